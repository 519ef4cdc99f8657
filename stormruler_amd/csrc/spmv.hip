// The operator-apply hot path: stormDivGrad (source_apps/playground/Playground.cpp:115-131)
// as a gather-form SpMV on a sliced-ELL / CSR-tail hybrid.
//
// Layout in HBM (built once per operator on the host, slot order = face order):
//   slice s = rows [64 s, 64 s + 64): one wavefront.  Everything the slice streams is ONE
//   contiguous record at byte offset slice_off[s]:
//        [ ext : 64 x f64 ][ col : W x 64 x i32 ][ val : W x 64 x f64 ]      (512 + 768 W bytes)
//   Inside col / val the slots are stored in PAIRS, lane-major: lane l finds (slot 2p, slot 2p+1)
//   of its row as one int2 / one double2, so a wave reads 512 B of columns and 1 KiB of weights
//   per instruction (8- and 16-byte accesses per lane; an odd last slot follows column-major).  Packing ext/col/val of a slice
//   into one record leaves the kernel three DRAM streams (records, x, y) instead of five;
//   on MI355X the number of concurrent streams, not L2 locality, decided the rate (measured:
//   profiles/r01_notes.md).
//   W = min(longest row of the slice, ell_cap); what does not fit goes to a CSR tail handled
//   by a wave-per-row kernel with a __shfl_down reduction.  Padding slots carry weight 0 and
//   the row's own index as column.
// Value-dictionary records (VARIANT 2, chosen at build time when it is lossless): when ext and the
//   weights of the whole operator take at most 256 distinct fp64 values and no slice is wider than
//   7 slots -- any mesh made of a few repeated cell shapes, the 256^3 box of the benchmark being the
//   extreme case -- a slice stores, instead of 8 (W + 1) bytes of fp64 per row, ONE 8-byte word per
//   row holding W + 1 byte indices into a dictionary of the distinct values (bit patterns, so the
//   arithmetic is unchanged):
//        [ idx : 64 x u64 (byte 0 = ext, byte k+1 = slot k) ][ col : W x 64 x i32 ]   (512 + 256 W bytes)
//   The kernel keeps the dictionary in LDS (2 KiB per block).  At W = 6 a row then streams
//   24 + 8 + 8 + 8 = 48 bytes instead of 96.  Operators that do not qualify keep the fp64 records.
// Value + offset dictionaries (format 2): if, in addition, the column offsets col - row take at most
//   256 distinct values (a mesh numbered block-wise: the box has 7 -- 0, +-1, +-nx, +-nx ny -- a z-slab of
//   it with halo planes a few more), the columns become byte indices as well and a row is ONE 16-byte word
//        byte 0 = ext, bytes 1..7 = weight of slot 0..6, bytes 8..14 = offset of slot 0..6
//   so a slice is a 1 KiB record whatever its width, and a row streams 16 + 8 + 8 = 32 bytes.
// Paired rows (format 3, the default when it applies): with a third of the bytes the format-2 kernel is no
//   longer HBM-bound but bound by the number of vector-memory instructions a row costs (one 8-byte gather per
//   neighbour through the texture-address path; halving that count in a diagnostic build took 16 % off the
//   kernel).  Format 3 gives each LANE two consecutive rows (2p, 2p+1) whose neighbour lists are merged into
//   one list of column offsets (the shortest common supersequence of the two rows' offset lists, <= 7 long;
//   a row that lacks an offset of the merged list gets weight 0 there, which leaves its sum unchanged): one
//   16-byte load then fetches x[2p + off], x[2p + 1 + off] -- the neighbour of BOTH rows -- and x_i, y_i move
//   as 16-byte pairs too.  Record of a 128-row group: [64 x (weights of row 2p : u64, of row 2p+1 : u64)]
//   [64 x offsets : u64], index bytes pre-scaled (value index * 8, offset index * 4) so a bit-field extract is
//   the LDS byte address; 12 + 8 + 8 = 28 bytes per row and 10 instead of 18 vector-memory instructions per
//   row pair.  Needs <= 32 distinct values, <= 64 distinct offsets, < 2^28 columns, and every 16-byte gather in
//   bounds of [-kVecGuard, n + 3] (vectors carry a zero guard in front and zero padding behind).  A weight-0
//   slot still reads its partner's neighbour, so it contributes 0 * (x_nb - x_i): exact for finite x; a
//   non-finite x_nb reaches one more row than the face loop would carry it to.
// Arithmetic:  y_i = beta x_i + alpha ( sum_k w_ik (x[col_ik] - x_i) + ext_i x_i )
//   -- the difference form of the reference's flux  (c[out] - c[in]), which keeps the
//   cancellation behaviour of the face loop (no large diagonal * x_i term).
//
// Translation units (NOTES.md, "Source layout and dispatch of the apply", has the table): spmv_device.hpp -- argument structs, helpers, the launch
// interface; spmv_sell.hip, spmv_dict.hip, spmv_pair.hip, spmv_lattice.hip -- one per kernel family, each exporting
// launchers that take a RangeLaunch; spmv_build.hip -- the host-side build and the create entry points; this file --
// the dispatch (which launches make up an apply, which format takes each), the diagonal and the apply entry points.
#include <algorithm>

#include "spmv_device.hpp"

namespace storm {


// fused (peer-window transport): the interior launch carries the send, the boundary launch reads the window.
static int interior_blocks(const storm_hip_op *op, bool accumulate, int n_send_blocks) {
  CanonTileArgs T;
  int nbt = 0;
  if (!accumulate && canon_tile_geometry(op, &T, &nbt, true)) return nbt + n_send_blocks;
  return -1;
}

// The dispatch table: a launch goes to the first format that takes it.
struct SpmvFormat {
  const char *name;
  bool (*run)(const RangeLaunch &L);
};
static const SpmvFormat kSpmvFormats[] = {
    {"format 4 on a lattice: tiles of 1024 rows x 2|4 planes (spmv_lattice.hip)", spmv_tile_run},
    {"formats 4, 5: paired rows, common offsets (spmv_pair.hip)", spmv_canon_run},
    {"format 3: paired rows, per-lane offsets (spmv_pair.hip)", spmv_pair_run},
    {"formats 1, 2 of one width: value (+ offset) dictionaries (spmv_dict.hip)", spmv_dict_run},
    {"format 0 and mixed-width dictionary records: sliced ELL (spmv_sell.hip)", spmv_sell_run},
};

static int launch_range(const storm_hip_op *op, Scal alpha, Scal beta, const double *x, double *y,
                        const int *slice_list, int64_t n_launch, DotArgs dot, bool want_dot,
                        const int *done, bool accumulate, const IpcFused *fused = nullptr, const CgFuseArgs *cg_fuse = nullptr) {
  if (n_launch <= 0) return STORM_HIP_OK;
  storm_hip_ctx *c = op->ctx;
  const bool prof = c->opt_profile_spmv != 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (prof) {
    while (c->prof_events.size() < c->prof_used + 2) {
      hipEvent_t ev;
      HIP_TRY(hipEventCreate(&ev));
      c->prof_events.push_back(ev);
    }
    // the events are attached to the kernel dispatch itself (hipExtLaunchKernelGGL): they are
    // stamped at the kernel's begin and end, the quantity rocprofv3's kernel trace reports
    ev0 = c->prof_events[c->prof_used], ev1 = c->prof_events[c->prof_used + 1];
  }
  int nb = blocks_for(op, n_launch, op->d_bnd_pack != nullptr && slice_list != nullptr && slice_list == op->d_boundary);
  if (slice_list == nullptr && !accumulate && n_launch == op->n_slices) {
    CanonTileArgs T;
    int nbt = 0;
    if (canon_tile_geometry(op, &T, &nbt)) nb = nbt;  // the tiled format-4 kernel (spmv_tile_run takes it on the same test)
  } else if (slice_list != nullptr && slice_list == op->d_interior) {
    const int nbi = interior_blocks(op, accumulate, fused ? fused->sp.n_blocks : 0);
    if (nbi >= 0) nb = nbi;  // ... over the interior planes of a partitioned operator, plus the sending blocks
    else if (fused != nullptr) STORM_TRY(comm_ipc_send(op, x, fused->w, fused->sp));  // a kernel that cannot send: stand-alone
  }
  RangeLaunch L{op, nb, alpha, beta, x, y, slice_list, n_launch, dot, want_dot, done, ev0, ev1, accumulate, fused,
                want_dot ? cg_fuse : nullptr};
  bool taken = false;
  for (const SpmvFormat &f : kSpmvFormats)
    if ((taken = f.run(L))) break;
  STORM_REQUIRE(taken, "spmv: no kernel takes this operator");
  HIP_TRY(hipGetLastError());
  if (prof) c->prof_used += 2;
  return STORM_HIP_OK;
}


// The fused CG step (CgFuseArgs) applies to an operator whose unsplit apply runs the tiled format-4 kernel.
bool spmv_can_fuse_cg(const storm_hip_op *op) {
  CanonTileArgs T;
  int nb = 0;
  if (op->halo.n_nbrs == 0 && op->d_bnd_pack == nullptr && op->tail_rows == 0 && canon_tile_geometry(op, &T, &nb)) return true;
  // ... or a partitioned (mixed) operator on the peer-window transport, whose boundary launch reads the window itself
  MarchArgs M;
  // ... or on RCCL: the boundary planes of p' are packed by a small kernel and travel on the comm stream under the march
  return op->halo.n_nbrs > 0 && op->d_bnd_pack != nullptr && op->tail_rows == 0 &&
         ((comm_is_ipc(op->ctx) && op->ctx->opt_ipc_fused != 0) || (comm_is_rccl(op->ctx) && op->ctx->opt_rccl_fused != 0)) &&
         op->n_boundary > 0 && cg_march_geometry(op, &M, &nb, true);
}

bool spmv_can_march(const storm_hip_op *op) {
  MarchArgs M;
  int nb = 0;
  return op->halo.n_nbrs == 0 && op->d_bnd_pack == nullptr && op->tail_rows == 0 && cg_march_geometry(op, &M, &nb);
}

int spmv_grid_blocks(const storm_hip_op *op) {
  CanonTileArgs T;
  int nb = 0;
  if (canon_tile_geometry(op, &T, &nb)) return nb;  // (an unsplit launch of a format-4 lattice operator)
  return blocks_for(op, op->n_slices);
}

int spmv_launch(const storm_hip_op *op, Scal alpha, Scal beta, const double *x, double *y,
                const SpmvDot *sd, const int *done, bool accumulate) {
  storm_hip_ctx *c = op->ctx;
  const bool fuse_dot = sd != nullptr && op->tail_rows == 0;
  STORM_REQUIRE(op->halo.n_nbrs == 0 || c->comm != nullptr,
                "operator has a halo plan but the context has no communicator (call storm_hip_ctx_comm_init)");
  // (a mixed operator -- format 4 inside, format 3 where rows read halo columns -- always runs as its two lists)
  const bool exchange = op->halo.n_nbrs > 0;
  const bool split = exchange || op->d_bnd_pack != nullptr;
  DotArgs dot{nullptr, nullptr, 0, 0, 0};
  // peer-window transport + paired records: the fused form of the exchange (IpcFused)
  const bool fuse_x = exchange && comm_is_ipc(c) && op->pair != 0 && c->opt_ipc_fused != 0 && op->n_boundary > 0;
  IpcFused fx;
  if (fuse_x) STORM_TRY(comm_ipc_exchange(op, &fx.w, &fx.sp, &fx.rp));
  const int nbi_tile = split ? interior_blocks(op, accumulate, 0) : -1;  // (send blocks carry no partials)
  const int nb_int = split ? (nbi_tile >= 0 ? nbi_tile : blocks_for(op, op->n_interior)) : spmv_grid_blocks(op);
  const int nb_bnd = split ? blocks_for(op, op->n_boundary, op->d_bnd_pack != nullptr) : 0;
  const int nb_total = nb_int + nb_bnd;
  if (fuse_dot) {
    STORM_REQUIRE(8 * (int64_t)nb_total <= c->partials_capacity,
                  "spmv: %d blocks exceed the partials workspace", nb_total);
    dot = DotArgs{sd->w, sd->partials, sd->yy ? 1 : 0, 4 * nb_total, 0};  // one partial per wave
  }
  if (sd && sd->nblocks_out) *sd->nblocks_out = fuse_dot ? 4 * nb_total : 0;
  if (sd && sd->ticketed_out) *sd->ticketed_out = 0;
  if (fuse_dot && !split && op->pair >= 2 && sd->out[0] != nullptr && c->opt_ticket_reduce != 0 && c->comm == nullptr &&
      nb_total <= kTicketGroup * kTicketMaxGroups && 2 * nb_total <= c->partials_capacity &&
      (!sd->yy || sd->out[1] != nullptr)) {
    dot.tickets = c->d_tickets, dot.part2 = c->d_ticket_sums, dot.out0 = sd->out[0], dot.out1 = sd->out[1];
    dot.nblocks_total = nb_total;
    if (sd->ticketed_out) *sd->ticketed_out = 1;
  }

  if (!split) {
    CgFuseArgs cgf{};
    const bool cg_fused = sd != nullptr && sd->cg.r != nullptr;
    if (cg_fused) {
      STORM_REQUIRE(spmv_can_fuse_cg(op) && fuse_dot && !accumulate && sd->w == x,
                    "spmv: the fused CG step needs the tiled format-4 kernel");
      STORM_REQUIRE(sd->cg.x != nullptr, "spmv: the fused CG step updates x");
      dot.tickets = nullptr, dot.nblocks_total = 4 * nb_total;  // (the tiled form of the step leaves per-wave partials)
      if (sd->ticketed_out) *sd->ticketed_out = 0;
      cgf = CgFuseArgs{sd->cg.iteration, sd->cg.my_iteration, sd->cg.ca, sd->cg.cb, sd->cg.x, sd->cg.r, sd->cg.p_out, sd->cg.ca_imm, sd->cg.cb_imm};
    }
    MarchArgs M;
    int nb_march = 0;
    if (cg_fused && cg_march_geometry(op, &M, &nb_march)) {
      // the z-marching step kernel: its own grid, its own (fewer) partial slots -- or, where the caller takes the sum
      // from the slab (sd->out[0]), the reduction finished in the kernel by tickets
      dot.nblocks_total = 4 * nb_march;
      if (sd->nblocks_out) *sd->nblocks_out = 4 * nb_march;
      if (sd->out[0] != nullptr && c->opt_ticket_reduce != 0 && c->comm == nullptr && (!sd->yy || sd->out[1] != nullptr) &&
          nb_march <= kTicketGroup * kTicketMaxGroups && 2 * (int64_t)nb_march <= c->partials_capacity) {
        dot.tickets = c->d_tickets, dot.part2 = c->d_ticket_sums, dot.out0 = sd->out[0], dot.out1 = sd->yy ? sd->out[1] : nullptr;
        dot.nblocks_total = nb_march;
        if (sd->ticketed_out) *sd->ticketed_out = 1;
      }
      return spmv_march_run(op, M, nb_march, alpha, beta, x, y, dot, done, cgf, IpcSendArgs{});
    }
    STORM_REQUIRE(!cg_fused || (cgf.ca != nullptr && cgf.cb != nullptr && cgf.iteration != nullptr),
                  "spmv: a fused CG step with immediate coefficients needs the marching kernel");
    STORM_TRY(launch_range(op, alpha, beta, x, y, nullptr, op->n_slices, dot, fuse_dot, done, accumulate, nullptr,
                           cg_fused ? &cgf : nullptr));
  } else {
    const bool cg_fused_part = sd != nullptr && sd->cg.r != nullptr;
    if (cg_fused_part) {
      // The fused CG step on a partitioned operator (peer-window transport): ONE marching launch updates x and forms
      // p' on every owned plane, applies the operator to the interior planes and -- its first blocks -- sends p' of the
      // boundary rows; the boundary launch then reads p' (owned columns) and the window (halo columns).
      MarchArgs M;
      int nb_march = 0;
      const bool over_rccl = !fuse_x && comm_is_rccl(c);
      STORM_REQUIRE((fuse_x || over_rccl) && fuse_dot && !accumulate && sd->w == x && cg_march_geometry(op, &M, &nb_march, true),
                    "spmv: the fused CG step on a partitioned operator needs the peer-window or the RCCL transport and a lattice");
      // RCCL: p' = r + beta p of the rows to send is formed by a small kernel on the comm stream and travels while the march
      // below runs; the boundary launch waits for the planes (they land in p_out's halo tail)
      if (over_rccl) STORM_TRY(comm_halo_exchange_begin_direction(op, x, sd->cg.r, sd->cg.cb, sd->cg.p_out));
      const CgFuseArgs cgf{sd->cg.iteration, sd->cg.my_iteration, sd->cg.ca, sd->cg.cb, sd->cg.x, sd->cg.r, sd->cg.p_out, 0.0, 0.0};
      const int nb_b = blocks_for(op, op->n_boundary, true);
      STORM_REQUIRE(8 * (int64_t)(nb_march + nb_b) <= c->partials_capacity, "spmv: %d blocks exceed the partials workspace", nb_march + nb_b);
      dot = DotArgs{sd->cg.p_out, sd->partials, sd->yy ? 1 : 0, 4 * (nb_march + nb_b), 0};
      if (sd->nblocks_out) *sd->nblocks_out = 4 * (nb_march + nb_b);
      IpcSendArgs S{};
      if (fuse_x) S = IpcSendArgs{fx.w, fx.sp};  // (RCCL: no sending blocks in the march)
      // (the march kernel's own dots use w = p' from its registers; DotArgs::w only has to be non-null there)
      STORM_TRY(spmv_march_run(op, M, nb_march, alpha, beta, x, y, dot, done, cgf, S));
      dot.block_offset = 4 * nb_march;
      // the boundary groups: z = A p' from p_out and the window (RCCL: p_out's halo tail); <p', z> partials behind the march's
      if (over_rccl) STORM_TRY(comm_halo_exchange_end(op));
      STORM_TRY(launch_range(op, alpha, beta, sd->cg.p_out, y, op->d_boundary, op->n_boundary, dot, fuse_dot, done, accumulate,
                             fuse_x ? &fx : nullptr));
      return STORM_HIP_OK;
    }
    // interior rows overlap the halo exchange running on the comm stream
    if (exchange && !fuse_x) STORM_TRY(comm_halo_exchange_begin(op, const_cast<double *>(x)));
    if (fuse_x && op->n_interior == 0) STORM_TRY(comm_ipc_send(op, x, fx.w, fx.sp));
    STORM_TRY(launch_range(op, alpha, beta, x, y, op->d_interior, op->n_interior, dot, fuse_dot, done, accumulate,
                           fuse_x ? &fx : nullptr));
    if (exchange && !fuse_x) STORM_TRY(comm_halo_exchange_end(op));
    dot.block_offset = 4 * nb_int;
    STORM_TRY(launch_range(op, alpha, beta, x, y, op->d_boundary, op->n_boundary, dot, fuse_dot, done, accumulate,
                           fuse_x ? &fx : nullptr));
  }
  if (op->tail_rows > 0) {
    STORM_TRY(spmv_tail_run(op, alpha, x, y, done));
  }
  return STORM_HIP_OK;
}


// ---- diagonal of beta*I + alpha*M (for a Jacobi preconditioner) -----------------------------------
// In the difference form  (Mx)_i = sum_k w_ik (x_col - x_i) + ext_i x_i  the coefficient of x_i is
// ext_i - sum_k w_ik (no slot has col == i: build_op receives off-diagonal entries only, and padding
// slots carry w = 0).  One lane per row, same slot addressing as build_op; not a hot kernel.
__global__ __launch_bounds__(kBlock) void diag_sell_kernel(const char *__restrict__ pack,
                                                           const int64_t *__restrict__ slice_off, int64_t n_rows,
                                                           const double *__restrict__ dict, int fmt2, double alpha,
                                                           double beta, double *__restrict__ d) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t s = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  const int64_t r = s * kWave + lane;
  if (r >= n_rows) return;
  if (fmt2 >= 3) {  // paired rows: 128-row groups, weights of row r in word (r % 128), bytes pre-scaled by 8
    const uint64_t iw = reinterpret_cast<const uint64_t *>(pack + (r >> 7) * (fmt2 == 4 ? kCanonRecBytes : kPairRecBytes))[r & 127];
    double sum = 0.0;
    for (int k = 0; k < 7; ++k) sum += dict[((unsigned)(iw >> (8 * (k + 1))) & 0xffu) >> 3];
    d[r] = beta + alpha * (dict[((unsigned)iw & 0xffu) >> 3] - sum);
    return;
  }
  const char *rec = pack + slice_off[s];
  if (dict) {  // value-dictionary record (format 2: 16-byte words, weights in bytes 1..7 of the first half)
    const int w = fmt2 ? 7 : (int)((slice_off[s + 1] - slice_off[s] - kExtBytes) / kColSlotBytes);
    const uint64_t iw = reinterpret_cast<const uint64_t *>(rec)[fmt2 ? 2 * lane : lane];
    double sum = 0.0;
    for (int k = 0; k < w; ++k) sum += dict[(unsigned)(iw >> (8 * (k + 1))) & 0xffu];
    d[r] = beta + alpha * (dict[(unsigned)iw & 0xffu] - sum);
    return;
  }
  const int w = (int)((slice_off[s + 1] - slice_off[s] - kExtBytes) / kSlotBytes);
  const double *val = reinterpret_cast<const double *>(rec + kExtBytes + (int64_t)w * (kWave * 4));
  const int np2 = w >> 1;
  double sum = 0.0;
  for (int k = 0; k < w; ++k) {
    const int at = (k < 2 * np2) ? ((k >> 1) * kWave + lane) * 2 + (k & 1) : np2 * 2 * kWave + lane;
    sum += val[at];
  }
  d[r] = beta + alpha * (reinterpret_cast<const double *>(rec)[lane] - sum);
}

__global__ void diag_tail_kernel(int64_t n_tail, const int *__restrict__ tail_row,
                                 const int64_t *__restrict__ tail_ptr, const double *__restrict__ tail_val,
                                 double alpha, double *__restrict__ d) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tail) return;
  double sum = 0.0;
  for (int64_t k = tail_ptr[t]; k < tail_ptr[t + 1]; ++k) sum += tail_val[k];
  d[tail_row[t]] -= alpha * sum;  // each overflowing row appears once in the tail
}

__global__ void safe_invert_kernel(int64_t n, double *__restrict__ d) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = d[i] == 0.0 ? 0.0 : 1.0 / d[i];  // safe_inverse, Crow/MathUtils.hpp:54-58
}


}  // namespace storm

using namespace storm;

extern "C" {

int storm_hip_op_apply(const storm_hip_op *op, double alpha, double beta, const storm_hip_vec *x,
                       storm_hip_vec *y) {
  STORM_REQUIRE(op && x && y, "op_apply: null argument");
  STORM_REQUIRE(x->ctx == op->ctx && y->ctx == op->ctx, "op_apply: context mismatch");
  STORM_REQUIRE(x != y && x->d != y->d, "op_apply: x and y must not alias");
  STORM_REQUIRE(x->n_owned == op->n_rows && y->n_owned == op->n_rows, "op_apply: operator has %lld rows, x %lld, y %lld",
                (long long)op->n_rows, (long long)x->n_owned, (long long)y->n_owned);
  STORM_REQUIRE(x->n_halo >= op->n_halo, "op_apply: x has %lld halo rows, operator needs %lld", (long long)x->n_halo,
                (long long)op->n_halo);
  if (op->ctx->opt_lazy != 0 && op->ctx->callback_depth == 0 && op->ctx->api_done == nullptr)
    return lazy_push_apply(op, alpha, beta, x->d, y->d);  // (option lazy_statements: lazy.hip)
  STORM_TRY(lazy_sync(op->ctx));
  return spmv_launch(op, host_scal(alpha), host_scal(beta), x->d, y->d, nullptr, op->ctx->api_done);
}

int storm_hip_op_apply_add(const storm_hip_op *op, double alpha, const storm_hip_vec *x, storm_hip_vec *y) {
  STORM_REQUIRE(op && x && y, "op_apply_add: null argument");
  STORM_REQUIRE(x->ctx == op->ctx && y->ctx == op->ctx, "op_apply_add: context mismatch");
  STORM_REQUIRE(x != y && x->d != y->d, "op_apply_add: x and y must not alias");
  STORM_REQUIRE(x->n_owned == op->n_rows && y->n_owned == op->n_rows, "op_apply_add: operator has %lld rows, x %lld, y %lld",
                (long long)op->n_rows, (long long)x->n_owned, (long long)y->n_owned);
  STORM_REQUIRE(x->n_halo >= op->n_halo, "op_apply_add: x has %lld halo rows, operator needs %lld", (long long)x->n_halo,
                (long long)op->n_halo);
  STORM_TRY(lazy_sync(op->ctx));
  return spmv_launch(op, host_scal(alpha), host_scal(0.0), x->d, y->d, nullptr, op->ctx->api_done, true);
}

int storm_hip_op_get_diagonal(const storm_hip_op *op, double alpha, double beta, int invert, storm_hip_vec *d) {
  STORM_REQUIRE(op && d, "op_get_diagonal: null argument");
  STORM_REQUIRE(d->ctx == op->ctx, "op_get_diagonal: context mismatch");
  STORM_REQUIRE(d->n_owned == op->n_rows, "op_get_diagonal: operator has %lld rows, d %lld", (long long)op->n_rows,
                (long long)d->n_owned);
  if (op->n_rows == 0) return STORM_HIP_OK;
  storm_hip_ctx *c = op->ctx;
  HIP_TRY(hipSetDevice(c->device));
  STORM_TRY(lazy_sync(c));
  const int64_t n64 = (op->n_rows + kWave - 1) / kWave;  // the kernel walks 64-row groups whatever the format
  const int nb = (int)((n64 + (kBlock / kWave) - 1) / (kBlock / kWave));
  hipLaunchKernelGGL(diag_sell_kernel, dim3(nb), dim3(kBlock), 0, c->stream, op->d_pack, op->d_slice_off, op->n_rows,
                     op->d_dict, op->pair >= 2 ? op->pair + 2 : op->pair ? 3 : (int)(op->offs_size > 0), alpha, beta, d->d);
  if (op->tail_rows > 0)
    hipLaunchKernelGGL(diag_tail_kernel, dim3((int)((op->tail_rows + 255) / 256)), dim3(256), 0, c->stream,
                       op->tail_rows, op->d_tail_row, op->d_tail_ptr, op->d_tail_val, alpha, d->d);
  if (invert)
    hipLaunchKernelGGL(safe_invert_kernel, dim3((int)((op->n_rows + 255) / 256)), dim3(256), 0, c->stream, op->n_rows,
                       d->d);
  HIP_TRY(hipGetLastError());
  return STORM_HIP_OK;
}

int storm_hip_op_get_stats(const storm_hip_op *op, storm_hip_op_stats *s) {
  STORM_REQUIRE(op && s, "op_get_stats: null argument");
  s->n_rows = op->n_rows;
  s->n_cols = op->n_rows + op->n_halo;
  s->nnz_offdiag = op->nnz;
  s->ell_slots = op->ell_slots;
  s->tail_nnz = op->tail_nnz;
  s->tail_rows = op->tail_rows;
  s->n_slices = op->n_slices;
  s->max_row_len = op->max_row_len;
  s->n_interior_slices = op->n_interior_slices;
  s->device_bytes = op->device_bytes;
  s->record_bytes = op->pack_bytes;
  s->value_dictionary_size = op->dict_size;
  s->offset_dictionary_size = op->offs_size;
  s->paired_rows = op->pair;
  CanonTileArgs T;
  int nbt = 0;
  s->tiled_planes = canon_tile_geometry(op, &T, &nbt, op->d_bnd_pack != nullptr) ? canon_tile_planes(op) : 0;  // (mixed operator: its interior planes)
  s->spmv_blocks = spmv_grid_blocks(op);
  s->xcd_run_blocks = op->xcd_group_sell;
  return STORM_HIP_OK;
}

int storm_hip_op_destroy(storm_hip_op *op) {
  if (!op) return STORM_HIP_OK;
  if (op->ctx) (void)storm_hip_ctx_sync(op->ctx);
  (void)hipFree(op->d_interior);
  (void)hipFree(op->d_boundary);
  (void)hipFree(op->d_slice_off);
  (void)hipFree(op->d_pack);
  (void)hipFree(op->d_bnd_pack);
  (void)hipFree(op->d_dict);
  (void)hipFree(op->d_offs);
  (void)hipFree(op->d_tail_row);
  (void)hipFree(op->d_tail_ptr);
  (void)hipFree(op->d_tail_col);
  (void)hipFree(op->d_tail_val);
  (void)hipFree(op->d_lat_pack);
  (void)hipFree(op->d_lat_off);
  (void)hipFree(op->halo.d_send_idx);
  (void)hipFree(op->halo.d_sendbuf);
  delete op;
  return STORM_HIP_OK;
}

}  // extern "C"
