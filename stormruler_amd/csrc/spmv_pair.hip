// Operator apply, paired rows: format 3 (per-lane offset bytes) and formats 4 / 5 (common offsets as kernel arguments).
// Record layouts: the header of spmv.hip and the comments in spmv_device.hpp.
#include "spmv_device.hpp"

namespace storm {


// Format 3: one lane = rows (2p, 2p + 1), one wave = 128 rows.  See the header comment.
// HALO: columns >= n_rows are not read from x's tail but from the peer window (each value polled until its tag is
// this exchange's); the kernel's last block acknowledges the planes.
template <bool DOT, int W, bool HALO>
__global__ __launch_bounds__(kBlock) void spmv_pair_kernel(SellArgs A, Scal alpha_s, Scal beta_s,
                                                           const double *__restrict__ x, double *__restrict__ y,
                                                           const int *__restrict__ slice_list,
                                                           int64_t n_launch_slices, DotArgs dot, const int *done,
                                                           IpcRecvArgs H) {
  const int done_flag = done ? *done : 0;
  __shared__ double dict_sh[32];
  __shared__ int offs_sh[64];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bidx = (int)blockIdx.x;
  const int lb = A.xcd_group > 1 ? xcd_remap_grouped(bidx, gridDim.x, A.xcd_group)
                                 : (A.xcd_group == 1 ? xcd_remap(bidx, gridDim.x) : bidx);
  const int64_t sl = (int64_t)lb * (kBlock / kWave) + wave;
  const bool active = sl < n_launch_slices;  // wave-uniform
  const uint32_t slice = (uint32_t)(slice_list ? slice_list[active ? sl : 0] : (active ? sl : 0));
  const double alpha = ld_scal2(alpha_s), beta = ld_scal2(beta_s);
  const uint32_t last_row = (uint32_t)(A.n_rows - 1);
  const bool w_is_x = DOT && dot.w == x;
  const bool w_load = DOT && dot.w != nullptr && !w_is_x;
  const char *xb = reinterpret_cast<const char *>(x);
  const char *xg_base = xb - (size_t)kVecGuard * 8;  // start of the zero guard in front of x
  char *yb = reinterpret_cast<char *>(y);

  const uint32_t r0 = slice * (2 * kWave) + 2 * lane;  // row A; row B = r0 + 1
  const bool valid_a = active && r0 <= last_row, valid_b = active && r0 + 1 <= last_row;
  const uint32_t rc = r0 <= last_row ? r0 : (last_row & ~1u);  // pairs past the end re-read the last pair
  const char *rec = A.pack + (size_t)(A.rec_by_pos ? (uint32_t)(active ? sl : 0) : slice) * kPairRecBytes;
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const u64x2 vw = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(rec) + lane);
  const uint64_t jw = __builtin_nontemporal_load(reinterpret_cast<const uint64_t *>(rec + 2 * kWave * 8) + lane);
  const double2v xi = *reinterpret_cast<const double2v *>(xb + (size_t)(rc << 3));
  double2v yo = {0.0, 0.0}, wi = {0.0, 0.0};
  if (A.accumulate) yo = *reinterpret_cast<const double2v *>(yb + (size_t)(rc << 3));
  if (w_load) wi = *reinterpret_cast<const double2v *>(reinterpret_cast<const char *>(dot.w) + (size_t)(rc << 3));
  {
    // tables: <= 64 entries each, one load per lane (the allocations hold kDictSize entries).  One copy per
    // block at a fixed LDS address; every wave stores the same words before it reads them: no barrier.
    const int o0 = A.offs[lane];
    const double d0 = A.dict[lane & 31];
    offs_sh[lane] = o0;
    if (lane < 32) dict_sh[lane] = d0;
  }
  __builtin_amdgcn_wave_barrier();  // this wave's copy of the tables is complete (same-wave LDS order)
  double2v xg[W > 0 ? W : 1];
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const unsigned ob = (unsigned)(jw >> (8 * k)) & 0xffu;  // = offset index * 4
    const int off = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(offs_sh) + ob);
    // both rows' neighbour.  The index is biased by the guard so that it is never negative (the host checked
    // rc + off >= -kVecGuard): the address is a uniform base plus an UNSIGNED 32-bit byte offset.
    const int ca = (int)rc + off;  // column of row A's neighbour; row B's is ca + 1
    if (HALO && ca + 1 >= (int)A.n_rows) {
      // (an absent slot's column may point anywhere: beyond the halo rows it reads as 0, like x's zero padding)
      const int ha = ca - (int)A.n_rows, hb = ha + 1;
      double va, vb;
      ipc_halo_pair(H.w, H.rp, ha, hb, H.n_halo, &va, &vb);
      xg[k].x = ha < 0 ? x[ca] : va;
      xg[k].y = vb;
    } else {
      xg[k] = *reinterpret_cast<const double2v *>(xg_base + (size_t)((rc + (uint32_t)(off + kVecGuard)) << 3));
    }
  }
  double acc_a = 0.0, acc_b = 0.0;
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const unsigned ba = (unsigned)(vw.x >> (8 * (k + 1))) & 0xffu, bb = (unsigned)(vw.y >> (8 * (k + 1))) & 0xffu;
    acc_a += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ba) * (xg[k].x - xi.x);
    acc_b += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + bb) * (xg[k].y - xi.y);
  }
  const double ext_a = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)vw.x & 0xffu));
  const double ext_b = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)vw.y & 0xffu));
  double2v yi;
  yi.x = (A.accumulate ? yo.x : beta * xi.x) + alpha * (acc_a + ext_a * xi.x);
  yi.y = (A.accumulate ? yo.y : beta * xi.y) + alpha * (acc_b + ext_b * xi.y);
  if (!done_flag) {
    if (valid_b) __builtin_nontemporal_store(yi, reinterpret_cast<double2v *>(yb + (size_t)(rc << 3)));
    else if (valid_a) y[rc] = yi.x;  // the odd last row
  }
  if (HALO) ipc_halo_ack_last_block(H.w, H.rp);  // (every thread of every block gets here)
  if (done_flag) return;
  if (DOT) {
    yi.x = valid_a ? yi.x : 0.0;
    yi.y = valid_b ? yi.y : 0.0;
    double a = dot.w ? (w_is_x ? xi.x : wi.x) * yi.x + (w_is_x ? xi.y : wi.y) * yi.y : 0.0;
    double b = yi.x * yi.x + yi.y * yi.y;
    a = wave_sum_to_lane63(a);
    if (dot.yy) b = wave_sum_to_lane63(b);
    if (lane == kWave - 1) {
      const int slot = dot.block_offset + (int)blockIdx.x * (kBlock / kWave) + wave;
      dot.partials[slot] = a;
      if (dot.yy) dot.partials[dot.nblocks_total + slot] = b;
    }
  }
}

template <bool DOT, int K, int M1, int G>
__global__ __launch_bounds__(kBlock) void spmv_canon_kernel(SellArgs A, CanonArgs C, Scal alpha_s, Scal beta_s,
                                                            const double *__restrict__ x, double *__restrict__ y,
                                                            const int *__restrict__ slice_list,
                                                            int64_t n_launch_slices, DotArgs dot, const int *done) {
  const int done_flag = done ? *done : 0;
  __shared__ double dict_sh[32];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bidx = C.reverse ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
  int lb = bidx;
  if (C.xcd_shift >= 0) {  // xcd_remap_grouped for a power-of-two run length
    if (bidx < C.xcd_full) {
      const int xcd = bidx & (kNumXcd - 1), j = bidx >> 3;
      lb = ((((j >> C.xcd_shift) << 3) + xcd) << C.xcd_shift) + (j & ((1 << C.xcd_shift) - 1));
    }
  } else if (A.xcd_group > 1) {
    lb = xcd_remap_grouped(bidx, gridDim.x, A.xcd_group);
  } else if (A.xcd_group == 1) {
    lb = xcd_remap(bidx, gridDim.x);
  }
  const int64_t sl0 = ((int64_t)lb * (kBlock / kWave) + wave) * G;
  const double alpha = ld_scal2(alpha_s), beta = ld_scal2(beta_s);
  const uint32_t last_row = (uint32_t)(A.n_rows - 1);
  const bool w_is_x = DOT && dot.w == x;
  const bool w_load = DOT && dot.w != nullptr && !w_is_x;
  const char *xb = reinterpret_cast<const char *>(x);
  const char *xg_base = xb - (size_t)kVecGuard * 8;  // start of the zero guard in front of x
  char *yb = reinterpret_cast<char *>(y);
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

  // Issue order matters (loads return in order; a wait for one load waits for every earlier one): the tables first
  // -- their LDS copies are needed before anything else can be consumed -- then per group the record, the own rows,
  // all gathers and the two outer neighbours back to back; nothing is consumed before the last load is in flight.
  const double dict_word = A.dict[lane & 31];
  bool valid_a[G], valid_b[G];
  uint32_t rc[G];
  u64x2 vw[G];
  double2v xi[G], yo[G], wi[G], xg[G][K];
  double e[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const bool active = sl0 + g < n_launch_slices;  // wave-uniform
    const uint32_t slice = (uint32_t)(slice_list ? slice_list[active ? sl0 + g : 0] : (active ? sl0 + g : 0));
    const uint32_t r0 = slice * (2 * kWave) + 2 * lane;  // row A; row B = r0 + 1
    valid_a[g] = active && r0 <= last_row, valid_b[g] = active && r0 + 1 <= last_row;
    rc[g] = r0 <= last_row ? r0 : (last_row & ~1u);  // pairs past the end re-read the last pair
    vw[g] = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(A.pack + (size_t)slice * kCanonRecBytes) + lane);
    xi[g] = *reinterpret_cast<const double2v *>(xb + (size_t)(rc[g] << 3));
    yo[g] = double2v{0.0, 0.0}, wi[g] = double2v{0.0, 0.0};
    if (A.accumulate) yo[g] = *reinterpret_cast<const double2v *>(yb + (size_t)(rc[g] << 3));
    if (w_load) wi[g] = *reinterpret_cast<const double2v *>(reinterpret_cast<const char *>(dot.w) + (size_t)(rc[g] << 3));
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (M1 >= 0 && (k == M1 || k == M1 + 1)) continue;
      int t = (int)rc[g] + C.off[k] + kVecGuard;  // guard-relative, clamped: an absent neighbour may point anywhere
      t = t < 0 ? 0 : t;
      t = t > C.max_gather ? C.max_gather : t;
      // (32-bit byte offset from a uniform base: n_rows + n_halo < 2^28 is a condition of the paired formats)
#if defined(STORM_CANON_EXPERIMENT) && STORM_CANON_EXPERIMENT >= 2  // (measurement only: no gathers either)
      xg[g][k] = xi[g] + (double)t;
#else
      xg[g][k] = *reinterpret_cast<const double2v *>(xg_base + (size_t)((uint32_t)t << 3));
#endif
    }
    e[g] = 0.0;
    if (M1 >= 0 && (lane == 0 || lane == kWave - 1))  // x[rc - 1] of lane 0, x[rc + 2] of lane 63
      e[g] = *reinterpret_cast<const double *>(xg_base + (size_t)((rc[g] + (uint32_t)(kVecGuard + (lane == 0 ? -1 : 2))) << 3));
  }
  if (lane < 32) dict_sh[lane] = dict_word;  // one copy per block, every wave stores the same words: no barrier
  __builtin_amdgcn_wave_barrier();  // this wave's copy of the tables is complete (same-wave LDS order)
  double dot_a = 0.0, dot_b = 0.0;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (M1 >= 0) {
      // x[rc - 1] and x[rc + 2] are the neighbouring lanes' own rows
      const double left = dpp_shift<0x138>(xi[g].y);   // wave_shr:1 -- lane i receives lane i - 1
      const double right = dpp_shift<0x130>(xi[g].x);  // wave_shl:1 -- lane i receives lane i + 1
      xg[g][M1 >= 0 ? M1 : 0].x = lane == 0 ? e[g] : left;
      xg[g][M1 >= 0 ? M1 : 0].y = xi[g].x;
      xg[g][M1 >= 0 ? M1 + 1 : 0].x = xi[g].y;
      xg[g][M1 >= 0 ? M1 + 1 : 0].y = lane == kWave - 1 ? e[g] : right;
    }
    double acc_a = 0.0, acc_b = 0.0;
#ifdef STORM_CANON_EXPERIMENT  // (measurement only: the kernel's memory floor -- no table lookups, one add per neighbour)
#pragma unroll
    for (int k = 0; k < K; ++k) acc_a += xg[g][k].x, acc_b += xg[g][k].y;
    acc_a += __longlong_as_double((long long)vw[g].x), acc_b += __longlong_as_double((long long)vw[g].y);
#else
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const unsigned ba = (unsigned)(vw[g].x >> (8 * (k + 1))) & 0xffu, bb = (unsigned)(vw[g].y >> (8 * (k + 1))) & 0xffu;
      acc_a += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ba) * (xg[g][k].x - xi[g].x);
      acc_b += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + bb) * (xg[g][k].y - xi[g].y);
    }
#endif
    const double ext_a = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)vw[g].x & 0xffu));
    const double ext_b = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)vw[g].y & 0xffu));
    double2v yi;
    yi.x = (A.accumulate ? yo[g].x : beta * xi[g].x) + alpha * (acc_a + ext_a * xi[g].x);
    yi.y = (A.accumulate ? yo[g].y : beta * xi[g].y) + alpha * (acc_b + ext_b * xi[g].y);
    if (!done_flag) {
      double2v *yp = reinterpret_cast<double2v *>(yb + (size_t)(rc[g] << 3));
      if (valid_b[g]) *yp = yi;
      else if (valid_a[g]) y[rc[g]] = yi.x;  // the odd last row
    }
    if (DOT) {
      yi.x = valid_a[g] ? yi.x : 0.0;
      yi.y = valid_b[g] ? yi.y : 0.0;
      // (group by group, rows in order: with G == 1 exactly the sums of the one-group kernel)
      const double a = dot.w ? (w_is_x ? xi[g].x : wi[g].x) * yi.x + (w_is_x ? xi[g].y : wi[g].y) * yi.y : 0.0;
      const double b = yi.x * yi.x + yi.y * yi.y;
      dot_a = g == 0 ? a : dot_a + a;
      dot_b = g == 0 ? b : dot_b + b;
    }
  }
  if (done_flag) return;
  if (DOT) {
    dot_a = wave_sum_to_lane63(dot_a);
    if (dot.yy) dot_b = wave_sum_to_lane63(dot_b);
    if (dot.tickets == nullptr) {
      if (lane == kWave - 1) {
        const int slot = dot.block_offset + bidx * (kBlock / kWave) + wave;
        dot.partials[slot] = dot_a;
        if (dot.yy) dot.partials[dot.nblocks_total + slot] = dot_b;
      }
    } else {  // the reduction finishes here: block partial, then two levels of tickets
      __shared__ double wave_part[2 * (kBlock / kWave)];
      if (lane == kWave - 1) wave_part[wave] = dot_a, wave_part[kBlock / kWave + wave] = dot.yy ? dot_b : 0.0;
      __syncthreads();
      if (wave != 0) return;
      const double mine[2] = {(wave_part[0] + wave_part[1]) + (wave_part[2] + wave_part[3]),
                              (wave_part[4] + wave_part[5]) + (wave_part[6] + wave_part[7])};
      double total[2];
      const TicketArgs t{dot.tickets, dot.partials, dot.part2};
      if (ticket_reduce_wave0<2>(t, mine, dot.yy ? 2 : 1, (unsigned)bidx, gridDim.x, total) && lane == 0) {
        *dot.out0 = total[0];
        if (dot.yy) *dot.out1 = total[1];
      }
    }
  }
}

template <bool DOT>
static void launch_canon(const RangeLaunch &L, SellArgs A) {
  const storm_hip_op *op = L.op;
  const int nb = L.nb, group = A.xcd_group;
  hipStream_t st = op->ctx->stream;
  const Scal alpha = L.alpha, beta = L.beta;
  const double *x = L.x;
  double *y = L.y;
  const int *slice_list = L.slice_list;
  const int64_t n_launch = L.n_launch;
  const DotArgs dot = L.dot;
  const int *done = L.done;
  hipEvent_t ev0 = L.ev0, ev1 = L.ev1;
  CanonArgs C;
  for (int k = 0; k < 7; ++k) C.off[k] = op->canon_off[k];
  C.max_gather = (int)(op->n_rows + op->n_halo) + kVecGuard + 2;
  C.reverse = op->ctx->spmv_reverse;
  C.xcd_shift = -1, C.xcd_full = 0;
  if (group > 1 && (group & (group - 1)) == 0) {
    while ((1 << (C.xcd_shift + 1)) <= group) ++C.xcd_shift;
    const int span = kNumXcd * group;
    C.xcd_full = (nb / span) * span;
  }
#define CANON_GO2(K_, M1_, G_)                                                                                       \
  hipExtLaunchKernelGGL((spmv_canon_kernel<DOT, K_, M1_, G_>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A, C, alpha, \
                      beta, x, y, slice_list, n_launch, dot, done)
#define CANON_GO(K_, M1_)                                      \
  do {                                                         \
  const bool two = canon_groups(op) == 2;                    \
  if (two) CANON_GO2(K_, M1_, 2);                            \
  else CANON_GO2(K_, M1_, 1);                                \
  } while (0)
  if (op->canon_k == 6) CANON_GO(6, 2);
  else if (op->canon_k == 4) CANON_GO(4, 1);
  else CANON_GO(2, 0);
#undef CANON_GO
#undef CANON_GO2
}

bool spmv_canon_run(const RangeLaunch &L) {
  if (L.op->pair < 2 || boundary_of_mixed(L)) return false;
  int width = 0;
  const SellArgs A = paired_args(L, &width);
  if (L.want_dot) launch_canon<true>(L, A);
  else launch_canon<false>(L, A);
  return true;
}

template <bool DOT>
static void launch_pair(const RangeLaunch &L, SellArgs A, int width) {
  const storm_hip_op *op = L.op;
  const int nb = L.nb;
  hipStream_t st = op->ctx->stream;
  const Scal alpha = L.alpha, beta = L.beta;
  const double *x = L.x;
  double *y = L.y;
  const int *slice_list = L.slice_list;
  const int64_t n_launch = L.n_launch;
  const DotArgs dot = L.dot;
  const int *done = L.done;
  hipEvent_t ev0 = L.ev0, ev1 = L.ev1;
  const IpcFused *fused = L.fused;
  IpcRecvArgs H{};
  const bool halo_reads = fused != nullptr && slice_list != nullptr && slice_list == op->d_boundary;
  if (halo_reads) H.w = fused->w, H.rp = fused->rp, H.n_halo = (int)op->n_halo;
#define PAIR_GO(W_)                                                                                                         \
  do {                                                                                                                      \
    if (halo_reads)                                                                                                         \
      hipExtLaunchKernelGGL((spmv_pair_kernel<DOT, W_, true>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A, alpha, beta, \
                            x, y, slice_list, n_launch, dot, done, H);                                                      \
    else                                                                                                                    \
      hipExtLaunchKernelGGL((spmv_pair_kernel<DOT, W_, false>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A, alpha, beta, \
                            x, y, slice_list, n_launch, dot, done, H);                                                      \
  } while (0)
  switch (width) {
    case 1: PAIR_GO(1); break;
    case 2: PAIR_GO(2); break;
    case 3: PAIR_GO(3); break;
    case 4: PAIR_GO(4); break;
    case 5: PAIR_GO(5); break;
    case 6: PAIR_GO(6); break;
    default: PAIR_GO(7); break;
  }
#undef PAIR_GO
}

bool spmv_pair_run(const RangeLaunch &L) {
  if (L.op->pair == 0) return false;
  int width = 0;
  const SellArgs A = paired_args(L, &width);
  if (L.want_dot) launch_pair<true>(L, A, width);
  else launch_pair<false>(L, A, width);
  return true;
}

}  // namespace storm
