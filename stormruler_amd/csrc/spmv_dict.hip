// Operator apply, value-dictionary records of one uniform width (formats 1 and 2): SPW slices per wavefront.
// Record layout: the header of spmv.hip.
#include "spmv_device.hpp"

namespace storm {


// Value-dictionary records of one uniform width W: SPW consecutive slices per wavefront.
// Compared with the general kernel above: (i) every wave keeps its own copy of the dictionary in LDS
// (written and read by the same wave: no block barrier between the record loads and the lookups);
// (ii) all record loads of the wave's SPW slices are issued first, then all SPW x W gathers, then
// the lookups and FMAs -- with half the bytes per row the kernel is bound by memory-level
// parallelism per wave rather than by HBM, and a wave with one slice had too little in flight;
// (iii) one fused-dot partial per wave covers SPW slices.
// FMT 1: [idx u64][columns i32] records; FMT 2: one 16-byte word per row, columns = row + offs[byte].
// Branch-free but for the store: rows past the end (ragged last slice) and slices past the launch are
// redirected to valid memory (row n-1, slice 0) and masked at the store / in the partials.
template <bool DOT, int W, int SPW, int FMT>
__global__ __launch_bounds__(kBlock) void spmv_dict_kernel(SellArgs A, Scal alpha_s, Scal beta_s,
                                                           const double *__restrict__ x, double *__restrict__ y,
                                                           const int *__restrict__ slice_list,
                                                           int64_t n_launch_slices, DotArgs dot, const int *done) {
  constexpr int NP = W / 2;
  constexpr int64_t kRec = FMT == 2 ? (int64_t)kWave * 16 : kExtBytes + (int64_t)kColSlotBytes * W;
  const int done_flag = done ? *done : 0;
  __shared__ double dict_s[kBlock / kWave][kDictSize];
  __shared__ int offs_s[FMT == 2 ? kBlock / kWave : 1][FMT == 2 ? kDictSize : 1];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double *dw = dict_s[wave];
  int *ow = offs_s[FMT == 2 ? wave : 0];
  const int bidx = (int)blockIdx.x;
  const int lb = A.xcd_group > 1 ? xcd_remap_grouped(bidx, gridDim.x, A.xcd_group)
                                 : (A.xcd_group == 1 ? xcd_remap(bidx, gridDim.x) : bidx);
  const int64_t sl0 = ((int64_t)lb * (kBlock / kWave) + wave) * SPW;
  const double alpha = ld_scal2(alpha_s), beta = ld_scal2(beta_s);
  const int64_t last_row = A.n_rows - 1;
  const bool w_is_x = DOT && dot.w == x;
  const bool w_load = DOT && dot.w != nullptr && !w_is_x;

  uint64_t iw[SPW], jw[SPW];
  int2v c[SPW][NP > 0 ? NP : 1];
  int ct[SPW];
  double xi[SPW], wi[SPW], yo[SPW];
  int64_t row[SPW];
  bool valid[SPW];
#pragma unroll
  for (int u = 0; u < SPW; ++u) {
    const int64_t sl = sl0 + u;
    const bool active = sl < n_launch_slices;  // wave-uniform
    const int64_t slice = slice_list ? (int64_t)slice_list[active ? sl : 0] : (active ? sl : 0);
    const int64_t r = slice * kWave + lane;
    valid[u] = active && r <= last_row;
    row[u] = r <= last_row ? r : last_row;
    const char *rec = A.pack + slice * kRec;
    ct[u] = 0;
    jw[u] = 0;
    if (FMT == 2) {
      typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
      const u64x2 word = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(rec) + lane);
      iw[u] = word.x, jw[u] = word.y;
    } else {
      iw[u] = __builtin_nontemporal_load(reinterpret_cast<const uint64_t *>(rec) + lane);
      const int2v *cp2 = reinterpret_cast<const int2v *>(rec + kExtBytes) + lane;
#pragma unroll
      for (int q = 0; q < NP; ++q) c[u][q] = __builtin_nontemporal_load(cp2 + q * kWave);
      if (W & 1)
        ct[u] = __builtin_nontemporal_load(reinterpret_cast<const int *>(rec + kExtBytes + NP * (kWave * 8)) + lane);
    }
    xi[u] = x[row[u]];
    wi[u] = 0.0;
    yo[u] = 0.0;
    if (A.accumulate) yo[u] = y[row[u]];
    if (w_load) wi[u] = dot.w[row[u]];  // (w == x, CG's <p, Ap>, reuses xi at the end: no copy here, a
  }                                      //  copy would wait for xi in the middle of the load issue)
  // The wave's own copy of the tables, requested AFTER the record loads (memory returns in order, so
  // the copy costs no extra round trip).  The tables are allocated with kDictSize entries: the first
  // 64 are copied unconditionally, the rest under a scalar branch that operators with a handful of
  // distinct values never take.
  {
    const int o0 = FMT == 2 ? A.offs[lane] : 0;
    const double d0 = A.dict[lane];
    if (FMT == 2) ow[lane] = o0;
    dw[lane] = d0;
    if (A.dict_size > kWave || A.offs_size > kWave) {
#pragma unroll
      for (int j = 1; j < kDictSize / kWave; ++j) {
        if (FMT == 2) ow[lane + j * kWave] = A.offs[lane + j * kWave];
        dw[lane + j * kWave] = A.dict[lane + j * kWave];
      }
    }
  }
  double xg[SPW][W > 0 ? W : 1];
  if (FMT == 2) __builtin_amdgcn_wave_barrier();  // the wave's offset table is complete (same-wave LDS order)
#pragma unroll
  for (int u = 0; u < SPW; ++u) {
    if (FMT == 2) {
      // padding slots (and every slot of a row past the end) carry offset 0 and weight 0
      const double *xr = x + row[u];
#pragma unroll
      for (int k = 0; k < W; ++k) xg[u][k] = xr[ow[(unsigned)(jw[u] >> (8 * k)) & 0xffu]];
    } else {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        xg[u][2 * q] = x[c[u][q].x];
        xg[u][2 * q + 1] = x[c[u][q].y];
      }
      if (W & 1) xg[u][W - 1] = x[ct[u]];
    }
  }
  __builtin_amdgcn_wave_barrier();  // the wave's dictionary copy is complete (same-wave LDS order)
  double da = 0.0, db = 0.0;
#pragma unroll
  for (int u = 0; u < SPW; ++u) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < W; ++k) acc += dw[(unsigned)(iw[u] >> (8 * (k + 1))) & 0xffu] * (xg[u][k] - xi[u]);
    const double ext = dw[(unsigned)iw[u] & 0xffu];
    double yi = (A.accumulate ? yo[u] : beta * xi[u]) + alpha * (acc + ext * xi[u]);
    if (valid[u] && !done_flag) __builtin_nontemporal_store(yi, y + row[u]);
    yi = valid[u] ? yi : 0.0;
    if (DOT) {
      da += (w_is_x ? xi[u] : wi[u]) * yi;
      db += yi * yi;
    }
  }
  if (done_flag) return;
  if (DOT) {
    double a = dot.w ? da : 0.0;
    a = wave_sum_to_lane63(a);
    if (dot.yy) db = wave_sum_to_lane63(db);
    if (lane == kWave - 1) {
      const int slot = dot.block_offset + (int)blockIdx.x * (kBlock / kWave) + wave;
      dot.partials[slot] = a;
      if (dot.yy) dot.partials[dot.nblocks_total + slot] = db;
    }
  }
}

template <bool DOT, int SPW>
static void launch_dict(const storm_hip_op *op, int nb, Scal alpha, Scal beta, const double *x, double *y,
                        const int *slice_list, int64_t n_launch, DotArgs dot, const int *done, hipEvent_t ev0,
                        hipEvent_t ev1, bool accumulate) {
  // slice lists (interior / boundary sets) are not contiguous: no XCD grouping there
  const int group = slice_list ? 0 : (int)op->ctx->opt_spmv_xcd_remap;
  SellArgs A{op->d_pack, op->d_slice_off, op->n_rows, op->uniform_width, group, op->d_dict, op->dict_size,
             op->d_offs, op->offs_size, (int)accumulate};
  hipStream_t st = op->ctx->stream;
#define DICT_GO(W_)                                                                                                 \
  do {                                                                                                              \
    if (op->offs_size > 0)                                                                                          \
      hipExtLaunchKernelGGL((spmv_dict_kernel<DOT, W_, SPW, 2>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A,     \
                            alpha, beta, x, y, slice_list, n_launch, dot, done);                                    \
    else                                                                                                            \
      hipExtLaunchKernelGGL((spmv_dict_kernel<DOT, W_, SPW, 1>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A,     \
                            alpha, beta, x, y, slice_list, n_launch, dot, done);                                    \
  } while (0)
  switch (op->uniform_width) {
    case 1: DICT_GO(1); break;
    case 2: DICT_GO(2); break;
    case 3: DICT_GO(3); break;
    case 4: DICT_GO(4); break;
    case 5: DICT_GO(5); break;
    case 6: DICT_GO(6); break;
    default: DICT_GO(7); break;
  }
#undef DICT_GO
}

bool spmv_dict_run(const RangeLaunch &L) {
  const storm_hip_op *op = L.op;
  if (op->pair || !(op_spw(op) >= 1 && op->dict_size > 0 && op->uniform_width > 0)) return false;
#define DICT_SPW(S_)                                                                                                         \
  do {                                                                                                                       \
    if (L.want_dot)                                                                                                          \
      launch_dict<true, S_>(op, L.nb, L.alpha, L.beta, L.x, L.y, L.slice_list, L.n_launch, L.dot, L.done, L.ev0, L.ev1, L.accumulate);  \
    else                                                                                                                     \
      launch_dict<false, S_>(op, L.nb, L.alpha, L.beta, L.x, L.y, L.slice_list, L.n_launch, L.dot, L.done, L.ev0, L.ev1, L.accumulate); \
  } while (0)
  switch (op_spw(op)) {
    case 1: DICT_SPW(1); break;
    case 2: DICT_SPW(2); break;
    default: DICT_SPW(4); break;
  }
#undef DICT_SPW
  return true;
}

}  // namespace storm
