// Operator apply, sliced-ELL records: formats 0 (fp64 records) and 1 / 2 of mixed width through the general kernel, and
// the CSR tail.  Record layout: the header of spmv.hip.
#include "spmv_device.hpp"

namespace storm {



// sum_k w_k (x[col_k] - x_i) over slots [S0, S0 + W) of a record whose slice has `width` slots
// (W compile-time, S0 even).  Pairs are read as int2 / double2, an odd last slot unpaired.
template <bool NT, int W>
__device__ __forceinline__ double row_sum(const char *rec, int width, int lane, const double *__restrict__ x,
                                          double xi, int s0 = 0) {
  constexpr int NP = W / 2;
  const int npair_total = width >> 1;
  const int2v *cp2 = reinterpret_cast<const int2v *>(rec + kExtBytes) + lane + (s0 >> 1) * kWave;
  const char *vbase = rec + kExtBytes + (int64_t)width * (kWave * 4);
  const double2v *vp2 = reinterpret_cast<const double2v *>(vbase) + lane + (s0 >> 1) * kWave;
  int2v c[NP > 0 ? NP : 1];
  double2v v[NP > 0 ? NP : 1];
  int ct = 0;
  double vt = 0.0;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    c[q] = NT ? __builtin_nontemporal_load(cp2 + q * kWave) : cp2[q * kWave];
    v[q] = NT ? __builtin_nontemporal_load(vp2 + q * kWave) : vp2[q * kWave];
  }
  if (W & 1) {  // the slice's unpaired last slot
    ct = ld_i<NT>(reinterpret_cast<const int *>(rec + kExtBytes + (int64_t)npair_total * (kWave * 8)) + lane);
    vt = ld_d<NT>(reinterpret_cast<const double *>(vbase + (int64_t)npair_total * (kWave * 16)) + lane);
  }
  double xg[W > 0 ? W : 1];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    xg[2 * q] = x[c[q].x];
    xg[2 * q + 1] = x[c[q].y];
  }
  if (W & 1) xg[W - 1] = x[ct];
  double acc = 0.0;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    acc += v[q].x * (xg[2 * q] - xi);
    acc += v[q].y * (xg[2 * q + 1] - xi);
  }
  if (W & 1) acc += vt * (xg[W - 1] - xi);
  return acc;
}

// Rows wider than 8 slots: chunks of 8, then the remainder.
template <bool NT>
__device__ __forceinline__ double row_sum_wide(const char *rec, int width, int lane, const double *__restrict__ x,
                                               double xi) {
  double acc = 0.0;
  int s0 = 0;
  for (; s0 + 8 <= width; s0 += 8) acc += row_sum<NT, 8>(rec, width, lane, x, xi, s0);
  switch (width - s0) {
    case 1: acc += row_sum<NT, 1>(rec, width, lane, x, xi, s0); break;
    case 2: acc += row_sum<NT, 2>(rec, width, lane, x, xi, s0); break;
    case 3: acc += row_sum<NT, 3>(rec, width, lane, x, xi, s0); break;
    case 4: acc += row_sum<NT, 4>(rec, width, lane, x, xi, s0); break;
    case 5: acc += row_sum<NT, 5>(rec, width, lane, x, xi, s0); break;
    case 6: acc += row_sum<NT, 6>(rec, width, lane, x, xi, s0); break;
    case 7: acc += row_sum<NT, 7>(rec, width, lane, x, xi, s0); break;
    default: break;
  }
  return acc;
}

// Value-dictionary record: columns as in row_sum, the weight of slot k is dict[byte k + 1 of iw] (LDS).
template <bool NT, int W>
__device__ __forceinline__ double row_sum_cv(const char *rec, int width, int lane, const double *__restrict__ x,
                                             double xi, uint64_t iw, const double *dict) {
  constexpr int NP = W / 2;
  const int npair_total = width >> 1;
  const int2v *cp2 = reinterpret_cast<const int2v *>(rec + kExtBytes) + lane;
  int2v c[NP > 0 ? NP : 1];
  int ct = 0;
#pragma unroll
  for (int q = 0; q < NP; ++q) c[q] = NT ? __builtin_nontemporal_load(cp2 + q * kWave) : cp2[q * kWave];
  if (W & 1) ct = ld_i<NT>(reinterpret_cast<const int *>(rec + kExtBytes + (int64_t)npair_total * (kWave * 8)) + lane);
  double xg[W > 0 ? W : 1];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    xg[2 * q] = x[c[q].x];
    xg[2 * q + 1] = x[c[q].y];
  }
  if (W & 1) xg[W - 1] = x[ct];
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < W; ++k) acc += dict[(unsigned)(iw >> (8 * (k + 1))) & 0xffu] * (xg[k] - xi);
  return acc;
}

// One wavefront per slice, one row per lane, 4 slices per 256-thread block.
//   NT      : record / y traffic marked non-temporal so it does not evict x from L2 (+15 %).
//   DOT     : epilogue writes per-block partials of <w, y> and <y, y> (fused reductions).
//   VARIANT : 0 fp64 records, 2 value-dictionary records (x is gathered straight from global memory: L1 / L2 / the
//             Infinity Cache serve the reuse; staging the block's own rows in LDS gave no gain and is gone)
template <bool NT, bool DOT, int VARIANT, bool XCD>
__global__ __launch_bounds__(kBlock) void spmv_sell_kernel(SellArgs A, Scal alpha_s, Scal beta_s,
                                                           const double *__restrict__ x,
                                                           double *__restrict__ y,
                                                           const int *__restrict__ slice_list,
                                                           int64_t n_launch_slices, DotArgs dot,
                                                           const int *done) {
  // The `done` predicate is only needed before the first store: issue its (scalar) load now and
  // test it after the gathers, so it never sits at the head of a wave's dependency chain.
  const int done_flag = done ? *done : 0;
  __shared__ double dict_s[VARIANT == 2 ? kDictSize : 1];
  if (VARIANT == 2) {
    static_assert(kDictSize == kBlock, "one dictionary entry per thread");
    dict_s[threadIdx.x] = A.dict[threadIdx.x];
    __syncthreads();
  }
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: slice math runs on the SALU
  const int bidx = (int)blockIdx.x;
  const int lb = XCD ? (A.xcd_group > 1 ? xcd_remap_grouped(bidx, gridDim.x, A.xcd_group) : xcd_remap(bidx, gridDim.x))
                     : bidx;
  const int64_t sl = (int64_t)lb * (kBlock / kWave) + wave;
  const bool wave_active = sl < n_launch_slices;
  const double alpha = ld_scal2(alpha_s), beta = ld_scal2(beta_s);

  double yi = 0.0, wi = 0.0;
  int64_t row = 0;
  bool valid = false;
  double xi = 0.0;
  int64_t slice = 0;
  if (wave_active) {
    slice = slice_list ? (int64_t)slice_list[sl] : sl;
    row = slice * kWave + lane;
    valid = row < A.n_rows;
    xi = valid ? x[row] : 0.0;
    if (DOT && dot.w) wi = (dot.w == x) ? xi : (valid ? dot.w[row] : 0.0);  // early: off the tail of the chain
  }
  if (wave_active) {
    constexpr int kSlot = (VARIANT == 2) ? kColSlotBytes : kSlotBytes;
    int64_t base;
    int width;
    if (A.uniform_width > 0) {
      width = A.uniform_width;
      base = slice * (int64_t)(kExtBytes + kSlot * width);
    } else {
      base = A.slice_off[slice];
      width = (int)((A.slice_off[slice + 1] - base - kExtBytes) / kSlot);
    }
    const char *rec = A.pack + base;
    double ext, acc;
    if (VARIANT == 2) {
      const uint64_t *ip = reinterpret_cast<const uint64_t *>(rec) + lane;
      const uint64_t iw = NT ? __builtin_nontemporal_load(ip) : *ip;
      ext = dict_s[(unsigned)iw & 0xffu];
      switch (width) {  // build_op guarantees width <= 7 for these records
        case 0: acc = 0.0; break;
        case 1: acc = row_sum_cv<NT, 1>(rec, 1, lane, x, xi, iw, dict_s); break;
        case 2: acc = row_sum_cv<NT, 2>(rec, 2, lane, x, xi, iw, dict_s); break;
        case 3: acc = row_sum_cv<NT, 3>(rec, 3, lane, x, xi, iw, dict_s); break;
        case 4: acc = row_sum_cv<NT, 4>(rec, 4, lane, x, xi, iw, dict_s); break;
        case 5: acc = row_sum_cv<NT, 5>(rec, 5, lane, x, xi, iw, dict_s); break;
        case 6: acc = row_sum_cv<NT, 6>(rec, 6, lane, x, xi, iw, dict_s); break;
        default: acc = row_sum_cv<NT, 7>(rec, 7, lane, x, xi, iw, dict_s); break;
      }
    } else {
    ext = ld_d<NT>(reinterpret_cast<const double *>(rec) + lane);
    // The width is wave-uniform: dispatch to a body with the width as a compile-time constant,
    // so all (col, val) loads of the row are issued back to back, then all gathers, then the
    // FMAs -- no branch (and no s_waitcnt) between the gathers of one row.
    switch (width) {
      case 0: acc = 0.0; break;
      case 1: acc = row_sum<NT, 1>(rec, 1, lane, x, xi); break;
      case 2: acc = row_sum<NT, 2>(rec, 2, lane, x, xi); break;
      case 3: acc = row_sum<NT, 3>(rec, 3, lane, x, xi); break;
      case 4: acc = row_sum<NT, 4>(rec, 4, lane, x, xi); break;
      case 5: acc = row_sum<NT, 5>(rec, 5, lane, x, xi); break;
      case 6: acc = row_sum<NT, 6>(rec, 6, lane, x, xi); break;
      case 7: acc = row_sum<NT, 7>(rec, 7, lane, x, xi); break;
      case 8: acc = row_sum<NT, 8>(rec, 8, lane, x, xi); break;
      default: acc = row_sum_wide<NT>(rec, width, lane, x, xi); break;
    }
    }
    yi = (A.accumulate ? (valid ? y[row] : 0.0) : beta * xi) + alpha * (acc + ext * xi);
    if (valid && !done_flag) {
      if (NT) __builtin_nontemporal_store(yi, y + row);
      else y[row] = yi;
    }
    if (!valid) yi = 0.0;
  }
  if (done_flag) return;  // block-uniform
  if (DOT) {
    // One partial per WAVE (64-lane DPP tree, lane 63 stores): no LDS, no block barrier.
    // (A per-block partial with __syncthreads cost 6 % of the kernel: every wave of a block had
    // to outlive its slowest sibling.)  The 4x longer partial arrays are folded by the two-pass
    // final reduction in solvers.hip.
    double a = dot.w ? wi * yi : 0.0, b = dot.yy ? yi * yi : 0.0;
    a = wave_sum_to_lane63(a);
    if (dot.yy) b = wave_sum_to_lane63(b);
    if (lane == kWave - 1) {
      const int slot = dot.block_offset + (int)blockIdx.x * (kBlock / kWave) + wave;
      dot.partials[slot] = a;
      if (dot.yy) dot.partials[dot.nblocks_total + slot] = b;
    }
  }
}


// CSR tail: one wavefront per overflowing row; the lanes' partial products are folded
// with __shfl_down and lane 0 adds the row's remainder to y.
__global__ __launch_bounds__(kBlock) void spmv_tail_kernel(int64_t n_tail, const int *__restrict__ tail_row,
                                                           const int64_t *__restrict__ tail_ptr,
                                                           const int *__restrict__ tail_col,
                                                           const double *__restrict__ tail_val,
                                                           Scal alpha_s, const double *__restrict__ x,
                                                           double *__restrict__ y, const int *done) {
  if (done && *done) return;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t t = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  if (t >= n_tail) return;
  const double alpha = ld_scal2(alpha_s);
  const int r = tail_row[t];
  const double xi = x[r];
  double acc = 0.0;
  for (int64_t k = tail_ptr[t] + lane; k < tail_ptr[t + 1]; k += kWave)
    acc += tail_val[k] * (x[tail_col[k]] - xi);
  acc = wave_sum_down(acc);
  if (lane == 0) y[r] += alpha * acc;
}

template <bool NT, bool DOT, int VARIANT>
static void launch_sell(const storm_hip_op *op, int nb, Scal alpha, Scal beta, const double *x, double *y,
                        const int *slice_list, int64_t n_launch, DotArgs dot, const int *done, hipEvent_t ev0,
                        hipEvent_t ev1, bool accumulate) {
  SellArgs A{op->d_pack, op->d_slice_off, op->n_rows, op->uniform_width, op->xcd_group_sell, op->d_dict, op->dict_size,
             op->d_offs, op->offs_size, (int)accumulate};
  constexpr int LV = VARIANT;
  hipStream_t st = op->ctx->stream;
  if (slice_list == nullptr && op->xcd_group_sell != 0) {
    hipExtLaunchKernelGGL((spmv_sell_kernel<NT, DOT, VARIANT, true>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A, alpha,
                       beta, x, y, slice_list, n_launch, dot, done);
  } else if (slice_list == nullptr) {
    hipExtLaunchKernelGGL((spmv_sell_kernel<NT, DOT, VARIANT, false>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A, alpha,
                       beta, x, y, slice_list, n_launch, dot, done);
  } else if (op->xcd_group_sell != 0) {
    // listed slices (interior / boundary sets of a partitioned operator): the LDS window does not
    // apply, the XCD grouping still does -- the interior list is consecutive but for a few gaps
    hipExtLaunchKernelGGL((spmv_sell_kernel<NT, DOT, LV, true>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A, alpha, beta,
                       x, y, slice_list, n_launch, dot, done);
  } else {
    hipExtLaunchKernelGGL((spmv_sell_kernel<NT, DOT, LV, false>), dim3(nb), dim3(kBlock), 0, st, ev0, ev1, 0, A, alpha, beta,
                       x, y, slice_list, n_launch, dot, done);
  }
}

bool spmv_sell_run(const RangeLaunch &L) {
  const storm_hip_op *op = L.op;
  const bool nt = op->ctx->opt_nt != 0;
#define SPMV_GO(NT_, DOT_, VAR_) \
  launch_sell<NT_, DOT_, VAR_>(op, L.nb, L.alpha, L.beta, L.x, L.y, L.slice_list, L.n_launch, L.dot, L.done, L.ev0, L.ev1, L.accumulate)
#define SPMV_VAR(VAR_)                                                                        \
  do {                                                                                        \
    if (nt) { if (L.want_dot) SPMV_GO(true, true, VAR_); else SPMV_GO(true, false, VAR_); }   \
    else    { if (L.want_dot) SPMV_GO(false, true, VAR_); else SPMV_GO(false, false, VAR_); } \
  } while (0)
  if (op->dict_size > 0) SPMV_VAR(2);
  else SPMV_VAR(0);
#undef SPMV_VAR
#undef SPMV_GO
  return true;
}

int spmv_tail_run(const storm_hip_op *op, Scal alpha, const double *x, double *y, const int *done) {
  const int nb = (int)((op->tail_rows + 3) / 4);
  hipLaunchKernelGGL(spmv_tail_kernel, dim3(nb), dim3(kBlock), 0, op->ctx->stream, op->tail_rows, op->d_tail_row,
                     op->d_tail_ptr, op->d_tail_col, op->d_tail_val, alpha, x, y, done);
  HIP_TRY(hipGetLastError());
  return STORM_HIP_OK;
}

}  // namespace storm
