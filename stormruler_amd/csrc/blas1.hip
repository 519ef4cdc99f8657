// BLAS-1 kernels: the element loops and reductions the Krylov bodies run on
// (reference: Bittern/MatrixAlgorithms.hpp:58-81 matrix_for_each, :162-205 reduce).
//
// All kernels are HBM-bound streams: 16-byte (double2) accesses per lane, a
// grid capped at a few resident blocks per CU with a grid-stride loop, and --
// for reductions -- per-lane partial sums folded with 64-wide __shfl_down,
// then across the 4 waves of a block through LDS, then one fixed-order final
// pass over the per-block partials.  The summation tree depends only on
// (n, grid), so results are bitwise reproducible run to run.
#include "common.hpp"

namespace storm {

__device__ __forceinline__ double ld_scal(const Scal &s) { return s.p ? (*s.p) * s.sign : s.v; }

static inline int ew_blocks(const storm_hip_ctx *c, int64_t n) {
  // each thread moves double2 x 2 per trip
  const int64_t need = (n / 2 + kBlock * 2 - 1) / (kBlock * 2);
  const int64_t cap = (int64_t)c->num_cus * 8;
  return (int)(need < 1 ? 1 : (need > cap ? cap : need));
}

// ---- elementwise ---------------------------------------------------------------------
struct EwPtrs {
  double *y;
  const double *x0, *x1;
};

template <class F>
__global__ __launch_bounds__(kBlock) void ew_kernel(int64_t n, EwPtrs p, F f, const int *done, int rev) {
  if (done && *done) return;
  f.prepare();
  const int64_t n2 = n >> 1;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  double2 *__restrict__ y2 = reinterpret_cast<double2 *>(p.y);
  const double2 *__restrict__ a2 = reinterpret_cast<const double2 *>(p.x0);
  const double2 *__restrict__ b2 = reinterpret_cast<const double2 *>(p.x1);
#pragma unroll 2
  for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < n2; i0 += stride) {
    const int64_t i = rev ? n2 - 1 - i0 : i0;
    double2 vy = make_double2(0, 0), va = make_double2(0, 0), vb = make_double2(0, 0);
    if (F::reads_y) vy = y2[i];
    if (F::nin > 0) va = a2[i];
    if (F::nin > 1) vb = b2[i];
    double2 o;
    o.x = f(vy.x, va.x, vb.x);
    o.y = f(vy.y, va.y, vb.y);
    y2[i] = o;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    const double vy = F::reads_y ? p.y[i] : 0.0;
    const double va = F::nin > 0 ? p.x0[i] : 0.0;
    const double vb = F::nin > 1 ? p.x1[i] : 0.0;
    p.y[i] = f(vy, va, vb);
  }
}

struct FillF {
  static constexpr bool reads_y = false;
  static constexpr int nin = 0;
  double v;
  __device__ void prepare() {}
  __device__ double operator()(double, double, double) const { return v; }
};
struct CopyF {
  static constexpr bool reads_y = false;
  static constexpr int nin = 1;
  __device__ void prepare() {}
  __device__ double operator()(double, double a, double) const { return a; }
};
struct ScaleF {
  static constexpr bool reads_y = true;
  static constexpr int nin = 0;
  Scal s;
  bool divide;
  double sv;
  __device__ void prepare() { sv = ld_scal(s); }
  __device__ double operator()(double y, double, double) const { return divide ? y / sv : y * sv; }
};
// y = a*x0 + b*x1 (either input may be y itself)
struct AxpbzF {
  static constexpr bool reads_y = false;
  static constexpr int nin = 2;
  Scal a, b;
  double av, bv;
  __device__ void prepare() { av = ld_scal(a), bv = ld_scal(b); }
  __device__ double operator()(double, double x0, double x1) const { return av * x0 + bv * x1; }
};
// p = r + beta*(p - omega*v)     SolverBiCgStab.hpp:119
struct BicgPF {
  static constexpr bool reads_y = true;
  static constexpr int nin = 2;
  Scal beta, omega;
  double bv, wv;
  __device__ void prepare() { bv = ld_scal(beta), wv = ld_scal(omega); }
  __device__ double operator()(double p, double r, double v) const { return r + bv * (p - wv * v); }
};

template <class F>
static int launch_ew(storm_hip_ctx *c, int64_t n, EwPtrs p, F f, const int *done) {
  if (n <= 0) return STORM_HIP_OK;
  hipLaunchKernelGGL(ew_kernel<F>, dim3(ew_blocks(c, n)), dim3(kBlock), 0, c->stream, n, p, f, done,
                     c->next_dir());
  HIP_TRY(hipGetLastError());
  return STORM_HIP_OK;
}

int k_fill(storm_hip_ctx *c, double *y, int64_t n, double v) {
  return launch_ew(c, n, EwPtrs{y, nullptr, nullptr}, FillF{v}, nullptr);
}
int k_copy(storm_hip_ctx *c, double *y, const double *x, int64_t n, const int *done) {
  return launch_ew(c, n, EwPtrs{y, x, nullptr}, CopyF{}, done);
}
int k_scale(storm_hip_ctx *c, double *y, int64_t n, Scal s, bool divide, const int *done) {
  return launch_ew(c, n, EwPtrs{y, nullptr, nullptr}, ScaleF{s, divide, 0.0}, done);
}
int k_axpbz(storm_hip_ctx *c, double *y, Scal a, const double *x, Scal b, const double *z, int64_t n,
            const int *done) {
  return launch_ew(c, n, EwPtrs{y, x, z}, AxpbzF{a, b, 0.0, 0.0}, done);
}
int k_bicg_p(storm_hip_ctx *c, double *p, const double *r, Scal beta, Scal omega, const double *v,
             int64_t n, const int *done) {
  return launch_ew(c, n, EwPtrs{p, r, v}, BicgPF{beta, omega, 0.0, 0.0}, done);
}

// ---- reductions -------------------------------------------------------------------------

// Sum over the 256 threads of a block, fixed order; result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double *lds4) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  __syncthreads();  // lds4 may still be read by a previous call
  if (lane == 0) lds4[wave] = v;
  __syncthreads();
  return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

constexpr int kDotChunk = 8;
struct DotPtrs {
  const double *b[kDotChunk];
};

template <int KB>
__global__ __launch_bounds__(kBlock) void multi_dot_kernel(int64_t n, const double *__restrict__ a,
                                                           DotPtrs bs, double *__restrict__ partials,
                                                           const int *done) {
  if (done && *done) return;
  __shared__ double lds4[4];
  double acc[KB];
#pragma unroll
  for (int j = 0; j < KB; ++j) acc[j] = 0.0;
  const int64_t n2 = n >> 1;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const double2 *__restrict__ a2 = reinterpret_cast<const double2 *>(a);
#pragma unroll 2
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
    const double2 va = a2[i];
#pragma unroll
    for (int j = 0; j < KB; ++j) {
      const double2 vb = reinterpret_cast<const double2 *>(bs.b[j])[i];
      acc[j] += va.x * vb.x;
      acc[j] += va.y * vb.y;
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < KB; ++j) acc[j] += a[n - 1] * bs.b[j][n - 1];
  }
#pragma unroll
  for (int j = 0; j < KB; ++j) {
    const double s = block_sum(acc[j], lds4);
    if (threadIdx.x == 0) partials[(int64_t)j * gridDim.x + blockIdx.x] = s;
  }
}

// out[j] = sum_b partials[j * nblocks + b]; one block per j, fixed order.
__global__ __launch_bounds__(kBlock) void reduce_final_kernel(const double *__restrict__ partials,
                                                              int nblocks, double *__restrict__ out,
                                                              const int *done) {
  if (done && *done) return;
  __shared__ double lds4[4];
  const double *p = partials + (int64_t)blockIdx.x * nblocks;
  double v = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += kBlock) v += p[i];
  const double s = block_sum(v, lds4);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

static inline int reduce_blocks(const storm_hip_ctx *c, int64_t n) {
  const int64_t need = (n / 2 + kBlock * 4 - 1) / (kBlock * 4);
  const int64_t cap = (int64_t)c->num_cus * 4;
  int64_t b = need < 1 ? 1 : (need > cap ? cap : need);
  if (b > kMaxReduceBlocks) b = kMaxReduceBlocks;
  return (int)b;
}

int k_reduce_final(storm_hip_ctx *c, const double *partials, int nblocks, int k, double *d_out,
                   const int *done) {
  hipLaunchKernelGGL(reduce_final_kernel, dim3(k), dim3(kBlock), 0, c->stream, partials, nblocks,
                     d_out, done);
  HIP_TRY(hipGetLastError());
  return STORM_HIP_OK;
}

int k_multi_dot(storm_hip_ctx *c, const double *a, const double *const *bs, int k, int64_t n,
                double *d_out, const int *done) {
  STORM_REQUIRE(k >= 1 && k <= kMaxMulti, "multi_dot: k = %d outside [1, %d]", k, kMaxMulti);
  const int nb = reduce_blocks(c, n);
  for (int j0 = 0; j0 < k; j0 += kDotChunk) {
    const int kb = (k - j0) < kDotChunk ? (k - j0) : kDotChunk;
    DotPtrs ptrs;
    for (int j = 0; j < kDotChunk; ++j) ptrs.b[j] = bs[j0 + (j < kb ? j : 0)];
    double *part = c->d_partials + (int64_t)j0 * nb;
    const dim3 g(nb), b(kBlock);
    switch (kb) {
      case 1: hipLaunchKernelGGL(multi_dot_kernel<1>, g, b, 0, c->stream, n, a, ptrs, part, done); break;
      case 2: hipLaunchKernelGGL(multi_dot_kernel<2>, g, b, 0, c->stream, n, a, ptrs, part, done); break;
      case 3: hipLaunchKernelGGL(multi_dot_kernel<3>, g, b, 0, c->stream, n, a, ptrs, part, done); break;
      case 4: hipLaunchKernelGGL(multi_dot_kernel<4>, g, b, 0, c->stream, n, a, ptrs, part, done); break;
      case 5: hipLaunchKernelGGL(multi_dot_kernel<5>, g, b, 0, c->stream, n, a, ptrs, part, done); break;
      case 6: hipLaunchKernelGGL(multi_dot_kernel<6>, g, b, 0, c->stream, n, a, ptrs, part, done); break;
      case 7: hipLaunchKernelGGL(multi_dot_kernel<7>, g, b, 0, c->stream, n, a, ptrs, part, done); break;
      default: hipLaunchKernelGGL(multi_dot_kernel<8>, g, b, 0, c->stream, n, a, ptrs, part, done); break;
    }
    HIP_TRY(hipGetLastError());
  }
  return k_reduce_final(c, c->d_partials, nb, k, d_out, done);
}

// ---- multi-axpy ------------------------------------------------------------------------------
constexpr int kAxpyChunk = 8;
struct AxpyArgs {
  const double *x[kAxpyChunk];
  double c[kAxpyChunk];
  const double *dc;  // device coefficients (override c[] when non-null)
  double sign;
};

template <int KB>
__global__ __launch_bounds__(kBlock) void multi_axpy_kernel(int64_t n, double *__restrict__ y,
                                                            AxpyArgs a, const int *done) {
  if (done && *done) return;
  double cf[KB];
#pragma unroll
  for (int j = 0; j < KB; ++j) cf[j] = a.dc ? a.dc[j] * a.sign : a.c[j];
  const int64_t n2 = n >> 1;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  double2 *__restrict__ y2 = reinterpret_cast<double2 *>(y);
#pragma unroll 2
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
    double2 vy = y2[i];
#pragma unroll
    for (int j = 0; j < KB; ++j) {
      const double2 vx = reinterpret_cast<const double2 *>(a.x[j])[i];
      vy.x += cf[j] * vx.x;
      vy.y += cf[j] * vx.y;
    }
    y2[i] = vy;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    double vy = y[n - 1];
#pragma unroll
    for (int j = 0; j < KB; ++j) vy += cf[j] * a.x[j][n - 1];
    y[n - 1] = vy;
  }
}

static int multi_axpy_impl(storm_hip_ctx *c, double *y, const double *h_coef, const double *d_coef,
                           double sign, const double *const *xs, int k, int64_t n, const int *done) {
  if (n <= 0 || k <= 0) return STORM_HIP_OK;
  const dim3 g(ew_blocks(c, n)), b(kBlock);
  for (int j0 = 0; j0 < k; j0 += kAxpyChunk) {
    const int kb = (k - j0) < kAxpyChunk ? (k - j0) : kAxpyChunk;
    AxpyArgs a;
    for (int j = 0; j < kAxpyChunk; ++j) {
      a.x[j] = xs[j0 + (j < kb ? j : 0)];
      a.c[j] = (h_coef && j < kb) ? h_coef[j0 + j] : 0.0;
    }
    a.dc = d_coef ? d_coef + j0 : nullptr;
    a.sign = sign;
    switch (kb) {
      case 1: hipLaunchKernelGGL(multi_axpy_kernel<1>, g, b, 0, c->stream, n, y, a, done); break;
      case 2: hipLaunchKernelGGL(multi_axpy_kernel<2>, g, b, 0, c->stream, n, y, a, done); break;
      case 3: hipLaunchKernelGGL(multi_axpy_kernel<3>, g, b, 0, c->stream, n, y, a, done); break;
      case 4: hipLaunchKernelGGL(multi_axpy_kernel<4>, g, b, 0, c->stream, n, y, a, done); break;
      case 5: hipLaunchKernelGGL(multi_axpy_kernel<5>, g, b, 0, c->stream, n, y, a, done); break;
      case 6: hipLaunchKernelGGL(multi_axpy_kernel<6>, g, b, 0, c->stream, n, y, a, done); break;
      case 7: hipLaunchKernelGGL(multi_axpy_kernel<7>, g, b, 0, c->stream, n, y, a, done); break;
      default: hipLaunchKernelGGL(multi_axpy_kernel<8>, g, b, 0, c->stream, n, y, a, done); break;
    }
    HIP_TRY(hipGetLastError());
  }
  return STORM_HIP_OK;
}

int k_multi_axpy(storm_hip_ctx *c, double *y, const double *d_coef, double sign,
                 const double *const *xs, int k, int64_t n, const int *done) {
  return multi_axpy_impl(c, y, nullptr, d_coef, sign, xs, k, n, done);
}

}  // namespace storm

using namespace storm;

// ---- C ABI --------------------------------------------------------------------------------------
static int check_pair(const storm_hip_vec *a, const storm_hip_vec *b, const char *what) {
  STORM_REQUIRE(a && b, "%s: null vector", what);
  STORM_REQUIRE(a->ctx == b->ctx, "%s: vectors belong to different contexts", what);
  STORM_REQUIRE(a->n_owned == b->n_owned, "%s: size mismatch (%lld vs %lld owned rows)", what,
                (long long)a->n_owned, (long long)b->n_owned);
  return STORM_HIP_OK;
}

// Finish a host-visible reduction: sum over ranks, copy k scalars to the host.
static int finish_reduction(storm_hip_ctx *c, int k, double *out) {
  if (c->comm != nullptr) STORM_TRY(comm_allreduce_sum(c, c->d_scalars, k));
  HIP_TRY(hipMemcpyAsync(c->h_scalars, c->d_scalars, sizeof(double) * (size_t)k, hipMemcpyDeviceToHost,
                         c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int j = 0; j < k; ++j) out[j] = c->h_scalars[j];
  return STORM_HIP_OK;
}

extern "C" {

int storm_hip_fill(storm_hip_vec *y, double value) {
  STORM_REQUIRE(y, "fill: null vector");
  return k_fill(y->ctx, y->d, y->n_owned, value);
}

int storm_hip_copy(storm_hip_vec *y, const storm_hip_vec *x) {
  STORM_TRY(check_pair(y, x, "copy"));
  return k_copy(y->ctx, y->d, x->d, y->n_owned, nullptr);
}

int storm_hip_scale(storm_hip_vec *y, double s) {
  STORM_REQUIRE(y, "scale: null vector");
  return k_scale(y->ctx, y->d, y->n_owned, host_scal(s), false, nullptr);
}

int storm_hip_div_scalar(storm_hip_vec *y, double s) {
  STORM_REQUIRE(y, "div_scalar: null vector");
  return k_scale(y->ctx, y->d, y->n_owned, host_scal(s), true, nullptr);
}

int storm_hip_axpy(storm_hip_vec *y, double a, const storm_hip_vec *x) {
  STORM_TRY(check_pair(y, x, "axpy"));
  return k_axpbz(y->ctx, y->d, host_scal(a), x->d, host_scal(1.0), y->d, y->n_owned, nullptr);
}

int storm_hip_xpay(storm_hip_vec *y, const storm_hip_vec *x, double b) {
  STORM_TRY(check_pair(y, x, "xpay"));
  return k_axpbz(y->ctx, y->d, host_scal(1.0), x->d, host_scal(b), y->d, y->n_owned, nullptr);
}

int storm_hip_axpbz(storm_hip_vec *y, double a, const storm_hip_vec *x, double b, const storm_hip_vec *z) {
  STORM_TRY(check_pair(y, x, "axpbz"));
  STORM_TRY(check_pair(y, z, "axpbz"));
  return k_axpbz(y->ctx, y->d, host_scal(a), x->d, host_scal(b), z->d, y->n_owned, nullptr);
}

int storm_hip_bicgstab_p(storm_hip_vec *p, const storm_hip_vec *r, double beta, double omega,
                         const storm_hip_vec *v) {
  STORM_TRY(check_pair(p, r, "bicgstab_p"));
  STORM_TRY(check_pair(p, v, "bicgstab_p"));
  return k_bicg_p(p->ctx, p->d, r->d, host_scal(beta), host_scal(omega), v->d, p->n_owned, nullptr);
}

int storm_hip_multi_dot(const storm_hip_vec *a, const storm_hip_vec *const *bs, int k, double *out) {
  STORM_REQUIRE(a && bs && out, "multi_dot: null argument");
  STORM_REQUIRE(k >= 1 && k <= kMaxMulti, "multi_dot: k = %d outside [1, %d]", k, kMaxMulti);
  const double *ptrs[kMaxMulti];
  for (int j = 0; j < k; ++j) {
    STORM_TRY(check_pair(a, bs[j], "multi_dot"));
    ptrs[j] = bs[j]->d;
  }
  storm_hip_ctx *c = a->ctx;
  if (a->n_owned == 0) {
    HIP_TRY(hipMemsetAsync(c->d_scalars, 0, sizeof(double) * (size_t)k, c->stream));
  } else {
    STORM_TRY(k_multi_dot(c, a->d, ptrs, k, a->n_owned, c->d_scalars, nullptr));
  }
  return finish_reduction(c, k, out);
}

int storm_hip_dot(const storm_hip_vec *a, const storm_hip_vec *b, double *result) {
  STORM_REQUIRE(result, "dot: null result");
  return storm_hip_multi_dot(a, &b, 1, result);
}

int storm_hip_norm2(const storm_hip_vec *a, double *result) {
  STORM_REQUIRE(result, "norm2: null result");
  double s = 0.0;
  STORM_TRY(storm_hip_multi_dot(a, &a, 1, &s));
  *result = sqrt(s);
  return STORM_HIP_OK;
}

int storm_hip_multi_axpy(storm_hip_vec *y, const double *coefs, const storm_hip_vec *const *xs, int k) {
  STORM_REQUIRE(y && coefs && xs, "multi_axpy: null argument");
  STORM_REQUIRE(k >= 1 && k <= kMaxMulti, "multi_axpy: k = %d outside [1, %d]", k, kMaxMulti);
  const double *ptrs[kMaxMulti];
  for (int j = 0; j < k; ++j) {
    STORM_TRY(check_pair(y, xs[j], "multi_axpy"));
    ptrs[j] = xs[j]->d;
  }
  return multi_axpy_impl(y->ctx, y->d, coefs, nullptr, 1.0, ptrs, k, y->n_owned, nullptr);
}

}  // extern "C"
