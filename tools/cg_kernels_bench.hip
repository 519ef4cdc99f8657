// Micro-benchmark of the two BLAS-1 kernels of a fused CG iteration (csrc/solvers.hip: cg_r_kernel, cg_xp_kernel)
// in several shapes, at 256^3 doubles per vector, rotating over three buffer sets (the 256 MiB Infinity Cache must
// not decide).  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/cg_kernels_bench.hip -o /tmp/cgk && /tmp/cgk
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double double2v __attribute__((ext_vector_type(2)));
constexpr int kBlock = 256;
__device__ __forceinline__ double2v ldv(const double2v *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stv(double2v *p, double2v v) { __builtin_nontemporal_store(v, p); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
  v += dpp_mov<0xb1, 0xf>(v);
  v += dpp_mov<0x4e, 0xf>(v);
  v += dpp_mov<0x114, 0xf>(v);
  v += dpp_mov<0x118, 0xf>(v);
  v += dpp_mov<0x142, 0xa>(v);
  v += dpp_mov<0x143, 0xc>(v);
  return v;
}
__device__ __forceinline__ double block_sum_lds(double v, double *lds4) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) lds4[wave] = v;
  __syncthreads();
  return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

// RED: 0 none, 1 shfl + LDS + barrier, one partial per block (current), 2 DPP, one partial per wave,
//      3 DPP per wave + LDS + barrier, one partial per block
template <int U, int RED, bool SCAL>
__global__ __launch_bounds__(kBlock) void cg_r_k(long n2, const double *scal, double2v *__restrict__ r,
                                                 const double2v *__restrict__ z, double *__restrict__ partials) {
  __shared__ double lds4[4];
  double alpha = 0.5;
  if (SCAL) {
    if (scal[2] != 0.0) return;
    alpha = scal[1] == 0.0 ? 0.0 : scal[0] / scal[1];
  }
  double acc = 0.0;
  const long base = (long)blockIdx.x * (kBlock * U) + threadIdx.x;
  double2v vr[U], vz[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long i = base + u * kBlock;
    if (i < n2) vr[u] = ldv(r + i), vz[u] = ldv(z + i);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long i = base + u * kBlock;
    if (i < n2) {
      vr[u] -= alpha * vz[u];
      stv(r + i, vr[u]);
      acc += vr[u].x * vr[u].x;
      acc += vr[u].y * vr[u].y;
    }
  }
  if (RED == 1) {
    const double s = block_sum_lds(acc, lds4);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
  } else if (RED == 2) {
    const double s = wave_sum_to_lane63(acc);
    if ((threadIdx.x & 63) == 63) partials[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
  } else if (RED == 3) {
    const double s = wave_sum_to_lane63(acc);
    if ((threadIdx.x & 63) == 63) lds4[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
  }
}

template <int U, bool NTLD, bool NTST>
__global__ __launch_bounds__(kBlock) void cg_xp_k(long n2, const double *scal, double2v *__restrict__ x,
                                                  double2v *__restrict__ p, const double2v *__restrict__ r) {
  const double alpha = scal[0], beta = scal[1];
  const long base = (long)blockIdx.x * (kBlock * U) + threadIdx.x;
  double2v vx[U], vp[U], vr[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long i = base + u * kBlock;
    if (i < n2) {
      vx[u] = NTLD ? ldv(x + i) : x[i], vp[u] = NTLD ? ldv(p + i) : p[i], vr[u] = NTLD ? ldv(r + i) : r[i];
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long i = base + u * kBlock;
    if (i < n2) {
      vx[u] += alpha * vp[u];
      const double2v np = vr[u] + beta * vp[u];
      if (NTST) stv(x + i, vx[u]), stv(p + i, np);
      else x[i] = vx[u], p[i] = np;
    }
  }
}

static hipEvent_t e0, e1;
template <class F>
static double timed(F &&launch, int reps) {
  for (int i = 0; i < 3; ++i) launch(i);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) launch(i);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps * 1e3;  // us
}

int main() {
  const long n = 256L * 256 * 256, n2 = n / 2;
  hipEventCreate(&e0), hipEventCreate(&e1);
  double *buf[9], *partials, *scal;
  for (auto &b : buf) hipMalloc(&b, n * 8), hipMemset(b, 0, n * 8);
  hipMalloc(&partials, 8 * 65536 * 4);
  hipMalloc(&scal, 64);
  const double hs[3] = {1.0, 2.0, 0.0};
  hipMemcpy(scal, hs, 24, hipMemcpyHostToDevice);
  const int reps = 30;
#define CGR(U, RED, SCAL)                                                                                          \
  printf("cg_r  U=%d RED=%d SCAL=%d : %7.1f us  %5.0f GB/s\n", U, RED, SCAL,                                       \
         t = timed([&](int i) { hipLaunchKernelGGL((cg_r_k<U, RED, SCAL>), dim3((n2 + kBlock * U - 1) / (kBlock * U)), \
                                                   dim3(kBlock), 0, 0, n2, scal, (double2v *)buf[3 * (i % 3)],       \
                                                   (const double2v *)buf[3 * (i % 3) + 1], partials); }, reps),      \
         24.0 * n / t * 1e-3)
  double t;
  CGR(4, 0, false); CGR(4, 1, false); CGR(4, 1, true); CGR(4, 2, true); CGR(4, 3, true);
  CGR(8, 0, false); CGR(8, 1, true); CGR(8, 2, true); CGR(8, 3, true);
  CGR(2, 1, true); CGR(2, 2, true);
#define CGXP(U, NTLD, NTST)                                                                                          \
  printf("cg_xp U=%d NTLD=%d NTST=%d : %7.1f us  %5.0f GB/s\n", U, NTLD, NTST,                                       \
         t = timed([&](int i) { hipLaunchKernelGGL((cg_xp_k<U, NTLD, NTST>), dim3((n2 + kBlock * U - 1) / (kBlock * U)), \
                                                   dim3(kBlock), 0, 0, n2, scal, (double2v *)buf[3 * (i % 3)],         \
                                                   (double2v *)buf[3 * (i % 3) + 1], (const double2v *)buf[3 * (i % 3) + 2]); }, reps), \
         40.0 * n / t * 1e-3)
  CGXP(4, true, true); CGXP(2, true, true); CGXP(8, true, true); CGXP(4, false, true); CGXP(4, true, false);
  CGXP(4, false, false); CGXP(2, false, true); CGXP(1, true, true);
  return 0;
}
