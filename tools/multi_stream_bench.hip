// Micro-benchmark: a kernel that reads w and K more vectors at the same index and writes w back (the shape of the
// Gram-Schmidt passes, multi-dot and multi-axpy) -- what decides its HBM rate on MI355X?
//   * SKEW  : bytes between the vectors' base addresses beyond the vector length (0 = the same alignment for all)
//   * shape : trips per thread (U sequential trips of one 16-byte access per stream) and blocks
//   * one thread block = 256 threads
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -D__HIP_PLATFORM_AMD__ -Istormruler_amd/csrc -Iinclude tools/multi_stream_bench.hip -o tools/multi_stream_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <vector>
#include "common.hpp"
#include "blas1_device.hpp"
using storm::double2v;



// ---- the library's own kernels (text copied from csrc/solvers.hip and csrc/blas1.hip when this tool was last edited: keep
// in step) ----------------------------------------------------------------------------------------------------------
#include "ticket_device.hpp"
#include "solver_device.hpp"
#include "wave_device.hpp"
namespace storm {
__global__ __launch_bounds__(kBlock) void mgs_pair_kernel(int64_t n, const int *done, double *__restrict__ w,
                                                          const double *ha, const double *hb,
                                                          const double *__restrict__ qa, const double *__restrict__ qb,
                                                          const double *__restrict__ qc, const double *__restrict__ qd,
                                                          double *out_c, double *out_d, TicketArgs tickets, int nt) {
  if (done && *done) return;
  __shared__ double lds4[4];
  const double va = qa ? *ha : 0.0, vb = qb ? *hb : 0.0;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  const int64_t n2 = n >> 1;
  double2v *w2 = reinterpret_cast<double2v *>(w);
  const double2v *a2 = reinterpret_cast<const double2v *>(qa), *b2 = reinterpret_cast<const double2v *>(qb);
  const double2v *c2 = reinterpret_cast<const double2v *>(qc), *d2 = reinterpret_cast<const double2v *>(qd);
  nt_dispatch(nt, [&](auto nt) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
    double2v vw = ldv(w2 + i, nt), xa = {0.0, 0.0}, xb = {0.0, 0.0}, xc = {0.0, 0.0}, xd = {0.0, 0.0};
    if (qa) xa = ldv(a2 + i, nt);
    if (qb) xb = ldv(b2 + i, nt);
    if (qc) xc = ldv(c2 + i, nt);
    if (qd) xd = ldv(d2 + i, nt);
    if (qa) {
      vw -= va * xa;
      if (qb) vw -= vb * xb;
      stv(w2 + i, vw, nt);
    }
    if (qc) {
      s0 += vw.x * xc.x, s0 += vw.y * xc.y;
      if (qd) s1 += vw.x * xd.x, s1 += vw.y * xd.y, s2 += xc.x * xd.x, s2 += xc.y * xd.y;
    } else {
      s0 += vw.x * vw.x, s0 += vw.y * vw.y;
    }
  }
  });
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    double vw = w[i];
    if (qa) {
      vw -= va * qa[i];
      if (qb) vw -= vb * qb[i];
      w[i] = vw;
    }
    if (qc) {
      s0 += vw * qc[i];
      if (qd) s1 += vw * qd[i], s2 += qc[i] * qd[i];
    } else {
      s0 += vw * vw;
    }
  }
  const double mine[3] = {block_sum256(s0, lds4), block_sum256(s1, lds4), block_sum256(s2, lds4)};
  if (threadIdx.x >= kWave) return;
  double total[3];
  if (ticket_reduce_wave0<3>(tickets, mine, qd ? 3 : 1, blockIdx.x, gridDim.x, total) && threadIdx.x == 0) {
    *out_c = total[0];
    if (qd) *out_d = total[1] - total[0] * total[2];
  }
}

template <int T>
struct MgsMultiArgs {
  const double *h[T];   // coefficients of the vectors to subtract (device, finished by the previous pass)
  const double *qa[T];  // the vectors to subtract
  const double *qc[T];  // the vectors to project on next
  double *out[T];       // where their coefficients go (nc == 0: out[0] = the norm's square)
  int na, nc;
};
template <int T, int TRIPS>
__global__ __launch_bounds__(kBlock) void mgs_multi_kernel(int64_t n, const int *done, double *__restrict__ w, MgsMultiArgs<T> a,
                                                           TicketArgs tickets, int nt) {
  if (done && *done) return;
  constexpr int NG = T * (T - 1) / 2, NV = T + NG;
  __shared__ double lds[4][NV];
  double hv[T];
#pragma unroll
  for (int j = 0; j < T; ++j) hv[j] = j < a.na ? *a.h[j] : 0.0;
  double acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.0;
  const int64_t n2 = n >> 1;
  double2v *w2 = reinterpret_cast<double2v *>(w);
  auto fold = [&](double wx, double wy, const double2v (&xc)[T]) {
    if (a.nc == 0) {
      acc[0] += wx * wx, acc[0] += wy * wy;
      return;
    }
    int g = T;
#pragma unroll
    for (int j = 0; j < T; ++j) {
      acc[j] += wx * xc[j].x, acc[j] += wy * xc[j].y;
#pragma unroll
      for (int i = 0; i < j; ++i, ++g) acc[g] += xc[i].x * xc[j].x, acc[g] += xc[i].y * xc[j].y;
    }
  };
  nt_dispatch(nt, [&](auto nt) {
  for (int64_t base = (int64_t)blockIdx.x * (kBlock * TRIPS) + threadIdx.x; base < n2;
       base += (int64_t)gridDim.x * (kBlock * TRIPS)) {
#pragma unroll
    for (int u = 0; u < TRIPS; ++u) {
      const int64_t i = base + u * kBlock;
      if (i >= n2) break;
      double2v vw = ldv(w2 + i, nt), xa[T], xc[T];
#pragma unroll
      for (int j = 0; j < T; ++j) {
        xa[j] = double2v{0.0, 0.0}, xc[j] = double2v{0.0, 0.0};
        if (j < a.na) xa[j] = ldv(reinterpret_cast<const double2v *>(a.qa[j]) + i, nt);
        if (j < a.nc) xc[j] = ldv(reinterpret_cast<const double2v *>(a.qc[j]) + i, nt);
      }
      if (a.na > 0) {
#pragma unroll
        for (int j = 0; j < T; ++j)
          if (j < a.na) vw -= hv[j] * xa[j];
        stv(w2 + i, vw, nt);
      }
      fold(vw.x, vw.y, xc);
    }
  }
  });
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    double vw = w[i];
    double2v xc[T];
#pragma unroll
    for (int j = 0; j < T; ++j) {
      if (j < a.na) vw -= hv[j] * a.qa[j][i];
      xc[j] = double2v{j < a.nc ? a.qc[j][i] : 0.0, 0.0};
    }
    if (a.na > 0) w[i] = vw;
    fold(vw, 0.0, xc);
  }
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const double s = wave_sum_to_lane63(acc[v]);  // (DPP: ten shuffle trees through the LDS crossbar cost ~1 us per wave)
    if (lane == kWave - 1) lds[wave][v] = s;
  }
  __syncthreads();
  if (threadIdx.x >= kWave) return;
  double mine[NV], total[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) mine[v] = (lds[0][v] + lds[1][v]) + (lds[2][v] + lds[3][v]);
  const int nv = a.nc == 0 ? 1 : NV;
  if (ticket_reduce_wave0<NV>(tickets, mine, nv, blockIdx.x, gridDim.x, total) && threadIdx.x == 0) {
    if (a.nc == 0) {
      *a.out[0] = total[0];
      return;
    }
    double hn[T];
    int g = T;
#pragma unroll
    for (int j = 0; j < T; ++j) {
      double v = total[j];
#pragma unroll
      for (int i = 0; i < j; ++i, ++g) v -= hn[i] * total[g];
      hn[j] = v;
      if (j < a.nc) *a.out[j] = v;
    }
  }
}

// (csrc/blas1.hip)
// Sum over the 256 threads of a block, fixed order; result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double *lds4) {
  v = wave_sum_down(v);  // (the __shfl_down tree's order and bits, without the LDS crossbar: wave_device.hpp)
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  __syncthreads();  // lds4 may still be read by a previous call
  if (lane == 0) lds4[wave] = v;
  __syncthreads();
  return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

// KB sums at once, ONE pair of barriers: the same wave trees and the same (w0 + w1) + (w2 + w3) as block_sum, so the same
// bits; sums[j] valid in every thread.
template <int KB>
__device__ __forceinline__ void block_sum_multi(const double (&v)[KB], double (*lds)[4], double (&sums)[KB]) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  double w[KB];
#pragma unroll
  for (int j = 0; j < KB; ++j) w[j] = wave_sum_down(v[j]);
  __syncthreads();  // (the buffer may still be read by a previous call)
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < KB; ++j) lds[j][wave] = w[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < KB; ++j) sums[j] = (lds[j][0] + lds[j][1]) + (lds[j][2] + lds[j][3]);
}

template <int KB>
__global__ __launch_bounds__(kBlock) void multi_dot_kernel(int64_t n, const double *__restrict__ a,
                                                           DotPtrs bs, double *__restrict__ partials,
                                                           const int *done, int nt) {
  if (done && *done) return;
  __shared__ double lds[KB][4];
  double acc[KB], sums[KB];
#pragma unroll
  for (int j = 0; j < KB; ++j) acc[j] = 0.0;
  multi_dot_accumulate<KB>(n, a, bs, nt, acc);
  const unsigned bx = (nt & 2) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  block_sum_multi<KB>(acc, lds, sums);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < KB; ++j) partials[(int64_t)j * gridDim.x + bx] = sums[j];
  }
}

// The same with the reduction finished in the kernel (ticket_device.hpp): out[j] = <a, bs.b[j]>, one launch.
// host_words != null: the last block also stores the sums into pinned HOST memory, each as two self-validating words
// { tag | low half }, { tag | high half } (one atomic system-scope store each: no ordering between them and a flag to
// rely on) -- the host polls them instead of copying and waiting on the stream.
template <int KB>
__global__ __launch_bounds__(kBlock) void multi_dot_ticket_kernel(int64_t n, const double *__restrict__ a, DotPtrs bs,
                                                                  TicketArgs tickets, double *__restrict__ out,
                                                                  const int *done, int nt,
                                                                  unsigned long long *host_words, unsigned tag) {
  if (done && *done) return;
  __shared__ double lds[KB][4];
  double acc[KB];
#pragma unroll
  for (int j = 0; j < KB; ++j) acc[j] = 0.0;
  multi_dot_accumulate<KB>(n, a, bs, nt, acc);
  double mine[KB], total[KB];
  block_sum_multi<KB>(acc, lds, mine);
  if (threadIdx.x >= kWave) return;
  const unsigned bx = (nt & 2) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  if (ticket_reduce_wave0<KB>(tickets, mine, KB, bx, gridDim.x, total) && threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < KB; ++j) {
      out[j] = total[j];
      if (host_words) {
        const unsigned long long t = (unsigned long long)tag << 32;
        __hip_atomic_store(host_words + 2 * j, t | (unsigned)__double2loint(total[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_words + 2 * j + 1, t | (unsigned)__double2hiint(total[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}


}  // namespace storm

template <int K>
struct Ptrs {
  const double2v *q[K];
};

// MODE 0: U sequential trips (loads of trip u + 1 after the arithmetic of trip u); MODE 1: all loads of U trips first
// PRO: the solver kernels' prologue (a device flag read before anything else); EPI: their epilogue (NV block sums through LDS,
// published with returning atomic exchanges, a ticket drawn with a returning atomic add, the last of 64 folds the group)
template <int K, int U, int MODE, bool WRITE, bool PRO = false, int EPI = 0>
__global__ __launch_bounds__(256) void multi_k(double2v *__restrict__ w, Ptrs<K> p, long n2, double *out, const int *flag = nullptr,
                                               double *part = nullptr, int *cnt = nullptr) {
  if (PRO && flag && *flag) return;
  __shared__ double lds[4][12];
  double acc = 0.0;
  for (long base = (long)blockIdx.x * (256 * U) + threadIdx.x; base < n2; base += (long)gridDim.x * (256 * U)) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long i = base + u * 256;
        if (i >= n2) break;
        double2v vw = __builtin_nontemporal_load(w + i), x[K];
#pragma unroll
        for (int j = 0; j < K; ++j) x[j] = __builtin_nontemporal_load(p.q[j] + i);
#pragma unroll
        for (int j = 0; j < K; ++j) vw -= 1e-9 * x[j], acc += vw.x * x[j].x + vw.y * x[j].y;
        if (WRITE) __builtin_nontemporal_store(vw, w + i);
      }
    } else {
      double2v vw[U], x[U][K];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long i = base + u * 256;
        if (i < n2) {
          vw[u] = __builtin_nontemporal_load(w + i);
#pragma unroll
          for (int j = 0; j < K; ++j) x[u][j] = __builtin_nontemporal_load(p.q[j] + i);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long i = base + u * 256;
        if (i < n2) {
#pragma unroll
          for (int j = 0; j < K; ++j) vw[u] -= 1e-9 * x[u][j], acc += vw[u].x * x[u][j].x + vw[u].y * x[u][j].y;
          if (WRITE) __builtin_nontemporal_store(vw[u], w + i);
        }
      }
    }
  }
  if (EPI > 0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < EPI; ++v) {
      double sv = acc * (v + 1);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) sv += __shfl_down(sv, off, 64);
      if (lane == 0) lds[wave][v] = sv;
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    int go = 0;
    if (lane == 0) {
      unsigned long long seen = 0;
#pragma unroll
      for (int v = 0; v < EPI; ++v)
        seen ^= __hip_atomic_exchange(reinterpret_cast<unsigned long long *>(part + (size_t)v * gridDim.x + blockIdx.x),
                                      (unsigned long long)__double_as_longlong((lds[0][v] + lds[1][v]) + (lds[2][v] + lds[3][v])),
                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" : : "v"(seen) : "memory");
      int *c = cnt + (size_t)(1 + blockIdx.x / 64) * 16;
      if (__hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 63) {
        __hip_atomic_store(c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        go = 1;
      }
    }
    go = __shfl(go, 0, 64);
    if (go) {
      double tot = 0.0;
      for (int v = 0; v < EPI; ++v)
        tot += __hip_atomic_load(part + (size_t)v * gridDim.x + (blockIdx.x / 64) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tot == 1.2345e300) *out = tot;
    }
    return;
  }
  if (acc == 1.2345e300) *out = acc;
}

static int *g_flag, *g_cnt;
static int g_pool_vectors = 0, g_rotate_w = 0;
static size_t g_pitch = 0, g_pool_bytes = 0;
static double *g_part;
template <int K, int U, int MODE, bool WRITE, bool PRO = false, int EPI = 0>
static double run(double2v *w, Ptrs<K> p0, long n2, int blocks, double *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int reps = 24;
  // like a Gram-Schmidt sweep: every launch reads the NEXT K vectors of the pool (g_pool_vectors of them, g_pitch apart)
  auto ptrs = [&](int rep) {
    Ptrs<K> p = p0;
    if (g_pool_vectors > 0)
      for (int j = 0; j < K; ++j)
        p.q[j] = reinterpret_cast<const double2v *>(reinterpret_cast<const char *>(w) + (size_t)(1 + (rep * (K + 1) + j) % g_pool_vectors) * g_pitch);
    return p;
  };
  // g_rotate_w: w too is another vector of the pool every launch (nothing is served by the Infinity Cache)
  auto wptr = [&](int rep) {
    if (!g_rotate_w || g_pool_vectors <= 0) return w;
    return reinterpret_cast<double2v *>(reinterpret_cast<char *>(w) + (size_t)(1 + (rep * (K + 1) + K) % g_pool_vectors) * g_pitch);
  };
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((multi_k<K, U, MODE, WRITE, PRO, EPI>), dim3(blocks), dim3(256), 0, 0, wptr(i), ptrs(i), n2, out, g_flag, g_part, g_cnt);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((multi_k<K, U, MODE, WRITE, PRO, EPI>), dim3(blocks), dim3(256), 0, 0, wptr(3 + i), ptrs(3 + i), n2, out, g_flag, g_part, g_cnt);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0), hipEventDestroy(e1);
  const double bytes = 16.0 * n2 * (K + 1 + (WRITE ? 1 : 0));
  return bytes * reps / (ms * 1e-3) / 1e12;
}

template <int K>
static void sweep(char *pool, long n, long skew, double *out) {
  const long n2 = n / 2;
  const long pitch = n * 8 + skew;
  g_pitch = (size_t)pitch;
  double2v *w = reinterpret_cast<double2v *>(pool);
  Ptrs<K> p;
  for (int j = 0; j < K; ++j) p.q[j] = reinterpret_cast<const double2v *>(pool + (j + 1) * pitch);
  const int b1 = (int)((n2 + 255) / 256), b2 = (int)((n2 + 511) / 512), b4 = (int)((n2 + 1023) / 1024);
  printf("K=%d skew=%7ld | rw: 1 trip %.2f  2 trips %.2f  4 trips %.2f  4 trips/2048 blocks %.2f | 2 ahead %.2f  4 ahead %.2f | read-only: 1 trip %.2f 4 trips %.2f 2 ahead %.2f\n",
         K, skew, run<K, 1, 0, true>(w, p, n2, b1, out), run<K, 2, 0, true>(w, p, n2, b2, out), run<K, 4, 0, true>(w, p, n2, b4, out),
         run<K, 4, 0, true>(w, p, n2, 2048, out), run<K, 2, 1, true>(w, p, n2, b2, out), run<K, 4, 1, true>(w, p, n2, b4, out),
         run<K, 1, 0, false>(w, p, n2, b1, out), run<K, 4, 0, false>(w, p, n2, b4, out), run<K, 2, 1, false>(w, p, n2, b2, out));
  printf("      with the solvers' prologue / epilogue | rw 1 trip: plain %.2f  flag %.2f  3 sums %.2f  10 sums %.2f  flag + 3 sums %.2f | rw 4 trips: plain %.2f  flag + 10 sums %.2f | 2 ahead: flag + 3 sums %.2f  flag + 10 sums %.2f\n",
         run<K, 1, 0, true>(w, p, n2, b1, out), run<K, 1, 0, true, true, 0>(w, p, n2, b1, out), run<K, 1, 0, true, false, 3>(w, p, n2, b1, out),
         run<K, 1, 0, true, false, 10>(w, p, n2, b1, out), run<K, 1, 0, true, true, 3>(w, p, n2, b1, out), run<K, 4, 0, true>(w, p, n2, b4, out),
         run<K, 4, 0, true, true, 10>(w, p, n2, b4, out), run<K, 2, 1, true, true, 3>(w, p, n2, b2, out), run<K, 2, 1, true, true, 10>(w, p, n2, b2, out));
  fflush(stdout);
}

static double time_launches(int reps, const std::function<void(int)> &launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch(i);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) launch(3 + i);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 / reps;
}

static void library_kernels(char *pool, long n, double *out) {
  using namespace storm;
  double *w = reinterpret_cast<double *>(pool);
  auto vec = [&](int i) { return reinterpret_cast<const double *>(pool + (size_t)(1 + i % g_pool_vectors) * g_pitch); };
  double *h;
  hipMalloc((void **)&h, 8 * 64), hipMemset(h, 0, 8 * 64);
  double *part1, *part2;
  hipMalloc((void **)&part1, 8 * 12 * 40000), hipMalloc((void **)&part2, 8 * 12 * 2048);
  TicketArgs t{g_cnt, part1, part2};
  const int64_t n2 = n / 2;
  for (int nt : {1, 0}) {
    for (int nbp : {32768, 21845, 16384, 8192}) {
      const double s = time_launches(24, [&](int r) {
        hipLaunchKernelGGL(mgs_pair_kernel, dim3(nbp), dim3(256), 0, 0, (int64_t)n, g_flag, w, h, h + 1, vec(4 * r), vec(4 * r + 1),
                           vec(4 * r + 2), vec(4 * r + 3), h + 2, h + 3, t, nt);
      });
      printf("mgs_pair_kernel (2 subtracted, 2 projected: 6 streams) nt=%d blocks=%5d: %.1f us  %.2f TB/s\n", nt, nbp, s * 1e6, 48.0 * n / s / 1e12);
    }
    auto multi = [&](auto trips_tag, int nbm) {
      constexpr int TRIPS = decltype(trips_tag)::value;
      MgsMultiArgs<4> a{};
      a.na = 4, a.nc = 4;
      const double s = time_launches(24, [&](int r) {
        for (int j = 0; j < 4; ++j) a.h[j] = h + j, a.qa[j] = vec(8 * r + j), a.qc[j] = vec(8 * r + 4 + j), a.out[j] = h + 8 + j;
        hipLaunchKernelGGL((mgs_multi_kernel<4, TRIPS>), dim3(nbm), dim3(256), 0, 0, (int64_t)n, g_flag, w, a, t, nt);
      });
      printf("mgs_multi_kernel<4, %d trips> (4 + 4: 10 streams) nt=%d blocks=%5d: %.1f us  %.2f TB/s\n", TRIPS, nt, nbm, s * 1e6, 80.0 * n / s / 1e12);
    };
    multi(std::integral_constant<int, 4>{}, 8192);
    multi(std::integral_constant<int, 2>{}, 16384);
    multi(std::integral_constant<int, 1>{}, 32768);
  }
  for (int nt : {1, 0})
    for (int nb : {8192, 4096, 2048}) {
      DotPtrs bp;
      const double s8 = time_launches(24, [&](int r) {
        for (int j = 0; j < 8; ++j) bp.b[j] = vec(9 * r + 1 + j);
        hipLaunchKernelGGL(multi_dot_ticket_kernel<8>, dim3(nb), dim3(256), 0, 0, (int64_t)n, vec(9 * r), bp, t, h, g_flag, nt, (unsigned long long *)nullptr, 0u);
      });
      const double s1 = time_launches(24, [&](int r) {
        for (int j = 0; j < 8; ++j) bp.b[j] = vec(2 * r + 1);
        hipLaunchKernelGGL(multi_dot_ticket_kernel<1>, dim3(nb), dim3(256), 0, 0, (int64_t)n, vec(2 * r), bp, t, h, g_flag, nt, (unsigned long long *)nullptr, 0u);
      });
      const double s0 = time_launches(24, [&](int r) {
        for (int j = 0; j < 8; ++j) bp.b[j] = vec(r);
        hipLaunchKernelGGL(multi_dot_ticket_kernel<1>, dim3(nb), dim3(256), 0, 0, (int64_t)n, vec(r), bp, t, h, g_flag, nt, (unsigned long long *)nullptr, 0u);
      });
      printf("multi_dot_ticket_kernel nt=%d blocks=%5d: k=8 %.1f us %.2f TB/s | dot %.1f us %.2f TB/s | norm2 %.1f us %.2f TB/s\n", nt, nb, s8 * 1e6,
             72.0 * n / s8 / 1e12, s1 * 1e6, 16.0 * n / s1 / 1e12, s0 * 1e6, 8.0 * n / s0 / 1e12);
    }
  fflush(stdout);
  (void)out;
}

// Does the DISTANCE between the vectors matter at the megabyte scale (the library's vectors are separate allocations of
// 128 MiB + a little, 2 MiB aligned)?  K + 1 vectors `pitch` apart, all read at the same index, one written.
template <int K>
static void pitch_sweep(char *pool, long n, double *out) {
  const long n2 = n / 2;
  const int b2 = (int)((n2 + 511) / 512);
  const int pool_vectors = g_pool_vectors;
  g_pool_vectors = 0;  // (fixed vectors: the question is their placement)
  std::vector<long> extras = {0L, 2L, 64L};
  for (long m = 1; m <= 40; ++m) extras.push_back(m * 1024);
  for (long m : {48L, 64L, 96L}) extras.push_back(m * 1024);
  for (long extra_kib : extras) {
    const long pitch = n * 8 + extra_kib * 1024;
    if ((size_t)pitch * (K + 1) > g_pool_bytes) continue;
    double2v *w = reinterpret_cast<double2v *>(pool);
    Ptrs<K> p;
    for (int j = 0; j < K; ++j) p.q[j] = reinterpret_cast<const double2v *>(pool + (j + 1) * pitch);
    printf("K=%d pitch = vector + %6ld KiB: rw 2 ahead %.2f  1 trip %.2f | read-only 4 trips %.2f TB/s\n", K, extra_kib,
           run<K, 2, 1, true>(w, p, n2, b2, out), run<K, 1, 0, true>(w, p, n2, 2 * b2, out), run<K, 4, 0, false>(w, p, n2, b2 / 2, out));
    fflush(stdout);
  }
  g_pool_vectors = pool_vectors;
}

int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : (1L << 24);  // doubles per vector (256^3)
  const long max_skew = 1 << 17;
  g_pool_vectors = argc > 2 ? atoi(argv[2]) : 30;  // 0: the same K vectors every launch (the Infinity Cache then serves part of them)
  const int nvec = 1 + (g_pool_vectors > 8 ? g_pool_vectors : 8);
  char *pool;
  double *out;
  const bool contiguous = argc > 4 && atoi(argv[4]) != 0;  // physically contiguous pool (hipDeviceMallocContiguous)
  if (contiguous) {
    if (hipExtMallocWithFlags((void **)&pool, (size_t)(n * 8 + max_skew) * nvec, hipDeviceMallocContiguous) != hipSuccess) return 2;
    printf("(physically contiguous pool)\n");
  } else if (hipMalloc((void **)&pool, (size_t)(n * 8 + max_skew) * nvec) != hipSuccess) {
    return 1;
  }
  hipMalloc((void **)&out, 8);
  hipMalloc((void **)&g_flag, 4), hipMemset(g_flag, 0, 4);
  hipMalloc((void **)&g_cnt, 4 * 16 * 2049), hipMemset(g_cnt, 0, 4 * 16 * 2049);
  hipMalloc((void **)&g_part, 8 * 12 * 40000);
  hipMemset(pool, 0, (size_t)(n * 8 + max_skew) * nvec);
  g_pool_bytes = (size_t)(n * 8 + max_skew) * nvec;
  if (argc > 3 && !strcmp(argv[3], "pitch")) {
    pitch_sweep<6>(pool, n, out);
    return 0;
  }
  g_pitch = (size_t)(n * 8);
  library_kernels(pool, n, out);
  for (long skew : {0L, 0L}) {
    printf("---- w %s\n", g_rotate_w ? "rotates through the pool too (no Infinity Cache reuse at all)" : "is the same vector every launch (as in a Gram-Schmidt sweep)");
    sweep<1>(pool, n, skew, out);
    sweep<2>(pool, n, skew, out);
    sweep<4>(pool, n, skew, out);
    sweep<8>(pool, n, skew, out);
    g_rotate_w = 1;
  }
  return 0;
}
