// Shared by blas1.hip and krylov.hip: the accumulation loop of the multi-dot kernels (ONE definition, so that the
// partial sums -- and with them every reduction's bits -- do not depend on which kernel ran the loop).
#pragma once
#include <type_traits>

#include "common.hpp"
#include "wave_device.hpp"

namespace storm {

typedef double double2v __attribute__((ext_vector_type(2)));
// Non-temporal access of the streaming kernels.  A select between a non-temporal and a plain access of the same address
// (`nt ? __builtin_nontemporal_load(p) : *p`) is folded by the compiler into ONE plain access -- rounds 1-3 shipped the
// BLAS-1 and solver kernels without a single `nt` instruction although option blas1_nt was on (found in round 4 from the
// ISA; tools/multi_stream_bench.hip).  So the choice is made at compile time: a kernel wraps its streaming loop in
//   nt_dispatch(nt, [&](auto nt) { ... ld2(p, nt) ... st2(p, v, nt) ... });
// which instantiates the loop twice and takes one uniform branch per kernel; inside, `nt` is a std::true_type /
// std::false_type and picks the overload.  Which one a launch gets: stream_nt() (common.hpp) -- non-temporal when the
// vectors are too long to be served by the Infinity Cache between kernels (measured: 256^3 +3 ... +9 %, 160^3 and
// smaller -3 ... -7 % with non-temporal accesses).
__device__ __forceinline__ double2v ld2(const double2v *p, std::true_type) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ double2v ld2(const double2v *p, std::false_type) { return *p; }
__device__ __forceinline__ void st2(double2v *p, double2v v, std::true_type) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st2(double2v *p, double2v v, std::false_type) { *p = v; }
template <class NT>
__device__ __forceinline__ double2v ldv(const double2v *p, NT nt) { return ld2(p, nt); }
template <class NT>
__device__ __forceinline__ void stv(double2v *p, double2v v, NT nt) { st2(p, v, nt); }
template <class Body>
__device__ __forceinline__ void nt_dispatch(int nt, Body &&body) {  // bit 0 of nt decides (other bits: the caller's)
  if (nt & 1) body(std::true_type{});
  else body(std::false_type{});
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

constexpr int kDotChunk = 8;
struct DotPtrs {
  const double *b[kDotChunk];
};

// acc[j] += this thread's share of <a, bs.b[j]>, j < KB.
template <int KB>
__device__ __forceinline__ void multi_dot_accumulate(int64_t n, const double *__restrict__ a, const DotPtrs &bs, int nt,
                                                     double (&acc)[KB]) {
  // nt: bit 0 = non-temporal loads, bit 1 = blocks dealt out from the far end (a block keeps its rows)
  const unsigned bx = (nt & 2) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  nt &= 1;
  const int64_t n2 = n >> 1;
  const double2v *__restrict__ a2 = reinterpret_cast<const double2v *>(a);
  constexpr int U = KB <= 2 ? kUnroll : (KB <= 4 ? 2 : 1);  // many streams: few accesses per stream in flight (tools/cg_kernels_bench.hip)
  const bool same_vector = KB == 1 && bs.b[0] == a;
  nt_dispatch(nt, [&](auto nt) {
  for (int64_t base = (int64_t)bx * (kBlock * kUnroll) + threadIdx.x; base < n2;
       base += (int64_t)gridDim.x * (kBlock * kUnroll)) {
#pragma unroll
    for (int u0 = 0; u0 < kUnroll; u0 += U) {
      double2v va[U], vb[U][KB];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = base + (u0 + u) * kBlock;
        if (i < n2) {
          va[u] = ld2(a2 + i, nt);
#pragma unroll
          for (int j = 0; j < KB; ++j) {
            // (<a, a>: one load, not two of the same address -- the same operands, the same sum)
            if (KB == 1 && same_vector) vb[u][j] = va[u];
            else vb[u][j] = ld2(reinterpret_cast<const double2v *>(bs.b[j]) + i, nt);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = base + (u0 + u) * kBlock;
        if (i < n2) {
#pragma unroll
          for (int j = 0; j < KB; ++j) {
            acc[j] += va[u].x * vb[u][j].x;
            acc[j] += va[u].y * vb[u][j].y;
          }
        }
      }
    }
  }
  });
  if ((n & 1) && bx == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < KB; ++j) acc[j] += a[n - 1] * bs.b[j][n - 1];
  }
}

}  // namespace storm
