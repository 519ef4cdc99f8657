// Micro-benchmark: which shape of a streaming copy kernel reaches the HBM rate on MI355X?
// hipcc --offload-arch=gfx950 -O3 tools/stream_bench.hip -o tools/stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double double2v __attribute__((ext_vector_type(2)));

template <bool NT_LD, bool NT_ST, int UNROLL>
__global__ __launch_bounds__(256) void copy_k(const double2v* __restrict__ a, double2v* __restrict__ b, long n2) {
  const long stride = (long)gridDim.x * 256 * UNROLL;
  for (long i = (long)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n2; i += stride) {
    double2v v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (i + u * 256 < n2) v[u] = NT_LD ? __builtin_nontemporal_load(a + i + u * 256) : a[i + u * 256];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      if (i + u * 256 < n2) {
        if (NT_ST) __builtin_nontemporal_store(v[u], b + i + u * 256);
        else b[i + u * 256] = v[u];
      }
  }
}

template <bool NT_LD, bool NT_ST, int UNROLL>
float run(const double* a, double* b, long n, int blocks, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((copy_k<NT_LD, NT_ST, UNROLL>), dim3(blocks), dim3(256), 0, 0, (const double2v*)a, (double2v*)b, n / 2);
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((copy_k<NT_LD, NT_ST, UNROLL>), dim3(blocks), dim3(256), 0, 0, (const double2v*)a, (double2v*)b, n / 2);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return 16.0 * n * reps / (ms * 1e-3) / 1e9;
}

int main() {
  const long n = 1L << 27;  // 1 GiB per buffer
  double *a, *b;
  hipMalloc(&a, n * 8), hipMalloc(&b, n * 8);
  hipMemset(a, 1, n * 8), hipMemset(b, 0, n * 8);
  const int cus = 256;
  for (int bpc : {4, 8, 16, 32, 0}) {
    int blocks = bpc ? cus * bpc : (int)((n / 2 + 255) / 256);
    printf("blocks/CU %2d: plain u1 %.0f  u2 %.0f  u4 %.0f | ntst u2 %.0f  u4 %.0f | ntld+ntst u2 %.0f u4 %.0f | ntld u2 %.0f\n", bpc,
           run<false, false, 1>(a, b, n, blocks, 10), run<false, false, 2>(a, b, n, bpc ? blocks : blocks / 2, 10),
           run<false, false, 4>(a, b, n, bpc ? blocks : blocks / 4, 10), run<false, true, 2>(a, b, n, bpc ? blocks : blocks / 2, 10),
           run<false, true, 4>(a, b, n, bpc ? blocks : blocks / 4, 10), run<true, true, 2>(a, b, n, bpc ? blocks : blocks / 2, 10),
           run<true, true, 4>(a, b, n, bpc ? blocks : blocks / 4, 10), run<true, false, 2>(a, b, n, bpc ? blocks : blocks / 2, 10));
  }
  return 0;
}
