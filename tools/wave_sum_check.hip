// Is wave_sum_down() (csrc/wave_device.hpp: permlane swaps + DPP row shifts) bit-identical, in lane 0, to the
// __shfl_down tree it replaces?  hipcc --offload-arch=gfx950 -O3 -Istormruler_amd/csrc tools/wave_sum_check.hip -o tools/wave_sum_check
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "wave_device.hpp"

__global__ void check_kernel(const double *in, double *out_ref, double *out_new, double *out_all) {
  double v = in[blockIdx.x * 64 + threadIdx.x];
  double a = v;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
  const double b = storm::wave_sum_down(v);
  const double c = storm::wave_sum_all(v);
  if (threadIdx.x == 0) out_ref[blockIdx.x] = a, out_new[blockIdx.x] = b;
  out_all[blockIdx.x * 64 + threadIdx.x] = c;
}

int main() {
  const int waves = 4096;
  std::vector<double> h(waves * 64);
  std::mt19937_64 rng(42);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  for (size_t i = 0; i < h.size(); ++i) h[i] = u(rng) * std::pow(10.0, (double)(rng() % 12) - 6.0);
  double *d_in, *d_a, *d_b, *d_c;
  hipMalloc((void **)&d_in, h.size() * 8), hipMalloc((void **)&d_a, waves * 8), hipMalloc((void **)&d_b, waves * 8);
  hipMalloc((void **)&d_c, h.size() * 8);
  hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check_kernel, dim3(waves), dim3(64), 0, 0, d_in, d_a, d_b, d_c);
  std::vector<double> a(waves), b(waves), c(h.size());
  hipMemcpy(a.data(), d_a, waves * 8, hipMemcpyDeviceToHost), hipMemcpy(b.data(), d_b, waves * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), d_c, h.size() * 8, hipMemcpyDeviceToHost);
  int bad = 0, bad_all = 0;
  for (int w = 0; w < waves; ++w) {
    if (std::memcmp(&a[w], &b[w], 8) != 0) ++bad;
    for (int l = 0; l < 64; ++l)
      if (std::memcmp(&a[w], &c[w * 64 + l], 8) != 0) ++bad_all;
  }
  printf("wave_sum_down: %d of %d waves differ from the __shfl_down tree; wave_sum_all: %d of %d lanes differ\n", bad, waves, bad_all,
         waves * 64);
  return bad || bad_all ? 1 : 0;
}
