// Does the 256 MB Infinity Cache keep the END of what a streaming kernel just touched, so that the next kernel
// gains by sweeping the other way?  W: r = r - a z over n rows (reads r, z; writes r), forward.  Then R: sum of r,
// forward or backward (block index reversed).  Prints microseconds of R.
//   hipcc -O3 --offload-arch=gfx950 tools/mall_probe.hip -o tools/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double double2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void w_kernel(int64_t n2, double2v *r, const double2v *z, double a, int reverse) {
  const int64_t b = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  for (int u = 0; u < 4; ++u) {
    const int64_t i = b * 1024 + u * 256 + threadIdx.x;
    if (i < n2) r[i] = r[i] - a * z[i];
  }
}
__global__ __launch_bounds__(256) void r_kernel(int64_t n2, const double2v *r, double *out, int reverse) {
  const int64_t b = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
  double s = 0.0;
  for (int u = 0; u < 4; ++u) {
    const int64_t i = b * 1024 + u * 256 + threadIdx.x;
    if (i < n2) s += r[i].x + r[i].y;
  }
  if (s == 123.456) out[0] = s;
}
int main() {
  const int64_t n = 16777216, n2 = n / 2;
  double2v *r, *z;
  double *out;
  hipMalloc(&r, n * 8), hipMalloc(&z, n * 8), hipMalloc(&out, 8);
  hipMemset(r, 0, n * 8), hipMemset(z, 0, n * 8);
  const int nb = (int)((n2 + 1023) / 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  // cold / warm: the same read right after the write, or after 1 GB of other traffic
  double2v *big;
  hipMalloc(&big, (size_t)1 << 30);
  hipMemset(big, 0, (size_t)1 << 30);
  for (int cold = 0; cold < 2; ++cold) {
    std::vector<float> ts;
    for (int it = 0; it < 30; ++it) {
      hipLaunchKernelGGL(w_kernel, dim3(nb), dim3(256), 0, 0, n2, r, z, 0.5, 0);
      if (cold) hipLaunchKernelGGL(r_kernel, dim3(65536), dim3(256), 0, 0, (int64_t)1 << 26, big, out, 0);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(r_kernel, dim3(nb), dim3(256), 0, 0, n2, r, out, 0);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      ts.push_back(ms * 1000.f);
    }
    std::sort(ts.begin(), ts.end());
    printf("read of r %s: %.1f us (median), %.1f (min)\n", cold ? "after 1 GB of other reads" : "right after it was written",
           ts[ts.size() / 2], ts[0]);
  }
  for (int wrev = 0; wrev < 2; ++wrev)
    for (int rrev = 0; rrev < 2; ++rrev) {
      std::vector<float> ts;
      for (int it = 0; it < 30; ++it) {
        hipLaunchKernelGGL(w_kernel, dim3(nb), dim3(256), 0, 0, n2, r, z, 0.5, wrev);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(r_kernel, dim3(nb), dim3(256), 0, 0, n2, r, out, rrev);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ts.push_back(ms * 1000.f);
      }
      std::sort(ts.begin(), ts.end());
      printf("writer %s, reader %s: %.1f us (median), %.1f (min)  [134 MB read]\n", wrev ? "backward" : "forward",
             rrev ? "backward" : "forward", ts[ts.size() / 2], ts[0]);
    }
  return 0;
}
