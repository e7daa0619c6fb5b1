// Wall time per dependent launch on one stream, unprofiled: tiny kernels, and 256-workgroup kernels of 512 threads.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_tiny(float* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.0f; }
__global__ __launch_bounds__(512) void k_wide(float* x, int spin) {
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.0f) x[blockIdx.x] = v;
}
template <class F> static double timeit(hipStream_t s, int n, F f) {
  for (int i = 0; i < 50; ++i) f();
  (void)hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) f();
  (void)hipStreamSynchronize(s);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
int main() {
  float* x; (void)hipMalloc(&x, 1 << 20);
  hipStream_t s; (void)hipStreamCreate(&s);
  const int n = 3000;
  printf("tiny kernel, 1 block x 64:            %.2f us per launch\n", timeit(s, n, [&] { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, x); }));
  printf("256 blocks x 512, no work:            %.2f us per launch\n", timeit(s, n, [&] { hipLaunchKernelGGL(k_wide, dim3(256), dim3(512), 0, s, x, 0); }));
  printf("2560 blocks x 256 (k_tiny body):      %.2f us per launch\n", timeit(s, n, [&] { hipLaunchKernelGGL(k_tiny, dim3(2560), dim3(256), 0, s, x); }));
  for (int spin : {1000, 4000, 16000}) {
    const double a = timeit(s, n, [&] { hipLaunchKernelGGL(k_wide, dim3(256), dim3(512), 0, s, x, spin); });
    const double b = timeit(s, n, [&] { hipLaunchKernelGGL(k_wide, dim3(256), dim3(512), 0, s, x, spin); hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, x); });
    const double c = timeit(s, n, [&] { hipLaunchKernelGGL(k_wide, dim3(256), dim3(512), 0, s, x, spin); hipLaunchKernelGGL(k_tiny, dim3(2560), dim3(256), 0, s, x); });
    printf("spin %5d: wide alone %.2f us; wide + tiny %.2f us (+%.2f); wide + 2560-block tiny %.2f us (+%.2f)\n", spin, a, b, b - a, c, c - a);
  }
  return 0;
}
