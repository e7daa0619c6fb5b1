// Rate of integer atomics on random rows of a counter table (the index build's k_count and the fused kernel's cursor
// draws), by memory scope and with / without a returned value.  One thread per atomic.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <random>
template <int MODE> __global__ void k_atom(const int* __restrict__ ids, int n, int* cnt, int* sink) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int id = ids[t];
  int r = 0;
  if (MODE == 0) atomicAdd(&cnt[id], 1);                                                                     // device scope, result unused
  if (MODE == 1) r = atomicAdd(&cnt[id], 1);                                                                 // device scope, returning
  if (MODE == 2) __hip_atomic_fetch_add(&cnt[id], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);        // workgroup scope
  if (MODE == 3) r = __hip_atomic_fetch_add(&cnt[id], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);    // ... returning
  if (MODE == 4) __hip_atomic_fetch_add(&cnt[id], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (MODE == 5) cnt[id] = t;                                                                                // plain scattered store
  if (MODE == 6) r = cnt[id];                                                                                // plain scattered load
  if (r == 0x7fffffff) sink[0] = r;
}
template <class F> static double timeit(hipStream_t s, int n, F f) {
  for (int i = 0; i < 10; ++i) f();
  (void)hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) f();
  (void)hipStreamSynchronize(s);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
int main() {
  hipStream_t s; (void)hipStreamCreate(&s);
  const char* names[] = {"device, no return", "device, returning", "workgroup, no return", "workgroup, returning", "agent, no return", "plain store", "plain load"};
  for (int rows : {28589, 5000000}) {
    for (int n : {80000, 200000, 800000}) {
      std::vector<int> h(n);
      std::mt19937 g(1);
      for (auto& v : h) v = g() % rows;
      int *ids, *cnt, *sink;
      (void)hipMalloc(&ids, 4 * n); (void)hipMalloc(&cnt, 4 * (size_t)rows); (void)hipMalloc(&sink, 64);
      (void)hipMemcpy(ids, h.data(), 4 * n, hipMemcpyHostToDevice);
      (void)hipMemset(cnt, 0, 4 * (size_t)rows);
      const dim3 grid((n + 255) / 256), blk(256);
      double base = timeit(s, 200, [&] { hipLaunchKernelGGL(k_atom<7>, grid, blk, 0, s, ids, n, cnt, sink); });
      printf("rows %8d  n %7d  (empty kernel %.1f us)\n", rows, n, base);
#define RUN(M) { double t = timeit(s, 200, [&] { hipLaunchKernelGGL(k_atom<M>, grid, blk, 0, s, ids, n, cnt, sink); }); \
                 printf("   %-22s %7.1f us  %6.2f G/s\n", names[M], t, n / (t - base * 0.0) / 1e3); }
      RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
      (void)hipFree(ids); (void)hipFree(cnt); (void)hipFree(sink);
    }
  }
  return 0;
}
