// What does a kernel boundary cost behind a kernel that wrote B bytes?  (scripts/micro: measurement only)
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/boundary.hip -o gpurun_out/boundary && rocprofv3 --kernel-trace ... gpurun_out/boundary
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k_write(float* out, size_t n4, int spin) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  f32x4 v = {(float)t, 1.0f, 2.0f, 3.0f};
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;   // some compute so the kernel lasts a while
  for (size_t k = t; k < n4; k += (size_t)gridDim.x * blockDim.x) {
    float* p = out + 4 * k;
    if (MODE == 0) *(f32x4*)p = v;
    else if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" ::"v"(p), "v"(v));
    else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 0" ::"v"(p), "v"(v));
    else asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 0" ::"v"(p), "v"(v));
  }
}
__global__ void k_tiny_after_plain(float* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.0f; }
__global__ void k_tiny_after_sc1(float* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.0f; }
__global__ void k_tiny_after_sc0sc1(float* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.0f; }
__global__ void k_tiny_after_nt(float* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.0f; }
__global__ void k_tiny_after_tiny(float* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.0f; }
__global__ __launch_bounds__(256) void k_read_after(const float* in, size_t n4, float* out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  for (size_t k = t; k < n4; k += (size_t)gridDim.x * blockDim.x) acc += *(const f32x4*)(in + 4 * k);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.0f;
}
int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 32;
  const size_t n4 = mb * 1024 * 1024 / 16;
  float *buf, *x;
  hipMalloc(&buf, n4 * 16); hipMalloc(&x, 4096);
  hipMemset(x, 0, 4096);
  hipStream_t s; hipStreamCreate(&s);
  for (int rep = 0; rep < 30; ++rep) {
    hipLaunchKernelGGL(k_write<0>, dim3(2048), dim3(256), 0, s, buf, n4, 200);
    hipLaunchKernelGGL(k_tiny_after_plain, dim3(1), dim3(64), 0, s, x);
    hipLaunchKernelGGL(k_tiny_after_tiny, dim3(1), dim3(64), 0, s, x);
    hipLaunchKernelGGL(k_write<1>, dim3(2048), dim3(256), 0, s, buf, n4, 200);
    hipLaunchKernelGGL(k_tiny_after_sc1, dim3(1), dim3(64), 0, s, x);
    hipLaunchKernelGGL(k_tiny_after_tiny, dim3(1), dim3(64), 0, s, x);
    hipLaunchKernelGGL(k_write<2>, dim3(2048), dim3(256), 0, s, buf, n4, 200);
    hipLaunchKernelGGL(k_tiny_after_sc0sc1, dim3(1), dim3(64), 0, s, x);
    hipLaunchKernelGGL(k_tiny_after_tiny, dim3(1), dim3(64), 0, s, x);
    hipLaunchKernelGGL(k_write<3>, dim3(2048), dim3(256), 0, s, buf, n4, 200);
    hipLaunchKernelGGL(k_tiny_after_nt, dim3(1), dim3(64), 0, s, x);
    hipLaunchKernelGGL(k_tiny_after_tiny, dim3(1), dim3(64), 0, s, x);
    // and a reader behind each flavour
    hipLaunchKernelGGL(k_write<0>, dim3(2048), dim3(256), 0, s, buf, n4, 200);
    hipLaunchKernelGGL(k_read_after, dim3(2048), dim3(256), 0, s, buf, n4, x);
    hipLaunchKernelGGL(k_write<1>, dim3(2048), dim3(256), 0, s, buf, n4, 200);
    hipLaunchKernelGGL(k_read_after, dim3(2048), dim3(256), 0, s, buf, n4, x);
  }
  hipStreamSynchronize(s);
  printf("done %zu MB\n", mb);
  return 0;
}
