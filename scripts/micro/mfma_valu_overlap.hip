// Do v_mfma_f32_16x16x4_f32 (exact fp32, what k_fwd_bwd's fp32 variant uses) and plain vector instructions of the OTHER
// wavefront of a SIMD overlap?  One workgroup of 8 wavefronts per CU: wavefronts w and w + 4 share a SIMD.  Role A (waves 0-3)
// issues NM MFMAs over four independent accumulators, role B (waves 4-7) NV vector instructions (v_fma_f32, or packed
// v_pk_fma_f32) over eight independent registers.  Cycles (s_memtime) of the slower role for: A alone, B alone, both.
//   both ~ max(A, B): the pipes overlap across wavefronts;  both ~ A + B: they share the SIMD's issue / ALUs.
// The same with v_mfma_f32_16x16x16_bf16 for role A.  hipcc --offload-arch=gfx950 -O3 -o ab_run/mfma_valu_overlap ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KIND>   // role A: 0 = f32 16x16x4, 1 = bf16 16x16x16
__device__ __forceinline__ void role_mfma(int n, f32x4 (&acc)[4], float x) {
  const s16x4 bx = {(short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80};
  for (int i = 0; i < n; i += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bx, bx, acc[j], 0, 0, 0);
    }
  }
}
template <int KIND>   // role B: 0 = v_fma_f32, 1 = v_pk_fma_f32, 2 = v_exp_f32, 3 = v_mov_b32_dpp-ish adds
__device__ __forceinline__ void role_valu(int n, float (&v)[8], float x) {
  for (int i = 0; i < n; i += 8) {
    if constexpr (KIND == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j]) : "v"(x));
    } else if constexpr (KIND == 1) {
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        f32x2 t = {v[j], v[j + 1]}, xx = {x, x};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(t) : "v"(xx));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(t) : "v"(xx));
        v[j] = t[0]; v[j + 1] = t[1];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
    }
  }
}

template <int MK, int VK>
__global__ __launch_bounds__(512) void k(int nm, int nv, int mode, unsigned long long* out, float* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool roleA = wave < 4;
  f32x4 acc[4];
  float v[8];
  const float x = 1.0f + 1e-7f * lane;
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (f32x4)(0.0f);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = x + j;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (roleA) { if (mode & 1) role_mfma<MK>(nm, acc, x); }
  else { if (mode & 2) role_valu<VK>(nv, v, x); }
  asm volatile("s_nop 0" ::: "memory");
  float s = 0.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += v[j];
  asm volatile("" : "+v"(s));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

template <int MK, int VK>
static void run(const char* what, int nm, int nv) {
  unsigned long long* d; float* sink;
  const int grid = 256;
  (void)hipMalloc(&d, grid * 8 * sizeof(*d));
  (void)hipMalloc(&sink, 512 * sizeof(float));
  double res[4] = {0, 0, 0, 0};
  for (int mode = 1; mode <= 3; ++mode) {
    std::vector<unsigned long long> h(grid * 8);
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL((k<MK, VK>), dim3(grid), dim3(512), 0, 0, nm, nv, mode, d, sink);
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h.data(), d, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    // per workgroup: the slower role's slowest wavefront; median over workgroups
    std::vector<double> w;
    for (int b = 0; b < grid; ++b) {
      unsigned long long m = 0;
      for (int x = 0; x < 8; ++x) m = std::max(m, h[b * 8 + x]);
      w.push_back((double)m);
    }
    std::sort(w.begin(), w.end());
    res[mode] = w[w.size() / 2];
  }
  printf("%-44s nm %5d nv %5d | MFMA alone %8.0f (%.1f cyc each) | VALU alone %8.0f (%.2f cyc each) | both %8.0f | max %.0f sum %.0f -> overlap %.0f %%\n",
         what, nm, nv, res[1], res[1] / nm, res[2], res[2] / nv, res[3], std::max(res[1], res[2]), res[1] + res[2],
         100.0 * (res[1] + res[2] - res[3]) / std::min(res[1], res[2]));
  (void)hipFree(d); (void)hipFree(sink);
}

int main() {
  // sized so that each role alone takes about the same time (f32 MFMA 32 cycles, v_fma 4, v_pk_fma 8?, v_exp 8)
  run<0, 0>("f32 16x16x4 MFMA  |  v_fma_f32", 1024, 8192);
  run<0, 1>("f32 16x16x4 MFMA  |  v_pk_fma_f32", 1024, 4096);
  run<0, 2>("f32 16x16x4 MFMA  |  v_exp_f32", 1024, 4096);
  run<1, 0>("bf16 16x16x16 MFMA |  v_fma_f32", 2048, 8192);
  run<1, 1>("bf16 16x16x16 MFMA |  v_pk_fma_f32", 2048, 4096);
  run<0, 0>("f32 MFMA | v_fma_f32 (VALU = half)", 1024, 4096);
  run<0, 0>("f32 MFMA | v_fma_f32 (VALU = double)", 1024, 16384);
  return 0;
}
