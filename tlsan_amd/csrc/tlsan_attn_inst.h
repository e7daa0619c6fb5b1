// tlsan_attn_inst.h -- per-D instantiation + launcher of k_fwd_bwd (one translation unit per D
// so the big kernels compile in parallel).
#pragma once
#include "tlsan_attn.h"

template <int D, int DH>
static size_t fwd_smem_bytes(bool train) {
  using G = Geo<D, DH>;
  return sizeof(float) * ((train ? G::NSB * G::PSTR : 0) + 2 * G::NSB * G::LSTR + G::NW * 4 + G::NSB * 2 * TLSAN_LS_MAX + (G::USE_SW ? 2 * (2 * DH * DH + 2 * DH) : 0) + G::NW * G::WSCR);
}

template <int D, int DH>
static hipError_t launch_fwd_bwd_impl(bool train, const FwdArgs& a, int grid, hipStream_t st) {
  const size_t smem = fwd_smem_bytes<D, DH>(train);
  if (train) {
    auto k = k_fwd_bwd<D, DH, true>;
    if (smem > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, st, a);
  } else {
    auto k = k_fwd_bwd<D, DH, false>;
    if (smem > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem, st, a);
  }
  return hipGetLastError();
}

