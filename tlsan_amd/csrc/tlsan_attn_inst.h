// tlsan_attn_inst.h -- per-D instantiation + launcher of k_fwd_bwd (one translation unit per D
// so the big kernels compile in parallel).
#pragma once
#include <atomic>
#include <hip/hip_ext.h>
#include "tlsan_attn.h"
#define TLSAN_MAX_DEVICES 16   // devices one process may drive (per-device launch attributes below)
struct LaunchEvents { hipEvent_t start, stop; };   // optional time stamps of the dispatch (tlsan_profile_*), or NULLs
#ifndef TLSAN_STAMPS
#define TLSAN_STAMPS 0
#endif

// (mirrors the carve-up at the top of k_fwd_bwd; flat = the FLAT variant of the streamed windows, tlsan_attn.h)
template <int D, int DH, int NWV = 0>
static size_t fwd_smem_bytes(bool train, bool lstream, bool fuse_dk, int Sn, bool cseg, bool drop) {
  using G = Geo<D, DH, NWV>;
  const bool flat = lstream && !drop;
  const bool flatg = flat && G::NB > 1;      // (d = 256: statistics / long vectors in global memory, no LDS copy of the weights, no long slots)
  const int lsc = lstream ? TLSAN_LS_CAP : TLSAN_LS_MAX;
  const int pstr = (flat ? 0 : lsc) + ((cseg || flatg || G::NSB < 16 || TLSAN_STAMPS) ? ((Sn + 3) & ~3) : TLSAN_SN_CAP) + 4;      // position slots per sample (k_fwd_bwd: PSTR), twice with CSEG
  const int nf = flat ? G::NSB * TLSAN_LS_CAP : 0;
  return sizeof(float) * ((train ? G::NSB * pstr * (cseg ? 2 : 1) : 0) + 2 * G::NSB * G::LSTR + ((train && ((G::FUSE_DK && fuse_dk) || (flat && !flatg))) ? G::NSB * G::LSTR : 0) + G::NW * 4 + ((G::NB == 1 && !lstream) ? G::NW * 2 * G::SPW * 16 : 0) /* LKEY: session keys (sSK) */ + G::NSB * 2 * lsc +
                          ((G::USE_SW && !flatg) ? ((G::NB > 1 && !lstream) ? 2 * (4 * G::NB * G::NB * 256 + 2 * DH) : 2 * (2 * DH * DH + 2 * DH)) : 0) + G::NW * G::WSCR +
                          ((G::KEEP_A && train && !lstream) ? G::NW * TLSAN_LS_MAX * G::NB * 256 : 0) +
                          nf * (3 + (train ? 1 : 0) + ((train && cseg) ? 1 : 0)) + ((flat && !flatg && train) ? 2 * G::NSB * G::LSTR : 0) + (flatg ? G::NSB : 0) +
                          ((flat && !train) ? G::NSB : 0) /* evaluation: the slots' samples (sSb) */ +
                          (TLSAN_STAMPS ? G::NW * 32 * 2 : 0) /* diagnostic stamps */);
}

template <int D, int DH, bool TRAIN, bool LSTREAM, int DT, bool DROP = false, int MM = TLSAN_MATRIX_F32, bool CSEG = false, int NWV = 0>
static hipError_t launch_variant_dt(const FwdArgs& a, int grid, hipStream_t st, LaunchEvents ev) {
  if constexpr (TRAIN && !CSEG) {   // tables with thousands of categories: the variant with category segments
    if (a.cseg) return launch_variant_dt<D, DH, TRAIN, LSTREAM, DT, DROP, MM, true, NWV>(a, grid, st, ev);
  }
  // (the copy of `long` for the fused dK product only in launches that fuse: at d = 64 it is what decides whether two
  //  workgroups fit a CU's LDS -- 8192 sequences, not fused: 77 us/step with it left out, 95 with it)
  const size_t smem = fwd_smem_bytes<D, DH, NWV>(TRAIN, LSTREAM, a.fuse_dk != 0, a.b.Sn, a.cseg != 0, DROP);
  auto k = k_fwd_bwd<D, DH, TRAIN, LSTREAM, DT, DROP, MM, CSEG, NWV>;
  // (per kernel variant AND device: the attribute is raised once, not on every launch; relaxed atomics -- two threads
  //  racing on a first launch both raise it, which is harmless)
  static std::atomic<size_t> smem_set[TLSAN_MAX_DEVICES];
  if (smem > 48 * 1024) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<size_t>* slot = (dev >= 0 && dev < TLSAN_MAX_DEVICES) ? &smem_set[dev] : nullptr;
    if (slot == nullptr || smem > slot->load(std::memory_order_relaxed)) {
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (slot != nullptr) slot->store(smem, std::memory_order_relaxed);
    }
  }
  // (ev: optional pair of events attached to THIS dispatch -- its own begin / end time stamps, what a kernel trace
  //  reports -- instead of two events recorded around it, which are barrier packets of their own and read 3 us long)
  if (ev.start != nullptr) hipExtLaunchKernelGGL(k, dim3(grid), dim3(Geo<D, DH, NWV>::NW * 64), smem, st, ev.start, ev.stop, 0, a);
  else hipLaunchKernelGGL(k, dim3(grid), dim3(Geo<D, DH, NWV>::NW * 64), smem, st, a);
  return hipGetLastError();
}

// bf16 table storage and bf16 matrix products: window in registers or streamed; dropout with either kind of matrix product
template <int D, int DH, bool TRAIN, bool LSTREAM>
static hipError_t launch_variant(const FwdArgs& a, int grid, hipStream_t st, LaunchEvents ev) {
  if (a.p.matrix_dtype == TLSAN_MATRIX_BF16) {  // bf16 matrix products: either window form, either table storage
    if (a.drop_thr != 0) {   // dropout (model.py:428-431): training only
      if constexpr (TRAIN) {
        if (a.p.table_dtype == TLSAN_TABLE_BF16) return launch_variant_dt<D, DH, TRAIN, LSTREAM, TLSAN_TABLE_BF16, true, TLSAN_MATRIX_BF16>(a, grid, st, ev);
        return launch_variant_dt<D, DH, TRAIN, LSTREAM, TLSAN_TABLE_F32, true, TLSAN_MATRIX_BF16>(a, grid, st, ev);
      } else {
        return hipErrorNotSupported;
      }
    }
    if (a.p.table_dtype == TLSAN_TABLE_BF16) return launch_variant_dt<D, DH, TRAIN, LSTREAM, TLSAN_TABLE_BF16, false, TLSAN_MATRIX_BF16>(a, grid, st, ev);
    return launch_variant_dt<D, DH, TRAIN, LSTREAM, TLSAN_TABLE_F32, false, TLSAN_MATRIX_BF16>(a, grid, st, ev);
  }
  if (a.p.table_dtype == TLSAN_TABLE_BF16) {
    if (a.drop_thr != 0) {   // dropout on bf16 tables: training, fp32 matrix products
      if constexpr (TRAIN) return launch_variant_dt<D, DH, TRAIN, LSTREAM, TLSAN_TABLE_BF16, true>(a, grid, st, ev);
      else return hipErrorNotSupported;
    }
    return launch_variant_dt<D, DH, TRAIN, LSTREAM, TLSAN_TABLE_BF16>(a, grid, st, ev);
  }
  if (a.drop_thr != 0) {  // dropout: training, fp32 tables
    if constexpr (TRAIN) return launch_variant_dt<D, DH, TRAIN, LSTREAM, TLSAN_TABLE_F32, true>(a, grid, st, ev);
    else return hipErrorNotSupported;
  }
  return launch_variant_dt<D, DH, TRAIN, LSTREAM, TLSAN_TABLE_F32>(a, grid, st, ev);
}

template <int D, int DH>
static hipError_t launch_fwd_bwd_impl(bool train, bool lstream, const FwdArgs& a, int grid, hipStream_t st, LaunchEvents ev) {
  if (train) return lstream ? launch_variant<D, DH, true, true>(a, grid, st, ev) : launch_variant<D, DH, true, false>(a, grid, st, ev);
  return lstream ? launch_variant<D, DH, false, true>(a, grid, st, ev) : launch_variant<D, DH, false, false>(a, grid, st, ev);
}

// one window form only (a translation unit of its own for the streamed form of d = 256, which is compiled with other flags)
template <int D, int DH, bool LSTREAM>
static hipError_t launch_fwd_bwd_form(bool train, const FwdArgs& a, int grid, hipStream_t st, LaunchEvents ev) {
  if (train) return launch_variant<D, DH, true, LSTREAM>(a, grid, st, ev);
  return launch_variant<D, DH, false, LSTREAM>(a, grid, st, ev);
}

// training step with the window in registers, no dropout, as NWV-wavefront workgroups (d = 128: 4 wavefronts, 8 samples)
template <int D, int DH, int NWV>
static hipError_t launch_train_nw(const FwdArgs& a, int grid, hipStream_t st, LaunchEvents ev) {
  if (a.drop_thr != 0) return hipErrorNotSupported;
  const bool tb = a.p.table_dtype == TLSAN_TABLE_BF16;
  if (a.p.matrix_dtype == TLSAN_MATRIX_BF16) {
    if (tb) return launch_variant_dt<D, DH, true, false, TLSAN_TABLE_BF16, false, TLSAN_MATRIX_BF16, false, NWV>(a, grid, st, ev);
    return launch_variant_dt<D, DH, true, false, TLSAN_TABLE_F32, false, TLSAN_MATRIX_BF16, false, NWV>(a, grid, st, ev);
  }
  if (tb) return launch_variant_dt<D, DH, true, false, TLSAN_TABLE_BF16, false, TLSAN_MATRIX_F32, false, NWV>(a, grid, st, ev);
  return launch_variant_dt<D, DH, true, false, TLSAN_TABLE_F32, false, TLSAN_MATRIX_F32, false, NWV>(a, grid, st, ev);
}
