// tlsan_attn.h -- the fused per-sample kernel: gathers -> long-term feature-wise attention
// -> bridge -> short-term feature-wise attention -> logit (-> loss -> full backward).
//
// Follows reference TLSAN/model.py:84-137 (forward), :164-172 (loss), and what
// tf.gradients (:198) differentiates.  Phases per workgroup pass over NSB samples:
//   P1  per wave   gather long rows, scale by gamma*usert*hist_t, FWA block 1 -> long
//   P2  workgroup  bridge = long . K + k0          (16x16x4 f32 MFMA, A from LDS, B = K^T rows)
//   P3  per wave   gather session rows, FWA block 2, logit, BCE, backward of block 2
//   P4  workgroup  dlong = dbridge . K^T           (MFMA, B = K rows)
//   P5  per wave   backward of block 1, per-use gradient rows, usert / gamma gradients
// Per-use gradient rows are written once with plain 16-B stores at destination-sorted
// positions (drawn from the fill cursors of the inverted index with one returning integer
// atomic per use); k_apply_* sum each row's contiguous segment with exact arithmetic.
//
// The kernel is bound by the latency of ONE wavefront's dependent chain (4096 samples are
// only 2 wavefronts per SIMD), so the code is organised to keep that chain short:
// index/row loads are unconditional (clamped) and batched -- no per-lane branches around
// loads; attention weights live in LDS; the dW products of position p are issued after the
// map chain of position p+1 (double-buffered transpose scratch); cross-lane reductions are
// batched after the loops.
#pragma once
#include "tlsan_common.h"
#ifndef TLSAN_STAMPS
#define TLSAN_STAMPS 0   // 1: the diagnostic build with in-kernel cycle stamps (scripts/stamps.py)
#endif

// ---------------------------------------------------------------------------------------
// feature_wise_attention forward (model.py:370-394) over NPOS static positions held in
// registers.  e[p] = raw rows, sc[p] = per-position scale.  Returns out = sum_p a[p] x[p]
// plus the per-channel softmax statistics (mx = max score, Z = 1/sum exp) from which the
// backward recomputes a[p] = exp(m2[p] - mx) * Z (exactly 0 on masked positions).
// Positions are processed two per (wave-uniform) branch so their MFMA chains interleave.
// DROP: dropout on the inputs of both maps (model.py:428-431), pattern drop_scale4(dc, net, p, ., chb).
template <int NB, int NPOS, bool DROP = false, int MM = TLSAN_MATRIX_F32>
__device__ __forceinline__ void fwa_forward(const typename MMT<MM>::opd (&FT1)[NB][NB], const f32x4 (&b1)[NB],
                                            const typename MMT<MM>::opd (&FT2)[NB][NB], const f32x4 (&b2)[NB],
                                            const f32x4 (&e)[NPOS][NB], const float (&sc)[NPOS],
                                            int n_valid, int pmax, f32x4 (&mx)[NB],
                                            f32x4 (&Z)[NB], f32x4 (&out)[NB], float* __restrict__ sAw,
                                            const DropCtx& dc = DropCtx{}, int net = 0, const int* chb = nullptr,
                                            f32x4 (*aout)[NB] = nullptr) {
  f32x4 a[NPOS][NB];
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) mx[kb] = (f32x4)(TLSAN_NEG);
#pragma unroll
  for (int p0 = 0; p0 < NPOS; p0 += 2) {
    if (p0 < pmax) {  // wave-uniform
#pragma unroll
      for (int p = p0; p < p0 + 2 && p < NPOS; ++p) {
        f32x4 xv[NB], z[NB], m2[NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) xv[kb] = e[p][kb] * sc[p];
        if constexpr (DROP) {
#pragma unroll
          for (int kb = 0; kb < NB; ++kb) xv[kb] *= drop_scale4(dc, net, p, 0, chb[kb]);
        }
        map_apply<NB, MM>(FT1, b1, xv, z);  // model.py:380 (relu below)
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
          for (int i = 0; i < 4; ++i) z[kb][i] = fmaxf(z[kb][i], 0.0f);
        if constexpr (DROP) {
#pragma unroll
          for (int kb = 0; kb < NB; ++kb) z[kb] *= drop_scale4(dc, net, p, 1, chb[kb]);
        }
        map_apply<NB, MM>(FT2, b2, z, m2);  // model.py:382
        const bool valid = p < n_valid;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
          // model.py:384: m2 + (1-mask)*(-1e30) == -1e30 exactly in fp32
          a[p][kb] = valid ? m2[kb] : (f32x4)(TLSAN_NEG);
#pragma unroll
          for (int i = 0; i < 4; ++i) mx[kb][i] = fmaxf(mx[kb][i], a[p][kb][i]);
        }
      }
    } else {
#pragma unroll
      for (int p = p0; p < p0 + 2 && p < NPOS; ++p)
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) a[p][kb] = (f32x4)(TLSAN_NEG);
    }
  }
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) Z[kb] = (f32x4)(0.0f);
#pragma unroll
  for (int p = 0; p < NPOS; ++p)
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float ev = exp2s(a[p][kb][i] - mx[kb][i]);  // softmax over positions, model.py:386
        a[p][kb][i] = ev;
        Z[kb][i] += ev;
      }
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    out[kb] = (f32x4)(0.0f);
#pragma unroll
    for (int i = 0; i < 4; ++i) Z[kb][i] = fast_rcp(Z[kb][i]);
  }
#pragma unroll
  for (int p = 0; p < NPOS; ++p)
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      a[p][kb] = a[p][kb] * Z[kb];
      out[kb] += a[p][kb] * (e[p][kb] * sc[p]);  // model.py:387
      if (sAw != nullptr) *(f32x4*)(sAw + (p * NB + kb) * 256) = a[p][kb];  // [position][lane] float4s
      if (aout != nullptr) aout[p][kb] = a[p][kb];                            // (evaluation: FwdArgs.att0)
    }
}

// ---------------------------------------------------------------------------------------
// Backward of one position of feature_wise_attention, first half.  Inputs in C-layout:
// x (scaled row), z1 = x W1 + b1 (recomputed by the caller, bitwise equal to the forward),
// a (softmax weight), outv (block output), dout (gradient of block output).
// Produces dx, accumulates db1 / db2, and writes the four 16x16 tiles (x, dz1, m1, dm2) of
// this position transposed-ready into the LDS scratch `T` for bwd_dw.
// DROP: k1 / k2 are the dropout scales of this position's map inputs (x (.) k1 entered map 1,
// relu(z1) (.) k2 entered map 2): the staged operands of the dW products and the two
// back-propagated vectors carry them.
template <int NB, int TSTR, bool DROP = false, int MM = TLSAN_MATRIX_F32>
__device__ __forceinline__ void bwd_compute(const typename MMT<MM>::opd (&FN2)[NB][NB],
                                            const typename MMT<MM>::opd (&FN1)[NB][NB], const f32x4 (&xv)[NB],
                                            const f32x4 (&z1)[NB], const f32x4 (&av)[NB],
                                            const f32x4 (&outv)[NB], const f32x4 (&dout)[NB],
                                            float* __restrict__ T, int q, int r,
                                            f32x4 (&db1)[NB], f32x4 (&db2)[NB], f32x4 (&dx)[NB],
                                            const f32x4* k1 = nullptr, const f32x4* k2 = nullptr) {
  f32x4 m1[NB], dm2[NB], dm1[NB], dz1[NB], dxm[NB], zero[NB];
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    zero[kb] = (f32x4)(0.0f);
    dm2[kb] = av[kb] * dout[kb] * (xv[kb] - outv[kb]);  // softmax-over-positions backward
#pragma unroll
    for (int i = 0; i < 4; ++i) m1[kb][i] = fmaxf(z1[kb][i], 0.0f);
  }
  map_apply<NB, MM>(FN2, zero, dm2, dm1);  // dm1 = dm2 . W2^T
  if constexpr (DROP) {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      dm1[kb] *= k2[kb];
      m1[kb] *= k2[kb];   // what map 2 saw: the dW2 operand
    }
  }
#pragma unroll
  for (int kb = 0; kb < NB; ++kb)
#pragma unroll
    for (int i = 0; i < 4; ++i) dz1[kb][i] = z1[kb][i] > 0.0f ? dm1[kb][i] : 0.0f;
  map_apply<NB, MM>(FN1, zero, dz1, dxm);  // dz1 . W1^T
  if constexpr (DROP) {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) dxm[kb] *= k1[kb];
  }
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    dx[kb] = av[kb] * dout[kb] + dxm[kb];
    db1[kb] += dz1[kb];
    db2[kb] += dm2[kb];
  }
  // tiles in the scratch: [0,NB) x, [NB,2NB) dz1, [2NB,3NB) m1, [3NB,4NB) dm2; each [16][TSTR]
  const int wofs = r * TSTR + 4 * q;
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    *(f32x4*)(T + (0 * NB + kb) * 16 * TSTR + wofs) = DROP ? xv[kb] * k1[kb] : xv[kb];   // what map 1 saw: the dW1 operand
    *(f32x4*)(T + (1 * NB + kb) * 16 * TSTR + wofs) = dz1[kb];
    *(f32x4*)(T + (2 * NB + kb) * 16 * TSTR + wofs) = m1[kb];
    *(f32x4*)(T + (3 * NB + kb) * 16 * TSTR + wofs) = dm2[kb];
  }
  __builtin_amdgcn_wave_barrier();
}

// Second half: dW1 += x^T dz1, dW2 += m1^T dm2 over the 16 (sample, head) columns of the tile
// staged in `T` (MFMA with the column index as the K dimension).  DS operations of a
// wavefront execute in order, so reading what other lanes of the same wave wrote needs no
// barrier, only program order.
template <int NB, int TSTR, int MM = TLSAN_MATRIX_F32>
__device__ __forceinline__ void bwd_dw(const float* __restrict__ T, int q, int r,
                                       f32x4 (&dW1)[NB][NB], f32x4 (&dW2)[NB][NB]) {
  f32x4 ax[NB], bz[NB], am[NB], bd[NB];
#pragma unroll
  for (int kb = 0; kb < NB; ++kb)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int rofs = (4 * q + s) * TSTR + r;
      ax[kb][s] = T[(0 * NB + kb) * 16 * TSTR + rofs];
      bz[kb][s] = T[(1 * NB + kb) * 16 * TSTR + rofs];
      am[kb][s] = T[(2 * NB + kb) * 16 * TSTR + rofs];
      bd[kb][s] = T[(3 * NB + kb) * 16 * TSTR + rofs];
    }
  if constexpr (MM == TLSAN_MATRIX_F32) {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int jb = 0; jb < NB; ++jb)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          dW1[kb][jb] = TLSAN_MFMA(ax[kb][s], bz[jb][s], dW1[kb][jb]);
          dW2[kb][jb] = TLSAN_MFMA(am[kb][s], bd[jb][s], dW2[kb][jb]);
        }
  } else {
    typename MMT<MM>::opd pax[NB], pbz[NB], pam[NB], pbd[NB];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      pax[kb] = mm_pack<MM>(ax[kb]);
      pbz[kb] = mm_pack<MM>(bz[kb]);
      pam[kb] = mm_pack<MM>(am[kb]);
      pbd[kb] = mm_pack<MM>(bd[kb]);
    }
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
        dW1[kb][jb] = mm_mma<MM>(pax[kb], pbz[jb], dW1[kb][jb]);
        dW2[kb][jb] = mm_mma<MM>(pam[kb], pbd[jb], dW2[kb][jb]);
      }
  }
  __builtin_amdgcn_wave_barrier();
}

// 16-B gather of channels [c, c+4) of the concatenated row [item_emb[it] || cate_emb[cat[it]]]
// (model.py:84-86,105-107,111-113)
// (the ADDRESS is selected, then one load is issued: a load in each arm of a ?: makes hipcc branch around both)
template <int DT = TLSAN_TABLE_F32>
__device__ __forceinline__ f32x4 gather_item4c(const FwdArgs& a, int it, int ct, int c) {
  const bool item = c < a.di;
  const float* base = item ? a.p.item_emb : a.p.cate_emb;
  const size_t idx = item ? (size_t)it * a.p.ld_item + c : (size_t)ct * a.dc + (c - a.di);
  return tbl_ld4<DT>(base, idx);
}

template <int DT = TLSAN_TABLE_F32>
__device__ __forceinline__ typename TblRaw<DT>::type gather_item4c_raw(const FwdArgs& a, int it, int ct, int c) {
  const bool item = c < a.di;
  const float* base = item ? a.p.item_emb : a.p.cate_emb;
  const size_t idx = item ? (size_t)it * a.p.ld_item + c : (size_t)ct * a.dc + (c - a.di);
  return tbl_ld4_raw<DT>(base, idx);
}

template <int DT = TLSAN_TABLE_F32>
__device__ __forceinline__ f32x4 gather_item4(const FwdArgs& a, int it, int c) {
  return gather_item4c<DT>(a, it, a.p.item_cate[it], c);
}

// Per-pass, per-wave gradient accumulators of one attention block, and their deterministic
// reduction over the workgroup's wavefronts into the pass's partial record.
// LDS staging: stage[wave*WSCR + vec*256 + lane*4 + i]; vec order: dW1[kb][jb], dW2[kb][jb], db1[kb],
// db2[kb], extra[kb] (extra = dk0 for block 2).
template <int NB>
struct AccSet {
  f32x4 dW1[NB][NB], dW2[NB][NB], db1[NB], db2[NB];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int x = 0; x < NB; ++x) {
      db1[x] = db2[x] = (f32x4)(0.0f);
#pragma unroll
      for (int y = 0; y < NB; ++y) dW1[x][y] = dW2[x][y] = (f32x4)(0.0f);
    }
  }
};

// `stage` is the wave's OWN scratch region (stride WSCR), so staging never touches the
// transpose tiles another wavefront may still be using.
template <int NB, int CPS, bool EXTRA>
__device__ __forceinline__ void stage_accs(AccSet<NB>& A, f32x4 (&extra)[NB], float* __restrict__ stage,
                                           int lane) {
  float* base = stage + lane * 4;
  int v = 0;
#pragma unroll
  for (int kb = 0; kb < NB; ++kb)
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) *(f32x4*)(base + (v++) * 256) = A.dW1[kb][jb];
#pragma unroll
  for (int kb = 0; kb < NB; ++kb)
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) *(f32x4*)(base + (v++) * 256) = A.dW2[kb][jb];
  // biases: per-lane partial over the 16 columns r -> xor-reduce over r (owner r == 0)
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    f32x4 t = A.db1[kb];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = lanes_sum<16>(t[i]);
    *(f32x4*)(base + (v++) * 256) = t;
  }
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    f32x4 t = A.db2[kb];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = lanes_sum<16>(t[i]);
    *(f32x4*)(base + (v++) * 256) = t;
  }
  if constexpr (EXTRA) {
    // dk0: sum over the samples of the wave (column index bits >= CPS), owner s_loc == 0
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      f32x4 t = extra[kb];
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = stride_sum<CPS>(t[i]);
      *(f32x4*)(base + (v++) * 256) = t;
    }
  }
}

// Half of the above (Geo::SPLIT): PART 0 = {dW1, db1, extra}, PART 1 = {dW2, db2}.
template <int NB, int CPS, bool EXTRA, int PART>
__device__ __forceinline__ void stage_part(AccSet<NB>& A, f32x4 (&extra)[NB], float* __restrict__ stage, int lane) {
  float* base = stage + lane * 4;
  int v = 0;
#pragma unroll
  for (int kb = 0; kb < NB; ++kb)
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) *(f32x4*)(base + (v++) * 256) = PART == 0 ? A.dW1[kb][jb] : A.dW2[kb][jb];
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    f32x4 t = PART == 0 ? A.db1[kb] : A.db2[kb];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = lanes_sum<16>(t[i]);
    *(f32x4*)(base + (v++) * 256) = t;
  }
  if constexpr (EXTRA && PART == 0) {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      f32x4 t = extra[kb];
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = stride_sum<CPS>(t[i]);
      *(f32x4*)(base + (v++) * 256) = t;
    }
  }
}

template <typename G, bool EXTRA, int PART>
__device__ __forceinline__ void reduce_part(const float* __restrict__ stage, float* __restrict__ out, int pW, int pB,
                                            int tid) {
  constexpr int NB = G::NB, CW = G::CW, CPS = G::CPS, NW = G::NW;
  constexpr int NV = NB * NB + NB + ((EXTRA && PART == 0) ? NB : 0);
  for (int o = tid; o < NV * 256; o += NW * 64) {
    const int v = o >> 8, l4 = o & 255, ln = l4 >> 2, i = l4 & 3;
    const int q = ln >> 4, r = ln & 15;
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += stage[(size_t)w * G::WSCR + o];
    if (v < NB * NB) {
      const int kb = v / NB, jb = v % NB;
      out[pW + (16 * kb + 4 * q + i) * CW + 16 * jb + r] = s;
    } else if (v < NB * NB + NB) {
      const int kb = v - NB * NB;
      if (r == 0) out[pB + 16 * kb + 4 * q + i] = s;
    } else {
      const int kb = v - (NB * NB + NB);
      const int s_loc = r / CPS, col = r % CPS;
      if (s_loc == 0) out[G::P_K0 + col * CW + 16 * kb + 4 * q + i] = s;
    }
  }
}

// all threads: sum the staged vectors over the wavefronts (fixed order) and scatter them to
// the partial record.  pW1/pB1/pW2/pB2 = section offsets of this attention block.
template <typename G, bool EXTRA>
__device__ __forceinline__ void reduce_staged(const float* __restrict__ stage, float* __restrict__ out,
                                              int pW1, int pB1, int pW2, int pB2, int tid) {
  constexpr int NB = G::NB, CW = G::CW, CPS = G::CPS, NW = G::NW;
  constexpr int NV = 2 * NB * NB + 2 * NB + (EXTRA ? NB : 0);
  for (int o = tid; o < NV * 256; o += NW * 64) {
    const int v = o >> 8, l4 = o & 255, ln = l4 >> 2, i = l4 & 3;
    const int q = ln >> 4, r = ln & 15;
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += stage[(size_t)w * G::WSCR + o];
    if (v < 2 * NB * NB) {
      const int m = v % (NB * NB), kb = m / NB, jb = m % NB;
      const int idx = (16 * kb + 4 * q + i) * CW + 16 * jb + r;
      out[(v < NB * NB ? pW1 : pW2) + idx] = s;
    } else if (v < 2 * NB * NB + 2 * NB) {
      const int m = v - 2 * NB * NB, kb = m % NB;
      if (r == 0) out[(m < NB ? pB1 : pB2) + 16 * kb + 4 * q + i] = s;
    } else {
      const int kb = v - (2 * NB * NB + 2 * NB);
      const int s_loc = r / CPS, col = r % CPS;
      if (s_loc == 0) out[G::P_K0 + col * CW + 16 * kb + 4 * q + i] = s;
    }
  }
}

// one online-softmax step of a streamed attention block (running max mx, sum Z, weighted sum N)
template <int NB>
__device__ __forceinline__ void online_step(f32x4 (&mx)[NB], f32x4 (&Z)[NB], f32x4 (&N)[NB],
                                            const f32x4 (&m2)[NB], const f32x4 (&xv)[NB]) {
#pragma unroll
  for (int kb = 0; kb < NB; ++kb)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float mn = fmaxf(mx[kb][i], m2[kb][i]);
      const float so = exp2s(mx[kb][i] - mn), ev = exp2s(m2[kb][i] - mn);
      Z[kb][i] = Z[kb][i] * so + ev;
      N[kb][i] = N[kb][i] * so + ev * xv[kb][i];
      mx[kb][i] = mn;
    }
}

// LSTREAM: the long block is streamed like the short one (any Ls <= TLSAN_LS_CAP); otherwise its
// Ls <= TLSAN_LS_MAX positions stay in registers between forward and backward.
// CSEG (FwdArgs.cseg, tables with thousands of categories): the category half of every item use's gradient row goes to
// the category's own segment of Gc.  A compile-time variant: as a run-time flag its tests sat inside the pipelined
// loops of the variant that does not need it (C3 shape: 61.0 -> 62.9 us/step).
// NWV (0 = the width's default): wavefronts per workgroup, see Geo.  The register budget is that of two wavefronts per
// SIMD either way (__launch_bounds__(512)): an 8-wavefront workgroup fills a CU, two 4-wavefront ones share it.
template <int D, int DH, bool TRAIN, bool LSTREAM, int DT = TLSAN_TABLE_F32, bool DROP = false, int MM = TLSAN_MATRIX_F32, bool CSEG = false, int NWV = 0>
__global__ __launch_bounds__(512) void k_fwd_bwd(FwdArgs a) {
  static_assert(!DROP || TRAIN, "dropout: train steps only");
  static_assert(!CSEG || TRAIN, "category segments: train steps only");
  using G = Geo<D, DH, NWV>;
  static_assert(G::NSB == 16 || (G::NSB == 8 && !LSTREAM), "8-sample workgroups: windows in registers only");
  using opd = typename MMT<MM>::opd;   // an operand of one 16-deep contraction (tlsan_common.h)
  constexpr int NB = G::NB, CPS = G::CPS, SPW = G::SPW, NW = G::NW, NSB = G::NSB;
  constexpr int LS = LSTREAM ? 1 : TLSAN_LS_MAX;             // positions held in registers
  constexpr int LSC = LSTREAM ? TLSAN_LS_CAP : TLSAN_LS_MAX;  // position slots in the LDS tables
  constexpr int LSTR = G::LSTR, TSTR = G::TSTR, CW = G::CW;
  constexpr int WB = 2 * DH * DH + 2 * DH;    // floats of one attention block's weights
  // per-sample position slots: long, session (the batch's padded session length, rounded up to 4), 3 singles
  // (CSEG keeps two such arrays and sizes them by the batch; otherwise the slot count is a compile-time constant)

  constexpr bool LKEY = NB == 1 && !LSTREAM;   // row keys reach a sample's lanes through the LDS (see g_item below, P1, fetch_row_of)
  constexpr int NLK = 16;                                    // session entries per chunk of keys
  const bool FUSE_RT = a.fuse_dk != 0;   // (this launch forms the dK partials itself; as a compile-time constant: -0.2 us/step, not worth a variant)
  // FLAT (streamed windows): the window positions of the workgroup's 16 samples form ONE list that is dealt out evenly
  // to its 16 column groups (a wavefront's lanes that share a sample slot) -- see P1.  Its entries, the per-sample
  // softmax statistics and the long-term vectors live in the LDS, where any group can reach them.
  constexpr bool FLAT = LSTREAM && !DROP;   // (dropout keeps a window per column group: its pattern is indexed by (sample, position))
  constexpr int NF = FLAT ? NSB * TLSAN_LS_CAP : 0;      // entries of the flat list (every window at the cap)
  // NB > 1 (d = 256): the LDS has no room for the per-sample statistics and long-term vectors beside the list -- they
  // go through global memory (FwdArgs.gStat, gLong: 32 KB per workgroup, L2-resident), the attention weights are
  // read from `dense` instead of an LDS copy, and the position tables keep session slots only
  constexpr bool FLATG = FLAT && NB > 1;
  constexpr bool USE_SW = G::USE_SW && !FLATG;          // attention weights staged in LDS (when they fit)
  constexpr int LSCP = FLAT ? 0 : LSC;                  // long slots of the position tables (FLAT: the list holds them)
  // (8-sample workgroups size them by the batch as well: two workgroups must fit a CU's LDS)
  // (the diagnostic stamps build as well: its 2 KB of stamps must fit beside the 160 KB the d = 128 training kernel fills)
  const int SNS = (CSEG || FLATG || NSB < 16 || TLSAN_STAMPS) ? ((a.b.Sn + 3) & ~3) : TLSAN_SN_CAP;
  const int PSTR = LSCP + SNS + 4;
  const int P_TGT = LSCP + SNS, P_USR = P_TGT + 1, P_UC = P_TGT + 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                   // [NSB][LSTR]  long -> dbridge
  float* sB0 = sA + NSB * LSTR;       // [NSB][LSTR]  bridge -> dlong  (FLAT: placed in front of sT instead, see sPart)
  float* sL = FLAT ? sB0 : sB0 + NSB * LSTR;   // [NSB][LSTR]  long, kept for the fused dK product (TRAIN && FUSE_DK) and for FLAT's backward
  float* sS = sL + ((TRAIN && ((G::FUSE_DK && FUSE_RT) || (FLAT && !FLATG))) ? NSB * LSTR : 0);  // [NW][4] scalar staging  (sL exists in launches that fuse: tlsan_attn_inst.h)
  int* sSK = (int*)(sS + NW * 4);     // LKEY: [NW][2][SPW][NLK] item ids | categories of the current chunk of session entries
  float* sH = (float*)(sSK + (LKEY ? NW * 2 * SPW * NLK : 0));   // [NSB][2*LS] hist_t and usert*hist_t of the pass  (FLAT: [2][NF], by flat index)
  float* sW = sH + NSB * 2 * LSC;     // [2][WB] attention weights (W1,b1,W2,b2) of both blocks  (PERM: [2][WBP], see below)
  // PERM (two 16-channel blocks per column, d = 256): the weight fragments are re-read from the LDS at every position
  // (Geo::AT_USE), and read from the row-major copy that costs 8 two-dword reads with computed addresses per fragment,
  // four-way bank-conflicted (a fragment's lanes (q, r) read W[(16 kb + 4 q + s) * 32 + 16 jb + r]: the four q hit the
  // same banks) -- 26 M conflict cycles per launch, 28 % of the kernel's time (profiles/r04_pmc_d256_summary.txt).  The LDS
  // copy is therefore kept in FRAGMENT order, every matrix twice: table [pair][lane][4] with pair = the fragment's
  // (out block, in block) -- a fragment is NB * NB conflict-free 16-byte reads at constant offsets from the lane's base.
  // Per attention block: [W1 as T fragments | W2 as T | W1 as N | W2 as N | b1 | b2].
  constexpr bool PERM = USE_SW && NB > 1 && !LSTREAM;   // (streamed windows with dropout keep the row-major copy: the 16 KB more do not fit beside their position tables)
  constexpr int PP = NB * NB * 256;            // floats of one fragment table
  constexpr int WBP = 4 * PP + 2 * DH;         // floats of one attention block's weights in fragment order
  int* sP = (int*)(sW + (USE_SW ? (PERM ? 2 * WBP : 2 * WB) : 0));  // [NSB][PSTR] destination-sorted row of every use
  // CSEG (FwdArgs.cseg, many categories): the category half of an item use's gradient row goes to the category's own
  // segment of Gc -- its position, drawn from the category's cursor, sits in sPc beside the item position in sP
  int* sPc = sP + (TRAIN ? NSB * PSTR : 0);      // [NSB][PSTR], only when a.cseg
  int* sFid = sPc + ((TRAIN && CSEG) ? NSB * PSTR : 0);   // FLAT: [NF] item id, category, (slot << 8 | position) of every list entry,
  int* sFct = sFid + NF;                                  //       its destination rows (TRAIN), and below the per-sample statistics
  int* sFst = sFct + NF;
  int* sFpos = sFst + NF;
  int* sFcpos = sFpos + (TRAIN ? NF : 0);
  float* sMx = (float*)(sFcpos + ((TRAIN && CSEG) ? NF : 0));   // [NSB][LSTR] per-channel max of the window's scores
  float* sIz = sMx + ((FLAT && !FLATG && TRAIN) ? NSB * LSTR : 0);   // [NSB][LSTR] 1 / sum of exponentials
  int* sBx = (int*)(sIz + ((FLAT && !FLATG && TRAIN) ? NSB * LSTR : 0));   // FLATG: [NSB] the slots' samples (rows of gStat / gLong)
  int* sSb = sBx + (FLATG ? NSB : 0);                   // evaluation, FLAT: [NSB] the slots' samples (rows of FwdArgs.att0)
  float* sB = FLAT ? (float*)(sSb + ((!TRAIN && FLAT) ? NSB : 0)) : sB0;
  float* sT = FLAT ? sB + NSB * LSTR : (float*)sFid;  // per-wave transpose scratch / staging
  // FLAT: the partial softmax states of P1, 32 slots of [3][D], lie over sB and sT (neither is touched before P2)
  float* sPart = sB;
  static_assert(!FLAT || 32 * 3 * D <= NSB * LSTR + NW * G::WSCR, "partial states must fit sB + sT");
  float* sFht = sH;
  float* sFuh = sH + NF;
  constexpr bool KEEP_A = G::KEEP_A && TRAIN && !LSTREAM;
  constexpr bool PIPE5 = KEEP_A && !DROP;       // long backward as a skewed software pipeline (see P5)
  constexpr bool LPF = NB == 1;                 // streamed windows: the next position's row is prefetched
  // ... and the next chunk's ids / weights / categories are loaded a chunk ahead (d = 256 in fp32 has no registers for
  // either: 568 -> 618 us at Ls = 90 with this one; with bf16 matrix operands it has: 328 -> 321 us)
  constexpr bool LCH = NB == 1 || MM == TLSAN_MATRIX_BF16;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int q = lane >> 4, r = lane & 15;
  const int s_loc = r / CPS, col = r % CPS;
  const int srow = wave * SPW + s_loc;  // sample row inside the workgroup pass
  float* T = sT + wave * G::WSCR;
  float* sAw = sT + NW * G::WSCR + wave * (LS * NB * 256) + lane * 4;  // this lane's slot of the kept softmax weights
  int chb[NB];
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) chb[kb] = col * CW + 16 * kb + 4 * q;
  // chl: where the lane's float4 of 16-channel block kb sits inside a [slot][channel] row of the partial softmax states
  // (sPart, streamed windows).  Indexed by the channel itself the eight column groups of a sample are CW dwords apart --
  // the same banks: a 16-byte store is served in groups of 8 consecutive lanes (one sample's eight columns at one q), all
  // eight on ONE set of four banks, 8 LDS cycles per group instead of 1, six such stores per list entry and wavefront
  // (58 % of the C5 kernel's LDS cycles were bank conflicts: profiles/r04_pmc_c5_summary.txt).  Each column's block is
  // therefore rotated by 4 * col dwords (CW = 32; by 4 * (col / 2) at CW = 16, where two columns span the 32 banks): the
  // eight lanes of a group then cover all 32 banks.  Writer and reader address by channel, so both see the same rotation.
  int chl[NB];
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) chl[kb] = col * CW + ((16 * kb + 4 * q + 4 * (NB > 1 ? col : (col >> 1))) & (CW - 1));
  const bool lead = (q == 0) && (col == 0);  // one lane per sample
  // LKEY (windows in registers, one 16-channel block per column): the lane's half of a concatenated [item_emb | cate_emb]
  // row as per-lane constants -- table, row stride, offset of its four channels -- so that a gather's address is
  // key * g_ld + g_off with key = the item id or the item's category (see P1)
  // (element index of the lane's piece of row `key`: one v_mad_u64_u32 -- ids, strides and offsets are non-negative 32-bit values)
  auto g_idx = [&](int key, int ld, int off) -> size_t { return (size_t)((unsigned long long)(unsigned)key * (unsigned)ld + (unsigned)off); };
  const bool g_item = chb[0] < a.di;
  const float* g_base = g_item ? a.p.item_emb : a.p.cate_emb;
  const int g_ld = g_item ? a.p.ld_item : a.dc;
  const int g_off = g_item ? chb[0] : chb[0] - a.di;
  // where channels [c, c+4) of an item use's gradient row go: the use's row of Gi, or (CSEG, category half) the use's
  // row in its category's segment of Gc
  auto use_dst = [&](int pos, int cpos, int c) -> float* {
    return (CSEG && c >= a.di) ? a.Gc + (size_t)cpos * a.dc + (c - a.di) : a.Gi + (size_t)pos * D + c;
  };

  // channels [ch, ch + 4) of position p of sample b in an attention-weight tensor of T positions (FwdArgs.att0 / att1)
  auto att_at = [&](float* base, int T_, int b, int p, int ch) -> float* {
    return base + (((size_t)(ch / DH) * a.b.B + b) * T_ + p) * DH + (ch % DH);
  };
  if constexpr (TRAIN) {
    // (tlsan_step_out.started: this kernel running means everything queued before the step is complete)
    if (a.started != nullptr && blockIdx.x == 0 && tid == 0)
      __hip_atomic_store(a.started, a.started_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const float* dn = a.p.dense;
  const float gamma = dn[a.lay.gamma];
  const float P = a.p.scale ? *a.p.scale : 1.0f;  // tables hold W / P (lazy L2 decay)
  const int Ls = a.Ls, Sn = a.b.Sn, B = a.b.B;
  // attention weights -> LDS once per workgroup (both blocks are contiguous runs of `dense`)
  if constexpr (PERM) {
    for (int o = tid; o < 2 * WBP; o += NW * 64) {
      const int m = o / WBP, x = o - m * WBP;
      const float* src = dn + (m ? a.lay.f2_W1 : a.lay.f1_W1);   // [W1 | b1 | W2 | b2] of the block, row-major
      float v;
      if (x < 4 * PP) {
        const int tab = x / PP, y = x - tab * PP, pair = y >> 8, ln = (y >> 2) & 63, s_ = y & 3;
        const int qq = ln >> 4, rr = ln & 15, pa = pair / NB, pb = pair % NB;
        const float* W = src + ((tab & 1) ? DH * DH + DH : 0);    // W1 or W2
        // T fragment (tab 0, 1): F[jb = pa][kb = pb][s] = W[16 kb + 4 q + s][16 jb + r]; N (tab 2, 3): F[kb = pa][jb = pb][s] = W[16 kb + r][16 jb + 4 q + s]
        v = tab < 2 ? weff<DH>(W, 16 * pb + 4 * qq + s_, 16 * pa + rr) : weff<DH>(W, 16 * pa + rr, 16 * pb + 4 * qq + s_);
        if (tab == 1) v *= TLSAN_LOG2E;   // W2 as forward fragments: scores in units of log 2 (TLSAN_LOG2E)
      } else {
        const int y = x - 4 * PP;
        v = y < DH ? src[DH * DH + y] : src[2 * DH * DH + DH + (y - DH)] * TLSAN_LOG2E;
      }
      sW[o] = v;
    }
    __syncthreads();
  } else if constexpr (USE_SW) {
    for (int o = tid; o < 2 * WB; o += NW * 64)
      sW[o] = (o < WB) ? dn[a.lay.f1_W1 + o] : dn[a.lay.f2_W1 + (o - WB)];
    __syncthreads();
  }
  const float* wb1 = USE_SW ? sW : dn + a.lay.f1_W1;
  const float* wb2 = USE_SW ? sW + (PERM ? WBP : WB) : dn + a.lay.f2_W1;
  // (PERM: wXW1 / wXW2 are the T-fragment tables; the N tables lie 2 * PP floats behind them)
  const float *w1W1 = wb1, *w1b1 = wb1 + (PERM ? 4 * PP : DH * DH), *w1W2 = PERM ? wb1 + PP : w1b1 + DH, *w1b2 = PERM ? w1b1 + DH : w1W2 + DH * DH;
  const float *w2W1 = wb2, *w2b1 = wb2 + (PERM ? 4 * PP : DH * DH), *w2W2 = PERM ? wb2 + PP : w2b1 + DH, *w2b2 = PERM ? w2b1 + DH : w2W2 + DH * DH;
// (FLATG reads the weights from global memory: their per-lane addresses, formed early, were kept in spilled registers, and
//  every reload from scratch memory drains the vector-memory queue -- 62 such drains in the C5 kernel; behind an opaque zero
//  the address arithmetic stays at the use: d = 256, Ls = 90 344 -> 312 us/step, the C5 shape 412 -> 366, in bf16 317 -> 287)
#define LD_Q (q + (FLATG ? opaque_zero(q) : 0))
// (the same for the rows fetched inside loops at d = 256: the lane's channel offset -- with it the choice of table and the
//  64-bit base of the lane's half of the row, invariants of those loops -- is formed behind the id of the row)
#define CH_USE(c_, id_) ((c_) + ((NB > 1 && LSTREAM) ? opaque_zero(id_) : 0))   // (streamed: Ls = 90 296 -> 284 us/step, C5 375 -> 358; with the window in registers it added drains)
#define LD_T(W_, F_) do { if constexpr (PERM) load_frag_P<NB, MM>((W_), lane, F_); else load_frag_T<DH, NB, MM>((W_), LD_Q, r, F_); } while (0)
// the forward fragment of W2 and the bias b2: in units of log 2 (tlsan_common.h, TLSAN_LOG2E; PERM: the LDS tables hold them scaled)
#define LD_T2(W_, F_) do { if constexpr (PERM) load_frag_P<NB, MM>((W_), lane, F_); else load_frag_T<DH, NB, MM>((W_), LD_Q, r, F_, TLSAN_LOG2E); } while (0)
#define LD_B2(b_, out_) load_bias<DH, NB>((b_), LD_Q, out_, PERM ? 1.0f : TLSAN_LOG2E)
#define LD_N(W_, F_) do { if constexpr (PERM) load_frag_P<NB, MM>((W_) + 2 * PP, lane, F_); else load_frag_N<DH, NB, MM>((W_), LD_Q, r, F_); } while (0)

  // diagnostic cycle stamps: only in a -DTLSAN_STAMPS=1 build (scripts/stamps.py loads it through TLSAN_LIB_PATH;
  // the production kernel carries no stamp code).  Kept in the LDS while the pass runs -- a global store per stamp
  // would sit in front of every later vmcnt(0) wait and distort what it measures -- and copied out at the end.
#ifndef TLSAN_STAMPS
#define TLSAN_STAMPS 0
#endif
#if TLSAN_STAMPS
  unsigned long long* sStamp = (unsigned long long*)(sT + NW * G::WSCR + ((G::KEEP_A && TRAIN && !LSTREAM) ? NW * LS * NB * 256 : 0)) + wave * 32;
#define TLSAN_STAMP(k)                                                                       \
  do {                                                                                       \
    if (a.stamps != nullptr && lane == 0) sStamp[k] = __builtin_amdgcn_s_memtime();          \
  } while (0)
  if (a.stamps != nullptr && lane < 32) sStamp[lane] = 0;   // (a stamp the wavefront never reaches reads 0, not what the LDS held)
#else
#define TLSAN_STAMP(k) do { } while (0)
#endif

  for (int g = blockIdx.x; g < a.ngroups; g += gridDim.x) {
    // later passes of a workgroup: the previous pass's closing reduction reads EVERY wavefront's scratch (sT) -- a wavefront
    // that ran ahead into this pass would write its own (sPerm, the window's keys; FLAT: the partial states of P1)
    if (g != blockIdx.x) __syncthreads();
    TLSAN_STAMP(0);
#if TLSAN_STAMPS
    if (a.stamps != nullptr && lane == 0) sStamp[28] = __builtin_amdgcn_s_memrealtime();   // (100 MHz, one clock for the whole device)
#endif
    // Which sample a slot of the pass takes.  The 16 samples are ranked by cost (session length first, then window
    // length) and dealt out in that order: the two waves of a SIMD do not run alike -- the first-dispatched half of
    // the workgroup wins the issue arbitration and runs its phases ~30 % faster (MI355X_MICROARCH.md, two waves per
    // SIMD) -- so it gets the longest samples, and a wavefront's two samples have similar lengths (its trip counts
    // are the longer one's).  Any assignment gives the same sums up to fp32 rounding; this one is a fixed function
    // of the batch, so results stay bitwise reproducible.
    // (lane r of a row speaks for candidate r % NSB: with 8 samples per pass the upper half of a row repeats the lower)
    int bidx;
    int offv = 0, f_total = 0;   // FLAT: lane r = where slot r's window starts in the flat list; the list's length
    {
      const int rc = r % NSB;
      const int cand = a.perm != nullptr ? a.perm[g * NSB + rc] : g * NSB + rc;   // (the batch's samples ranked and dealt out: BalArgs)
      const bool cv = cand < B;
      const int cl = cv ? min(a.b.sl[cand], Ls) : 0, cs = cv ? min(a.b.sl_new[cand], Sn) : 0;
      // (a window per column group -- streamed windows with dropout -- : the window length decides, up to 90 positions
      //  against a session's few)
      const int key = cv ? ((((LSTREAM && !FLAT) ? ((cl << 12) | (cs << 4)) : ((cs << 12) | (cl << 4))) | (15 - rc)) + 1) : -rc;   // distinct; larger = heavier
      int rank = 0;
#pragma unroll
      for (int j = 0; j < NSB; ++j) rank += (__builtin_amdgcn_readlane(key, j) > key) ? 1 : 0;
      int* sPerm = (int*)T;                       // (the wave's own scratch: free until P3)
      // (windows held in registers, two samples per wavefront: the wavefront with the k-th longest session also takes
      // the k-th shortest -- its lanes then share the long one, see HELP in P3)
      const int slot = (SPW == 2 && !DROP) ? (rank < NSB / 2 ? 2 * rank : 2 * (NSB - 1 - rank) + 1) : rank;
      if (q == 0 && r < NSB) sPerm[slot] = r;
      if (FLAT && q == 0) sPerm[16 + slot] = cl;
      wave_lds_fence();
      bidx = __shfl(cand, sPerm[srow], 16);   // (candidate of lane sPerm[srow] of the row)
      if constexpr (FLAT) {   // every wavefront works out the whole list's layout for itself: no barrier
        const int nlj = sPerm[16 + r];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int x = __builtin_amdgcn_readlane(nlj, j);
          offv += (j < r) ? x : 0;
          f_total += x;
        }
      }
      wave_lds_fence();
    }
    const bool vs = bidx < B;
    DropCtx dc;
    dc.seed = a.drop_seed; dc.thr = a.drop_thr; dc.inv = a.drop_inv; dc.sbase = 2u * ((uint32_t)bidx + a.drop_sample0);
    const int bb = vs ? bidx : 0;
    const int uid = a.b.u[bb];
    const int it_i = a.b.i[bb];
    int n_l = vs ? a.b.sl[bb] : 0;
    n_l = min(n_l, Ls);
    int n_s = vs ? a.b.sl_new[bb] : 0;
    n_s = min(n_s, Sn);
    float loss_acc = 0.0f, sq_acc = 0.0f, dgam = 0.0f;
    // ------------------------------------------------------------------ P1: long block
    // All loads are unconditional on clamped (always valid) addresses so that they are issued
    // back to back: three dependent round trips (ids -> categories -> rows) for all LS positions
    // together; padding is applied afterwards by selects (padded slots contribute exactly 0).
    f32x4 e1[LS][NB], long4[NB], mx1[NB], iz1[NB];
    int idk_w = 0, ctk_w = 0;   // this lane's window entry (id, category): NB > 1 re-gathers the window's rows in P5 (E1RG below)
    int posv[(TRAIN && LSTREAM) ? LS + 3 : 1];   // (streamed window: the three single uses, drawn by the lead lane)
    int upos = 0;                                // (window in registers: the cursor draw of this lane's use slot)
    int ucpos = 0, cposv = 0, scpos0 = 0;        // (CSEG: the category-cursor draws of the same uses)
    const int pmax1 = wave_max_samples<CPS>(n_l);
    // ---- streamed long block (LSTREAM): ids, scales and positions live one per lane of the sample
    // (lane kk = entry base + kk of the current chunk) and are broadcast with cross-lane reads
    constexpr int NLc = 4 * CPS;
    const int kkl = q * CPS + col;
    int lid = 0, lct = 0;
    float lht = 0.0f, lut = 0.0f;
    auto load_lchunk = [&](int base) {
      const int t = min(base + kkl, Ls - 1);
      lid = a.b.hist_i[(size_t)bb * Ls + t];
      lht = a.b.hist_t[(size_t)bb * Ls + t];
      lut = a.p.usert_emb[(size_t)uid * a.p.ld_usert + t];
      lct = a.p.item_cate[lid];
    };
    // the NEXT chunk's entries, loaded while the current chunk is processed (its categories one position later, when
    // the ids have arrived) and taken over at the chunk boundary: no dependent id -> category -> row trip per chunk
    int nlid = 0, nlct = 0;
    float nlht = 0.0f, nlut = 0.0f;
    auto stage_lchunk_ids = [&](int base) {
      const int t = min(base + kkl, Ls - 1);
      nlid = a.b.hist_i[(size_t)bb * Ls + t];
      nlht = a.b.hist_t[(size_t)bb * Ls + t];
      nlut = a.p.usert_emb[(size_t)uid * a.p.ld_usert + t];
    };
    auto stage_lchunk_cats = [&]() { nlct = a.p.item_cate[nlid]; };
    auto take_lchunk = [&]() { lid = nlid; lct = nlct; lht = nlht; lut = nlut; };
    // entry p of the lane's own sample, from the loaded chunk
    auto pick_lentry = [&](int p, int& it, int& ct, float& uth) {
      const int k = p % NLc;
      it = sample_pick<CPS>(lid, k / CPS, k % CPS, s_loc);
      ct = sample_pick<CPS>(lct, k / CPS, k % CPS, s_loc);
      uth = sample_pick<CPS>(lut, k / CPS, k % CPS, s_loc) * sample_pick<CPS>(lht, k / CPS, k % CPS, s_loc);
    };
    auto fetch_lrow = [&](int p, f32x4 (&xr)[NB], float& scx, float& sce) {  // position p of the loaded chunk
      int it, ct;
      float uth;
      pick_lentry(p, it, ct, uth);
      scx = (gamma * P * P) * uth;  // x = e_stored * scx
      sce = (gamma * P) * uth;      // d x / d e_true
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) xr[kb] = gather_item4c<DT>(a, it, ct, CH_USE(chb[kb], it));
    };
    using raw4 = typename TblRaw<DT>::type;
    auto fetch_lrow_raw = [&](int p, raw4 (&xr)[NB], float& scx, float& sce) {  // the same, the row as loaded (widened at its use)
      int it, ct;
      float uth;
      pick_lentry(p, it, ct, uth);
      scx = (gamma * P * P) * uth;
      sce = (gamma * P) * uth;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) xr[kb] = gather_item4c_raw<DT>(a, it, ct, CH_USE(chb[kb], it));
    };
    // session ids (and their categories) are fetched once, one per lane of the sample
    // (lane k = q*CPS + col holds entry k of the current chunk of NL entries), then broadcast
    // with a cross-lane read: no dependent index loads inside the position loops.
    constexpr int NL = LKEY ? NLK : 4 * CPS;   // (LKEY: lanes kk >= NL hold clamped copies and neither file keys nor draw cursors)
    const int kk = q * CPS + col;
    int sid = 0, scat = 0;
    int* sSKi = sSK + wave * (2 * SPW * NLK);   // LKEY: this wavefront's [SPW][NLK] ids, then [SPW][NLK] categories
    int* sSKc = sSKi + SPW * NLK;
    auto file_chunk = [&]() {                   // (DS operations of a wavefront execute in order: the fences only bind the compiler)
      wave_lds_fence();
      if (kk < NL) {
        sSKi[s_loc * NLK + kk] = sid;
        sSKc[s_loc * NLK + kk] = scat;
      }
      wave_lds_fence();
    };
    auto load_chunk = [&](int base) {
      const int t = min(base + kk, max(Sn - 1, 0));
      sid = (Sn > 0) ? a.b.hist_i_new[(size_t)bb * Sn + t] : 0;
      scat = a.p.item_cate[sid];
      if constexpr (LKEY) file_chunk();
    };
    // row of session entry t0 + dt of the wavefront's sample s_sel (n_sel entries; its chunk must be loaded): normally
    // the lane's own sample and dt = 0; while one half of the lanes helps with a long session (HELP, see P3) both halves
    // fetch for that sample, the helping half the odd entry (t0, s_sel, n_sel and `two` are wave-uniform, dt per lane)
    // (the row comes back as loaded, with a flag: scaling or masking it right behind the load would make the wavefront
    //  wait for it there instead of where it is used -- see use_row)
    using srow4 = typename TblRaw<DT>::type;
    struct RowPF { srow4 r[NB]; bool v; };
    auto use_row = [&](const RowPF& o, f32x4 (&xr)[NB]) {
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) xr[kb] = o.v ? tbl_cvt<DT>(o.r[kb]) * P : (f32x4)(0.0f);
    };
    auto fetch_row_of = [&](int t0, int dt, bool two, RowPF& o, int s_sel, int n_sel) {
      const int k = t0 % NL;
      if constexpr (LKEY) {   // the key of the lane's half of the row: one LDS read at a per-lane address, one multiply-add
        const int kx = two ? min(k + dt, NL - 1) : k;
        const int key = ((g_item ? sSKi : sSKc) + s_sel * NLK)[kx];
        o.v = t0 + dt < n_sel;
        o.r[0] = tbl_ld4_raw<DT>(g_base, g_idx(key, g_ld, g_off));
        return;
      }
      int it = sample_pick<CPS>(sid, k / CPS, k % CPS, s_sel), ct = sample_pick<CPS>(scat, k / CPS, k % CPS, s_sel);
      if (two) {
        const int k1 = min(k + 1, NL - 1);
        const int it1 = sample_pick<CPS>(sid, k1 / CPS, k1 % CPS, s_sel), ct1 = sample_pick<CPS>(scat, k1 / CPS, k1 % CPS, s_sel);
        it = dt ? it1 : it;
        ct = dt ? ct1 : ct;
      }
      o.v = t0 + dt < n_sel;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) o.r[kb] = gather_item4c_raw<DT>(a, it, ct, CH_USE(chb[kb], it));
    };
    auto fetch_row = [&](int t, RowPF& o) { fetch_row_of(t, 0, false, o, s_loc, n_s); };
    int spos0 = 0;  // position of session use kk (first chunk): one returning atomic per use
    RowPF xnext;
    xnext.v = false;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) xnext.r[kb] = srow4{};
    int ucat = 0, ct_i = 0;   // category of the user's row / of the candidate: fetched with the window's ids
    if constexpr (FLAT) {
      // ---- FLAT: streamed windows as ONE work list per workgroup.  Window lengths are heavy-tailed (synthetic
      // histories: mean 14, 1 % at the cap of 90; real ones likewise) and a launch ends with its slowest wavefront:
      // with a window per column group -- even with a long one shared by a wavefront's two halves -- the wavefront
      // that holds a 90-entry window walks ~50 positions while the others of its workgroup walk ~15 and wait at the
      // barrier.  Here the (sample, position) pairs of the workgroup's 16 windows, in sample order, form a list of
      // f_total entries; column group j (slot j: its lanes) takes entries [j T, (j + 1) T), T = ceil(f_total / 16).
      // Forward: a group keeps an online-softmax state per run of entries of one sample and leaves it in the LDS
      // (slot sample + group: runs are enumerated in order, so the sum is unique); after a barrier each sample's own
      // lanes merge its runs in order.  Backward (P5): positions are independent given the sample's statistics,
      // output and output gradient, which every group reads from the LDS.  Results are a fixed function of the batch.
      ucat = a.b.u_cate[bb];
      ct_i = a.p.item_cate[it_i];
      opd FT1[NB][NB], FT2[NB][NB];
      f32x4 b1[NB], b2[NB];
      LD_T(w1W1, FT1);
      LD_T2(w1W2, FT2);
      load_bias<DH, NB>(w1b1, LD_Q, b1);
      LD_B2(w1b2, b2);
      if constexpr (TRAIN) {
        posv[LS] = (lead && vs) ? atomicAdd(&a.cur_item[it_i], 1) : 0;
        cposv = (lead && vs && CSEG) ? atomicAdd(&a.cur_uc[ct_i], 1) : 0;
        posv[LS + 1] = (lead && vs) ? atomicAdd(&a.cur_user[uid], 1) : 0;
        posv[LS + 2] = (lead && vs && (a.uc_by_sample + opaque_zero(lane)) == 0) ? atomicAdd(&a.cur_uc[a.b.u_cate[bb]], 1) : bidx;
      }
      if constexpr (!TRAIN) {
        if (lead) sSb[srow] = bidx;    // (FwdArgs.att0: whose scores a list entry's are)
      }
      const int wv = __builtin_amdgcn_readfirstlane(wave);
      int offm = __builtin_amdgcn_readlane(offv, wv * SPW);   // where this lane's own window starts in the list
#pragma unroll
      for (int sx = 1; sx < SPW; ++sx) {
        const int x = __builtin_amdgcn_readlane(offv, wv * SPW + sx);
        offm = (s_loc == sx) ? x : offm;
      }
      // the wavefront's own windows -> the list (lane kkl of a sample's lanes: entry base + kkl), with the cursor draws
      for (int base = 0; base < pmax1; base += NLc) {  // wave-uniform
        const int t = base + kkl, tc = min(t, Ls - 1);
        const int id = a.b.hist_i[(size_t)bb * Ls + tc];
        const float ht = a.b.hist_t[(size_t)bb * Ls + tc], ut = a.p.usert_emb[(size_t)uid * a.p.ld_usert + tc];
        const int ct = a.p.item_cate[id];
        if (t < n_l) {
          const int i = offm + t;
          sFid[i] = id;
          sFct[i] = ct;
          sFst[i] = (srow << 8) | t;
          sFht[i] = ht;
          sFuh[i] = ut * ht;
          if constexpr (TRAIN) {
            sFpos[i] = atomicAdd(&a.cur_item[id], 1);
            if constexpr (CSEG) sFcpos[i] = atomicAdd(&a.cur_uc[ct], 1);
          }
        }
      }
      __syncthreads();
      TLSAN_STAMP(21);
      const int FTn = (f_total + NSB - 1) / NSB;               // entries per column group (wave-uniform)
      const int i0 = srow * FTn, iend = min(i0 + FTn, f_total);
      const int ilast = max(f_total - 1, 0);
      struct Ent { int id, ct, st; float uh; };
      auto read_ent = [&](int idx, Ent& e) {
        const int ic = min(idx, ilast);
        e.id = sFid[ic];
        e.ct = sFct[ic];
        e.st = sFst[ic];
        e.uh = sFuh[ic];
      };
      auto issue = [&](const Ent& e, raw4 (&row)[NB]) {
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) row[kb] = gather_item4c_raw<DT>(a, e.id, e.ct, CH_USE(chb[kb], e.id));
      };
      f32x4 smx[NB], sZ[NB], sN[NB];
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        smx[kb] = (f32x4)(TLSAN_NEG);
        sZ[kb] = sN[kb] = (f32x4)(0.0f);
      }
      int s_prev = -1;
      // one entry: both maps, one online-softmax step into the state of the current run (a fresh one where the sample
      // changes), the state left in the run's slot.  No branches: an entry past the group's share changes nothing.
      auto compute = [&](int idx, int st, const f32x4 (&xv)[NB]) {
        const bool v = idx < iend;
        const float mk = v ? 1.0f : 0.0f;
        const int s_cur = st >> 8;
        const bool fresh = v && s_cur != s_prev;
        f32x4 z[NB], m2[NB];
        map_apply<NB, MM>(FT1, b1, xv, z);
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
          for (int i = 0; i < 4; ++i) z[kb][i] = fmaxf(z[kb][i], 0.0f);
        map_apply<NB, MM>(FT2, b2, z, m2);
        if constexpr (!TRAIN) {   // attention weights wanted: the raw scores now, normalised by the sample's own lanes after the merge
          if (a.att0 != nullptr && v) {
            const int bs = sSb[s_cur];
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) *(f32x4*)att_at(a.att0, Ls, bs, st & 255, chb[kb]) = m2[kb];
          }
        }
        const int slot = v ? s_cur + srow : 31;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float mx0 = fresh ? TLSAN_NEG : smx[kb][i], Z0 = fresh ? 0.0f : sZ[kb][i], N0 = fresh ? 0.0f : sN[kb][i];
            const float m2m = v ? m2[kb][i] : TLSAN_NEG;
            const float mn = fmaxf(mx0, m2m);
            const float so = exp2s(mx0 - mn), ev = exp2s(m2m - mn) * mk;
            sZ[kb][i] = Z0 * so + ev;
            sN[kb][i] = N0 * so + ev * xv[kb][i];
            smx[kb][i] = mn;
          }
          float* ps = sPart + slot * 3 * D + chl[kb];
          *(f32x4*)(ps) = smx[kb];
          *(f32x4*)(ps + D) = sZ[kb];
          *(f32x4*)(ps + 2 * D) = sN[kb];
        }
        s_prev = v ? s_cur : s_prev;
      };
      auto take = [&](int idx, const Ent& e, const raw4 (&row)[NB], f32x4 (&xv)[NB]) {
        const float scx = (idx < iend) ? (gamma * P * P) * e.uh : 0.0f;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) xv[kb] = tbl_cvt<DT>(row[kb]) * scx;
      };
      // (d = 256: one row at a time -- two in flight as at d <= 128 were no faster there, the kernel is bound by its
      //  matrix work and its spilled registers: C5 shape 439 vs 430 us/step)
      if (FTn > 0 && NB > 1 && DT == TLSAN_TABLE_F32) {
        // (d = 256, fp32 tables: one row in flight ahead of the entry being computed.  Nothing in round 3, when every reload
        //  of a spilled address drained the queue and the prefetch with it; under the final compiler switches the C5 shape
        //  goes 325 -> 317 us/step, small tables and bf16 tables stay equal; the same in the list's backward costs 26 us --
        //  32 more live registers there)
        Ent e, en;
        raw4 rw[NB], rn[NB];
        read_ent(i0, e);
        issue(e, rw);
        for (int k = 0; k < FTn; ++k) {
          f32x4 xa[NB];
          read_ent(i0 + k + 1, en);
          issue(en, rn);
          take(i0 + k, e, rw, xa);
          compute(i0 + k, e.st, xa);
          e = en;
#pragma unroll
          for (int kb = 0; kb < NB; ++kb) rw[kb] = rn[kb];
        }
      } else if (FTn > 0 && NB > 1) {
        for (int k = 0; k < FTn; ++k) {
          Ent e;
          raw4 rw[NB];
          f32x4 xa[NB];
          read_ent(i0 + k, e);
          issue(e, rw);
          take(i0 + k, e, rw, xa);
          compute(i0 + k, e.st, xa);
        }
      } else
      if (FTn > 0) {   // wave-uniform
        // NPF rows in flight, each fetched NPF entries ahead of its use (the item table does not fit an XCD's L2: a
        // gather is ~2 k cycles away, an entry's own work a few hundred); the entry after that is read from the list
        // meanwhile.  No stores in this loop: the compiler's own vmcnt counting keeps the later rows in flight.
        constexpr int NPF = 4;
        Ent eU[NPF], nU[NPF];
        raw4 rU[NPF][NB];
#pragma unroll
        for (int u = 0; u < NPF; ++u) read_ent(i0 + u, eU[u]);
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
          issue(eU[u], rU[u]);
          read_ent(i0 + NPF + u, nU[u]);
        }
        for (int k = 0; k < FTn; k += NPF) {
#pragma unroll
          for (int u = 0; u < NPF; ++u) {
            f32x4 xa[NB];
            take(i0 + k + u, eU[u], rU[u], xa);
            const int stU = eU[u].st;
            eU[u] = nU[u];
            issue(eU[u], rU[u]);
            read_ent(i0 + k + u + 2 * NPF, nU[u]);
            compute(i0 + k + u, stU, xa);
          }
        }
      }
      TLSAN_STAMP(22);
      __syncthreads();
      // the sample's own lanes merge its runs, in list order
      int gf = 0, gl = -1;
      if (n_l > 0) {
        gf = offm / FTn;
        gl = (offm + n_l - 1) / FTn;
      }
      int glo = __builtin_amdgcn_readlane(gf, 0), ghi = __builtin_amdgcn_readlane(gl, 0);
#pragma unroll
      for (int sx = 1; sx < SPW; ++sx) {
        const int f1 = __builtin_amdgcn_readlane(gf, sx * CPS), l1 = __builtin_amdgcn_readlane(gl, sx * CPS);
        const bool e0 = ghi < glo, e1 = l1 < f1;   // (a sample without entries has no runs)
        glo = e0 ? f1 : (e1 ? glo : min(glo, f1));
        ghi = e0 ? l1 : (e1 ? ghi : max(ghi, l1));
      }
      f32x4 Zl[NB];
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        mx1[kb] = (f32x4)(TLSAN_NEG);
        Zl[kb] = (f32x4)(0.0f);
        long4[kb] = (f32x4)(0.0f);
      }
      for (int gs = glo; gs <= ghi; ++gs) {  // wave-uniform
        const bool in = gs >= gf && gs <= gl;
        const int slot = in ? srow + gs : 31;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
          const float* ps = sPart + slot * 3 * D + chl[kb];
          const f32x4 pm = *(const f32x4*)(ps), pZ = *(const f32x4*)(ps + D), pN = *(const f32x4*)(ps + 2 * D);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float mn = fmaxf(mx1[kb][i], pm[i]);
            const float sa = exp2s(mx1[kb][i] - mn), sb = exp2s(pm[i] - mn);
            mx1[kb][i] = in ? mn : mx1[kb][i];
            Zl[kb][i] = in ? Zl[kb][i] * sa + pZ[i] * sb : Zl[kb][i];
            long4[kb][i] = in ? long4[kb][i] * sa + pN[i] * sb : long4[kb][i];
          }
        }
      }
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          iz1[kb][i] = Zl[kb][i] > 0.0f ? fast_rcp(Zl[kb][i]) : 0.0f;  // samples past the batch: no position
          long4[kb][i] *= iz1[kb][i];
        }
        if constexpr (TRAIN && FLATG) {
          if (vs) {
            *(f32x4*)(a.gStat + (size_t)bidx * 2 * D + chb[kb]) = mx1[kb];
            *(f32x4*)(a.gStat + (size_t)bidx * 2 * D + D + chb[kb]) = iz1[kb];
          }
        } else if constexpr (TRAIN) {
          *(f32x4*)(sMx + srow * LSTR + chb[kb]) = mx1[kb];
          *(f32x4*)(sIz + srow * LSTR + chb[kb]) = iz1[kb];
        }
      }
      if constexpr (TRAIN && FLATG) {
        if (lead) sBx[srow] = bb;
      }
      if constexpr (!TRAIN) {
        // (the raw scores were stored by whichever column groups walked this sample's entries, all before the barrier above)
        if (a.att0 != nullptr && vs) {
          for (int p = 0; p < Ls; ++p) {
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              float* pa = att_at(a.att0, Ls, bidx, p, chb[kb]);
              f32x4 w = (f32x4)(0.0f);
              if (p < n_l) {
                const f32x4 raw = *(const f32x4*)pa;
#pragma unroll
                for (int i = 0; i < 4; ++i) w[i] = exp2s(raw[i] - mx1[kb][i]) * iz1[kb][i];
              }
              *(f32x4*)pa = w;
            }
          }
        }
      }
    } else if constexpr (LSTREAM) {
      ucat = a.b.u_cate[bb];
      ct_i = a.p.item_cate[it_i];
      opd FT1[NB][NB], FT2[NB][NB];
      f32x4 b1[NB], b2[NB], Zl[NB];
      LD_T(w1W1, FT1);
      LD_T2(w1W2, FT2);
      load_bias<DH, NB>(w1b1, LD_Q, b1);
      LD_B2(w1b2, b2);
      if constexpr (TRAIN) {
        posv[LS] = (lead && vs) ? atomicAdd(&a.cur_item[it_i], 1) : 0;
        cposv = (lead && vs && CSEG) ? atomicAdd(&a.cur_uc[ct_i], 1) : 0;
        posv[LS + 1] = (lead && vs) ? atomicAdd(&a.cur_user[uid], 1) : 0;
        // (a per-lane copy of the flag: a scalar branch here would split the batch of returning atomics)
        posv[LS + 2] = (lead && vs && (a.uc_by_sample + opaque_zero(lane)) == 0) ? atomicAdd(&a.cur_uc[a.b.u_cate[bb]], 1) : bidx;
      }
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        mx1[kb] = (f32x4)(TLSAN_NEG);
        Zl[kb] = (f32x4)(0.0f);
        long4[kb] = (f32x4)(0.0f);
      }
      if constexpr (LCH) { stage_lchunk_ids(0); stage_lchunk_cats(); }
      for (int base = 0; base < pmax1; base += NLc) {  // wave-uniform
        if constexpr (LCH) {
          take_lchunk();
          if (base + NLc < pmax1) stage_lchunk_ids(base + NLc);
        } else {
          load_lchunk(base);
        }
        if constexpr (TRAIN) {
          const int t = base + kkl;
          if (t < Ls) {
            const bool vt = vs && t < n_l;
            sP[srow * PSTR + t] = vt ? atomicAdd(&a.cur_item[lid], 1) : 0;
            if constexpr (CSEG) sPc[srow * PSTR + t] = vt ? atomicAdd(&a.cur_uc[lct], 1) : 0;
            sH[srow * 2 * LSC + t] = vt ? lht : 0.0f;
            sH[srow * 2 * LSC + LSC + t] = vt ? lut * lht : 0.0f;
          }
        }
        const int pend = min(base + NLc, pmax1);
        // the row of position p + 1 is fetched while position p is processed (left to the top of its own iteration,
        // every position waited out a full gather: most of a streamed window's time)
        // (d = 256 has no registers for it: 567 -> 642 us at Ls = 90 with the prefetch, so it keeps the fetch in place;
        //  d = 128: 268 -> 241 us)
        raw4 xn[NB];
        float scxn = 0.0f, scen = 0.0f;
        if constexpr (LPF) fetch_lrow_raw(base, xn, scxn, scen);
        for (int p = base; p < pend; ++p) {
          f32x4 xv[NB], z[NB], m2[NB];
          float scx = scxn, sce = scen;
          (void)sce;
          if constexpr (LPF) {
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) xv[kb] = tbl_cvt<DT>(xn[kb]);
            if (p + 1 < pend) fetch_lrow_raw(p + 1, xn, scxn, scen);
          } else {
            fetch_lrow(p, xv, scx, sce);
          }
          if constexpr (LCH) {
            if (p == base + 2 && base + NLc < pmax1) stage_lchunk_cats();   // (the next chunk's ids are here by now)
          }
          const bool vp = p < n_l;
#pragma unroll
          for (int kb = 0; kb < NB; ++kb) xv[kb] = vp ? xv[kb] * scx : (f32x4)(0.0f);
          if constexpr (DROP) {
            f32x4 xd[NB];
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) xd[kb] = xv[kb] * drop_scale4(dc, 0, p, 0, chb[kb]);
            map_apply<NB, MM>(FT1, b1, xd, z);
          } else {
            map_apply<NB, MM>(FT1, b1, xv, z);
          }
#pragma unroll
          for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) z[kb][i] = fmaxf(z[kb][i], 0.0f);
          if constexpr (DROP) {
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) z[kb] *= drop_scale4(dc, 0, p, 1, chb[kb]);
          }
          map_apply<NB, MM>(FT2, b2, z, m2);
          if (vp) online_step<NB>(mx1, Zl, long4, m2, xv);
        }
      }
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          iz1[kb][i] = Zl[kb][i] > 0.0f ? fast_rcp(Zl[kb][i]) : 0.0f;  // samples past the batch: no position
          long4[kb][i] *= iz1[kb][i];
        }
    } else
    {
      // Four explicit issue stages, each one dependent round trip deep: ids / scales -> categories -> rows ->
      // cursor draws.  sched_barrier keeps the compiler from interleaving a stage's loads with their first uses
      // (left alone it serialises them into one round trip per position).  The returning atomics come LAST:
      // vmcnt retires in issue order, so rows issued behind them would not count as arrived before every atomic
      // of the wavefront has returned (about 3 k cycles with all CUs drawing at once) -- as it was when each
      // use drew its cursor from the lead lane in a block of its own, with a vmcnt(0) in the middle.
      // Now lane k of a sample's 4*CPS lanes owns use slot k (long positions, candidate, user, u_cate): ONE
      // atomic instruction per wavefront draws them all and lane k files its result in the LDS itself.
      // The window's ids, time weights, usert entries and categories are loaded ONE ENTRY PER LANE (lane k of the
      // sample's lanes: entry k) -- four coalesced wave instructions instead of forty that each fetch the same two
      // dwords for all lanes of a sample (the vector-memory front end, not latency, paces the head of the kernel) --
      // and handed to the sample's lanes with v_readlane (sample_pick).
      float sc1[LS];
      constexpr int NU = LS + 3;
      static_assert(4 * CPS >= NU, "a sample's lanes must cover its use slots");
      const int kku = q * CPS + col;
      const int kc = min(kku, Ls - 1);
      const int id_k = a.b.hist_i[(size_t)bb * Ls + kc];
      const float ht_k = a.b.hist_t[(size_t)bb * Ls + kc];
      const float ut_k = a.p.usert_emb[(size_t)uid * a.p.ld_usert + kc];
      ucat = a.b.u_cate[bb];
      ct_i = a.p.item_cate[it_i];   // (with the stage that already waits for the sample's scalars: no trip of its own later)
      sid = (Sn > 0) ? a.b.hist_i_new[(size_t)bb * Sn + min(kk, max(Sn - 1, 0))] : 0;   // first chunk of session ids (load_chunk(0)), with the window's ids
      // (round 2 measured the short block's first ids / categories / cursor draws / first row riding with these stages
      //  instead of three trips of their own at the start of P3: no gain in the step time then, profiles/r02_ab.  Round 4
      //  with the stamps: those trips were 1.9 k cycles at the head of P3 on every workgroup's critical path.  Now the ids
      //  and categories ride here (two registers) and the first ROW is issued at the end of P1, in flight over barrier 1
      //  and P2: P3's head 1.9 k -> 0.5 k cycles, critical path 68.2 k -> 67.0 k, step -0.4 us, d = 64 -0.9 us
      //  (profiles/r04_sess_prefetch_ab.md); the cursor draws stay in P3)
      __builtin_amdgcn_sched_barrier(0);
      const int ct_k = a.p.item_cate[id_k];
      scat = a.p.item_cate[sid];   // ... and their categories with the window's
      __builtin_amdgcn_sched_barrier(0);
      idk_w = id_k;
      ctk_w = ct_k;
      if constexpr (LKEY) {
        // The window's keys reach the sample's lanes through the LDS: lane k files (id, category) of entry k in the wave's
        // own scratch (free until P3), every lane reads back the ten keys of ITS half of the row -- item ids for the
        // lanes whose channels lie in item_emb, categories for the others -- as three 16-byte broadcast reads, and a
        // row address is one multiply-add on per-lane constants.  (v_readlane + select per key and lane, sample_pick,
        // were 5 vector instructions per pick, 20 picks, plus the table select per gather: ~250 of the kernel's 3 100.)
        int* sKI = (int*)T + 32;            // [SPW][16] item ids      (T[0, 32): sPerm, read for the last time above)
        int* sKC = sKI + SPW * 16;          // [SPW][16] categories
        if (kku < 16) {
          sKI[s_loc * 16 + kku] = id_k;
          sKC[s_loc * 16 + kku] = ct_k;
        }
        file_chunk();   // (the first chunk of session keys with them)
        const int* ksrc = (g_item ? sKI : sKC) + s_loc * 16;
        int keyv[12];
#pragma unroll
        for (int k4 = 0; k4 < 3; ++k4) {
          const int4 kq = *(const int4*)(ksrc + 4 * k4);
          keyv[4 * k4] = kq.x; keyv[4 * k4 + 1] = kq.y; keyv[4 * k4 + 2] = kq.z; keyv[4 * k4 + 3] = kq.w;
        }
        static_assert(LS <= 12, "three 16-byte reads cover the window's keys");
#pragma unroll
        for (int p = 0; p < LS; ++p) e1[p][0] = tbl_ld4<DT>(g_base, g_idx(keyv[p], g_ld, g_off));
      } else {
#pragma unroll
      for (int p = 0; p < LS; ++p) {
        const int it = sample_pick<CPS>(id_k, p / CPS, p % CPS, s_loc), ct = sample_pick<CPS>(ct_k, p / CPS, p % CPS, s_loc);
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) e1[p][kb] = gather_item4c<DT>(a, it, ct, chb[kb]);
      }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (TRAIN) {
        const bool is_long = kku < LS;
        const int id = is_long ? id_k : (kku == LS ? it_i : (kku == LS + 1 ? uid : ucat));
        int32_t* cur = kku <= LS ? a.cur_item : (kku == LS + 1 ? a.cur_user : a.cur_uc);
        const bool act = vs && (is_long ? kku < n_l : (kku < LS + 2 || (kku == LS + 2 && a.uc_by_sample == 0)));
        upos = act ? atomicAdd(cur + id, 1) : (kku == LS + 2 ? bidx : 0);   // (u_cate rows in sample order: position = sample)
        if constexpr (CSEG) ucpos = (act && kku <= LS) ? atomicAdd(&a.cur_uc[is_long ? ct_k : ct_i], 1) : 0;
      }
      __builtin_amdgcn_sched_barrier(0);
      // padded slots: scale 0 (model.py:384 gives them weight exactly 0; their clamped rows are finite table rows),
      // so no select on the rows themselves -- the forward can start on position 0 while later rows are in flight
      const float uth_k = (kku < n_l) ? ut_k * ht_k : 0.0f;
      if ((TRAIN || LKEY) && kku < LS) {   // lane k files its own entry: hist_t and usert * hist_t of the pass (0 on padded slots)
        sH[srow * 2 * LSC + kku] = (kku < n_l) ? ht_k : 0.0f;
        sH[srow * 2 * LSC + LSC + kku] = uth_k;
      }
      if constexpr (LKEY) {   // usert * hist_t of the sample's positions: read back from sH (filed above by lane k), three 16-byte reads
        wave_lds_fence();
        static_assert(LSC == 10, "usert * hist_t occupies floats 10 .. 19 of a sample's sH row");
        const f32x4 u2 = *(const f32x4*)(sH + srow * 2 * LSC + 8), u3 = *(const f32x4*)(sH + srow * 2 * LSC + 12), u4 = *(const f32x4*)(sH + srow * 2 * LSC + 16);
        const float uthv[10] = {u2[2], u2[3], u3[0], u3[1], u3[2], u3[3], u4[0], u4[1], u4[2], u4[3]};
#pragma unroll
        for (int p = 0; p < LS; ++p) sc1[p] = (gamma * P * P) * uthv[p];
      } else {
#pragma unroll
      for (int p = 0; p < LS; ++p) sc1[p] = (gamma * P * P) * sample_pick<CPS>(uth_k, p / CPS, p % CPS, s_loc);  // model.py:100-102,109
      }
      TLSAN_STAMP(21);
      opd FT1[NB][NB], FT2[NB][NB];
      f32x4 b1[NB], b2[NB];
      LD_T(w1W1, FT1);
      LD_T2(w1W2, FT2);
      load_bias<DH, NB>(w1b1, LD_Q, b1);
      LD_B2(w1b2, b2);
      TLSAN_STAMP(22);
      f32x4 att_r[TRAIN ? 1 : LS][NB];
      fwa_forward<NB, LS, DROP, MM>(FT1, b1, FT2, b2, e1, sc1, n_l, pmax1, mx1, iz1, long4, KEEP_A ? sAw : nullptr, dc, 0, chb,
                                    TRAIN ? nullptr : att_r);
      if constexpr (!TRAIN) {
        if (a.att0 != nullptr && vs) {
#pragma unroll
          for (int p = 0; p < LS; ++p)
            if (p < Ls) {
#pragma unroll
              for (int kb = 0; kb < NB; ++kb) *(f32x4*)att_at(a.att0, Ls, bidx, p, chb[kb]) = att_r[p][kb];
            }
        }
      }
    }
    TLSAN_STAMP(23);
    // (d = 256: the addresses below are formed HERE, from the ids, behind a zero the compiler cannot see through -- formed
    //  early they sat in spilled 64-bit registers, and each reload from scratch memory drained the vector-memory queue)
    const int ozt = (NB > 1 && MM == TLSAN_MATRIX_F32) ? opaque_zero(tid) : 0;   // (fp32 operands: 182 -> 177 us/step at Ls = 10; with bf16 operands the kernel spills elsewhere and this cost 0.8 us)
    const int bidx_t = bidx + ozt, uid_t = uid + ozt, ucat_t = ucat + ozt, it_t = it_i + ozt, ct_t = ct_i + ozt;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      *(f32x4*)(sA + srow * LSTR + chb[kb]) = long4[kb];
      if constexpr (TRAIN && G::FUSE_DK) {
        if (FUSE_RT || FLAT) *(f32x4*)(sL + srow * LSTR + chb[kb]) = long4[kb];
        if (!FUSE_RT && vs) *(f32x4*)(a.gLong + (size_t)bidx_t * D + chb[kb]) = long4[kb];
      } else if (TRAIN && vs) {
        *(f32x4*)(a.gLong + (size_t)bidx_t * D + chb[kb]) = long4[kb];
      }
    }
    // rows the short block needs that depend only on ids: issue now, consume after P2
    f32x4 uemb[NB], iemb[NB];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      const int c = chb[kb];
      {
        const bool usr = c < a.di;
        const float* ub = usr ? a.p.user_emb : a.p.cate_emb;
        const size_t ui = usr ? (size_t)uid_t * a.p.ld_user + c : (size_t)ucat_t * a.dc + (c - a.di);
        uemb[kb] = tbl_ld4<DT>(ub, ui) * P;
      }
      if constexpr (LKEY) iemb[kb] = tbl_ld4<DT>(g_base, g_idx(g_item ? it_t : ct_t, g_ld, g_off)) * P;
      else iemb[kb] = gather_item4c<DT>(a, it_t, ct_t, c) * P;
    }
    const float ib_i = a.p.item_b[(size_t)it_t * a.p.ld_itemb];
    if constexpr (!LSTREAM) {   // the first session row as well (ids and categories came with the window's): it is in flight over barrier 1 and P2
      if (wave_max_samples<CPS>(n_s + 1) > 1) fetch_row(0, xnext);
    }

    // B fragments of the bridge GEMM (K^T rows, L2-resident) do not depend on the barrier:
    // fetch them first so their latency overlaps the wait for the slowest wavefront
    // (only the first tile's: with two tiles per wavefront -- 8-sample workgroups -- the second's are fetched while the
    //  first is computed; all of them up front were 64 registers across the barrier and P1's tail spilled)
    opd bfr[D / 16];
    auto load_B = [&](const float* Bmat, int t, opd (&dst)[D / 16]) {
      const int ct = (wave * G::TPW + t) % G::NT;
      const float* Brow = Bmat + (size_t)(16 * ct + r) * D + 4 * q;
#pragma unroll
      for (int kc = 0; kc < D / 16; ++kc) dst[kc] = mm_pack<MM>(*(const f32x4*)(Brow + 16 * kc));
    };
    load_B(a.p.dense_KT, 0, bfr);
    TLSAN_STAMP(1);
    __syncthreads();
    TLSAN_STAMP(2);
    // ------------------------------------------------------------------ P2: bridge GEMM
    // bridge[s][j] = sum_k long[s][k] K[k][j] + k0[j]   (tf.layers.dense, model.py:347)
#pragma unroll
    for (int t = 0; t < G::TPW; ++t) {
      const int task = wave * G::TPW + t, rt = task / G::NT, ct = task % G::NT;
      opd bnx[D / 16];
      if (t + 1 < G::TPW) load_B(a.p.dense_KT, t + 1, bnx);
      f32x4 acc0 = (f32x4)(dn[a.lay.k0 + 16 * ct + r]), acc1 = (f32x4)(0.0f);
      const float* Arow = sA + (NSB >= 16 ? 16 * rt + r : (16 * rt + r) % NSB) * LSTR + 4 * q;   // (NSB = 8: rows 8..15 of the tile repeat 0..7, their results are dropped)
#pragma unroll
      for (int kc = 0; kc < D / 16; kc += 2) {
        const opd av0 = mm_pack<MM>(*(const f32x4*)(Arow + 16 * kc)), av1 = mm_pack<MM>(*(const f32x4*)(Arow + 16 * kc + 16));
        if constexpr (MM == TLSAN_MATRIX_F32) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc0 = TLSAN_MFMA(av0[s], bfr[kc][s], acc0);
            acc1 = TLSAN_MFMA(av1[s], bfr[kc + 1][s], acc1);
          }
        } else {   // (one 16x16x32 per pair of k-chunks, two accumulators alternating)
          if ((kc >> 1) & 1) acc1 = mm_mma2<MM>(av0, av1, bfr[kc], bfr[kc + 1], acc1);
          else acc0 = mm_mma2<MM>(av0, av1, bfr[kc], bfr[kc + 1], acc0);
        }
      }
      acc0 += acc1;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (NSB >= 16 || 4 * q < NSB) sB[(16 * rt + 4 * q + i) * LSTR + 16 * ct + r] = acc0[i];
      if (t + 1 < G::TPW) {
#pragma unroll
        for (int kc = 0; kc < D / 16; ++kc) bfr[kc] = bnx[kc];
      }
    }
    // (the short block's weight fragments: LDS reads issued before barrier 2, see P5)
    opd FT1[NB][NB], FT2[NB][NB];
    f32x4 b1[NB], b2[NB];
    LD_T(w2W1, FT1);
    LD_T2(w2W2, FT2);
    load_bias<DH, NB>(w2b1, LD_Q, b1);
    LD_B2(w2b2, b2);
    TLSAN_STAMP(3);
    __syncthreads();
    TLSAN_STAMP(4);
    // ------------------------------------------------------------------ P3: short block
    // positions: 0 = bridge, 1..n_s = session rows (model.py:350); streamed with an online
    // softmax so registers do not grow with the session length.
    const int n_pos = n_s + 1;  // model.py:355: rep_length = sl_new + 1
    const int pmax2 = wave_max_samples<CPS>(n_pos);
    if constexpr (TRAIN) {
      if constexpr (LSTREAM) {
        if (lead) {
          sP[srow * PSTR + P_TGT] = posv[LS];
          sP[srow * PSTR + P_USR] = posv[LS + 1];
          sP[srow * PSTR + P_UC] = posv[LS + 2];
          if constexpr (CSEG) sPc[srow * PSTR + P_TGT] = cposv;
        }
      } else {
        const int kku = q * CPS + col;   // use slot of this lane (P_TGT, P_USR, P_UC are consecutive)
        if (kku < LS + 3) sP[srow * PSTR + (kku < LS ? kku : P_TGT + (kku - LS))] = upos;
        if (CSEG && kku <= LS) sPc[srow * PSTR + (kku < LS ? kku : P_TGT)] = ucpos;
        // item_b's gradient is 0 for every use but the candidate's: the lane that drew a window use's position clears it
        // here, ONE store instruction for the wavefront's twenty uses (it was a lead-lane store inside a branch at every
        // position of the long backward: ten pairs of exec-mask edges in the pipelined loop)
        if (vs && kku < n_l) a.Gb[upos] = 0.0f;
      }
    }
    if constexpr (LSTREAM) load_chunk(0);
    if constexpr (TRAIN) {
      spos0 = (vs && kk < n_s && kk < NL) ? atomicAdd(&a.cur_item[sid], 1) : 0;
      if constexpr (CSEG) scpos0 = (vs && kk < n_s && kk < NL) ? atomicAdd(&a.cur_uc[scat], 1) : 0;
    }
    if constexpr (LSTREAM) { if (pmax2 > 1) fetch_row(0, xnext); }
    TLSAN_STAMP(24);
    f32x4 mx[NB], Zs[NB], short4[NB];
    {
      f32x4 xv[NB], z[NB];
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) xv[kb] = *(const f32x4*)(sB + srow * LSTR + chb[kb]);
      if constexpr (DROP) {
        f32x4 xd[NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) xd[kb] = xv[kb] * drop_scale4(dc, 1, 0, 0, chb[kb]);
        map_apply<NB, MM>(FT1, b1, xd, z);
      } else {
        map_apply<NB, MM>(FT1, b1, xv, z);
      }
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int i = 0; i < 4; ++i) z[kb][i] = fmaxf(z[kb][i], 0.0f);
      if constexpr (DROP) {
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) z[kb] *= drop_scale4(dc, 1, 0, 1, chb[kb]);
      }
      map_apply<NB, MM>(FT2, b2, z, mx);  // position 0 is always valid: running max = its score
      if constexpr (!TRAIN) {
        if (a.att1 != nullptr && vs) {
#pragma unroll
          for (int kb = 0; kb < NB; ++kb) *(f32x4*)att_at(a.att1, Sn + 1, bidx, 0, chb[kb]) = mx[kb];   // (raw, normalised below)
        }
      }
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        Zs[kb] = (f32x4)(1.0f);
        short4[kb] = xv[kb];
      }
    }
    // ---- HELP: a long session is shared by the wavefront's two halves.  The kernel ends with its slowest workgroup,
    // and that is the one holding the batch's longest session (P3 costs ~3.8 k cycles per session position; one
    // sample in a few thousand has six or more).  Once the shorter of the wavefront's two sessions is exhausted, its
    // half of the lanes takes every second remaining entry of the longer one: the MFMA columns are then (entry 2j,
    // 8 heads | entry 2j+1, 8 heads) of ONE sample, the helping half keeps a partial online-softmax state of its own,
    // and the two are merged with one cross-lane step (lane r <-> r ^ 8: row_ror:8) -- forward here, and in the
    // backward the helper reads the sample's statistics / output gradient the same way.  Session steps of the
    // wavefront: lo + ceil((hi - lo) / 2) instead of hi.
    constexpr bool HELP = SPW == 2 && !DROP;
    int h_lo = 0, h_hi = 0, h_L = 0;
    bool h_on = false;
    if constexpr (HELP) {
      const int ns0 = __builtin_amdgcn_readlane(n_s, 0), ns1 = __builtin_amdgcn_readlane(n_s, CPS);
      h_lo = min(ns0, ns1);
      h_hi = max(ns0, ns1);
      h_L = ns0 >= ns1 ? 0 : 1;
      h_on = (h_hi - h_lo >= 2) && h_hi <= NL;   // (sessions beyond one chunk of ids: rare, left to the plain loop)
      if constexpr (!TRAIN) h_on = h_on && a.att1 == nullptr;   // (attention weights wanted: every lane stays with its own sample)
    }
    const int nsess = h_on ? h_lo + (h_hi - h_lo + 1) / 2 : pmax2 - 1;   // session steps of this wavefront
    const bool helper = HELP && h_on && s_loc != h_L;
    struct Sel { int s, t0, dt, n; bool two; };   // (sample, even entry, +1 for the helping half, its length, shared step)
    auto sel_of = [&](int u) -> Sel {             // session step u of this wavefront -> what this lane works on
      if (HELP && h_on && u >= h_lo) return Sel{h_L, h_lo + 2 * (u - h_lo), helper ? 1 : 0, h_hi, true};
      return Sel{s_loc, u, 0, n_s, false};
    };
    if constexpr (HELP) {
      if (h_on && pmax2 > 1 && h_lo == 0) {   // (the row prefetched above was the lane's own entry 0: the helper needs entry 1)
        const Sel e = sel_of(0);
        fetch_row_of(e.t0, e.dt, e.two, xnext, e.s, e.n);
      }
    }
    for (int u = 0; u < nsess; ++u) {  // wave-uniform trip count
      const int p = u + 1;             // (position of the step in the block: 0 is the bridge)
      const Sel e = sel_of(u);
      const bool vt = e.t0 + e.dt < e.n;
      if constexpr (HELP) {
        if (h_on && u == h_lo && helper) {   // the helping half parks its own (finished) state and starts a partial one
#pragma unroll
          for (int kb = 0; kb < NB; ++kb) {
            *(f32x4*)(T + ((0 * NB + kb) * 64 + lane) * 4) = mx[kb];
            *(f32x4*)(T + ((1 * NB + kb) * 64 + lane) * 4) = Zs[kb];
            *(f32x4*)(T + ((2 * NB + kb) * 64 + lane) * 4) = short4[kb];
            mx[kb] = (f32x4)(TLSAN_NEG);
            Zs[kb] = (f32x4)(0.0f);
            short4[kb] = (f32x4)(0.0f);
          }
        }
      }
      f32x4 xv[NB], z[NB], m2[NB];
      if constexpr (G::AT_USE && !LSTREAM && G::AT_USE_T) {  // weight fragments from LDS at the use
        const int zz = opaque_zero(p);
        LD_T(w2W1 + zz, FT1);
        load_bias<DH, NB>(w2b1 + zz, LD_Q, b1);
        LD_T2(w2W2 + zz, FT2);
        LD_B2(w2b2 + zz, b2);
      }

      use_row(xnext, xv);
      if (u + 1 < nsess) {  // prefetch the next row while this one is processed
        if (!h_on && (p % NL) == 0) load_chunk(p);
        const Sel e2 = sel_of(u + 1);
        fetch_row_of(e2.t0, e2.dt, e2.two, xnext, e2.s, e2.n);
      }
      if constexpr (DROP) {
        f32x4 xd[NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) xd[kb] = xv[kb] * drop_scale4(dc, 1, p, 0, chb[kb]);
        map_apply<NB, MM>(FT1, b1, xd, z);
      } else {
        map_apply<NB, MM>(FT1, b1, xv, z);
      }
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int i = 0; i < 4; ++i) z[kb][i] = fmaxf(z[kb][i], 0.0f);
      if constexpr (DROP) {
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) z[kb] *= drop_scale4(dc, 1, p, 1, chb[kb]);
      }
      map_apply<NB, MM>(FT2, b2, z, m2);
      if constexpr (!TRAIN) {
        if (a.att1 != nullptr && vs && vt) {
#pragma unroll
          for (int kb = 0; kb < NB; ++kb) *(f32x4*)att_at(a.att1, Sn + 1, bidx, p, chb[kb]) = m2[kb];
        }
      }
      if (vt) {
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float mn = fmaxf(mx[kb][i], m2[kb][i]);
            const float so = exp2s(mx[kb][i] - mn), ev = exp2s(m2[kb][i] - mn);
            Zs[kb][i] = Zs[kb][i] * so + ev;
            short4[kb][i] = short4[kb][i] * so + ev * xv[kb][i];
            mx[kb][i] = mn;
          }
      }
    }
    if constexpr (HELP) {
      if (h_on) {   // merge the helper's partial state into the sample's, give the helper its own state back
        wave_lds_fence();
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
          const f32x4 own_mx = *(const f32x4*)(T + ((0 * NB + kb) * 64 + lane) * 4);
          const f32x4 own_Z = *(const f32x4*)(T + ((1 * NB + kb) * 64 + lane) * 4);
          const f32x4 own_N = *(const f32x4*)(T + ((2 * NB + kb) * 64 + lane) * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float o_mx = dpp_f32<TLSAN_DPP_ROR(8)>(mx[kb][i]), o_Z = dpp_f32<TLSAN_DPP_ROR(8)>(Zs[kb][i]);
            const float o_N = dpp_f32<TLSAN_DPP_ROR(8)>(short4[kb][i]);
            const float mn = fmaxf(mx[kb][i], o_mx);
            const float sa = exp2s(mx[kb][i] - mn), sb = exp2s(o_mx - mn);
            mx[kb][i] = helper ? own_mx[i] : mn;
            Zs[kb][i] = helper ? own_Z[i] : Zs[kb][i] * sa + o_Z * sb;
            short4[kb][i] = helper ? own_N[i] : short4[kb][i] * sa + o_N * sb;
          }
        }
        wave_lds_fence();
      }
    }
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Zs[kb][i] = fast_rcp(Zs[kb][i]);
        short4[kb][i] *= Zs[kb][i];
      }
    if constexpr (!TRAIN) {
      if (a.att1 != nullptr && vs) {   // raw scores -> weights (model.py:386), zero past the session
        for (int p = 0; p <= Sn; ++p) {
#pragma unroll
          for (int kb = 0; kb < NB; ++kb) {
            float* pa = att_at(a.att1, Sn + 1, bidx, p, chb[kb]);
            f32x4 w = (f32x4)(0.0f);
            if (p <= n_s) {
              const f32x4 raw = *(const f32x4*)pa;
#pragma unroll
              for (int i = 0; i < 4; ++i) w[i] = exp2s(raw[i] - mx[kb][i]) * Zs[kb][i];
            }
            *(f32x4*)pa = w;
          }
        }
      }
    }
    if constexpr (TRAIN) {  // publish the session positions (the atomics are long back by now)
      if (vs && kk < n_s && kk < NL) {
        sP[srow * PSTR + LSCP + kk] = spos0;
        a.Gb[spos0] = 0.0f;      // (a session use's item_b gradient is 0: cleared by the lane that drew the position)
      }
      if (CSEG && vs && kk < n_s && kk < NL) sPc[srow * PSTR + LSCP + kk] = scpos0;
      for (int base = NL; base < pmax2 - 1; base += NL) {  // sessions longer than one chunk (rare)
        const int t = base + kk;
        if (vs && t < n_s && kk < NL) {
          const int sidt = a.b.hist_i_new[(size_t)bb * Sn + t];
          const int spt = atomicAdd(&a.cur_item[sidt], 1);
          sP[srow * PSTR + LSCP + t] = spt;
          a.Gb[spt] = 0.0f;
          if constexpr (CSEG) sPc[srow * PSTR + LSCP + t] = atomicAdd(&a.cur_uc[a.p.item_cate[sidt]], 1);
        }
      }
    }
    TLSAN_STAMP(25);
    // u_t = short + [user_emb[u] || cate_emb[u_cate]]   (model.py:93-95,135)
    f32x4 ut4[NB];
    float part = 0.0f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      ut4[kb] = short4[kb] + uemb[kb];
      part += dot4(ut4[kb], iemb[kb]);
      if constexpr (!TRAIN) {   // (evaluation's output: not a branch of the training kernel)
        if (a.u_t != nullptr && vs) *(f32x4*)(a.u_t + (size_t)bidx * D + chb[kb]) = ut4[kb];
      }
    }
    const float logit = sample_sum<CPS>(part) + ib_i;  // model.py:137
    if (lead && vs && a.logits_i != nullptr) a.logits_i[bidx] = logit;
    if (!TRAIN && a.b.j != nullptr && a.logits_j != nullptr) {  // second candidate (eval_auc's negative; never in a train step)
      const int it_j = a.b.j[bb];
      float pj = 0.0f;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) pj += dot4(ut4[kb], gather_item4<DT>(a, it_j, chb[kb]) * P);
      const float lj = sample_sum<CPS>(pj) + a.p.item_b[(size_t)it_j * a.p.ld_itemb];
      if (lead && vs) a.logits_j[bidx] = lj;
    }
    TLSAN_STAMP(5);
    if constexpr (TRAIN) {
      float* prec = a.partials + (size_t)g * G::NPB;
      // BCE with logits, mean over the batch (model.py:171)
      const float yv = a.b.y[bb];
      const float en = __expf(-fabsf(logit));
      const float lb = fmaxf(logit, 0.0f) - logit * yv + __logf(1.0f + en);
      const float rden = fast_rcp(1.0f + en);
      const float sg = logit >= 0.0f ? rden : en * rden;
      const float dl = vs ? (sg - yv) * a.inv_B : 0.0f;
      const int pos_t = sP[srow * PSTR + P_TGT], pos_u = sP[srow * PSTR + P_USR], pos_c = sP[srow * PSTR + P_UC];
      const int cpos_t = CSEG ? sPc[srow * PSTR + P_TGT] : 0;
      if (lead && vs) {
        a.Gb[pos_t] = dl;  // per-use item_b gradient
        loss_acc += lb;
        sq_acc += dl * dl;
      }
      f32x4 dout[NB], dk0[NB];
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        dout[kb] = iemb[kb] * dl;  // d loss / d u_t
        dk0[kb] = (f32x4)(0.0f);
        const f32x4 gi = ut4[kb] * dl;
        if (vs) {
          const int c = chb[kb];
          // user use: [user_emb half] -> Gu (grouped by user), [u_cate half] -> Gc (grouped by category)
          float* up = (c < a.di) ? a.Gu + (size_t)pos_u * a.WU + c : a.Gc + (size_t)pos_c * a.dc + (c - a.di);
          st4_out(up, dout[kb]);
          st4_out(use_dst(pos_t, cpos_t, c), gi);  // candidate use
          sq_acc += dot4(dout[kb], dout[kb]) + dot4(gi, gi);
        }
      }
      TLSAN_STAMP(26);
      {
        opd FN1[NB][NB], FN2[NB][NB];
        LD_N(w2W1, FN1);
        LD_N(w2W2, FN2);
        AccSet<NB> acc;
        acc.zero();
        if (pmax2 - 1 > NL) load_chunk(0);  // (wave-uniform) the forward loop moved past chunk 0
        f32x4 xbr[NB];      // position 0: the bridge row (LDS)
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) xbr[kb] = *(const f32x4*)(sB + srow * LSTR + chb[kb]);
        RowPF xn2;          // positions 1..: the session row fetched one step ahead ...
        xn2.v = false;
        // (d = 256 keeps the take-over at the top of the row's own step: eight more live registers there cost the streamed
        //  bf16 variants 12 %: 241 -> 270 us/step at d = 256, Ls = 90)
        constexpr bool P3TAKE = NB == 1;
        f32x4 xnv[NB];      // ... and taken over at the END of the step that fetched it, before that step's stores: a wait
                            // for a load that is older than a store waits for the store as well (vmcnt retires in order,
                            // the stores sit in branches), so taken at the top of its own step it waited out the
                            // previous step's stores
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
          xn2.r[kb] = srow4{};
          xnv[kb] = (f32x4)(0.0f);
        }
        for (int p = 0; p <= nsess; ++p) {  // wave-uniform trip count: the bridge, then the session steps
          const Sel e = sel_of(p > 0 ? p - 1 : 0);
          const int s_sel = e.s, t = e.t0 + e.dt;
          const bool vt = p == 0 || t < e.n;
          const bool shared = p > 0 && e.two;                          // (wave-uniform) both halves work on one sample
          const bool oth = shared && helper;                           // this lane works for the other half's sample
          f32x4 xv[NB], z1[NB], zr[NB], m2[NB], av[NB], dx[NB];
          if constexpr (G::AT_USE && !LSTREAM) {  // weight fragments from LDS at the use
            const int zz = opaque_zero(p);
            if constexpr (G::AT_USE_T) {
              LD_T(w2W1 + zz, FT1);
              load_bias<DH, NB>(w2b1 + zz, LD_Q, b1);
              LD_T2(w2W2 + zz, FT2);
              LD_B2(w2b2 + zz, b2);
            }
            LD_N(w2W1 + zz, FN1);
            LD_N(w2W2 + zz, FN2);
          }

          if (p == 0) {
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) xv[kb] = xbr[kb];
          } else if constexpr (P3TAKE) {
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) xv[kb] = xnv[kb];
          } else {
            use_row(xn2, xv);
          }
          if (p < nsess) {  // the row of the next session step
            if (!h_on && p > 0 && (p % NL) == 0) load_chunk(p);
            const Sel e2 = sel_of(p);
            fetch_row_of(e2.t0, e2.dt, e2.two, xn2, e2.s, e2.n);
          }
          f32x4 k1[NB], k2[NB];
          if constexpr (DROP) {
            f32x4 xd[NB];
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              k1[kb] = drop_scale4(dc, 1, p, 0, chb[kb]);
              k2[kb] = drop_scale4(dc, 1, p, 1, chb[kb]);
              xd[kb] = xv[kb] * k1[kb];
            }
            map_apply<NB, MM>(FT1, b1, xd, z1);
          } else {
            map_apply<NB, MM>(FT1, b1, xv, z1);
          }
#pragma unroll
          for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i) zr[kb][i] = fmaxf(z1[kb][i], 0.0f);
          if constexpr (DROP) {
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) zr[kb] *= k2[kb];
          }
          map_apply<NB, MM>(FT2, b2, zr, m2);
          // statistics, output and output gradient of the sample this lane works for (the other half's, read
          // across the row, while helping)
          f32x4 outs[NB], douts[NB];
          if (shared) {
#pragma unroll
            for (int kb = 0; kb < NB; ++kb)
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float o_mx = dpp_f32<TLSAN_DPP_ROR(8)>(mx[kb][i]), o_Z = dpp_f32<TLSAN_DPP_ROR(8)>(Zs[kb][i]);
                const float o_out = dpp_f32<TLSAN_DPP_ROR(8)>(short4[kb][i]), o_do = dpp_f32<TLSAN_DPP_ROR(8)>(dout[kb][i]);
                const float mxs = oth ? o_mx : mx[kb][i], zss = oth ? o_Z : Zs[kb][i];
                outs[kb][i] = oth ? o_out : short4[kb][i];
                douts[kb][i] = oth ? o_do : dout[kb][i];
                av[kb][i] = vt ? exp2s(m2[kb][i] - mxs) * zss : 0.0f;
              }
          } else {
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              outs[kb] = short4[kb];
              douts[kb] = dout[kb];
#pragma unroll
              for (int i = 0; i < 4; ++i) av[kb][i] = vt ? exp2s(m2[kb][i] - mx[kb][i]) * Zs[kb][i] : 0.0f;
            }
          }
          bwd_compute<NB, TSTR, DROP, MM>(FN2, FN1, xv, z1, av, outs, douts, T, q, r, acc.db1, acc.db2, dx, k1, k2);
          bwd_dw<NB, TSTR, MM>(T, q, r, acc.dW1, acc.dW2);
          if (P3TAKE && p < nsess) {   // (wave-uniform)
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) asm volatile("" : "+v"(xn2.r[kb]));   // (the fetched registers are read HERE)
            use_row(xn2, xnv);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (p == 0) {
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              *(f32x4*)(sA + srow * LSTR + chb[kb]) = dx[kb];  // dbridge
              if (vs) {
                if (!G::FUSE_DK || !FUSE_RT) *(f32x4*)(a.gDB + (size_t)bidx * D + chb[kb]) = dx[kb];
                dk0[kb] += dx[kb];
              }
            }
          } else if ((vs || oth) && vt) {
            const int pos = sP[(wave * SPW + s_sel) * PSTR + LSCP + t];
            const int cpos = CSEG ? sPc[(wave * SPW + s_sel) * PSTR + LSCP + t] : 0;
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              st4_out(use_dst(pos, cpos, chb[kb]), dx[kb]);
              sq_acc += dot4(dx[kb], dx[kb]);
            }
          }
        }
        TLSAN_STAMP(27);
        if constexpr (G::SPLIT) {  // two halves through the (smaller) staging area
          stage_part<NB, CPS, true, 0>(acc, dk0, T, lane);
          __syncthreads();
          reduce_part<G, true, 0>(sT, prec, G::P_F2W1, G::P_F2B1, tid);
          __syncthreads();
          stage_part<NB, CPS, true, 1>(acc, dk0, T, lane);
        } else {
          stage_accs<NB, CPS, true>(acc, dk0, T, lane);
        }
      }
      load_B(dn + a.lay.K, 0, bfr);  // B fragments of the dlong GEMM (K rows), before the barrier
      TLSAN_STAMP(6);
      __syncthreads();
      TLSAN_STAMP(7);
      if constexpr (G::SPLIT) reduce_part<G, true, 1>(sT, prec, G::P_F2W2, G::P_F2B2, tid);
      else reduce_staged<G, true>(sT, prec, G::P_F2W1, G::P_F2B1, G::P_F2W2, G::P_F2B2, tid);
      TLSAN_STAMP(30);
      // ---------------------------------------------------------------- P4: dlong GEMM
      // dlong[s][k] = sum_j dbridge[s][j] K[k][j]
#pragma unroll
      for (int t = 0; t < G::TPW; ++t) {
        const int task = wave * G::TPW + t, rt = task / G::NT, kt = task % G::NT;
        opd bnx[D / 16];
        if (t + 1 < G::TPW) load_B(dn + a.lay.K, t + 1, bnx);
        f32x4 acc0 = (f32x4)(0.0f), acc1 = (f32x4)(0.0f);
        const float* Arow = sA + (NSB >= 16 ? 16 * rt + r : (16 * rt + r) % NSB) * LSTR + 4 * q;
#pragma unroll
        for (int jc = 0; jc < D / 16; jc += 2) {
          const opd av0 = mm_pack<MM>(*(const f32x4*)(Arow + 16 * jc)), av1 = mm_pack<MM>(*(const f32x4*)(Arow + 16 * jc + 16));
          if constexpr (MM == TLSAN_MATRIX_F32) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              acc0 = TLSAN_MFMA(av0[s], bfr[jc][s], acc0);
              acc1 = TLSAN_MFMA(av1[s], bfr[jc + 1][s], acc1);
            }
          } else {
            if ((jc >> 1) & 1) acc1 = mm_mma2<MM>(av0, av1, bfr[jc], bfr[jc + 1], acc1);
            else acc0 = mm_mma2<MM>(av0, av1, bfr[jc], bfr[jc + 1], acc0);
          }
        }
        acc0 += acc1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (NSB >= 16 || 4 * q < NSB) sB[(16 * rt + 4 * q + i) * LSTR + 16 * kt + r] = acc0[i];
        if (t + 1 < G::TPW) {
#pragma unroll
          for (int jc = 0; jc < D / 16; ++jc) bfr[jc] = bnx[jc];
        }
      }
      TLSAN_STAMP(31);
      if constexpr (G::FUSE_DK) {
        if (FUSE_RT && tid == 0) *(int*)(sS + 3) = 0;   // ticket counter of the pass's dK tile groups (taken behind P5; this barrier and the next lie between)
      }
      // the long block's weight fragments (LDS copies that no barrier guards) are read BEFORE barrier 4: the compiler does
      // not move LDS reads across a barrier, and P5's first map waited for them (kernel -0.5 us, Movies-TV shape -1.5 us/step: profiles/r04_frag_early_ab.md)
      opd FN1[NB][NB], FN2[NB][NB];
      LD_T(w1W1, FT1);
      load_bias<DH, NB>(w1b1, LD_Q, b1);
      LD_T2(w1W2, FT2);
      LD_B2(w1b2, b2);
      LD_N(w1W1, FN1);
      LD_N(w1W2, FN2);
      TLSAN_STAMP(8);
      __syncthreads();
      TLSAN_STAMP(9);
      // ---------------------------------------------------------------- P5: long backward
      {
        f32x4 dlong[NB], dummy[NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) dlong[kb] = *(const f32x4*)(sB + srow * LSTR + chb[kb]);
        // ---- pieces of the software-pipelined loops below (PIPE5: window in registers; FLAT: the streamed windows' list)
        static_assert(!PIPE5 || NB == 1, "one 16-channel block per column");
        f32x4 ta[NB], tb[NB];   // transposed tiles of the previous position: (x, dz1), then (m1, dm2)
        auto read_tiles = [&](int t0, int t1) {
#pragma unroll
          for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const int rofs = (4 * q + s) * TSTR + r;
              ta[kb][s] = T[(t0 * NB + kb) * 16 * TSTR + rofs];
              tb[kb][s] = T[(t1 * NB + kb) * 16 * TSTR + rofs];
            }
        };
        auto dw_prod = [&](f32x4 (&dW)[NB][NB]) {
#pragma unroll
          for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) dW[kb][jb] = mm_mma<MM>(mm_pack<MM>(ta[kb]), mm_pack<MM>(tb[jb]), dW[kb][jb]);
        };
        // a map as two half-chains whose sum is taken a group later (the sum right behind the chain would wait for it)
        auto map_issue = [&](const opd (&F)[NB][NB], const f32x4 (&v)[NB], f32x4 (&h0)[NB], f32x4 (&h1)[NB], const f32x4* bias) {
          if constexpr (MM == TLSAN_MATRIX_F32) {
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) {
              h0[ob] = bias ? TLSAN_MFMA(F[ob][0][0], v[0][0], bias[ob]) : TLSAN_MFMA(F[ob][0][0], v[0][0], (f32x4)(0.0f));
              h1[ob] = TLSAN_MFMA(F[ob][0][2], v[0][2], (f32x4)(0.0f));
              h0[ob] = TLSAN_MFMA(F[ob][0][1], v[0][1], h0[ob]);
              h1[ob] = TLSAN_MFMA(F[ob][0][3], v[0][3], h1[ob]);
            }
          } else {
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) {
              h0[ob] = bias ? mm_mma<MM>(F[ob][0], mm_pack<MM>(v[0]), bias[ob]) : mm_mma<MM>(F[ob][0], mm_pack<MM>(v[0]), (f32x4)(0.0f));
              h1[ob] = (f32x4)(0.0f);
            }
          }
        };
        if constexpr (LSTREAM) {
          AccSet<NB> acc;
          acc.zero();
          if constexpr (FLATG) {
          // ---- FLAT at d = 256 (see P1): this column group's share of the workgroup's list, one entry after the other
          // (the three-stage pipeline of d <= 128 below does not fit the registers).  The next entry's row, statistics
          // and long-term vector are fetched at the top of an iteration and taken over just before its stores.
          const int FTn = (f_total + NSB - 1) / NSB;
          const int i0 = srow * FTn, iend = min(i0 + FTn, f_total);
          const int ilast = max(f_total - 1, 0);
          if (FTn > 0) {   // wave-uniform
          raw4 en[NB];
          f32x4 mxn[NB], izn[NB], lon[NB];
          int stN = 0;
          float uhN = 0.0f;
          auto fetch = [&](int idx) {
            const int ic = min(idx, ilast);
            const int id = sFid[ic], ct = sFct[ic];
            stN = sFst[ic];
            uhN = sFuh[ic];
            const int sb = sBx[stN >> 8];
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) en[kb] = gather_item4c_raw<DT>(a, id, ct, CH_USE(chb[kb], id));
            const float* gs = a.gStat + (size_t)sb * 2 * D;
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              mxn[kb] = *(const f32x4*)(gs + chb[kb]);
              izn[kb] = *(const f32x4*)(gs + D + chb[kb]);
              lon[kb] = *(const f32x4*)(a.gLong + (size_t)sb * D + chb[kb]);
            }
          };
          f32x4 ev[NB], mxc[NB], izc[NB], loc[NB];
          float uhc = 0.0f;
          int stc = 0;
          bool vc = false;
          auto take = [&](int idx) {
            vc = idx < iend;
            stc = stN;
            uhc = vc ? uhN : 0.0f;
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              ev[kb] = vc ? tbl_cvt<DT>(en[kb]) : (f32x4)(0.0f);
              mxc[kb] = mxn[kb];
              izc[kb] = izn[kb];
              loc[kb] = lon[kb];
            }
          };
          // (the next entry's row / statistics in flight over the iteration, taken over before the stores: measured, no
          //  faster at d = 256)
          for (int k = 0; k < FTn; ++k) {       // wave-uniform
            fetch(i0 + k);
            take(i0 + k);
            const int ic = min(i0 + k, ilast);
            const int sc = stc >> 8, tcur = stc & 255;
            const int posp = sFpos[ic], cposp = CSEG ? sFcpos[ic] : 0;
            const float htc = sFht[ic];
            const int posu = sP[sc * PSTR + P_USR];
            const float scx = (gamma * P * P) * uhc, sce = (gamma * P) * uhc;
            const float mk = vc ? 1.0f : 0.0f;
            f32x4 dl[NB], xv[NB], z1[NB], zr[NB], m2[NB], av[NB], dx[NB];
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              dl[kb] = *(const f32x4*)(sB + sc * LSTR + chb[kb]);
              xv[kb] = ev[kb] * scx;
            }
            map_apply<NB, MM>(FT1, b1, xv, z1);
#pragma unroll
            for (int kb = 0; kb < NB; ++kb)
#pragma unroll
              for (int i = 0; i < 4; ++i) zr[kb][i] = fmaxf(z1[kb][i], 0.0f);
            map_apply<NB, MM>(FT2, b2, zr, m2);
#pragma unroll
            for (int kb = 0; kb < NB; ++kb)
#pragma unroll
              for (int i = 0; i < 4; ++i) av[kb][i] = exp2s(fminf(m2[kb][i] - mxc[kb][i], 0.0f)) * (izc[kb][i] * mk);
            bwd_compute<NB, TSTR, false, MM>(FN2, FN1, xv, z1, av, loc, dl, T, q, r, acc.db1, acc.db2, dx);
            bwd_dw<NB, TSTR, MM>(T, q, r, acc.dW1, acc.dW2);
            float dsp = 0.0f;
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) dsp += dot4(dx[kb], ev[kb]);
            const float ds = sample_sum<CPS>(dsp) * P;  // d loss / d scale of this entry
            const bool vst = vc;
            const float uhs = uhc;
            __builtin_amdgcn_sched_barrier(0);
            if (vst) {
              if (lead) a.Gb[posp] = 0.0f;
#pragma unroll
              for (int kb = 0; kb < NB; ++kb) {
                const f32x4 de = dx[kb] * sce;
                st4_out(use_dst(posp, cposp, chb[kb]), de);
                sq_acc += dot4(de, de);
              }
              if (lead) {
                const float gt = ds * (gamma * htc);  // d / d usert_emb[u][position]
                a.Gu[(size_t)posu * a.WU + a.di + tcur] = gt;
                sq_acc += gt * gt;
                dgam += ds * (P * uhs);
              }
            }
          }
          }
          } else if constexpr (FLAT) {
          // ---- FLAT (see P1): this column group's share of the workgroup's list, through a software pipeline three
          // stages deep -- an iteration runs the FORWARD recomputation of entry k (z1 = x W1 + b1, m2 = relu(z1) W2 + b2,
          // a = exp(m2 - max) / sum), the BACKWARD maps of entry k - 1 and the two dW products of entry k - 2, whose
          // transposed operands it reads back from the LDS first, with the rows of entries k + 1 and k + 2 in flight.
          // MFMA groups of an iteration, each one's inputs made at least a group earlier:
          //   z1(k) . dm1(k-1) . m2(k) . dW1(k-2) . dxm(k-1) . dW2(k-2)
          // -- 24 MFMAs back to back with the vector code, the exponentials, the LDS traffic and the row stores in their
          // shadow, where a plain loop runs six dependent map / product chains and an LDS round trip per entry one after
          // the other.  First iteration: no backward entry yet (zero operands, nothing stored, the tiles zeroed below);
          // last: the forward entry past the share (a zero row).  An entry's sample can be any of the workgroup's: its
          // statistics, output and output gradient come from the LDS (read at the top of the iteration that runs its
          // forward stage; the backward stage's pair rides along a stage).
          // Vector-memory operations of an iteration: the row fetch at its top, the row stores at its end, the fetched
          // row taken over just BEFORE those stores: vmcnt retires in order and the stores sit in branches the compiler
          // cannot count, so any wait for a load that is older than a store waits for the store as well -- taken over at
          // the top of its own iteration, every row waited out the previous iteration's stores, a full round trip.
          const int wofs = r * TSTR + 4 * q;
#pragma unroll
          for (int t = 0; t < 4; ++t) *(f32x4*)(T + t * 16 * TSTR + wofs) = (f32x4)(0.0f);
          const int FTn = (f_total + NSB - 1) / NSB;
          const int i0 = srow * FTn, iend = min(i0 + FTn, f_total);
          const int ilast = max(f_total - 1, 0);
          if (FTn > 0) {   // wave-uniform
          // TWO rows in flight (E0, E1: entry j in E[j & 1]), the loop unrolled by two so that each buffer's fetch and
          // take-over are straight-line code: iteration k fetches entry k + 2 at its top and takes entry k + 1 over just
          // before its stores -- a gather is ~2 k cycles away (the item table does not fit an XCD's L2), an iteration is
          // ~1.6 k, and with one row in flight every iteration waited for the fetch it had issued itself.
          struct RowBuf { raw4 r[NB]; int st; float uh; };
          RowBuf E0, E1;
          int idN, ctN, stN, stF = 0, stB = 0;   // entry after the ones in flight / forward stage / backward stage
          float uhN;
          f32x4 xvF[NB], evF[NB], z1F[NB], avF[NB], xvB[NB], evB[NB], loB[NB], dlB[NB];
#pragma unroll
          for (int kb = 0; kb < NB; ++kb) z1F[kb] = avF[kb] = xvB[kb] = evB[kb] = loB[kb] = dlB[kb] = (f32x4)(0.0f);
          float sceF = 0.0f, sceB = 0.0f;
          bool vF = false, vB = false;
          int posuB = 0;
          auto fetch_into = [&](RowBuf& E) {    // the entry read from the list last goes out; the one after it is read
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) E.r[kb] = gather_item4c_raw<DT>(a, idN, ctN, CH_USE(chb[kb], idN));
            E.st = stN;
            E.uh = uhN;
          };
          auto read_next = [&](int idx) {
            const int ic = min(idx, ilast);
            idN = sFid[ic];
            ctN = sFct[ic];
            stN = sFst[ic];
            uhN = sFuh[ic];
          };
          auto take_row = [&](int idx, const RowBuf& E) {   // a fetched row becomes the forward stage's entry
            vF = idx < iend;
            const float uth = vF ? E.uh : 0.0f;
            stF = E.st;
            sceF = (gamma * P) * uth;      // d x / d e_true
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              evF[kb] = vF ? tbl_cvt<DT>(E.r[kb]) : (f32x4)(0.0f);
              xvF[kb] = evF[kb] * ((gamma * P * P) * uth);  // x = e_stored * scale
            }
          };
          read_next(i0);
          fetch_into(E0);
          read_next(i0 + 1);
          fetch_into(E1);
          read_next(i0 + 2);
          take_row(i0, E0);
          auto body = [&](int k, RowBuf& Ef, RowBuf& Et) __attribute__((always_inline)) {
            f32x4 z1[NB], av[NB], m1[NB], dm2[NB], dz1[NB], dx[NB], ha[NB], hb[NB], hc[NB], hd[NB], mxf[NB], izf[NB], loF[NB], dlF[NB];
            const float sce = sceB;
            const bool stv = vB;
            const float mkF = vF ? 1.0f : 0.0f;
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              z1[kb] = z1F[kb];
              av[kb] = avF[kb];
            }
            fetch_into(Ef);                                          // entry k + 2
            read_next(i0 + k + 3);
            // backward stage's entry: where its rows go; forward stage's entry: its sample's vectors
            const int ib = max(min(i0 + k - 1, ilast), 0);
            const int posp = sFpos[ib], cposp = CSEG ? sFcpos[ib] : 0;
            const float htB = sFht[ib], uhB = sFuh[ib];
            const int sF = stF >> 8;
            const int posuF = sP[sF * PSTR + P_USR];
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              mxf[kb] = *(const f32x4*)(sMx + sF * LSTR + chb[kb]);
              izf[kb] = *(const f32x4*)(sIz + sF * LSTR + chb[kb]);
              loF[kb] = *(const f32x4*)(sL + sF * LSTR + chb[kb]);
              dlF[kb] = *(const f32x4*)(sB + sF * LSTR + chb[kb]);
            }
            read_tiles(0, 1);                                        // x, dz1 of entry k - 2
            __builtin_amdgcn_sched_barrier(0);
            map_issue(FT1, xvF, ha, hb, b1);                         // G1: z1(k)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) dm2[kb] = av[kb] * dlB[kb] * (xvB[kb] - loB[kb]);  // softmax-over-positions backward
            __builtin_amdgcn_sched_barrier(0);
            map_issue(FN2, dm2, hc, hd, nullptr);                    // G2: dm1(k - 1) = dm2 . W2^T
            __builtin_amdgcn_sched_barrier(0);
            f32x4 zrF[NB];
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              z1F[kb] = ha[kb] + hb[kb];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                zrF[kb][i] = fmaxf(z1F[kb][i], 0.0f);
                m1[kb][i] = fmaxf(z1[kb][i], 0.0f);
              }
              acc.db2[kb] += dm2[kb];
            }
            __builtin_amdgcn_sched_barrier(0);
            map_issue(FT2, zrF, ha, hb, b2);                         // G3: m2(k)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              const f32x4 dm1 = hc[kb] + hd[kb];
#pragma unroll
              for (int i = 0; i < 4; ++i) dz1[kb][i] = z1[kb][i] > 0.0f ? dm1[i] : 0.0f;
            }
            __builtin_amdgcn_sched_barrier(0);
            dw_prod(acc.dW1);                                        // G4: dW1 += x^T dz1 of entry k - 2
            read_tiles(2, 3);                                        //     its m1, dm2 (before this entry's tiles overwrite them)
            __builtin_amdgcn_sched_barrier(0);
            map_issue(FN1, dz1, hc, hd, nullptr);                    // G5: dxm(k - 1) = dz1 . W1^T
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
              for (int i = 0; i < 4; ++i)   // (m2 <= max on valid entries: the clamp changes nothing there; no branch around the exponentials)
                avF[kb][i] = exp2s(fminf((ha[kb][i] + hb[kb][i]) - mxf[kb][i], 0.0f)) * (izf[kb][i] * mkF);
              *(f32x4*)(T + (0 * NB + kb) * 16 * TSTR + wofs) = xvB[kb];  // entry k - 1's tiles (behind the reads above: in-order DS)
              *(f32x4*)(T + (1 * NB + kb) * 16 * TSTR + wofs) = dz1[kb];
              *(f32x4*)(T + (2 * NB + kb) * 16 * TSTR + wofs) = m1[kb];
              *(f32x4*)(T + (3 * NB + kb) * 16 * TSTR + wofs) = dm2[kb];
              acc.db1[kb] += dz1[kb];
            }
            __builtin_amdgcn_sched_barrier(0);
            dw_prod(acc.dW2);                                        // G6: dW2 += m1^T dm2 of entry k - 2
            __builtin_amdgcn_sched_barrier(0);
            float dsp = 0.0f;
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              dx[kb] = av[kb] * dlB[kb] + (hc[kb] + hd[kb]);
              dsp += dot4(dx[kb], evB[kb]);
            }
            const float ds = sample_sum<CPS>(dsp) * P;  // d loss / d scale of entry k - 1
            const int tB = stB & 255, posu = posuB;
            // the forward stage's entry moves on to the backward stage; entry k + 1's row takes its place -- BEFORE the stores
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              xvB[kb] = xvF[kb];
              evB[kb] = evF[kb];
              loB[kb] = loF[kb];
              dlB[kb] = dlF[kb];
            }
            sceB = sceF;
            vB = vF;
            stB = stF;
            posuB = posuF;
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) asm volatile("" : "+v"(Et.r[kb]));   // (the fetched registers are read HERE: left alone the compiler sinks the take-over below the stores' branches)
            take_row(i0 + k + 1, Et);
            __builtin_amdgcn_sched_barrier(0);
            if (stv) {
              if (lead) a.Gb[posp] = 0.0f;
#pragma unroll
              for (int kb = 0; kb < NB; ++kb) {
                const f32x4 de = dx[kb] * sce;
                st4_out(use_dst(posp, cposp, chb[kb]), de);
                sq_acc += dot4(de, de);
              }
              if (lead) {
                const float gt = ds * (gamma * htB);  // d / d usert_emb[u][position]
                a.Gu[(size_t)posu * a.WU + a.di + tB] = gt;
                sq_acc += gt * gt;
                dgam += ds * (P * uhB);
              }
            }
          };
          TLSAN_STAMP(16);
          for (int k = 0; k <= FTn; k += 2) {   // wave-uniform (an odd count runs one empty iteration more)
            body(k, E0, E1);
            body(k + 1, E1, E0);
          }
          TLSAN_STAMP(18);
          read_tiles(0, 1);                     // drain: the dW products of the last entry
          dw_prod(acc.dW1);
          read_tiles(2, 3);
          dw_prod(acc.dW2);
          }
          } else {
          if constexpr (LCH) { stage_lchunk_ids(0); stage_lchunk_cats(); }
          for (int base = 0; base < pmax1; base += NLc) {  // wave-uniform
            if constexpr (LCH) {
              take_lchunk();
              if (base + NLc < pmax1) stage_lchunk_ids(base + NLc);
            } else {
              load_lchunk(base);
            }
            const int pend = min(base + NLc, pmax1);
            raw4 en[NB];                  // (the next position's row, in flight while this one is processed)
            float scxn = 0.0f, scen = 0.0f;
            if constexpr (LPF) fetch_lrow_raw(base, en, scxn, scen);
            for (int p = base; p < pend; ++p) {
              const bool vp = p < n_l;
              f32x4 ev[NB], xv[NB], z1[NB], zr[NB], m2[NB], av[NB], dx[NB];
              float scx = scxn, sce = scen;
              if constexpr (LPF) {
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) ev[kb] = tbl_cvt<DT>(en[kb]);
                if (p + 1 < pend) fetch_lrow_raw(p + 1, en, scxn, scen);
              } else {
                fetch_lrow(p, ev, scx, sce);
              }
              if constexpr (LCH) {
                if (p == base + 2 && base + NLc < pmax1) stage_lchunk_cats();
              }
#pragma unroll
              for (int kb = 0; kb < NB; ++kb) {
                ev[kb] = vp ? ev[kb] : (f32x4)(0.0f);
                xv[kb] = ev[kb] * scx;
              }
              f32x4 k1[NB], k2[NB];
              if constexpr (DROP) {
                f32x4 xd[NB];
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) {
                  k1[kb] = drop_scale4(dc, 0, p, 0, chb[kb]);
                  k2[kb] = drop_scale4(dc, 0, p, 1, chb[kb]);
                  xd[kb] = xv[kb] * k1[kb];
                }
                map_apply<NB, MM>(FT1, b1, xd, z1);
              } else {
                map_apply<NB, MM>(FT1, b1, xv, z1);
              }
#pragma unroll
              for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i) zr[kb][i] = fmaxf(z1[kb][i], 0.0f);
              if constexpr (DROP) {
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) zr[kb] *= k2[kb];
              }
              map_apply<NB, MM>(FT2, b2, zr, m2);
#pragma unroll
              for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  av[kb][i] = vp ? exp2s(m2[kb][i] - mx1[kb][i]) * iz1[kb][i] : 0.0f;
              bwd_compute<NB, TSTR, DROP, MM>(FN2, FN1, xv, z1, av, long4, dlong, T, q, r, acc.db1, acc.db2, dx, k1, k2);
              bwd_dw<NB, TSTR, MM>(T, q, r, acc.dW1, acc.dW2);
              float dsp = 0.0f;
#pragma unroll
              for (int kb = 0; kb < NB; ++kb) dsp += dot4(dx[kb], ev[kb]);
              const float ds = sample_sum<CPS>(dsp) * P;  // d loss / d scale[p]
              if (vs && vp) {
                const int pos = sP[srow * PSTR + p];
                const int cpos = CSEG ? sPc[srow * PSTR + p] : 0;
                if (lead) a.Gb[pos] = 0.0f;
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) {
                  const f32x4 de = dx[kb] * sce;
                  st4_out(use_dst(pos, cpos, chb[kb]), de);
                  sq_acc += dot4(de, de);
                }
                if (lead) {
                  const float gt = ds * (gamma * sH[srow * 2 * LSC + p]);  // d / d usert_emb[u][p]
                  a.Gu[(size_t)sP[srow * PSTR + P_USR] * a.WU + a.di + p] = gt;
                  sq_acc += gt * gt;
                  dgam += ds * (P * sH[srow * 2 * LSC + LSC + p]);
                }
              }
            }
          }
          }
          if (lead && vs)  // padded long slots and the row's alignment padding
            for (int p = a.di + n_l; p < a.WU; ++p) a.Gu[(size_t)sP[srow * PSTR + P_USR] * a.WU + p] = 0.0f;
          if constexpr (G::SPLIT) {
          stage_part<NB, CPS, false, 0>(acc, dummy, T, lane);
          __syncthreads();
          reduce_part<G, false, 0>(sT, prec, G::P_F1W1, G::P_F1B1, tid);
          __syncthreads();
          stage_part<NB, CPS, false, 1>(acc, dummy, T, lane);
        } else {
          stage_accs<NB, CPS, false>(acc, dummy, T, lane);
        }
        } else {
        AccSet<NB> acc;
        acc.zero();
        float dsp[LS];  // per-lane partials of d loss / d scale[p]; reduced after the loop
        if constexpr (PIPE5) {
        // ---- software pipeline, skewed by one position: iteration p runs the map chain of position p and the two dW
        // products of position p - 1 (whose transposed operands it reads back from the LDS first).  Left to itself
        // the compiler emits every position as map (4 MFMAs) -> wait -> vector code -> map -> ... -> LDS round trip ->
        // dW (8 MFMAs), one dependent chain of ~1.1 k cycles with the matrix pipe idle between the links; here the
        // MFMA sequence of an iteration is  z1(p) . dm1(p) . dW1(p-1) . dxm(p) . dW2(p-1)  -- each group's inputs were
        // produced at least one group earlier, so the groups issue back to back and the vector code, the LDS writes of
        // this position's tiles and the LDS reads of the previous one's run in their shadow.  One transpose buffer
        // suffices: LDS operations of a wavefront execute in order, so the reads of p - 1 (top of the iteration) precede
        // the writes of p.  The groups are pinned with sched_barrier (the scheduler would re-cluster each chain).
        const int wofs = r * TSTR + 4 * q;
#pragma unroll
        for (int p = 0; p < LS; ++p) dsp[p] = 0.0f;
        int posp_c = sP[srow * PSTR], cposp_c = CSEG ? sPc[srow * PSTR] : 0;
        float uth_c = sH[srow * 2 * LSC + LSC];
        f32x4 av_c[NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) av_c[kb] = *(const f32x4*)(sAw + kb * 256);
#pragma unroll
        for (int p = 0; p <= LS; ++p) {
          if (p < LS && p < pmax1) {   // wave-uniform
            const bool vp = p < n_l;
            if (p == 1) TLSAN_STAMP(16);
            if (p == 2) TLSAN_STAMP(17);
            if (p == 9) TLSAN_STAMP(18);
            const int posp = posp_c, cposp = cposp_c;
            const float uth = uth_c;
            const float scp = (gamma * P * P) * uth;  // x = e_stored * scp
            const float sce = (gamma * P) * uth;      // d x / d e_true
            f32x4 xv[NB], av[NB], z1[NB], m1[NB], dm2[NB], dz1[NB], dx[NB], ha[NB], hb[NB], hc[NB], hd[NB];
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              xv[kb] = e1[p][kb] * scp;
              av[kb] = av_c[kb];
            }
            if (p >= 1) read_tiles(0, 1);                            // x, dz1 of p - 1
            if (p + 1 < LS) {                                        // the next position's LDS operands, a whole iteration ahead
              posp_c = sP[srow * PSTR + p + 1];
              if constexpr (CSEG) cposp_c = sPc[srow * PSTR + p + 1];
              uth_c = sH[srow * 2 * LSC + LSC + p + 1];
#pragma unroll
              for (int kb = 0; kb < NB; ++kb) av_c[kb] = *(const f32x4*)(sAw + ((p + 1) * NB + kb) * 256);
            }
            __builtin_amdgcn_sched_barrier(0);
            map_issue(FT1, xv, ha, hb, b1);                          // G1: z1(p)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) dm2[kb] = av[kb] * dlong[kb] * (xv[kb] - long4[kb]);  // softmax-over-positions backward
            __builtin_amdgcn_sched_barrier(0);
            map_issue(FN2, dm2, hc, hd, nullptr);                    // G2: dm1(p) = dm2 . W2^T
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              z1[kb] = ha[kb] + hb[kb];
#pragma unroll
              for (int i = 0; i < 4; ++i) m1[kb][i] = fmaxf(z1[kb][i], 0.0f);
              acc.db2[kb] += dm2[kb];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (p >= 1) {
              dw_prod(acc.dW1);                                      // G3: dW1(p-1) += x^T dz1
              read_tiles(2, 3);                                      //     m1, dm2 of p - 1 (before this position's tiles overwrite them)
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              const f32x4 dm1 = hc[kb] + hd[kb];
#pragma unroll
              for (int i = 0; i < 4; ++i) dz1[kb][i] = z1[kb][i] > 0.0f ? dm1[i] : 0.0f;
            }
            __builtin_amdgcn_sched_barrier(0);
            map_issue(FN1, dz1, ha, hb, nullptr);                    // G4: dxm(p) = dz1 . W1^T
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {                        // this position's tiles (behind the reads above: in-order DS)
              *(f32x4*)(T + (0 * NB + kb) * 16 * TSTR + wofs) = xv[kb];
              *(f32x4*)(T + (1 * NB + kb) * 16 * TSTR + wofs) = dz1[kb];
              *(f32x4*)(T + (2 * NB + kb) * 16 * TSTR + wofs) = m1[kb];
              *(f32x4*)(T + (3 * NB + kb) * 16 * TSTR + wofs) = dm2[kb];
              acc.db1[kb] += dz1[kb];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (p >= 1) dw_prod(acc.dW2);                            // G5: dW2(p-1) += m1^T dm2
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
              dx[kb] = av[kb] * dlong[kb] + (ha[kb] + hb[kb]);
              dsp[p] += dot4(dx[kb], e1[p][kb]);
            }
            if (p == 1) TLSAN_STAMP(15);
            if (vs && vp) {     // (Gb[posp]: cleared by the lane that drew the position, head of P3)
#pragma unroll
              for (int kb = 0; kb < NB; ++kb) {
                const f32x4 de = dx[kb] * sce;
                st4_out(use_dst(posp, cposp, chb[kb]), de);
                sq_acc += dot4(de, de);
              }
            }
          } else if (p >= 1 && p == pmax1) {   // drain: the dW products of the last position
            read_tiles(0, 1);
            dw_prod(acc.dW1);
            read_tiles(2, 3);
            dw_prod(acc.dW2);
          }
        }
        } else {
        // E1RG (two 16-channel blocks per column, d = 256): the window's rows are NOT carried from P1 to here in registers
        // (80 of them, live across three phases: the stamps showed 27 k cycles of spill traffic at the end of P1 and 19 k at
        // the end of P3 where d = 128 spends 1 k) but fetched again, one position ahead -- they are L2 / Infinity-Cache
        // resident, and a reload from scratch memory costs the same trip.  d = 256, Ls = 10: 206 -> 183 us/step, with bf16
        // tables and operands 178 -> 139 (profiles/r04_d256_regather_ab.md); spilled registers 166 -> 114 / 223 -> 84
        constexpr bool E1RG = NB > 1 && !LSTREAM;
        f32x4 e_nx[NB];
        auto e1_fetch = [&](int p_, f32x4 (&dst)[NB]) {
          const int it = sample_pick<CPS>(idk_w, p_ / CPS, p_ % CPS, s_loc), ct = sample_pick<CPS>(ctk_w, p_ / CPS, p_ % CPS, s_loc);
#pragma unroll
          for (int kb = 0; kb < NB; ++kb) dst[kb] = gather_item4c<DT>(a, it, ct, chb[kb]);
        };
        if constexpr (E1RG) e1_fetch(0, e_nx);
#pragma unroll
        for (int p = 0; p < LS; ++p) {
          dsp[p] = 0.0f;
          if (p < pmax1) {
            const bool vp = p < n_l;
            if (p == 1) TLSAN_STAMP(16);
            if (p == 2) TLSAN_STAMP(17);
            if (p == 9) TLSAN_STAMP(18);
            const int posp = sP[srow * PSTR + p];   // (read with the position's other LDS operands; used by its stores)
            const int cposp = CSEG ? sPc[srow * PSTR + p] : 0;
            const float uth = sH[srow * 2 * LSC + LSC + p];
            const float scp = (gamma * P * P) * uth;  // x = e_stored * scp
            const float sce = (gamma * P) * uth;      // d x / d e_true
            f32x4 xv[NB], z1[NB], zr[NB], m2[NB], av[NB], dx[NB];
          if constexpr (G::AT_USE && !LSTREAM) {  // weight fragments from LDS at the use
            const int zz = opaque_zero(p);
            if constexpr (G::AT_USE_T) {
              LD_T(w1W1 + zz, FT1);
              load_bias<DH, NB>(w1b1 + zz, LD_Q, b1);
              LD_T2(w1W2 + zz, FT2);
              LD_B2(w1b2 + zz, b2);
            }
            LD_N(w1W1 + zz, FN1);
            LD_N(w1W2 + zz, FN2);
          }
            f32x4 ep[NB];   // the position's row: from P1's registers, or (E1RG) fetched again one position ahead
            if constexpr (E1RG) {
#pragma unroll
              for (int kb = 0; kb < NB; ++kb) ep[kb] = e_nx[kb];
              if (p + 1 < LS && p + 1 < pmax1) e1_fetch(p + 1, e_nx);
            } else {
#pragma unroll
              for (int kb = 0; kb < NB; ++kb) ep[kb] = e1[p][kb];
            }
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) xv[kb] = ep[kb] * scp;
            if (p == 1) TLSAN_STAMP(12);
            f32x4 k1[NB], k2[NB];
            if constexpr (DROP) {
              f32x4 xd[NB];
#pragma unroll
              for (int kb = 0; kb < NB; ++kb) {
                k1[kb] = drop_scale4(dc, 0, p, 0, chb[kb]);
                k2[kb] = drop_scale4(dc, 0, p, 1, chb[kb]);
                xd[kb] = xv[kb] * k1[kb];
              }
              map_apply<NB, MM>(FT1, b1, xd, z1);
            } else {
              map_apply<NB, MM>(FT1, b1, xv, z1);
            }
            if constexpr (KEEP_A) {
#pragma unroll
              for (int kb = 0; kb < NB; ++kb) av[kb] = *(const f32x4*)(sAw + (p * NB + kb) * 256);
            } else {
#pragma unroll
              for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i) zr[kb][i] = fmaxf(z1[kb][i], 0.0f);
              if constexpr (DROP) {
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) zr[kb] *= k2[kb];
              }
              map_apply<NB, MM>(FT2, b2, zr, m2);
#pragma unroll
              for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  av[kb][i] = vp ? exp2s(m2[kb][i] - mx1[kb][i]) * iz1[kb][i] : 0.0f;
            }
            if (p == 1) TLSAN_STAMP(13);
            bwd_compute<NB, TSTR, DROP, MM>(FN2, FN1, xv, z1, av, long4, dlong, T, q, r, acc.db1, acc.db2, dx, k1, k2);
            if (p == 1) TLSAN_STAMP(14);
            bwd_dw<NB, TSTR, MM>(T, q, r, acc.dW1, acc.dW2);
            if (p == 1) TLSAN_STAMP(15);
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) dsp[p] += dot4(dx[kb], ep[kb]);
            if (vs && vp) {     // (Gb[posp]: cleared by the lane that drew the position, head of P3)
#pragma unroll
              for (int kb = 0; kb < NB; ++kb) {
                const f32x4 de = dx[kb] * sce;
                st4_out(use_dst(posp, cposp, chb[kb]), de);
                sq_acc += dot4(de, de);
              }
            }
          }
        }
        }
        TLSAN_STAMP(19);
        // usert_emb / gamma gradients.  Branch-free: every lane of a sample forms the same ten sums (vector-ALU
        // cross-lane steps only, independent chains), the sample's hist_t / usert*hist_t rows come from the LDS
        // as five 16-B reads, and the lead lane stores its [Ls | pad] piece of the user's gradient row as 16-B
        // pieces.  (As ten lead-lane blocks, each with its own LDS reads and scalar store, this took 7.5 k cycles.)
        {
          static_assert(LSC % 2 == 0 && (2 * LSC) % 4 == 0, "sH rows are read as float4");
          constexpr int NH4 = 2 * LSC / 4;
          f32x4 sh4[NH4];
#pragma unroll
          for (int k = 0; k < NH4; ++k) sh4[k] = *(const f32x4*)(sH + srow * 2 * LSC + 4 * k);
          const int pos_u = sP[srow * PSTR + P_USR];
          const float own = (lead && vs) ? 1.0f : 0.0f;   // one lane per sample counts
          constexpr int NG = (LS + 3) / 4 + 1;            // float4 pieces that can hold [0, WU - di): Ls <= LS entries + padding
          f32x4 gt4[NG];
#pragma unroll
          for (int k = 0; k < NG; ++k) gt4[k] = (f32x4)(0.0f);
#pragma unroll
          for (int p = 0; p < LS; ++p) {
            const float ds = sample_sum<CPS>(dsp[p]) * P;  // d loss / d scale[p] (e_true = P * e_stored)
            const bool vp = p < n_l;                        // (n_l <= Ls: padded slots and the row's padding stay 0)
            const float ht = sh4[p / 4][p % 4], uth = sh4[(LSC + p) / 4][(LSC + p) % 4];
            const float gt = vp ? ds * (gamma * ht) : 0.0f;  // d / d usert_emb[u][p]
            gt4[p / 4][p % 4] = gt;
            sq_acc += own * (gt * gt);
            dgam += (vp ? ds * (P * uth) : 0.0f) * own;
          }
          if (lead && vs) {
            float* gu = a.Gu + (size_t)pos_u * a.WU + a.di;
            const int n4 = (a.WU - a.di) >> 2;
#pragma unroll
            for (int k = 0; k < NG; ++k)
              if (k < n4) st4_out(gu + 4 * k, gt4[k]);
          }
        }
        TLSAN_STAMP(20);
        if constexpr (G::SPLIT) {
          stage_part<NB, CPS, false, 0>(acc, dummy, T, lane);
          __syncthreads();
          reduce_part<G, false, 0>(sT, prec, G::P_F1W1, G::P_F1B1, tid);
          __syncthreads();
          stage_part<NB, CPS, false, 1>(acc, dummy, T, lane);
        } else {
          stage_accs<NB, CPS, false>(acc, dummy, T, lane);
        }
        }
      }
      // scalars of this pass: wave-reduce, stage, one thread sums the waves in fixed order
      {
        const float s0 = wave_sum(dgam), s1 = wave_sum(loss_acc), s2 = wave_sum(sq_acc);
        if (lane == 0) {
          sS[wave * 4 + 0] = s0;
          sS[wave * 4 + 1] = s1;
          sS[wave * 4 + 2] = s2;
        }
      }
      if constexpr (G::FUSE_DK) {
       if (FUSE_RT) {
        // ---- dK partial of this pass: C[k][j] = sum over the pass's samples of long[s][k] * dbridge[s][j]
        // (A from sL, B from sA: both [sample][channel] rows in the LDS, untouched since P4).  Nothing in the kernel
        // consumes it, so it is not formed between barriers 3 and 4, where all eight wavefronts waited for the matrix pipe
        // (3.7 k cycles of every workgroup's critical path), but here, behind the long backward, by whoever arrives first:
        // the wavefronts draw groups of four 16 x 16 tiles from a counter in the LDS until none is left -- the
        // wavefronts that finish P5 early (shorter windows, the first-dispatched half) take them while the others still
        // run, and the last to arrive finds the counter exhausted.  A tile's value does not depend on who forms it.
        // Samples are the K dimension: NSB / 4 k-steps of 4 samples.  Group (quadrant, ta): lane (q, r) reads channels
        // 4r .. 4r+3 of sample 4*step + q from both operands as 16-B pieces; element ta of the A piece and element tb of
        // the B piece feed tile tb, which holds rows 4m + ta, columns 4n + tb of the 64 x 64 quadrant -- the accumulators
        // of one i are four consecutive columns: 16-B stores, 256 B contiguous per 16 lanes.
        constexpr int NQ = D / 64;                       // quadrants per side
        constexpr int NGRP = NQ * NQ * 4;
        int* tk = (int*)(sS + 3);
        float* kp = a.Kp + (size_t)blockIdx.x * D * D;
        const bool first = g == (int)blockIdx.x;         // later passes of this workgroup add to its partial
        for (;;) {
          int t = 0;
          if (lane == 0) t = atomicAdd(tk, 1);
          t = __builtin_amdgcn_readfirstlane(t);
          if (t >= NGRP) break;
          const int quad = t >> 2, ta = t & 3;
          const int M0 = (quad / NQ) * 64, N0 = (quad % NQ) * 64;
          f32x4 acc[4];
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) acc[tb] = (f32x4)(0.0f);
#pragma unroll
          for (int step = 0; step < NSB / 4; ++step) {
            const float va = sL[(4 * step + q) * LSTR + M0 + 4 * r + ta];
            const f32x4 vb = *(const f32x4*)(sA + (4 * step + q) * LSTR + N0 + 4 * r);
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) acc[tb] = TLSAN_MFMA(va, vb[tb], acc[tb]);
          }
          // acc[tb][i] = C[M0 + 4 (4q + i) + ta][N0 + 4 r + tb]
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            f32x4 v;
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) v[tb] = acc[tb][i];
            float* dst = kp + (size_t)(M0 + 4 * (4 * q + i) + ta) * D + N0 + 4 * r;
            if (!first) v += *(const f32x4*)dst;
            st4_out(dst, v);
          }
        }
       }
      }
      TLSAN_STAMP(10);
      __syncthreads();
      if constexpr (G::SPLIT) reduce_part<G, false, 1>(sT, prec, G::P_F1W2, G::P_F1B2, tid);
      else reduce_staged<G, false>(sT, prec, G::P_F1W1, G::P_F1B1, G::P_F1W2, G::P_F1B2, tid);
      if (tid < 3) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += sS[w * 4 + tid];
        prec[G::P_GAMMA + tid] = s;
      }
      TLSAN_STAMP(11);
#if TLSAN_STAMPS
      if (a.stamps != nullptr && lane == 0) sStamp[29] = __builtin_amdgcn_s_memrealtime();
#endif
#if TLSAN_STAMPS
      if (a.stamps != nullptr && lane < 32) a.stamps[((size_t)blockIdx.x * NW + wave) * 32 + lane] = sStamp[lane];
#endif
    }
    // Next pass: sA is rewritten in P1 (last read in P4), sB in P2 (last read at the top of
    // P5), sH rows are wave-private, sT/sS are rewritten in P3/P5 (last read right above; the
    // P1->P2 barrier separates) -> no extra barrier.
  }
}
