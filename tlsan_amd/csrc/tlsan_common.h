// tlsan_common.h -- shared device helpers for the gfx950 TLSAN kernels.
//
// Register layout used by every attention kernel ("C-layout"): a wavefront (64 lanes) is
// viewed as 4 quarters q = lane>>4 of 16 columns r = lane&15.  A column is one
// (sample, 16*NB-channel group) pair; the lane owns channels  col*CW + 16*kb + 4*q + {0..3}
// of that sample as one float4 per 16-block kb.  This is exactly the C/D layout of
// v_mfma_f32_16x16x4_f32 (row = 4*q + reg, col = lane&15) with the channel on the MFMA row
// and the (sample, head) pair on the MFMA column, so:
//   * the per-head maps  m = W^T x  are MFMAs whose B operand is the lane's own float4
//     (k-step s uses k = 4q+s; the weight fragment is pre-permuted to match), and whose
//     result lands back in the same layout -- map1 -> relu -> map2 chain with no data
//     movement;
//   * the per-channel softmax over sequence positions is purely in-lane (positions are
//     different registers of the same lane): no shuffles, no LDS;
//   * a float4 is 16 contiguous bytes of an embedding row, so gathers are 16-B loads and
//     one wave-instruction covers whole 256/512-B rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tlsan.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef short s16x4 __attribute__((ext_vector_type(4)));

#define TLSAN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ---- arithmetic of the matrix products (tlsan_params.matrix_dtype) ---------------------------------------
// MM = TLSAN_MATRIX_F32: v_mfma_f32_16x16x4_f32, exact fp32 (the reference's precision); an operand of one
// 16-deep contraction is the lane's f32x4 (k-step s = element s), i.e. four chained MFMAs.
// MM = TLSAN_MATRIX_BF16: the same contraction as ONE v_mfma_f32_16x16x16_bf16 (lane (q, .) supplies
// k = 4q .. 4q+3 -- exactly the four channels it owns): both operands rounded to bfloat16 (nearest even),
// products and sums in fp32.  1/8 of the matrix-pipe time and a quarter of the dependent chain; the price
// is 8 significant bits per operand (BASELINE.json configs[2] names bf16; a labelled build extension).
template <int MM> struct MMT { typedef f32x4 opd; };
template <> struct MMT<TLSAN_MATRIX_BF16> { typedef s16x4 opd; };

template <int MM>
__device__ __forceinline__ typename MMT<MM>::opd mm_pack(f32x4 v) {
  if constexpr (MM == TLSAN_MATRIX_F32) {
    return v;
  } else {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};       // (one v_cvt_pk_bf16_f32 per pair)
    uint2 u;
    u.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));
    u.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
    return __builtin_bit_cast(s16x4, u);
  }
}

// acc += A (16 x 16-deep) . B  for one 16-channel contraction whose operands are held as above
template <int MM>
__device__ __forceinline__ f32x4 mm_mma(typename MMT<MM>::opd a, typename MMT<MM>::opd b, f32x4 acc) {
  if constexpr (MM == TLSAN_MATRIX_F32) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = TLSAN_MFMA(a[s], b[s], acc);
    return acc;
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
  }
}
// acc += A0 . B0 + A1 . B1 for two 16-deep contractions that share the accumulator: bf16 operands as ONE gfx950
// v_mfma_f32_16x16x32_bf16 (round 6) -- lane (q, .) supplies k-slots 8q .. 8q+7, which are simply the lane's four values of
// the first contraction followed by its four of the second (the k order inside a contraction is free as long as both
// operands agree), so the fragments of the two 16-deep products are concatenated as they stand: half the matrix
// instructions and half the dependent chain of a d = 256 map (two 16-channel blocks per column) and of the bridge GEMMs.
template <int MM>
__device__ __forceinline__ f32x4 mm_mma2(typename MMT<MM>::opd a0, typename MMT<MM>::opd a1, typename MMT<MM>::opd b0,
                                         typename MMT<MM>::opd b1, f32x4 acc) {
  if constexpr (MM == TLSAN_MATRIX_F32) {
    return mm_mma<MM>(a1, b1, mm_mma<MM>(a0, b0, acc));
  } else {
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    const s16x8 A = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7), B = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc, 0, 0, 0);
  }
}
#define TLSAN_NEG (-1e30f)  // VERY_NEGATIVE_NUMBER, reference TLSAN/model.py:10-11
// The attention scores m2 = relu(x W1 + b1) W2 + b2 (model.py:380-382) are only ever used as exponents of the softmax over
// positions (model.py:386).  The kernels keep them in units of log 2: the FORWARD fragments of W2 and b2 are multiplied by
// log2(e) where they are loaded (load_frag_T / load_bias with scl; the fragment-order LDS tables hold them scaled), every
// exponential of a score difference is then one v_exp_f32 (exp2s) instead of v_mul + v_exp, and the mask constant and
// the running maxima live in the same units.  The backward never differentiates through the scaled copy: dm2 is formed
// from the softmax weights, and dm1 = dm2 W2^T, dW2 = m1^T dm2 use the unscaled W2.
// (bf16 matrix products, ADVICE r5: the forward fragment is bf16(W2 * log2 e) while the backward's is bf16(W2), so the
//  two see weights that differ by up to one bf16 ulp per entry -- the forward's effective W2 is bf16(W2 log2 e) / log2 e.
//  Known and accepted: it is inside the rounding the bf16 extension already makes (its tests hold gradients to 15 % in L2
//  against the fp64 oracle); scaling the fp32 accumulator instead would put a v_mul back behind every score.)
#define TLSAN_LOG2E 1.44269504088896340736f
__device__ __forceinline__ float exp2s(float x) { return __builtin_amdgcn_exp2f(x); }

#define TLSAN_LS_MAX 10  // long-term windows up to this size stay in registers (reference default Ls = 10)
#define TLSAN_LS_CAP 96  // larger windows (up to the reference's max_length = 90) are streamed

// ---- table storage: fp32, or bf16 (upper half of the fp32 pattern) ---------------------------
// 4 consecutive elements at element index idx of a table.  The storage type is a COMPILE-TIME
// parameter of the kernels (a run-time branch around the loads makes the compiler serialise the
// gathers: +7 us on the fused kernel).
template <int DT>
__device__ __forceinline__ f32x4 tbl_ld4(const float* __restrict__ base, size_t idx) {
  if constexpr (DT == TLSAN_TABLE_F32) return *(const f32x4*)(base + idx);
  const uint2 v = *(const uint2*)((const uint16_t*)base + idx);
  f32x4 r;
  r[0] = __uint_as_float(v.x << 16);
  r[1] = __uint_as_float(v.x & 0xffff0000u);
  r[2] = __uint_as_float(v.y << 16);
  r[3] = __uint_as_float(v.y & 0xffff0000u);
  return r;
}
// the same in two steps -- the load as it comes from memory, the widening later -- for loads that are issued ahead of
// their use: arithmetic right behind the load would make the wave wait for it there
template <int DT> struct TblRaw { using type = f32x4; };
template <> struct TblRaw<TLSAN_TABLE_BF16> { using type = uint2; };
template <int DT>
__device__ __forceinline__ typename TblRaw<DT>::type tbl_ld4_raw(const float* __restrict__ base, size_t idx) {
  if constexpr (DT == TLSAN_TABLE_F32) return *(const f32x4*)(base + idx);
  else return *(const uint2*)((const uint16_t*)base + idx);
}
template <int DT>
__device__ __forceinline__ f32x4 tbl_cvt(typename TblRaw<DT>::type v) {
  if constexpr (DT == TLSAN_TABLE_F32) {
    return v;
  } else {
    f32x4 r;
    r[0] = __uint_as_float(v.x << 16);
    r[1] = __uint_as_float(v.x & 0xffff0000u);
    r[2] = __uint_as_float(v.y << 16);
    r[3] = __uint_as_float(v.y & 0xffff0000u);
    return r;
  }
}
// 16-byte store of one of the kernels' bulk outputs (per-use gradient rows, dK partials, row sums, table rows).
// (Round 3 measured these as write-through stores, `global_store_dwordx4 ... sc1`: no gain, profiles/r03_ab_summary.md.)
__device__ __forceinline__ void st4_out(float* p, f32x4 v) { *(f32x4*)p = v; }

// 32 well-mixed bits from (element index, stream): the random bits of the stochastic rounding
__device__ __forceinline__ uint32_t tbl_hash(uint32_t x, uint32_t stream) {
  x ^= stream * 0x9e3779b9u;
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// store 4 elements; bf16: stochastic rounding -- add 16 random bits below the kept half, truncate:
// E[stored] = value, so updates smaller than one bf16 ulp still move the parameter on average.
// `stream` = a per-step, per-table salt (deterministic).  w receives the values actually stored.
template <int DT>
__device__ __forceinline__ void tbl_st4(float* __restrict__ base, size_t idx, f32x4& w, uint32_t stream) {
  if constexpr (DT == TLSAN_TABLE_F32) {
    st4_out(base + idx, w);   // (read next by another kernel: written through, not left dirty in the L2)
    return;
  }
  const uint32_t h0 = tbl_hash((uint32_t)idx, stream), h1 = tbl_hash((uint32_t)idx + 0x68bc21ebu, stream ^ 0x2545f491u);
  uint32_t b[4];
  b[0] = (__float_as_uint(w[0]) + (h0 & 0xffffu)) & 0xffff0000u;
  b[1] = (__float_as_uint(w[1]) + (h0 >> 16)) & 0xffff0000u;
  b[2] = (__float_as_uint(w[2]) + (h1 & 0xffffu)) & 0xffff0000u;
  b[3] = (__float_as_uint(w[3]) + (h1 >> 16)) & 0xffff0000u;
  uint2 v;
  v.x = (b[0] >> 16) | b[1];
  v.y = (b[2] >> 16) | b[3];
  *(uint2*)((uint16_t*)base + idx) = v;
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = __uint_as_float(b[i]);
}

// a zero the optimiser cannot see through (keeps LDS weight loads at their use; see Geo::AT_USE)
__device__ __forceinline__ int opaque_zero(int x) {
  int z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z) : "v"(x));
  return z;
}

// NW_: wavefronts per workgroup (0 = the default of the width).  d = 128 also runs as 4-wavefront workgroups of 8 samples
// (Geo<128, 16, 4>), two of which fit a CU.  Measured in round 4 (profiles/r04_nw4_ab.md): at 4096 sequences -- two such
// workgroups per CU instead of one of eight wavefronts -- the kernel is 15 % SLOWER (43.3 vs 37.7 us: the bridge GEMMs'
// tiles are half empty, and starting one of a CU's two workgroups late never shortens the launch: a wavefront is bound
// by its own dependent chain, not by its SIMD partner being in the same phase); it pays for small batches only, where
// it puts a wavefront alone on a SIMD (B = 256: 45.6 -> 42.2 us/step, 1024: 53.7 -> 52.4), and is used there.
template <int D_, int DH_, int NW_ = 0>
struct Geo {
  static constexpr int D = D_;
  static constexpr int DH = DH_;                // channels per head (d / num_heads)
  static constexpr int CW = DH < 16 ? 16 : DH;  // column width: one head, or 2 heads when DH == 8
  static constexpr int NB = CW / 16;            // 16-channel blocks per column
  static constexpr int CPS = D / CW;            // columns per sample
  static constexpr int SPW = 16 / CPS;          // samples per wavefront
  static constexpr int NW = NW_ != 0 ? NW_ : (D_ == 64 ? 4 : 8);   // wavefronts per workgroup
  static constexpr int NSB = NW * SPW;          // samples per workgroup pass
  static constexpr int RT = (NSB + 15) / 16;    // 16-row tiles of the bridge GEMM (NSB = 8: one tile, half of its rows unused)
  static constexpr int NT = D / 16;             // 16-col tiles of the bridge GEMM
  static constexpr int TPW = RT * NT / NW;      // bridge tiles per wavefront
  static constexpr int LSTR = D + 4;            // LDS row stride of [sample][channel] buffers
  // dK = long^T . dbridge (gradient of the bridge's kernel, model.py:347): for D <= 128 every workgroup forms the
  // product over its own 16 samples while both operands sit in the LDS (behind the long backward, drawn tile group by
  // tile group by whichever wavefront gets there first: k_fwd_bwd, end of P5), and leaves one
  // D x D partial per workgroup for k_dense_finalize -- no k_dk_partial launch, no [B, D] round trip through HBM.
  // (D = 256: the partials would be 256 KB per workgroup; the separate kernel stays.)
  // Costs and gains (round 2): the k_dk_partial launch (7.2 us + a gap) goes away, k_fwd_bwd grows by ~1.3 us (six
  // spilled registers at D = 128), k_dense_finalize reads 256 partials instead of 64.  With the next batch's index
  // built ONE step ahead this bought nothing (72.1-73.6 vs 71.0-71.8 us): the step was then bounded by a second path
  // of the same length -- k_fwd_bwd -> index build on the side stream (it cannot run beside that kernel) -> host
  // wake-up -> launch of the next step -- so a shorter tail only made the GPU wait for the host.  With the index built
  // TWO steps ahead (Model.train_async(after_next=)) that path is gone and the main stream's chain is the step:
  // 70.0 us without the fusion, 66.7 us with it (profiles/r02_ab/r02_ab_fuse2.txt).
  static constexpr bool FUSE_DK = D_ <= 128;
  static constexpr int TSTR = 20;                   // LDS row stride of the 16x16 transpose tiles
  // per-wave LDS scratch (floats): 4*NB transpose tiles, or the staged gradient accumulators
  static constexpr int TBUF = 4 * NB * 16 * TSTR;   // one buffer: x, dz1, m1, dm2 tiles (double-buffering them bought nothing)
  static constexpr bool KEEP_A = NB == 1;           // long block's softmax weights kept in LDS for the backward
  static constexpr bool USE_SW = true;              // attention weights staged in LDS
  // NB > 1 (d = 256): six 16-register weight fragments cannot stay in registers for a whole phase:
  // the window-in-registers variants re-read them from LDS at every position (behind a value the
  // optimiser cannot hoist): 274 -> 239 us/step at B=4096.  The streamed variants keep them in
  // registers (spilled): re-reading per position measured slower there (707 vs 659 us at Ls=90).
  static constexpr bool AT_USE = NB > 1;
  static constexpr bool AT_USE_T = true;            // (forward fragments as well: d=256, Ls=10: 239 us/step vs 261 without)
  static constexpr int WSCR_T = TBUF;
  // staged gradient accumulators: all 2 NB^2 + 3 NB vectors at once, or (NB > 1, to fit the LDS) in two
  // halves {dW1, db1, dk0} / {dW2, db2}
  static constexpr bool SPLIT = NB > 1;
  static constexpr int WSCR_A = (SPLIT ? (NB * NB + 2 * NB) : (2 * NB * NB + 3 * NB)) * 256;
  static constexpr int WSCR = WSCR_T > WSCR_A ? WSCR_T : WSCR_A;
  // per-workgroup partial record (effective CWxCW layout), see k_dense_finalize
  static constexpr int P_W = CW * CW;
  static constexpr int P_F1W1 = 0;
  static constexpr int P_F1B1 = P_F1W1 + P_W;
  static constexpr int P_F1W2 = P_F1B1 + CW;
  static constexpr int P_F1B2 = P_F1W2 + P_W;
  static constexpr int P_F2W1 = P_F1B2 + CW;
  static constexpr int P_F2B1 = P_F2W1 + P_W;
  static constexpr int P_F2W2 = P_F2B1 + CW;
  static constexpr int P_F2B2 = P_F2W2 + P_W;
  static constexpr int P_K0 = P_F2B2 + CW;
  static constexpr int P_GAMMA = P_K0 + D;
  static constexpr int P_LOSS = P_GAMMA + 1;
  static constexpr int P_SQ = P_LOSS + 1;
  static constexpr int NPB = P_SQ + 2;  // +1 pad: the scalar record is stored as one float4
  static_assert(CPS >= 1 && CPS <= 16 && (16 % CPS) == 0, "unsupported geometry");
  static_assert((RT * NT) % NW == 0, "bridge tiles must divide over the wavefronts");
};

// Effective CW x CW weight: the head's DHxDH matrix, or for DH == 8 the block-diagonal
// diag(W, W) so that two heads share one 16-wide MFMA block.
template <int DH>
__device__ __forceinline__ float weff(const float* __restrict__ W, int k, int j) {
  if constexpr (DH >= 16) {
    return W[k * DH + j];
  } else {
    return ((k / DH) == (j / DH)) ? W[(k % DH) * DH + (j % DH)] : 0.0f;
  }
}

// "T" fragment: A operand of  out[j] = sum_k W[k][j] v[k]   (forward maps):
//   F[jb][kb][s] on lane (q, m=r)  = W_eff[16kb + 4q + s][16jb + r]
template <int DH, int NB, int MM = TLSAN_MATRIX_F32>
__device__ __forceinline__ void load_frag_T(const float* __restrict__ W, int q, int r,
                                            typename MMT<MM>::opd (&F)[NB][NB], float scl = 1.0f) {
#pragma unroll
  for (int jb = 0; jb < NB; ++jb)
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      f32x4 t;
#pragma unroll
      for (int s = 0; s < 4; ++s) t[s] = weff<DH>(W, 16 * kb + 4 * q + s, 16 * jb + r) * scl;
      F[jb][kb] = mm_pack<MM>(t);
    }
}

// "N" fragment: A operand of  out[k] = sum_j W[k][j] v[j]   (backward maps):
//   F[kb][jb][s] on lane (q, m=r)  = W_eff[16kb + r][16jb + 4q + s]
template <int DH, int NB, int MM = TLSAN_MATRIX_F32>
__device__ __forceinline__ void load_frag_N(const float* __restrict__ W, int q, int r,
                                            typename MMT<MM>::opd (&F)[NB][NB]) {
#pragma unroll
  for (int kb = 0; kb < NB; ++kb)
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
      f32x4 t;
#pragma unroll
      for (int s = 0; s < 4; ++s) t[s] = weff<DH>(W, 16 * kb + r, 16 * jb + 4 * q + s);
      F[kb][jb] = mm_pack<MM>(t);
    }
}

// a fragment from a table kept in fragment order in the LDS ([pair = a * NB + b][lane][4]): NB * NB 16-byte reads at
// constant offsets from the lane's base, conflict-free (k_fwd_bwd: PERM)
template <int NB, int MM = TLSAN_MATRIX_F32>
__device__ __forceinline__ void load_frag_P(const float* __restrict__ tab, int lane, typename MMT<MM>::opd (&F)[NB][NB]) {
  const float* base = tab + lane * 4;
#pragma unroll
  for (int x = 0; x < NB; ++x)
#pragma unroll
    for (int y = 0; y < NB; ++y) F[x][y] = mm_pack<MM>(*(const f32x4*)(base + (x * NB + y) * 256));
}

// bias in C-layout: lane (q, .) reg i of block jb  <->  channel-in-column 16jb + 4q + i
template <int DH, int NB>
__device__ __forceinline__ void load_bias(const float* __restrict__ b, int q, f32x4 (&out)[NB], float scl = 1.0f) {
#pragma unroll
  for (int jb = 0; jb < NB; ++jb)
#pragma unroll
    for (int i = 0; i < 4; ++i) out[jb][i] = b[(16 * jb + 4 * q + i) % DH] * scl;
}

// out = bias + F (x) v  in C-layout (see header comment).  Two accumulators per output block
// (k-steps {0,1} and {2,3}) halve the dependent-MFMA chain: a lone wavefront stalls ~40 cycles
// on every back-to-back dependent v_mfma_f32_16x16x4_f32.
template <int NB, int MM = TLSAN_MATRIX_F32>
__device__ __forceinline__ void map_apply(const typename MMT<MM>::opd (&F)[NB][NB], const f32x4 (&bias)[NB],
                                          const f32x4 (&v)[NB], f32x4 (&out)[NB]) {
  if constexpr (MM == TLSAN_MATRIX_F32) {
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
      f32x4 acc0 = bias[ob], acc1 = (f32x4)(0.0f);
#pragma unroll
      for (int ib = 0; ib < NB; ++ib) {
        acc0 = TLSAN_MFMA(F[ob][ib][0], v[ib][0], acc0);
        acc1 = TLSAN_MFMA(F[ob][ib][2], v[ib][2], acc1);
        acc0 = TLSAN_MFMA(F[ob][ib][1], v[ib][1], acc0);
        acc1 = TLSAN_MFMA(F[ob][ib][3], v[ib][3], acc1);
      }
      out[ob] = acc0 + acc1;
    }
  } else {
    typename MMT<MM>::opd pv[NB];
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) pv[ib] = mm_pack<MM>(v[ib]);
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
      f32x4 acc = bias[ob];
      if constexpr (NB == 2) {
        acc = mm_mma2<MM>(F[ob][0], F[ob][1], pv[0], pv[1], acc);
      } else {
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) acc = mm_mma<MM>(F[ob][ib], pv[ib], acc);
      }
      out[ob] = acc;
    }
  }
}

// out = F (x) v  with no bias: the accumulators start from the instruction's inline constant 0 (no register zeroing)
template <int NB, int MM = TLSAN_MATRIX_F32>
__device__ __forceinline__ void map_apply0(const typename MMT<MM>::opd (&F)[NB][NB], const f32x4 (&v)[NB], f32x4 (&out)[NB]) {
  if constexpr (MM == TLSAN_MATRIX_F32) {
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
      f32x4 acc0 = TLSAN_MFMA(F[ob][0][0], v[0][0], (f32x4)(0.0f));
      f32x4 acc1 = TLSAN_MFMA(F[ob][0][2], v[0][2], (f32x4)(0.0f));
      acc0 = TLSAN_MFMA(F[ob][0][1], v[0][1], acc0);
      acc1 = TLSAN_MFMA(F[ob][0][3], v[0][3], acc1);
#pragma unroll
      for (int ib = 1; ib < NB; ++ib) {
        acc0 = TLSAN_MFMA(F[ob][ib][0], v[ib][0], acc0);
        acc1 = TLSAN_MFMA(F[ob][ib][2], v[ib][2], acc1);
        acc0 = TLSAN_MFMA(F[ob][ib][1], v[ib][1], acc0);
        acc1 = TLSAN_MFMA(F[ob][ib][3], v[ib][3], acc1);
      }
      out[ob] = acc0 + acc1;
    }
  } else {
    typename MMT<MM>::opd pv[NB];
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) pv[ib] = mm_pack<MM>(v[ib]);
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
      f32x4 acc;
      if constexpr (NB == 2) {
        acc = mm_mma2<MM>(F[ob][0], F[ob][1], pv[0], pv[1], (f32x4)(0.0f));
      } else {
        acc = mm_mma<MM>(F[ob][0], pv[0], (f32x4)(0.0f));
#pragma unroll
        for (int ib = 1; ib < NB; ++ib) acc = mm_mma<MM>(F[ob][ib], pv[ib], acc);
      }
      out[ob] = acc;
    }
  }
}

// ---- cross-lane helpers without the LDS ------------------------------------------------------------------
// __shfl / __shfl_xor compile to ds_bpermute_b32: an LDS round trip each, and hipcc waits lgkmcnt(0) after
// every one that sits in its own basic block -- the ten dependent 5-step sums at the end of the long backward
// alone cost 7.5 k cycles per wavefront (scripts/stamps.py).  Everything below stays in the vector ALU:
// DPP row operations inside a 16-lane row, v_permlane16/32_swap across the four rows, v_readlane for
// wave-uniform picks.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
#define TLSAN_DPP_XOR1 0xB1   // quad_perm:[1,0,3,2]
#define TLSAN_DPP_XOR2 0x4E   // quad_perm:[2,3,0,1]
#define TLSAN_DPP_HMIRROR 0x141  // row_half_mirror: i <-> 7 - i   (after the quad steps: the other quad's sum)
#define TLSAN_DPP_MIRROR 0x140   // row_mirror:      i <-> 15 - i  (after the half step: the other half's sum)
#define TLSAN_DPP_ROR(n) (0x120 + (n))  // row_ror:n: lane i reads lane (i + n) % 16 of its row

// sum over N consecutive lanes (aligned group of N = 2, 4, 8 or 16 inside a row); every lane gets the sum
template <int N>
__device__ __forceinline__ float lanes_sum(float v) {
  if constexpr (N >= 2) v += dpp_f32<TLSAN_DPP_XOR1>(v);
  if constexpr (N >= 4) v += dpp_f32<TLSAN_DPP_XOR2>(v);
  if constexpr (N >= 8) v += dpp_f32<TLSAN_DPP_HMIRROR>(v);
  if constexpr (N >= 16) v += dpp_f32<TLSAN_DPP_MIRROR>(v);
  return v;
}
// sum over the lanes {i, i + S, i + 2S, ...} of a row (S = 1 .. 16 a power of two); every lane gets the sum
template <int S>
__device__ __forceinline__ float stride_sum(float v) {
  if constexpr (S <= 8) v += dpp_f32<TLSAN_DPP_ROR(8)>(v);
  if constexpr (S <= 4) v += dpp_f32<TLSAN_DPP_ROR(4)>(v);
  if constexpr (S <= 2) v += dpp_f32<TLSAN_DPP_ROR(2)>(v);
  if constexpr (S <= 1) v += dpp_f32<TLSAN_DPP_ROR(1)>(v);
  return v;
}
// sum over the four rows of the wavefront (lane bits 4 and 5); every lane gets the sum.
// v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second,
// v_permlane32_swap the upper half of the first with the lower half of the second: fed two copies of v they
// leave (r0, r0, r2, r2) / (r1, r1, r3, r3), then (lo, lo) / (hi, hi).  (Inline asm: the builtin folds the
// two results of identical operands into one on ROCm 7.2; the s_nop covers the VALU-write -> permlane hazard.)
__device__ __forceinline__ float rows_sum(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  a += b;
  b = a;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float wave_sum(float v) { return rows_sum(lanes_sum<16>(v)); }

// max over the wavefront of a value that is uniform over the lanes of each sample (lane s * CPS of row 0
// speaks for sample s): SPW readlanes instead of six dependent cross-lane steps
template <int CPS>
__device__ __forceinline__ int wave_max_samples(int v) {
  int m = __builtin_amdgcn_readlane(v, 0);
#pragma unroll
  for (int s = 1; s < 16 / CPS; ++s) m = max(m, __builtin_amdgcn_readlane(v, s * CPS));
  return m;
}

// sum over the lanes that own one sample: quarters (lane bits 4,5) and the CPS columns
template <int CPS>
__device__ __forceinline__ float sample_sum(float v) {
  return rows_sum(lanes_sum<CPS>(v));
}

// the value lane (row, s * CPS + c) holds, for the sample s of the calling lane: `row` and `c` wave-uniform
// (one v_readlane per sample of the wavefront and a select, instead of a ds_bpermute)
template <int CPS>
__device__ __forceinline__ int sample_pick(int v, int row, int c, int s_loc) {
  const int base = __builtin_amdgcn_readfirstlane(row * 16 + c);
  int out = __builtin_amdgcn_readlane(v, base);
#pragma unroll
  for (int s = 1; s < 16 / CPS; ++s) {
    const int x = __builtin_amdgcn_readlane(v, base + s * CPS);
    out = (s_loc == s) ? x : out;
  }
  return out;
}
template <int CPS>
__device__ __forceinline__ float sample_pick(float v, int row, int c, int s_loc) {
  return __builtin_bit_cast(float, sample_pick<CPS>(__builtin_bit_cast(int, v), row, c, s_loc));
}

// 1 / x as one v_rcp_f32 (1 ulp) instead of the eleven-instruction IEEE division sequence; used on softmax
// normalisers (sums of exponentials of non-positive numbers, >= 1) and on 1 + e^-|x| (in (1, 2])
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// clip_by_global_norm's coefficient clip / max(norm, clip) (model.py:201), written so that a non-finite norm is NOT
// swallowed: fmaxf(NaN, clip) is clip, i.e. a diverged step would update with coefficient 1 and look healthy.  TF 1.8
// forms scale = clip * min(1 / norm, 1 / clip) with Eigen's (y < x ? y : x), which hands a NaN norm on to every clipped
// gradient; here NaN -> NaN (the table scale and every parameter of the step follow), +Inf -> 0.
__device__ __forceinline__ float clip_coef(float norm, float clip) { return norm <= clip ? 1.0f : clip / norm; }

__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
}

// make LDS traffic of one wavefront visible to its own lanes (DS ops of a wave are
// executed in order; this only stops the compiler from reordering across it)
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------
// Exact (order-independent) accumulation of fp32 values on a 2^-40 grid held in a double:
//   t = (v + BIG) - BIG rounds v to a multiple of 2^-40 (BIG = 1.5 * 2^12, ulp 2^-40);
//   sums of such t are exact in a double while |sum| < 2^12, so the result does not depend
//   on the order in which contributions are added -> bitwise reproducible scatter-add
//   without sorting the contribution lists.
#define TLSAN_EXACT_BIG 6144.0
__device__ __forceinline__ double exact_term(float v) {
  double t = (double)v + TLSAN_EXACT_BIG;
  return t - TLSAN_EXACT_BIG;
}

#define TLSAN_SN_CAP 96  // longest session the training kernel keeps positions for (reference: <= 90)

struct FwdArgs {
  tlsan_params p;
  tlsan_batch b;
  tlsan_dense_layout lay;
  int32_t Ls, di, dc;
  int32_t WU;       // floats per row of Gu: roundup4(di + Ls)
  int32_t ngroups;  // ceil(B / NSB)
  float inv_B;
  float* logits_i;
  float* logits_j;
  float* u_t;
  // evaluation only, optional: the two attention-weight tensors the reference keeps on the model (model.py:122,
  // 386-394: `soft` of feature_wise_attention, heads split along the batch axis)
  float* att0;      // [H * B, Ls, d / H]        long-term block:  row h * B + b
  float* att1;      // [H * B, 1 + Sn, d / H]    short-term block (position 0 = the bridge)
  // training only.  Per-use gradient rows are written at DESTINATION-SORTED positions: the
  // kernel draws the position of each use from the fill cursor of its destination row
  // (cursor = exclusive scan of the per-row use counts), so k_apply_* read contiguous segments.
  float* Gi;        // [n_item_uses, D]  [item half | cate half] rows, grouped by item id
  float* Gb;        // [n_item_uses]     item_b gradients (same positions; 0 for non-candidate uses)
  float* Gu;        // [B, WU]           [user_emb | usert_emb | pad] rows, grouped by user id
  float* Gc;        // [B, dc]           u_cate rows, grouped by category
  int32_t* cur_item; int32_t* cur_user; int32_t* cur_uc;
  const int32_t* perm;    // optional: perm[16 g + j] = sample j of workgroup pass g (>= B: none); BalArgs, tlsan_update.h
  int32_t uc_by_sample;   // != 0: Gc rows are written in SAMPLE order (row b = sample b) and the index holds the
                          // samples of every category (uc_list, k_uc_fill): no cursor is drawn for the u_cate use
  int32_t fuse_dk;  // != 0: this launch forms the dK partials itself (Geo::FUSE_DK builds; chosen per launch, tlsan_api.hip)
  int32_t cseg;     // != 0 (many categories): the category half of every item use's gradient row is written into the
                    // category's own segment of Gc (position drawn from cur_uc), not beside the item half in Gi
  float* gLong;     // [B, D]   long-term summaries (A operand of dK)   -- written only when the dK product is NOT fused
  float* gDB;       // [B, D]   d loss / d bridge     (B operand of dK)
  float* gStat;     // [B, 2, D] streamed windows at d = 256: per-channel max and 1 / sum of the long block's scores (P1 -> P5)
  float* Kp;        // fused dK (Geo::FUSE_DK): [gridDim.x][D*D] partial products long^T . dbridge, one per workgroup
  float* partials;  // [ngroups, NPB]
  unsigned long long* stamps;  // diagnostic only (NULL in production): [gridDim.x*8 waves][16]
  // dropout on the inputs of the attention maps (model.py:428-431), training with config['dropout'] > 0
  uint32_t drop_seed, drop_thr;   // kept iff hash < drop_thr (keep_prob * 2^32); drop_thr == 0: no dropout
  float drop_inv;                 // 1 / keep_prob
  uint32_t drop_sample0;          // index of this batch's first sample in the pattern (a rank's share of a global batch)
  uint32_t* started;              // optional host-visible word: workgroup 0 stores started_val there when the kernel begins
  uint32_t started_val;
};

// Keep / drop pattern of tf.nn.dropout as the scale the element is multiplied with (0 or 1/keep_prob):
// a counter-based hash of (seed, sample, block, position, map, channel) -- the forward and the two
// recomputations of the backward see the same pattern without storing it (oracle: dropout_scale).
struct DropCtx {
  uint32_t seed, thr;
  float inv;
  uint32_t sbase;  // 2 * global sample index
};
__device__ __forceinline__ f32x4 drop_scale4(const DropCtx& dc, int net, int pos, int which, int chan) {
  const uint32_t e0 = ((((dc.sbase + (uint32_t)net) * 128u + (uint32_t)pos) * 2u + (uint32_t)which) * 256u) + (uint32_t)chan;
  f32x4 k;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t h = (e0 + (uint32_t)i) ^ dc.seed;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    k[i] = h < dc.thr ? dc.inv : 0.0f;
  }
  return k;
}
