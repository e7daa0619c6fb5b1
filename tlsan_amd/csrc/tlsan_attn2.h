// tlsan_attn2.h -- k_fwd_bwd2: the fused training kernel in the "one sample per wavefront" layout.
//
// Same mathematics, phases and outputs as k_fwd_bwd (tlsan_attn.h; reference TLSAN/model.py:84-137
// forward, :164-172 loss, and what tf.gradients (:198) differentiates), for the shapes where one
// attention column is one 16-channel block (d = 64 or 128, 8 heads) and the long window fits in
// registers (Ls <= TLSAN_LS_MAX).  What changes is the assignment of work to lanes:
//
//   k_fwd_bwd :  16 columns of a wavefront = 2 samples x 8 heads; a lane walks ALL positions of
//                its sample;  8 wavefronts / workgroup, 2 per SIMD at B = 4096 (one workgroup per CU).
//   k_fwd_bwd2:  16 columns = PP position classes x CPS heads of ONE sample (PP = 2 at d=128,
//                4 at d=64); a lane walks the positions p = par (mod PP) only; 16 wavefronts per
//                workgroup under a 128-VGPR budget -> 4 per SIMD.
//
// The step is bound by the latency of a wavefront's dependent MFMA / LDS chain, not by any
// pipe: halving the chain and doubling the resident wavefronts is what this layout buys.  The
// softmax over positions (model.py:386) stays in-lane over the lane's own positions plus ONE
// cross-lane combine over the PP classes (lanes r and r ^ CPS ...: same quarter row).
// The dW products sum over all 16 columns of the tile = over (position class, head), which is
// exactly the sum the gradient needs.  Per-workgroup partial records, gradient-row buffers and
// every other interface are those of k_fwd_bwd, so the rest of the step is unchanged.
#pragma once
#include "tlsan_attn.h"

template <int D_, int DH_>
struct Geo2 : Geo<D_, DH_> {
  using Base = Geo<D_, DH_>;
  static_assert(Base::NB == 1, "k_fwd_bwd2 handles 16-channel columns only");
  static constexpr int CPS = Base::CPS;   // columns (16-channel blocks) per sample: 8 (d=128), 4 (d=64)
  static constexpr int PP = 16 / CPS;     // position classes sharing a wavefront
  static constexpr int NW = 16;           // wavefronts per workgroup = samples per workgroup
  static constexpr int NSB = 16;
  static constexpr int NT = D_ / 16;      // 16-column tiles of the two GEMMs (one per wavefront < NT)
  static constexpr int NPL = (TLSAN_LS_MAX + PP - 1) / PP;  // long positions per lane
  static constexpr int WSCR = 1280;       // per-wave LDS scratch: 4 transpose tiles [16][20] / 5 staged vectors
};

// sum / max over the PP position classes of a sample (lane bits CPS .. 8 of the column index)
template <int CPS>
__device__ __forceinline__ float pp_sum(float v) {
#pragma unroll
  for (int o = CPS; o < 16; o <<= 1) v += __shfl_xor(v, o);
  return v;
}
template <int CPS>
__device__ __forceinline__ float pp_max(float v) {
#pragma unroll
  for (int o = CPS; o < 16; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// 16-B access at (uniform base pointer) + (32-bit per-lane float index): the compiler keeps the base
// in SGPRs and the index in ONE VGPR (global_load ... v_off, s[base]) instead of a 64-bit address
// pair per lane that it would precompute, hoist and spill.  The launcher checks that every table
// and gradient buffer is smaller than 4 GiB.
__device__ __forceinline__ f32x4 ld4(const float* __restrict__ base, uint32_t idx) {
  return *(const f32x4*)((const char*)base + (size_t)(idx * 4u));
}
__device__ __forceinline__ void st4(float* __restrict__ base, uint32_t idx, f32x4 v) {
  *(f32x4*)((char*)base + (size_t)(idx * 4u)) = v;
}
template <class T>
__device__ __forceinline__ T ldg(const T* __restrict__ base, uint32_t idx) {
  return *(const T*)((const char*)base + (size_t)(idx * (uint32_t)sizeof(T)));
}
template <class T>
__device__ __forceinline__ void stg(T* __restrict__ base, uint32_t idx, T v) {
  *(T*)((char*)base + (size_t)(idx * (uint32_t)sizeof(T))) = v;
}
__device__ __forceinline__ int draw(int32_t* __restrict__ cur, uint32_t idx) {  // next free position of a destination row
  return atomicAdd((int32_t*)((char*)cur + (size_t)(idx * 4u)), 1);
}
// channels [c, c+4) of [item_emb[it] || cate_emb[ct]] (model.py:84-86,105-107,111-113); the lane is
// statically an item-half or a cate-half lane, so each half issues its own base+offset load
__device__ __forceinline__ f32x4 gather2(const FwdArgs& a, int it, int ct, int c) {
  if (c < a.di) return ld4(a.p.item_emb, (uint32_t)it * (uint32_t)a.p.ld_item + (uint32_t)c);
  return ld4(a.p.cate_emb, (uint32_t)ct * (uint32_t)a.dc + (uint32_t)(c - a.di));
}

// (opaque_zero, tlsan_common.h: weight fragments live in LDS; adding it to their address at every use keeps
// the loads AT the use instead of hoisted out of the position loops, where six fragments would pin 24 VGPRs)
// the two per-head maps of feature_wise_attention (model.py:380-382) with fragments fetched from LDS
template <int DH>
__device__ __forceinline__ void fwd_maps(const float* W1, const float* b1, const float* W2, const float* b2, int q, int r,
                                         const f32x4& x, f32x4& z1, f32x4& m2) {
  float F[1][1][4];
  f32x4 b[1], in[1], out[1];
  load_frag_T<DH, 1>(W1, q, r, F);
  load_bias<DH, 1>(b1, q, b);
  in[0] = x;
  map_apply<1>(F, b, in, out);
  z1 = out[0];
#pragma unroll
  for (int i = 0; i < 4; ++i) in[0][i] = fmaxf(z1[i], 0.0f);
  load_frag_T<DH, 1>(W2, q, r, F);
  load_bias<DH, 1>(b2, q, b);
  map_apply<1>(F, b, in, out);
  m2 = out[0];
}

template <int D, int DH>
static size_t fwd2_smem_bytes() {
  using G = Geo2<D, DH>;
  return sizeof(float) * (2 * G::NSB * G::LSTR + G::NW * 4 + G::NSB * 2 * TLSAN_LS_MAX + 2 * (2 * DH * DH + 2 * DH) +
                          G::NSB * (TLSAN_LS_MAX + TLSAN_SN_CAP + 4) + G::NW * G::WSCR + G::NW * 64 * 8 + G::NSB * 2 * TLSAN_LS_MAX);
}

template <int D, int DH>
__global__ __launch_bounds__(1024) void k_fwd_bwd2(FwdArgs a) {
  using G = Geo2<D, DH>;
  constexpr int CPS = G::CPS, PP = G::PP, NW = G::NW, NSB = G::NSB, NT = G::NT, NPL = G::NPL;
  constexpr int LS = TLSAN_LS_MAX;
  constexpr int LSTR = G::LSTR, TSTR = G::TSTR;
  constexpr int WB = 2 * DH * DH + 2 * DH;  // floats of one attention block's weights
  constexpr int PSTR = LS + TLSAN_SN_CAP + 4;
  constexpr int P_TGT = LS + TLSAN_SN_CAP, P_USR = P_TGT + 1, P_UC = P_TGT + 2;
  constexpr int NB = 1;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sA = smem;                   // [NSB][LSTR]  long -> dbridge
  float* sB = sA + NSB * LSTR;        // [NSB][LSTR]  bridge -> dlong
  float* sS = sB + NSB * LSTR;        // [NW][4] scalar staging
  float* sH = sS + NW * 4;            // [NSB][2*LS] hist_t and usert*hist_t
  float* sW = sH + NSB * 2 * LS;      // [2][WB] attention weights of both blocks
  int* sP = (int*)(sW + 2 * WB);      // [NSB][PSTR] destination-sorted row of every use
  float* sT = (float*)(sP + NSB * PSTR);  // [NW][WSCR] per-wave transpose scratch / staging
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: per-sample scalars live in SGPRs
  const int srow = wave;              // one sample per wavefront

  const float* dn = a.p.dense;
  const float gamma = dn[a.lay.gamma];
  const float P = a.p.scale ? *a.p.scale : 1.0f;  // tables hold W / P (lazy L2 decay)
  const int Ls = a.Ls, Sn = a.b.Sn, B = a.b.B;
  for (int o = tid; o < 2 * WB; o += NW * 64)
    sW[o] = (o < WB) ? dn[a.lay.f1_W1 + o] : dn[a.lay.f2_W1 + (o - WB)];
  __syncthreads();
  const float *w1W1 = sW, *w1b1 = w1W1 + DH * DH, *w1W2 = w1b1 + DH, *w1b2 = w1W2 + DH * DH;
  const float *w2W1 = sW + WB, *w2b1 = w2W1 + DH * DH, *w2W2 = w2b1 + DH, *w2b2 = w2W2 + DH * DH;

#define TLSAN_STAMP2(k)                                                                     \
  do {                                                                                      \
    if (a.stamps != nullptr && lane == 0)                                                   \
      a.stamps[((size_t)blockIdx.x * NW + wave) * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
  for (int g = blockIdx.x; g < a.ngroups; g += gridDim.x) {
    // Lane geometry is derived INSIDE the pass loop from a lane id the optimiser cannot prove
    // loop-invariant: hoisted out, the dozens of per-lane addresses built from it would be live
    // (and spilled) across the whole body, while the loop normally runs once.
    const int lane = (tid & 63) + opaque_zero(g);
    const int q = lane >> 4, r = lane & 15;
    const int col = r % CPS, par = r / CPS;
    float* T = sT + wave * G::WSCR;
    float* sM = sT + NW * G::WSCR + (wave * 64 + lane) * 8;  // this lane's long-block softmax statistics (max, 1/sum)
    int* sI = (int*)(sT + NW * G::WSCR + NW * 64 * 8) + wave * 2 * LS;  // [LS] item ids, [LS] categories of the long positions
    const int chb = col * 16 + 4 * q;   // first of the lane's 4 channels
    const bool own = par == 0;          // the class that writes per-sample (not per-position) data
    const bool lead = (q == 0) && (col == 0) && own;
    const bool plead = (q == 0) && (col == 0);  // one lane per position class
    TLSAN_STAMP2(0);
    const int bidx = g * NSB + srow;
    const bool vs = bidx < B;          // wave-uniform
    const int bb = vs ? bidx : 0;
    const int uid = __builtin_amdgcn_readfirstlane(a.b.u[bb]);
    const int it_i = __builtin_amdgcn_readfirstlane(a.b.i[bb]);
    const int ucat = __builtin_amdgcn_readfirstlane(a.b.u_cate[bb]);
    const int n_l = vs ? min(__builtin_amdgcn_readfirstlane(a.b.sl[bb]), Ls) : 0;
    const int n_s = vs ? min(__builtin_amdgcn_readfirstlane(a.b.sl_new[bb]), Sn) : 0;
    float loss_acc = 0.0f, sq_acc = 0.0f, dgam = 0.0f;
    // ------------------------------------------------------------------ P1: long block forward
    // lane's positions: p_j = PP*j + par.  Three load stages (ids/scales -> categories -> rows),
    // all on clamped addresses, padding applied by selects afterwards.
    f32x4 long4;
    {
      f32x4 e1[NPL], mx1, iz1;
      int its[NPL], cts[NPL];
      int posv[NPL], pos_t = 0, pos_u = 0, pos_c = 0;
      float sc1[NPL], hts[NPL], uts[NPL];
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const int pc = min(PP * j + par, Ls - 1);
        its[j] = ldg(a.b.hist_i, (uint32_t)(bb * Ls + pc));
        hts[j] = ldg(a.b.hist_t, (uint32_t)(bb * Ls + pc));
        uts[j] = ldg(a.p.usert_emb, (uint32_t)uid * (uint32_t)a.p.ld_usert + (uint32_t)pc);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NPL; ++j) cts[j] = ldg(a.p.item_cate, (uint32_t)its[j]);
      // destination-sorted row of every use: returning atomics issued now, published after P2
#pragma unroll
      for (int j = 0; j < NPL; ++j) posv[j] = (plead && PP * j + par < n_l) ? draw(a.cur_item, (uint32_t)its[j]) : 0;
      if (lead && vs) {
        pos_t = draw(a.cur_item, (uint32_t)it_i);
        pos_u = draw(a.cur_user, (uint32_t)uid);
        pos_c = a.uc_by_sample ? bidx : draw(a.cur_uc, (uint32_t)ucat);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NPL; ++j) e1[j] = gather2(a, its[j], cts[j], chb);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const int p = PP * j + par;
        const bool vp = p < n_l;
        sc1[j] = vp ? (gamma * P * P) * (uts[j] * hts[j]) : 0.0f;  // model.py:100-102,109
        if (plead && p < LS) {
          sH[srow * 2 * LS + p] = vp ? hts[j] : 0.0f;
          sH[srow * 2 * LS + LS + p] = vp ? uts[j] * hts[j] : 0.0f;
          sI[p] = its[j];       // P5 gathers the rows again (their registers go to the short block)
          sI[LS + p] = cts[j];
        }
        e1[j] = vp ? e1[j] : (f32x4)(0.0f);
      }
      TLSAN_STAMP2(12);
      float FT1[1][1][4], FT2[1][1][4];
      f32x4 b1[1], b2[1];
      load_frag_T<DH, 1>(w1W1, q, r, FT1);
      load_frag_T<DH, 1>(w1W2, q, r, FT2);
      load_bias<DH, 1>(w1b1, q, b1);
      load_bias<DH, 1>(w1b2, q, b2);
      // feature_wise_attention forward (model.py:370-394) over the lane's positions, then one
      // combine over the PP classes
      f32x4 av[NPL];
      mx1 = (f32x4)(TLSAN_NEG);
      const int jmax = (n_l + PP - 1) / PP;  // wave-uniform
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        if (j < jmax) {
          f32x4 xv[1], z[1], m2[1];
          xv[0] = e1[j] * sc1[j];
          map_apply<1>(FT1, b1, xv, z);
#pragma unroll
          for (int i = 0; i < 4; ++i) z[0][i] = fmaxf(z[0][i], 0.0f);
          map_apply<1>(FT2, b2, z, m2);
          av[j] = (PP * j + par < n_l) ? m2[0] : (f32x4)(TLSAN_NEG);  // model.py:384
#pragma unroll
          for (int i = 0; i < 4; ++i) mx1[i] = fmaxf(mx1[i], av[j][i]);
        } else {
          av[j] = (f32x4)(TLSAN_NEG);
        }
      }
      TLSAN_STAMP2(13);
      f32x4 Z = (f32x4)(0.0f);
#pragma unroll
      for (int i = 0; i < 4; ++i) mx1[i] = pp_max<CPS>(mx1[i]);
#pragma unroll
      for (int j = 0; j < NPL; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float ev = __expf(av[j][i] - mx1[i]);  // softmax over positions, model.py:386
          av[j][i] = ev;
          Z[i] += ev;
        }
      long4 = (f32x4)(0.0f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float zt = pp_sum<CPS>(Z[i]);
        iz1[i] = zt > 0.0f ? 1.0f / zt : 0.0f;  // samples past the batch: no position
      }
#pragma unroll
      for (int j = 0; j < NPL; ++j) long4 += (av[j] * iz1) * (e1[j] * sc1[j]);  // model.py:387
#pragma unroll
      for (int i = 0; i < 4; ++i) long4[i] = pp_sum<CPS>(long4[i]);
      TLSAN_STAMP2(14);
      *(f32x4*)(sM) = mx1;  // needed again in P5 only
      *(f32x4*)(sM + 4) = iz1;
      // publish the positions drawn above (the atomics returned during the attention maths)
      if (plead) {
#pragma unroll
        for (int j = 0; j < NPL; ++j)
          if (PP * j + par < LS) sP[srow * PSTR + PP * j + par] = posv[j];
      }
      if (lead) {
        sP[srow * PSTR + P_TGT] = pos_t;
        sP[srow * PSTR + P_USR] = pos_u;
        sP[srow * PSTR + P_UC] = pos_c;
      }
    }
    if (own) {
      *(f32x4*)(sA + srow * LSTR + chb) = long4;
      if (vs) st4(a.gLong, (uint32_t)(bidx * D + chb), long4);
    }
    TLSAN_STAMP2(1);
    __syncthreads();
    TLSAN_STAMP2(2);
    // ------------------------------------------------------------------ P2: bridge GEMM
    // bridge[s][j] = sum_k long[s][k] K[k][j] + k0[j]   (tf.layers.dense, model.py:347)
    if (wave < NT) {
      const int ct = wave;
      f32x4 bfr[D / 16];  // B fragments: K^T rows (L2-resident)
      const uint32_t brow = (uint32_t)((16 * wave + r) * D + 4 * q);
#pragma unroll
      for (int kc = 0; kc < D / 16; ++kc) bfr[kc] = ld4(a.p.dense_KT, brow + 16u * kc);
      f32x4 acc0 = (f32x4)(dn[a.lay.k0 + 16 * ct + r]), acc1 = (f32x4)(0.0f);
      const float* Arow = sA + r * LSTR + 4 * q;
#pragma unroll
      for (int kc = 0; kc < D / 16; kc += 2) {
        const f32x4 av0 = *(const f32x4*)(Arow + 16 * kc), av1 = *(const f32x4*)(Arow + 16 * kc + 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc0 = TLSAN_MFMA(av0[s], bfr[kc][s], acc0);
          acc1 = TLSAN_MFMA(av1[s], bfr[kc + 1][s], acc1);
        }
      }
      acc0 += acc1;
#pragma unroll
      for (int i = 0; i < 4; ++i) sB[(4 * q + i) * LSTR + 16 * ct + r] = acc0[i];
    }
    // rows the short block needs that depend only on ids
    f32x4 uemb, iemb;
    {
      if (chb < a.di) uemb = ld4(a.p.user_emb, (uint32_t)uid * (uint32_t)a.p.ld_user + (uint32_t)chb) * P;
      else uemb = ld4(a.p.cate_emb, (uint32_t)ucat * (uint32_t)a.dc + (uint32_t)(chb - a.di)) * P;
      iemb = gather2(a, it_i, __builtin_amdgcn_readfirstlane(ldg(a.p.item_cate, (uint32_t)it_i)), chb) * P;
    }
    const float ib_i = ldg(a.p.item_b, (uint32_t)it_i * (uint32_t)a.p.ld_itemb);
    // session ids (and categories): entry `lane` in (sid0, scat0), entry 64 + lane in (sid1, scat1)
    // (TLSAN_SN_CAP <= 128), broadcast to the lanes that need them by cross-lane reads
    static_assert(TLSAN_SN_CAP <= 128, "two session-id registers per lane");
    int sid0 = 0, scat0 = 0, sid1 = 0, scat1 = 0;
    if (Sn > 0) sid0 = ldg(a.b.hist_i_new, (uint32_t)(bb * Sn + min(lane, Sn - 1)));
    if (n_s > 64) sid1 = ldg(a.b.hist_i_new, (uint32_t)(bb * Sn + min(64 + lane, Sn - 1)));  // (wave-uniform, rare)
    scat0 = ldg(a.p.item_cate, (uint32_t)sid0);
    if (n_s > 64) scat1 = ldg(a.p.item_cate, (uint32_t)sid1);
    if (vs && lane < n_s) sP[srow * PSTR + LS + lane] = draw(a.cur_item, (uint32_t)sid0);
    if (vs && 64 + lane < n_s) sP[srow * PSTR + LS + 64 + lane] = draw(a.cur_item, (uint32_t)sid1);
    TLSAN_STAMP2(3);
    __syncthreads();
    TLSAN_STAMP2(4);
    // ------------------------------------------------------------------ P3: short block
    // positions: 0 = bridge, 1..n_s = session rows (model.py:350); class `par` walks t = par, par+PP, ...
    const int n_pos = n_s + 1;                 // model.py:355: rep_length = sl_new + 1
    const int jmax2 = (n_pos + PP - 1) / PP;   // wave-uniform
    auto fetch_pos = [&](int t) -> f32x4 {  // input row of position t
      const int e = min(max(t - 1, 0), 127);
      int it = __shfl(sid0, e & 63), ct = __shfl(scat0, e & 63);
      if (n_s > 64) {  // wave-uniform
        const int it1 = __shfl(sid1, e & 63), ct1 = __shfl(scat1, e & 63);
        it = e < 64 ? it : it1;
        ct = e < 64 ? ct : ct1;
      }
      const f32x4 v = gather2(a, it, ct, chb) * P;
      return t == 0 ? *(const f32x4*)(sB + srow * LSTR + chb) : (t < n_pos ? v : (f32x4)(0.0f));  // position 0 = bridge
    };
    f32x4 mx, Zs, short4;
    {
      f32x4 mxl = (f32x4)(TLSAN_NEG), Zl = (f32x4)(0.0f), Nl = (f32x4)(0.0f);
      f32x4 xnext = fetch_pos(par);
      for (int j = 0; j < jmax2; ++j) {  // wave-uniform trip count
        const int t = par + PP * j;
        f32x4 xv[1], z[1], m2[1];
        xv[0] = xnext;
        if (j + 1 < jmax2) xnext = fetch_pos(t + PP);
        const int zz = opaque_zero(j);
        fwd_maps<DH>(w2W1 + zz, w2b1 + zz, w2W2 + zz, w2b2 + zz, q, r, xv[0], z[0], m2[0]);
        if (t < n_pos) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float mn = fmaxf(mxl[i], m2[0][i]);
            const float so = __expf(mxl[i] - mn), ev = __expf(m2[0][i] - mn);
            Zl[i] = Zl[i] * so + ev;
            Nl[i] = Nl[i] * so + ev * xv[0][i];
            mxl[i] = mn;
          }
        }
      }
      // combine the PP classes (class 0 always holds position 0, so the maximum is finite)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mx[i] = pp_max<CPS>(mxl[i]);
        const float sc = __expf(mxl[i] - mx[i]);
        const float zt = pp_sum<CPS>(Zl[i] * sc);
        Zs[i] = 1.0f / zt;
        short4[i] = pp_sum<CPS>(Nl[i] * sc) * Zs[i];
      }
    }
    // u_t = short + [user_emb[u] || cate_emb[u_cate]]   (model.py:93-95,135)
    const f32x4 ut4 = short4 + uemb;
    if (a.u_t != nullptr && vs && own) st4(a.u_t, (uint32_t)(bidx * D + chb), ut4);
    const float logit = sample_sum<CPS>(dot4(ut4, iemb)) + ib_i;  // model.py:137 (every class: same value)
    if (lead && vs && a.logits_i != nullptr) a.logits_i[bidx] = logit;
    TLSAN_STAMP2(5);
    float* prec = a.partials + (size_t)g * G::NPB;
    // BCE with logits, mean over the batch (model.py:171)
    const float yv = a.b.y[bb];
    const float en = __expf(-fabsf(logit));
    const float lb = fmaxf(logit, 0.0f) - logit * yv + __logf(1.0f + en);
    const float sg = logit >= 0.0f ? 1.0f / (1.0f + en) : en / (1.0f + en);
    const float dl = vs ? (sg - yv) * a.inv_B : 0.0f;
    const f32x4 dout = iemb * dl;  // d loss / d u_t
    f32x4 dk0[1];
    dk0[0] = (f32x4)(0.0f);
    {
      const int p_t = sP[srow * PSTR + P_TGT], p_u = sP[srow * PSTR + P_USR], p_c = sP[srow * PSTR + P_UC];
      if (lead && vs) {
        stg(a.Gb, (uint32_t)p_t, dl);  // per-use item_b gradient
        loss_acc += lb;
        sq_acc += dl * dl;
      }
      if (own && vs) {
        const f32x4 gi = ut4 * dl;
        // user use: [user_emb half] -> Gu (grouped by user), [u_cate half] -> Gc (grouped by category)
        if (chb < a.di) st4(a.Gu, (uint32_t)p_u * (uint32_t)a.WU + (uint32_t)chb, dout);
        else st4(a.Gc, (uint32_t)p_c * (uint32_t)a.dc + (uint32_t)(chb - a.di), dout);
        st4(a.Gi, (uint32_t)p_t * (uint32_t)D + (uint32_t)chb, gi);  // candidate use
        sq_acc += dot4(dout, dout) + dot4(gi, gi);
      }
    }
    // ---- backward of the short block
    {
      AccSet<1> acc;
      acc.zero();
      f32x4 xn2 = fetch_pos(par);
      const f32x4 sh1[1] = {short4}, do1[1] = {dout};
      for (int j = 0; j < jmax2; ++j) {
        const int t = par + PP * j;
        const bool vt = t < n_pos;
        f32x4 xv[1], z1[1], m2[1], av[1], dx[1];
        xv[0] = xn2;
        if (j + 1 < jmax2) xn2 = fetch_pos(t + PP);
        const int zz = opaque_zero(j);
        fwd_maps<DH>(w2W1 + zz, w2b1 + zz, w2W2 + zz, w2b2 + zz, q, r, xv[0], z1[0], m2[0]);
#pragma unroll
        for (int i = 0; i < 4; ++i) av[0][i] = vt ? __expf(m2[0][i] - mx[i]) * Zs[i] : 0.0f;
        float FN1[1][1][4], FN2[1][1][4];
        load_frag_N<DH, 1>(w2W1 + zz, q, r, FN1);
        load_frag_N<DH, 1>(w2W2 + zz, q, r, FN2);
        bwd_compute<1, TSTR>(FN2, FN1, xv, z1, av, sh1, do1, T, q, r, acc.db1, acc.db2, dx);
        bwd_dw<1, TSTR>(T, q, r, acc.dW1, acc.dW2);
        if (t == 0) {
          *(f32x4*)(sA + srow * LSTR + chb) = dx[0];  // dbridge
          if (vs) {
            st4(a.gDB, (uint32_t)(bidx * D + chb), dx[0]);
            dk0[0] += dx[0];
          }
        } else if (vs && vt) {
          const int pos = sP[srow * PSTR + LS + (t - 1)];
          if (plead) stg(a.Gb, (uint32_t)pos, 0.0f);
          st4(a.Gi, (uint32_t)pos * (uint32_t)D + (uint32_t)chb, dx[0]);
          sq_acc += dot4(dx[0], dx[0]);
        }
      }
      stage_accs<1, CPS, true>(acc, dk0, T, lane);
    }
    // first long row of the backward: in flight across the barrier and P4
    f32x4 enext = gather2(a, sI[min(par, LS - 1)], sI[LS + min(par, LS - 1)], chb);
    TLSAN_STAMP2(6);
    __syncthreads();
    TLSAN_STAMP2(7);
    reduce_staged<G, true>(sT, prec, G::P_F2W1, G::P_F2B1, G::P_F2W2, G::P_F2B2, tid);
    // ---------------------------------------------------------------- P4: dlong GEMM
    // dlong[s][k] = sum_j dbridge[s][j] K[k][j]
    if (wave < NT) {
      const int kt = wave;
      f32x4 bfr[D / 16];  // B fragments: K rows
      const uint32_t brow = (uint32_t)(a.lay.K + (16 * wave + r) * D + 4 * q);
#pragma unroll
      for (int jc = 0; jc < D / 16; ++jc) bfr[jc] = ld4(dn, brow + 16u * jc);
      f32x4 acc0 = (f32x4)(0.0f), acc1 = (f32x4)(0.0f);
      const float* Arow = sA + r * LSTR + 4 * q;
#pragma unroll
      for (int jc = 0; jc < D / 16; jc += 2) {
        const f32x4 av0 = *(const f32x4*)(Arow + 16 * jc), av1 = *(const f32x4*)(Arow + 16 * jc + 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc0 = TLSAN_MFMA(av0[s], bfr[jc][s], acc0);
          acc1 = TLSAN_MFMA(av1[s], bfr[jc + 1][s], acc1);
        }
      }
      acc0 += acc1;
#pragma unroll
      for (int i = 0; i < 4; ++i) sB[(4 * q + i) * LSTR + 16 * kt + r] = acc0[i];
    }
    TLSAN_STAMP2(8);
    __syncthreads();
    TLSAN_STAMP2(9);
    // ---------------------------------------------------------------- P5: long backward
    {
      const f32x4 dl1[1] = {*(const f32x4*)(sB + srow * LSTR + chb)};
      const f32x4 lo1[1] = {long4};
      f32x4 dummy[1];
      const f32x4 mx1 = *(const f32x4*)(sM), iz1 = *(const f32x4*)(sM + 4);
      AccSet<1> acc;
      acc.zero();
      const int jmax = (n_l + PP - 1) / PP;  // wave-uniform
      const int p_u = sP[srow * PSTR + P_USR];
#pragma unroll 1
      for (int j = 0; j < jmax; ++j) {  // rolled on purpose: one position's temporaries at a time
        const int p = PP * j + par;
        const bool vp = p < n_l;
        const f32x4 e = vp ? enext : (f32x4)(0.0f);
        if (j + 1 < jmax) {  // next row while this one is processed
          const int pn = min(p + PP, LS - 1);
          enext = gather2(a, sI[pn], sI[LS + pn], chb);
        }
        const float uth = sH[srow * 2 * LS + LS + min(p, LS - 1)];
        const float scp = vp ? (gamma * P * P) * uth : 0.0f;  // x = e_stored * scp
        const float sce = (gamma * P) * uth;                  // d x / d e_true
        f32x4 xv[1], z1[1], m2[1], av[1], dx[1];
        xv[0] = e * scp;
        const int zz = opaque_zero(p);
        fwd_maps<DH>(w1W1 + zz, w1b1 + zz, w1W2 + zz, w1b2 + zz, q, r, xv[0], z1[0], m2[0]);
#pragma unroll
        for (int i = 0; i < 4; ++i) av[0][i] = vp ? __expf(m2[0][i] - mx1[i]) * iz1[i] : 0.0f;
        float FN1[1][1][4], FN2[1][1][4];
        load_frag_N<DH, 1>(w1W1 + zz, q, r, FN1);
        load_frag_N<DH, 1>(w1W2 + zz, q, r, FN2);
        bwd_compute<1, TSTR>(FN2, FN1, xv, z1, av, lo1, dl1, T, q, r, acc.db1, acc.db2, dx);
        bwd_dw<1, TSTR>(T, q, r, acc.dW1, acc.dW2);
        const float ds = sample_sum<CPS>(dot4(dx[0], e)) * P;  // d loss / d scale[p] (e_true = P * e_stored)
        if (vs && vp) {
          const int pos = sP[srow * PSTR + p];
          if (plead) stg(a.Gb, (uint32_t)pos, 0.0f);
          const f32x4 de = dx[0] * sce;
          st4(a.Gi, (uint32_t)pos * (uint32_t)D + (uint32_t)chb, de);
          sq_acc += dot4(de, de);
          if (plead) {  // usert_emb / gamma gradients of this position
            const float gt = ds * (gamma * sH[srow * 2 * LS + p]);  // d / d usert_emb[u][p]
            stg(a.Gu, (uint32_t)(p_u * a.WU + a.di + p), gt);
            sq_acc += gt * gt;
            dgam += ds * (P * uth);
          }
        }
      }
      if (plead && vs)  // padded long slots: zero gradient
        for (int p = n_l + ((par - n_l) % PP + PP) % PP; p < Ls; p += PP) stg(a.Gu, (uint32_t)(p_u * a.WU + a.di + p), 0.0f);
      if (lead && vs)
        for (int p = a.di + Ls; p < a.WU; ++p) stg(a.Gu, (uint32_t)(p_u * a.WU + p), 0.0f);
      stage_accs<1, CPS, false>(acc, dummy, T, lane);
    }
    // scalars of this pass: wave-reduce, stage, one thread sums the waves in fixed order
    {
      float s0 = dgam, s1 = loss_acc, s2 = sq_acc;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        s0 += __shfl_xor(s0, o);
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
      }
      if (lane == 0) {
        sS[wave * 4 + 0] = s0;
        sS[wave * 4 + 1] = s1;
        sS[wave * 4 + 2] = s2;
      }
    }
    TLSAN_STAMP2(10);
    __syncthreads();
    reduce_staged<G, false>(sT, prec, G::P_F1W1, G::P_F1B1, G::P_F1W2, G::P_F1B2, tid);
    if (tid < 3) {
      float s = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += sS[w * 4 + tid];
      prec[G::P_GAMMA + tid] = s;
    }
    TLSAN_STAMP2(11);
    // next pass: sA/sB/sP/sH rows are rewritten by their own wavefront in P1/P2 after the barriers
    // above; sT/sS were last read right above and are rewritten after the P1->P2 barrier
    __syncthreads();
  }
}

#undef TLSAN_STAMP2

template <int D, int DH>
static hipError_t launch_fwd_bwd2(const FwdArgs& a, int grid, hipStream_t st) {
  const size_t smem = fwd2_smem_bytes<D, DH>();
  auto k = k_fwd_bwd2<D, DH>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(k, dim3(grid), dim3(1024), smem, st, a);
  return hipGetLastError();
}
