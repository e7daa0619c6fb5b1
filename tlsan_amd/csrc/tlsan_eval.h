// tlsan_eval.h -- all-items scoring without materialising eval_logits (model.py:140-156).
// For every test row the metric ops only need the RANK of the label inside
//   u_t . [item_emb || cate_emb[item_cate]]^T + item_b
// under tf.nn.top_k's order (higher first, ties -> lower id first): hit@k == rank < k.
// Scores are f32 MFMA tiles (16 users x 16 items); the label's own score is produced by the
// same instruction sequence (k_eval_label) so equality tests are bit-consistent.
#pragma once
#include "tlsan_common.h"

struct EvalArgs {
  tlsan_params p;
  const float* u_t;        // [B, D]
  const int32_t* labels;   // [B]
  int32_t B, I, di, dc;
  float* s_label;          // [B]
  int32_t* ranks;          // [B], zeroed before k_eval_rank
  float* all_emb;          // [I, D] dense [item_emb || cate_emb[item_cate]] (model.py:89-90), or NULL
  // item-sharded scoring: local item n has the global id n * id_mul + id_add (labels are global ids)
  int32_t id_mul, id_add;
};

__device__ __forceinline__ f32x4 all_emb4(const EvalArgs& a, int it, int c) {
  if (a.p.table_dtype == TLSAN_TABLE_BF16)   // (evaluation: a run-time switch is good enough here)
    return (c < a.di) ? tbl_ld4<TLSAN_TABLE_BF16>(a.p.item_emb, (size_t)it * a.p.ld_item + c)
                      : tbl_ld4<TLSAN_TABLE_BF16>(a.p.cate_emb, (size_t)a.p.item_cate[it] * a.dc + (c - a.di));
  return (c < a.di) ? tbl_ld4<TLSAN_TABLE_F32>(a.p.item_emb, (size_t)it * a.p.ld_item + c)
                    : tbl_ld4<TLSAN_TABLE_F32>(a.p.cate_emb, (size_t)a.p.item_cate[it] * a.dc + (c - a.di));
}

template <int D>
__device__ __forceinline__ void load_user_frag(const EvalArgs& a, int u0, int q, int r,
                                               f32x4 (&af)[D / 16]) {
  const int u = u0 + r;
#pragma unroll
  for (int kc = 0; kc < D / 16; ++kc)
    af[kc] = (u < a.B) ? *(const f32x4*)(a.u_t + (size_t)u * D + 16 * kc + 4 * q) : (f32x4)(0.0f);
}

// scores[user 4q+i][item of column r]
template <int D>
__device__ __forceinline__ f32x4 score_tile(const EvalArgs& a, const f32x4 (&af)[D / 16], int item,
                                            int q) {
  f32x4 acc = (f32x4)(0.0f);
#pragma unroll
  for (int kc = 0; kc < D / 16; ++kc) {
    const f32x4 bv = all_emb4(a, item, 16 * kc + 4 * q);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = TLSAN_MFMA(af[kc][s], bv[s], acc);
  }
  return acc;
}

template <int D>
__global__ __launch_bounds__(64) void k_eval_label(EvalArgs a) {
  const int lane = threadIdx.x, q = lane >> 4, r = lane & 15;
  const int u0 = blockIdx.x * 16;
  f32x4 af[D / 16];
  load_user_frag<D>(a, u0, q, r, af);
  const int u = u0 + r;
  const int item = (u < a.B) ? a.labels[u] : 0;
  const float P = a.p.scale ? *a.p.scale : 1.0f;
  const f32x4 acc = score_tile<D>(a, af, item, q) * P;
  const float bias = a.p.item_b[(size_t)item * a.p.ld_itemb];
  // diagonal: user (4q+i) == column r
  if (u < a.B && q == (r >> 2)) {
    float v = acc[0];
    if ((r & 3) == 1) v = acc[1];
    if ((r & 3) == 2) v = acc[2];
    if ((r & 3) == 3) v = acc[3];
    a.s_label[u] = v + bias;
  }
}

// grid (ceil(B/16), item chunks); 4 wavefronts per workgroup stride over the item tiles
template <int D>
__global__ __launch_bounds__(256) void k_eval_rank(EvalArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4, r = lane & 15;
  const int u0 = blockIdx.x * 16;
  f32x4 af[D / 16];
  load_user_frag<D>(a, u0, q, r, af);
  float sl[4];
  int lab[4], cnt[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = u0 + 4 * q + i;
    sl[i] = (u < a.B) ? a.s_label[u] : 0.0f;
    lab[i] = (u < a.B) ? a.labels[u] : -1;
    cnt[i] = 0;
  }
  const float P = a.p.scale ? *a.p.scale : 1.0f;
  const int ntiles = (a.I + 15) / 16;
  for (int t = blockIdx.y * 4 + wave; t < ntiles; t += gridDim.y * 4) {
    const int n = t * 16 + r;
    const bool vn = n < a.I;
    const int item = vn ? n : a.I - 1;
    const f32x4 acc = score_tile<D>(a, af, item, q) * P;
    const float bias = a.p.item_b[(size_t)item * a.p.ld_itemb];
    const int gn = n * a.id_mul + a.id_add;  // global item id
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float s = acc[i] + bias;
      const bool ahead = vn && gn != lab[i] && (s > sl[i] || (s == sl[i] && gn < lab[i]));
      cnt[i] += ahead ? 1 : 0;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) cnt[i] += __shfl_xor(cnt[i], o);
    const int u = u0 + 4 * q + i;
    if (r == 0 && u < a.B && cnt[i] != 0) atomicAdd(&a.ranks[u], cnt[i]);
  }
}

// model.py:89-90: all_emb = concat(item_emb, gather(cate_emb, item_cate_list)) as one dense [I, D]
// matrix (stored values; the table scale P is applied to the scores).  One 16-B piece per thread.
template <int D>
__global__ void k_all_emb(EvalArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.I * (D / 4)) return;
  const int it = t / (D / 4), c = 4 * (t % (D / 4));
  *(f32x4*)(a.all_emb + (size_t)it * D + c) = all_emb4(a, it, c);
}

// Ranking against the dense all_emb: a wavefront scores 16 users x 64 items at a time (4
// independent MFMA accumulators; the A fragments of the user tile stay in registers) and strides
// over the item groups; grid (user tiles, enough item-group slices to fill the chip).  The chain
// of one 16x16 tile is the same instruction sequence on the same operand values as score_tile /
// k_eval_label, so equality with the label's own score is bit-consistent.
template <int D>
__global__ __launch_bounds__(256) void k_eval_rank_dense(EvalArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4, r = lane & 15;
  const int u0 = blockIdx.x * 16;
  f32x4 af[D / 16];
  load_user_frag<D>(a, u0, q, r, af);
  const float P = a.p.scale ? *a.p.scale : 1.0f;
  float sl[4];
  int lab[4], cnt[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = u0 + 4 * q + i;
    sl[i] = (u < a.B) ? a.s_label[u] : 0.0f;
    lab[i] = (u < a.B) ? a.labels[u] : -1;
    cnt[i] = 0;
  }
  for (int n0 = (blockIdx.y * 4 + wave) * 64; n0 < a.I; n0 += gridDim.y * 4 * 64) {  // first item of the group
    const float* rows[4];
    int item[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      item[tt] = min(n0 + 16 * tt + r, a.I - 1);
      rows[tt] = a.all_emb + (size_t)item[tt] * D + 4 * q;
    }
    f32x4 acc[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[tt] = (f32x4)(0.0f);
#pragma unroll
    for (int kc = 0; kc < D / 16; ++kc) {
      f32x4 bv[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) bv[tt] = *(const f32x4*)(rows[tt] + 16 * kc);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[tt] = TLSAN_MFMA(af[kc][s], bv[tt][s], acc[tt]);
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int n = n0 + 16 * tt + r;
      const bool vn = n < a.I;
      const int gn = n * a.id_mul + a.id_add;  // global item id
      const float bias = a.p.item_b[(size_t)item[tt] * a.p.ld_itemb];
      const f32x4 sc = acc[tt] * P;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float s = sc[i] + bias;
        const bool ahead = vn && gn != lab[i] && (s > sl[i] || (s == sl[i] && gn < lab[i]));
        cnt[i] += ahead ? 1 : 0;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) cnt[i] += __shfl_xor(cnt[i], o);
    const int u = u0 + 4 * q + i;
    if (r == 0 && u < a.B && cnt[i] != 0) atomicAdd(&a.ranks[u], cnt[i]);
  }
}
