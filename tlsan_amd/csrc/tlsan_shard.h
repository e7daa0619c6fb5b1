// tlsan_shard.h -- device side of the row-sharded multi-GPU step (tlsan_amd/dist.py): routing plan
// of a batch in the owner-major key space, the step's scalar summary after the all-reduce, and the
// owner-side deterministic apply of the row gradients received from every rank.  No reference
// counterpart (the reference is single-GPU, train.py:53,146); the arithmetic is the update of
// model.py:198-205 on the rows a rank owns.
#pragma once
#include "tlsan_rows.h"

struct RouteArgs {
  const int32_t* keys;      // [n_keys] key of every id the batch touches (duplicates fine)
  int32_t n_keys, R, G;     // R rows per owner, G owners: key = owner * R + local row
  const int32_t* prefix;    // [G*R] compact index of every key (exclusive scan of the flags)
  const int32_t* uniq;      // distinct keys, ascending (= grouped by owner: all-to-all send order)
  const int32_t* n_uniq;
  const int32_t* cate_by_key;  // category of an item key, -1 for user keys
  int32_t* flags;
  int32_t* sendbuf;         // [G][1 + cap]: per owner {count, row numbers inside the owner's shard ...}:
  int32_t cap;              //   equal-split all-to-all payload (count and ids travel together)
  int32_t* cate_c;          // [cate_pad] item -> category map of the compact table (-1 past n_uniq)
  int32_t cate_pad;
  int32_t* comp;            // [n_keys] compact row of every id of the batch
  int32_t* counts_out;      // optional [G]: the per-owner counts once more, e.g. straight into pinned host memory
  int32_t overflow_need;    // > 0: cap is too small for this batch -- every header becomes -overflow_need, no rows
};

__global__ void k_route_mark(RouteArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < a.n_keys) a.flags[a.keys[t]] = 1;
}

__global__ void k_route_finish(RouteArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int nu = *a.n_uniq;
  if (t < a.n_keys) a.comp[t] = a.prefix[a.keys[t]];
  if (t < nu) {
    const int k = a.uniq[t], g = k / a.R;
    if (a.overflow_need == 0) a.sendbuf[(size_t)g * (1 + a.cap) + 1 + (t - a.prefix[(size_t)g * a.R])] = k - g * a.R;
    a.cate_c[t] = a.cate_by_key[k];
    a.flags[k] = 0;  // the marks are zero at rest: no memset per step
  } else if (t < a.cate_pad) {
    a.cate_c[t] = -1;  // rows of the (padded) compact table that are not in use: in no category
  }
  if (t < a.G) {
    int c = (t + 1 < a.G ? a.prefix[(size_t)(t + 1) * a.R] : nu) - a.prefix[(size_t)t * a.R];
    if (a.overflow_need > 0) c = -a.overflow_need;
    a.sendbuf[(size_t)t * (1 + a.cap)] = c;
    if (a.counts_out) a.counts_out[t] = c;
  }
}

// Static-shape plan (tlsan_route_plan_static): every (source, owner) pair exchanges exactly `cap` row slots in both
// directions, so no size ever has to reach the host and the whole step can be recorded in a HIP graph.  The compact
// table of the step is numbered by slot: row of the j-th distinct key of owner g = g * cap + j  (at one rank: the
// plain compact numbering).  Unused slots: category -1, never referenced by the batch.
//   status[0] = largest per-owner count seen so far (atomicMax): the host compares it with cap when it next looks
//   (a batch that needs more than cap rows of one owner is truncated -- the step is then wrong and must be reported).
// status_host (nullable): a word in pinned host memory that receives the count of an overflowing owner directly (a
// system-scope store by the lane that saw it; any such value exceeds cap, which is all the host's per-step look asks) --
// round 5 copied `status` to the host behind every plan: a blit dispatch of 3.5 us on the plan's stream, every step
__global__ void k_route_finish_static(RouteArgs a, int32_t* status, int32_t* status_host) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int nu = *a.n_uniq;
  if (t < a.n_keys) {
    const int key = a.keys[t], g = key / a.R;
    const int j = a.prefix[key] - a.prefix[(size_t)g * a.R];
    a.comp[t] = g * a.cap + min(j, a.cap - 1);
  }
  if (t < a.G * a.cap) {
    const int g = t / a.cap, j = t - g * a.cap;
    const int base = a.prefix[(size_t)g * a.R];
    const int cnt = (g + 1 < a.G ? a.prefix[(size_t)(g + 1) * a.R] : nu) - base;
    int cat = -1;
    if (j < cnt) {
      const int k = a.uniq[base + j];
      a.sendbuf[(size_t)g * (1 + a.cap) + 1 + j] = k - g * a.R;
      cat = a.cate_by_key[k];
    }
    a.cate_c[t] = cat;
    if (j == 0) {
      a.sendbuf[(size_t)g * (1 + a.cap)] = min(cnt, a.cap);
      if (a.counts_out) a.counts_out[g] = cnt;
      if (cnt > a.cap) {
        atomicMax(status, cnt);
        if (status_host != nullptr) __hip_atomic_store(status_host, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  if (t < nu) a.flags[a.uniq[t]] = 0;  // the marks are zero at rest: no memset per step
}

// Owner side of the row fetch: the rows the G ranks asked for (recvbuf [G][1 + cap] as received:
// {count, local row numbers}) copied contiguously in source-rank order (= all-to-all send order),
// plus the flat list of those row numbers for the gradient apply.  One 16-lane group per row.
struct GatherArgs {
  const float* shard; int32_t ld, W;
  const int32_t* recvbuf; int32_t cap, G, n_recv, R;
  float* rows_out; int32_t* recv_rows;
};
__global__ __launch_bounds__(256) void k_shard_gather(GatherArgs a) {
  const int e = blockIdx.x * 16 + (threadIdx.x >> 4), l16 = threadIdx.x & 15;
  if (e >= a.n_recv) return;
  int s = 0, j = e;
  for (;;) {  // source rank of entry e (G <= 16 counts, L2-resident)
    const int c = a.recvbuf[(size_t)s * (1 + a.cap)];
    if (j < c || s + 1 >= a.G) break;
    j -= c;
    ++s;
  }
  int r = a.recvbuf[(size_t)s * (1 + a.cap) + 1 + j];
  r = min(max(r, 0), a.R - 1);
  if (l16 == 0) a.recv_rows[e] = r;
  const float* src = a.shard + (size_t)r * a.ld;
  float* dst = a.rows_out + (size_t)e * a.W;
  for (int c4 = l16; c4 < a.W / 4; c4 += 16) *(f32x4*)(dst + 4 * c4) = *(const f32x4*)(src + 4 * c4);
}

// Static-shape form: slot e = s * cap + j holds the j-th row rank s asked for (or nothing: recv_rows[e] = -1), which is
// also the equal-split all-to-all layout of the reply.  The lazy apply's slot marks (k_slot_mark64) are written here,
// one launch earlier: they depend on the row numbers only.
struct GatherStaticArgs {
  const float* shard; int32_t ld, W;
  const int32_t* recvbuf; int32_t cap, G, R;
  float* rows_out; int32_t* recv_rows;
  unsigned long long* slots64; const uint32_t* stamp_dev;   // nullable; the step's stamp lives on the device (graph replay)
};
__global__ __launch_bounds__(256) void k_shard_gather_static(GatherStaticArgs a) {
  const int e = blockIdx.x * 16 + (threadIdx.x >> 4), l16 = threadIdx.x & 15;
  if (e >= a.G * a.cap) return;
  const int s = e / a.cap, j = e - s * a.cap;
  const int c = a.recvbuf[(size_t)s * (1 + a.cap)];
  int r = j < c ? a.recvbuf[(size_t)s * (1 + a.cap) + 1 + j] : -1;
  if (r < 0 || r >= a.R) r = -1;
  if (l16 == 0) {
    a.recv_rows[e] = r;
    if (r >= 0 && a.slots64) a.slots64[(size_t)r * a.G + s] = ((unsigned long long)*a.stamp_dev << 32) | (unsigned)(e + 1);
  }
  if (r < 0) return;
  const float* src = a.shard + (size_t)r * a.ld;
  float* dst = a.rows_out + (size_t)e * a.W;
  for (int c4 = l16; c4 < a.W / 4; c4 += 16) *(f32x4*)(dst + 4 * c4) = *(const f32x4*)(src + 4 * c4);
}

// bf16 rows on the wire (ShardedModel(wire_dtype="bf16")): the owners keep fp32 rows (the weights that are updated);
// what travels to the ranks that use a row, and what their kernels gather from, is
//   [ d_emb embedding values as bf16, round to nearest even | the row's fp32 tail | pad ]   (pitch bytes per slot)
// where the tail is item_b for item rows (column d_emb of the fused row) and the Ls position weights for user rows
// (columns d_emb .. d_emb + Ls): tlsan_params then points its bf16 tables and its fp32 item_b / usert_emb into the
// same slots with byte-derived strides.  The gradients travel back in fp32 as before.
struct GatherWireArgs {
  GatherStaticArgs g;
  int32_t d_emb, tail;      // embedding width (floats of the fused row that become bf16), fp32 tail floats copied after them
  int32_t pitch;            // bytes per slot of rows_out (multiple of 16)
};
__global__ __launch_bounds__(256) void k_shard_gather_wire_bf16(GatherWireArgs w) {
  const GatherStaticArgs& a = w.g;
  const int e = blockIdx.x * 16 + (threadIdx.x >> 4), l16 = threadIdx.x & 15;
  if (e >= a.G * a.cap) return;
  const int s = e / a.cap, j = e - s * a.cap;
  const int c = a.recvbuf[(size_t)s * (1 + a.cap)];
  int r = j < c ? a.recvbuf[(size_t)s * (1 + a.cap) + 1 + j] : -1;
  if (r < 0 || r >= a.R) r = -1;
  if (l16 == 0) {
    a.recv_rows[e] = r;
    if (r >= 0 && a.slots64) a.slots64[(size_t)r * a.G + s] = ((unsigned long long)*a.stamp_dev << 32) | (unsigned)(e + 1);
  }
  if (r < 0) return;
  const float* src = a.shard + (size_t)r * a.ld;
  char* dst = (char*)a.rows_out + (size_t)e * w.pitch;
  for (int c4 = l16; c4 < w.d_emb / 4; c4 += 16) {
    const s16x4 pk = mm_pack<TLSAN_MATRIX_BF16>(*(const f32x4*)(src + 4 * c4));   // v_cvt_pk_bf16_f32: round to nearest even
    *(s16x4*)(dst + 8 * c4) = pk;
  }
  for (int t = l16; t < w.tail; t += 16) *(float*)(dst + 2 * w.d_emb + 4 * t) = src[w.d_emb + t];
}

// ------------------------------------------------------------------------------------------
// After the all-reduce of flat = [dense grads | cate grads | loss | per-use squares | table squares]
// (sums over the G ranks of per-rank MEANS over their own batches): global norm, clip coefficient,
// loss, step size (device scalar for the owners' apply) and the SGD update of the replicated dense
// parameters + the K^T copy.  Every rank computes identical bits.
struct SummaryArgs {
  const float* flat; int32_t n_dense, n_cate, G;
  float lr, reg, clip;
  const double* S_cate;   // sum of squares of the (replicated) category table
  float* dense; float* dense_KT; int32_t D, K_off, k0_off;
  float* step_dev; float* loss_out; float* gnorm_out;   // step_dev[0] = lr * coef, step_dev[1] = coef
  OptCtx oc;                       // optimizers other than SGD (oc.opt != 0): accumulators of the dense parameters
  float* dense_s1; float* dense_s2;
  const float* P_dev;              // lazy L2 (SGD): table scale, tables = P * stored; step_dev[2] = lr coef / P_new,
};                                 //   step_dev[3] = P_new = P (1 - lr coef reg), committed by k_shard_apply_lazy

// Every workgroup recomputes the norm (n_dense L2-resident floats, same fixed tree -> same bits)
// and then updates its own 1024-element slice: no second launch, no cross-workgroup hand-over.
__global__ __launch_bounds__(1024) void k_shard_summary(SummaryArgs a) {
  __shared__ double sh[1024];
  __shared__ float sh_step, sh_coef;
  const int tid = threadIdx.x;
  const float inv_g = 1.0f / (float)a.G;
  double s = 0.0;
  for (int k0 = tid; k0 < a.n_dense; k0 += 1024 * 8) {  // 8 loads in flight
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = a.flat[k0 + 1024 * u < a.n_dense ? k0 + 1024 * u : k0];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + 1024 * u < a.n_dense) {
        const double g = (double)(v[u] * inv_g);
        s += g * g;
      }
  }
  sh[tid] = s;
  __syncthreads();
  for (int o = 512; o >= 1; o >>= 1) {
    if (tid < o) sh[tid] += sh[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    const float* tail = a.flat + a.n_dense + a.n_cate;
    const float P = a.P_dev ? *a.P_dev : 1.0f;
    const double S_tot = ((double)tail[2] + *a.S_cate) * (double)P * (double)P;   // stored sums -> true tables
    const double sq = (double)tail[1] * (double)(inv_g * inv_g) + (double)a.reg * (double)a.reg * S_tot + sh[0];
    const float norm = (float)sqrt(sq);
    const float coef = clip_coef(norm, a.clip);  // clip_by_global_norm (model.py:201)
    sh_step = coef * a.lr;
    sh_coef = coef;
    if (blockIdx.x == 0) {
      a.step_dev[0] = sh_step;
      a.step_dev[1] = coef;
      if (a.P_dev) {
        const float Pn = P * (1.0f - sh_step * a.reg);
        a.step_dev[2] = sh_step / Pn;
        a.step_dev[3] = Pn;
      }
      *a.gnorm_out = norm;
      *a.loss_out = tail[0] * inv_g + a.reg * 0.5f * (float)S_tot;
    }
  }
  __syncthreads();
  const float step = sh_step;
  const int k = blockIdx.x * 1024 + tid;
  if (k < a.n_dense) {
    float w = a.dense[k];
    if (a.oc.opt == TLSAN_OPT_SGD) {
      w -= step * (a.flat[k] * inv_g);
    } else {  // adam | rmsprop | adadelta on the clipped gradient (tlsan_optimizer)
      float s1 = a.dense_s1[k], s2 = a.dense_s2[k];
      opt_elem(a.oc, w, sh_coef * (a.flat[k] * inv_g), s1, s2);
      a.dense_s1[k] = s1; a.dense_s2[k] = s2;
    }
    a.dense[k] = w;
    if (k >= a.K_off && k < a.k0_off) {
      const int idx = k - a.K_off;
      a.dense_KT[(size_t)(idx % a.D) * a.D + idx / a.D] = w;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Owner-side apply.  The rows received from source rank s are distinct, so a destination row has at
// most one contribution per source: k_slot_mark files entry e of source s under slots[row][s]
// (plain stores, no counting sort), k_shard_apply sums a row's contributions in source order
// (fixed order -> bitwise reproducible), applies the dense-L2 SGD update to EVERY local row
// (model.py:164-169,198-205) and clears the slots.  Category rows (replicated table, gradient
// already summed by the all-reduce) are the trailing workgroups.
#define SHARD_GMAX 16
#define SHARD_NCH 4  // 16 lanes x 4 chunks x 4 floats = 256 columns max
struct ShardApplyArgs {
  float* shard; int32_t ld, cI, R, W, reg_item, reg_user;
  const float* vals; int32_t ldv; const int32_t* rows; int32_t n_recv;
  int32_t src_off[SHARD_GMAX + 1]; int32_t G;
  int32_t* slots;  // [R][G], zero at rest
  float gscale; const float* step_dev; float reg;
  float* cate_emb; int32_t C, dc; const float* g_cate;
  double* part_out; int32_t nb_rows, nb_cate;
  // optimizers other than SGD: accumulators laid out like the shard rows / like cate_emb; step_dev[1] = coef
  OptCtx oc;
  float* shard_s1; float* shard_s2; float* cate_s1; float* cate_s2;
  int32_t bias_col;   // column of item_b inside an item row: touched where gathered (sparse RMSProp / Adadelta)
};

__global__ void k_slot_mark(ShardApplyArgs a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.n_recv) return;
  int s = 0;
  while (s + 1 < a.G && e >= a.src_off[s + 1]) ++s;
  const int r = a.rows[e];
  if (r >= 0 && r < a.R) a.slots[(size_t)r * a.G + s] = e + 1;
}

__global__ __launch_bounds__(256) void k_shard_apply(ShardApplyArgs a) {
  __shared__ double shd[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = lane >> 4, l16 = lane & 15;
  const int blk = blockIdx.x;
  const bool is_cate = blk >= a.nb_rows;
  const int row = (is_cate ? blk - a.nb_rows : blk) * AP_ROWS_PB + wave * 4 + grp;
  const bool vr = row < (is_cate ? a.C : a.R);
  const int rc = vr ? row : 0;
  const int width = is_cate ? a.dc : a.W;
  const int W4 = width / 4;
  const int reg_cols = is_cate ? a.dc : (rc < a.cI ? a.reg_item : a.reg_user);
  float* Wr = is_cate ? a.cate_emb + (size_t)rc * a.dc : a.shard + (size_t)rc * a.ld;
  const float step = *a.step_dev;
  f32x4 w[SHARD_NCH];
#pragma unroll
  for (int ch = 0; ch < SHARD_NCH; ++ch)
    if (l16 + 16 * ch < W4) w[ch] = *(const f32x4*)(Wr + 4 * (l16 + 16 * ch));
  double acc[SHARD_NCH][4];
#pragma unroll
  for (int ch = 0; ch < SHARD_NCH; ++ch)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[ch][i] = 0.0;
  if (is_cate) {
#pragma unroll
    for (int ch = 0; ch < SHARD_NCH; ++ch)
      if (l16 + 16 * ch < W4) {
        const f32x4 v = *(const f32x4*)(a.g_cate + (size_t)rc * a.dc + 4 * (l16 + 16 * ch));
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[ch][i] = (double)v[i];
      }
  } else {
    int32_t* sl = a.slots + (size_t)rc * a.G;
    for (int s0 = 0; s0 < a.G; s0 += 4) {  // 4 sources in flight (clamped addresses, masked sum)
      int e[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) e[u] = (s0 + u < a.G && vr) ? sl[s0 + u] : 0;
      f32x4 v[4][SHARD_NCH];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* src = a.vals + (size_t)(e[u] > 0 ? e[u] - 1 : 0) * a.ldv;
#pragma unroll
        for (int ch = 0; ch < SHARD_NCH; ++ch)
          if (l16 + 16 * ch < W4) v[u][ch] = *(const f32x4*)(src + 4 * (l16 + 16 * ch));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (e[u] > 0) {
#pragma unroll
          for (int ch = 0; ch < SHARD_NCH; ++ch)
            if (l16 + 16 * ch < W4) {
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[ch][i] += (double)v[u][ch][i];
            }
          if (l16 == 0) sl[s0 + u] = 0;
        }
    }
  }
  double part = 0.0;
  if (vr) {
#pragma unroll
    for (int ch = 0; ch < SHARD_NCH; ++ch) {
      const int c4 = l16 + 16 * ch;
      if (c4 < W4) {
        if (a.oc.opt == TLSAN_OPT_SGD) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bool rg = 4 * c4 + i < reg_cols;
            const float g = a.gscale * (float)acc[ch][i] + (rg ? a.reg * w[ch][i] : 0.0f);
            w[ch][i] -= step * g;
            if (rg) part += (double)w[ch][i] * (double)w[ch][i];
          }
        } else {
          float* S1 = (is_cate ? a.cate_s1 + (size_t)rc * a.dc : a.shard_s1 + (size_t)rc * a.ld) + 4 * c4;
          float* S2 = (is_cate ? a.cate_s2 + (size_t)rc * a.dc : a.shard_s2 + (size_t)rc * a.ld) + 4 * c4;
          f32x4 m1 = *(const f32x4*)S1, m2 = *(const f32x4*)S2;
          const float coef = a.step_dev[1];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int col = 4 * c4 + i;
            const bool rg = col < reg_cols;
            const float g = a.gscale * (float)acc[ch][i] + (rg ? a.reg * w[ch][i] : 0.0f);
            // item_b reaches the optimizer as the gathered rows only: sparse RMSProp / Adadelta leave the rest alone
            const bool skip = !is_cate && rc < a.cI && col == a.bias_col && g == 0.0f && a.oc.opt != TLSAN_OPT_ADAM;
            if (!skip) {
              float wi = w[ch][i], a1 = m1[i], a2 = m2[i];
              opt_elem(a.oc, wi, coef * g, a1, a2);
              w[ch][i] = wi; m1[i] = a1; m2[i] = a2;
            }
            if (rg) part += (double)w[ch][i] * (double)w[ch][i];
          }
          *(f32x4*)S1 = m1;
          *(f32x4*)S2 = m2;
        }
        *(f32x4*)(Wr + 4 * c4) = w[ch];
      }
    }
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) part += __shfl_xor(part, o);
  if (lane == 0) shd[wave] = part;
  __syncthreads();
  if (tid == 0) a.part_out[blk] = shd[0] + shd[1] + shd[2] + shd[3];
}

// ---- lazy-L2 owner update (SGD): W = P * W_stored with one scale P that every rank advances alike, so an
// owner touches only the rows whose gradients arrived (tlsan_shard_apply_lazy).  Received entries are
// filed under slots64[row][source] = (stamp << 32 | entry + 1): a per-step stamp instead of clearing, so
// nothing has to be zero at rest; the entry of the lowest source owns its row and adds the contributions
// of all sources in source order (fixed order -> bitwise reproducible).
struct ShardLazyArgs {
  float* shard; int32_t ld, cI, R, W, reg_item, reg_user;
  const float* vals; int32_t ldv; const int32_t* rows; int32_t n_recv;
  int32_t src_off[SHARD_GMAX + 1]; int32_t G;
  unsigned long long* slots64; uint32_t stamp;
  float gscale; const float* step_dev;
  float* cate_emb; int32_t C, dc; const float* g_cate;
  double* part_out; int32_t nb_rows, nb_cate;
  float* P_dev;
  uint32_t* stamp_dev;   // static-shape step: the stamp is read from (and advanced on) the device, `stamp` is unused
};

__global__ void k_slot_mark64(ShardLazyArgs a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.n_recv) return;
  const uint32_t stamp = a.stamp_dev ? *a.stamp_dev : a.stamp;
  int s = 0;
  while (s + 1 < a.G && e >= a.src_off[s + 1]) ++s;
  const int r = a.rows[e];
  if (r >= 0 && r < a.R) a.slots64[(size_t)r * a.G + s] = ((unsigned long long)stamp << 32) | (unsigned)(e + 1);
}

// one workgroup's share of the lazy apply; returns this thread's part of the change of the sums of squares
__device__ __forceinline__ double shard_apply_lazy_part(const ShardLazyArgs& a) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = lane >> 4, l16 = lane & 15;
  const int blk = blockIdx.x;
  const bool is_cate = blk >= a.nb_rows;
  const float step = a.step_dev[0], lazy = a.step_dev[2];
  const uint32_t stamp = a.stamp_dev ? *a.stamp_dev : a.stamp;
  if (blk == 0 && tid == 0) *a.P_dev = a.step_dev[3];   // (nothing in this launch reads P)
  double part = 0.0;
  if (is_cate) {  // replicated category table: every row, W_stored -= (step / P_new) * g
    const int row = (blk - a.nb_rows) * AP_ROWS_PB + wave * 4 + grp;
    if (row < a.C) {
      float* Wr = a.cate_emb + (size_t)row * a.dc;
      for (int c4 = l16; c4 < a.dc / 4; c4 += 16) {
        f32x4 w = *(const f32x4*)(Wr + 4 * c4);
        const f32x4 g = *(const f32x4*)(a.g_cate + (size_t)row * a.dc + 4 * c4);
        w -= (lazy * a.gscale) * g;
        *(f32x4*)(Wr + 4 * c4) = w;
#pragma unroll
        for (int i = 0; i < 4; ++i) part += (double)w[i] * (double)w[i];
      }
    }
  } else {
    const int e = blk * AP_ROWS_PB + wave * 4 + grp;
    const int r = e < a.n_recv ? a.rows[e] : -1;   // (static-shape exchange: slots nobody filled hold -1)
    if (r >= 0 && r < a.R) {
      int s = 0;
      while (s + 1 < a.G && e >= a.src_off[s + 1]) ++s;
      const unsigned long long* sl = a.slots64 + (size_t)r * a.G;
      bool owner = true;
      for (int s1 = 0; s1 < s; ++s1) owner = owner && (uint32_t)(sl[s1] >> 32) != stamp;
      if (owner) {
        const int W4 = a.W / 4;
        const int reg_cols = r < a.cI ? a.reg_item : a.reg_user;
        float* Wr = a.shard + (size_t)r * a.ld;
        for (int c4 = l16; c4 < W4; c4 += 16) {
          double acc[4] = {0.0, 0.0, 0.0, 0.0};
          for (int s2 = s; s2 < a.G; ++s2) {  // contributions in source order
            const unsigned long long v = sl[s2];
            // (the entry is bounds-checked as well: a stamp alone cannot vouch for a slot written by an
            //  earlier life of the model, e.g. after a restore to an earlier step)
            if ((uint32_t)(v >> 32) == stamp && (uint32_t)v - 1u < (uint32_t)a.n_recv) {
              const f32x4 g = *(const f32x4*)(a.vals + (size_t)((uint32_t)v - 1u) * a.ldv + 4 * c4);
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[i] += (double)g[i];
            }
          }
          f32x4 w = *(const f32x4*)(Wr + 4 * c4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bool rg = 4 * c4 + i < reg_cols;
            const float w0 = w[i];
            // regularised columns are stored / P; item_b (and the padding) are not scaled
            w[i] = w0 - (rg ? lazy : step) * (a.gscale * (float)acc[i]);
            if (rg) part += (double)w[i] * (double)w[i] - (double)w0 * (double)w0;
          }
          *(f32x4*)(Wr + 4 * c4) = w;
        }
      }
    }
  }
  return part;
}

__global__ __launch_bounds__(256) void k_shard_apply_lazy(ShardLazyArgs a) {
  __shared__ double shd[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double part = shard_apply_lazy_part(a);
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) part += __shfl_xor(part, o);
  if (lane == 0) shd[wave] = part;
  __syncthreads();
  if (tid == 0) a.part_out[blockIdx.x] = shd[0] + shd[1] + shd[2] + shd[3];
}

// out[0] += sum(part[0, n0)) (changes of the stored shard rows' sum of squares), out[1] = sum(part[n0, n0+n1))
// stamp_dev (nullable): the static-shape step keeps its stamp on the device; every workgroup of the apply has read this
// step's value by now, the next step files under the next one (never 0, the value of a cleared slot)
__global__ __launch_bounds__(256) void k_reduce_lazy2(const double* part, int n0, int n1, double* out, float* sq_f32, uint32_t* stamp_dev) {
  __shared__ double shd[256];
  const int b = blockIdx.x;
  const double s = block_sum_double(part + (b ? n0 : 0), b ? n1 : n0, shd);
  if (threadIdx.x == 0) {
    if (b == 0) {
      out[0] += s;
      if (sq_f32) *sq_f32 = (float)out[0];
      if (stamp_dev) {
        const uint32_t n = *stamp_dev + 1u;
        *stamp_dev = n == 0xFFFFFFFFu ? 1u : n;
      }
    } else {
      out[1] = s;
    }
  }
}

// out[0] = sum(part[0, n0)), out[1] = sum(part[n0, n0+n1)); sq_f32 (nullable) = (float)out[0]
__global__ __launch_bounds__(256) void k_reduce_double2(const double* part, int n0, int n1, double* out, float* sq_f32) {
  __shared__ double shd[256];
  const int b = blockIdx.x;
  const double s = block_sum_double(part + (b ? n0 : 0), b ? n1 : n0, shd);
  if (threadIdx.x == 0) {
    out[b] = s;
    if (b == 0 && sq_f32) *sq_f32 = (float)s;
  }
}
