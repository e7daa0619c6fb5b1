// tlsan_update.h -- everything after the fused forward/backward kernel:
//   k_index<FILL>     inverted index (destination row -> list of per-use gradient rows)
//   k_index_scan      exclusive scan of the per-row counts
//   k_dk_partial      dK = long^T . dbridge  (split over the batch, f32 MFMA)
//   k_dense_finalize  fixed-order reduction of all dense-parameter gradient partials
//   k_apply_rows      exact segment-sum of the per-use rows + clip + SGD (model.py:198-205)
// Determinism: integer atomics only build *which* rows belong to a destination; the float
// sums are order-independent (exact_term) or in a fixed order, so two runs are bitwise equal.
#pragma once
#include "tlsan_common.h"

struct IdxArgs {
  tlsan_batch b;
  const int32_t* item_cate;
  int32_t Ls, S;
  int32_t* cnt_item; int32_t* cnt_cate; int32_t* cnt_user;   // persistent, zero at rest
  int32_t* cur_item; int32_t* cur_cate; int32_t* cur_user;   // fill cursors (start = offsets)
  int32_t* list_item; int32_t* list_cate; int32_t* list_user;
};

// one thread per (sample, slot); slot order: [0,Ls) long, [Ls,Ls+Sn) session, Ls+Sn candidate,
// Ls+Sn+1 user.  The contribution code stored in the lists is c = b*S + slot = the row of G.
template <bool FILL>
__global__ void k_index(IdxArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int B = a.b.B, S = a.S, Ls = a.Ls, Sn = a.b.Sn;
  if (t >= B * S) return;
  const int b = t / S, slot = t - b * S;
  int item = -1, cate = -1, user = -1;
  if (slot < Ls) {
    if (slot < min(a.b.sl[b], Ls)) item = a.b.hist_i[(size_t)b * Ls + slot];
  } else if (slot < Ls + Sn) {
    const int k = slot - Ls;
    if (k < min(a.b.sl_new[b], Sn)) item = a.b.hist_i_new[(size_t)b * Sn + k];
  } else if (slot == Ls + Sn) {
    item = a.b.i[b];
  } else {
    user = a.b.u[b];
    cate = a.b.u_cate[b];
  }
  if (item >= 0) cate = a.item_cate[item];
  if constexpr (!FILL) {
    if (item >= 0) atomicAdd(&a.cnt_item[item], 1);
    if (cate >= 0) atomicAdd(&a.cnt_cate[cate], 1);
    if (user >= 0) atomicAdd(&a.cnt_user[user], 1);
  } else {
    if (item >= 0) a.list_item[atomicAdd(&a.cur_item[item], 1)] = t;
    if (cate >= 0) a.list_cate[atomicAdd(&a.cur_cate[cate], 1)] = t;
    if (user >= 0) a.list_user[atomicAdd(&a.cur_user[user], 1)] = t;
  }
}

struct ScanArgs {
  const int32_t* cnt[3];
  int32_t* off[3];
  int32_t* cur[3];
  int32_t n[3];
};

// blockIdx.x selects the table; 1024 threads x 4 ids per iteration
__global__ __launch_bounds__(1024) void k_index_scan(ScanArgs a) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int which = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t* cnt = a.cnt[which];
  int32_t* off = a.off[which];
  int32_t* cur = a.cur[which];
  const int n = a.n[which];
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 4096) {
    const int i0 = base + tid * 4;
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (i0 + k < n) ? cnt[i0 + k] : 0;
    const int tsum = v[0] + v[1] + v[2] + v[3];
    int inc = tsum;  // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int wbase = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) wbase += (w < wave) ? wsum[w] : 0;
    int run = carry + wbase + inc - tsum;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i0 + k < n) {
        off[i0 + k] = run;
        cur[i0 + k] = run;
      }
      run += v[k];
    }
    __syncthreads();
    if (tid == 1023) carry = run;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// dK[k][j] = sum_b long[b][k] * dbridge[b][j]  (gradient of tf.layers.dense's kernel,
// model.py:347) for the samples [blockIdx.x*chunk, +chunk).  C[M=k][N=j], K-dim = samples.
template <int D>
__global__ __launch_bounds__(512) void k_dk_partial(const float* __restrict__ gLong,
                                                    const float* __restrict__ gDB, int B, int chunk,
                                                    float* __restrict__ Kp) {
  constexpr int NT = D / 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, q = lane >> 4, r = lane & 15;
  const int b0 = blockIdx.x * chunk, b1 = min(B, b0 + chunk);
  float* out = Kp + (size_t)blockIdx.x * D * D;
  for (int tile = wave; tile < NT * NT; tile += 8) {
    const int kt = tile / NT, jt = tile % NT;
    f32x4 acc = (f32x4)(0.0f);
    for (int bs = b0; bs < b1; bs += 16) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int bi = bs + 4 * q + s;
        const bool v = bi < b1;
        const float av = v ? gLong[(size_t)bi * D + 16 * kt + r] : 0.0f;
        const float bv = v ? gDB[(size_t)bi * D + 16 * jt + r] : 0.0f;
        acc = TLSAN_MFMA(av, bv, acc);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[(size_t)(16 * kt + 4 * q + i) * D + 16 * jt + r] = acc[i];
  }
}

// ------------------------------------------------------------------------------------------
struct FinArgs {
  tlsan_dense_layout lay;
  const float* partials;  // [nrec][NPB]
  int32_t nrec;
  const float* Kp;        // [nsplit][D*D]
  int32_t nsplit;
  float* gd;              // [n_dense] reduced dense gradients
  float* sqd;             // [gridDim.x - 1] per-block sum of gd^2
  float* scal;            // [0] = sum of per-sample BCE, [1] = sum of squares of per-use rows
  const double* S_part;   // per-row-block sums of squares of the regularised tables
  int32_t n_spart;
  double* S_total;
};

// fixed-order sum of doubles by one 256-thread block
__device__ __forceinline__ double block_sum_double(const double* __restrict__ v, int n, double* sh) {
  const int tid = threadIdx.x;
  double s = 0.0;
  for (int k = tid; k < n; k += 256) s += v[k];
  sh[tid] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (tid < o) sh[tid] += sh[tid + o];
    __syncthreads();
  }
  return sh[0];
}

// grid = ceil(n_dense/256) + 1; the last block reduces S_part -> S_total
template <int D, int DH>
__global__ __launch_bounds__(256) void k_dense_finalize(FinArgs a) {
  using G = Geo<D, DH>;
  constexpr int CW = G::CW, NPB = G::NPB, HPC = CW / DH;  // heads per 16-wide column block
  __shared__ double shd[256];
  const int tid = threadIdx.x;
  if (blockIdx.x == gridDim.x - 1) {
    const double s = block_sum_double(a.S_part, a.n_spart, shd);
    if (tid == 0) *a.S_total = s;
    return;
  }
  const int n = blockIdx.x * 256 + tid;
  const tlsan_dense_layout& L = a.lay;
  float g = 0.0f;
  if (n < L.n_dense) {
    if (n >= L.K && n < L.k0) {
      const int idx = n - L.K;
      for (int s = 0; s < a.nsplit; ++s) g += a.Kp[(size_t)s * D * D + idx];
    } else {
      // map the true parameter index to 1..HPC entries of the effective-layout record
      int e[2] = {-1, -1};
      const int wofs[4] = {L.f1_W1, L.f1_W2, L.f2_W1, L.f2_W2};
      const int bofs[4] = {L.f1_b1, L.f1_b2, L.f2_b1, L.f2_b2};
      const int pw[4] = {G::P_F1W1, G::P_F1W2, G::P_F2W1, G::P_F2W2};
      const int pb[4] = {G::P_F1B1, G::P_F1B2, G::P_F2B1, G::P_F2B2};
      for (int m = 0; m < 4; ++m) {
        if (n >= wofs[m] && n < wofs[m] + DH * DH) {
          const int k = (n - wofs[m]) / DH, j = (n - wofs[m]) % DH;
          for (int h = 0; h < HPC; ++h) e[h] = pw[m] + (h * DH + k) * CW + h * DH + j;
        }
        if (n >= bofs[m] && n < bofs[m] + DH) {
          const int j = n - bofs[m];
          for (int h = 0; h < HPC; ++h) e[h] = pb[m] + h * DH + j;
        }
      }
      if (n >= L.k0 && n < L.k0 + D) e[0] = G::P_K0 + (n - L.k0);
      if (n == L.gamma) e[0] = G::P_GAMMA;
      for (int rec = 0; rec < a.nrec; ++rec) {
        const float* p = a.partials + (size_t)rec * NPB;
        float t = p[e[0]];
        if (HPC > 1 && e[1] >= 0) t += p[e[1]];
        g += t;
      }
    }
    a.gd[n] = g;
  }
  shd[tid] = (double)g * (double)g;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (tid < o) shd[tid] += shd[tid + o];
    __syncthreads();
  }
  if (tid == 0) a.sqd[blockIdx.x] = (float)shd[0];
  if (blockIdx.x == 0 && tid < 2) {
    float s = 0.0f;
    for (int rec = 0; rec < a.nrec; ++rec) s += a.partials[(size_t)rec * NPB + G::P_LOSS + tid];
    a.scal[tid] = s;
  }
}

__global__ __launch_bounds__(256) void k_reduce_double(const double* v, int n, double* out) {
  __shared__ double shd[256];
  const double s = block_sum_double(v, n, shd);
  if (threadIdx.x == 0) *out = s;
}

__global__ void k_transpose_K(const float* __restrict__ K, float* __restrict__ KT, int D) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < D * D) KT[(size_t)(t % D) * D + t / D] = K[t];
}

// ------------------------------------------------------------------------------------------
enum { AP_UPDATE = 0, AP_GRADS = 1, AP_SUMSQ = 2, AP_ROWNORM = 3 };

struct ApplyArgs {
  tlsan_params p;
  tlsan_grads_out go;
  tlsan_dense_layout lay;
  int32_t I, U, C, Ls, D, di, dc, S, Sn;
  const float* G; const float* GT; const float* dlogit;
  int32_t* cnt_item; int32_t* cnt_cate; int32_t* cnt_user;
  const int32_t* off_item; const int32_t* off_cate; const int32_t* off_user;
  const int32_t* list_item; const int32_t* list_cate; const int32_t* list_user;
  const float* gd; const float* sqd; int32_t nsqd; const float* scal;
  double* part_out;        // UPDATE/SUMSQ: new sums of squares per row block; ROWNORM: sum g^2
  const double* S_total;   // sum of squares of the four regularised tables (current params)
  const double* rownorm;   // dedup mode: sum over rows of |g_row|^2 (from the ROWNORM pass)
  float lr, reg, clip, inv_B;
  int32_t norm_mode;
  float* out_loss; float* out_gnorm;
  int32_t nbI, nbU, nbC, nbD;
};

// add the rows list[first], list[first+stride], ... (each: 4*W4 floats at G[c*D + colofs]) exactly
__device__ __forceinline__ void accum_list(const int32_t* __restrict__ list, int n, int first,
                                           int stride, const float* __restrict__ G, int D, int colofs,
                                           int W4, int l16, double (&acc)[2][4]) {
  int k = first;
  for (; k + stride < n; k += 2 * stride) {  // two independent rows in flight
    const int c0 = list[k], c1 = list[k + stride];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const int c4 = l16 + 16 * ch;
      if (c4 < W4) {
        const f32x4 v0 = *(const f32x4*)(G + (size_t)c0 * D + colofs + 4 * c4);
        const f32x4 v1 = *(const f32x4*)(G + (size_t)c1 * D + colofs + 4 * c4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[ch][i] += exact_term(v0[i]) + exact_term(v1[i]);
      }
    }
  }
  if (k < n) {
    const int c0 = list[k];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const int c4 = l16 + 16 * ch;
      if (c4 < W4) {
        const f32x4 v0 = *(const f32x4*)(G + (size_t)c0 * D + colofs + 4 * c4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[ch][i] += exact_term(v0[i]);
      }
    }
  }
}

// Apply one parameter row of width 4*W4 held by lanes l16 (chunks l16, l16+16):
//   g = (float)acc + reg*w;  UPDATE: w -= lr*coef*g.  Returns this lane's partial for part_out.
template <int MODE>
__device__ __forceinline__ double apply_row(float* __restrict__ Wrow, float* __restrict__ Grow,
                                            const double (&acc)[2][4], int W4, int l16, float reg,
                                            float step) {
  double part = 0.0;
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {
    const int c4 = l16 + 16 * ch;
    if (c4 < W4) {
      f32x4 w = *(const f32x4*)(Wrow + 4 * c4);
      if constexpr (MODE == AP_SUMSQ) {
#pragma unroll
        for (int i = 0; i < 4; ++i) part += (double)w[i] * (double)w[i];
      } else {
        f32x4 g;
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = (float)acc[ch][i] + reg * w[i];
        if constexpr (MODE == AP_GRADS) *(f32x4*)(Grow + 4 * c4) = g;
        if constexpr (MODE == AP_ROWNORM) {
#pragma unroll
          for (int i = 0; i < 4; ++i) part += (double)g[i] * (double)g[i];
        }
        if constexpr (MODE == AP_UPDATE) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            w[i] -= step * g[i];
            part += (double)w[i] * (double)w[i];
          }
          *(f32x4*)(Wrow + 4 * c4) = w;
        }
      }
    }
  }
  return part;
}

__device__ __forceinline__ void combine_groups(double (&acc)[2][4]) {
#pragma unroll
  for (int ch = 0; ch < 2; ++ch)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[ch][i] += __shfl_xor(acc[ch][i], 16);
      acc[ch][i] += __shfl_xor(acc[ch][i], 32);
    }
}

// Block layout: [0,nbI) item rows (one row per wavefront), [nbI,nbI+nbU) user rows (user_emb +
// usert_emb), then nbC category rows (one row per workgroup: long lists), then nbD blocks of
// 256 dense parameters.
template <int MODE>
__global__ __launch_bounds__(256) void k_apply_rows(ApplyArgs a) {
  __shared__ double shd[4 * 16 * 8];
  __shared__ float sh_coef;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = lane >> 4, l16 = lane & 15;
  const int D = a.D, S = a.S;
  float coef = 1.0f;
  if constexpr (MODE == AP_UPDATE || MODE == AP_GRADS) {
    if (tid == 0) {
      // global norm (model.py:201).  tf18: per-use rows + (reg*W)^2 + dense; dedup: summed rows.
      double sq = 0.0;
      for (int k = 0; k < a.nsqd; ++k) sq += (double)a.sqd[k];
      const double St = *a.S_total;
      if (a.norm_mode == TLSAN_NORM_TF18)
        sq += (double)a.scal[1] + (double)a.reg * (double)a.reg * St;
      else
        sq += *a.rownorm;
      const float norm = (float)sqrt(sq);
      sh_coef = a.clip / fmaxf(norm, a.clip);
      if (blockIdx.x == 0) {
        if (a.out_gnorm) *a.out_gnorm = norm;
        if (a.out_loss) *a.out_loss = a.scal[0] * a.inv_B + a.reg * (float)(0.5 * St);
      }
    }
    __syncthreads();
    coef = sh_coef;
  }
  const float step = a.lr * coef;
  double part = 0.0;
  const int blk = blockIdx.x;
  if (blk < a.nbI + a.nbU) {
    const bool is_item = blk < a.nbI;
    const int row = is_item ? blk * 4 + wave : (blk - a.nbI) * 4 + wave;
    const int nrows = is_item ? a.I : a.U;
    if (row < nrows) {
      double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      double sacc = 0.0;  // item: bias gradient (lane 0 of each group); user: usert_emb[p = l16]
      int n = 0;
      if constexpr (MODE != AP_SUMSQ) {
        int32_t* cntp = (is_item ? a.cnt_item : a.cnt_user) + row;
        n = *cntp;
        if (n > 0) {
          const int32_t* list = (is_item ? a.list_item + a.off_item[row] : a.list_user + a.off_user[row]);
          accum_list(list, n, grp, 4, a.G, D, 0, a.di / 4, l16, acc);
          for (int k = grp; k < n; k += 4) {
            const int c = list[k];
            const int b = c / S;
            if (is_item) {
              if (l16 == 0 && (c - b * S) == a.Ls + a.Sn) sacc += exact_term(a.dlogit[b]);
            } else if (l16 < a.Ls) {
              sacc += exact_term(a.GT[(size_t)b * a.Ls + l16]);
            }
          }
          combine_groups(acc);
          sacc += __shfl_xor(sacc, 16);
          sacc += __shfl_xor(sacc, 32);
          if constexpr (MODE != AP_ROWNORM) {
            if (lane == 0) *cntp = 0;  // counters are zero at rest
          }
        }
      }
      if (grp == 0) {
        float* W = (is_item ? a.p.item_emb : a.p.user_emb) + (size_t)row * a.di;
        float* Gr = nullptr;
        if constexpr (MODE == AP_GRADS) Gr = (is_item ? a.go.item_emb : a.go.user_emb) + (size_t)row * a.di;
        part += apply_row<MODE>(W, Gr, acc, a.di / 4, l16, a.reg, step);
        // the narrow companions: item_b[row] (not regularised, model.py:164-169) / usert_emb[row]
        if (is_item) {
          if (l16 == 0) {
            const float g = (float)sacc;
            if constexpr (MODE == AP_GRADS) a.go.item_b[row] = g;
            if constexpr (MODE == AP_ROWNORM) part += (double)g * (double)g;
            if constexpr (MODE == AP_UPDATE) {
              if (n > 0) a.p.item_b[row] -= step * g;
            }
          }
        } else if (l16 < a.Ls) {
          float* wp = a.p.usert_emb + (size_t)row * a.Ls + l16;
          float w = *wp;
          const float g = (float)sacc + a.reg * w;
          if constexpr (MODE == AP_SUMSQ) part += (double)w * (double)w;
          if constexpr (MODE == AP_GRADS) a.go.usert_emb[(size_t)row * a.Ls + l16] = g;
          if constexpr (MODE == AP_ROWNORM) part += (double)g * (double)g;
          if constexpr (MODE == AP_UPDATE) {
            w -= step * g;
            *wp = w;
            part += (double)w * (double)w;
          }
        }
      }
    }
  } else if (blk < a.nbI + a.nbU + a.nbC) {
    const int row = blk - a.nbI - a.nbU;
    double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    int n = 0;
    if constexpr (MODE != AP_SUMSQ) {
      n = a.cnt_cate[row];
      if (n > 0) {  // workgroup-uniform
        accum_list(a.list_cate + a.off_cate[row], n, wave * 4 + grp, 16, a.G, D, a.di, a.dc / 4, l16, acc);
        combine_groups(acc);
        if (grp == 0) {
#pragma unroll
          for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int i = 0; i < 4; ++i) shd[((wave * 16 + l16) * 2 + ch) * 4 + i] = acc[ch][i];
        }
        __syncthreads();
        if (wave == 0 && grp == 0) {
#pragma unroll
          for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              double s = 0.0;
              for (int w = 0; w < 4; ++w) s += shd[((w * 16 + l16) * 2 + ch) * 4 + i];
              acc[ch][i] = s;
            }
        }
        __syncthreads();
        if constexpr (MODE != AP_ROWNORM) {
          if (tid == 0) a.cnt_cate[row] = 0;
        }
      }
    }
    if (wave == 0 && grp == 0) {
      float* W = a.p.cate_emb + (size_t)row * a.dc;
      float* Gr = nullptr;
      if constexpr (MODE == AP_GRADS) Gr = a.go.cate_emb + (size_t)row * a.dc;
      part += apply_row<MODE>(W, Gr, acc, a.dc / 4, l16, a.reg, step);
    }
  } else {
    if constexpr (MODE == AP_UPDATE || MODE == AP_GRADS) {
      const int n = (blk - a.nbI - a.nbU - a.nbC) * 256 + tid;
      if (n < a.lay.n_dense) {
        const float g = a.gd[n];
        if constexpr (MODE == AP_GRADS) {
          a.go.dense[n] = g;
        } else {
          const float w = a.p.dense[n] - step * g;
          a.p.dense[n] = w;
          if (n >= a.lay.K && n < a.lay.k0) {
            const int idx = n - a.lay.K;
            a.p.dense_KT[(size_t)(idx % D) * D + idx / D] = w;
          }
        }
      }
    }
    return;
  }
  if constexpr (MODE != AP_GRADS) {
    // per-workgroup partial (fixed order): lanes -> waves -> block
    __syncthreads();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) part += __shfl_xor(part, o);
    if (lane == 0) shd[wave] = part;
    __syncthreads();
    if (tid == 0) a.part_out[blk] = shd[0] + shd[1] + shd[2] + shd[3];
  }
}
