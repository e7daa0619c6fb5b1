// tlsan_rows.h -- generic deterministic scatter-apply on one row table (tlsan_rows_apply):
// the owner-side half of the sharded multi-GPU step.  Same machinery as k_apply_rows:
// integer atomics only decide WHICH contributions belong to a row; the float sum is exact
// (exact_term), so any order gives the same bits.
#pragma once
#include "tlsan_update.h"

struct GIdxArgs {
  const int32_t* dest;
  int32_t n, nrows;
  int32_t* cnt;
  int32_t* cur;
  int32_t* list;
};

template <bool FILL>
__global__ void k_gidx(GIdxArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.n) return;
  const int d = a.dest[t];
  if (d < 0 || d >= a.nrows) return;  // out-of-range destinations are ignored
  if constexpr (!FILL) atomicAdd(&a.cnt[d], 1);
  else a.list[atomicAdd(&a.cur[d], 1)] = t;
}

// The same counting sort (cnt, off, list of a destination array) in ONE launch, counters in the LDS: for the
// per-step category index of the sharded step's compact table (a few 10^4 rows, <= 8192 categories), where the
// four-launch form above (memset + count + scan + fill) costs more in launch latency than in work.  CSR_SMALL_WG
// workgroups, each owning a contiguous range of destination rows: every workgroup reads all of dest (L2-resident),
// counts its own rows in the LDS and, in one register counter, everything that sorts before its range -- that is
// its base offset, so the workgroups need nothing from each other.
#define CSR_SMALL_MAXROWS 8192
#define CSR_SMALL_MAXN (1 << 18)
#define CSR_SMALL_WG 8
__global__ __launch_bounds__(1024) void k_csr_small(GIdxArgs a, int32_t* __restrict__ off) {
  __shared__ int s_cnt[CSR_SMALL_MAXROWS / CSR_SMALL_WG + 1];
  __shared__ int s_part[1024];
  __shared__ int s_below[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int span = (a.nrows + CSR_SMALL_WG - 1) / CSR_SMALL_WG;
  const int r0 = (int)blockIdx.x * span, r1 = min(r0 + span, a.nrows), nr = max(r1 - r0, 0);
  for (int c = tid; c < nr; c += 1024) s_cnt[c] = 0;
  __syncthreads();
  int below = 0;
  for (int t0 = tid; t0 < a.n; t0 += 1024 * 8) {   // 8 loads in flight per thread
    int d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u] = t0 + 1024 * u < a.n ? a.dest[t0 + 1024 * u] : -1;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (d[u] >= r0 && d[u] < r1) atomicAdd(&s_cnt[d[u] - r0], 1);
      below += (d[u] >= 0 && d[u] < r0) ? 1 : 0;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) below += __shfl_xor(below, o);
  if (lane == 0) s_below[wave] = below;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) base += s_below[w];
  const int per = (nr + 1023) / 1024, lo = tid * per;
  int sum = 0;
  for (int c = lo; c < min(lo + per, nr); ++c) sum += s_cnt[c];
  s_part[tid] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {   // inclusive scan of the chunk sums
    const int v = tid >= o ? s_part[tid - o] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  int run = base + s_part[tid] - sum;
  for (int c = lo; c < min(lo + per, nr); ++c) {
    const int n = s_cnt[c];
    a.cnt[r0 + c] = n;
    off[r0 + c] = run;
    s_cnt[c] = run;   // cursor of the fill pass
    run += n;
  }
  __syncthreads();
  for (int t0 = tid; t0 < a.n; t0 += 1024 * 8) {
    int d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u] = t0 + 1024 * u < a.n ? a.dest[t0 + 1024 * u] : -1;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (d[u] >= r0 && d[u] < r1) a.list[atomicAdd(&s_cnt[d[u] - r0], 1)] = t0 + 1024 * u;
  }
}

struct RowsArgs {
  float* W;
  int32_t ld, nrows, width, reg_cols;
  const float* G;
  int32_t ldg;
  const int32_t* cnt;
  const int32_t* off;
  const int32_t* list;
  float gscale;
  const float* step_dev;
  float reg;
  double* part_out;  // [gridDim.x]
};

#define ROWS_NCH 3  // 16 lanes x 3 chunks x 4 floats = 192 columns max

__device__ __forceinline__ void rows_accum(const int32_t* __restrict__ list, int lo, int hi, int stride,
                                           const float* __restrict__ G, int ldg, int W4, int l16,
                                           double (&acc)[ROWS_NCH][4]) {
  for (int k = lo; k < hi; k += 2 * stride) {
    const int c0 = list[k];
    const int c1 = (k + stride < hi) ? list[k + stride] : -1;
#pragma unroll
    for (int ch = 0; ch < ROWS_NCH; ++ch) {
      const int c4 = l16 + 16 * ch;
      if (c4 < W4) {
        const f32x4 v0 = *(const f32x4*)(G + (size_t)c0 * ldg + 4 * c4);
        const f32x4 v1 = c1 >= 0 ? *(const f32x4*)(G + (size_t)c1 * ldg + 4 * c4) : (f32x4)(0.0f);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[ch][i] += exact_term(v0[i]) + exact_term(v1[i]);
      }
    }
  }
}

// 16 rows per workgroup: one row per 16-lane group, long lists shared by the wavefront
__global__ __launch_bounds__(256) void k_rows_apply(RowsArgs a) {
  __shared__ double shd[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = lane >> 4, l16 = lane & 15;
  const int row = blockIdx.x * AP_ROWS_PB + wave * 4 + grp;
  const bool vr = row < a.nrows;
  const int W4 = a.width / 4;
  const float step = *a.step_dev;
  double acc[ROWS_NCH][4];
#pragma unroll
  for (int ch = 0; ch < ROWS_NCH; ++ch)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[ch][i] = 0.0;
  int n = 0, off = 0;
  if (vr) {
    n = a.cnt[row];
    off = a.off[row];
  }
  rows_accum(a.list + off, 0, min(n, AP_OWN), 1, a.G, a.ldg, W4, l16, acc);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int ng = __shfl(n, g * 16);
    if (ng > AP_OWN) {  // wave-uniform
      const int og = __shfl(off, g * 16);
      double t[ROWS_NCH][4];
#pragma unroll
      for (int ch = 0; ch < ROWS_NCH; ++ch)
#pragma unroll
        for (int i = 0; i < 4; ++i) t[ch][i] = 0.0;
      rows_accum(a.list + og, AP_OWN + grp, ng, 4, a.G, a.ldg, W4, l16, t);
#pragma unroll
      for (int ch = 0; ch < ROWS_NCH; ++ch)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          t[ch][i] += __shfl_xor(t[ch][i], 16);
          t[ch][i] += __shfl_xor(t[ch][i], 32);
          if (grp == g) acc[ch][i] += t[ch][i];
        }
    }
  }
  double part = 0.0;
  if (vr) {
    float* Wr = a.W + (size_t)row * a.ld;
#pragma unroll
    for (int ch = 0; ch < ROWS_NCH; ++ch) {
      const int c4 = l16 + 16 * ch;
      if (c4 < W4) {
        f32x4 w = *(const f32x4*)(Wr + 4 * c4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool rg = 4 * c4 + i < a.reg_cols;
          const float g = a.gscale * (float)acc[ch][i] + (rg ? a.reg * w[i] : 0.0f);
          w[i] -= step * g;
          if (rg) part += (double)w[i] * (double)w[i];
        }
        *(f32x4*)(Wr + 4 * c4) = w;
      }
    }
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) part += __shfl_xor(part, o);
  if (lane == 0) shd[wave] = part;
  __syncthreads();
  if (tid == 0 && a.part_out) a.part_out[blockIdx.x] = shd[0] + shd[1] + shd[2] + shd[3];
}
