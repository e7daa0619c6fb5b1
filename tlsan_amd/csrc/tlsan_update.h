// tlsan_update.h -- everything after the fused forward/backward kernel:
//   k_count           uses per destination row (integer atomics)
//   k_index_scan      exclusive scan of the counts = first sorted position of every row; the
//                     fused kernel then draws one position per use from these cursors
//   k_dk_partial      dK = long^T . dbridge  (split over the batch, f32 MFMA)
//   k_dense_finalize  fixed-order reduction of all dense-parameter gradient partials
//   k_apply           one launch for all tables: exact sum of every destination row's contiguous
//                     segment(s) of per-use gradient rows + clip + SGD (model.py:198-205)
// Determinism: integer atomics only build *which* rows belong to a destination; the float
// sums are order-independent (exact_term) or in a fixed order, so two runs are bitwise equal.
#pragma once
#include "tlsan_common.h"
#include <type_traits>

// index slots of the state: the batch being trained and up to two announced successors (tlsan_batch_index)
#define TLSAN_INDEX_SLOTS 3
// first bytes of the persistent state buffer
struct StateHdr {
  // ---- read-mostly line
  float P;                 // scale of the four regularised tables: W_true = P * W_stored (1 unless lazy L2)
  float P_prev;            // P before the current step's commit: what k_apply scales with
  int32_t n_uniq[TLSAN_INDEX_SLOTS][2];    // [index slot][item, user]: rows that received a gradient (k_index_scan)
  double St;               // sum of squares of the four STORED tables (true value: P^2 * St)
  float coef;              // global-norm clip coefficient of the current step (model.py:201)
  uint32_t nstep;          // update steps taken (salt of the stochastic rounding of bf16 tables)
  int32_t spart_n;         // leading records of S_delta the last update may have written
  int32_t n_hot[TLSAN_INDEX_SLOTS];        // [index slot] item rows with more than AP_HOT uses (k_index_scan; listed in the state)
  uint32_t folded;         // the step (nstep) whose S_delta records are already part of St (a step is folded once)
  // the speculative one-pass lazy update (k_finalize_update / k_spec_commit): the step summary leaves the table scale
  // AFTER the step and the step's salt here; P and nstep themselves are committed by the second launch, so that both
  // stay put while the first launch's row workgroups read them
  float P_next;
  uint32_t spec_salt;
  float pad0[13];
  // ---- its own 128-B line: hammered by atomics, must not share a line with anything that is read
  int32_t ticket;          // arrival counter of k_dense_finalize: the last workgroup writes the step summary
  int32_t pad1[31];
};
static_assert(sizeof(StateHdr) == 256, "StateHdr layout");

// A workgroup's change of the stored tables' sum of squares, tagged with the step that made it (StateHdr::nstep after
// that step).  The next step's finalize adds up the records of the previous step, once (StateHdr::folded), and ignores
// older ones: nothing is ever cleared.  (Round 2 kept plain doubles that the finalize cleared as it consumed them;
// tlsan_state_renorm then rescaled a sum that lacked the last step's changes.)
struct DeltaRec {
  double v;
  unsigned long long tag;
};

#ifndef TLSAN_KCH
#define TLSAN_KCH 64     // dK partials a finalize thread has in flight (dense_finalize_block)
#endif
#define FIN_SMALL_PB 64  // small dense parameters per finalize workgroup (dense_finalize_block; the host's nbS)
#ifndef FIN_SMALL_KCH
#define FIN_SMALL_KCH 16 // records a lane has in flight there
#endif
#define AP_HOT 48        // item rows with more uses than this are summed by a workgroup of their own ...
#define AP_HOT_CAP 64    // ... when there are at most this many (the head of a Zipf distribution: a handful)
struct CountArgs {
  int32_t* n_hot;       // reset here for the scan that follows
  tlsan_batch b;
  int32_t Ls;
  int32_t* cnt_item; int32_t* cnt_user; int32_t* cnt_uc;  // persistent, zero at rest
  // CSEG (many categories): every item use also counts into its item's category -- the category half of its gradient row
  // will sit in that category's segment of Gc, next to the u_cate uses (ApplyArgs.cseg)
  const int32_t* item_cate;
  int32_t cseg;
  int32_t ncate;        // categories (rows of cnt_uc)
  int32_t* flag_user;   // optional: [ceil(U / 256)] set where a user row of that 256-row piece is counted (ScanArgs.flag)
  int32_t skip_users;   // != 0: the user side of the index comes from the counting sort of the batch's user ids (IsortArgs): no counts
};

// Use counts per destination row: one thread per (sample, slot); slots [0,Ls) long positions,
// [Ls,Ls+Sn) session positions, Ls+Sn candidate; the first `nbs` blocks take the samples' single uses instead (user
// row + u_cate row), 256 samples each.
// Category rows find their item-side gradients through the segments of their items (see k_apply),
// so only the u_cate uses are counted per category -- through a histogram in the LDS when the table is small:
// atomics on ONE address execute one after the other (~100 ns each across the XCDs), and with few categories (15 in
// Movies-TV: 273 samples per category) the 4096 u_cate counts alone took 30 us of this kernel's 50.
#define COUNT_LDS_CATES 4096
// one thread per sample (b; nthr threads in the block): the user use and the u_cate use
__device__ __forceinline__ void count_samples_block(const CountArgs& a, int* hist, int b, int nthr) {
  const int B = a.b.B;
  if (b == 0 && a.n_hot) *a.n_hot = 0;
  const bool small = a.ncate <= COUNT_LDS_CATES;
  if (small) {
    for (int c = threadIdx.x; c < a.ncate; c += nthr) hist[c] = 0;
    __syncthreads();
  }
  if (b < B) {
    if (!a.skip_users) {
      atomicAdd(&a.cnt_user[a.b.u[b]], 1);
      if (a.flag_user) a.flag_user[a.b.u[b] >> 8] = 1;
    }
    if (small) atomicAdd(&hist[a.b.u_cate[b]], 1);
    else atomicAdd(&a.cnt_uc[a.b.u_cate[b]], 1);
  }
  if (small) {
    __syncthreads();
    for (int c = threadIdx.x; c < a.ncate; c += nthr)
      if (hist[c] != 0) atomicAdd(&a.cnt_uc[c], hist[c]);
  }
}

__global__ __launch_bounds__(256) void k_count(CountArgs a) {
  __shared__ int hist[COUNT_LDS_CATES];
  const int B = a.b.B, Ls = a.Ls, Sn = a.b.Sn, S = Ls + Sn + 1;
  const int nbs = (B + 255) / 256;
  if ((int)blockIdx.x < nbs) {   // ---- one thread per sample: the user use
    count_samples_block(a, hist, blockIdx.x * 256 + threadIdx.x, 256);
    return;
  }
  const int t = (blockIdx.x - nbs) * 256 + threadIdx.x;
  if (t >= B * S) return;
  const int b = t / S, slot = t - b * S;
  if (slot < Ls) {
    if (slot < min(a.b.sl[b], Ls)) {
      const int id = a.b.hist_i[(size_t)b * Ls + slot];
      atomicAdd(&a.cnt_item[id], 1);
      if (a.cseg) atomicAdd(&a.cnt_uc[a.item_cate[id]], 1);
    }
  } else if (slot < Ls + Sn) {
    const int k = slot - Ls;
    if (k < min(a.b.sl_new[b], Sn)) {
      const int id = a.b.hist_i_new[(size_t)b * Sn + k];
      atomicAdd(&a.cnt_item[id], 1);
      if (a.cseg) atomicAdd(&a.cnt_uc[a.item_cate[id]], 1);
    }
  } else {
    const int id = a.b.i[b];
    atomicAdd(&a.cnt_item[id], 1);
    if (a.cseg) atomicAdd(&a.cnt_uc[a.item_cate[id]], 1);
  }
}

// ------------------------------------------------------------------------------------------
// Device-resident batcher: the reference's DataInput.__next__ / DataInputTest.__next__
// (TLSAN/input.py:17-54, 70-107) over a sample set packed as CSR (tlsan_amd/input.py PackedSet).
// One thread per (sample, slot): slots [0,Ls) the long window -- sl = min(len, Ls), the LAST Ls
// items when the history is longer (input.py:41-45), left-aligned otherwise (:47-49), zeros past
// sl -- slots [Ls, Ls+Sn) the current session padded with zeros, slot Ls+Sn the scalars.
// Samples of every category for the u_cate uses (counting sort by category, after the scan): the
// fused kernel then writes those gradient rows in sample order and draws no cursor for them -- with
// few categories (15 in Movies-TV) 4096 returning atomics on 15 addresses cost it 20 us.
// (the cursor draws of a block's 256 samples go through the LDS as well when the table is small: one returning atomic
//  per block and category instead of one per sample -- 13 -> 4 us with 15 categories)
__global__ __launch_bounds__(256) void k_uc_fill(const int32_t* __restrict__ u_cate, int B, int ncate, int32_t* cur_uc, int32_t* uc_list) {
  __shared__ int hist[COUNT_LDS_CATES];
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (ncate > COUNT_LDS_CATES) {
    if (b < B) uc_list[atomicAdd(&cur_uc[u_cate[b]], 1)] = b;
    return;
  }
  for (int c = threadIdx.x; c < ncate; c += 256) hist[c] = 0;
  __syncthreads();
  const int c = b < B ? u_cate[b] : 0;
  const int rank = b < B ? atomicAdd(&hist[c], 1) : 0;
  __syncthreads();
  for (int k = threadIdx.x; k < ncate; k += 256) {
    const int n = hist[k];
    if (n != 0) hist[k] = atomicAdd(&cur_uc[k], n);   // the block's first position in the category
  }
  __syncthreads();
  if (b < B) uc_list[hist[c] + rank] = b;
}

struct PackArgs {
  tlsan_packed set;
  const int32_t* order;  // sample permutation of the epoch (train.py:191 shuffles the list)
  int32_t lo, Ls, is_test;
  tlsan_batch out;
};

__global__ void k_batch_pack(PackArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int B = a.out.B, Sn = a.out.Sn, Ls = a.Ls, S = Ls + Sn + 1;
  if (t >= B * S) return;
  const int b = t / S, slot = t - b * S;
  const int smp = a.order[a.lo + b];
  if (slot < Ls) {
    const int h0 = a.set.hist_off[smp], len = a.set.hist_off[smp + 1] - h0;
    const int sl = min(len, Ls), src = h0 + max(len - Ls, 0) + slot;
    const bool v = slot < sl;
    const_cast<int32_t*>(a.out.hist_i)[(size_t)b * Ls + slot] = v ? a.set.hist[src] : 0;
    const_cast<float*>(a.out.hist_t)[(size_t)b * Ls + slot] = v ? a.set.hist_t[src] : 0.0f;
  } else if (slot < Ls + Sn) {
    const int k = slot - Ls;
    const int s0 = a.set.sess_off[smp], n = a.set.sess_off[smp + 1] - s0;
    const_cast<int32_t*>(a.out.hist_i_new)[(size_t)b * Sn + k] = k < n ? a.set.sess[s0 + k] : 0;
  } else {
    const int len = a.set.hist_off[smp + 1] - a.set.hist_off[smp];
    const_cast<int32_t*>(a.out.u)[b] = a.set.u[smp];
    const_cast<int32_t*>(a.out.u_cate)[b] = a.set.cate[smp];
    const_cast<int32_t*>(a.out.sl)[b] = min(len, Ls);                                   // input.py:31
    const_cast<int32_t*>(a.out.sl_new)[b] = a.set.sess_off[smp + 1] - a.set.sess_off[smp];  // :32
    const_cast<int32_t*>(a.out.i)[b] = a.set.target[smp];
    if (a.is_test) const_cast<int32_t*>(a.out.j)[b] = a.set.second[smp];
    else const_cast<float*>(a.out.y)[b] = (float)a.set.second[smp];
  }
}

// Which samples share a workgroup of the fused kernel (ScanArgs.bal; one extra block of the index scan's launch).
// The kernel's launch is one workgroup of 16 samples per CU and ends with its slowest workgroup; sample costs are
// heavy-tailed (window length when windows are streamed, session length otherwise).  The batch is ranked by that
// cost (stable counting sort, descending) and dealt out in snake order: round j hands one sample to every group,
// walking the groups forwards on even rounds and backwards on odd ones, so every group gets one sample of each
// sixteenth of the ranking.  perm[16 g + j] = sample j of group g (>= B: none).  A fixed function of the batch.
struct BalArgs {
  const int32_t* sl; const int32_t* sl_new;
  int32_t B, Ls, Sn, by_window;
  int32_t blk;        // index of the block that does this (the first one behind the scan's blocks)
  int32_t* perm;      // NULL: no balancing
};
// Windows in registers (by_window == 0; round 5): a workgroup's time is its longest wavefront's -- session steps in P3,
// window positions in P1 / P5 -- and the launch ends with the workgroups that hold one of the batch's few long sessions
// (25 of 4096 sessions have four or more entries at the bench shape: +1.9 us in P3).  Those workgroups are given short
// WINDOWS to make up for it: samples with sessions of two or more rank first (longest first) and are dealt out in snake
// order as above; the others are ranked by window length, shortest first, and handed out in contiguous runs -- group 0,
// which holds the longest session, gets the shortest windows, the last groups get full windows only (which most
// wavefronts have anyway: half the batch's windows are full).
#define BAL_KEYS 192    // by_window: costs 0 .. 96 (TLSAN_LS_CAP); else 11 * min(session, 15) + window for sessions >= 2, 10 - window below
#define BAL_TAIL_KEY 10 // (by_window == 0) the largest key of a sample with a session of one entry or none

template <int NWV = 16>
__device__ __forceinline__ void balance_block(const BalArgs& b) {   // NWV wavefronts
  constexpr int NT = NWV * 64;
  static_assert(NT >= BAL_KEYS, "a thread per cost");
  __shared__ int wcnt[NWV][BAL_KEYS];  // samples of every cost per wavefront -> where the wavefront's first one of that cost ranks
  __shared__ int start[BAL_KEYS];      // rank of the first sample of every cost
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = b.B, G = (B + 15) / 16;
  const int CH = (B + NT - 1) / NT, nper = 64 * CH;   // wavefront w owns samples [w nper, (w + 1) nper), 64 per round
  for (int o = tid; o < NWV * BAL_KEYS; o += NT) (&wcnt[0][0])[o] = 0;
  __syncthreads();
  auto key_of = [&](int i) {
    if (b.by_window > 0) return min(max(min(b.sl[i], b.Ls), 0), BAL_KEYS - 1);
    const int cs = min(max(min(b.sl_new[i], b.Sn), 0), 15), cl = min(max(min(b.sl[i], b.Ls), 0), 10);
    return cs >= 2 ? 11 * cs + cl : BAL_TAIL_KEY - cl;
  };
  for (int c = 0; c < CH; ++c) {
    const int i = wave * nper + c * 64 + lane;
    if (i < B) atomicAdd(&wcnt[wave][key_of(i)], 1);
  }
  __syncthreads();
  if (tid < BAL_KEYS) {
    int run = 0;
    for (int w = 0; w < NWV; ++w) {
      const int x = wcnt[w][tid];
      wcnt[w][tid] = run;
      run += x;
    }
    start[tid] = run;   // (the cost's total, for the moment)
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int k = BAL_KEYS - 1; k >= 0; --k) {   // descending: the costliest samples rank first
      const int x = start[k];
      start[k] = run;
      run += x;
    }
  }
  __syncthreads();
  // rounds dealt out in snake order: all of them (by_window), or those that hold the samples with sessions of two or more
  const int n_heavy = b.by_window > 0 ? 16 * G : start[BAL_TAIL_KEY];   // (rank of the first sample of the tail)
  const int H = min(16, (n_heavy + G - 1) / G);
  // (by_window == 0: the groups are numbered backwards -- the full-window groups, the slowest ones now, get the lowest
  //  block numbers and start first; the launch places its workgroups over 0.6 us.  by_window < 0: not, for A/B)
  const bool rev = b.by_window == 0;
  auto place = [&](int rank, int sample) {
    if (rank < H * G) {
      const int j = rank / G, idx = rank - j * G;
      const int g = (j & 1) ? G - 1 - idx : idx;
      b.perm[(rev ? G - 1 - g : g) * 16 + j] = sample;
    } else {   // the tail, shortest windows first: a contiguous run per group
      const int r2 = rank - H * G, per = 16 - H;
      const int g = r2 / per;
      b.perm[(rev ? G - 1 - g : g) * 16 + H + (r2 - g * per)] = sample;
    }
  };
  volatile int* mine = &wcnt[wave][0];   // (wave-private from here on: DS operations of a wavefront execute in order)
  for (int c = 0; c < CH; ++c) {
    const int i = wave * nper + c * 64 + lane;
    const bool v = i < B;
    const int k = v ? key_of(i) : 255;   // (8 bits; the lanes past the batch form a group of their own)
    // lanes of this round with the same cost: eight ballots, one per bit of the cost
    unsigned long long m = ~0ull;
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const unsigned long long bb = __ballot((k >> bit) & 1);
      m &= ((k >> bit) & 1) ? bb : ~bb;
    }
    const unsigned long long below = m & ((1ull << lane) - 1ull);
    const int kk = v ? k : 0;
    const int base = start[kk] + mine[kk];                 // ranks taken by earlier wavefronts and earlier rounds
    if (v) place(base + __popcll(below), i);               // within a cost and a round: lane order = sample order
    if (v && below == 0ull) mine[k] = mine[k] + __popcll(m);   // the group's first lane counts the round in
  }
  if (B + tid < 16 * G) place(B + tid, B);   // (a last group that is not full)
}

// ------------------------------------------------------------------------------------------
// The USER side of a batch's destination index without a pass over the user table (one more block of the scan's
// launch): a batch holds B user uses (one per sample), so the used rows, their counts and first positions come from a
// sort of the B ids, B <= USORT_MAX, where the counting form reads (and two scan kernels walk) a counter per table row:
// 10 M users are 2442 of a 10 M + 5 M-row index's 3666 scan blocks.  Writes exactly what the scan writes for a table
// with `sparse` set: cur / off of the used rows, their records (ascending), their number, off[U].
// Round 3 sorted with a bitonic network: 78 barrier-separated phases, ~45 us for ONE block -- and a block that sits on a
// CU that long keeps the fused kernel of the next step, which needs every CU's whole LDS, from placing its last
// workgroup (k_fwd_bwd 45 -> 72 us every other step at 10 M users / 5 M items).  Now a bucket sort in eight phases:
// 1024 buckets of consecutive ids (counted, scanned, filled through LDS cursors), then every id ranks itself inside
// its bucket by counting (smaller ids, and equal ones that came earlier in the bucket) -- a handful of compares per id
// for ids that are spread over the table, O(bucket) each if a batch repeats one user thousands of times.
#define USORT_MAX 4096
#define USORT_NB 1024
struct UsortArgs {
  const int32_t* u; int32_t B, U;
  int32_t* cur; int32_t* off; int4* urec; int32_t* n_uniq;
  int32_t blk;        // index of the block that does this; u == NULL: none
};

__device__ __forceinline__ void usort_block(const UsortArgs& a) {   // 1024 threads
  __shared__ int key[USORT_MAX];      // the ids: as they come, later sorted
  __shared__ int srt[USORT_MAX];      // grouped by bucket
  __shared__ int ustart[USORT_MAX];   // bucket counts | cursors, later the first sorted position of every run
  __shared__ int bst[USORT_NB];       // first position of every bucket
  __shared__ int wtot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int CH = USORT_MAX / 1024;
  int shift = 0;
  while (((a.U - 1) >> shift) >= USORT_NB) ++shift;
  int* bcnt = ustart;                 // [USORT_NB] uses per bucket
  int* bcur = ustart + USORT_NB;      // [USORT_NB] fill cursors
  bcnt[tid] = 0;
  int x[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int i = c * 1024 + tid;
    x[c] = i < a.B ? a.u[i] : -1;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < CH; ++c)
    if (x[c] >= 0) atomicAdd(&bcnt[x[c] >> shift], 1);
  __syncthreads();
  {   // exclusive scan of the bucket counts (one per thread)
    const int v = bcnt[tid];
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int pre = inc - v;
#pragma unroll
    for (int w = 0; w < 16; ++w) pre += (w < wave) ? wtot[w] : 0;
    bst[tid] = pre;
    bcur[tid] = pre;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < CH; ++c)
    if (x[c] >= 0) srt[atomicAdd(&bcur[x[c] >> shift], 1)] = x[c];
  __syncthreads();
  // every entry of srt ranks itself inside its bucket
  int dst[CH], val[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int p = c * 1024 + tid;
    dst[c] = -1;
    if (p < a.B) {
      const int v = srt[p], bk = v >> shift, s0 = bst[bk], n = bcnt[bk];
      int rank = 0;
      for (int j = s0; j < s0 + n; ++j) {
        const int y = srt[j];
        rank += (y < v || (y == v && j < p)) ? 1 : 0;
      }
      dst[c] = s0 + rank;
      val[c] = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < CH; ++c)
    if (dst[c] >= 0) key[dst[c]] = val[c];
  __syncthreads();
  // runs of equal ids: a thread takes CH consecutive sorted entries
  const int i0 = tid * CH;
  int mine = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int i = i0 + c;
    mine += (i < a.B && (i == 0 || key[i] != key[i - 1])) ? 1 : 0;
  }
  int inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  int j = inc - mine, nu = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    j += (w < wave) ? wtot[w] : 0;
    nu += wtot[w];
  }
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int i = i0 + c;
    if (i < a.B && (i == 0 || key[i] != key[i - 1])) ustart[j++] = i;
  }
  __syncthreads();
  for (int r = tid; r < nu; r += 1024) {
    const int st = ustart[r], row = key[st], cnt = (r + 1 < nu ? ustart[r + 1] : a.B) - st;
    a.urec[r] = make_int4(row, st, cnt, 0);
    a.cur[row] = st;
    a.off[row] = st;
  }
  if (tid == 0) {
    *a.n_uniq = nu;
    a.off[a.U] = a.B;
  }
}

// ------------------------------------------------------------------------------------------
// The ITEM side of a batch's destination index without a counter per table row (round 4): a partitioned counting sort of
// the batch's item uses (window, session, candidate).  With millions of items the counting form costs a global atomic
// per use on a table that does not fit the caches and two passes over every counter (k_scan_block_sums + k_index_scan:
// 16 + 41 us for 5 M items on an otherwise idle GPU); here the work is proportional to the batch.  The ids are split
// into buckets of 2^shift consecutive ids (at most IS_MAXB buckets of at most IS_BSZ ids: tables up to 16 M rows), and a
// bucket's counters live in the LDS:
//   k_isort_hist      blocks of IS_BLK_SLOTS use slots: LDS histogram over the buckets -> one row of `bh` per block
//                     (its leading blocks take the samples' single uses, as k_count's do)
//   k_isort_scatter   same blocks: a bucket's first position = uses of lower buckets + this bucket's uses in earlier
//                     blocks (column sums of bh, read by every block from the L2); ids scattered into their bucket's
//                     range of `ids` (order inside a bucket: whatever the LDS atomics give -- it does not matter)
//   k_isort_bucket    one block per bucket: counts of its ids in the LDS, scanned -> for every used id its first sorted
//                     position (cur / off, written for used rows only) and its record (id, first, uses) at the
//                     bucket's own range of `tmp`; category segments: uses added to the id's category counter, one
//                     atomic per used ROW instead of one per use
//   isort_finish      (blocks of k_index_scan's launch, beside the category scan) records compacted into urec in id
//                     order, hot rows listed, n_uniq, off[I]
// Writes what the scan writes for a table with `sparse` set -- a fixed function of the batch (records ascending by id;
// the hot list's order is the atomics', as before).  Consumers must reach off / cur through ids or records only: the
// lazy-L2 SGD step with category segments (tlsan_api.hip: build_index).
#define IS_MAXB 2048
#define IS_BSZ 8192
#define IS_BLK_SLOTS 4096
#define ISORT_MAX_SLOTS (1 << 20)
struct IsortArgs {
  int32_t on;               // 0: the item side is counted per row (k_count)
  tlsan_batch b; int32_t Ls;
  int32_t nbu;              // leading blocks of k_isort_hist that take the samples' single uses (1024 samples each)
  int32_t n, shift, nb;     // rows of the table; nb buckets of 2^shift ids
  int32_t nslots, nblk;     // B * (Ls + Sn + 1) use slots in nblk blocks of IS_BLK_SLOTS
  int32_t* bh;              // [nblk][nb] uses per block and bucket
  int32_t* ids;             // [<= nslots] the valid slots' ids, grouped by bucket
  int32_t* bstart;          // [nb + 1] first position of every bucket
  int32_t* nd;              // [nb] used rows of every bucket
  int4* tmp;                // [<= nslots] records, at the bucket's own positions
  int32_t* cur; int32_t* off; int4* urec; int32_t* n_uniq;
  int32_t* hot_n; int32_t* hot_list;                           // rows with more than AP_HOT uses
  const int32_t* item_cate; int32_t* cnt_uc;                   // category segments: uses counted into the row's category
  int32_t blk, nfin;        // k_index_scan's launch: first finishing block, their number (16 buckets each)
};

// item id of use slot t (k_count's enumeration: sample-major; long positions, session positions, the candidate), -1: padding
__device__ __forceinline__ int isort_slot_id(const IsortArgs& a, int t) {
  if (t >= a.nslots) return -1;
  const tlsan_batch& b = a.b;
  const int Ls = a.Ls, Sn = b.Sn, S = Ls + Sn + 1;
  const int smp = t / S, slot = t - smp * S;
  if (slot < Ls) return slot < min(b.sl[smp], Ls) ? b.hist_i[(size_t)smp * Ls + slot] : -1;
  if (slot < Ls + Sn) return (slot - Ls) < min(b.sl_new[smp], Sn) ? b.hist_i_new[(size_t)smp * Sn + (slot - Ls)] : -1;
  return b.i[smp];
}
#define IS_PT (IS_BLK_SLOTS / 1024)   // slots per thread

__global__ __launch_bounds__(1024) void k_isort_hist(IsortArgs a, CountArgs ca) {
  __shared__ int h[COUNT_LDS_CATES > IS_MAXB ? COUNT_LDS_CATES : IS_MAXB];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < a.nbu) {     // the samples' single uses (u_cate row; user row unless sorted)
    count_samples_block(ca, h, blockIdx.x * 1024 + tid, 1024);
    return;
  }
  const int blk = (int)blockIdx.x - a.nbu;
  h[tid] = 0;
  h[tid + 1024] = 0;
  __syncthreads();
  int id[IS_PT];
#pragma unroll
  for (int k = 0; k < IS_PT; ++k) id[k] = isort_slot_id(a, blk * IS_BLK_SLOTS + k * 1024 + tid);
#pragma unroll
  for (int k = 0; k < IS_PT; ++k)
    if (id[k] >= 0) atomicAdd(&h[id[k] >> a.shift], 1);
  __syncthreads();
  for (int c = tid; c < a.nb; c += 1024) a.bh[(size_t)blk * a.nb + c] = h[c];
}

__global__ __launch_bounds__(1024) void k_isort_scatter(IsortArgs a) {
  __shared__ int base[IS_MAXB];
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, blk = blockIdx.x;
  // the ids first: their loads fly while the column sums are formed
  int id[IS_PT];
#pragma unroll
  for (int k = 0; k < IS_PT; ++k) id[k] = isort_slot_id(a, blk * IS_BLK_SLOTS + k * 1024 + tid);
  // thread t owns buckets 2t and 2t + 1 (consecutive: the exclusive scan over the threads' pairs is the scan over buckets)
  int below[2] = {0, 0}, total[2] = {0, 0};
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int c = 2 * tid + e;
    if (c < a.nb) {
      const int32_t* col = a.bh + c;
      for (int j0 = 0; j0 < a.nblk; j0 += 8) {   // 8 loads in flight
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = col[(size_t)min(j0 + u, a.nblk - 1) * a.nb];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int xx = (j0 + u < a.nblk) ? v[u] : 0;
          total[e] += xx;
          below[e] += (j0 + u < blk) ? xx : 0;
        }
      }
    }
  }
  const int tsum = total[0] + total[1];
  int inc = tsum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int start = inc - tsum;
#pragma unroll
  for (int w = 0; w < 16; ++w) start += (w < wave) ? wsum[w] : 0;
  base[2 * tid] = start + below[0];
  base[2 * tid + 1] = start + total[0] + below[1];
  if (blk == 0) {
    if (2 * tid < a.nb) a.bstart[2 * tid] = start;
    if (2 * tid + 1 < a.nb) a.bstart[2 * tid + 1] = start + total[0];
    if (2 * tid == a.nb - 1 || 2 * tid + 1 == a.nb - 1) a.bstart[a.nb] = start + tsum;   // (the last bucket's owner: nothing lies behind it)
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < IS_PT; ++k)
    if (id[k] >= 0) a.ids[atomicAdd(&base[id[k] >> a.shift], 1)] = id[k];
}

// (its FIRST block, when us.u is set, is the user side's sort: independent of everything here and the longest block of
//  the launch -- dispatched first it runs beside the bucket blocks; inside k_index_scan's launch it was the long pole)
__global__ __launch_bounds__(1024) void k_isort_bucket(IsortArgs a, UsortArgs us) {
  const int ub = us.u != nullptr ? 1 : 0;
  if (ub && blockIdx.x == 0) {
    usort_block(us);
    return;
  }
  __shared__ int h[IS_BSZ];
  __shared__ long long wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = (int)blockIdx.x - ub;
  const int lo = a.bstart[b], n = a.bstart[b + 1] - lo;
  if (n == 0) {   // (block-uniform)
    if (tid == 0) a.nd[b] = 0;
    return;
  }
  const int id0 = b << a.shift;
  constexpr int PER = IS_BSZ / 1024;
#pragma unroll
  for (int e = 0; e < PER; ++e) h[e * 1024 + tid] = 0;
  __syncthreads();
  for (int j = tid; j < n; j += 1024) atomicAdd(&h[a.ids[lo + j] - id0], 1);
  __syncthreads();
  int c[PER];
  long long tsum = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    c[k] = h[tid * PER + k];
    tsum += (long long)c[k] + ((long long)(c[k] > 0) << 32);
  }
  long long inc = tsum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const long long t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  long long run = inc - tsum;
#pragma unroll
  for (int w = 0; w < 16; ++w) run += (w < wave) ? wsum[w] : 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    if (c[k] > 0) {
      const int id = id0 + tid * PER + k, first = lo + (int)(run & 0xffffffffLL);
      a.cur[id] = first;
      a.off[id] = first;
      a.tmp[lo + (int)(run >> 32)] = make_int4(id, first, c[k], 0);
      if (a.cnt_uc != nullptr) atomicAdd(&a.cnt_uc[a.item_cate[id]], c[k]);
      run += (long long)c[k] + (1LL << 32);
    }
  }
  if (tid == 1023) a.nd[b] = (int)(run >> 32);
}

// one finishing block (k_index_scan's launch): 16 buckets, one per wavefront -- records to their place in urec
__device__ __forceinline__ void isort_finish_block(const IsortArgs& a, int j) {
  __shared__ int dbase[IS_MAXB];
  __shared__ int wtot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int v0 = 2 * tid < a.nb ? a.nd[2 * tid] : 0, v1 = 2 * tid + 1 < a.nb ? a.nd[2 * tid + 1] : 0;
  int inc = v0 + v1;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  int pre = inc - (v0 + v1), nu = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    pre += (w < wave) ? wtot[w] : 0;
    nu += wtot[w];
  }
  dbase[2 * tid] = pre;
  dbase[2 * tid + 1] = pre + v0;
  __syncthreads();
  const int bb = j * 16 + wave;
  if (bb < a.nb) {
    const int n = a.nd[bb], d0 = dbase[bb];
    const int4* src = a.tmp + a.bstart[bb];
    for (int r = lane; r < n; r += 64) {
      const int4 rec = src[r];
      a.urec[d0 + r] = rec;
      if (rec.z > AP_HOT) {
        const int hh = atomicAdd(a.hot_n, 1);
        if (hh < AP_HOT_CAP) a.hot_list[hh] = d0 + r;
      }
    }
  }
  if (j == 0 && tid == 0) {
    *a.n_uniq = nu;
    a.off[a.n] = a.bstart[a.nb];
  }
}

struct ScanArgs {
  const int32_t* cnt[3];
  int32_t* off[3];
  int32_t* cur[3];
  int32_t n[3];
  int32_t blk0[3];     // first block of each table in the grid
  int32_t* uniq[3];    // optional: ids with cnt > 0, ascending
  int32_t* n_uniq[3];  // optional: how many
  int4* urec[3];       // optional: (id, first position, count) of the ids with cnt > 0 (lazy L2: rows to update)
  int32_t total[3];    // != 0: off has n+1 entries, off[n] = sum of all counts
  long long* bsum;     // optional [gridDim.x]: per-chunk packed sums (k_scan_block_sums) -- large tables
  int32_t* hot_n[3]; int32_t* hot_list[3];   // optional (with urec): slots of the rows with more than AP_HOT uses
  int32_t sparse;      // bit t set: off / cur of table t are written for the rows with cnt > 0 only -- the user table of a
                       // batch's destination index, which is reached through the batch's ids and the used-row records
                       // only (10 M users: 80 MB of writes per step otherwise; the item offsets are walked per category)
  // optional per table: [ceil(n / 256)] marks of the 256-row pieces that hold a count (set by k_count, cleared here: zero
  // at rest).  A wavefront of either scan kernel covers exactly one piece and skips an unmarked one without reading it:
  // 4096 samples mark at most 4096 pieces of a 10 M-row user table's 39 k (40 MB of counters per pass otherwise)
  int32_t* flag[3];
  int32_t* bs_ticket;   // optional (with bsum): arrival counter of k_scan_block_sums, zero at rest -- its last block scans the sums
  BalArgs bal;         // optional (bal.perm): one more block ranks the batch's samples for the fused kernel's workgroups
  UsortArgs us;        // optional (us.u): one more block builds the user side of the index from a sort of the batch's ids
  IsortArgs is;        // optional (is.on): is.nfin more blocks finish the item side built by the k_isort_* launches
};
#define SCAN_TWO_LEVEL_BLOCKS 16  // tables of more chunks than this take the two-launch form

// Exclusive scan of the per-row counts, one launch: block j of a table owns ids
// [4096 j, 4096 j + 4096); it first sums every count that precedes its chunk (coalesced
// re-read of at most n ints from L2: cheaper than a second launch or a serial carry chain),
// then scans its own chunk.  The number of non-zero counts is scanned alongside (high 32 bits
// of a packed 64-bit sum) to compact the list of used rows.
// The re-read is quadratic in the number of chunks, so for large tables (millions of rows) a
// first launch leaves one packed sum per chunk (k_scan_block_sums) and the blocks add up the
// preceding CHUNK sums instead (ScanArgs.bsum).
__global__ __launch_bounds__(1024) void k_scan_block_sums(ScanArgs a) {
  __shared__ long long wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int which = 0;
  if ((int)blockIdx.x >= a.blk0[1]) which = 1;
  if ((int)blockIdx.x >= a.blk0[2]) which = 2;
  const int32_t* __restrict__ cnt = a.cnt[which];
  const int n = a.n[which];
  const int i0 = ((int)blockIdx.x - a.blk0[which]) * 4096 + tid * 4;
  long long part = 0;
  int c4[4] = {0, 0, 0, 0};
  const bool marked = a.flag[which] == nullptr || i0 >= n || a.flag[which][i0 >> 8] != 0;   // (wave-uniform)
  if (!marked) {
  } else if (i0 + 3 < n) {               // (chunks start at multiples of 4096: 16-byte aligned)
    const int4 v = *(const int4*)(cnt + i0);
    c4[0] = v.x; c4[1] = v.y; c4[2] = v.z; c4[3] = v.w;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) c4[k] = (i0 + k < n) ? cnt[i0 + k] : 0;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) part += (long long)c4[k] + ((long long)(c4[k] > 0) << 32);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
  if (lane == 0) wsum[wave] = part;
  __syncthreads();
  __shared__ int s_last;
  if (tid == 0) {
    long long t = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += wsum[w];
    if (a.bs_ticket == nullptr) {
      a.bsum[blockIdx.x] = t;
    } else {   // published with a returning atomic: the last block to arrive reads every sum with device-scope loads
      const unsigned long long old = atomicExch((unsigned long long*)&a.bsum[blockIdx.x], (unsigned long long)t);
      asm volatile("" ::"v"(old));
      s_last = atomicAdd(a.bs_ticket, 1) == (int)gridDim.x - 1;
    }
  }
  if (a.bs_ticket == nullptr) return;
  __syncthreads();
  if (!s_last) return;
  // ---- the last block: the sums of every table's chunks -> exclusive prefixes, in place (k_index_scan then reads ONE
  // value per block where every block used to add up all the sums before its own: 3663 chunks of a 10 M + 5 M-row index,
  // 54 MB of reads)
  __shared__ long long carry;
  for (int t = 0; t < 3; ++t) {
    const int b0 = a.blk0[t], b1 = t < 2 ? a.blk0[t + 1] : (int)gridDim.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int k0 = b0; k0 < b1; k0 += 1024) {
      const int k = k0 + tid;
      const long long v = k < b1 ? __hip_atomic_load(&a.bsum[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      long long inc = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const long long x = __shfl_up(inc, o);
        if (lane >= o) inc += x;
      }
      if (lane == 63) wsum[wave] = inc;
      __syncthreads();
      long long pre = carry + inc - v;
#pragma unroll
      for (int w = 0; w < 16; ++w) pre += (w < wave) ? wsum[w] : 0;
      if (k < b1) a.bsum[k] = pre;
      __syncthreads();
      if (tid == 1023) carry = pre + v;
      __syncthreads();
    }
  }
  if (tid == 0) *a.bs_ticket = 0;   // zero at rest
}

// one scan block of NT threads: chunk `blk` of its table, 4 NT consecutive counts.  (Round 6 also ran the scan of cache-resident
// tables as 256-thread workgroups over chunks of 1024 counts, so that its blocks find a slot beside the row-sum workgroups:
// the kernel itself 16 -> 28 us beside the step, the step equal or 1-2 us slower -- profiles/r06_ab_scan_small.txt; removed.)
template <int NT>
__device__ __forceinline__ void index_scan_block(const ScanArgs& a, int blk) {
  constexpr int CHK = 4 * NT;
  __shared__ long long wsum[NT / 64];
  __shared__ long long prefix;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int which = 0;
  if (blk >= a.blk0[1]) which = 1;
  if (blk >= a.blk0[2]) which = 2;
  const int32_t* __restrict__ cnt = a.cnt[which];
  const int n = a.n[which];
  const int base = (blk - a.blk0[which]) * CHK;
  auto pack = [](int c) { return (long long)c + ((long long)(c > 0) << 32); };
  // ---- packed sum over cnt[0, base)
  long long part = 0;
  if (a.bsum != nullptr && a.bs_ticket != nullptr) {
    part = tid == 0 ? a.bsum[blk] : 0;   // (k_scan_block_sums left the exclusive prefix of this block's table)
  } else if (a.bsum != nullptr) {
    for (int k = a.blk0[which] + tid; k < blk; k += NT) part += a.bsum[k];
  } else {
    for (int k = tid * 4; k < base; k += CHK) {
      const int4 v = *(const int4*)(cnt + k);  // base is a multiple of the chunk -> always in range
      part += pack(v.x) + pack(v.y) + pack(v.z) + pack(v.w);
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
  if (lane == 0) wsum[wave] = part;
  __syncthreads();
  if (tid == 0) {
    long long t = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += wsum[w];
    prefix = t;
  }
  __syncthreads();
  const long long pre = prefix;
  __syncthreads();
  // ---- own chunk
  const int i0 = base + tid * 4;
  int v[4] = {0, 0, 0, 0};
  const bool full = i0 + 3 < n;          // (chunks start at multiples of 4 NT: 16-byte accesses)
  const bool marked = a.flag[which] == nullptr || i0 >= n || a.flag[which][i0 >> 8] != 0;   // (wave-uniform: a wavefront = one 256-row piece)
  if (!marked) {
  } else if (full) {
    const int4 t = *(const int4*)(cnt + i0);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (i0 + k < n) ? cnt[i0 + k] : 0;
  }
  const long long tsum = pack(v[0]) + pack(v[1]) + pack(v[2]) + pack(v[3]);
  long long inc = tsum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const long long t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  long long run = pre + inc - tsum;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) run += (w < wave) ? wsum[w] : 0;
  int32_t* off = a.off[which];
  int32_t* cur = a.cur[which];
  int32_t* uniq = a.uniq[which];
  int4* urec = a.urec[which];
  const bool dense4 = full && !((a.sparse >> which) & 1);   // the four offsets as one 16-byte store each
  if (dense4) {
    const int o0 = (int)(run & 0xffffffffLL);
    const int4 o4 = make_int4(o0, o0 + v[0], o0 + v[0] + v[1], o0 + v[0] + v[1] + v[2]);
    *(int4*)(off + i0) = o4;
    if (cur) *(int4*)(cur + i0) = o4;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (i0 + k < n) {
      const int o = (int)(run & 0xffffffffLL);
      if (!dense4 && (!((a.sparse >> which) & 1) || v[k] > 0)) {
        off[i0 + k] = o;
        if (cur) cur[i0 + k] = o;
      }
      if (uniq && v[k] > 0) uniq[(int)(run >> 32)] = i0 + k;
      if (urec && v[k] > 0) urec[(int)(run >> 32)] = make_int4(i0 + k, o, v[k], 0);
      if (urec && a.hot_n[which] && v[k] > AP_HOT) {
        const int h = atomicAdd(a.hot_n[which], 1);
        if (h < AP_HOT_CAP) a.hot_list[which][h] = (int)(run >> 32);
      }
      run += pack(v[k]);
      if (i0 + k == n - 1) {
        if (a.n_uniq[which]) *a.n_uniq[which] = (int)(run >> 32);
        if (a.total[which]) off[n] = (int)(run & 0xffffffffLL);
      }
    }
  }
  if (a.flag[which] != nullptr && marked && lane == 0 && i0 < n) a.flag[which][i0 >> 8] = 0;   // zero at rest
}

__global__ __launch_bounds__(1024) void k_index_scan(ScanArgs a) {
  if (a.bal.perm != nullptr && (int)blockIdx.x == a.bal.blk) {
    balance_block<16>(a.bal);
    return;
  }
  if (a.us.u != nullptr && (int)blockIdx.x == a.us.blk) {
    usort_block(a.us);
    return;
  }
  if (a.is.on != 0 && (int)blockIdx.x >= a.is.blk) {
    isort_finish_block(a.is, (int)blockIdx.x - a.is.blk);
    return;
  }
  index_scan_block<1024>(a, blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// dK[k][j] = sum_b long[b][k] * dbridge[b][j]  (gradient of tf.layers.dense's kernel,
// model.py:347): C[M=k][N=j], K-dim = samples, f32 MFMA.
// Grid: (D/64)^2 output quadrants of 64x64 x nsplit batch splits; a workgroup is 8 wavefronts and
// wavefront w of split s owns the samples [(8s+w)*spw, +spw).  A wavefront computes a whole 64x64
// quadrant for its samples from two coalesced 16-B loads per MFMA k-step: lane (q, r) reads
// channels 4r..4r+3 of sample s0+q from both operands, and float t of the load feeds MFMA tile t,
// i.e. tile ta x tb covers rows 4m+ta, columns 4n+tb -- 16 independent accumulators per k-step
// and the float4 write-out is contiguous again.  The wavefronts are summed through LDS in a
// fixed order, so a launch leaves nsplit (<= DK_SPLITS_MAX) partial matrices for k_dense_finalize.
#define DK_WAVES 4   // wavefronts per workgroup: 4 -> 64 splits x (D/64)^2 quadrants = 256 workgroups at B = 4096, one
                                 // wavefront per SIMD on every CU (8 left half the chip idle with two wavefronts per SIMD)
#define DK_SPLITS_MAX (256 / DK_WAVES)
// (D = 256: sixteen quadrants per split and 64 KB of LDS per workgroup, two per CU -- at most 32 splits, so that the
//  launch's 512 workgroups are resident at once; with 64 it ran in two rounds: d = 256, Ls = 10 177 -> 175 us/step, bf16 139.6 -> 136.8)
static inline int dk_nsplit(int B, int D) {
  const int cap = D > 128 ? DK_SPLITS_MAX / 2 : DK_SPLITS_MAX;
  const int n = (B + DK_WAVES * 16 - 1) / (DK_WAVES * 16);
  return n < cap ? n : cap;
}
static inline int dk_spw(int B, int D) { const int per = (B + dk_nsplit(B, D) * DK_WAVES - 1) / (dk_nsplit(B, D) * DK_WAVES); return (per + 3) / 4 * 4; }
#define DK_SMEM_BYTES (DK_WAVES * 64 * 64 * 4)
template <int D>
__global__ __launch_bounds__(DK_WAVES * 64) void k_dk_partial(const float* __restrict__ gLong,
                                                              const float* __restrict__ gDB, int B, int spw,
                                                              float* __restrict__ Kp) {
  constexpr int NQ = D / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [DK_WAVES wavefronts][64][64]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, r = lane & 15;
  const int quad = blockIdx.x % (NQ * NQ), split = blockIdx.x / (NQ * NQ);
  const int M0 = (quad / NQ) * 64, N0 = (quad % NQ) * 64;
  const int s_begin = (split * DK_WAVES + wave) * spw, s_end = min(s_begin + spw, B);
  f32x4 acc[4][4];
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = (f32x4)(0.0f);
  for (int s0 = s_begin; s0 < s_end; s0 += 16) {
    f32x4 va[4], vb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // k-step j: samples s0 + 4j + q (clamped address, zeroed past the end)
      const int sm = s0 + 4 * j + q;
      const int sc = sm < s_end ? sm : s_begin;
      va[j] = *(const f32x4*)(gLong + (size_t)sc * D + M0 + 4 * r);
      vb[j] = *(const f32x4*)(gDB + (size_t)sc * D + N0 + 4 * r);
      if (sm >= s_end) va[j] = (f32x4)(0.0f);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = TLSAN_MFMA(va[j][ta], vb[j][tb], acc[ta][tb]);
  }
  // acc[ta][tb][i] = C[M0 + 4 (4q + i) + ta][N0 + 4 r + tb]
  float* W = smem + wave * 4096;
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v;
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) v[tb] = acc[ta][tb][i];
      *(f32x4*)(W + (16 * q + 4 * i + ta) * 64 + 4 * r) = v;
    }
  __syncthreads();
  float* out = Kp + (size_t)split * D * D;
#pragma unroll
  for (int k = 0; k < 1024 / (DK_WAVES * 64); ++k) {
    const int f = tid + DK_WAVES * 64 * k;  // float4 index inside the 64x64 quadrant
    f32x4 v = *(const f32x4*)(smem + 4 * f);
#pragma unroll
    for (int w_ = 1; w_ < DK_WAVES; ++w_) v += *(const f32x4*)(smem + w_ * 4096 + 4 * f);
    *(f32x4*)(out + (size_t)(M0 + f / 16) * D + N0 + 4 * (f % 16)) = v;
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
  float* sqd;             // [nbK + nbS] per-block sum of gd^2
  float* scal;            // [0] = sum of per-sample BCE, [1] = sum of squares of per-use rows
  const DeltaRec* S_delta; // per-workgroup changes of the regularised tables' sum of squares, tagged by step
  int32_t n_spart;
  double* S_total;
  // step summary (written by the last workgroup to arrive)
  StateHdr* hdr;
  float lr, reg, clip, inv_B;
  int32_t norm_mode;
  int32_t commit;          // lazy L2 update: advance the table scale P (P_prev keeps the old value)
  int32_t count_step;      // an update follows (train step, not tlsan_grads): advance hdr->nstep
  int32_t spec;            // speculative one-pass lazy update: neither P nor nstep are touched here (commit = count_step = 0);
                           // the scale after the step and the step's salt go to hdr->P_next / hdr->spec_salt (k_spec_commit)
  float* out_loss; float* out_gnorm; float* out_sq;
};

// The step's scalars, computed once by the last workgroup of k_dense_finalize instead of by every
// workgroup of k_apply: global norm (tf18: per-use rows + (reg*W)^2 + dense; model.py:198-201),
// clip coefficient, loss with the L2 term (model.py:181-196), and the new table scale for lazy L2.
// Dedup-norm mode finishes the coefficient in k_clip_dedup (it needs the per-row sums first).
//
// Hand-over without a device-wide fence (a release fence would write back the whole L2, which
// holds the step's gradient rows): the few scalars other workgroups produced are published with
// returning device-scope atomics (pub_*; the wait for the returned value orders them before the
// ticket) and read back here with device-scope atomic loads.
__device__ __forceinline__ void pub_f32(float* p, float v) {
  const float old = atomicExch(p, v);
  asm volatile("" ::"v"(old));  // wait for the return: the exchange has been performed
}
__device__ __forceinline__ void pub_f64(double* p, double v) {
  const unsigned long long old = atomicExch((unsigned long long*)p, (unsigned long long)__double_as_longlong(v));
  asm volatile("" ::"v"(old));
}
__device__ __forceinline__ float acq_f32(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double acq_f64(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void step_summary(const FinArgs& a, int nsqd, double* shd) {
  const int tid = threadIdx.x;
  // (thread 0's scalars first: their round trip overlaps the partial sums')
  float P = 0.0f, sc0 = 0.0f, sc1 = 0.0f;
  double St0 = 0.0;
  if (tid == 0) {
    P = a.hdr->P;
    St0 = acq_f64(a.S_total);
    sc0 = acq_f32(a.scal + 0);
    sc1 = acq_f32(a.scal + 1);
  }
  double sq = 0.0;
  for (int k = tid; k < nsqd; k += 256) sq += (double)acq_f32(a.sqd + k);
  shd[tid] = sq;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (tid < o) shd[tid] += shd[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    const double St = St0 * (double)P * (double)P;  // true tables = P * stored
    sq = shd[0] + (double)sc1 + (double)a.reg * (double)a.reg * St;
    const float norm = (float)sqrt(sq);
    const float coef = clip_coef(norm, a.clip);
    a.hdr->coef = coef;
    a.hdr->P_prev = P;
    if (a.count_step) a.hdr->nstep += 1;
    if (a.commit) a.hdr->P = P * (1.0f - a.lr * coef * a.reg);
    if (a.spec) {
      a.hdr->P_next = P * (1.0f - a.lr * coef * a.reg);
      a.hdr->spec_salt = a.hdr->nstep + 1;
    }
    if (a.norm_mode == TLSAN_NORM_TF18 && a.out_gnorm) *a.out_gnorm = norm;
    if (a.out_loss) *a.out_loss = sc0 * a.inv_B + a.reg * (float)(0.5 * St);
    if (a.out_sq) *a.out_sq = sc1;
    a.hdr->ticket = 0;
  }
}

// fixed-order sum of doubles by one 256-thread block
__device__ __forceinline__ double block_sum_double(const double* __restrict__ v, int n, double* sh) {
  const int tid = threadIdx.x;
  double s = 0.0;
  for (int k0 = tid; k0 < n; k0 += 256 * 8) {  // 8 loads in flight (clamped addresses, masked sum)
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = v[k0 + 256 * u < n ? k0 + 256 * u : k0];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += k0 + 256 * u < n ? t[u] : 0.0;
  }
  sh[tid] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (tid < o) sh[tid] += sh[tid + o];
    __syncthreads();
  }
  return sh[0];
}

// The apply kernels leave per-workgroup CHANGES of the tables' sum of squares (DeltaRec, tagged by step): add the records
// of the last update to the running sum, once (StateHdr::folded).  One 256-thread workgroup, fixed order.
__device__ __forceinline__ void fold_delta(const DeltaRec* __restrict__ recs, int n_spart, StateHdr* hdr, double* S_total, double* shd) {
  const int tid = threadIdx.x;
  const unsigned long long tag = hdr->nstep;     // (the records of the last update; this step's summary has not run yet)
  // (a step's records are added once: a gradient-only call between two updates finds them folded)
  const int np = hdr->folded == (uint32_t)tag ? 0 : min(n_spart, hdr->spart_n);
  double s = 0.0;
  for (int k0 = tid; k0 < np; k0 += 256 * 8) {     // 8 records in flight (clamped addresses, masked sum), fixed order
    DeltaRec t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = recs[k0 + 256 * u < np ? k0 + 256 * u : k0];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (k0 + 256 * u < np && t[u].tag == tag) ? t[u].v : 0.0;
  }
  shd[tid] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (tid < o) shd[tid] += shd[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    pub_f64(S_total, *S_total + shd[0]);
    hdr->folded = (uint32_t)tag;
  }
}

// (tlsan_state_renorm: the sum of squares must be complete before it is rescaled)
__global__ __launch_bounds__(256) void k_fold_delta(const DeltaRec* recs, int n_spart, StateHdr* hdr, double* S_total) {
  __shared__ double shd[256];
  fold_delta(recs, n_spart, hdr, S_total, shd);
}

// Grid: [0, nbK) blocks reduce the D*D kernel gradient over the batch splits (one thread per
// entry); [nbK, nbK+nbS) blocks reduce the small parameters over the per-pass partial records
// with a lane per parameter (64 per block; a wavefront per quarter of the records, fixed order);
// the last block reduces S_part -> S_total.  Every sum has a fixed order -> deterministic.
template <int D, int DH>
__device__ __forceinline__ void dense_finalize_block(const FinArgs& a, int nbK, int nbS, int blk, double* shd, int* sh_last) {
  using G = Geo<D, DH>;
  constexpr int CW = G::CW, NPB = G::NPB, HPC = CW / DH;  // heads per 16-wide column block
  const int tid = threadIdx.x;
  const tlsan_dense_layout& L = a.lay;
  if (blk == nbK + nbS) {
    // the apply kernels leave per-workgroup CHANGES of the tables' sum of squares: fold them in
    // and clear them (consumed exactly once)
    // (only the entries the last update can have written: a lazy update of a 10^7-row table leaves
    //  a few thousand, not rows / 16)
    fold_delta(a.S_delta, a.n_spart, a.hdr, a.S_total, shd);
    __syncthreads();  // shd is reused below
  }
  float g = 0.0f;
  bool owner = false;
  if (blk == nbK + nbS) {
  } else if (blk < nbK) {
    const int idx = blk * 256 + tid;
    if (idx < D * D) {
      // the partials in chunks of KCH, all loads of a chunk in flight at once (clamped addresses, masked sum), fixed order
      // (64 since round 5: 256 partials in four dependent rounds instead of eight, about 2 us each; the sum's order does
      //  not depend on the chunk.  Measured and not kept, profiles/r05_cate_lists.md: 64 entries per workgroup with a
      //  wavefront per class of partials -- 256 workgroups, one round -- delays the row sums behind them by more than
      //  it gains; 256 B of padding between the partials, against channel conflicts of the 64 KB stride, is slower)
      //  (32 when there are no more partials than that -- k_dk_partial's launches at d = 256: clamped loads are still loads)
      float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, g3 = 0.0f;
      auto chunks = [&](auto kch) {
        constexpr int KCH = decltype(kch)::value;
        for (int s0 = 0; s0 < a.nsplit; s0 += KCH) {
          float v[KCH];
#pragma unroll
          for (int u = 0; u < KCH; ++u)
            v[u] = a.Kp[(size_t)min(s0 + u, a.nsplit - 1) * D * D + idx];
#pragma unroll
          for (int u = 0; u < KCH; u += 4) {
            g0 += s0 + u + 0 < a.nsplit ? v[u + 0] : 0.0f;
            g1 += s0 + u + 1 < a.nsplit ? v[u + 1] : 0.0f;
            g2 += s0 + u + 2 < a.nsplit ? v[u + 2] : 0.0f;
            g3 += s0 + u + 3 < a.nsplit ? v[u + 3] : 0.0f;
          }
        }
      };
      if (a.nsplit > 32) chunks(std::integral_constant<int, TLSAN_KCH>());
      else chunks(std::integral_constant<int, 32>());
      g = (g0 + g1) + (g2 + g3);
      a.gd[L.K + idx] = g;
      owner = true;
    }
  } else if constexpr (D > 128) {
    // d = 256: FIN_SMALL_PB = 64 parameters per workgroup, a lane per parameter: consecutive parameters are consecutive entries of a
    // record, so a wavefront's load touches two lines of ONE record; the four wavefronts take a quarter of the records
    // each, 16 loads in flight, fixed order.  (Until round 6: 16 lanes per parameter, each lane its own records -- every
    // load instruction touched 16 lines 18 KB apart: at d = 256 its 281 workgroups took 8.6 us each and held a third of the
    // row launch's slots for its first 10 us.)
    const int lane = tid & 63, wave = tid >> 6;
    const int m = (blk - nbK) * FIN_SMALL_PB + lane;
    const int n_small = L.n_dense - D * D;
    const bool valid = m < n_small;
    int n = 0, e0 = 0, e1 = 0;
    bool two = false;
    if (valid) {
      n = m < L.K ? m : m + D * D;
      // map the true parameter index to 1..HPC entries of the effective-layout record
      int e[2] = {-1, -1};
      const int wofs[4] = {L.f1_W1, L.f1_W2, L.f2_W1, L.f2_W2};
      const int bofs[4] = {L.f1_b1, L.f1_b2, L.f2_b1, L.f2_b2};
      const int pw[4] = {G::P_F1W1, G::P_F1W2, G::P_F2W1, G::P_F2W2};
      const int pb[4] = {G::P_F1B1, G::P_F1B2, G::P_F2B1, G::P_F2B2};
      for (int mm = 0; mm < 4; ++mm) {
        if (n >= wofs[mm] && n < wofs[mm] + DH * DH) {
          const int k = (n - wofs[mm]) / DH, j = (n - wofs[mm]) % DH;
          for (int h = 0; h < HPC; ++h) e[h] = pw[mm] + (h * DH + k) * CW + h * DH + j;
        }
        if (n >= bofs[mm] && n < bofs[mm] + DH) {
          const int j = n - bofs[mm];
          for (int h = 0; h < HPC; ++h) e[h] = pb[mm] + h * DH + j;
        }
      }
      if (n >= L.k0 && n < L.k0 + D) e[0] = G::P_K0 + (n - L.k0);
      if (n == L.gamma) e[0] = G::P_GAMMA;
      two = HPC > 1 && e[1] >= 0;
      e0 = e[0];
      e1 = two ? e[1] : e[0];
    }
    const int q = (a.nrec + 3) / 4, r_lo = wave * q, r_hi = min(a.nrec, r_lo + q);   // this wavefront's records
    float t = 0.0f;
    if (valid) {
      // (chunks of FIN_SMALL_KCH loads in flight; of 4 when a wavefront has no more records than that -- small batches: clamped loads
      //  are still loads; the sum's order does not depend on the chunk)
      auto chunks = [&](auto kch) {
        constexpr int KCH = decltype(kch)::value;
        for (int r0 = r_lo; r0 < r_hi; r0 += KCH) {
          float v0[KCH], v1[KCH];
#pragma unroll
          for (int u = 0; u < KCH; ++u) {
            const float* p = a.partials + (size_t)(r0 + u < r_hi ? r0 + u : r0) * NPB;
            v0[u] = p[e0];
            if (HPC > 1) v1[u] = p[e1];
          }
#pragma unroll
          for (int u = 0; u < KCH; ++u)
            if (r0 + u < r_hi) t += (HPC > 1 && two) ? v0[u] + v1[u] : v0[u];
        }
      };
      if (q > 4) chunks(std::integral_constant<int, FIN_SMALL_KCH>());
      else chunks(std::integral_constant<int, 4>());
    }
    float* shf = (float*)shd;   // (256 floats of the 256 doubles)
    shf[tid] = t;
    __syncthreads();
    if (wave == 0 && valid) {
      g = (shf[lane] + shf[64 + lane]) + (shf[128 + lane] + shf[192 + lane]);
      a.gd[n] = g;
      owner = true;
    }
    __syncthreads();   // (shd is reused below)
  
  } else {
    // (d <= 128: 16 lanes per parameter, each lane its own records -- 77 / 23 such workgroups, never the launch's long pole;
    //  the form above costs the narrow row kernels two spilled registers)
    const int m = (blk - nbK) * 16 + (tid >> 4), rl = tid & 15;
    const int n_small = L.n_dense - D * D;
    if (m < n_small) {
      const int n = m < L.K ? m : m + D * D;
      // map the true parameter index to 1..HPC entries of the effective-layout record
      int e[2] = {-1, -1};
      const int wofs[4] = {L.f1_W1, L.f1_W2, L.f2_W1, L.f2_W2};
      const int bofs[4] = {L.f1_b1, L.f1_b2, L.f2_b1, L.f2_b2};
      const int pw[4] = {G::P_F1W1, G::P_F1W2, G::P_F2W1, G::P_F2W2};
      const int pb[4] = {G::P_F1B1, G::P_F1B2, G::P_F2B1, G::P_F2B2};
      for (int mm = 0; mm < 4; ++mm) {
        if (n >= wofs[mm] && n < wofs[mm] + DH * DH) {
          const int k = (n - wofs[mm]) / DH, j = (n - wofs[mm]) % DH;
          for (int h = 0; h < HPC; ++h) e[h] = pw[mm] + (h * DH + k) * CW + h * DH + j;
        }
        if (n >= bofs[mm] && n < bofs[mm] + DH) {
          const int j = n - bofs[mm];
          for (int h = 0; h < HPC; ++h) e[h] = pb[mm] + h * DH + j;
        }
      }
      if (n >= L.k0 && n < L.k0 + D) e[0] = G::P_K0 + (n - L.k0);
      if (n == L.gamma) e[0] = G::P_GAMMA;
      float t = 0.0f;
      const bool two = HPC > 1 && e[1] >= 0;
      const int e1 = two ? e[1] : e[0];
      for (int r0 = rl; r0 < a.nrec; r0 += 16 * 8) {  // 8 records in flight per lane, fixed order
        float v0[8], v1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float* p = a.partials + (size_t)(r0 + 16 * u < a.nrec ? r0 + 16 * u : r0) * NPB;
          v0[u] = p[e[0]];
          v1[u] = p[e1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (r0 + 16 * u < a.nrec) t += two ? v0[u] + v1[u] : v0[u];
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) t += __shfl_xor(t, o);
      g = t;
      if (rl == 0) {
        a.gd[n] = g;
        owner = true;
      }
    }
  
  }
  shd[tid] = owner ? (double)g * (double)g : 0.0;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (tid < o) shd[tid] += shd[tid + o];
    __syncthreads();
  }
  if (tid == 0 && blk < nbK + nbS) pub_f32(a.sqd + blk, (float)shd[0]);
  // (in the workgroup that folds the sum-of-squares records: it ends early; the dK entry blocks end the dense chain)
  if (blk == nbK + nbS && tid < 32) {  // loss sum and per-use square sum: 16 lanes each, fixed order
    const int which = tid >> 4, rl = tid & 15;
    float t = 0.0f;
    for (int r0 = rl; r0 < a.nrec; r0 += 16 * 8) {
      float v0[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v0[u] = a.partials[(size_t)(r0 + 16 * u < a.nrec ? r0 + 16 * u : r0) * NPB + G::P_LOSS + which];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r0 + 16 * u < a.nrec) t += v0[u];
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) t += __shfl_xor(t, o);
    if (rl == 0) pub_f32(a.scal + which, t);
  }
  // ---- the last workgroup to arrive sees every other one's results and writes the step summary
  __syncthreads();
  if (tid == 0) *sh_last = atomicAdd(&a.hdr->ticket, 1) == nbK + nbS;  // nbK + nbS + 1 finalize workgroups
  __syncthreads();
  if (*sh_last) step_summary(a, nbK + nbS, shd);
}

template <int D, int DH>
__global__ __launch_bounds__(256) void k_dense_finalize(FinArgs a, int nbK, int nbS) {
  __shared__ double shd[256];
  __shared__ int sh_last;
  dense_finalize_block<D, DH>(a, nbK, nbS, blockIdx.x, shd, &sh_last);
}

// dedup-norm mode: norm^2 = sum over destination rows of |summed row gradient|^2 (ROWNORM pass)
// + dense gradients; overrides the coefficient / norm of the step summary
__global__ __launch_bounds__(256) void k_clip_dedup(const double* rown_part, int nrow, const float* sqd, int nsqd,
                                                    StateHdr* hdr, float clip, float* out_gnorm) {
  __shared__ double shd[256];
  const int tid = threadIdx.x;
  double s = 0.0;
  for (int k = tid; k < nrow; k += 256) s += rown_part[k];
  for (int k = tid; k < nsqd; k += 256) s += (double)sqd[k];
  shd[tid] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (tid < o) shd[tid] += shd[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    const float norm = (float)sqrt(shd[0]);
    hdr->coef = clip_coef(norm, clip);
    if (out_gnorm) *out_gnorm = norm;
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
enum { AP_UPDATE = 0, AP_GRADS = 1, AP_SUMSQ = 2, AP_ROWNORM = 3,
       AP_PRESUM = 4 };  // PRESUM: only the exact per-row sums, left in Rc / Ri / Rb / Ru for k_update_lazy

struct ApplyArgs {
  tlsan_params p;
  tlsan_grads_out go;
  tlsan_dense_layout lay;
  int32_t I, U, C, Ls, D, di, dc, WU;
  const float* Gi; const float* Gb; const float* Gu; const float* Gc;
  int32_t* cnt_item; int32_t* cnt_user; int32_t* cnt_uc;
  const int32_t* off_item; const int32_t* off_user; const int32_t* off_uc;   // n+1 entries each
  const int4* urec_item; const int4* urec_user;   // lazy L2: (row, first position, uses) of the rows used this step
  const int32_t* cate_off; const int32_t* cate_cnt; const int32_t* cate_items;  // static CSR
  const int32_t* uc_list;  // optional: samples of every category (segments off_uc); then Gc is in sample order
  int32_t cseg;            // != 0 (many categories): a category's segment of Gc holds its u_cate uses AND the category halves
                           // of its items' uses (k_fwd_bwd, FwdArgs.cseg): the category blocks sum that one segment and
                           // do not walk the category's items
  const float* gd;
  float* Rc; float* Ri; float* Rb; float* Ru;   // PRESUM -> k_update_lazy: summed rows [C][dc], [slot][di], [slot], [slot][WU]
  int32_t presum_rows;     // PRESUM: write the item / user sums to the rows of `go` instead (tlsan_grads with reg = 0)
  // PRESUM with few, large categories (Movies-TV: 15): csplit > 1 workgroups share a category (each takes
  // every csplit-th pass of cpass items and its share of the u_cate uses) and add their exact partial
  // sums into Rc64 with double atomics --
  // sums of 2^-40-grid values are exact in any order, so the result stays bitwise reproducible;
  // k_update_lazy rounds them to float (as a single workgroup would have) and clears them
  int32_t csplit, cpass;
  // cpos != 0 (categories of at most 256 items -- one pass): the sharing workgroups all scan the category's items and
  // each takes an equal slice of the concatenated USE POSITIONS instead of every csplit-th group of items: a hot item no
  // longer makes its share the launch's longest chain (Digital-Music, batch 2048: one share 14 us, the rest 7)
  int32_t cpos;
  // hot item rows (more than AP_HOT uses) get a workgroup each in the row-sum pass: nbH = AP_HOT_CAP such
  // workgroups lead the grid, the item-row workgroups leave those rows to them (when the list did not overflow)
  const int32_t* hot_n; const int32_t* hot_list; int32_t nbH;
  double* Rc64;            // [C][dc], zero at rest (state)
  double* part_out;        // SUMSQ: sum of squares per workgroup; ROWNORM: sum g^2
  DeltaRec* delta_out;     // UPDATE: change of the stored tables' sum of squares per workgroup, tagged with the step
  StateHdr* hdr;           // P, P_prev, coef (read); spart_n (written by an update)
  const int32_t* n_uniq_item; const int32_t* n_uniq_user;   // used-row counts of this step's index slot
  float lr, reg;
  int32_t nbI, nbU, nbC, nbD;
  // k_finalize_update: item-row workgroups LAUNCHED (0: nbI).  The host sizes nbI for the most rows the batch can touch
  // (C5: 25.8 k blocks of 16 rows; 4.4 k are real, the rest start, find nothing and leave -- 5 us of the launch's slots);
  // with nbI_l < nbI a workgroup takes the blocks nbI_l apart until the used rows end
  int32_t nbI_l;
  int32_t ufirst;          // k_finalize_update: the user-row workgroups lead the item-row workgroups
  // optimizers other than SGD (dense UPDATE only): accumulator tables shaped like p, see tlsan_optimizer
  tlsan_params s1, s2;
  int32_t opt;
  float ob1, ob2, oeps, oalpha;   // oalpha: Adam's lr * sqrt(1 - beta2^t) / (1 - beta1^t)
  unsigned long long* stamps;  // debug: 8 s_memtime stamps per workgroup (tlsan_debug_stamps)
};

#define AP_OWN 8        // uses a 16-lane group sums alone before the wavefront helps
#define AP_ROWS_PB 16   // item / user rows per workgroup (4 wavefronts x 4 groups)

// exact sum of rows lo, lo+stride, ... < hi of a [.., ld] buffer (columns 4*c4..), 4 in flight
template <int NCH>
__device__ __forceinline__ void seg_accum(const float* __restrict__ Gs, int ld, int lo, int hi, int stride,
                                          int W4, int l16, double (&acc)[NCH][4]) {
  for (int k = lo; k < hi; k += 4 * stride) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c4 = l16 + 16 * ch;
      if (c4 < W4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[u] = (k + u * stride < hi) ? *(const f32x4*)(Gs + (size_t)(k + u * stride) * ld + 4 * c4) : (f32x4)(0.0f);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[ch][i] += exact_term(v[u][i]);
      }
    }
  }
}

template <int NCH>
__device__ __forceinline__ void zero_acc(double (&acc)[NCH][4]) {
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[ch][i] = 0.0;
}

// sum over the four 16-lane groups of a wavefront (exact doubles -> order irrelevant)
template <int NCH>
__device__ __forceinline__ void combine_groups(double (&acc)[NCH][4]) {
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[ch][i] += __shfl_xor(acc[ch][i], 16);
      acc[ch][i] += __shfl_xor(acc[ch][i], 32);
    }
}

// accum: add to the record this step's first launch left (the correcting pass of a speculative update, k_spec_commit)
__device__ __forceinline__ void block_delta_store(double part, double* shd, DeltaRec* dst, unsigned long long tag, bool accum = false) {
  __syncthreads();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) part += __shfl_xor(part, o);
  if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double v = shd[0] + shd[1] + shd[2] + shd[3];
    dst->v = (accum && dst->tag == tag) ? dst->v + v : v;
    dst->tag = tag;
  }
}

__device__ __forceinline__ void block_part_store(double part, double* shd, double* dst) {
  __syncthreads();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) part += __shfl_xor(part, o);
  if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) *dst = shd[0] + shd[1] + shd[2] + shd[3];
}

// One parameter element of a regularised table under the update of model.py:198-205.
// Stored value w (true parameter P*w), exact gradient sum gs of the TRUE parameter.
//   dense : w <- w - step * (gs / P + reg * w)            (every row, every step)
//   lazy  : w <- w - (step / P_new) * gs, P_new = P (1 - step reg), P committed once per step
// Returns the gradient (GRADS / ROWNORM) and accumulates the block partial.
template <int MODE, bool LAZY>
__device__ __forceinline__ float apply_elem(float& w, float gs, float P, float invP, float reg, float step,
                                            float lazy_scale, double& part) {
  const float g = gs + reg * (P * w);  // gradient of the true parameter
  if constexpr (MODE == AP_SUMSQ) part += (double)w * (double)w;
  if constexpr (MODE == AP_ROWNORM) part += (double)g * (double)g;
  if constexpr (MODE == AP_UPDATE) {
    const float w0 = w;
    if constexpr (LAZY) w = w0 - lazy_scale * gs;
    else w = w0 - step * (gs * invP + reg * w0);
    part += (double)w * (double)w - (double)w0 * (double)w0;
  }
  return g;
}

// One element under TF 1.8's Adam / RMSProp / Adadelta (training_ops.cc; the Sparse* forms are the
// same arithmetic per row).  g: clipped gradient of the parameter; s1, s2: its two accumulators.
struct OptCtx {
  int opt;
  float lr, b1, b2, eps, alpha;
};
__device__ __forceinline__ void opt_elem(const OptCtx& o, float& w, float g, float& s1, float& s2) {
  if (o.opt == TLSAN_OPT_ADAM) {
    s1 = s1 * o.b1 + g * (1.0f - o.b1);
    s2 = s2 * o.b2 + (g * g) * (1.0f - o.b2);
    w -= o.alpha * s1 / (sqrtf(s2) + o.eps);
  } else if (o.opt == TLSAN_OPT_RMSPROP) {  // b1 = decay, b2 = momentum
    s1 = s1 * o.b1 + (g * g) * (1.0f - o.b1);
    s2 = s2 * o.b2 + o.lr * g / sqrtf(s1 + o.eps);
    w -= s2;
  } else {  // Adadelta, b1 = rho
    s1 = s1 * o.b1 + (g * g) * (1.0f - o.b1);
    const float upd = sqrtf(s2 + o.eps) / sqrtf(s1 + o.eps) * g;
    w -= upd * o.lr;
    s2 = s2 * o.b1 + (upd * upd) * (1.0f - o.b1);
  }
}

// ------------------------------------------------------------------------------------------
// k_apply: ONE launch applies the step to every embedding table and the dense parameters.
// Block layout: [0,nbC) one category row per workgroup, then nbI blocks of item rows and nbU
// blocks of user rows (user_emb + usert_emb; one row per 16-lane group), then nbD blocks of 256
// dense parameters.  Category blocks come first: they have the longest dependent chain.
//
// Item uses are stored destination-sorted as rows [item half | cate half] of Gi: the item
// blocks sum the item halves of their row's contiguous segment, the category blocks sum the
// cate halves of the segments of all items of the category (static CSR of item_cate) plus the
// category's u_cate uses (segment of Gc).  Nothing is passed between blocks, so rows and
// categories need no second launch.  Every sum is exact (exact_term) -> order-free, bitwise
// reproducible.
//
// The kernel is a chain of dependent memory round trips, so every load that does not depend
// on another is issued up front: used-row records, clip-norm partials, the parameter row, and
// the first AP_OWN gradient rows of a segment in one batch (clamped addresses instead of
// branches, which the compiler would serialise).
// LAZY: the row blocks walk the compacted records of used rows (k_index_scan) instead of every row.
// NCH = float4 chunks per lane: 16 lanes x NCH x 4 floats >= the widest row (d_item, WU, d_cate).
#define AP_CAP 2048  // LDS list of use positions of one category pass

// exact sum of the rows listed in sh_pos[0, T): >= 0 -> cate half of Gi[pos], < 0 -> Gc[~pos];
// the 16 groups of the workgroup stride over the list, AP_OWN rows in flight per group
template <int NCH>
__device__ __forceinline__ void list_accum(const ApplyArgs& a, const int* sh_pos, int T, int gid, int l16, int W4,
                                           double (&acc)[NCH][4]) {
  for (int k = gid; k < T; k += 16 * AP_OWN) {
    f32x4 v[AP_OWN][NCH];
#pragma unroll
    for (int u = 0; u < AP_OWN; ++u) {
      const int kk = k + 16 * u;
      const int pos = sh_pos[kk < T ? kk : k];
      const float* src = pos >= 0 ? a.Gi + (size_t)pos * a.D + a.di : a.Gc + (size_t)(~pos) * a.dc;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int c4 = l16 + 16 * ch;
        if (c4 < W4) v[u][ch] = *(const f32x4*)(src + 4 * c4);
      }
    }
#pragma unroll
    for (int u = 0; u < AP_OWN; ++u) {
      if (k + 16 * u < T) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          if (l16 + 16 * ch < W4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[ch][i] += exact_term(v[u][ch][i]);
          }
        }
      }
    }
  }
}

struct ApCtx {
  int tid, wave, lane, grp, l16, gid, blk;
  float P, invP, step, lazy_scale;
  uint32_t salt;      // per-step salt of the stochastic rounding (bf16 tables)
  float coef;         // clip coefficient (optimizers other than SGD)
  OptCtx oc;
  bool accum = false; // UPDATE: add the block's change of the sum of squares to its record of this step (k_spec_commit)
};

#define AP_STAMP(k)                                                                      \
  do {                                                                                   \
    if (a.stamps != nullptr && x.tid == 0) a.stamps[(size_t)x.blk * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)

// ================= one category row per workgroup =================
template <int MODE, bool LAZY, int NCH, int DT, bool CSPLIT = false>
__device__ __forceinline__ void apply_cate_block(const ApplyArgs& a, const ApCtx& x, double* shd, double* shp,
                                                 int* sh_pos, int* sh_lo, int* sh_n, int* sh_wtot) {
  constexpr bool RESET = MODE == AP_UPDATE || MODE == AP_GRADS || MODE == AP_PRESUM;  // counters are zero at rest
  const int tid = x.tid, wave = x.wave, lane = x.lane, grp = x.grp, l16 = x.l16, gid = x.gid;
  int c = x.blk, split = 0, nsplit = 1;
  if constexpr (MODE == AP_PRESUM && CSPLIT) {  // (a compile-time variant: the common single-workgroup case pays nothing)
    nsplit = a.csplit; c = x.blk % a.C; split = x.blk / a.C;
  }
  const int W4 = a.dc / 4;
  const size_t wrow = (size_t)c * a.dc;  // element index of the row in cate_emb
  f32x4 w[NCH];
  if constexpr (MODE != AP_PRESUM) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
      if (l16 + 16 * ch < W4) w[ch] = tbl_ld4<DT>(a.p.cate_emb, wrow + 4 * (l16 + 16 * ch));
  }
  double acc[NCH][4];
  zero_acc(acc);
  double part = 0.0;
  int nu = 0;
  if constexpr (MODE != AP_SUMSQ) {
    // (CSEG: nothing to walk -- the category halves of the items' uses sit in this category's segment of Gc, with the
    //  u_cate uses: `nu` below counts both)
    const int i0 = a.cseg ? 0 : a.cate_off[c], ni = a.cseg ? 0 : a.cate_cnt[c];
    int ou = a.off_uc[c];
    const int nu_all = a.off_uc[c + 1] - ou;
    nu = nu_all;
    int PS = 256;  // items per pass
    bool by_pos = false;     // (CSPLIT) shares are slices of the use positions, not groups of items
    if constexpr (CSPLIT) {  // this workgroup's share of the u_cate uses and its pass size
      by_pos = a.cpos != 0;
      if (!by_pos) PS = a.cpass;
      const int chunk = (nu_all + nsplit - 1) / nsplit;
      ou += split * chunk;
      nu = max(0, min(chunk, nu_all - split * chunk));
    }
    const int pstart = by_pos ? 0 : split * PS, pstep = by_pos ? PS : nsplit * PS;
    bool first = true;
    for (int p0 = pstart; first || p0 < ni; p0 += pstep, first = false) {
      int lo = 0, n = 0;
      if (tid < PS && p0 + tid < ni) {
        const int item = a.cate_items[i0 + p0 + tid];
        lo = a.off_item[item];
        n = a.off_item[item + 1] - lo;
      }
      int inc = n;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
      }
      if (lane == 63) sh_wtot[wave] = inc;
      __syncthreads();
      int pre = inc - n;
#pragma unroll
      for (int w_ = 0; w_ < 4; ++w_) pre += (w_ < wave) ? sh_wtot[w_] : 0;
      int T = (sh_wtot[0] + sh_wtot[1]) + (sh_wtot[2] + sh_wtot[3]);
      const bool last = p0 + pstep >= ni;  // this workgroup's last pass
      const int extra = last ? nu : 0;  // the u_cate uses ride along with the last pass
      // by_pos: this workgroup's slice [s_lo, s_hi) of the pass's T concatenated use positions
      int s_lo = 0, s_hi = T;
      if (by_pos) {
        s_lo = (int)((long long)T * split / nsplit);
        s_hi = (int)((long long)T * (split + 1) / nsplit);
        T = s_hi - s_lo;
      }
      if (T + extra <= AP_CAP) {
        if constexpr (CSPLIT) {
          for (int j = max(0, s_lo - pre); j < min(n, s_hi - pre); ++j) sh_pos[pre + j - s_lo] = lo + j;
        } else {   // (the whole pass: no slice arithmetic in the registers of the one-workgroup-per-category form)
          for (int j = 0; j < n; ++j) sh_pos[pre + j] = lo + j;
        }
        for (int j = tid; j < extra; j += 256) sh_pos[T + j] = ~(a.uc_list ? a.uc_list[ou + j] : ou + j);
        __syncthreads();
        AP_STAMP(1);
        list_accum<NCH>(a, sh_pos, T + extra, gid, l16, W4, acc);
        AP_STAMP(2);
      } else {  // very hot category: segment after segment, the 16 groups striding over each
        sh_lo[tid] = lo;
        sh_n[tid] = n;
        __syncthreads();
        const int cnt = max(0, min(PS, ni - p0));
        int run = 0;   // (by_pos) concatenated position of the segment's first use
        for (int t = 0; t < cnt; ++t) {
          const int nt = sh_n[t], lt = sh_lo[t];
          if constexpr (CSPLIT) {
            const int o_lo = max(run, s_lo), o_hi = min(run + nt, s_hi);   // the part of the segment inside the slice
            if (o_hi > o_lo) seg_accum<NCH>(a.Gi + a.di, a.D, lt + (o_lo - run) + gid, lt + (o_hi - run), 16, W4, l16, acc);
            run += nt;
          } else {
            if (nt > 0) seg_accum<NCH>(a.Gi + a.di, a.D, lt + gid, lt + nt, 16, W4, l16, acc);
          }
        }
        if (last) {
          if (a.uc_list == nullptr) {
            seg_accum<NCH>(a.Gc, a.dc, ou + gid, ou + nu, 16, W4, l16, acc);
          } else {  // (rows in sample order: through the category's sample list)
            for (int k = ou + gid; k < ou + nu; k += 16) {
              const float* src = a.Gc + (size_t)a.uc_list[k] * a.dc;
#pragma unroll
              for (int ch = 0; ch < NCH; ++ch)
                if (l16 + 16 * ch < W4) {
                  const f32x4 v = *(const f32x4*)(src + 4 * (l16 + 16 * ch));
#pragma unroll
                  for (int i = 0; i < 4; ++i) acc[ch][i] += exact_term(v[i]);
                }
            }
          }
        }
      }
      __syncthreads();  // sh_pos / sh_wtot are rewritten by the next pass
    }
    combine_groups(acc);
    if (grp == 0) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int i = 0; i < 4; ++i) shd[((wave * 16 + l16) * NCH + ch) * 4 + i] = acc[ch][i];
    }
    __syncthreads();
    if (wave == 0 && grp == 0) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          double s = 0.0;
          for (int w_ = 0; w_ < 4; ++w_) s += shd[((w_ * 16 + l16) * NCH + ch) * 4 + i];
          acc[ch][i] = s;
        }
    }
  }
  AP_STAMP(3);
  if constexpr (MODE == AP_PRESUM) {
    if (wave == 0 && grp == 0) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int c4 = l16 + 16 * ch;
        if (c4 < W4) {
          if constexpr (CSPLIT) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (acc[ch][i] != 0.0) unsafeAtomicAdd(a.Rc64 + wrow + 4 * c4 + i, acc[ch][i]);  // (hardware f64 add, no CAS loop)
          } else {
            f32x4 g;
#pragma unroll
            for (int i = 0; i < 4; ++i) g[i] = (float)acc[ch][i];
            *(f32x4*)(a.Rc + wrow + 4 * c4) = g;
          }
        }
      }
    }
    if (tid == 0 && split == 0 && nu > 0) a.cnt_uc[c] = 0;
    return;
  }
  if (wave == 0 && grp == 0) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c4 = l16 + 16 * ch;
      if (c4 < W4) {
        f32x4 g;
        const f32x4 w0 = w[ch];
        double pe = 0.0;  // (UPDATE: the change of the sum of squares is taken from the values actually stored)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float wi = w[ch][i];
          g[i] = apply_elem<MODE, LAZY>(wi, (float)acc[ch][i], x.P, x.invP, a.reg, x.step, x.lazy_scale, pe);
          w[ch][i] = wi;
        }
        if constexpr (MODE == AP_GRADS) *(f32x4*)(a.go.cate_emb + (size_t)c * a.dc + 4 * c4) = g;
        if constexpr (MODE == AP_UPDATE && !LAZY) {
          if (a.opt != TLSAN_OPT_SGD) {
            f32x4 m1 = *(const f32x4*)(a.s1.cate_emb + wrow + 4 * c4), m2 = *(const f32x4*)(a.s2.cate_emb + wrow + 4 * c4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float wi = w0[i], a1 = m1[i], a2 = m2[i];
              opt_elem(x.oc, wi, x.coef * g[i], a1, a2);
              w[ch][i] = wi; m1[i] = a1; m2[i] = a2;
            }
            *(f32x4*)(a.s1.cate_emb + wrow + 4 * c4) = m1;
            *(f32x4*)(a.s2.cate_emb + wrow + 4 * c4) = m2;
          }
        }
        if constexpr (MODE == AP_UPDATE) {
          tbl_st4<DT>(a.p.cate_emb, wrow + 4 * c4, w[ch], x.salt ^ 0x3c6ef372u);
#pragma unroll
          for (int i = 0; i < 4; ++i) part += (double)w[ch][i] * (double)w[ch][i] - (double)w0[i] * (double)w0[i];
        } else {
          part += pe;
        }
      }
    }
  }
  if constexpr (RESET) {
    if (tid == 0 && nu > 0) a.cnt_uc[c] = 0;
  }
  if constexpr (MODE == AP_UPDATE) block_delta_store(part, shp, &a.delta_out[x.blk], x.salt, x.accum);
  else if constexpr (MODE != AP_GRADS) block_part_store(part, shp, &a.part_out[x.blk]);
}

// ================= 16 item rows or 16 user rows per workgroup (one row per 16-lane group) =========
// NCH float4 chunks per lane cover the row (item rows: the item half only), OWN gradient rows are
// in flight per group; longer segments are finished by the whole wavefront.
// C0: first 16-byte chunk per lane this call covers (a row wider than 16 lanes x NCH chunks is covered by two calls:
// the row sums of 154-float user rows -- d = 128 with 90-entry windows -- without the wide variant's registers)
template <int MODE, bool LAZY, bool IS_ITEM, int NCH, int OWN, int DT, int C0 = 0>
__device__ __forceinline__ void apply_rows_block(const ApplyArgs& a, const ApCtx& x, int slot0, double* shp) {
  constexpr bool RESET = MODE == AP_UPDATE || MODE == AP_GRADS || MODE == AP_PRESUM;
  const int lane = x.lane, grp = x.grp, l16 = x.l16;
  const int slot = slot0 + x.gid;
  int row = 0, off = 0, n = 0;
  bool vr;
  double part = 0.0;
  if constexpr (LAZY) {
    const int nuq = IS_ITEM ? *a.n_uniq_item : *a.n_uniq_user;
    if (slot0 >= nuq) return;  // (workgroup-uniform) nothing left: lazy rows past the used ones leave no partial
    const int4 r = (IS_ITEM ? a.urec_item : a.urec_user)[slot];  // (row, first position, uses)
    vr = slot < nuq;
    if (vr) { row = r.x; off = r.y; n = r.z; }
    if constexpr ((MODE == AP_PRESUM || MODE == AP_UPDATE) && IS_ITEM) {
      if (a.nbH > 0 && n > AP_HOT && *a.hot_n <= AP_HOT_CAP) { vr = false; n = 0; }   // a hot-row workgroup sums (speculative one-pass update: updates) it
    }
  } else {
    vr = slot < (IS_ITEM ? a.I : a.U);
    if (vr) row = slot;
    if constexpr (MODE != AP_SUMSQ) {
      const int32_t* o = IS_ITEM ? a.off_item : a.off_user;
      off = o[row];
      n = vr ? o[row + 1] - off : 0;
    }
  }
  AP_STAMP(1);
  const float* Gs = IS_ITEM ? a.Gi : a.Gu;
  const int ld = IS_ITEM ? a.D : a.WU;
  const int W4 = (IS_ITEM ? a.di : a.WU) / 4;
  // ---- the parameter row
  float* Wtab = IS_ITEM ? a.p.item_emb : a.p.user_emb;
  const size_t wrow = IS_ITEM ? (size_t)row * a.p.ld_item : (size_t)row * a.p.ld_user;  // element index of the row
  float* Trow = a.p.usert_emb + (size_t)row * a.p.ld_usert;
  f32x4 w[NCH];
  float wb = 0.0f;
  if constexpr (MODE != AP_PRESUM) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int cc = 4 * (l16 + 16 * (ch + C0));
      if (cc < a.di) {
        w[ch] = tbl_ld4<DT>(Wtab, wrow + cc);
      } else if (!IS_ITEM) {
#pragma unroll
        for (int i = 0; i < 4; ++i) w[ch][i] = (cc + i - a.di < a.Ls) ? Trow[cc + i - a.di] : 0.0f;
      }
    }
    if (IS_ITEM && l16 == 0) wb = a.p.item_b[(size_t)row * a.p.ld_itemb];
  }
  // ---- exact sum of the row's segment
  double acc[NCH][4];
  zero_acc(acc);
  double bacc = 0.0;  // item_b gradient of the row (item rows)
  if constexpr (MODE != AP_SUMSQ) {
    const int n_own = min(n, OWN);
    {
      f32x4 v[OWN][NCH];
      const int last = max(n_own - 1, 0);
#pragma unroll
      for (int u = 0; u < OWN; ++u) {
        const float* src = Gs + (size_t)(off + min(u, last)) * ld;  // (buffers carry a pad row)
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
          if (l16 + 16 * (ch + C0) < W4) v[u][ch] = *(const f32x4*)(src + 4 * (l16 + 16 * (ch + C0)));
      }
      float gb = 0.0f;
      if (IS_ITEM) gb = a.Gb[off + min(l16, last)];
#pragma unroll
      for (int u = 0; u < OWN; ++u) {
        if (u < n_own) {
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch)
            if (l16 + 16 * (ch + C0) < W4) {
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[ch][i] += exact_term(v[u][ch][i]);
            }
        }
      }
      if (IS_ITEM && l16 < n_own) bacc = exact_term(gb);
    }
    AP_STAMP(2);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ng = __shfl(n, g * 16);
      if (ng > OWN) {  // wave-uniform: the four groups split the rest of group g's segment
        const int og = __shfl(off, g * 16);
        double t[NCH][4];
        zero_acc(t);
        seg_accum<NCH>(Gs + 64 * C0, ld, og + OWN + grp, og + ng, 4, W4 - 16 * C0, l16, t);
        double tb = 0.0;
        if (IS_ITEM)
          for (int k = og + OWN + lane; k < og + ng; k += 64) tb += exact_term(a.Gb[k]);
        combine_groups(t);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) tb += __shfl_xor(tb, o);
        if (grp == g) {
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[ch][i] += t[ch][i];
          if (l16 == 0) bacc += tb;
        }
      }
    }
    if (IS_ITEM) {  // fold the group's 16 partial bias sums (exact doubles: any order)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) bacc += __shfl_xor(bacc, o);
    }
  }
  AP_STAMP(3);
  if constexpr (MODE == AP_PRESUM) {
    if (vr) {
      const bool by_row = a.presum_rows != 0;  // (workgroup-uniform)
      float* R = IS_ITEM ? a.Ri + (size_t)slot * a.di : a.Ru + (size_t)slot * a.WU;
      if (by_row) R = IS_ITEM ? a.go.item_emb + (size_t)row * a.go.ld_item : a.go.user_emb + (size_t)row * a.go.ld_user;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int c4 = l16 + 16 * (ch + C0);
        if (c4 < W4) {
          f32x4 g;
#pragma unroll
          for (int i = 0; i < 4; ++i) g[i] = (float)acc[ch][i];
          if (!by_row || 4 * c4 < a.di) {
            *(f32x4*)(R + 4 * c4) = g;
          } else {  // usert_emb columns of a user row
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int p = 4 * c4 + i - a.di;
              if (p < a.Ls) a.go.usert_emb[(size_t)row * a.go.ld_usert + p] = g[i];
            }
          }
        }
      }
      if (l16 == 0) {
        if (IS_ITEM) (by_row ? a.go.item_b[(size_t)row * a.go.ld_itemb] : a.Rb[slot]) = (float)bacc;
        if (n > 0) (IS_ITEM ? a.cnt_item : a.cnt_user)[row] = 0;
      }
      if (a.presum_rows == 2) {
        // fused rows [item_emb | item_b | pad] / [user_emb | usert_emb | pad] of one width (the sharded step):
        // a row is written by exactly one of the two views -- clear the rest of it, so that the caller's
        // buffer need not be zeroed
        const int first = IS_ITEM ? a.di + 1 : a.di + a.Ls, width = IS_ITEM ? a.go.ld_item : a.go.ld_user;
        for (int c = first + l16; c < width; c += 16) R[c] = 0.0f;
      }
    }
    return;
  }
  if (vr) {
    // column cc of the row: cc < di -> item_emb / user_emb;  user rows, di <= cc < di+Ls -> usert_emb
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int cc = 4 * (l16 + 16 * (ch + C0));
      if (cc >= 4 * W4) continue;
      f32x4 g;
      if (cc < a.di) {
        const f32x4 w0 = w[ch];
        double pe = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float wi = w[ch][i];
          g[i] = apply_elem<MODE, LAZY>(wi, (float)acc[ch][i], x.P, x.invP, a.reg, x.step, x.lazy_scale, pe);
          w[ch][i] = wi;
        }
        if constexpr (MODE == AP_UPDATE && !LAZY) {
          if (a.opt != TLSAN_OPT_SGD) {
            float* S1 = IS_ITEM ? a.s1.item_emb + (size_t)row * a.s1.ld_item : a.s1.user_emb + (size_t)row * a.s1.ld_user;
            float* S2 = IS_ITEM ? a.s2.item_emb + (size_t)row * a.s2.ld_item : a.s2.user_emb + (size_t)row * a.s2.ld_user;
            f32x4 m1 = *(const f32x4*)(S1 + cc), m2 = *(const f32x4*)(S2 + cc);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float wi = w0[i], a1 = m1[i], a2 = m2[i];
              opt_elem(x.oc, wi, x.coef * g[i], a1, a2);
              w[ch][i] = wi; m1[i] = a1; m2[i] = a2;
            }
            *(f32x4*)(S1 + cc) = m1;
            *(f32x4*)(S2 + cc) = m2;
          }
        }
        if constexpr (MODE == AP_GRADS) {
          if (!a.go.sparse || n > 0)
            *(f32x4*)((IS_ITEM ? a.go.item_emb + (size_t)row * a.go.ld_item : a.go.user_emb + (size_t)row * a.go.ld_user) + cc) = g;
        }
        if constexpr (MODE == AP_UPDATE) {
          tbl_st4<DT>(Wtab, wrow + cc, w[ch], x.salt ^ (IS_ITEM ? 0x85ebca6bu : 0xc2b2ae35u));
#pragma unroll
          for (int i = 0; i < 4; ++i) part += (double)w[ch][i] * (double)w[ch][i] - (double)w0[i] * (double)w0[i];
        } else {
          part += pe;
        }
      } else if (!IS_ITEM) {  // usert_emb columns (scalar: Ls need not be a multiple of 4)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int p = cc + i - a.di;
          if (p < a.Ls) {
            float wi = w[ch][i];
            const float w00 = wi;
            const float gg = apply_elem<MODE, LAZY>(wi, (float)acc[ch][i], x.P, x.invP, a.reg, x.step, x.lazy_scale, part);
            if constexpr (MODE == AP_GRADS) {
              if (!a.go.sparse || n > 0) a.go.usert_emb[(size_t)row * a.go.ld_usert + p] = gg;
            }
            if constexpr (MODE == AP_UPDATE && !LAZY) {
              if (a.opt != TLSAN_OPT_SGD) {
                float* q1 = a.s1.usert_emb + (size_t)row * a.s1.ld_usert + p;
                float* q2 = a.s2.usert_emb + (size_t)row * a.s2.ld_usert + p;
                float a1 = *q1, a2 = *q2;
                part -= (double)wi * (double)wi;
                wi = w00;
                opt_elem(x.oc, wi, x.coef * gg, a1, a2);
                part += (double)wi * (double)wi;
                *q1 = a1; *q2 = a2;
              }
            }
            if constexpr (MODE == AP_UPDATE) Trow[p] = wi;
          }
        }
      }
    }
    if (IS_ITEM && l16 == 0) {  // item_b[row]: not regularised (model.py:164-169), never scaled
      const float g = (float)bacc;
      if constexpr (MODE == AP_GRADS) {
        if (!a.go.sparse || n > 0) a.go.item_b[(size_t)row * a.go.ld_itemb] = g;
      }
      if constexpr (MODE == AP_ROWNORM) part += (double)g * (double)g;
      if constexpr (MODE == AP_UPDATE) {
        bool sgd = true;
        if constexpr (!LAZY) sgd = a.opt == TLSAN_OPT_SGD;
        if (sgd) {
          if (n > 0) a.p.item_b[(size_t)row * a.p.ld_itemb] = wb - x.step * g;
        } else if (g != 0.0f || a.opt == TLSAN_OPT_ADAM) {
          // sparse Adam decays m, v of every row; the sparse RMSProp / Adadelta kernels touch the rows the
          // candidates gathered -- recognised by their non-zero gradient (a candidate whose sigmoid
          // saturates to exactly y has gradient 0 and is skipped here, where TF would still decay its slots)
          float* q1 = a.s1.item_b + (size_t)row * a.s1.ld_itemb;
          float* q2 = a.s2.item_b + (size_t)row * a.s2.ld_itemb;
          float a1 = *q1, a2 = *q2, wi = wb;
          opt_elem(x.oc, wi, x.coef * g, a1, a2);
          a.p.item_b[(size_t)row * a.p.ld_itemb] = wi;
          *q1 = a1; *q2 = a2;
        }
      }
    }
    if constexpr (RESET) {
      if (n > 0 && l16 == 0) (IS_ITEM ? a.cnt_item : a.cnt_user)[row] = 0;
    }
  }
  if constexpr (MODE == AP_UPDATE) block_delta_store(part, shp, &a.delta_out[x.blk], x.salt, x.accum);
  else if constexpr (MODE != AP_GRADS) block_part_store(part, shp, &a.part_out[x.blk]);
}

// ================= CSEG: 16 category rows per workgroup (one per 16-lane group) =================
// With ApplyArgs.cseg a category's gradient is ONE contiguous segment of Gc (its u_cate uses and the category halves of
// its items' uses), so category rows are summed like item and user rows -- a 16-lane group per row, OWN rows in flight,
// the wavefront finishing long segments -- instead of by a workgroup each (10 k categories: 10 k workgroups of a
// few uses, each with the list machinery of apply_cate_block).  Block b handles categories [16 b, 16 b + 16).
template <int MODE, bool LAZY, int NCH, int OWN, int DT>
__device__ __forceinline__ void apply_cseg_block(const ApplyArgs& a, const ApCtx& x, int c0, double* shp) {
  constexpr bool RESET = MODE == AP_UPDATE || MODE == AP_GRADS || MODE == AP_PRESUM;
  const int lane = x.lane, grp = x.grp, l16 = x.l16;
  const int c = c0 + x.gid;
  const bool vr = c < a.C;
  const int cc = vr ? c : 0;
  const int W4 = a.dc / 4;
  int off = 0, n = 0;
  if constexpr (MODE != AP_SUMSQ) {
    off = a.off_uc[cc];
    n = vr ? a.off_uc[cc + 1] - off : 0;
  }
  const size_t wrow = (size_t)cc * a.dc;
  f32x4 w[NCH];
  if constexpr (MODE != AP_PRESUM) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
      if (l16 + 16 * ch < W4) w[ch] = tbl_ld4<DT>(a.p.cate_emb, wrow + 4 * (l16 + 16 * ch));
  }
  double acc[NCH][4];
  zero_acc(acc);
  double part = 0.0;
  if constexpr (MODE != AP_SUMSQ) {
    const int n_own = min(n, OWN);
    {
      f32x4 v[OWN][NCH];
      const int last = max(n_own - 1, 0);
#pragma unroll
      for (int u = 0; u < OWN; ++u) {
        const float* src = a.Gc + (size_t)(off + min(u, last)) * a.dc;  // (the buffer carries a pad row)
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
          if (l16 + 16 * ch < W4) v[u][ch] = *(const f32x4*)(src + 4 * (l16 + 16 * ch));
      }
#pragma unroll
      for (int u = 0; u < OWN; ++u)
        if (u < n_own) {
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch)
            if (l16 + 16 * ch < W4) {
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[ch][i] += exact_term(v[u][ch][i]);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ng = __shfl(n, g * 16);
      if (ng > OWN) {  // wave-uniform: the four groups split the rest of group g's segment
        const int og = __shfl(off, g * 16);
        double t[NCH][4];
        zero_acc(t);
        seg_accum<NCH>(a.Gc, a.dc, og + OWN + grp, og + ng, 4, W4, l16, t);
        combine_groups(t);
        if (grp == g) {
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[ch][i] += t[ch][i];
        }
      }
    }
  }
  if constexpr (MODE == AP_PRESUM) {
    if (vr) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int c4 = l16 + 16 * ch;
        if (c4 < W4) {
          f32x4 g;
#pragma unroll
          for (int i = 0; i < 4; ++i) g[i] = (float)acc[ch][i];
          *(f32x4*)(a.Rc + wrow + 4 * c4) = g;
        }
      }
      if (l16 == 0 && n > 0) a.cnt_uc[c] = 0;
    }
    return;
  }
  if (vr) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c4 = l16 + 16 * ch;
      if (c4 >= W4) continue;
      f32x4 g;
      const f32x4 w0 = w[ch];
      double pe = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float wi = w[ch][i];
        g[i] = apply_elem<MODE, LAZY>(wi, (float)acc[ch][i], x.P, x.invP, a.reg, x.step, x.lazy_scale, pe);
        w[ch][i] = wi;
      }
      if constexpr (MODE == AP_GRADS) *(f32x4*)(a.go.cate_emb + (size_t)c * a.dc + 4 * c4) = g;
      if constexpr (MODE == AP_UPDATE && !LAZY) {
        if (a.opt != TLSAN_OPT_SGD) {
          f32x4 m1 = *(const f32x4*)(a.s1.cate_emb + wrow + 4 * c4), m2 = *(const f32x4*)(a.s2.cate_emb + wrow + 4 * c4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float wi = w0[i], a1 = m1[i], a2 = m2[i];
            opt_elem(x.oc, wi, x.coef * g[i], a1, a2);
            w[ch][i] = wi; m1[i] = a1; m2[i] = a2;
          }
          *(f32x4*)(a.s1.cate_emb + wrow + 4 * c4) = m1;
          *(f32x4*)(a.s2.cate_emb + wrow + 4 * c4) = m2;
        }
      }
      if constexpr (MODE == AP_UPDATE) {
        tbl_st4<DT>(a.p.cate_emb, wrow + 4 * c4, w[ch], x.salt ^ 0x3c6ef372u);
#pragma unroll
        for (int i = 0; i < 4; ++i) part += (double)w[ch][i] * (double)w[ch][i] - (double)w0[i] * (double)w0[i];
      } else {
        part += pe;
      }
    }
    if constexpr (RESET) {
      if (l16 == 0 && n > 0) a.cnt_uc[c] = 0;
    }
  }
  if constexpr (MODE == AP_UPDATE) block_delta_store(part, shp, &a.delta_out[x.blk], x.salt, x.accum);
  else if constexpr (MODE != AP_GRADS) block_part_store(part, shp, &a.part_out[x.blk]);
}

// ================= one hot item row per workgroup (PRESUM) =================
// UPD (the speculative one-pass update, k_finalize_update / k_spec_commit): the workgroup updates the row itself --
// w -= x.lazy_scale * sum, item_b -= x.step * sum_b -- and leaves its change of the sum of squares in record
// nbC + nbI + nbU + h of S_delta (hot workgroups own the records behind the row blocks').
template <int NCH, bool UPD = false, int DT = TLSAN_TABLE_F32>
__device__ __forceinline__ void presum_hot_block(const ApplyArgs& a, int h, double* shd, double* shp, const ApCtx* xp = nullptr) {
  const int nh = *a.hot_n;
  if (nh > AP_HOT_CAP || h >= nh) return;  // (workgroup-uniform) list overflowed: the item-row workgroups kept the rows
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = lane >> 4, l16 = lane & 15, gid = wave * 4 + grp;
  const int slot = a.hot_list[h];
  const int4 r = a.urec_item[slot];
  const int row = r.x, off = r.y, n = r.z;
  const int W4 = a.di / 4;
  f32x4 w_row[NCH];
  float wb_row = 0.0f;
  if constexpr (UPD) {   // (the parameter row, requested with everything else)
    if (wave == 0 && grp == 0) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch)
        if (l16 + 16 * ch < W4) w_row[ch] = tbl_ld4<DT>(a.p.item_emb, (size_t)row * a.p.ld_item + 4 * (l16 + 16 * ch));
      if (l16 == 0) wb_row = a.p.item_b[(size_t)row * a.p.ld_itemb];
    }
  }
  double acc[NCH][4];
  zero_acc(acc);
  for (int k = off + gid; k < off + n; k += 16 * AP_OWN) {
    f32x4 v[AP_OWN][NCH];
#pragma unroll
    for (int u = 0; u < AP_OWN; ++u) {
      const int kk = k + 16 * u < off + n ? k + 16 * u : k;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch)
        if (l16 + 16 * ch < W4) v[u][ch] = *(const f32x4*)(a.Gi + (size_t)kk * a.D + 4 * (l16 + 16 * ch));
    }
#pragma unroll
    for (int u = 0; u < AP_OWN; ++u)
      if (k + 16 * u < off + n) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
          if (l16 + 16 * ch < W4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[ch][i] += exact_term(v[u][ch][i]);
          }
      }
  }
  double tb = 0.0;
  for (int k = off + tid; k < off + n; k += 256) tb += exact_term(a.Gb[k]);
  combine_groups(acc);
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) tb += __shfl_xor(tb, o);
  if (grp == 0) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int i = 0; i < 4; ++i) shd[((wave * 16 + l16) * NCH + ch) * 4 + i] = acc[ch][i];
  }
  if (lane == 0) shp[wave] = tb;
  __syncthreads();
  if constexpr (UPD) {
    const ApCtx& x = *xp;
    double part = 0.0;
    if (wave == 0 && grp == 0) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const int c4 = l16 + 16 * ch;
        if (c4 < W4) {
          const f32x4 w0 = w_row[ch];
          f32x4 wn;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            double s_ = 0.0;
            for (int w_ = 0; w_ < 4; ++w_) s_ += shd[((w_ * 16 + l16) * NCH + ch) * 4 + i];
            wn[i] = w0[i] - x.lazy_scale * (float)s_;      // (apply_elem<AP_UPDATE, lazy>)
          }
          tbl_st4<DT>(a.p.item_emb, (size_t)row * a.p.ld_item + 4 * c4, wn, x.salt ^ 0x85ebca6bu);
#pragma unroll
          for (int i = 0; i < 4; ++i) part += (double)wn[i] * (double)wn[i] - (double)w0[i] * (double)w0[i];
        }
      }
      if (l16 == 0) {
        const float gb = (float)((shp[0] + shp[1]) + (shp[2] + shp[3]));
        a.p.item_b[(size_t)row * a.p.ld_itemb] = wb_row - x.step * gb;    // not regularised, never scaled
        a.cnt_item[row] = 0;
      }
    }
    block_delta_store(part, shp, &a.delta_out[a.nbC + a.nbI + a.nbU + h], x.salt, x.accum);
    return;
  }
  if (wave == 0 && grp == 0) {
    const bool by_row = a.presum_rows != 0;
    float* R = by_row ? a.go.item_emb + (size_t)row * a.go.ld_item : a.Ri + (size_t)slot * a.di;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c4 = l16 + 16 * ch;
      if (c4 < W4) {
        f32x4 g;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          double s = 0.0;
          for (int w_ = 0; w_ < 4; ++w_) s += shd[((w_ * 16 + l16) * NCH + ch) * 4 + i];
          g[i] = (float)s;
        }
        *(f32x4*)(R + 4 * c4) = g;
      }
    }
    if (l16 == 0) {
      const float gb = (float)((shp[0] + shp[1]) + (shp[2] + shp[3]));
      (by_row ? a.go.item_b[(size_t)row * a.go.ld_itemb] : a.Rb[slot]) = gb;
      a.cnt_item[row] = 0;
    }
    if (a.presum_rows == 2)
      for (int c = a.di + 1 + l16; c < a.go.ld_item; c += 16) R[c] = 0.0f;
  }
}

// WIDE: d_item / d_cate above 64 or d_item + Ls above 128 columns (more float4 chunks per lane)
template <int MODE, bool LAZY, bool WIDE, int DT = TLSAN_TABLE_F32>
__global__ __launch_bounds__(256) void k_apply(ApplyArgs a) {
  constexpr int NC = WIDE ? 2 : 1, NI = WIDE ? 2 : 1, NU = WIDE ? 4 : 2;
  __shared__ double shd[4 * 16 * NC * 4];
  __shared__ double shp[4];
  __shared__ int sh_pos[AP_CAP];
  __shared__ int sh_lo[256], sh_n[256];
  __shared__ int sh_wtot[4];
  ApCtx x;
  x.tid = threadIdx.x; x.wave = x.tid >> 6; x.lane = x.tid & 63; x.grp = x.lane >> 4; x.l16 = x.lane & 15;
  x.gid = x.wave * 4 + x.grp;  // 16 groups
  x.blk = blockIdx.x;
  AP_STAMP(0);
  if (a.stamps != nullptr && x.tid == 0) a.stamps[(size_t)x.blk * 8 + 4] = __builtin_amdgcn_s_memrealtime();
  // (the step summary of k_dense_finalize already advanced hdr->P for a lazy update)
  x.P = (MODE == AP_UPDATE && LAZY) ? a.hdr->P_prev : a.hdr->P;
  x.invP = 1.0f / x.P;
  x.step = MODE == AP_UPDATE ? a.lr * a.hdr->coef : 0.0f;
  x.lazy_scale = x.step / (x.P * (1.0f - x.step * a.reg));
  x.salt = a.hdr->nstep;
  x.coef = MODE == AP_UPDATE ? a.hdr->coef : 0.0f;
  if (MODE == AP_UPDATE && x.blk == 0 && x.tid == 0) a.hdr->spart_n = a.nbC + a.nbI + a.nbU;
  x.oc.opt = a.opt; x.oc.lr = a.lr; x.oc.b1 = a.ob1; x.oc.b2 = a.ob2; x.oc.eps = a.oeps; x.oc.alpha = a.oalpha;
  const int blk = x.blk;
  if (blk < a.nbC) {
    if (a.cseg) apply_cseg_block<MODE, LAZY, NC, AP_OWN, DT>(a, x, blk * AP_ROWS_PB, shp);   // (nbC = ceil(C / 16) then)
    else apply_cate_block<MODE, LAZY, NC, DT>(a, x, shd, shp, sh_pos, sh_lo, sh_n, sh_wtot);
  } else if (blk < a.nbC + a.nbI) {
    apply_rows_block<MODE, LAZY, true, NI, AP_OWN, DT>(a, x, (blk - a.nbC) * AP_ROWS_PB, shp);
  } else if (blk < a.nbC + a.nbI + a.nbU) {
    apply_rows_block<MODE, LAZY, false, NU, AP_OWN / 2, DT>(a, x, (blk - a.nbC - a.nbI) * AP_ROWS_PB, shp);
  } else {
    // ================= 256 dense parameters =================
    const int nd = (blk - a.nbC - a.nbI - a.nbU) * 256 + x.tid;
    if constexpr (MODE == AP_UPDATE || MODE == AP_GRADS) {
      if (nd < a.lay.n_dense) {
        const float g = a.gd[nd];
        float w0 = 0.0f;
        if constexpr (MODE == AP_UPDATE) w0 = a.p.dense[nd];
        if constexpr (MODE == AP_GRADS) {
          a.go.dense[nd] = g;
        } else {
          float wn = w0 - x.step * g;
          if constexpr (!LAZY) {
            if (a.opt != TLSAN_OPT_SGD) {
              float a1 = a.s1.dense[nd], a2 = a.s2.dense[nd];
              wn = w0;
              opt_elem(x.oc, wn, x.coef * g, a1, a2);
              a.s1.dense[nd] = a1; a.s2.dense[nd] = a2;
            }
          }
          a.p.dense[nd] = wn;
          if (nd >= a.lay.K && nd < a.lay.k0) {
            const int idx = nd - a.lay.K;
            a.p.dense_KT[(size_t)(idx % a.D) * a.D + idx / a.D] = wn;
          }
        }
      }
    }
  }
  AP_STAMP(6);
  // (slots 4/5: the device-wide 100 MHz clock at start/end; s_memtime is not synchronised across the chip)
  if (a.stamps != nullptr && x.tid == 0) a.stamps[(size_t)x.blk * 8 + 5] = __builtin_amdgcn_s_memrealtime();
}
#undef AP_STAMP

// ------------------------------------------------------------------------------------------
// The lazy-L2 train step splits the apply pass so that the row sums (which need neither the clip
// coefficient nor the dense gradients) overlap the dense finalize instead of waiting for it:
//   k_finalize_presum : workgroups [0, nbK+nbS] are k_dense_finalize's, the rest are k_apply's in
//                       PRESUM mode (exact per-row sums -> Rc / Ri / Rb / Ru, counters reset)
//   k_update_lazy     : elementwise w -= scale * sum for the used rows + the dense parameters
// Same arithmetic per element as k_apply<AP_UPDATE, lazy> (the sums are rounded to float there too).
// (the narrow form is held to 96 registers -- five workgroups per CU: left alone, the compiler takes 124 for the 64 loads the
//  dK entry blocks keep in flight and costs the launch a fifth of its residency; held, it needs 91 and spills nothing)
#ifndef PRESUM_WPE_WIDE
#define PRESUM_WPE_WIDE 4     // (128 registers, a few spilled in the wide row roles; three -- 138, nothing spilled -- measured slower: d = 256, Ls = 90 244.8 vs 238.5 us/step)
#endif
template <int D, int DH, bool WIDE, bool CSPLIT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WIDE ? PRESUM_WPE_WIDE : 5))) void k_finalize_presum(FinArgs f, int nbK, int nbS, ApplyArgs a) {
  constexpr int NC = WIDE ? 2 : 1, NI = WIDE ? 2 : 1, NU = WIDE ? 4 : 2;
  __shared__ double shd[4 * 16 * NC * 4 > 256 ? 4 * 16 * NC * 4 : 256];
  __shared__ double shp[4];
  __shared__ int sh_pos[AP_CAP];
  __shared__ int sh_lo[256], sh_n[256];
  __shared__ int sh_wtot[4];
  __shared__ int sh_last;
  const int nfin = nbK + nbS + 1;
  if ((int)blockIdx.x < nfin) {
    // (debug stamps: the finalize workgroups are listed after the apply workgroups)
    unsigned long long* stp = a.stamps ? a.stamps + (size_t)(gridDim.x - nfin + blockIdx.x) * 8 : nullptr;
    if (stp && threadIdx.x == 0) { stp[0] = __builtin_amdgcn_s_memtime(); stp[4] = __builtin_amdgcn_s_memrealtime(); }
    dense_finalize_block<D, DH>(f, nbK, nbS, blockIdx.x, shd, &sh_last);
    if (stp && threadIdx.x == 0) { stp[6] = __builtin_amdgcn_s_memtime(); stp[5] = __builtin_amdgcn_s_memrealtime(); }
    return;
  }
  ApCtx x;
  x.tid = threadIdx.x; x.wave = x.tid >> 6; x.lane = x.tid & 63; x.grp = x.lane >> 4; x.l16 = x.lane & 15;
  x.gid = x.wave * 4 + x.grp;
  x.blk = blockIdx.x - nfin;
  x.P = 1.0f; x.invP = 1.0f; x.step = 0.0f; x.lazy_scale = 0.0f; x.salt = 0u;
  if (x.blk < a.nbH) {   // hot item rows lead the grid (no debug stamps)
    presum_hot_block<NI>(a, x.blk, shd, shp);
    return;
  }
  x.blk -= a.nbH;
  unsigned long long* stp = a.stamps ? a.stamps + (size_t)x.blk * 8 : nullptr;
  if (stp && x.tid == 0) { stp[0] = __builtin_amdgcn_s_memtime(); stp[4] = __builtin_amdgcn_s_memrealtime(); }
  const int blk = x.blk;
  if (blk < a.nbC) {
    if (!CSPLIT && a.cseg) apply_cseg_block<AP_PRESUM, true, NC, AP_OWN, TLSAN_TABLE_F32>(a, x, blk * AP_ROWS_PB, shp);
    else apply_cate_block<AP_PRESUM, true, NC, TLSAN_TABLE_F32, CSPLIT>(a, x, shd, shp, sh_pos, sh_lo, sh_n, sh_wtot);
  }
  else if (blk < a.nbC + a.nbI) apply_rows_block<AP_PRESUM, true, true, NI, AP_OWN, TLSAN_TABLE_F32>(a, x, (blk - a.nbC) * AP_ROWS_PB, shp);
  else {
    apply_rows_block<AP_PRESUM, true, false, NU, AP_OWN / 2, TLSAN_TABLE_F32>(a, x, (blk - a.nbC - a.nbI) * AP_ROWS_PB, shp);
    if constexpr (!WIDE) {   // user rows wider than 128 floats with narrow item / category rows: the second half of the row
      if (a.WU > 128) apply_rows_block<AP_PRESUM, true, false, NU, AP_OWN / 2, TLSAN_TABLE_F32, NU>(a, x, (blk - a.nbC - a.nbI) * AP_ROWS_PB, shp);
    }
  }
  if (stp && x.tid == 0) { stp[6] = __builtin_amdgcn_s_memtime(); stp[5] = __builtin_amdgcn_s_memrealtime(); }
}

// 16 category rows, one per 16-lane group: w -= lazy_scale * (the row's presummed gradient: Rc, or Rc64 -- the exact doubles the
// split category workgroups left, rounded to float as a single workgroup would have, and cleared).  Returns the lane's share
// of the change of the stored table's sum of squares.  (k_update_lazy; k_spec_commit<.., CSPL>)
template <int NC, int DT>
__device__ __forceinline__ double update_cate_rows(const ApplyArgs& a, int c, int l16, float lazy_scale, uint32_t salt) {
  double part = 0.0;
  if (c < a.C) {
    const size_t wrow = (size_t)c * a.dc;
    f32x4 w[NC], g[NC];
#pragma unroll
    for (int ch = 0; ch < NC; ++ch)
      if (4 * (l16 + 16 * ch) < a.dc) {
        w[ch] = tbl_ld4<DT>(a.p.cate_emb, wrow + 4 * (l16 + 16 * ch));
        if (a.csplit > 1) {
          double* r64 = a.Rc64 + wrow + 4 * (l16 + 16 * ch);
#pragma unroll
          for (int i = 0; i < 4; ++i) { g[ch][i] = (float)r64[i]; r64[i] = 0.0; }
        } else {
          g[ch] = *(const f32x4*)(a.Rc + wrow + 4 * (l16 + 16 * ch));
        }
      }
#pragma unroll
    for (int ch = 0; ch < NC; ++ch)
      if (4 * (l16 + 16 * ch) < a.dc) {
        const f32x4 w0 = w[ch];
        w[ch] = w0 - lazy_scale * g[ch];
        tbl_st4<DT>(a.p.cate_emb, wrow + 4 * (l16 + 16 * ch), w[ch], salt ^ 0x3c6ef372u);
#pragma unroll
        for (int i = 0; i < 4; ++i) part += (double)w[ch][i] * (double)w[ch][i] - (double)w0[i] * (double)w0[i];
      }
  }
  return part;
}

// ------------------------------------------------------------------------------------------
// The lazy-L2 step for tables that live in HBM (round 6): the SPECULATIVE one-pass update.
// The split form above sends every summed row through memory (written by the row-sum launch, read by k_update_lazy beside
// the parameter row's read-modify-write): at 10 M users / 5 M items that round trip is a third of the tail's traffic.  One
// pass over the used rows (segment sums and the row's update by the same lanes, k_apply<AP_UPDATE, lazy>) avoids it but
// needs the clip coefficient first, i.e. the finalize's whole chain in front of it (C5: 26 us).  clip_by_global_norm's
// coefficient is 1 unless the global norm exceeds the clip (model.py:201) -- so:
//   k_finalize_update : the finalize's workgroups lead the grid; the row workgroups update with coefficient 1 beside them.
//                       Neither P nor nstep change during the launch (FinArgs.spec): the summary leaves P_next / spec_salt.
//   k_spec_commit     : the dense parameters (which need the reduced gradients), the commit of P and nstep, and -- only if
//                       the coefficient turned out to be < 1 (or not finite) -- a correcting pass over the same rows:
//                       w += (scale_spec - scale_true) * sum, i.e. w_old - scale_true * sum up to one rounding.
// Unclipped steps are bit-equal to the one-pass form; results stay a fixed function of the batch.  Category segments only.
// (rows of tables with millions of rows are used once or twice per batch: two gradient rows in flight per 16-lane group
//  instead of AP_OWN = 8 clamped loads of the same row -- 60 registers fewer, five workgroups per CU instead of three;
//  longer segments are finished by the whole wavefront as everywhere, and the sums are exact: same bits)
#ifndef SPEC_OWN
#define SPEC_OWN 2
#endif
#define SPEC_FIX_BLOCKS 512
#define SPEC_ITEM_BLOCKS 2048   // item-row workgroups k_finalize_update launches at most (ApplyArgs.nbI_l)
// (the wide form -- rows of 128 floats and more, C5 -- is held to four waves per SIMD: 149 registers left alone, i.e. three;
//  at four 88 bytes per lane spill in the user-row role and C5 runs 281.5 -> 273.5 us/step.  The narrow form: five (fp32
//  tables) / four (bf16 tables) where the caches hold the tables -- the bench shape 56.9 -> 55.4 us/step against the split
//  form, at three it LOSES to it (59.3) --, three (LOWOCC) where they live in HBM: at d = 128 with 10 M / 5 M tables the step
//  is bound by the index stream, whose 1024-thread blocks find no slot beside five row workgroups per CU -- 80.5 us/step
//  with three, 92 with five: profiles/r06_lazy_one_pass.md)
#ifndef SPEC_WPE
#define SPEC_WPE 4
#endif
#ifndef SPEC_WPE_NARROW
#define SPEC_WPE_NARROW 5        // fp32 tables: 93 registers, nothing spilled
#define SPEC_WPE_NARROW_BF16 4   // bf16 tables (the stochastic rounding's hash): 113 registers; at five, 180 bytes per lane spill
#endif
// LOWOCC (narrow form, tables in HBM): three waves per SIMD -- see the note above
// CSPL (narrow form; few, large categories -- Movies-TV: 15 -- that several workgroups share, category_split): the category
// workgroups of this launch only SUM (exact doubles added into Rc64, as in k_finalize_presum<.., CSPLIT>) and the category
// rows are updated by k_spec_commit<.., CSPL>, which knows the coefficient; item and user rows as everywhere.  a.nbC is then
// the number of category-row blocks of the COMMIT launch (16 rows each: they own the records [0, nbC) of S_delta); this
// launch carries C * csplit category workgroups.  User rows of up to 256 floats (d = 128 with 90-entry windows) in two
// passes of the narrow form, as the row-sum launch takes them.
template <int D, int DH, bool WIDE, int DT, bool LOWOCC = false, bool CSPL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WIDE ? SPEC_WPE : (LOWOCC ? 3 : (DT == TLSAN_TABLE_F32 ? SPEC_WPE_NARROW : SPEC_WPE_NARROW_BF16))))) void k_finalize_update(FinArgs f, int nbK, int nbS, ApplyArgs a) {
  static_assert(!(WIDE && CSPL), "shared categories: narrow form only");
  constexpr int NC = WIDE ? 2 : 1, NI = WIDE ? 2 : 1, NU = WIDE ? 4 : 2;
  constexpr int OWN = WIDE ? SPEC_OWN : AP_OWN;
  __shared__ double shd[4 * 16 * NC * 4 > 256 ? 4 * 16 * NC * 4 : 256];
  __shared__ double shp[4];
  __shared__ int sh_pos[AP_CAP];
  __shared__ int sh_lo[256], sh_n[256];
  __shared__ int sh_wtot[4];
  __shared__ int sh_last;
  const int nfin = nbK + nbS + 1;
  if ((int)blockIdx.x < nfin) {
    // (debug stamps, scripts/stamps_apply.py: the finalize workgroups are listed after the row workgroups)
    unsigned long long* stp = a.stamps ? a.stamps + (size_t)(gridDim.x - nfin - a.nbH + blockIdx.x) * 8 : nullptr;
    if (stp && threadIdx.x == 0) { stp[0] = __builtin_amdgcn_s_memtime(); stp[4] = __builtin_amdgcn_s_memrealtime(); }
    dense_finalize_block<D, DH>(f, nbK, nbS, blockIdx.x, shd, &sh_last);
    if (stp && threadIdx.x == 0) { stp[6] = __builtin_amdgcn_s_memtime(); stp[5] = __builtin_amdgcn_s_memrealtime(); }
    return;
  }
  ApCtx x;
  x.tid = threadIdx.x; x.wave = x.tid >> 6; x.lane = x.tid & 63; x.grp = x.lane >> 4; x.l16 = x.lane & 15;
  x.gid = x.wave * 4 + x.grp;
  x.blk = blockIdx.x - nfin;
  x.P = a.hdr->P;               // (stable: this launch's summary does not commit)
  x.invP = 1.0f / x.P;
  x.step = a.lr;                // coefficient 1
  x.lazy_scale = x.step / (x.P * (1.0f - x.step * a.reg));
  x.salt = a.hdr->nstep + 1;    // what the step's salt and record tag will be (hdr->spec_salt)
  x.coef = 1.0f;
  if (x.blk == 0 && x.tid == 0) a.hdr->spart_n = a.nbC + a.nbI + a.nbU + a.nbH;
  if (x.blk < a.nbH) {          // hot item rows lead the row workgroups
    presum_hot_block<NI, true, DT>(a, x.blk, shd, shp, &x);
    return;
  }
  x.blk -= a.nbH;
  unsigned long long* stp = a.stamps ? a.stamps + (size_t)x.blk * 8 : nullptr;
  if (stp && x.tid == 0) { stp[0] = __builtin_amdgcn_s_memtime(); stp[4] = __builtin_amdgcn_s_memrealtime(); }
  if constexpr (CSPL) {
    const int nbCg = a.C * a.csplit;       // category workgroups of this launch (category, share)
    if (x.blk < nbCg) {
      apply_cate_block<AP_PRESUM, true, NC, DT, true>(a, x, shd, shp, sh_pos, sh_lo, sh_n, sh_wtot);
    } else {
      // (the row blocks' records follow the commit launch's category blocks': [nbC | nbI | nbU].  Two-pass user blocks are the
      //  longer ones and lead the item blocks -- the launch ends when its last-placed blocks do)
      int rb = x.blk - nbCg;
      const int nbIl = a.nbI_l > 0 ? a.nbI_l : a.nbI;     // item-row workgroups launched (ApplyArgs.nbI_l)
      const bool ufirst = a.WU > 128;
      const bool is_user = ufirst ? rb < a.nbU : rb >= nbIl;
      if (is_user) rb -= ufirst ? 0 : nbIl; else rb -= ufirst ? a.nbU : 0;
      x.blk = a.nbC + (is_user ? a.nbI : 0) + rb;
      if (!is_user) {
        const int nuq = *a.n_uniq_item;
        for (int g = rb; g * AP_ROWS_PB < nuq && g < a.nbI; g += nbIl) {
          x.blk = a.nbC + g;
          apply_rows_block<AP_UPDATE, true, true, NI, OWN, DT>(a, x, g * AP_ROWS_PB, shp);
          __syncthreads();   // (the shared scratch is reused by the next block of rows)
        }
      } else {
        apply_rows_block<AP_UPDATE, true, false, NU, AP_OWN / 2, DT>(a, x, rb * AP_ROWS_PB, shp);
        if (ufirst) {                        // the second half of a wide user row (its change of the sum of squares: added to the record)
          __syncthreads();
          x.accum = true;
          apply_rows_block<AP_UPDATE, true, false, NU, AP_OWN / 2, DT, NU>(a, x, rb * AP_ROWS_PB, shp);
        }
      }
    }
  } else {
    const int blk = x.blk;
    if (blk < a.nbC) {
      // (the wide form takes category segments only -- lazy_one_pass, tlsan_api.hip: the item-walk category workgroups in its
      //  kernel cost the row roles 44 more spilled bytes per lane)
      if (WIDE || a.cseg) apply_cseg_block<AP_UPDATE, true, NC, OWN, DT>(a, x, blk * AP_ROWS_PB, shp);
      else apply_cate_block<AP_UPDATE, true, NC, DT>(a, x, shd, shp, sh_pos, sh_lo, sh_n, sh_wtot);
    } else {
      // (records: [nbC | nbI | nbU] whatever the order of the workgroups.  The wide form's user rows -- 220 floats at C5,
      //  5.7-11 us a workgroup -- lead the item rows: placed last they WERE the launch's last 8 us: 60.5 -> 55)
      const int nbIl = a.nbI_l > 0 ? a.nbI_l : a.nbI;
      int rb = blk - a.nbC;
      const bool uf = a.ufirst != 0;
      const bool is_user = uf ? rb < a.nbU : rb >= nbIl;
      if (is_user) {
        rb -= uf ? 0 : nbIl;
        x.blk = a.nbC + a.nbI + rb;
        apply_rows_block<AP_UPDATE, true, false, NU, (WIDE ? SPEC_OWN : AP_OWN / 2), DT>(a, x, rb * AP_ROWS_PB, shp);
      } else {
        rb -= uf ? a.nbU : 0;
        const int nuq = *a.n_uniq_item;
        for (int g = rb; g * AP_ROWS_PB < nuq && g < a.nbI; g += nbIl) {
          x.blk = a.nbC + g;
          apply_rows_block<AP_UPDATE, true, true, NI, OWN, DT>(a, x, g * AP_ROWS_PB, shp);
          __syncthreads();   // (the shared scratch is reused by the next block of rows)
        }
      }
    }
  }
  if (stp && x.tid == 0) { stp[6] = __builtin_amdgcn_s_memtime(); stp[5] = __builtin_amdgcn_s_memrealtime(); }
}

// grid: nbD blocks of 256 dense parameters, (CSPL: a.nbC blocks of 16 category rows, updated here from the shared categories'
// exact sums with the step's true coefficient,) then at most SPEC_FIX_BLOCKS correcting workgroups (which return at once
// when the step was not clipped)
template <bool WIDE, int DT, bool CSPL = false>
__global__ __launch_bounds__(256) void k_spec_commit(ApplyArgs a) {
  constexpr int NC = WIDE ? 2 : 1, NI = WIDE ? 2 : 1, NU = WIDE ? 4 : 2;
  constexpr int OWN = WIDE ? SPEC_OWN : AP_OWN;
  __shared__ double shd[4 * 16 * NC * 4];
  __shared__ double shp[4];
  __shared__ int sh_pos[AP_CAP];
  __shared__ int sh_lo[256], sh_n[256];
  __shared__ int sh_wtot[4];
  const int tid = threadIdx.x;
  const float coef = a.hdr->coef;
  if ((int)blockIdx.x < a.nbD) {
    if (blockIdx.x == 0 && tid == 0) {   // (nothing in this launch reads P or nstep: P_prev / spec_salt hold what it needs)
      a.hdr->P = a.hdr->P_next;
      a.hdr->nstep += 1;
    }
    const float step = a.lr * coef;
    const int nd = blockIdx.x * 256 + tid;
    if (nd < a.lay.n_dense) {
      const float wn = a.p.dense[nd] - step * a.gd[nd];
      a.p.dense[nd] = wn;
      if (nd >= a.lay.K && nd < a.lay.k0) {
        const int idx = nd - a.lay.K;
        a.p.dense_KT[(size_t)(idx % a.D) * a.D + idx / a.D] = wn;
      }
    }
    return;
  }
  const float st_true = a.lr * coef;
  int fix0 = a.nbD;              // first correcting workgroup
  if constexpr (CSPL) {
    fix0 += a.nbC;
    if ((int)blockIdx.x < fix0) {   // 16 category rows: nothing speculative about them
      const float Pp = a.hdr->P_prev;
      const int cb = (int)blockIdx.x - a.nbD;
      const double part = update_cate_rows<NC, DT>(a, cb * 16 + (tid >> 4), tid & 15, st_true / (Pp * (1.0f - st_true * a.reg)), a.hdr->spec_salt);
      block_delta_store(part, shp, &a.delta_out[cb], a.hdr->spec_salt);
      return;
    }
  }
  if (coef == 1.0f) return;   // (block-uniform) the speculation held.  (A NaN coefficient takes the correcting pass and poisons the rows.)
  ApCtx x;
  x.tid = tid; x.wave = tid >> 6; x.lane = tid & 63; x.grp = x.lane >> 4; x.l16 = x.lane & 15;
  x.gid = x.wave * 4 + x.grp;
  x.P = a.hdr->P_prev;
  x.invP = 1.0f / x.P;
  x.step = st_true - a.lr;                                                   // item_b: w_spec - (st_true - lr) g = w_old - st_true g
  x.lazy_scale = st_true / (x.P * (1.0f - st_true * a.reg)) - a.lr / (x.P * (1.0f - a.lr * a.reg));
  x.salt = a.hdr->spec_salt;
  x.coef = coef;
  x.accum = true;
  // (the launch carries at most SPEC_FIX_BLOCKS correcting workgroups, each walking row blocks with the grid's stride: an
  //  unclipped step -- nearly every step -- pays for a few hundred workgroups that return at once, not for one per 16 rows)
  if constexpr (CSPL) {
    for (int v = (int)blockIdx.x - fix0; v < a.nbH + a.nbI + a.nbU; v += (int)gridDim.x - fix0) {
      if (v < a.nbH) {
        presum_hot_block<NI, true, DT>(a, v, shd, shp, &x);
      } else {
        const int rb = v - a.nbH;
        x.blk = a.nbC + rb;
        if (rb < a.nbI) {
          apply_rows_block<AP_UPDATE, true, true, NI, OWN, DT>(a, x, rb * AP_ROWS_PB, shp);
        } else {
          apply_rows_block<AP_UPDATE, true, false, NU, AP_OWN / 2, DT>(a, x, (rb - a.nbI) * AP_ROWS_PB, shp);
          if (a.WU > 128) {
            __syncthreads();
            apply_rows_block<AP_UPDATE, true, false, NU, AP_OWN / 2, DT, NU>(a, x, (rb - a.nbI) * AP_ROWS_PB, shp);
          }
        }
      }
      __syncthreads();
    }
    return;
  }
  for (int v = (int)blockIdx.x - a.nbD; v < a.nbH + a.nbC + a.nbI + a.nbU; v += (int)gridDim.x - a.nbD) {
    if (v < a.nbH) {
      presum_hot_block<NI, true, DT>(a, v, shd, shp, &x);
    } else {
      const int blk = v - a.nbH;
      x.blk = blk;
      if (blk < a.nbC) {
        if (WIDE || a.cseg) apply_cseg_block<AP_UPDATE, true, NC, OWN, DT>(a, x, blk * AP_ROWS_PB, shp);
        else apply_cate_block<AP_UPDATE, true, NC, DT>(a, x, shd, shp, sh_pos, sh_lo, sh_n, sh_wtot);
      }
      else if (blk < a.nbC + a.nbI) apply_rows_block<AP_UPDATE, true, true, NI, OWN, DT>(a, x, (blk - a.nbC) * AP_ROWS_PB, shp);
      else apply_rows_block<AP_UPDATE, true, false, NU, (WIDE ? SPEC_OWN : AP_OWN / 2), DT>(a, x, (blk - a.nbC - a.nbI) * AP_ROWS_PB, shp);
    }
    __syncthreads();   // (the shared scratch is reused by the next block of rows)
  }
}

// split category sums (Rc64, exact doubles) -> float output, and back to zero at rest (tlsan_grads)
__global__ void k_rc64_to_float(double* __restrict__ r64, float* __restrict__ out, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    out[t] = (float)r64[t];
    r64[t] = 0.0;
  }
}

// grid: nbC16 = ceil(C / 16) blocks of category rows, nbI / nbU blocks of used item / user rows
// (one row per 16-lane group), nbD blocks of 256 dense parameters
template <bool WIDE, int DT>
__global__ __launch_bounds__(256) void k_update_lazy(ApplyArgs a, int nbC16) {
  constexpr int NC = WIDE ? 2 : 1, NI = WIDE ? 2 : 1, NU = WIDE ? 4 : 2;
  __shared__ double shp[4];
  const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, gid = tid >> 4, blk = blockIdx.x;
  const float P = a.hdr->P_prev;  // (the step summary already advanced hdr->P)
  const float step = a.lr * a.hdr->coef;
  const float lazy_scale = step / (P * (1.0f - step * a.reg));
  const uint32_t salt = a.hdr->nstep;
  if (blk == 0 && tid == 0) a.hdr->spart_n = nbC16 + a.nbI + a.nbU;
  double part = 0.0;
  if (blk < nbC16) {
    part = update_cate_rows<NC, DT>(a, blk * 16 + gid, l16, lazy_scale, salt);
  } else if (blk < nbC16 + a.nbI) {
    const int slot0 = (blk - nbC16) * AP_ROWS_PB, slot = slot0 + gid;
    const int nuq = *a.n_uniq_item;
    if (slot0 >= nuq) return;
    if (slot < nuq) {
      const int row = a.urec_item[slot].x;
      const size_t wrow = (size_t)row * a.p.ld_item;
      f32x4 w[NI], g[NI];
      float wb = 0.0f, gb = 0.0f;
#pragma unroll
      for (int ch = 0; ch < NI; ++ch)
        if (4 * (l16 + 16 * ch) < a.di) {
          w[ch] = tbl_ld4<DT>(a.p.item_emb, wrow + 4 * (l16 + 16 * ch));
          g[ch] = *(const f32x4*)(a.Ri + (size_t)slot * a.di + 4 * (l16 + 16 * ch));
        }
      if (l16 == 0) { wb = a.p.item_b[(size_t)row * a.p.ld_itemb]; gb = a.Rb[slot]; }
#pragma unroll
      for (int ch = 0; ch < NI; ++ch)
        if (4 * (l16 + 16 * ch) < a.di) {
          const f32x4 w0 = w[ch];
          w[ch] = w0 - lazy_scale * g[ch];
          tbl_st4<DT>(a.p.item_emb, wrow + 4 * (l16 + 16 * ch), w[ch], salt ^ 0x85ebca6bu);
#pragma unroll
          for (int i = 0; i < 4; ++i) part += (double)w[ch][i] * (double)w[ch][i] - (double)w0[i] * (double)w0[i];
        }
      if (l16 == 0) a.p.item_b[(size_t)row * a.p.ld_itemb] = wb - step * gb;  // not regularised, never scaled
    }
  } else if (blk < nbC16 + a.nbI + a.nbU) {
    const int slot0 = (blk - nbC16 - a.nbI) * AP_ROWS_PB, slot = slot0 + gid;
    const int nuq = *a.n_uniq_user;
    if (slot0 >= nuq) return;
    if (slot < nuq) {
      const int row = a.urec_user[slot].x;
      const size_t wrow = (size_t)row * a.p.ld_user;
      float* Trow = a.p.usert_emb + (size_t)row * a.p.ld_usert;
      f32x4 w[NU], g[NU];
#pragma unroll
      for (int ch = 0; ch < NU; ++ch) {
        const int cc = 4 * (l16 + 16 * ch);
        if (cc < a.WU) g[ch] = *(const f32x4*)(a.Ru + (size_t)slot * a.WU + cc);
        if (cc < a.di) {
          w[ch] = tbl_ld4<DT>(a.p.user_emb, wrow + cc);
        } else if (cc < a.WU) {
#pragma unroll
          for (int i = 0; i < 4; ++i) w[ch][i] = (cc + i - a.di < a.Ls) ? Trow[cc + i - a.di] : 0.0f;
        }
      }
#pragma unroll
      for (int ch = 0; ch < NU; ++ch) {
        const int cc = 4 * (l16 + 16 * ch);
        if (cc < a.di) {
          const f32x4 w0 = w[ch];
          w[ch] = w0 - lazy_scale * g[ch];
          tbl_st4<DT>(a.p.user_emb, wrow + cc, w[ch], salt ^ 0xc2b2ae35u);
#pragma unroll
          for (int i = 0; i < 4; ++i) part += (double)w[ch][i] * (double)w[ch][i] - (double)w0[i] * (double)w0[i];
        } else if (cc < a.WU) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int p = cc + i - a.di;
            if (p < a.Ls) {
              const float w0 = w[ch][i], wn = w0 - lazy_scale * g[ch][i];
              Trow[p] = wn;
              part += (double)wn * (double)wn - (double)w0 * (double)w0;
            }
          }
        }
      }
    }
  } else {
    const int nd = (blk - nbC16 - a.nbI - a.nbU) * 256 + tid;
    if (nd < a.lay.n_dense) {
      const float wn = a.p.dense[nd] - step * a.gd[nd];
      a.p.dense[nd] = wn;
      if (nd >= a.lay.K && nd < a.lay.k0) {
        const int idx = nd - a.lay.K;
        a.p.dense_KT[(size_t)(idx % a.D) * a.D + idx / a.D] = wn;
      }
    }
    return;
  }
  block_delta_store(part, shp, &a.delta_out[blk], salt);
}


// stored *= P for one table (tlsan_state_renorm)
// (dt: storage type of the table; width % 4 == 0 for bf16 tables; bf16 values are rounded stochastically)
__global__ void k_scale_table(float* W, int rows, int width, int ld, const StateHdr* hdr, int dt, uint32_t salt) {
  const float P = hdr->P;
  if (dt == TLSAN_TABLE_F32) {
    const size_t n = (size_t)rows * width;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
      const size_t r = t / width, c = t % width;
      W[r * ld + c] *= P;
    }
    return;
  }
  const size_t n4 = (size_t)rows * (width / 4);
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (size_t)gridDim.x * blockDim.x) {
    const size_t r = t / (width / 4), c = 4 * (t % (width / 4));
    f32x4 w = tbl_ld4<TLSAN_TABLE_BF16>(W, r * ld + c) * P;
    tbl_st4<TLSAN_TABLE_BF16>(W, r * ld + c, w, salt ^ hdr->nstep);
  }
}

__global__ void k_renorm_commit(StateHdr* hdr) {
  const double P = hdr->P;
  hdr->St *= P * P;
  hdr->P = 1.0f;
}
