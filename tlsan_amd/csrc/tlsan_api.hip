// tlsan_api.hip -- the C ABI declared in include/tlsan.h: argument checking, workspace carving
// and kernel sequencing.  No allocation, no synchronisation; everything is enqueued on the
// caller's stream (so a whole step can be captured into a hipGraph).
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <sched.h>
#include <thread>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tlsan_common.h"
#include "tlsan_eval.h"
#include "tlsan_update.h"
#include "tlsan_shard.h"

struct LaunchEvents { hipEvent_t start, stop; };   // optional time stamps of the dispatch (tlsan_attn_inst.h)
hipError_t tlsan_launch_fwd_bwd_d64(bool train, bool lstream, const FwdArgs& a, int grid, hipStream_t st, LaunchEvents ev);
hipError_t tlsan_launch_fwd_bwd_d128(bool train, bool lstream, const FwdArgs& a, int grid, hipStream_t st, LaunchEvents ev);
hipError_t tlsan_launch_fwd_bwd_d256(bool train, bool lstream, const FwdArgs& a, int grid, hipStream_t st, LaunchEvents ev);
hipError_t tlsan_launch_fwd_bwd_d128w4(const FwdArgs& a, int grid, hipStream_t st, LaunchEvents ev);   // training, 8-sample workgroups

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define CHECK_LAUNCH(what)                                                         \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ != hipSuccess) return fail(TLSAN_E_LAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
  } while (0)

// ---- profiling ring (tlsan_profile_*): events at the kernel boundaries of a train step ----
#define PROF_MAX_STEPS 4096
#define PROF_MARKS 6
static unsigned long long* g_stamps = nullptr;
#define TLSAN_APPLY_STAMP_OFF (1 << 20)  // k_apply's stamps start this many entries into the debug buffer
static int g_prof_level = 0;
static int g_prof_n = 0;
static int g_prof_stride = 1;  // record every g_prof_stride-th step (tlsan_profile_stride)
static int g_prof_tick = 0;    // steps seen since the ring was enabled
static hipEvent_t* g_prof_ev = nullptr;  // [PROF_MAX_STEPS][PROF_MARKS], created on first enable
static void prof_mark(int mark, hipStream_t hs) {
  // (the sampled step of every stride is the middle one: not the first step after the caller's fence, whose kernel
  //  starts on an idle GPU and runs 5-8 % longer than the others)
  if (g_prof_level == 0 || g_prof_n >= PROF_MAX_STEPS || (g_prof_tick % g_prof_stride) != g_prof_stride / 2) return;
  if (g_prof_level == 1) return;   // (the kernel's own events: prof_kernel_events)
  (void)hipEventRecord(g_prof_ev[g_prof_n * PROF_MARKS + mark], hs);
}

// level 1 (the fused kernel's own duration): the two events ride on the kernel's dispatch packet (hipExtLaunchKernelGGL) --
// its begin / end time stamps, what a kernel trace reports -- instead of being recorded around it as barrier packets of
// their own, which read ~3 us longer and delay the step that carries them
static LaunchEvents prof_kernel_events() {
  LaunchEvents ev = {nullptr, nullptr};
  if (g_prof_level != 1 || g_prof_n >= PROF_MAX_STEPS || (g_prof_tick % g_prof_stride) != g_prof_stride / 2) return ev;
  ev.start = g_prof_ev[g_prof_n * PROF_MARKS + 1];
  ev.stop = g_prof_ev[g_prof_n * PROF_MARKS + 2];
  return ev;
}

static void prof_step_done() {
  if (g_prof_level == 0) return;
  if ((g_prof_tick % g_prof_stride) == g_prof_stride / 2 && g_prof_n < PROF_MAX_STEPS) ++g_prof_n;
  ++g_prof_tick;
}

static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct Shape {  // derived geometry of the supported (d, heads) combinations
  int D, DH, NSB, NPB, CW;
};

static int shape_of(const tlsan_dims* d, Shape* s) {
  if (!d) return fail(TLSAN_E_BADARG, "dims is NULL");
  if (d->num_heads <= 0 || d->d % d->num_heads) return fail(TLSAN_E_BADARG, "d %% num_heads != 0");
  const int D = d->d, DH = D / d->num_heads;
  if (d->d_item + d->d_cate != D) return fail(TLSAN_E_BADARG, "d_item + d_cate != d (model.py:100-109)");
  if (d->d_item % 4 || d->d_cate % 4 || d->d_item > 128 || d->d_cate > 128)
    return fail(TLSAN_E_UNSUPPORTED, "embedding widths must be multiples of 4 and <= 128");
  if (d->Ls < 1 || d->Ls > TLSAN_LS_CAP) return fail(TLSAN_E_UNSUPPORTED, "Ls must be in 1..%d", TLSAN_LS_CAP);
  if (d->user_count < 1 || d->item_count < 1 || d->cate_count < 1) return fail(TLSAN_E_BADARG, "empty table");
  s->D = D;
  s->DH = DH;
  if (D == 64 && DH == 8) { using G = Geo<64, 8>; s->NSB = G::NSB; s->NPB = G::NPB; s->CW = G::CW; }
  else if (D == 128 && DH == 16) { using G = Geo<128, 16>; s->NSB = G::NSB; s->NPB = G::NPB; s->CW = G::CW; }
  else if (D == 256 && DH == 32) { using G = Geo<256, 32>; s->NSB = G::NSB; s->NPB = G::NPB; s->CW = G::CW; }
  else return fail(TLSAN_E_UNSUPPORTED, "unsupported (hidden_units=%d, num_heads=%d): this build has 64/8, 128/8, 256/8", D, d->num_heads);
  return TLSAN_OK;
}

#define EVAL_DENSE_MAX ((size_t)256 << 20)  // all-items scoring materialises all_emb (model.py:89-90) up to this size
struct Ws {  // carve-up of the caller's scratch buffer
  float *Rc, *Ri, *Rb, *Ru;  // summed rows of the split lazy update
  float *Gi, *Gb, *Gu, *Gc, *gLong, *gDB, *gStat, *partials, *Kp, *gd, *sqd, *scal, *logits, *s_label;
  float* all_emb;  // evaluation: dense [I, D] item matrix (NULL when it would exceed EVAL_DENSE_MAX bytes)
  double* rownorm_part;
  double* rownorm;
  size_t bytes;
  int ngroups, nfin, nbK, nbS, WU;
};

static int ru4(int x) { return (x + 3) / 4 * 4; }
// the dK product rides in k_fwd_bwd for D <= 128 (Geo::FUSE_DK); its launch then has at most this many workgroups
// (each loops over its passes), so that the per-workgroup partials stay a few tens of MB at any batch size
// ... when there is at most one partial per CU: more of them (d = 64 at 8192 sequences: 512) make the reduction in
// k_dense_finalize the long pole of its launch (77 -> 91 us/step), and the separate k_dk_partial launch (<= 64 partials)
// is the better deal again
#define FUSED_DK_MAX_GROUPS 256
// tables with at least this many categories take the category-segment path (cate_seg below; TLSAN_CSEG_MIN=<n> for tests
// and experiments, read once per process)
#define TLSAN_CSEG_MIN_CATES 2048
static int cseg_min_cates() {
  static const int v = [] { const char* e = getenv("TLSAN_CSEG_MIN"); return e ? atoi(e) : TLSAN_CSEG_MIN_CATES; }();
  return v;
}
static bool fused_dk(int D, int ngroups) { return D <= 128 && ngroups <= FUSED_DK_MAX_GROUPS; }
// Samples per workgroup pass of a TRAINING launch of the fused kernel (= per partial record): the width's NSB, or 8 at
// d = 128 (4-wavefront workgroups: Geo<128, 16, 4>) with the window in registers and no dropout, when the batch is so
// small that 16-sample workgroups would leave three quarters of the CUs idle -- a wavefront then runs alone on its SIMD
// (profiles/r04_nw4_ab.md; at larger batches the 16-sample workgroups are faster).
// TLSAN_NW4 (A/B measurements; read once): 0 never, 1 as described, 2 always.
static int train_group(const Shape& s, const tlsan_dims* d, const tlsan_batch* b, const tlsan_hparams* hp) {
  static const int mode = [] { const char* e = getenv("TLSAN_NW4"); return e ? atoi(e) : 1; }();
  if (s.D != 128 || d->Ls > TLSAN_LS_MAX || (hp && hp->dropout != 0.0f) || mode == 0) return s.NSB;
  if (mode == 1 && (b->B + s.NSB - 1) / s.NSB > 64) return s.NSB;
  return 8;
}
static int fwd_train_grid(int ngroups) { return ngroups; }    // (fused: ngroups <= FUSED_DK_MAX_GROUPS, one pass per workgroup)

// windows longer than TLSAN_LS_MAX are streamed (the list form of the long block); shorter ones stay in registers
static bool streamed(int Ls) { return Ls > TLSAN_LS_MAX; }

static void carve(const tlsan_dims* d, const Shape& s, int B, int Sn, char* base, Ws* w) {
  size_t o = 0;
  auto take = [&](size_t n) { char* p = base ? base + o : nullptr; o += al(n); return p; };
  const size_t NI = (size_t)B * (d->Ls + Sn + 1), D = s.D;
  tlsan_dense_layout L;
  tlsan_dense_layout_of(d, &L);
  w->WU = ru4(d->d_item + d->Ls);
  w->ngroups = (B + 7) / 8;  // partial records: one per workgroup pass (the smallest pass any launch may take: train_group)
  // dK partials: one per batch split of k_dk_partial, or (fused into the forward/backward kernel, D <= 128) one per
  // workgroup of that launch -- which of the two, and how many, is the launch's choice (run_backward)
  const int ngroups = (B + s.NSB - 1) / s.NSB;
  w->nbK = (s.D * s.D + 255) / 256;
  const int small_pb = s.D > 128 ? FIN_SMALL_PB : 16;      // (small parameters per finalize workgroup: dense_finalize_block)
  w->nbS = (L.n_dense - s.D * s.D + small_pb - 1) / small_pb;
  w->nfin = w->nbK + w->nbS;
  // (+1 row: k_apply reads clamped addresses instead of branching, see AP_OWN)
  w->Gi = (float*)take(sizeof(float) * (NI + 1) * D);
  w->Gb = (float*)take(sizeof(float) * (NI + 1));
  w->Gu = (float*)take(sizeof(float) * (size_t)(B + 1) * w->WU);
  // (CSEG: one row per u_cate use AND per item use -- sized for it whenever the table shape can take that path)
  w->Gc = (float*)take(sizeof(float) * (size_t)(B + 1 + (d->cate_count >= cseg_min_cates() ? NI : 0)) * d->d_cate);
  w->gLong = (float*)take(sizeof(float) * B * D);
  w->gDB = (float*)take(sizeof(float) * B * D);
  // streamed windows at d = 256: per-sample softmax statistics of the long block (k_fwd_bwd, FLATG: no room in the LDS)
  w->gStat = (float*)take((D > 128 && streamed(d->Ls)) ? sizeof(float) * (size_t)B * 2 * D : 0);
  w->partials = (float*)take(sizeof(float) * w->ngroups * s.NPB);
  // (sized so that the workspace of a batch also holds every smaller batch: the fused form of a smaller batch can
  //  need more partials than the split form of a larger one)
  int kp_slots = dk_nsplit(B, s.D);
  if (s.D <= 128) {
    const int ng8 = s.D == 128 ? (B + 7) / 8 : ngroups;    // (8-sample workgroups: twice the groups)
    const int fmax = ng8 < FUSED_DK_MAX_GROUPS ? ng8 : FUSED_DK_MAX_GROUPS;
    if (fmax > kp_slots) kp_slots = fmax;
  }
  w->Kp = (float*)take(sizeof(float) * kp_slots * D * D);
  w->gd = (float*)take(sizeof(float) * L.n_dense);
  w->sqd = (float*)take(sizeof(float) * w->nfin);
  w->scal = (float*)take(sizeof(float) * 4);
  w->logits = (float*)take(sizeof(float) * B);
  w->s_label = (float*)take(sizeof(float) * B);
  {
    const size_t ae = sizeof(float) * (size_t)d->item_count * D;
    w->all_emb = ae <= EVAL_DENSE_MAX ? (float*)take(ae) : nullptr;
  }
  const size_t nrowblk = (size_t)(d->item_count + 15) / 16 + (d->user_count + 15) / 16 + d->cate_count;
  {  // summed rows of the used rows (lazy update: k_finalize_presum -> k_update_lazy)
    const size_t ri = (NI < (size_t)d->item_count ? NI : (size_t)d->item_count) + AP_ROWS_PB;
    const size_t ru = ((size_t)B < (size_t)d->user_count ? (size_t)B : (size_t)d->user_count) + AP_ROWS_PB;
    w->Ri = (float*)take(sizeof(float) * ri * d->d_item);
    w->Rb = (float*)take(sizeof(float) * ri);
    w->Ru = (float*)take(sizeof(float) * ru * w->WU);
    w->Rc = (float*)take(sizeof(float) * (size_t)d->cate_count * d->d_cate);
  }
  w->rownorm_part = (double*)take(8 * nrowblk);
  w->rownorm = (double*)take(8);
  w->bytes = o;
}

#define BAL_CAP (1 << 14)      // batches up to this many samples are ranked for the fused kernel's workgroups (BalArgs, tlsan_update.h)
// Streamed windows: a workgroup's time is proportional to its windows' total length (d = 64, Ls = 90: 98 -> 87 us/step;
// d = 256 at the C5 shape 496 -> 471; d = 128: the kernel alone 94 -> 72 us, 104 -> 93 beside the index build of the batch
// after next).  Windows in registers: evenly loaded workgroups (ranked by session length alone) ran no faster in round 4
// (bench 61.0 vs 61.1 us/step; Amazon session lengths 65.6 vs 66.2) -- the launch ends with the workgroups that hold one
// of the few long sessions, however they are dealt; what does help is giving THOSE workgroups the batch's shortest
// windows (round 5, BalArgs: bench 58.2 -> 57.15, Amazon session lengths 61.85 -> 59.05, d = 256 169.4 -> 162.0:
// profiles/r05_ab_bal_reg.txt).
// (round 5: windows in registers as well, with a ranking that gives the workgroups of the batch's longest sessions its
//  shortest windows -- BalArgs; TLSAN_BAL_REG=0: streamed windows only, as before)
static int bal_reg_mode() {   // 0 off, 1 on, 2 on without the reversed group numbering (A/B)
  static const int reg = [] { const char* e = getenv("TLSAN_BAL_REG"); return e ? atoi(e) : 1; }();
  return reg;
}
static bool balanced(const tlsan_dims* d, const tlsan_batch* b) {
  if (!b || b->B <= 16 || b->B > BAL_CAP) return false;
  if (streamed(d->Ls)) return true;
  // windows in registers: where it was measured to win -- one round of workgroups (<= 256 groups of 16: a second round
  // evens the workgroups out by itself, and 16384 sequences ran 5 % SLOWER ranked: 207 vs 196 us/step) and tables that the
  // caches hold (10 M users / 5 M items: 101.7 vs 97.4 -- full windows side by side mean more HBM rows in flight at once);
  // profiles/r05_ab_bal_reg_shapes2.txt
  return bal_reg_mode() != 0 && b->Sn > 0 && (b->B + 15) / 16 <= FUSED_DK_MAX_GROUPS && d->item_count <= (1 << 18);
}
#define UC_LIST_CAP (1 << 18)  // batches up to this many samples may keep a per-category sample list in the state
// ... and do when a category sees many of the batch's samples (cursor atomics on few addresses inside the
// fused kernel cost more than the extra launch on the index stream): more than 32 samples per category
static inline bool uc_by_list(const tlsan_dims* d, const tlsan_batch* b) {
  return b && b->B <= UC_LIST_CAP && (long)b->B > 32L * d->cate_count;
}
// many categories (the 10 k of BASELINE.json configs[4]): the category half of every item use's gradient row is written
// into the category's own segment of Gc (FwdArgs.cseg) -- a category block of the apply pass then sums one contiguous
// segment instead of walking the category's items for their segments (500 items per category at 5 M items, of which a
// batch uses five: k_finalize_presum 196 us at that shape).  With few categories the cursor draws would pile up on few
// addresses (as the u_cate uses did, uc_by_list): the item walk stays.
// (the category counts of CSEG need the item -> category map: tlsan_batch_index takes it as an argument -- round 3 kept
//  a process-global registry state -> map filled by tlsan_state_init, which a re-allocated map or a recycled state
//  address left stale)
// Tables of at least this many rows take their side of a lazy-SGD step's index from a sort of the batch's ids instead of
// a counter per row -- when its consumers reach it through ids and records only: the user table (UsortArgs: a bucket
// sort inside one block), the item table with category segments (IsortArgs: a partitioned counting sort, tlsan_update.h).
// TLSAN_ISORT_MIN=<rows> for tests (read once per process; it also sizes the state).
static int isort_min_rows() {
  static const int v = [] { const char* e = getenv("TLSAN_ISORT_MIN"); return e ? atoi(e) : (1 << 16); }();
  return v;
}
static bool isort_rows_ok(int rows) { return rows >= isort_min_rows() && (long)rows <= (long)IS_MAXB * IS_BSZ; }
static inline bool cate_seg(const tlsan_dims* d, const tlsan_batch* b) {
  return d->cate_count >= cseg_min_cates() && !uc_by_list(d, b);
}
struct St {  // persistent state
  // two index slots (a batch's destination index depends only on its ids, so it lives with the
  // state, not in the per-call workspace whose layout follows the batch shape):
  int32_t *cnt_item[TLSAN_INDEX_SLOTS], *cnt_uc[TLSAN_INDEX_SLOTS], *cnt_user[TLSAN_INDEX_SLOTS];   // use counters, zero at rest
  int32_t *off_item[TLSAN_INDEX_SLOTS], *off_uc[TLSAN_INDEX_SLOTS], *off_user[TLSAN_INDEX_SLOTS];   // segment offsets (n+1 entries)
  int32_t *cur_item[TLSAN_INDEX_SLOTS], *cur_uc[TLSAN_INDEX_SLOTS], *cur_user[TLSAN_INDEX_SLOTS];   // fill cursors
  int4 *urec_item[TLSAN_INDEX_SLOTS], *urec_user[TLSAN_INDEX_SLOTS];   // (row, first position, uses) of the used rows
  int32_t *cate_off, *cate_cnt, *cate_cur, *cate_items;   // static CSR category -> items
  StateHdr* hdr;
  double *S_part, *S_total;
  DeltaRec* S_delta;                                      // per-workgroup changes of the sum of squares, tagged by step (tlsan_update.h)
  long long* scan_bsum[TLSAN_INDEX_SLOTS];                                // per-chunk sums of the index scan (large tables), per slot
  int32_t* perm[TLSAN_INDEX_SLOTS];                                       // samples of every workgroup of the fused kernel (BalArgs), BAL_CAP each
  int32_t* scan_ticket;                                                   // [index slot] arrivals of k_scan_block_sums (ScanArgs.bs_ticket), zero at rest
  int32_t* flag_user[TLSAN_INDEX_SLOTS];                                  // 256-row pieces of the user table that hold a count (ScanArgs.flag), zero at rest
  int32_t* uc_list[TLSAN_INDEX_SLOTS];                                    // samples of every category (u_cate uses), UC_LIST_CAP each
  double* Rc64;                                           // category sums of a split PRESUM pass, zero at rest
  int32_t* hot_list[TLSAN_INDEX_SLOTS];                                   // slots (urec_item) of the hot item rows, AP_HOT_CAP each
  // the item side of the index from a partitioned counting sort of the batch's ids (IsortArgs; tables of isort_min_rows() rows or more)
  int32_t *is_bh[TLSAN_INDEX_SLOTS], *is_ids[TLSAN_INDEX_SLOTS], *is_bstart[TLSAN_INDEX_SLOTS], *is_nd[TLSAN_INDEX_SLOTS];
  int4* is_tmp[TLSAN_INDEX_SLOTS];
  size_t bytes;
  int nbI, nbU, nbC;
};

static void carve_state(const tlsan_dims* d, char* base, St* s) {
  size_t o = 0;
  auto take = [&](size_t n) { char* p = base ? base + o : nullptr; o += al(n); return p; };
  s->nbI = (d->item_count + AP_ROWS_PB - 1) / AP_ROWS_PB;
  s->nbU = (d->user_count + AP_ROWS_PB - 1) / AP_ROWS_PB;
  s->nbC = d->cate_count;
  s->hdr = (StateHdr*)take(sizeof(StateHdr));  // must stay first: tlsan_state_scale(state) == state
  for (int k = 0; k < TLSAN_INDEX_SLOTS; ++k) {
    s->cnt_item[k] = (int32_t*)take(4 * (size_t)d->item_count);
    s->cnt_uc[k] = (int32_t*)take(4 * (size_t)d->cate_count);
    s->cnt_user[k] = (int32_t*)take(4 * (size_t)d->user_count);
    s->off_item[k] = (int32_t*)take(4 * ((size_t)d->item_count + 1));
    s->off_uc[k] = (int32_t*)take(4 * ((size_t)d->cate_count + 1));
    s->off_user[k] = (int32_t*)take(4 * ((size_t)d->user_count + 1));
    s->cur_item[k] = (int32_t*)take(4 * (size_t)d->item_count);
    s->cur_uc[k] = (int32_t*)take(4 * (size_t)d->cate_count);
    s->cur_user[k] = (int32_t*)take(4 * (size_t)d->user_count);
    s->urec_item[k] = (int4*)take(16 * ((size_t)d->item_count + AP_ROWS_PB));
    s->urec_user[k] = (int4*)take(16 * ((size_t)d->user_count + AP_ROWS_PB));
  }
  s->cate_off = (int32_t*)take(4 * (size_t)d->cate_count);
  s->cate_cnt = (int32_t*)take(4 * (size_t)d->cate_count);
  s->cate_cur = (int32_t*)take(4 * (size_t)d->cate_count);
  s->cate_items = (int32_t*)take(4 * (size_t)d->item_count);
  s->S_part = (double*)take(8 * (size_t)(s->nbI + s->nbU + s->nbC));
  s->S_delta = (DeltaRec*)take(sizeof(DeltaRec) * ((size_t)(s->nbI + s->nbU + s->nbC) + AP_HOT_CAP));   // (+ the hot-row workgroups of a one-pass update)
  for (int k = 0; k < TLSAN_INDEX_SLOTS; ++k) s->uc_list[k] = (int32_t*)take(4 * (size_t)UC_LIST_CAP);
  s->Rc64 = (double*)take(8 * (size_t)d->cate_count * d->d_cate);
  for (int k = 0; k < TLSAN_INDEX_SLOTS; ++k) s->hot_list[k] = (int32_t*)take(4 * (size_t)AP_HOT_CAP);
  for (int k = 0; k < TLSAN_INDEX_SLOTS; ++k)
    s->scan_bsum[k] = (long long*)take(8 * ((size_t)(d->item_count + 4095) / 4096 + (d->cate_count + 4095) / 4096 +
                                            (d->user_count + 4095) / 4096));
  for (int k = 0; k < TLSAN_INDEX_SLOTS; ++k) s->perm[k] = (int32_t*)take(4 * (size_t)BAL_CAP);
  s->scan_ticket = (int32_t*)take(4 * 64);
  for (int k = 0; k < TLSAN_INDEX_SLOTS; ++k) s->flag_user[k] = (int32_t*)take(4 * (((size_t)d->user_count + 255) / 256));
  {
    const bool on = isort_rows_ok(d->item_count);
    for (int k = 0; k < TLSAN_INDEX_SLOTS; ++k) {
      s->is_bh[k] = (int32_t*)take(on ? 4 * (size_t)(ISORT_MAX_SLOTS / IS_BLK_SLOTS) * IS_MAXB : 0);
      s->is_ids[k] = (int32_t*)take(on ? 4 * (size_t)ISORT_MAX_SLOTS : 0);
      s->is_bstart[k] = (int32_t*)take(on ? 4 * (size_t)(IS_MAXB + 1) : 0);
      s->is_nd[k] = (int32_t*)take(on ? 4 * (size_t)IS_MAXB : 0);
      s->is_tmp[k] = (int4*)take(on ? 16 * (size_t)ISORT_MAX_SLOTS : 0);
    }
  }
  s->S_total = base ? &s->hdr->St : nullptr;
  s->bytes = o;
}

extern "C" {

int tlsan_abi_version(void) { return TLSAN_ABI_VERSION; }
const char* tlsan_last_error(void) { return g_err; }

int tlsan_dense_layout_of(const tlsan_dims* d, tlsan_dense_layout* L) {
  if (!d || !L || d->num_heads <= 0) return fail(TLSAN_E_BADARG, "null dims/layout");
  const int D = d->d, dh = D / d->num_heads;
  int o = 0;
  L->f1_W1 = o; o += dh * dh;
  L->f1_b1 = o; o += dh;
  L->f1_W2 = o; o += dh * dh;
  L->f1_b2 = o; o += dh;
  L->K = o; o += D * D;
  L->k0 = o; o += D;
  L->f2_W1 = o; o += dh * dh;
  L->f2_b1 = o; o += dh;
  L->f2_W2 = o; o += dh * dh;
  L->f2_b2 = o; o += dh;
  L->gamma = o; o += 1;
  L->n_dense = o;
  return TLSAN_OK;
}

size_t tlsan_workspace_bytes(const tlsan_dims* d, int32_t max_B, int32_t max_Sn) {
  Shape s;
  if (shape_of(d, &s) != TLSAN_OK || max_B < 1 || max_Sn < 0) return 0;
  Ws w;
  carve(d, s, max_B, max_Sn, nullptr, &w);
  return w.bytes;
}

size_t tlsan_state_bytes(const tlsan_dims* d) {
  Shape s;
  if (shape_of(d, &s) != TLSAN_OK) return 0;
  St st;
  carve_state(d, nullptr, &st);
  return st.bytes;
}

static int check_params(const tlsan_params* p) {
  if (!p || !p->item_emb || !p->item_b || !p->user_emb || !p->usert_emb || !p->cate_emb || !p->dense ||
      !p->dense_KT || !p->item_cate)
    return fail(TLSAN_E_BADARG, "NULL parameter pointer");
  if (p->table_dtype != TLSAN_TABLE_F32 && p->table_dtype != TLSAN_TABLE_BF16) return fail(TLSAN_E_BADARG, "table_dtype");
  if (p->matrix_dtype != TLSAN_MATRIX_F32 && p->matrix_dtype != TLSAN_MATRIX_BF16) return fail(TLSAN_E_BADARG, "matrix_dtype");
  if (p->table_dtype == TLSAN_TABLE_BF16 && ((p->ld_item | p->ld_user) % 4))
    return fail(TLSAN_E_UNSUPPORTED, "bf16 tables need row strides that are multiples of 4 elements");
  return TLSAN_OK;
}

// fill in the default (dense) row strides
static tlsan_params norm_params(const tlsan_params* p, const tlsan_dims* d) {
  tlsan_params q = *p;
  if (q.ld_item == 0) q.ld_item = d->d_item;
  if (q.ld_itemb == 0) q.ld_itemb = 1;
  if (q.ld_user == 0) q.ld_user = d->d_item;
  if (q.ld_usert == 0) q.ld_usert = d->Ls;
  return q;
}

static int check_batch(const tlsan_dims* d, const tlsan_batch* b, bool train) {
  if (!b || b->B < 1 || b->Sn < 0) return fail(TLSAN_E_BADARG, "bad batch (B=%d, Sn=%d)", b ? b->B : -1, b ? b->Sn : -1);
  if (!b->u || !b->i || !b->hist_i || !b->hist_t || !b->sl || !b->sl_new || !b->u_cate || (b->Sn > 0 && !b->hist_i_new))
    return fail(TLSAN_E_BADARG, "NULL batch pointer");
  if (train && !b->y) return fail(TLSAN_E_BADARG, "training needs labels y");
  if ((size_t)b->B * (d->Ls + b->Sn + 2) >= ((size_t)1 << 31)) return fail(TLSAN_E_UNSUPPORTED, "B*S overflows int32");
  if (train && b->Sn > TLSAN_SN_CAP) return fail(TLSAN_E_UNSUPPORTED, "training supports sessions up to %d items (got Sn=%d)", TLSAN_SN_CAP, b->Sn);
  return TLSAN_OK;
}

static void fill_apply(ApplyArgs& A, const tlsan_dims* d, const Shape& s, const tlsan_params* p, const tlsan_batch* b,
                       const tlsan_hparams* hp, const Ws& w, const St& st, const tlsan_dense_layout& L) {
  const int k = hp ? hp->index_slot : 0;
  memset(&A, 0, sizeof(A));
  A.p = norm_params(p, d);
  A.lay = L;
  A.I = d->item_count; A.U = d->user_count; A.C = d->cate_count; A.Ls = d->Ls; A.D = s.D;
  A.di = d->d_item; A.dc = d->d_cate; A.WU = ru4(d->d_item + d->Ls);
  A.Gi = w.Gi; A.Gb = w.Gb; A.Gu = w.Gu; A.Gc = w.Gc;
  A.cnt_item = st.cnt_item[k]; A.cnt_uc = st.cnt_uc[k]; A.cnt_user = st.cnt_user[k];
  A.off_item = st.off_item[k]; A.off_uc = st.off_uc[k]; A.off_user = st.off_user[k];
  A.n_uniq_item = st.hdr ? &st.hdr->n_uniq[k][0] : nullptr; A.n_uniq_user = st.hdr ? &st.hdr->n_uniq[k][1] : nullptr;
  A.cate_off = st.cate_off; A.cate_cnt = st.cate_cnt; A.cate_items = st.cate_items;
  A.uc_list = uc_by_list(d, b) ? st.uc_list[k] : nullptr;
  A.cseg = (b && cate_seg(d, b)) ? 1 : 0;
  A.Rc64 = st.Rc64;
  A.csplit = 1; A.cpass = 256; A.cpos = 0;
  A.hot_n = st.hdr ? &st.hdr->n_hot[k] : nullptr; A.hot_list = st.hot_list[k]; A.nbH = 0;
  A.gd = w.gd;
  A.Rc = w.Rc; A.Ri = w.Ri; A.Rb = w.Rb; A.Ru = w.Ru;
  A.part_out = st.S_part; A.delta_out = st.S_delta; A.hdr = st.hdr;
  A.urec_item = st.urec_item[k]; A.urec_user = st.urec_user[k];
  if (hp) { A.lr = hp->lr; A.reg = hp->reg; }
  A.nbI = st.nbI; A.nbU = st.nbU; A.nbC = st.nbC; A.nbD = (L.n_dense + 255) / 256;
  if (A.cseg) A.nbC = (A.C + AP_ROWS_PB - 1) / AP_ROWS_PB;     // (16 categories per workgroup: apply_cseg_block)
  A.stamps = g_stamps ? g_stamps + TLSAN_APPLY_STAMP_OFF : nullptr;
}

// one apply pass = one launch: category rows, item/user rows (+ dense parameters).
// lazy (UPDATE only): the row blocks walk the compacted records of used rows.
static void lazy_blocks(ApplyArgs& A, int B, int Sn) {  // at most min(rows, uses) rows were used
  const long ni = (long)B * (A.Ls + Sn + 1);
  A.nbI = (int)(((ni < A.I ? ni : A.I) + AP_ROWS_PB - 1) / AP_ROWS_PB);
  A.nbU = ((B < A.U ? B : A.U) + AP_ROWS_PB - 1) / AP_ROWS_PB;
}

static bool apply_wide(const ApplyArgs& A) { return A.di > 64 || A.dc > 64 || A.WU > 128; }  // more float4 chunks per lane

// few, large categories: several workgroups per category in the row-sum pass, every one with its share of the items and of the
// u_cate uses (estimated from the batch shape; up to 64 per category).
//  * from 512 uses per category on: about 128 uses per workgroup (round 3; Movies-TV's 15 categories at batch 4096), within
//    a budget of ~700 category workgroups (round 6, below);
//  * round 6 -- where the launch has SLOTS TO SPARE (its other workgroups and the category workgroups all resident at once:
//    small batches), from ~100 uses on and ~48 per workgroup: a category workgroup is a chain of dependent trips (3 us
//    before its first gradient row arrives) plus ~0.03 us per use, and such a launch ends with its longest chain --
//    Digital-Music's 53 categories at batch 1024 (300 uses each) took 9-13 us where everything else had finished after 7:
//    51.4 -> 48.2 us/step.  Where the launch is bound by slots (batch 4096: the bench's 673 categories of ~100 uses,
//    Movies-TV) every workgroup more costs its lead-in again: 64 instead of 46 per category at Movies-TV, Ls = 10:
//    59.1 -> 61.5 (profiles/r06_ab_csplit.txt).
// TLSAN_CSPLIT_FINE=0 (read once): the first rule only (A/B).
static void category_split(ApplyArgs& A, const tlsan_dims* d, const tlsan_batch* b) {
  // category segments (A.cseg) sum a category as ONE contiguous segment, 16 categories per workgroup
  // (apply_cseg_block): there is nothing to split, and the split kernels decode blocks as (category, share)
  if (A.cseg) return;
  const long uses = ((long)b->B * (d->Ls + b->Sn + 2) + d->cate_count - 1) / d->cate_count;
  const int per = (d->item_count + d->cate_count - 1) / d->cate_count;
  static const int fine = [] { const char* e = getenv("TLSAN_CSPLIT_FINE"); return e ? atoi(e) : 1; }();
  // (the first rule's budget of category workgroups, round 6: every share repeats the category's lead-in, and a launch
  //  bound by slots pays for it 1:1 -- 673 categories of 620 uses (Ls = 90) as four shares each: 2 692 workgroups of 8.8 us,
  //  24 of the launch's 32 k slot-us; unshared: d = 256 239 -> 224 us/step, d = 128 103.5 -> 97.4.  Movies-TV's 15
  //  categories run best as ~46 shares each at Ls = 10 AND at Ls = 90 (64: 97.7, 46: 94.4, 30: 94.9, 11: 104), i.e. ~700
  //  category workgroups beside the rows' on 1 280 slots.  TLSAN_CSPLIT_PER / TLSAN_CSPLIT_BUDGET for A/B.)
  static const int per_share = [] { const char* e = getenv("TLSAN_CSPLIT_PER"); return e && atoi(e) > 0 ? atoi(e) : 128; }();
  static const int budget = [] { const char* e = getenv("TLSAN_CSPLIT_BUDGET"); return e && atoi(e) > 0 ? atoi(e) : 700; }();
  long n = uses > 512 ? (uses / per_share < 64 ? uses / per_share : 64) : 1;
  if (n > budget / d->cate_count) n = budget / d->cate_count;
  if (n < 1) n = 1;
  if (fine && uses > 96) {
    // the launch's other workgroups: the finalize's and the hot rows' (~206), 16 used item / user rows each (lazy_blocks)
    const long ni = (long)b->B * (d->Ls + b->Sn + 1);
    const long others = 206 + ((ni < d->item_count ? ni : d->item_count) + 15) / 16 + ((b->B < d->user_count ? b->B : d->user_count) + 15) / 16;
    long nf = uses / 48 < 64 ? uses / 48 : 64;
    const long spare = (1280 - others) / d->cate_count;     // (256 CUs x five 256-thread workgroups)
    if (nf > spare) nf = spare;
    if (nf > n) n = nf;
  }
  if (n > 1) {
    A.csplit = (int)n;
    const int ps = (per + A.csplit - 1) / A.csplit;
    A.cpass = ps < 1 ? 1 : (ps > 256 ? 256 : ps);
    // categories of at most 256 items (one pass of the walk: the static CSR's counts, not the average, would say; the
    // kernel takes further passes the same way if one is larger): shares by use position instead of by item, so that a
    // hot item does not make its share the launch's longest chain.  TLSAN_CSPLIT_POS=0: by item (A/B)
    static const int by_pos = [] { const char* e = getenv("TLSAN_CSPLIT_POS"); return e ? atoi(e) : 1; }();
    A.cpos = (by_pos && per <= 128) ? 1 : 0;
  }
}

// The lazy update as ONE pass over the used rows (round 6; k_finalize_update / k_spec_commit, tlsan_update.h).  The split
// form (row sums in the finalize's launch, then k_update_lazy) sends every summed row through memory -- written by one
// launch, read by the next beside the parameter row's read-modify-write -- and ends in a launch of its own; in the one-pass
// form a 16-lane group sums its row's segment and updates the row, speculating on clip coefficient 1, beside the finalize.
static bool tables_in_hbm(const tlsan_dims* d) {      // (well beyond the 256 MiB Infinity Cache)
  return 4.0 * ((double)d->item_count * d->d_item + (double)d->user_count * (d->d_item + d->Ls)) > 512e6;
}
// Where the one-pass form was measured to win (profiles/r06_lazy_one_pass.md): rows of up to 64 floats per table half
// (d <= 128) at any table size -- bench shape 56.9 -> 55.4 us/step, 8192 sequences 106.5 -> 103.6, Amazon session lengths
// 59.9 -> 57.8, 10 M / 5 M tables 97 -> 80 --; wider rows (d = 256) only where the tables live in HBM (C5 300 -> 267; with
// cache-resident tables it loses 2.5 us to the split form).  TLSAN_LAZY_ONE_PASS: 0 never, 1 (default) as described,
// 2 whenever the tables take category segments, 3 wherever the form is built.
// Returns 0 (the split form), 1 (one pass), or 2: one pass over the item and user rows while the category rows -- few, large
// categories (Movies-TV: 15) that several row-sum workgroups share, adding exact doubles with atomics (category_split) -- are
// summed beside them and updated by the commit launch (k_finalize_update / k_spec_commit<.., CSPL>; TLSAN_LAZY_CSPL=0: off).
static int lazy_one_pass(const tlsan_dims* d, const tlsan_batch* b, const ApplyArgs& A) {
  static const int mode = [] { const char* e = getenv("TLSAN_LAZY_ONE_PASS"); return e ? atoi(e) : 1; }();
  static const int cspl = [] { const char* e = getenv("TLSAN_LAZY_CSPL"); return e ? atoi(e) : 1; }();
  if (mode == 0) return 0;
  ApplyArgs T = A;
  category_split(T, d, b);
  if (T.csplit > 1) {
    if (!cspl || mode == 2 || A.di > 64 || A.dc > 64 || A.WU > 256) return 0;   // (built in the narrow form: d <= 128)
    if (mode == 1 && A.p.table_dtype == TLSAN_TABLE_BF16 && !tables_in_hbm(d)) return 0;   // (as below)
    return 2;
  }
  if (apply_wide(A) && !A.cseg) return 0;      // (the wide form is built for category segments only)
  if (mode >= 2) return mode == 2 ? A.cseg != 0 : 1;
  // bf16 tables: a clipped step rounds twice in this form -- the speculative write at the magnitude of w - lr g, the
  // correction at that of the result -- so its stored elements can be off by one ulp of the SPECULATIVE value (unbiased,
  // and only in clipped steps; fp32 tables: 2^-24 of it, far inside every bound).  Taken where it pays for that (tables in
  // HBM: C5 in bf16 227 -> 202 us/step); with cache-resident bf16 tables (0.4-1.0 us) the split form and its
  // one-rounding guarantee stay.
  if (A.p.table_dtype == TLSAN_TABLE_BF16 && !tables_in_hbm(d)) return 0;
  if (!apply_wide(A)) return 1;
  return A.cseg && tables_in_hbm(d) ? 1 : 0;
}

// second half of the split lazy update (the first half rides with the dense finalize, run_backward)
static int launch_update_lazy(ApplyArgs A, int B, int Sn, hipStream_t hs) {
  lazy_blocks(A, B, Sn);
  const int nbC16 = (A.C + 15) / 16;
  const dim3 g1(nbC16 + A.nbI + A.nbU + A.nbD), blk(256);
  const bool wide = apply_wide(A);
  if (A.p.table_dtype == TLSAN_TABLE_BF16) {
    if (wide) hipLaunchKernelGGL((k_update_lazy<true, TLSAN_TABLE_BF16>), g1, blk, 0, hs, A, nbC16);
    else hipLaunchKernelGGL((k_update_lazy<false, TLSAN_TABLE_BF16>), g1, blk, 0, hs, A, nbC16);
  } else {
    if (wide) hipLaunchKernelGGL((k_update_lazy<true, TLSAN_TABLE_F32>), g1, blk, 0, hs, A, nbC16);
    else hipLaunchKernelGGL((k_update_lazy<false, TLSAN_TABLE_F32>), g1, blk, 0, hs, A, nbC16);
  }
  CHECK_LAUNCH("k_update_lazy");
  return TLSAN_OK;
}

static int launch_apply(int mode, ApplyArgs A, bool with_dense, hipStream_t hs, bool lazy = false) {
  const dim3 g1(A.nbC + A.nbI + A.nbU + (with_dense ? A.nbD : 0)), blk(256);
  const bool wide = apply_wide(A);
  const bool bf16 = A.p.table_dtype == TLSAN_TABLE_BF16;
#define AP_LAUNCH(M, LZ)                                                                                     \
  do {                                                                                                       \
    if (bf16) {                                                                                              \
      if (wide) hipLaunchKernelGGL((k_apply<M, LZ, true, TLSAN_TABLE_BF16>), g1, blk, 0, hs, A);             \
      else hipLaunchKernelGGL((k_apply<M, LZ, false, TLSAN_TABLE_BF16>), g1, blk, 0, hs, A);                 \
    } else {                                                                                                 \
      if (wide) hipLaunchKernelGGL((k_apply<M, LZ, true>), g1, blk, 0, hs, A);                               \
      else hipLaunchKernelGGL((k_apply<M, LZ, false>), g1, blk, 0, hs, A);                                   \
    }                                                                                                        \
  } while (0)
  switch (mode) {
    case AP_UPDATE: if (lazy) AP_LAUNCH(AP_UPDATE, true); else AP_LAUNCH(AP_UPDATE, false); break;
    case AP_GRADS: AP_LAUNCH(AP_GRADS, false); break;
    case AP_SUMSQ: AP_LAUNCH(AP_SUMSQ, false); break;
    default: AP_LAUNCH(AP_ROWNORM, false); break;
  }
#undef AP_LAUNCH
  CHECK_LAUNCH("k_apply");
  return TLSAN_OK;
}

// exclusive scans of up to three count arrays in one launch (two for large tables, see k_index_scan);
// bsum: scratch of >= nscan packed sums, or NULL (then always the single launch)
static int launch_scan(ScanArgs& sa, int nscan, long long* bsum, hipStream_t hs) {
  int big = 0;
  for (int k = 0; k < 3; ++k) {
    const int nb = (k < 2 ? sa.blk0[k + 1] : nscan) - sa.blk0[k];
    if (nb > SCAN_TWO_LEVEL_BLOCKS) big = 1;
  }
  sa.bsum = nullptr;
  if (big && bsum) {
    sa.bsum = bsum;
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nscan), dim3(1024), 0, hs, sa);
    CHECK_LAUNCH("k_scan_block_sums");
  }
  sa.bal.blk = nscan;
  sa.us.blk = nscan + (sa.bal.perm ? 1 : 0);
  sa.is.blk = sa.us.blk + (sa.us.u ? 1 : 0);       // (the finishing blocks of the item side's counting sort come last)
  hipLaunchKernelGGL(k_index_scan, dim3(sa.is.blk + (sa.is.on ? sa.is.nfin : 0)), dim3(1024), 0, hs, sa);
  CHECK_LAUNCH("k_index_scan");
  return TLSAN_OK;
}

static int scan_compact_impl(const int32_t* cnt, int32_t n, int32_t* prefix, int32_t* uniq, int32_t* n_uniq, long long* bsum,
                             hipStream_t hs);

// static CSR category -> items from p->item_cate (counting sort with the generic index kernels)
static int build_cate_csr(const tlsan_dims* d, const tlsan_params* p, const St& st, hipStream_t hs) {
  const int I = d->item_count, C = d->cate_count;
  GIdxArgs gi;
  gi.dest = p->item_cate; gi.n = I; gi.nrows = C; gi.cnt = st.cate_cnt; gi.cur = st.cate_cur; gi.list = st.cate_items;
  if (I <= CSR_SMALL_MAXN && C <= CSR_SMALL_MAXROWS) {   // one launch (the sharded step rebuilds this every step)
    hipLaunchKernelGGL(k_csr_small, dim3(CSR_SMALL_WG), dim3(1024), 0, hs, gi, st.cate_off);
    CHECK_LAUNCH("k_csr_small");
    return TLSAN_OK;
  }
  if (hipMemsetAsync(st.cate_cnt, 0, 4 * (size_t)C, hs) != hipSuccess) return fail(TLSAN_E_LAUNCH, "memset cate_cnt");
  hipLaunchKernelGGL(k_gidx<false>, dim3((I + 255) / 256), dim3(256), 0, hs, gi);
  CHECK_LAUNCH("k_gidx<count>");
  ScanArgs sa;
  memset(&sa, 0, sizeof(sa));
  sa.cnt[0] = st.cate_cnt; sa.off[0] = st.cate_off; sa.cur[0] = st.cate_cur; sa.n[0] = C;
  const int nscan = (C + 4095) / 4096;
  sa.blk0[0] = 0; sa.blk0[1] = nscan; sa.blk0[2] = nscan;
  hipLaunchKernelGGL(k_index_scan, dim3(nscan), dim3(1024), 0, hs, sa);
  CHECK_LAUNCH("k_index_scan");
  hipLaunchKernelGGL(k_gidx<true>, dim3((I + 255) / 256), dim3(256), 0, hs, gi);
  CHECK_LAUNCH("k_gidx<fill>");
  return TLSAN_OK;
}

int tlsan_sync_derived(const tlsan_dims* d, const tlsan_params* p, void* stream) {
  Shape s;
  int rc = shape_of(d, &s);
  if (rc) return rc;
  if ((rc = check_params(p))) return rc;
  tlsan_dense_layout L;
  tlsan_dense_layout_of(d, &L);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_transpose_K, dim3((s.D * s.D + 255) / 256), dim3(256), 0, st, p->dense + L.K, p->dense_KT, s.D);
  CHECK_LAUNCH("k_transpose_K");
  return TLSAN_OK;
}

int tlsan_state_init(const tlsan_dims* d, const tlsan_params* p, void* state, void* stream) {
  Shape s;
  int rc = shape_of(d, &s);
  if (rc) return rc;
  if ((rc = check_params(p))) return rc;
  if (!state) return fail(TLSAN_E_WORKSPACE, "state is NULL");
  if ((rc = tlsan_sync_derived(d, p, stream))) return rc;
  St st;
  carve_state(d, (char*)state, &st);
  hipStream_t hs = (hipStream_t)stream;
  if (hipMemsetAsync(state, 0, st.bytes, hs) != hipSuccess) return fail(TLSAN_E_LAUNCH, "memset state");
  tlsan_dense_layout L;
  tlsan_dense_layout_of(d, &L);
  Ws w;
  memset(&w, 0, sizeof(w));
  ApplyArgs A;
  fill_apply(A, d, s, p, nullptr, nullptr, w, st, L);
  static const float one = 1.0f;  // table scale P = 1
  if (hipMemcpyAsync(&st.hdr->P, &one, sizeof(float), hipMemcpyHostToDevice, hs) != hipSuccess) return fail(TLSAN_E_LAUNCH, "init P");
  if ((rc = build_cate_csr(d, p, st, hs))) return rc;
  if ((rc = launch_apply(AP_SUMSQ, A, false, hs))) return rc;
  hipLaunchKernelGGL(k_reduce_double, dim3(1), dim3(256), 0, hs, st.S_part, st.nbI + st.nbU + st.nbC, st.S_total);
  CHECK_LAUNCH("k_reduce_double");
  // (per-step CHANGES of the sum travel as tagged records in S_delta, folded by the next step's k_dense_finalize)
  if (hipMemsetAsync(st.S_part, 0, 8 * (size_t)(st.nbI + st.nbU + st.nbC), hs) != hipSuccess) return fail(TLSAN_E_LAUNCH, "memset S_part");
  return TLSAN_OK;
}

const float* tlsan_state_scale(const void* state) { return (const float*)state; }

int tlsan_state_renorm(const tlsan_dims* d, const tlsan_params* p, void* state, void* stream) {
  Shape s;
  int rc = shape_of(d, &s);
  if (rc) return rc;
  if ((rc = check_params(p))) return rc;
  if (!state) return fail(TLSAN_E_WORKSPACE, "state is NULL");
  St st;
  carve_state(d, (char*)state, &st);
  const tlsan_params q = norm_params(p, d);
  hipStream_t hs = (hipStream_t)stream;
  const int dt = q.table_dtype;
  // (changes of the sum of squares the last update left as records: part of St before St is rescaled)
  hipLaunchKernelGGL(k_fold_delta, dim3(1), dim3(256), 0, hs, st.S_delta, st.nbI + st.nbU + st.nbC + AP_HOT_CAP, st.hdr, st.S_total);
  hipLaunchKernelGGL(k_scale_table, dim3(1024), dim3(256), 0, hs, q.item_emb, d->item_count, d->d_item, q.ld_item, st.hdr, dt, 0x1b873593u);
  hipLaunchKernelGGL(k_scale_table, dim3(1024), dim3(256), 0, hs, q.user_emb, d->user_count, d->d_item, q.ld_user, st.hdr, dt, 0xcc9e2d51u);
  hipLaunchKernelGGL(k_scale_table, dim3(256), dim3(256), 0, hs, q.usert_emb, d->user_count, d->Ls, q.ld_usert, st.hdr, TLSAN_TABLE_F32, 0u);
  hipLaunchKernelGGL(k_scale_table, dim3(64), dim3(256), 0, hs, q.cate_emb, d->cate_count, d->d_cate, d->d_cate, st.hdr, dt, 0xe6546b64u);
  hipLaunchKernelGGL(k_renorm_commit, dim3(1), dim3(1), 0, hs, st.hdr);
  CHECK_LAUNCH("tlsan_state_renorm");
  if (dt != TLSAN_TABLE_F32) {
    // the rounded values no longer scale exactly with P: take the sum of squares from what is stored
    tlsan_dense_layout L;
    tlsan_dense_layout_of(d, &L);
    Ws w;
    memset(&w, 0, sizeof(w));
    ApplyArgs A;
    fill_apply(A, d, s, p, nullptr, nullptr, w, st, L);
    // (per-step changes still pending in S_part are already part of the stored values: overwritten)
    if ((rc = launch_apply(AP_SUMSQ, A, false, hs))) return rc;
    hipLaunchKernelGGL(k_reduce_double, dim3(1), dim3(256), 0, hs, st.S_part, st.nbI + st.nbU + st.nbC, st.S_total);
    CHECK_LAUNCH("k_reduce_double");
    if (hipMemsetAsync(st.S_part, 0, 8 * (size_t)(st.nbI + st.nbU + st.nbC), hs) != hipSuccess) return fail(TLSAN_E_LAUNCH, "memset S_part");
  }
  return TLSAN_OK;
}

int tlsan_state_reindex(const tlsan_dims* d, const tlsan_params* p, void* state, void* stream) {
  Shape s;
  int rc = shape_of(d, &s);
  if (rc) return rc;
  if ((rc = check_params(p))) return rc;
  if (!state) return fail(TLSAN_E_WORKSPACE, "state is NULL");
  St st;
  carve_state(d, (char*)state, &st);
  hipStream_t hs = (hipStream_t)stream;
  const size_t skip = al(sizeof(StateHdr));  // keep P / St
  if (hipMemsetAsync((char*)state + skip, 0, st.bytes - skip, hs) != hipSuccess) return fail(TLSAN_E_LAUNCH, "memset state");
  return build_cate_csr(d, p, st, hs);
}

int tlsan_state_recategorize(const tlsan_dims* d, const tlsan_params* p, void* state, void* stream) {
  Shape s;
  int rc = shape_of(d, &s);
  if (rc) return rc;
  if ((rc = check_params(p))) return rc;
  if (!state) return fail(TLSAN_E_WORKSPACE, "state is NULL");
  St st;
  carve_state(d, (char*)state, &st);
  return build_cate_csr(d, p, st, (hipStream_t)stream);
}

static int launch_fwd(const Shape& s, bool train, const FwdArgs& a, hipStream_t hs, int grp = 0) {
  int grid = a.ngroups < 4096 ? a.ngroups : 4096;
  if (train && a.fuse_dk) grid = fwd_train_grid(a.ngroups);
  hipError_t e;
  const bool lstream = streamed(a.Ls);  // long windows are streamed, short ones stay in registers
  LaunchEvents ev = {nullptr, nullptr};
  if (train) ev = prof_kernel_events();
  if (train && s.D == 128 && grp == 8) e = tlsan_launch_fwd_bwd_d128w4(a, grid, hs, ev);
  else if (s.D == 64) e = tlsan_launch_fwd_bwd_d64(train, lstream, a, grid, hs, ev);
  else if (s.D == 128) e = tlsan_launch_fwd_bwd_d128(train, lstream, a, grid, hs, ev);
  else e = tlsan_launch_fwd_bwd_d256(train, lstream, a, grid, hs, ev);
  if (e == hipErrorNotSupported)
    return fail(TLSAN_E_UNSUPPORTED, "dropout > 0 is built for train steps only (and not for the 8-sample workgroup form)");
  if (e != hipSuccess) return fail(TLSAN_E_LAUNCH, "k_fwd_bwd: %s", hipGetErrorString(e));
  return TLSAN_OK;
}

static void fill_fwd(FwdArgs& a, const tlsan_dims* d, const Shape& s, const tlsan_params* p, const tlsan_batch* b,
                     const Ws& w, const tlsan_dense_layout& L) {
  memset(&a, 0, sizeof(a));
  a.p = norm_params(p, d);
  a.b = *b;
  a.lay = L;
  a.Ls = d->Ls; a.di = d->d_item; a.dc = d->d_cate;
  a.ngroups = (b->B + s.NSB - 1) / s.NSB;  // (training: overridden by run_backward)
  a.inv_B = 1.0f / (float)b->B;
  a.stamps = g_stamps;
}

int tlsan_forward(const tlsan_dims* d, const tlsan_params* p, const tlsan_batch* b, float* logits_i, float* logits_j,
                  float* u_t, void* ws, size_t ws_bytes, void* stream) {
  return tlsan_forward_att(d, p, b, logits_i, logits_j, u_t, nullptr, nullptr, ws, ws_bytes, stream);
}

int tlsan_forward_att(const tlsan_dims* d, const tlsan_params* p, const tlsan_batch* b, float* logits_i, float* logits_j,
                      float* u_t, float* att0, float* att1, void* ws, size_t ws_bytes, void* stream) {
  Shape s;
  int rc = shape_of(d, &s);
  if (rc) return rc;
  if ((rc = check_params(p))) return rc;
  if ((rc = check_batch(d, b, false))) return rc;
  (void)ws; (void)ws_bytes;
  tlsan_dense_layout L;
  tlsan_dense_layout_of(d, &L);
  Ws w;
  memset(&w, 0, sizeof(w));
  FwdArgs a;
  fill_fwd(a, d, s, p, b, w, L);
  a.logits_i = logits_i;
  a.logits_j = logits_j;
  a.u_t = u_t;
  a.att0 = att0; a.att1 = att1;
  return launch_fwd(s, false, a, (hipStream_t)stream);
}

// destination index of a batch into slot k: use counts per destination row -> first sorted position
// of every row (+ records of the used rows)
static int build_index(const tlsan_dims* d, const tlsan_batch* b, const int32_t* item_cate, const St& st, int k, hipStream_t hs,
                       bool sparse_users = false) {
  const bool cseg = cate_seg(d, b);
  if (cseg && !item_cate) return fail(TLSAN_E_BADARG, "tables with %d categories count item uses per category: item_cate is NULL", d->cate_count);
  int rc;
  CountArgs ca;
  memset(&ca, 0, sizeof(ca));
  ca.b = *b; ca.Ls = d->Ls;
  ca.n_hot = &st.hdr->n_hot[k];
  ca.cnt_item = st.cnt_item[k]; ca.cnt_user = st.cnt_user[k]; ca.cnt_uc = st.cnt_uc[k];
  ca.item_cate = cseg ? item_cate : nullptr; ca.cseg = cseg ? 1 : 0;
  ca.ncate = d->cate_count;
  ca.flag_user = st.flag_user[k];
  // a table's side from a sort of the batch's ids when only the used rows are wanted and the table is large: the user
  // table (UsortArgs); the item table (IsortArgs) when a lazy-SGD step with category segments consumes the index (it
  // reaches item offsets / cursors through ids and records only)
  const int nthr = b->B * (d->Ls + b->Sn + 1);
  const bool usort = sparse_users && b->B <= USORT_MAX && d->user_count >= isort_min_rows();
  const bool isort = sparse_users && cseg && nthr <= ISORT_MAX_SLOTS && isort_rows_ok(d->item_count);
  ca.skip_users = usort ? 1 : 0;
  ScanArgs sa;
  memset(&sa, 0, sizeof(sa));
  if (isort) {
    IsortArgs& ia = sa.is;
    ia.on = 1;
    ia.b = *b; ia.Ls = d->Ls;
    ia.nbu = (b->B + 1023) / 1024;   // (k_count's sample blocks: k_count does not run)
    ia.n = d->item_count;
    ia.shift = 0;   // few, full buckets: up to IS_BSZ ids each, at least 64 of them
    while (((ia.n - 1) >> ia.shift) >= IS_MAXB || ((1 << ia.shift) < IS_BSZ && ((ia.n - 1) >> ia.shift) >= 64)) ++ia.shift;
    ia.nb = ((ia.n - 1) >> ia.shift) + 1;
    ia.nslots = nthr;
    ia.nblk = (nthr + IS_BLK_SLOTS - 1) / IS_BLK_SLOTS;
    ia.bh = st.is_bh[k]; ia.ids = st.is_ids[k]; ia.bstart = st.is_bstart[k]; ia.nd = st.is_nd[k]; ia.tmp = st.is_tmp[k];
    ia.cur = st.cur_item[k]; ia.off = st.off_item[k]; ia.urec = st.urec_item[k];
    ia.n_uniq = &st.hdr->n_uniq[k][0]; ia.hot_n = &st.hdr->n_hot[k]; ia.hot_list = st.hot_list[k];
    ia.item_cate = item_cate; ia.cnt_uc = st.cnt_uc[k];
    ia.nfin = (ia.nb + 15) / 16;
    hipLaunchKernelGGL(k_isort_hist, dim3(ia.nbu + ia.nblk), dim3(1024), 0, hs, ia, ca);
    CHECK_LAUNCH("k_isort_hist");
    hipLaunchKernelGGL(k_isort_scatter, dim3(ia.nblk), dim3(1024), 0, hs, ia);
    CHECK_LAUNCH("k_isort_scatter");
    UsortArgs ua;   // (the user side's sort rides in this launch: as long as a bucket block, and needed by nothing before the step)
    memset(&ua, 0, sizeof(ua));
    if (usort) {
      ua.u = b->u; ua.B = b->B; ua.U = d->user_count;
      ua.cur = st.cur_user[k]; ua.off = st.off_user[k]; ua.urec = st.urec_user[k]; ua.n_uniq = &st.hdr->n_uniq[k][1];
    }
    hipLaunchKernelGGL(k_isort_bucket, dim3(ia.nb + (usort ? 1 : 0)), dim3(1024), 0, hs, ia, ua);
    CHECK_LAUNCH("k_isort_bucket");
  } else {
    hipLaunchKernelGGL(k_count, dim3((b->B + 255) / 256 + (nthr + 255) / 256), dim3(256), 0, hs, ca);
    CHECK_LAUNCH("k_count");
  }
  sa.cnt[0] = st.cnt_item[k]; sa.cnt[1] = st.cnt_uc[k]; sa.cnt[2] = st.cnt_user[k];
  sa.off[0] = st.off_item[k]; sa.off[1] = st.off_uc[k]; sa.off[2] = st.off_user[k];
  sa.cur[0] = st.cur_item[k]; sa.cur[1] = st.cur_uc[k]; sa.cur[2] = st.cur_user[k];
  sa.n[0] = isort ? 0 : d->item_count; sa.n[1] = d->cate_count; sa.n[2] = d->user_count;
  sa.blk0[0] = 0;
  sa.blk0[1] = (sa.n[0] + 4095) / 4096;
  sa.blk0[2] = sa.blk0[1] + (sa.n[1] + 4095) / 4096;
  if (usort) {
    sa.n[2] = 0;
    if (!isort) {   // (with the items sorted as well, k_isort_bucket's launch had it)
      sa.us.u = b->u; sa.us.B = b->B; sa.us.U = d->user_count;
      sa.us.cur = st.cur_user[k]; sa.us.off = st.off_user[k]; sa.us.urec = st.urec_user[k]; sa.us.n_uniq = &st.hdr->n_uniq[k][1];
    }
  }
  const int nscan = sa.blk0[2] + (sa.n[2] + 4095) / 4096;
  sa.urec[0] = st.urec_item[k]; sa.urec[2] = st.urec_user[k];
  sa.hot_n[0] = &st.hdr->n_hot[k]; sa.hot_list[0] = st.hot_list[k];
  sa.total[0] = sa.total[1] = sa.total[2] = 1;
  sa.flag[2] = st.flag_user[k];
  // (sa.bs_ticket = st.scan_ticket + k: the sums scanned by the last block of k_scan_block_sums, one prefix read per scan
  //  block -- measured slower: 3663 publishing atomics on consecutive words, 134 -> 162 us/step at 10 M / 5 M rows)
  // the user table of a lazy-L2 SGD step: its consumers reach off / cur through the batch's ids or the used-row records
  // only (the dense sweeps and tlsan_grads read the offsets of every row)
  // (category segments: nothing walks the item offsets per category either -- 5 M items: 40 MB of writes per step less)
  sa.sparse = sparse_users ? ((1 << 2) | (cseg ? 1 : 0)) : 0;
  sa.n_uniq[0] = &st.hdr->n_uniq[k][0]; sa.n_uniq[1] = nullptr; sa.n_uniq[2] = &st.hdr->n_uniq[k][1];
  // (no ranking block for a launch that will not read it -- the 8-sample workgroups of small d = 128 batches take
  //  the batch's own order; the index is built ahead of the step's hyper-parameters, so the groups are sized for a launch
  //  without dropout, and run_backward asks the same question)
  Shape shp;
  if ((rc = shape_of(d, &shp))) return rc;
  if (balanced(d, b) && train_group(shp, d, b, nullptr) == shp.NSB) {
    sa.bal.sl = b->sl; sa.bal.sl_new = b->sl_new;
    sa.bal.B = b->B; sa.bal.Ls = d->Ls; sa.bal.Sn = b->Sn;
    sa.bal.by_window = streamed(d->Ls) ? 1 : (bal_reg_mode() == 2 ? -1 : 0);
    sa.bal.perm = st.perm[k];
  }
  if ((rc = launch_scan(sa, nscan, st.scan_bsum[k], hs))) return rc;
  if (uc_by_list(d, b)) {
    hipLaunchKernelGGL(k_uc_fill, dim3((b->B + 255) / 256), dim3(256), 0, hs, b->u_cate, b->B, d->cate_count, st.cur_uc[k], st.uc_list[k]);
    CHECK_LAUNCH("k_uc_fill");
  }
  return TLSAN_OK;
}

// shared front half of train_step / grads: index build, fused fwd+bwd, dense-grad reduction
static int run_backward(const tlsan_dims* d, const Shape& s, const tlsan_params* p, const tlsan_batch* b,
                        const tlsan_hparams* hp, bool update, const tlsan_step_out* out, const Ws& w, const St& st,
                        const tlsan_dense_layout& L, hipStream_t hs, const ApplyArgs* presum = nullptr,
                        float* gd_out = nullptr, bool sparse_index = false, bool spec = false) {
  const bool commit = update && hp->l2_mode == TLSAN_L2_LAZY && !spec;
  const int k = hp->index_slot;
  int rc;
  prof_mark(0, hs);
  if (!hp->index_prebuilt && (rc = build_index(d, b, p->item_cate, st, k, hs, presum != nullptr || sparse_index))) return rc;
  // --- fused forward + backward
  FwdArgs a;
  fill_fwd(a, d, s, p, b, w, L);
  a.logits_i = (out && out->logits) ? out->logits : w.logits;
  if (out && out->started) { a.started = out->started; a.started_val = out->started_value; }
  a.Gi = w.Gi; a.Gb = w.Gb; a.Gu = w.Gu; a.Gc = w.Gc; a.WU = w.WU;
  a.cur_item = st.cur_item[k]; a.cur_user = st.cur_user[k]; a.cur_uc = st.cur_uc[k];
  a.uc_by_sample = uc_by_list(d, b) ? 1 : 0;
  a.cseg = cate_seg(d, b) ? 1 : 0;
  a.gLong = w.gLong; a.gDB = w.gDB; a.gStat = w.gStat; a.partials = w.partials; a.Kp = w.Kp;
  if (hp->dropout != 0.0f) {
    if (!(hp->dropout > 0.0f && hp->dropout < 1.0f)) return fail(TLSAN_E_BADARG, "dropout must be in [0, 1)");
    const float keep = (float)(1.0 - (double)hp->dropout);
    const double t = (double)keep * 4294967296.0;
    a.drop_thr = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    a.drop_inv = 1.0f / keep;
    a.drop_seed = hp->dropout_seed;
    a.drop_sample0 = hp->dropout_sample0;
  }
  const int grp = train_group(s, d, b, hp);  // samples per workgroup pass of the fused kernel (= per partial record)
  // (the ranking deals the batch out in groups of 16: the 8-sample workgroups of small d = 128 batches take the batch's order)
  a.perm = balanced(d, b) && grp == s.NSB && train_group(s, d, b, nullptr) == s.NSB ? st.perm[k] : nullptr;   // (what build_index left)
  a.ngroups = (b->B + grp - 1) / grp;
  a.fuse_dk = fused_dk(s.D, a.ngroups) ? 1 : 0;
  prof_mark(1, hs);
  if ((rc = launch_fwd(s, true, a, hs, grp))) return rc;
  prof_mark(2, hs);
  const int nsplit = a.fuse_dk ? fwd_train_grid(a.ngroups) : dk_nsplit(b->B, s.D);   // dK partials the finalize sums
  // --- dense-parameter gradients (D <= 128: the dK partials were left by k_fwd_bwd, one per workgroup)
  if (!a.fuse_dk) {
    const int spw = dk_spw(b->B, s.D), nq = (s.D / 64) * (s.D / 64);
    const dim3 grid(nq * nsplit), blk(DK_WAVES * 64);
#define DK_LAUNCH(DD)                                                                                           \
  do {                                                                                                          \
    (void)hipFuncSetAttribute((const void*)k_dk_partial<DD>, hipFuncAttributeMaxDynamicSharedMemorySize, DK_SMEM_BYTES); \
    hipLaunchKernelGGL(k_dk_partial<DD>, grid, blk, DK_SMEM_BYTES, hs, w.gLong, w.gDB, b->B, spw, w.Kp);        \
  } while (0)
    if (s.D == 64) DK_LAUNCH(64);
    else if (s.D == 128) DK_LAUNCH(128);
    else DK_LAUNCH(256);
#undef DK_LAUNCH
  }
  CHECK_LAUNCH("k_dk_partial");
  prof_mark(3, hs);
  FinArgs f;
  memset(&f, 0, sizeof(f));
  f.lay = L; f.partials = w.partials; f.nrec = (b->B + grp - 1) / grp; f.Kp = w.Kp; f.nsplit = nsplit;
  f.gd = gd_out ? gd_out : w.gd; f.sqd = w.sqd; f.scal = w.scal;
  f.S_delta = st.S_delta; f.n_spart = st.nbI + st.nbU + st.nbC + AP_HOT_CAP; f.S_total = st.S_total;
  f.hdr = st.hdr; f.lr = hp->lr; f.reg = hp->reg; f.clip = hp->clip; f.inv_B = 1.0f / (float)b->B;
  f.norm_mode = hp->norm_mode; f.commit = commit ? 1 : 0; f.count_step = update ? 1 : 0;
  f.out_loss = out ? out->loss : nullptr;
  f.out_gnorm = out ? out->gnorm : nullptr;
  f.out_sq = out ? out->sq_rows : nullptr;
  if (presum && spec) {
    // the one-pass form: the row workgroups UPDATE beside the finalize, with clip coefficient 1 (k_finalize_update; the commit
    // and -- after a clipped step -- the correction follow in k_spec_commit, tlsan_train_step_opt)
    ApplyArgs A = *presum;
    lazy_blocks(A, b->B, b->Sn);
    A.nbH = AP_HOT_CAP;      // hot item rows: a workgroup each, leading the row workgroups (they return at once where there are none)
    f.count_step = 0; f.spec = 1;
    const bool shared = A.csplit > 1;   // (lazy_one_pass form 2: A.nbC = the commit launch's category blocks; this launch carries C * csplit)
    // (item-row workgroups launched: at most SPEC_ITEM_BLOCKS -- ApplyArgs.nbI_l; TLSAN_SPEC_ITEM_BLOCKS=<n>, 0: all, for A/B)
    static const int item_cap = [] { const char* e = getenv("TLSAN_SPEC_ITEM_BLOCKS"); return e ? atoi(e) : SPEC_ITEM_BLOCKS; }();
    // (the shared-category form: the same cap -- Movies-TV's 1787 blocks stay below it; fewer, 1024 / 640 / 384, measured a
    //  loss there: profiles/r06_ab_hot_cate.txt; TLSAN_SPEC_ITEM_BLOCKS_SHARED=<n> for A/B)
    static const int item_cap_sh = [] { const char* e = getenv("TLSAN_SPEC_ITEM_BLOCKS_SHARED"); return e ? atoi(e) : SPEC_ITEM_BLOCKS; }();
    const int cap_here = shared ? item_cap_sh : item_cap;
    A.nbI_l = (cap_here > 0 && A.nbI > cap_here) ? cap_here : 0;
    static const int ufirst = [] { const char* e = getenv("TLSAN_SPEC_UFIRST"); return e ? atoi(e) : 1; }();
    A.ufirst = (ufirst && apply_wide(A) && !shared) ? 1 : 0;
    const dim3 grid(w.nfin + 1 + A.nbH + (shared ? A.C * A.csplit : A.nbC) + (A.nbI_l > 0 ? A.nbI_l : A.nbI) + A.nbU);
    static const int low_env = [] { const char* e = getenv("TLSAN_SPEC_LOWOCC"); return e ? atoi(e) : -1; }();   // (A/B: 0 / 1 force it)
    const bool wide = apply_wide(A) && !shared, bf16 = A.p.table_dtype == TLSAN_TABLE_BF16, low = low_env < 0 ? tables_in_hbm(d) : low_env != 0;
#define FU_LAUNCH(DD, HH)                                                                                                            \
  do {                                                                                                                               \
    if (shared) {                                                                                                                    \
      if (bf16) {                                                                                                                    \
        if (low) hipLaunchKernelGGL((k_finalize_update<DD, HH, false, TLSAN_TABLE_BF16, true, true>), grid, dim3(256), 0, hs, f, w.nbK, w.nbS, A);  \
        else hipLaunchKernelGGL((k_finalize_update<DD, HH, false, TLSAN_TABLE_BF16, false, true>), grid, dim3(256), 0, hs, f, w.nbK, w.nbS, A);     \
      } else {                                                                                                                       \
        if (low) hipLaunchKernelGGL((k_finalize_update<DD, HH, false, TLSAN_TABLE_F32, true, true>), grid, dim3(256), 0, hs, f, w.nbK, w.nbS, A);   \
        else hipLaunchKernelGGL((k_finalize_update<DD, HH, false, TLSAN_TABLE_F32, false, true>), grid, dim3(256), 0, hs, f, w.nbK, w.nbS, A);      \
      }                                                                                                                              \
    } else if (bf16) {                                                                                                                      \
      if (wide) hipLaunchKernelGGL((k_finalize_update<DD, HH, true, TLSAN_TABLE_BF16>), grid, dim3(256), 0, hs, f, w.nbK, w.nbS, A);  \
      else if (low) hipLaunchKernelGGL((k_finalize_update<DD, HH, false, TLSAN_TABLE_BF16, true>), grid, dim3(256), 0, hs, f, w.nbK, w.nbS, A); \
      else hipLaunchKernelGGL((k_finalize_update<DD, HH, false, TLSAN_TABLE_BF16>), grid, dim3(256), 0, hs, f, w.nbK, w.nbS, A);      \
    } else {                                                                                                                         \
      if (wide) hipLaunchKernelGGL((k_finalize_update<DD, HH, true, TLSAN_TABLE_F32>), grid, dim3(256), 0, hs, f, w.nbK, w.nbS, A);   \
      else if (low) hipLaunchKernelGGL((k_finalize_update<DD, HH, false, TLSAN_TABLE_F32, true>), grid, dim3(256), 0, hs, f, w.nbK, w.nbS, A); \
      else hipLaunchKernelGGL((k_finalize_update<DD, HH, false, TLSAN_TABLE_F32>), grid, dim3(256), 0, hs, f, w.nbK, w.nbS, A);       \
    }                                                                                                                                \
  } while (0)
    if (shared && s.D > 128) return fail(TLSAN_E_UNSUPPORTED, "shared categories in the one-pass update: d <= 128");   // (lazy_one_pass never asks)
    if (s.D == 64) FU_LAUNCH(64, 8);
    else if (s.D == 128) FU_LAUNCH(128, 16);
    else FU_LAUNCH(256, 32);
#undef FU_LAUNCH
    CHECK_LAUNCH("k_finalize_update");
    prof_mark(4, hs);
    return TLSAN_OK;
  }
  if (presum) {
    // lazy update: the exact row sums of the apply pass share the launch (they wait for nothing it produces)
    ApplyArgs A = *presum;
    lazy_blocks(A, b->B, b->Sn);
    A.nbC = A.cseg ? (A.C + AP_ROWS_PB - 1) / AP_ROWS_PB : A.C * A.csplit;
    A.nbH = AP_HOT_CAP;   // hot item rows: a workgroup each, leading the grid
    const dim3 grid(w.nfin + 1 + A.nbH + A.nbC + A.nbI + A.nbU);
    // (the row-sum launch covers user rows of up to 256 floats in two passes of its narrow form -- 92 registers, five
    //  workgroups per CU, instead of 135 and three; the sharded step's fused rows keep the wide form.  d = 128 with 90-entry
    //  windows: Movies-TV shape 106.6 -> 104.3 us/step, with 673 categories 116.2 -> 106.2: profiles/r04_presum_narrow_ab.md)
    const bool wide = A.di > 64 || A.dc > 64 || (A.WU > 128 && A.presum_rows != 0);
#define FP_LAUNCH(DD, HH)                                                                                         \
  do {                                                                                                            \
    if (A.csplit > 1) {                                                                                           \
      if (wide) hipLaunchKernelGGL((k_finalize_presum<DD, HH, true, true>), grid, dim3(256), 0, hs, f, w.nbK, w.nbS, A);  \
      else hipLaunchKernelGGL((k_finalize_presum<DD, HH, false, true>), grid, dim3(256), 0, hs, f, w.nbK, w.nbS, A);      \
    } else if (wide) hipLaunchKernelGGL((k_finalize_presum<DD, HH, true>), grid, dim3(256), 0, hs, f, w.nbK, w.nbS, A);   \
    else hipLaunchKernelGGL((k_finalize_presum<DD, HH, false>), grid, dim3(256), 0, hs, f, w.nbK, w.nbS, A);       \
  } while (0)
    if (s.D == 64) FP_LAUNCH(64, 8);
    else if (s.D == 128) FP_LAUNCH(128, 16);
    else FP_LAUNCH(256, 32);
#undef FP_LAUNCH
    CHECK_LAUNCH("k_finalize_presum");
    prof_mark(4, hs);
    return TLSAN_OK;
  }
  if (s.D == 64) hipLaunchKernelGGL((k_dense_finalize<64, 8>), dim3(w.nfin + 1), dim3(256), 0, hs, f, w.nbK, w.nbS);
  else if (s.D == 128) hipLaunchKernelGGL((k_dense_finalize<128, 16>), dim3(w.nfin + 1), dim3(256), 0, hs, f, w.nbK, w.nbS);
  else hipLaunchKernelGGL((k_dense_finalize<256, 32>), dim3(w.nfin + 1), dim3(256), 0, hs, f, w.nbK, w.nbS);
  CHECK_LAUNCH("k_dense_finalize");
  prof_mark(4, hs);
  return TLSAN_OK;
}

static int prep_step(const tlsan_dims* d, Shape* s, const tlsan_params* p, const tlsan_batch* b, const tlsan_hparams* hp,
                     void* state, void* ws, size_t ws_bytes, Ws* w, St* st) {
  int rc = shape_of(d, s);
  if (rc) return rc;
  if ((rc = check_params(p))) return rc;
  if ((rc = check_batch(d, b, true))) return rc;
  if (!hp) return fail(TLSAN_E_BADARG, "hparams is NULL");
  if (hp->l2_mode != TLSAN_L2_DENSE && hp->l2_mode != TLSAN_L2_LAZY) return fail(TLSAN_E_BADARG, "l2_mode");
  if (hp->index_slot < 0 || hp->index_slot >= TLSAN_INDEX_SLOTS) return fail(TLSAN_E_BADARG, "index_slot must be 0 .. %d", TLSAN_INDEX_SLOTS - 1);
  if (hp->l2_mode == TLSAN_L2_LAZY) {
    if (hp->norm_mode != TLSAN_NORM_TF18) return fail(TLSAN_E_UNSUPPORTED, "TLSAN_L2_LAZY supports norm_mode TF18 only");
    if (p->scale != tlsan_state_scale(state)) return fail(TLSAN_E_BADARG, "TLSAN_L2_LAZY needs params->scale == tlsan_state_scale(state)");
  }
  if (hp->norm_mode != TLSAN_NORM_TF18 && hp->norm_mode != TLSAN_NORM_DEDUP) return fail(TLSAN_E_BADARG, "norm_mode");
  if (!state || !ws) return fail(TLSAN_E_WORKSPACE, "state / ws is NULL");
  carve(d, *s, b->B, b->Sn, (char*)ws, w);
  if (w->bytes > ws_bytes) return fail(TLSAN_E_WORKSPACE, "workspace too small: need %zu have %zu", w->bytes, ws_bytes);
  carve_state(d, (char*)state, st);
  return TLSAN_OK;
}

// dedup-norm mode: per-row squared norms of the SUMMED gradients (ROWNORM pass), then the coefficient
static int clip_dedup(const ApplyArgs& A, const tlsan_hparams* hp, const tlsan_step_out* out, const Ws& w,
                      const St& st, const tlsan_batch* b, hipStream_t hs) {
  ApplyArgs R = A;
  R.part_out = w.rownorm_part;
  int rc = launch_apply(AP_ROWNORM, R, false, hs);
  if (rc) return rc;
  hipLaunchKernelGGL(k_clip_dedup, dim3(1), dim3(256), 0, hs, w.rownorm_part, st.nbI + st.nbU + st.nbC, w.sqd, w.nfin,
                     st.hdr, hp->clip, out ? out->gnorm : nullptr);
  CHECK_LAUNCH("k_clip_dedup");
  return TLSAN_OK;
}

int tlsan_batch_pack(const tlsan_packed* set, const int32_t* order, int32_t lo, const tlsan_batch* out, int32_t Ls,
                     int32_t is_test, void* stream) {
  if (!set || !order || !out) return fail(TLSAN_E_BADARG, "tlsan_batch_pack: NULL argument");
  if (!set->u || !set->cate || !set->hist_off || !set->sess_off || !set->target || !set->second)
    return fail(TLSAN_E_BADARG, "tlsan_batch_pack: NULL pointer in the packed set");
  if (out->B < 1 || out->Sn < 0 || Ls < 1 || lo < 0 || lo + out->B > set->n)
    return fail(TLSAN_E_BADARG, "tlsan_batch_pack: samples [%d, %d) outside the set of %d", lo, lo + out->B, set->n);
  if (!out->u || !out->i || !out->hist_i || !out->hist_t || !out->sl || !out->sl_new || !out->u_cate ||
      (out->Sn > 0 && !out->hist_i_new) || (is_test ? !out->j : !out->y))
    return fail(TLSAN_E_BADARG, "tlsan_batch_pack: NULL output array");
  if ((size_t)out->B * (Ls + out->Sn + 1) >= ((size_t)1 << 31)) return fail(TLSAN_E_UNSUPPORTED, "B*S overflows int32");
  PackArgs a;
  a.set = *set; a.order = order; a.lo = lo; a.Ls = Ls; a.is_test = is_test ? 1 : 0; a.out = *out;
  const int nthr = out->B * (Ls + out->Sn + 1);
  hipLaunchKernelGGL(k_batch_pack, dim3((nthr + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  CHECK_LAUNCH("k_batch_pack");
  return TLSAN_OK;
}

int tlsan_batch_index(const tlsan_dims* d, const tlsan_batch* b, const int32_t* item_cate, void* state, int32_t slot, void* stream) {
  Shape s; St st;
  int rc = shape_of(d, &s);
  if (rc) return rc;
  if ((rc = check_batch(d, b, true))) return rc;
  const bool lazy_sgd = (slot & TLSAN_INDEX_FOR_LAZY_SGD) != 0;
  slot &= ~TLSAN_INDEX_FOR_LAZY_SGD;
  if (slot < 0 || slot >= TLSAN_INDEX_SLOTS) return fail(TLSAN_E_BADARG, "index slot must be 0 .. %d", TLSAN_INDEX_SLOTS - 1);
  if (!state) return fail(TLSAN_E_WORKSPACE, "state is NULL");
  carve_state(d, (char*)state, &st);
  return build_index(d, b, item_cate, st, slot, (hipStream_t)stream, lazy_sgd);
}

static int check_slot(const tlsan_params* q, const char* name) {
  if (!q || !q->item_emb || !q->item_b || !q->user_emb || !q->usert_emb || !q->cate_emb || !q->dense)
    return fail(TLSAN_E_BADARG, "tlsan_optimizer: NULL table in %s", name);
  if (q->ld_item % 4 || q->ld_user % 4) return fail(TLSAN_E_UNSUPPORTED, "tlsan_optimizer: row strides of %s must be multiples of 4 floats", name);
  return TLSAN_OK;
}

int tlsan_train_step(const tlsan_dims* d, const tlsan_params* p, const tlsan_batch* b, const tlsan_hparams* hp,
                     const tlsan_step_out* out, void* state, void* ws, size_t ws_bytes, void* stream) {
  return tlsan_train_step_opt(d, p, b, hp, nullptr, out, state, ws, ws_bytes, stream);
}

int tlsan_train_step_opt(const tlsan_dims* d, const tlsan_params* p, const tlsan_batch* b, const tlsan_hparams* hp,
                         const tlsan_optimizer* opt, const tlsan_step_out* out, void* state, void* ws, size_t ws_bytes,
                         void* stream) {
  Shape s; Ws w; St st;
  int rc = prep_step(d, &s, p, b, hp, state, ws, ws_bytes, &w, &st);
  if (rc) return rc;
  const bool other = opt && opt->kind != TLSAN_OPT_SGD;
  if (other) {
    if (opt->kind != TLSAN_OPT_ADAM && opt->kind != TLSAN_OPT_RMSPROP && opt->kind != TLSAN_OPT_ADADELTA)
      return fail(TLSAN_E_BADARG, "tlsan_optimizer: kind %d", opt->kind);
    if (hp->l2_mode != TLSAN_L2_DENSE) return fail(TLSAN_E_UNSUPPORTED, "optimizers other than sgd update every row: l2_mode must be TLSAN_L2_DENSE");
    if ((rc = check_slot(opt->slot1, "slot1")) || (rc = check_slot(opt->slot2, "slot2"))) return rc;
    if (opt->kind == TLSAN_OPT_ADAM && opt->step < 1) return fail(TLSAN_E_BADARG, "tlsan_optimizer: Adam's step counts from 1");
  }
  hipStream_t hs = (hipStream_t)stream;
  tlsan_dense_layout L;
  tlsan_dense_layout_of(d, &L);
  ApplyArgs A;
  fill_apply(A, d, s, p, b, hp, w, st, L);
  if (other) {
    A.opt = opt->kind;
    A.s1 = norm_params(opt->slot1, d); A.s2 = norm_params(opt->slot2, d);
    A.ob1 = opt->beta1; A.ob2 = opt->beta2; A.oeps = opt->epsilon;
    if (opt->kind == TLSAN_OPT_ADAM)  // adam.py: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
      A.oalpha = (float)((double)hp->lr * sqrt(1.0 - pow((double)opt->beta2, opt->step)) / (1.0 - pow((double)opt->beta1, opt->step)));
  }
  static const int spec_on = [] { const char* e = getenv("TLSAN_LAZY_SPEC"); return e ? atoi(e) : 1; }();
  int form = hp->l2_mode == TLSAN_L2_LAZY ? lazy_one_pass(d, b, A) : 0;
  if (form == 2 && !spec_on) form = 0;     // (shared categories: built in the speculative form only)
  if (form != 0) {
    // ONE pass over the used rows (segment sums and the update of a row by the same lanes) instead of row sums beside the
    // finalize + an elementwise update: the summed rows make no round trip through memory (lazy_one_pass says where)
    if (form == 2) {   // shared categories: summed in the first launch (C * csplit workgroups), updated by 16-row blocks of the commit
      category_split(A, d, b);
      A.nbC = (A.C + 15) / 16;
    }
    ApplyArgs A1 = A;
    lazy_blocks(A1, b->B, b->Sn);
    if (spec_on) {   // the row update beside the finalize, with coefficient 1; then the commit (+ the correction of a clipped step)
      if ((rc = run_backward(d, s, p, b, hp, true, out, w, st, L, hs, &A, nullptr, true, true))) return rc;
      A1.nbH = AP_HOT_CAP;
      const int nrow = A1.nbH + (form == 2 ? 0 : A1.nbC) + A1.nbI + A1.nbU;
      const dim3 grid(A1.nbD + (form == 2 ? A1.nbC : 0) + (nrow < SPEC_FIX_BLOCKS ? nrow : SPEC_FIX_BLOCKS));
      const bool wide = apply_wide(A1) && form != 2;
      if (form == 2) {
        if (A1.p.table_dtype == TLSAN_TABLE_BF16) hipLaunchKernelGGL((k_spec_commit<false, TLSAN_TABLE_BF16, true>), grid, dim3(256), 0, hs, A1);
        else hipLaunchKernelGGL((k_spec_commit<false, TLSAN_TABLE_F32, true>), grid, dim3(256), 0, hs, A1);
      } else if (A1.p.table_dtype == TLSAN_TABLE_BF16) {
        if (wide) hipLaunchKernelGGL((k_spec_commit<true, TLSAN_TABLE_BF16>), grid, dim3(256), 0, hs, A1);
        else hipLaunchKernelGGL((k_spec_commit<false, TLSAN_TABLE_BF16>), grid, dim3(256), 0, hs, A1);
      } else {
        if (wide) hipLaunchKernelGGL((k_spec_commit<true, TLSAN_TABLE_F32>), grid, dim3(256), 0, hs, A1);
        else hipLaunchKernelGGL((k_spec_commit<false, TLSAN_TABLE_F32>), grid, dim3(256), 0, hs, A1);
      }
      CHECK_LAUNCH("k_spec_commit");
    } else {         // (A/B: the finalize's whole chain, then one pass over the rows)
      if ((rc = run_backward(d, s, p, b, hp, true, out, w, st, L, hs, nullptr, nullptr, true))) return rc;
      if ((rc = launch_apply(AP_UPDATE, A1, true, hs, true))) return rc;
    }
  } else if (hp->l2_mode == TLSAN_L2_LAZY) {   // row sums beside the finalize, then the short elementwise update
    category_split(A, d, b);
    if ((rc = run_backward(d, s, p, b, hp, true, out, w, st, L, hs, &A))) return rc;
    if ((rc = launch_update_lazy(A, b->B, b->Sn, hs))) return rc;
  } else {
    if ((rc = run_backward(d, s, p, b, hp, true, out, w, st, L, hs))) return rc;
    if (hp->norm_mode == TLSAN_NORM_DEDUP && (rc = clip_dedup(A, hp, out, w, st, b, hs))) return rc;
    if ((rc = launch_apply(AP_UPDATE, A, true, hs))) return rc;
  }
  prof_mark(5, hs);
  prof_step_done();
  return TLSAN_OK;
}

int tlsan_grads(const tlsan_dims* d, const tlsan_params* p, const tlsan_batch* b, const tlsan_hparams* hp,
                const tlsan_grads_out* g, const tlsan_step_out* out, void* state, void* ws, size_t ws_bytes, void* stream) {
  Shape s; Ws w; St st;
  int rc = prep_step(d, &s, p, b, hp, state, ws, ws_bytes, &w, &st);
  if (rc) return rc;
  if (!g || !g->item_emb || !g->item_b || !g->user_emb || !g->usert_emb || !g->cate_emb || !g->dense)
    return fail(TLSAN_E_BADARG, "NULL gradient output");
  hipStream_t hs = (hipStream_t)stream;
  tlsan_dense_layout L;
  tlsan_dense_layout_of(d, &L);
  ApplyArgs A;
  fill_apply(A, d, s, p, b, hp, w, st, L);
  A.go = *g;
  if (A.go.ld_item == 0) A.go.ld_item = d->d_item;
  if (A.go.ld_itemb == 0) A.go.ld_itemb = 1;
  if (A.go.ld_user == 0) A.go.ld_user = d->d_item;
  if (A.go.ld_usert == 0) A.go.ld_usert = d->Ls;
  if (A.go.ld_item % 4 || A.go.ld_user % 4) return fail(TLSAN_E_UNSUPPORTED, "gradient row strides must be multiples of 4 floats");
  if (hp->reg == 0.0f && g->sparse && hp->norm_mode == TLSAN_NORM_TF18) {
    // pure per-row sums of the used rows (what the sharded step asks for): they ride with the dense
    // finalize as in the lazy train step, written straight to the output rows -- no apply launch
    // (sparse == 2: the four outputs are views of ONE fused row table, see tlsan_grads_out)
    A.presum_rows = g->sparse == 2 ? 2 : 1;
    A.Rc = g->cate_emb;
    category_split(A, d, b);
    if ((rc = run_backward(d, s, p, b, hp, false, out, w, st, L, hs, &A, g->dense))) return rc;
    if (A.csplit > 1) {  // the split workgroups left exact double sums: round them into the output
      const int n = d->cate_count * d->d_cate;
      hipLaunchKernelGGL(k_rc64_to_float, dim3((n + 255) / 256), dim3(256), 0, hs, st.Rc64, g->cate_emb, n);
      CHECK_LAUNCH("k_rc64_to_float");
    }
    prof_mark(5, hs);
    prof_step_done();
    return TLSAN_OK;
  }
  if ((rc = run_backward(d, s, p, b, hp, false, out, w, st, L, hs))) return rc;
  if (hp->norm_mode == TLSAN_NORM_DEDUP && (rc = clip_dedup(A, hp, out, w, st, b, hs))) return rc;
  if ((rc = launch_apply(AP_GRADS, A, true, hs))) return rc;
  prof_mark(5, hs);
  prof_step_done();
  return TLSAN_OK;
}

static int eval_ranks_impl(const tlsan_dims* d, const tlsan_params* p, const float* u_t, const int32_t* labels, int32_t B,
                           int32_t* ranks, void* ws, size_t ws_bytes, void* stream, const float* s_label_in, int id_mul,
                           int id_add, float* s_label_out);

int tlsan_eval_ranks(const tlsan_dims* d, const tlsan_params* p, const float* u_t, const int32_t* labels, int32_t B,
                     int32_t* ranks, void* ws, size_t ws_bytes, void* stream) {
  return eval_ranks_impl(d, p, u_t, labels, B, ranks, ws, ws_bytes, stream, nullptr, 1, 0, nullptr);
}

int tlsan_eval_label_scores(const tlsan_dims* d, const tlsan_params* p, const float* u_t, const int32_t* labels, int32_t B,
                            float* scores, void* ws, size_t ws_bytes, void* stream) {
  if (!scores) return fail(TLSAN_E_BADARG, "tlsan_eval_label_scores: scores is NULL");
  return eval_ranks_impl(d, p, u_t, labels, B, nullptr, ws, ws_bytes, stream, nullptr, 1, 0, scores);
}

int tlsan_eval_counts_shard(const tlsan_dims* d, const tlsan_params* p, const float* u_t, const float* label_scores,
                            const int32_t* labels_global, int32_t B, int32_t id_mul, int32_t id_add, int32_t* counts,
                            void* ws, size_t ws_bytes, void* stream) {
  if (!label_scores || !counts || id_mul < 1 || id_add < 0) return fail(TLSAN_E_BADARG, "tlsan_eval_counts_shard: bad arguments");
  return eval_ranks_impl(d, p, u_t, labels_global, B, counts, ws, ws_bytes, stream, label_scores, id_mul, id_add, nullptr);
}

// s_label_in == NULL: the label's score is computed here (labels index THIS table); s_label_out != NULL: only that.
static int eval_ranks_impl(const tlsan_dims* d, const tlsan_params* p, const float* u_t, const int32_t* labels, int32_t B,
                           int32_t* ranks, void* ws, size_t ws_bytes, void* stream, const float* s_label_in, int id_mul,
                           int id_add, float* s_label_out) {
  Shape s;
  int rc = shape_of(d, &s);
  if (rc) return rc;
  if ((rc = check_params(p))) return rc;
  if (!u_t || !labels || (!ranks && !s_label_out) || B < 1) return fail(TLSAN_E_BADARG, "bad eval arguments");
  if (!ws) return fail(TLSAN_E_WORKSPACE, "ws is NULL");
  Ws w;
  carve(d, s, B, 0, (char*)ws, &w);
  if (w.bytes > ws_bytes) return fail(TLSAN_E_WORKSPACE, "workspace too small: need %zu have %zu", w.bytes, ws_bytes);
  hipStream_t hs = (hipStream_t)stream;
  EvalArgs e;
  memset(&e, 0, sizeof(e));
  e.p = norm_params(p, d); e.u_t = u_t; e.labels = labels; e.B = B; e.I = d->item_count; e.di = d->d_item; e.dc = d->d_cate;
  e.s_label = s_label_out ? s_label_out : (s_label_in ? const_cast<float*>(s_label_in) : w.s_label);
  e.ranks = ranks; e.all_emb = w.all_emb; e.id_mul = id_mul; e.id_add = id_add;
  if (ranks && hipMemsetAsync(ranks, 0, sizeof(int32_t) * (size_t)B, hs) != hipSuccess) return fail(TLSAN_E_LAUNCH, "memset ranks");
  const int ut = (B + 15) / 16;
  const int ntiles = (d->item_count + 15) / 16;
  int chunks = (ntiles + 3) / 4;
  const int want = (2048 + ut - 1) / ut;  // enough workgroups to fill the chip
  if (chunks > want) chunks = want;
  if (chunks < 1) chunks = 1;
  const int nae = (d->item_count * (s.D / 4) + 255) / 256;
  int ngrp = ((d->item_count + 63) / 64 + 3) / 4;  // workgroups (4 wavefronts x 64 items) along the items
  if (ngrp > want) ngrp = want;
#define EVAL_LAUNCH(DD)                                                                                  \
  do {                                                                                                   \
    if (!s_label_in) hipLaunchKernelGGL(k_eval_label<DD>, dim3(ut), dim3(64), 0, hs, e);                 \
    if (!ranks) break;                                                                                   \
    if (e.all_emb) {                                                                                     \
      hipLaunchKernelGGL(k_all_emb<DD>, dim3(nae), dim3(256), 0, hs, e);                                 \
      hipLaunchKernelGGL(k_eval_rank_dense<DD>, dim3(ut, ngrp), dim3(256), 0, hs, e);                    \
    } else {                                                                                             \
      hipLaunchKernelGGL(k_eval_rank<DD>, dim3(ut, chunks), dim3(256), 0, hs, e);                        \
    }                                                                                                    \
  } while (0)
  if (s.D == 64) EVAL_LAUNCH(64);
  else if (s.D == 128) EVAL_LAUNCH(128);
  else EVAL_LAUNCH(256);
#undef EVAL_LAUNCH
  CHECK_LAUNCH("k_eval");
  return TLSAN_OK;
}

struct RowsWs { int32_t *cnt, *off, *cur, *list; double* part; long long* bsum; size_t bytes; int nblk; };
static void carve_rows(int32_t nrows, int32_t n, char* base, RowsWs* w) {
  size_t o = 0;
  auto take = [&](size_t nb) { char* p = base ? base + o : nullptr; o += al(nb); return p; };
  w->nblk = (nrows + AP_ROWS_PB - 1) / AP_ROWS_PB;
  w->cnt = (int32_t*)take(4 * (size_t)nrows);
  w->off = (int32_t*)take(4 * (size_t)nrows);
  w->cur = (int32_t*)take(4 * (size_t)nrows);
  w->list = (int32_t*)take(4 * (size_t)(n > 0 ? n : 1));
  w->part = (double*)take(8 * (size_t)w->nblk);
  w->bsum = (long long*)take(8 * ((size_t)nrows + 4095) / 4096);
  w->bytes = o;
}

size_t tlsan_rows_apply_workspace(int32_t nrows, int32_t n) {
  if (nrows < 1 || n < 0) return 0;
  RowsWs w;
  carve_rows(nrows, n, nullptr, &w);
  return w.bytes;
}

int tlsan_rows_apply(float* W, int32_t ld, int32_t nrows, int32_t width, int32_t reg_cols, const float* grows,
                     int32_t ldg, const int32_t* dest, int32_t n, float gscale, const float* step_dev, float reg,
                     double* sumsq_out, void* ws, size_t ws_bytes, void* stream) {
  if (!W || !step_dev || nrows < 1 || n < 0 || (n > 0 && (!grows || !dest)))
    return fail(TLSAN_E_BADARG, "tlsan_rows_apply: bad pointer / size");
  if (width < 4 || width % 4 || width > 16 * 4 * ROWS_NCH || ld < width || (n > 0 && ldg < width) || reg_cols < 0 || reg_cols > width)
    return fail(TLSAN_E_UNSUPPORTED, "tlsan_rows_apply: width must be a multiple of 4 in 4..%d", 16 * 4 * ROWS_NCH);
  if (ld % 4 || ldg % 4) return fail(TLSAN_E_UNSUPPORTED, "tlsan_rows_apply: row strides must be multiples of 4 floats");
  if (!ws) return fail(TLSAN_E_WORKSPACE, "ws is NULL");
  RowsWs w;
  carve_rows(nrows, n, (char*)ws, &w);
  if (w.bytes > ws_bytes) return fail(TLSAN_E_WORKSPACE, "workspace too small: need %zu have %zu", w.bytes, ws_bytes);
  hipStream_t hs = (hipStream_t)stream;
  if (hipMemsetAsync(w.cnt, 0, 4 * (size_t)nrows, hs) != hipSuccess) return fail(TLSAN_E_LAUNCH, "memset cnt");
  GIdxArgs gi;
  gi.dest = dest; gi.n = n; gi.nrows = nrows; gi.cnt = w.cnt; gi.cur = w.cur; gi.list = w.list;
  if (n > 0) { hipLaunchKernelGGL(k_gidx<false>, dim3((n + 255) / 256), dim3(256), 0, hs, gi); CHECK_LAUNCH("k_gidx<count>"); }
  ScanArgs sa;
  memset(&sa, 0, sizeof(sa));
  sa.cnt[0] = w.cnt; sa.off[0] = w.off; sa.cur[0] = w.cur; sa.n[0] = nrows;
  const int nscan = (nrows + 4095) / 4096;
  sa.blk0[0] = 0; sa.blk0[1] = nscan; sa.blk0[2] = nscan;
  {
    const int rc = launch_scan(sa, nscan, w.bsum, hs);
    if (rc) return rc;
  }
  if (n > 0) { hipLaunchKernelGGL(k_gidx<true>, dim3((n + 255) / 256), dim3(256), 0, hs, gi); CHECK_LAUNCH("k_gidx<fill>"); }
  RowsArgs ra;
  ra.W = W; ra.ld = ld; ra.nrows = nrows; ra.width = width; ra.reg_cols = reg_cols; ra.G = grows; ra.ldg = ldg;
  ra.cnt = w.cnt; ra.off = w.off; ra.list = w.list; ra.gscale = gscale; ra.step_dev = step_dev; ra.reg = reg;
  ra.part_out = w.part;
  hipLaunchKernelGGL(k_rows_apply, dim3(w.nblk), dim3(256), 0, hs, ra);
  CHECK_LAUNCH("k_rows_apply");
  if (sumsq_out) {
    hipLaunchKernelGGL(k_reduce_double, dim3(1), dim3(256), 0, hs, w.part, w.nblk, sumsq_out);
    CHECK_LAUNCH("k_reduce_double");
  }
  return TLSAN_OK;
}

int tlsan_route_plan(const int32_t* keys, int32_t n_keys, int32_t R, int32_t G, const int32_t* cate_by_key,
                     int32_t* flags, int32_t* rank, int32_t* uniq, int32_t* n_uniq, int32_t* sendbuf, int32_t cap,
                     int32_t* cate_c, int32_t cate_pad, int32_t* comp, int32_t* counts_out, void* stream) {
  if (!keys || !cate_by_key || !flags || !rank || !uniq || !n_uniq || !sendbuf || !cate_c || !comp)
    return fail(TLSAN_E_BADARG, "tlsan_route_plan: NULL pointer");
  if (n_keys < 1 || R < 1 || G < 1 || (long long)R * G >= (1LL << 31)) return fail(TLSAN_E_BADARG, "tlsan_route_plan: bad sizes");
  if (cap < 0) return fail(TLSAN_E_BADARG, "tlsan_route_plan: cap < 0");
  const int need = R < n_keys ? R : n_keys;   // rows one owner can be asked for
  hipStream_t hs = (hipStream_t)stream;
  const int nkeys = R * G;
  if (cate_pad < 0) return fail(TLSAN_E_BADARG, "tlsan_route_plan: cate_pad < 0");
  RouteArgs a;
  memset(&a, 0, sizeof(a));
  a.keys = keys; a.n_keys = n_keys; a.R = R; a.G = G; a.prefix = rank; a.uniq = uniq; a.n_uniq = n_uniq;
  a.cate_by_key = cate_by_key; a.flags = flags; a.sendbuf = sendbuf; a.cap = cap;
  a.cate_c = cate_c; a.cate_pad = cate_pad; a.comp = comp; a.counts_out = counts_out;
  a.overflow_need = cap < need ? need : 0;
  int nt = n_keys > G ? n_keys : G;
  if (cate_pad > nt) nt = cate_pad;
  hipLaunchKernelGGL(k_route_mark, dim3((n_keys + 255) / 256), dim3(256), 0, hs, a);
  CHECK_LAUNCH("k_route_mark");
  // chunk sums of the scan: `uniq` receives at most min(n_keys, nkeys) entries, so when the key space
  // is larger than the batch its tail is free during the call (8-byte aligned slice)
  const int nscan = (nkeys + 4095) / 4096;
  const long long first = ((long long)(n_keys < nkeys ? n_keys : nkeys) + 1) / 2 * 2;
  long long* bsum = ((reinterpret_cast<uintptr_t>(uniq) & 7) == 0 && first + 2LL * nscan <= nkeys)
                        ? reinterpret_cast<long long*>(uniq + first) : nullptr;
  int rc = scan_compact_impl(flags, nkeys, rank, uniq, n_uniq, bsum, hs);
  if (rc) return rc;
  hipLaunchKernelGGL(k_route_finish, dim3((nt + 255) / 256), dim3(256), 0, hs, a);
  CHECK_LAUNCH("k_route_finish");
  return TLSAN_OK;
}

int tlsan_shard_gather(const float* shard, int32_t ld, int32_t R, int32_t W, const int32_t* recvbuf, int32_t cap,
                       int32_t G, int32_t n_recv, float* rows_out, int32_t* recv_rows, void* stream) {
  if (!shard || !recvbuf || n_recv < 0 || (n_recv > 0 && (!rows_out || !recv_rows)) || G < 1 || R < 1 || cap < 1)
    return fail(TLSAN_E_BADARG, "tlsan_shard_gather: bad arguments");
  if (W < 4 || W % 4 || ld < W || ld % 4) return fail(TLSAN_E_UNSUPPORTED, "tlsan_shard_gather: W, ld must be multiples of 4");
  if (n_recv == 0) return TLSAN_OK;
  GatherArgs a;
  a.shard = shard; a.ld = ld; a.W = W; a.recvbuf = recvbuf; a.cap = cap; a.G = G; a.n_recv = n_recv; a.R = R;
  a.rows_out = rows_out; a.recv_rows = recv_rows;
  hipLaunchKernelGGL(k_shard_gather, dim3((n_recv + 15) / 16), dim3(256), 0, (hipStream_t)stream, a);
  CHECK_LAUNCH("k_shard_gather");
  return TLSAN_OK;
}

static int shard_opt_ctx(const tlsan_shard_optimizer* o, float lr, OptCtx* oc) {
  memset(oc, 0, sizeof(*oc));
  if (!o || o->kind == TLSAN_OPT_SGD) return TLSAN_OK;
  if (o->scale) return fail(TLSAN_E_UNSUPPORTED, "tlsan_shard_optimizer: lazy L2 (scale) is for SGD only");
  if (o->kind != TLSAN_OPT_ADAM && o->kind != TLSAN_OPT_RMSPROP && o->kind != TLSAN_OPT_ADADELTA)
    return fail(TLSAN_E_BADARG, "tlsan_shard_optimizer: kind %d", o->kind);
  if (!o->shard_s1 || !o->shard_s2 || !o->cate_s1 || !o->cate_s2 || !o->dense_s1 || !o->dense_s2)
    return fail(TLSAN_E_BADARG, "tlsan_shard_optimizer: NULL accumulator");
  if (o->kind == TLSAN_OPT_ADAM && o->step < 1) return fail(TLSAN_E_BADARG, "tlsan_shard_optimizer: Adam's step counts from 1");
  oc->opt = o->kind; oc->lr = lr; oc->b1 = o->beta1; oc->b2 = o->beta2; oc->eps = o->epsilon;
  if (o->kind == TLSAN_OPT_ADAM)
    oc->alpha = (float)((double)lr * sqrt(1.0 - pow((double)o->beta2, o->step)) / (1.0 - pow((double)o->beta1, o->step)));
  return TLSAN_OK;
}

int tlsan_shard_summary(const float* flat, int32_t n_dense, int32_t n_cate, int32_t G, float lr, float reg, float clip,
                        const double* S_cate, float* dense, float* dense_KT, const tlsan_dims* d,
                        float* step_dev, float* loss_out, float* gnorm_out, void* stream) {
  return tlsan_shard_summary_opt(flat, n_dense, n_cate, G, lr, reg, clip, S_cate, dense, dense_KT, d, step_dev, loss_out,
                                 gnorm_out, nullptr, stream);
}

int tlsan_shard_summary_opt(const float* flat, int32_t n_dense, int32_t n_cate, int32_t G, float lr, float reg, float clip,
                            const double* S_cate, float* dense, float* dense_KT, const tlsan_dims* d,
                            float* step_dev, float* loss_out, float* gnorm_out, const tlsan_shard_optimizer* opt,
                            void* stream) {
  if (!flat || !S_cate || !dense || !dense_KT || !d || !step_dev || !loss_out || !gnorm_out || G < 1)
    return fail(TLSAN_E_BADARG, "tlsan_shard_summary: bad arguments");
  tlsan_dense_layout L;
  int rc = tlsan_dense_layout_of(d, &L);
  if (rc) return rc;
  if (n_dense != L.n_dense) return fail(TLSAN_E_BADARG, "tlsan_shard_summary: n_dense does not match dims");
  SummaryArgs a;
  a.flat = flat; a.n_dense = n_dense; a.n_cate = n_cate; a.G = G; a.lr = lr; a.reg = reg; a.clip = clip;
  a.S_cate = S_cate; a.dense = dense; a.dense_KT = dense_KT; a.D = d->d; a.K_off = L.K; a.k0_off = L.k0;
  a.step_dev = step_dev; a.loss_out = loss_out; a.gnorm_out = gnorm_out;
  if ((rc = shard_opt_ctx(opt, lr, &a.oc))) return rc;
  a.dense_s1 = opt ? opt->dense_s1 : nullptr; a.dense_s2 = opt ? opt->dense_s2 : nullptr;
  a.P_dev = opt ? opt->scale : nullptr;
  hipLaunchKernelGGL(k_shard_summary, dim3((n_dense + 1023) / 1024), dim3(1024), 0, (hipStream_t)stream, a);
  CHECK_LAUNCH("k_shard_summary");
  return TLSAN_OK;
}

size_t tlsan_shard_apply_workspace(int32_t R, int32_t C) {
  if (R < 1 || C < 1) return 0;
  return al(8 * (size_t)((R + AP_ROWS_PB - 1) / AP_ROWS_PB + (C + AP_ROWS_PB - 1) / AP_ROWS_PB));
}

int tlsan_shard_apply(float* shard, int32_t ld, int32_t cI, int32_t R, int32_t W, int32_t reg_item, int32_t reg_user,
                      const float* vals, int32_t ldv, const int32_t* rows, int32_t n_recv, const int32_t* src_off,
                      int32_t G, int32_t* slots, float gscale, const float* step_dev, float reg,
                      float* cate_emb, int32_t C, int32_t dc, const float* g_cate,
                      double* sumsq_out, float* sumsq_f32, void* ws, size_t ws_bytes, void* stream) {
  return tlsan_shard_apply_opt(shard, ld, cI, R, W, reg_item, reg_user, vals, ldv, rows, n_recv, src_off, G, slots, gscale,
                               step_dev, reg, cate_emb, C, dc, g_cate, sumsq_out, sumsq_f32, nullptr, 0.0f, ws, ws_bytes, stream);
}

int tlsan_shard_apply_opt(float* shard, int32_t ld, int32_t cI, int32_t R, int32_t W, int32_t reg_item, int32_t reg_user,
                          const float* vals, int32_t ldv, const int32_t* rows, int32_t n_recv, const int32_t* src_off,
                          int32_t G, int32_t* slots, float gscale, const float* step_dev, float reg,
                          float* cate_emb, int32_t C, int32_t dc, const float* g_cate,
                          double* sumsq_out, float* sumsq_f32, const tlsan_shard_optimizer* opt, float lr,
                          void* ws, size_t ws_bytes, void* stream) {
  if (!shard || !slots || !step_dev || !cate_emb || !g_cate || !sumsq_out || !src_off || n_recv < 0 ||
      (n_recv > 0 && (!vals || !rows)))
    return fail(TLSAN_E_BADARG, "tlsan_shard_apply: bad pointer / size");
  if (G < 1 || G > SHARD_GMAX) return fail(TLSAN_E_UNSUPPORTED, "tlsan_shard_apply: 1..%d ranks", SHARD_GMAX);
  if (W < 4 || W % 4 || W > 16 * 4 * SHARD_NCH || dc % 4 || dc > 16 * 4 * SHARD_NCH || ld < W || ld % 4 ||
      (n_recv > 0 && (ldv < W || ldv % 4)) || cI < 0 || cI > R || reg_item > W || reg_user > W)
    return fail(TLSAN_E_UNSUPPORTED, "tlsan_shard_apply: widths must be multiples of 4 up to %d", 16 * 4 * SHARD_NCH);
  if (!ws || ws_bytes < tlsan_shard_apply_workspace(R, C)) return fail(TLSAN_E_WORKSPACE, "tlsan_shard_apply: workspace too small");
  ShardApplyArgs a;
  memset(&a, 0, sizeof(a));
  a.shard = shard; a.ld = ld; a.cI = cI; a.R = R; a.W = W; a.reg_item = reg_item; a.reg_user = reg_user;
  a.vals = vals ? vals : shard; a.ldv = vals ? ldv : ld; a.rows = rows; a.n_recv = n_recv; a.G = G;
  for (int s = 0; s <= G; ++s) a.src_off[s] = src_off[s];
  if (a.src_off[0] != 0 || a.src_off[G] != n_recv) return fail(TLSAN_E_BADARG, "tlsan_shard_apply: src_off must run from 0 to n_recv");
  a.slots = slots; a.gscale = gscale; a.step_dev = step_dev; a.reg = reg;
  a.cate_emb = cate_emb; a.C = C; a.dc = dc; a.g_cate = g_cate;
  {
    const int rc_ = shard_opt_ctx(opt, lr, &a.oc);
    if (rc_) return rc_;
  }
  if (a.oc.opt != TLSAN_OPT_SGD) {
    a.shard_s1 = opt->shard_s1; a.shard_s2 = opt->shard_s2; a.cate_s1 = opt->cate_s1; a.cate_s2 = opt->cate_s2;
    a.bias_col = reg_item;   // fused item rows: [item_emb (reg_item columns) | item_b | pad]
  }
  a.part_out = (double*)ws;
  a.nb_rows = (R + AP_ROWS_PB - 1) / AP_ROWS_PB;
  a.nb_cate = (C + AP_ROWS_PB - 1) / AP_ROWS_PB;
  hipStream_t hs = (hipStream_t)stream;
  if (n_recv > 0) {
    hipLaunchKernelGGL(k_slot_mark, dim3((n_recv + 255) / 256), dim3(256), 0, hs, a);
    CHECK_LAUNCH("k_slot_mark");
  }
  hipLaunchKernelGGL(k_shard_apply, dim3(a.nb_rows + a.nb_cate), dim3(256), 0, hs, a);
  CHECK_LAUNCH("k_shard_apply");
  hipLaunchKernelGGL(k_reduce_double2, dim3(2), dim3(256), 0, hs, a.part_out, a.nb_rows, a.nb_cate, sumsq_out, sumsq_f32);
  CHECK_LAUNCH("k_reduce_double2");
  return TLSAN_OK;
}

static int scan_compact_impl(const int32_t* cnt, int32_t n, int32_t* prefix, int32_t* uniq, int32_t* n_uniq, long long* bsum,
                             hipStream_t hs) {
  if (!cnt || !prefix || n < 1) return fail(TLSAN_E_BADARG, "tlsan_scan_compact: bad arguments");
  ScanArgs sa;
  memset(&sa, 0, sizeof(sa));
  sa.cnt[0] = cnt; sa.off[0] = prefix; sa.cur[0] = nullptr; sa.n[0] = n;
  sa.uniq[0] = uniq; sa.n_uniq[0] = n_uniq;
  const int nscan = (n + 4095) / 4096;
  sa.blk0[0] = 0; sa.blk0[1] = nscan; sa.blk0[2] = nscan;
  return launch_scan(sa, nscan, bsum, hs);
}

// (no scratch: one launch whose prefix re-read grows with the square of n / 4096 -- meant for tables up
//  to a few hundred thousand entries; tlsan_route_plan scans its key space with chunk sums)
size_t tlsan_shard_apply_lazy_workspace(int32_t n_recv, int32_t C) {
  if (n_recv < 0 || C < 1) return 0;
  return al(8 * (size_t)((n_recv + AP_ROWS_PB - 1) / AP_ROWS_PB + (C + AP_ROWS_PB - 1) / AP_ROWS_PB + 1));
}

int tlsan_shard_apply_lazy(float* shard, int32_t ld, int32_t cI, int32_t R, int32_t W, int32_t reg_item, int32_t reg_user,
                           const float* vals, int32_t ldv, const int32_t* rows, int32_t n_recv, const int32_t* src_off,
                           int32_t G, uint64_t* slots64, uint32_t stamp, float gscale, const float* step_dev,
                           float* cate_emb, int32_t C, int32_t dc, const float* g_cate,
                           double* sumsq_out, float* sumsq_f32, float* scale, void* ws, size_t ws_bytes, void* stream) {
  if (!shard || !slots64 || !step_dev || !cate_emb || !g_cate || !sumsq_out || !src_off || !scale || n_recv < 0 ||
      (n_recv > 0 && (!vals || !rows)))
    return fail(TLSAN_E_BADARG, "tlsan_shard_apply_lazy: bad pointer / size");
  if (G < 1 || G > SHARD_GMAX) return fail(TLSAN_E_UNSUPPORTED, "tlsan_shard_apply_lazy: 1..%d ranks", SHARD_GMAX);
  if (W < 4 || W % 4 || dc % 4 || ld < W || ld % 4 || (n_recv > 0 && (ldv < W || ldv % 4)) || cI < 0 || cI > R ||
      reg_item > W || reg_user > W || stamp == 0)
    return fail(TLSAN_E_UNSUPPORTED, "tlsan_shard_apply_lazy: widths must be multiples of 4, stamp != 0");
  if (!ws || ws_bytes < tlsan_shard_apply_lazy_workspace(n_recv, C)) return fail(TLSAN_E_WORKSPACE, "tlsan_shard_apply_lazy: workspace too small");
  ShardLazyArgs a;
  memset(&a, 0, sizeof(a));
  a.shard = shard; a.ld = ld; a.cI = cI; a.R = R; a.W = W; a.reg_item = reg_item; a.reg_user = reg_user;
  a.vals = vals ? vals : shard; a.ldv = vals ? ldv : ld; a.rows = rows; a.n_recv = n_recv; a.G = G;
  for (int s = 0; s <= G; ++s) a.src_off[s] = src_off[s];
  if (a.src_off[0] != 0 || a.src_off[G] != n_recv) return fail(TLSAN_E_BADARG, "tlsan_shard_apply_lazy: src_off must run from 0 to n_recv");
  a.slots64 = (unsigned long long*)slots64; a.stamp = stamp; a.gscale = gscale; a.step_dev = step_dev;
  a.cate_emb = cate_emb; a.C = C; a.dc = dc; a.g_cate = g_cate; a.P_dev = scale;
  a.part_out = (double*)ws;
  a.nb_rows = (n_recv + AP_ROWS_PB - 1) / AP_ROWS_PB;
  if (a.nb_rows < 1) a.nb_rows = 1;   // (workgroup 0 commits the scale)
  a.nb_cate = (C + AP_ROWS_PB - 1) / AP_ROWS_PB;
  hipStream_t hs = (hipStream_t)stream;
  if (n_recv > 0) {
    hipLaunchKernelGGL(k_slot_mark64, dim3((n_recv + 255) / 256), dim3(256), 0, hs, a);
    CHECK_LAUNCH("k_slot_mark64");
  }
  hipLaunchKernelGGL(k_shard_apply_lazy, dim3(a.nb_rows + a.nb_cate), dim3(256), 0, hs, a);
  CHECK_LAUNCH("k_shard_apply_lazy");
  hipLaunchKernelGGL(k_reduce_lazy2, dim3(2), dim3(256), 0, hs, a.part_out, a.nb_rows, a.nb_cate, sumsq_out, sumsq_f32, (uint32_t*)nullptr);
  CHECK_LAUNCH("k_reduce_lazy2");
  return TLSAN_OK;
}

// ---- static-shape forms of the three calls above (include/tlsan.h): fixed `cap` row slots per (source, owner) pair
static int route_plan_static_impl(const int32_t* keys, int32_t n_keys, int32_t R, int32_t G, const int32_t* cate_by_key,
                                  int32_t* flags, int32_t* rank, int32_t* uniq, int32_t* n_uniq, int32_t* sendbuf, int32_t cap,
                                  int32_t* cate_c, int32_t* comp, int32_t* counts_out, int32_t* status, int32_t* status_host, void* stream);
int tlsan_route_plan_static(const int32_t* keys, int32_t n_keys, int32_t R, int32_t G, const int32_t* cate_by_key,
                            int32_t* flags, int32_t* rank, int32_t* uniq, int32_t* n_uniq, int32_t* sendbuf, int32_t cap,
                            int32_t* cate_c, int32_t* comp, int32_t* counts_out, int32_t* status, void* stream) {
  return route_plan_static_impl(keys, n_keys, R, G, cate_by_key, flags, rank, uniq, n_uniq, sendbuf, cap, cate_c, comp, counts_out, status, nullptr, stream);
}
static int route_plan_static_impl(const int32_t* keys, int32_t n_keys, int32_t R, int32_t G, const int32_t* cate_by_key,
                                  int32_t* flags, int32_t* rank, int32_t* uniq, int32_t* n_uniq, int32_t* sendbuf, int32_t cap,
                                  int32_t* cate_c, int32_t* comp, int32_t* counts_out, int32_t* status, int32_t* status_host, void* stream) {
  if (!keys || !cate_by_key || !flags || !rank || !uniq || !n_uniq || !sendbuf || !cate_c || !comp || !status)
    return fail(TLSAN_E_BADARG, "tlsan_route_plan_static: NULL pointer");
  if (n_keys < 1 || R < 1 || G < 1 || (long long)R * G >= (1LL << 31)) return fail(TLSAN_E_BADARG, "tlsan_route_plan_static: bad sizes");
  if (cap < 1 || (long long)cap * G >= (1LL << 31)) return fail(TLSAN_E_BADARG, "tlsan_route_plan_static: bad cap");
  hipStream_t hs = (hipStream_t)stream;
  const int nkeys = R * G;
  RouteArgs a;
  memset(&a, 0, sizeof(a));
  a.keys = keys; a.n_keys = n_keys; a.R = R; a.G = G; a.prefix = rank; a.uniq = uniq; a.n_uniq = n_uniq;
  a.cate_by_key = cate_by_key; a.flags = flags; a.sendbuf = sendbuf; a.cap = cap;
  a.cate_c = cate_c; a.cate_pad = G * cap; a.comp = comp; a.counts_out = counts_out;
  hipLaunchKernelGGL(k_route_mark, dim3((n_keys + 255) / 256), dim3(256), 0, hs, a);
  CHECK_LAUNCH("k_route_mark");
  const int nscan = (nkeys + 4095) / 4096;
  const long long first = ((long long)(n_keys < nkeys ? n_keys : nkeys) + 1) / 2 * 2;
  long long* bsum = ((reinterpret_cast<uintptr_t>(uniq) & 7) == 0 && first + 2LL * nscan <= nkeys)
                        ? reinterpret_cast<long long*>(uniq + first) : nullptr;
  int rc = scan_compact_impl(flags, nkeys, rank, uniq, n_uniq, bsum, hs);
  if (rc) return rc;
  int nt = n_keys > G * cap ? n_keys : G * cap;
  hipLaunchKernelGGL(k_route_finish_static, dim3((nt + 255) / 256), dim3(256), 0, hs, a, status, status_host);
  CHECK_LAUNCH("k_route_finish_static");
  return TLSAN_OK;
}

int tlsan_shard_gather_static(const float* shard, int32_t ld, int32_t R, int32_t W, const int32_t* recvbuf, int32_t cap,
                              int32_t G, float* rows_out, int32_t* recv_rows, uint64_t* slots64, const uint32_t* stamp,
                              void* stream) {
  if (!shard || !recvbuf || !rows_out || !recv_rows || G < 1 || R < 1 || cap < 1 || (slots64 && !stamp))
    return fail(TLSAN_E_BADARG, "tlsan_shard_gather_static: bad arguments");
  if (W < 4 || W % 4 || ld < W || ld % 4) return fail(TLSAN_E_UNSUPPORTED, "tlsan_shard_gather_static: W, ld must be multiples of 4");
  GatherStaticArgs a;
  a.shard = shard; a.ld = ld; a.W = W; a.recvbuf = recvbuf; a.cap = cap; a.G = G; a.R = R;
  a.rows_out = rows_out; a.recv_rows = recv_rows; a.slots64 = (unsigned long long*)slots64; a.stamp_dev = stamp;
  hipLaunchKernelGGL(k_shard_gather_static, dim3((G * cap + 15) / 16), dim3(256), 0, (hipStream_t)stream, a);
  CHECK_LAUNCH("k_shard_gather_static");
  return TLSAN_OK;
}

int tlsan_shard_gather_wire_bf16(const float* shard, int32_t ld, int32_t R, int32_t d_emb, int32_t tail,
                                 const int32_t* recvbuf, int32_t cap, int32_t G, void* rows_out, int32_t pitch,
                                 int32_t* recv_rows, uint64_t* slots64, const uint32_t* stamp, void* stream) {
  if (!shard || !recvbuf || !rows_out || !recv_rows || G < 1 || R < 1 || cap < 1 || (slots64 && !stamp))
    return fail(TLSAN_E_BADARG, "tlsan_shard_gather_wire_bf16: bad arguments");
  if (d_emb < 4 || d_emb % 4 || tail < 0 || ld < d_emb + tail || ld % 4 || pitch % 16 || pitch < 2 * d_emb + 4 * tail)
    return fail(TLSAN_E_UNSUPPORTED, "tlsan_shard_gather_wire_bf16: d_emb %% 4 == 0, pitch %% 16 == 0, pitch >= 2 d_emb + 4 tail");
  GatherWireArgs w;
  w.g.shard = shard; w.g.ld = ld; w.g.W = 0; w.g.recvbuf = recvbuf; w.g.cap = cap; w.g.G = G; w.g.R = R;
  w.g.rows_out = (float*)rows_out; w.g.recv_rows = recv_rows; w.g.slots64 = (unsigned long long*)slots64; w.g.stamp_dev = stamp;
  w.d_emb = d_emb; w.tail = tail; w.pitch = pitch;
  hipLaunchKernelGGL(k_shard_gather_wire_bf16, dim3((G * cap + 15) / 16), dim3(256), 0, (hipStream_t)stream, w);
  CHECK_LAUNCH("k_shard_gather_wire_bf16");
  return TLSAN_OK;
}

int tlsan_shard_apply_lazy_static(float* shard, int32_t ld, int32_t cI, int32_t R, int32_t W, int32_t reg_item, int32_t reg_user,
                                  const float* vals, int32_t ldv, const int32_t* rows, int32_t cap, int32_t G,
                                  uint64_t* slots64, uint32_t* stamp, int32_t marked, float gscale, const float* step_dev,
                                  float* cate_emb, int32_t C, int32_t dc, const float* g_cate,
                                  double* sumsq_out, float* sumsq_f32, float* scale,
                                  void* ws, size_t ws_bytes, void* stream) {
  if (!shard || !slots64 || !stamp || !step_dev || !cate_emb || !g_cate || !sumsq_out || !scale || !vals || !rows)
    return fail(TLSAN_E_BADARG, "tlsan_shard_apply_lazy_static: NULL pointer");
  if (G < 1 || G > SHARD_GMAX) return fail(TLSAN_E_UNSUPPORTED, "tlsan_shard_apply_lazy_static: 1..%d ranks", SHARD_GMAX);
  if (cap < 1 || (long long)cap * G >= (1LL << 31)) return fail(TLSAN_E_BADARG, "tlsan_shard_apply_lazy_static: bad cap");
  if (W < 4 || W % 4 || dc % 4 || ld < W || ld % 4 || ldv < W || ldv % 4 || cI < 0 || cI > R || reg_item > W || reg_user > W)
    return fail(TLSAN_E_UNSUPPORTED, "tlsan_shard_apply_lazy_static: widths must be multiples of 4");
  const int n_recv = G * cap;
  if (!ws || ws_bytes < tlsan_shard_apply_lazy_workspace(n_recv, C)) return fail(TLSAN_E_WORKSPACE, "tlsan_shard_apply_lazy_static: workspace too small");
  ShardLazyArgs a;
  memset(&a, 0, sizeof(a));
  a.shard = shard; a.ld = ld; a.cI = cI; a.R = R; a.W = W; a.reg_item = reg_item; a.reg_user = reg_user;
  a.vals = vals; a.ldv = ldv; a.rows = rows; a.n_recv = n_recv; a.G = G;
  for (int s = 0; s <= G; ++s) a.src_off[s] = s * cap;
  a.slots64 = (unsigned long long*)slots64; a.stamp_dev = stamp; a.gscale = gscale; a.step_dev = step_dev;
  a.cate_emb = cate_emb; a.C = C; a.dc = dc; a.g_cate = g_cate; a.P_dev = scale;
  a.part_out = (double*)ws;
  a.nb_rows = (n_recv + AP_ROWS_PB - 1) / AP_ROWS_PB;
  a.nb_cate = (C + AP_ROWS_PB - 1) / AP_ROWS_PB;
  hipStream_t hs = (hipStream_t)stream;
  if (!marked) {
    hipLaunchKernelGGL(k_slot_mark64, dim3((n_recv + 255) / 256), dim3(256), 0, hs, a);
    CHECK_LAUNCH("k_slot_mark64");
  }
  hipLaunchKernelGGL(k_shard_apply_lazy, dim3(a.nb_rows + a.nb_cate), dim3(256), 0, hs, a);
  CHECK_LAUNCH("k_shard_apply_lazy");
  // (the closing sums stay a launch of their own: taken by the last workgroup to finish they cost ~2000 same-address
  //  ticket atomics, 29 us against 7 + 4)
  hipLaunchKernelGGL(k_reduce_lazy2, dim3(2), dim3(256), 0, hs, a.part_out, a.nb_rows, a.nb_cate, sumsq_out, sumsq_f32, stamp);
  CHECK_LAUNCH("k_reduce_lazy2");
  return TLSAN_OK;
}

int tlsan_scan_compact(const int32_t* cnt, int32_t n, int32_t* prefix, int32_t* uniq, int32_t* n_uniq, void* stream) {
  return scan_compact_impl(cnt, n, prefix, uniq, n_uniq, nullptr, (hipStream_t)stream);
}

int tlsan_debug_stamps(void* device_buf) {
  g_stamps = (unsigned long long*)device_buf;
  return TLSAN_OK;
}

int tlsan_profile_stride(int every) {
  if (every < 1) return fail(TLSAN_E_BADARG, "profile stride must be >= 1");
  g_prof_stride = every;
  return TLSAN_OK;
}

int tlsan_profile_enable(int level) {
  if (level < 0 || level > 2) return fail(TLSAN_E_BADARG, "profile level must be 0..2");
  if (level > 0 && !g_prof_ev) {
    g_prof_ev = (hipEvent_t*)malloc(sizeof(hipEvent_t) * PROF_MAX_STEPS * PROF_MARKS);
    if (!g_prof_ev) return fail(TLSAN_E_WORKSPACE, "out of host memory");
    for (int k = 0; k < PROF_MAX_STEPS * PROF_MARKS; ++k)
      if (hipEventCreate(&g_prof_ev[k]) != hipSuccess) return fail(TLSAN_E_LAUNCH, "hipEventCreate");
  }
  g_prof_level = level;
  g_prof_n = 0;
  g_prof_tick = 0;
  return TLSAN_OK;
}

int tlsan_shard_plan_static(const tlsan_static_plan* p) {
  if (!p || !p->dims || !p->cp || !p->cb || !p->state) return fail(TLSAN_E_BADARG, "tlsan_shard_plan_static: NULL argument");
  hipStream_t s1 = (hipStream_t)p->stream, s2 = (hipStream_t)p->stream2;
  if (p->ev_fork && hipStreamWaitEvent(s1, (hipEvent_t)p->ev_fork, 0) != hipSuccess) return fail(TLSAN_E_LAUNCH, "wait(fork)");
  // (the overflow word reaches the pinned host copy by a store of the kernel that raises it: no copy behind the plan)
  int rc = route_plan_static_impl(p->keys, p->n_keys, p->R, p->G, p->cate_by_key, p->flags, p->rank, p->uniq, p->n_uniq,
                                  p->sendbuf, p->cap, p->cate_c, p->comp, nullptr, p->status, (int32_t*)p->status_host, p->stream);
  if (rc) return rc;
  if (p->ev_planned && hipEventRecord((hipEvent_t)p->ev_planned, s1) != hipSuccess) return fail(TLSAN_E_LAUNCH, "record(planned)");
  if ((rc = tlsan_state_recategorize(p->dims, p->cp, p->state, p->stream))) return rc;
  if (p->stream2 != nullptr) {
    if (p->ev_planned && hipStreamWaitEvent(s2, (hipEvent_t)p->ev_planned, 0) != hipSuccess) return fail(TLSAN_E_LAUNCH, "wait(planned)");
    if ((rc = tlsan_batch_index(p->dims, p->cb, p->cp->item_cate, p->state, 0, p->stream2))) return rc;
    if (p->ev_done1 && hipEventRecord((hipEvent_t)p->ev_done1, s2) != hipSuccess) return fail(TLSAN_E_LAUNCH, "record(done1)");
  } else {
    if ((rc = tlsan_batch_index(p->dims, p->cb, p->cp->item_cate, p->state, 0, p->stream))) return rc;
  }
  if (p->record_done0 && p->ev_done0 && hipEventRecord((hipEvent_t)p->ev_done0, s1) != hipSuccess) return fail(TLSAN_E_LAUNCH, "record(done0)");
  return TLSAN_OK;
}

// ---- the announced batches' plans on a launch thread of the library's own ------------------------------------------
// A plan is seven launches, a copy and four event operations on streams of its own; the step beside it is six launches
// on the main stream.  Issued by one host thread they cost it ~85 us per step for 77 us of kernels (the HIP runtime, not
// Python: scripts/shard_cprof.py), so the step was bound by its host.  With TLSAN_PLAN_ASYNC (phases bit) the plans are
// handed, by value, to one worker thread per process, which waits for the pinned word and issues them while the calling
// thread goes on with the main stream.  tlsan_shard_plans_flush() returns once the worker has issued everything handed
// to it (and reports its first error): call it before waiting on a plan's events, before re-using what a plan writes
// from the calling thread, and before a stream capture.
struct PlanJob {
  tlsan_static_plan p;
  tlsan_dims dims; tlsan_params cp; tlsan_batch cb;
  volatile uint32_t* word; uint32_t after;
  int device;
};
// (never destroyed: the worker sleeps on g_pcv when the process exits, and destroying a condition variable that has a
//  waiter blocks in glibc -- every process that had used the thread would hang at exit)
static std::mutex& g_pm = *new std::mutex;
static std::condition_variable& g_pcv = *new std::condition_variable;
static std::condition_variable& g_pidle = *new std::condition_variable;
static std::deque<PlanJob>& g_pq = *new std::deque<PlanJob>;
static bool g_pbusy = false, g_pstarted = false;
static int g_prc = 0;
static char g_pmsg[512] = "";

// a polite spin on a word the GPU writes: a pause instruction per poll, the time slice handed back every 64 polls -- the
// host normally runs ahead of the GPU, and a thread spinning flat out takes a core from the rank's own launch thread
static inline void spin_pause(unsigned long polls) {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
  if ((polls & 63) == 0) sched_yield();
}

static void plan_worker() {
  for (;;) {
    PlanJob j;
    bool skip = false;
    {
      std::unique_lock<std::mutex> lk(g_pm);
      g_pcv.wait(lk, [] { return !g_pq.empty(); });
      j = g_pq.front();
      g_pq.pop_front();
      g_pbusy = true;
      skip = g_prc != 0;     // (an earlier job failed: the error is latched until the caller flushes; later jobs are dropped, not issued)
    }
    int rc = TLSAN_OK;
    if (skip) {
      std::lock_guard<std::mutex> lk(g_pm);
      g_pbusy = false;
      if (g_pq.empty()) g_pidle.notify_all();
      continue;
    }
    if (hipSetDevice(j.device) != hipSuccess) rc = fail(TLSAN_E_LAUNCH, "plan worker: hipSetDevice(%d)", j.device);
    if (!rc && j.word != nullptr) {
      const auto t0 = std::chrono::steady_clock::now();
      unsigned long polls = 0;
      // (sequence numbers run over the full 32 bits on both sides of the ABI: "reached" is (int32)(word - after) >= 0)
      while ((int32_t)(*j.word - j.after) < 0) {
        spin_pause(++polls);
        if ((polls & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) {
          rc = fail(TLSAN_E_LAUNCH, "plan worker: step %u did not start within 30 s", j.after);
          break;
        }
      }
    }
    if (!rc) {
      j.p.dims = &j.dims; j.p.cp = &j.cp; j.p.cb = &j.cb;
      rc = tlsan_shard_plan_static(&j.p);
    }
    {
      std::lock_guard<std::mutex> lk(g_pm);
      if (rc && !g_prc) { g_prc = rc; snprintf(g_pmsg, sizeof(g_pmsg), "%s", g_err); }
      g_pbusy = false;
      if (g_pq.empty()) g_pidle.notify_all();
    }
  }
}

int tlsan_shard_plans_flush(void) {
  std::unique_lock<std::mutex> lk(g_pm);
  g_pidle.wait(lk, [] { return g_pq.empty() && !g_pbusy; });
  if (g_prc) {
    const int rc = g_prc;
    g_prc = 0;
    return fail(rc, "%s", g_pmsg);
  }
  return TLSAN_OK;
}

int tlsan_shard_step_static(const tlsan_static_step* s, int32_t phases, const tlsan_static_plan* const* plans, int32_t n_plans,
                            void* stream) {
  if (!s) return fail(TLSAN_E_BADARG, "tlsan_shard_step_static: NULL argument");
  int rc;
  if (phases & TLSAN_PHASE_GATHER) {
    if (s->wire) rc = tlsan_shard_gather_wire_bf16(s->shard, s->ld, s->R, s->d_emb, s->tail, s->recvbuf, s->cap, s->G, s->rows_out,
                                                   s->pitch, s->recv_rows, s->slots64, s->stamp, stream);
    else rc = tlsan_shard_gather_static(s->shard, s->ld, s->R, s->W, s->recvbuf, s->cap, s->G, (float*)s->rows_out, s->recv_rows,
                                        s->slots64, s->stamp, stream);
    if (rc) return rc;
  }
  if (phases & TLSAN_PHASE_GRADS) {
    if ((rc = tlsan_grads(s->dims, s->cp, s->cb, &s->hp, &s->go, &s->out, s->state, s->ws, s->ws_bytes, stream))) return rc;
  }
  if (phases & TLSAN_PHASE_SUMMARY) {
    if ((rc = tlsan_shard_summary_opt(s->flat, s->n_dense, s->n_cate, s->G, s->lr, s->reg, s->clip, s->S_cate, s->dense, s->dense_KT,
                                      s->dims_full, s->step_dev, s->loss_out, s->gnorm_out, s->opt, stream)))
      return rc;
  }
  if (phases & TLSAN_PHASE_APPLY) {
    if ((rc = tlsan_shard_apply_lazy_static(const_cast<float*>(s->shard), s->ld, s->cI, s->R, s->W, s->reg_item, s->reg_user, s->vals,
                                            s->ldv, s->recv_rows, s->cap, s->G, s->slots64, s->stamp, s->marked, s->gscale,
                                            s->step_dev, s->cate_emb, s->C, s->dc, s->g_cate, s->sumsq_out, s->sumsq_f32, s->scale,
                                            s->lws, s->lws_bytes, stream)))
      return rc;
  }
  if (plans && n_plans > 0 && (phases & TLSAN_PLAN_ASYNC)) {
    if (s->out.started == nullptr) return fail(TLSAN_E_BADARG, "TLSAN_PLAN_ASYNC needs the started word (out.started)");
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_pm);
    if (!g_pstarted) {
      std::thread(plan_worker).detach();
      g_pstarted = true;
    }
    for (int k = 0; k < n_plans; ++k) {
      if (!plans[k]) continue;
      PlanJob j;
      j.p = *plans[k]; j.dims = *plans[k]->dims; j.cp = *plans[k]->cp; j.cb = *plans[k]->cb;
      j.word = (volatile uint32_t*)s->out.started; j.after = s->plans_after; j.device = dev;
      g_pq.push_back(j);
    }
    g_pcv.notify_one();
  } else if (plans && n_plans > 0) {
    // The plans go to slots that earlier steps were the last to use: wait (on the host) until the pinned word says that
    // step `plans_after` has started -- everything queued before that step is then complete.  No event on the main stream.
    if (s->out.started != nullptr) {
      volatile uint32_t* w = (volatile uint32_t*)s->out.started;
      const auto t0 = std::chrono::steady_clock::now();
      unsigned long polls = 0;
      while ((int32_t)(*w - s->plans_after) < 0) {
        spin_pause(++polls);
        if ((polls & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30))
          return fail(TLSAN_E_LAUNCH, "tlsan_shard_step_static: step %u did not start within 30 s", s->plans_after);
      }
    }
    for (int k = 0; k < n_plans; ++k)
      if (plans[k] && (rc = tlsan_shard_plan_static(plans[k]))) return rc;
  }
  return TLSAN_OK;
}

int tlsan_profile_collect(float* host_ms, int max_steps) {
  if (!host_ms || max_steps < 0) return fail(TLSAN_E_BADARG, "bad profile buffer");
  int n = g_prof_n < max_steps ? g_prof_n : max_steps;
  for (int k = 0; k < n; ++k) {
    hipEvent_t* e = g_prof_ev + (size_t)k * PROF_MARKS;
    for (int sgm = 0; sgm < TLSAN_PROF_SEGMENTS; ++sgm) {
      float ms = 0.0f;
      if (g_prof_level == 2 || sgm == 1) {
        if (hipEventSynchronize(e[sgm + 1]) != hipSuccess || hipEventElapsedTime(&ms, e[sgm], e[sgm + 1]) != hipSuccess)
          return fail(TLSAN_E_LAUNCH, "hipEventElapsedTime");
      }
      host_ms[k * TLSAN_PROF_SEGMENTS + sgm] = ms;
    }
  }
  g_prof_n = 0;
  return n;
}

}  // extern "C"
