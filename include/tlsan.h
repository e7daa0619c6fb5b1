/*
 * tlsan.h -- C ABI of libtlsan_hip.so: the MI355X (gfx950) TLSAN hot path.
 *
 * The reference (TsingZ0/TLSAN) has no FFI: its hot path sits behind the Python class
 * `Model` (TLSAN/model.py:13-313) whose methods call `sess.run` on a TensorFlow-1.8 graph.
 * Each entry point below replaces one such `sess.run` fetch; the Python `tlsan_amd.Model`
 * (same constructor / method surface as the reference's Model) binds them with ctypes.
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer inside the structs is a DEVICE pointer unless marked "host";
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); all work is enqueued
 *     on it, nothing synchronises, nothing allocates (caller supplies `ws` and `state`);
 *   - return value: 0 = ok, negative = TLSAN_E_* (never throws across the ABI);
 *     tlsan_last_error() returns a host string for the calling thread's last failure;
 *   - single host thread per GPU, like the reference (one tf.Session, synchronous).
 */
#ifndef TLSAN_H_
#define TLSAN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TLSAN_ABI_VERSION 14

enum {
  TLSAN_OK = 0,
  TLSAN_E_BADARG = -1,      /* unsupported dims / null pointer / shape mismatch            */
  TLSAN_E_WORKSPACE = -2,   /* ws or state buffer too small                              */
  TLSAN_E_LAUNCH = -3,      /* HIP launch error (see tlsan_last_error)                   */
  TLSAN_E_UNSUPPORTED = -4  /* valid request outside what this build implements          */
};

/* Shapes: config keys read by model.py:64,72,76,81,102,116-117. */
typedef struct {
  int32_t user_count, item_count, cate_count;
  int32_t d;          /* hidden_units = d_item + d_cate = d_user + d_cate (model.py:100-109,135) */
  int32_t d_item;     /* itemid_embedding_size (== userid_embedding_size)                        */
  int32_t d_cate;     /* cateid_embedding_size                                                  */
  int32_t num_heads;  /* 8 in every config; d/num_heads in {8,16,32}                             */
  int32_t Ls;         /* long-term window (columns of hist_i / usert_emb), <= 96 (reference max_length = 90) */
} tlsan_dims;

/* Trainable variables of model.py:58-81 + attention/dense weights, fp32, row-major.
 * `dense` packs the small weights (layout: tlsan_dense_layout):
 *   fwa1{W1[dh,dh],b1[dh],W2[dh,dh],b2[dh]} | K[d,d] (tf.layers.dense kernel, model.py:347) |
 *   k0[d] | fwa2{W1,b1,W2,b2} | gamma[1]
 * `dense_KT` is K transposed, maintained by the library after every update
 * (tlsan_sync_derived must be called after the caller writes `dense` itself). */
typedef struct {
  float* item_emb;            /* [I, d_item] */
  float* item_b;              /* [I]         */
  float* user_emb;            /* [U, d_item] */
  float* usert_emb;           /* [U, Ls]     */
  float* cate_emb;            /* [C, d_cate] */
  float* dense;               /* [n_dense]   */
  float* dense_KT;            /* [d, d]      */
  const int32_t* item_cate;   /* [I] item -> category (model.py:85 graph constant) */
  /* Row strides in floats; 0 = densely packed (d_item, 1, d_item, Ls).  Non-zero strides let
   * the tables alias fused rows, e.g. [item_emb | item_b | pad] as exchanged between GPUs. */
  int32_t ld_item, ld_itemb, ld_user, ld_usert;
  /* Optional DEVICE scalar P (NULL = 1): the four L2-regularised tables hold W / P, i.e. the
   * true parameters are P * stored (item_b and `dense` are never scaled).  With
   * TLSAN_L2_LAZY the library keeps P = prod_t (1 - lr_t c_t reg) here instead of decaying
   * every row every step; point it at tlsan_state_scale(state). */
  const float* scale;
  /* Storage type of item_emb, user_emb and cate_emb: TLSAN_TABLE_F32 (0) or TLSAN_TABLE_BF16 (1).
   * With bf16 the three pointers address uint16 elements (row strides still in ELEMENTS); all
   * arithmetic stays fp32, updates are written back with stochastic rounding (deterministic: a
   * hash of step, table and element).  usert_emb, item_b and dense are always fp32.  A build
   * extension: the reference is fp32 throughout. */
  int32_t table_dtype;
  /* Arithmetic of the matrix products of the fused kernel (the two maps of both attention blocks, forward and
   * backward, their weight gradients, the bridge GEMM and its transpose): TLSAN_MATRIX_F32 (0) --
   * v_mfma_f32_16x16x4_f32, exact fp32, the reference's precision; TLSAN_MATRIX_BF16 (1) -- both operands of
   * every product rounded to bfloat16 (nearest even), products and sums in fp32
   * (v_mfma_f32_16x16x16_bf16).  Everything else (softmax, loss, gradients' sums, updates) is fp32 either
   * way.  A build extension for BASELINE.json configs[2] ("bf16"); logits then agree with the fp32 oracle
   * to about 1e-2, not 1e-4. */
  int32_t matrix_dtype;
} tlsan_params;
#define TLSAN_TABLE_F32 0
#define TLSAN_TABLE_BF16 1
#define TLSAN_MATRIX_F32 0
#define TLSAN_MATRIX_BF16 1

typedef struct {
  int32_t n_dense;
  int32_t f1_W1, f1_b1, f1_W2, f1_b2, K, k0, f2_W1, f2_b1, f2_W2, f2_b2, gamma;
} tlsan_dense_layout;

/* One batch = the placeholders of model.py:27-53, int32 / fp32, as fed by
 * model.py:210-222 (train) and :239-262 (eval). */
typedef struct {
  int32_t B;                  /* rows in this batch                                     */
  int32_t Sn;                 /* columns of hist_i_new (max session length in batch)    */
  const int32_t* u;           /* [B]                                                    */
  const int32_t* i;           /* [B] candidate item (train target / eval positive)      */
  const int32_t* j;           /* [B] optional second candidate (eval negative) or NULL  */
  const float* y;             /* [B] labels (train only, may be NULL for forward)       */
  const int32_t* hist_i;      /* [B, Ls]                                                */
  const int32_t* hist_i_new;  /* [B, Sn]                                                */
  const float* hist_t;        /* [B, Ls]                                                */
  const int32_t* sl;          /* [B] valid length of hist_i  (1..Ls)                    */
  const int32_t* sl_new;      /* [B] valid length of hist_i_new (0..Sn)                 */
  const int32_t* u_cate;      /* [B]                                                    */
} tlsan_batch;

enum { TLSAN_NORM_TF18 = 0, TLSAN_NORM_DEDUP = 1 };
/* L2 term of model.py:164-172.  DENSE: every row of the four tables is decayed every step, as the
 * reference's dense gradient does.  LAZY: algebraically the same update,
 *   W <- (1 - lr c reg) W - lr c G_sparse,
 * kept as W = P * W_stored with one global scale P, so only rows that received a gradient are
 * read and written (identical up to fp32 rounding; required for tables that do not fit a
 * per-step sweep). */
enum { TLSAN_L2_DENSE = 0, TLSAN_L2_LAZY = 1 };

/* Hyper-parameters of one step: model.py:172 (regulation_rate), :201 (max_gradient_norm),
 * lr placeholder :49; optimizer = sgd (:195). */
typedef struct {
  float lr;
  float reg;
  float clip;
  int32_t norm_mode;  /* TLSAN_NORM_*: how clip_by_global_norm's norm treats repeated ids */
  int32_t l2_mode;    /* TLSAN_L2_*                                                       */
  /* The destination index of a batch (use counts, segment offsets, used-row records) depends only on
   * the batch's ids.  The state holds three index slots (0..2): index_slot picks the one this
   * step uses; index_prebuilt != 0 says tlsan_batch_index already built it (e.g. on a second stream
   * while the previous step ran), otherwise the step builds it itself.  Defaults 0, 0. */
  int32_t index_slot;
  int32_t index_prebuilt;
  /* config['dropout'] (model.py:116-118, 428-431): tf.nn.dropout with keep_prob = 1 - dropout on
   * the inputs of the two linear maps of both attention blocks, train steps only (forward / evaluation
   * never drop).  TF's random stream is not reproducible; the keep / drop pattern is a hash of
   * (dropout_seed, sample, block, position, map, channel) -- pass a different seed every step.
   * 0 = off (the reference's default).  Supported for fp32 and bf16 tables, fp32 and bf16 matrix products. */
  float dropout;
  uint32_t dropout_seed;
  uint32_t dropout_sample0;   /* position of this batch's first sample in the pattern: a rank's offset into the global batch */
} tlsan_hparams;

/* Device-side results of a train step (all optional except loss).
 *
 * Divergence: the reference's loss goes NaN when training diverges (model.py:171, printed at train.py:205) and
 * tf.clip_by_global_norm hands a non-finite norm on to every gradient.  The same holds here: a NaN or Inf in any
 * gathered row or weight reaches `loss` and `gnorm` (plain IEEE arithmetic carries it through the logit, the BCE and
 * the square sums; the fused kernel's units are built with -fno-honor-nans, which only frees the compiler from
 * canonicalising in front of fmaxf -- no comparison there decides whether a value is reported), and the clip
 * coefficient is formed so that a non-finite norm is not swallowed (clip_coef: NaN -> NaN, +Inf -> 0), i.e. the
 * step's update and the lazy table scale carry it on.  No separate "finite" flag: a caller tests the loss, as the
 * reference's driver does.  Pinned by tests/test_gpu_parity.py::test_nonfinite_inputs_give_nonfinite_loss. */
typedef struct {
  float* loss;    /* [1]  mean BCE + reg * l2 (model.py:171-172), value BEFORE the update */
  float* gnorm;   /* [1]  global gradient norm used for clipping                           */
  float* logits;  /* [B]  model.py:137 (may be NULL)                                       */
  float* sq_rows; /* [1]  sum of squares of every per-use embedding-gradient row (the sparse
                     part of TF-1.8's global norm); may be NULL                              */
  /* Optional (may be NULL; ABI 13): a word in HOST-visible memory (hipHostMalloc) into which the step's first
   * kernel stores `started_value` when it begins to run -- everything queued on the stream before the step has
   * completed by then.  A caller that must know this (to reuse an index slot on another stream) polls the word
   * instead of recording an event between two steps: an event is a barrier packet in the queue and costs the next
   * kernel ~5 us. */
  uint32_t* started;
  uint32_t started_value;
} tlsan_step_out;

int tlsan_abi_version(void);
const char* tlsan_last_error(void);

int tlsan_dense_layout_of(const tlsan_dims* dims, tlsan_dense_layout* out);

/* Bytes of scratch for batches up to (max_B, max_Sn).  Contents need not persist. */
size_t tlsan_workspace_bytes(const tlsan_dims* dims, int32_t max_B, int32_t max_Sn);

/* Bytes of persistent optimizer-side state (running sums of squares of the four
 * regularised tables for the L2 term / clip norm; scatter index counters). */
size_t tlsan_state_bytes(const tlsan_dims* dims);

/* (Re)initialise `state` from the current parameters and refresh dense_KT.  Call once
 * after creating / restoring / externally modifying parameters. */
int tlsan_state_init(const tlsan_dims* dims, const tlsan_params* p, void* state, void* stream);

/* Address (inside `state`) of the table scale P maintained by TLSAN_L2_LAZY steps. */
const float* tlsan_state_scale(const void* state);


/* Fold the scale into the tables (stored *= P, P = 1).  Call before reading the tables as plain
 * parameters (checkpoint) or when P gets small (long lazy runs: P ~ exp(-sum lr c reg)). */
int tlsan_state_renorm(const tlsan_dims* dims, const tlsan_params* p, void* state, void* stream);

/* Clear the use counters and rebuild the static category->items index for p->item_cate without
 * touching the sums of squares (used by callers whose item table changes every step). */
int tlsan_state_reindex(const tlsan_dims* dims, const tlsan_params* p, void* state, void* stream);

/* Rebuild only the category->items index for p->item_cate (same dims as before, use counters
 * untouched -- they are zero between steps): for callers whose item -> category map changes every
 * step while the table shape stays (the padded compact table of the sharded path). */
int tlsan_state_recategorize(const tlsan_dims* dims, const tlsan_params* p, void* state, void* stream);

/* Refresh derived copies (dense_KT) after the caller wrote p->dense. */
int tlsan_sync_derived(const tlsan_dims* dims, const tlsan_params* p, void* stream);

/* Forward only -- replaces `sess.run(self.logits)` (model.py:239-262).
 * logits_i[B] for candidate b->i; logits_j[B] for b->j when both are non-NULL;
 * u_t[B,d] (model.py:135) when non-NULL (input of tlsan_eval_ranks). */
int tlsan_forward(const tlsan_dims* dims, const tlsan_params* p, const tlsan_batch* b,
                  float* logits_i, float* logits_j, float* u_t,
                  void* ws, size_t ws_bytes, void* stream);

/* The same, and the two attention-weight tensors the reference keeps on the model (model.py:122: self.att0, self.att1 =
 * `soft` of feature_wise_attention, model.py:386-394, heads split along the batch axis) when non-NULL:
 *   att0[num_heads * B, Ls, d / num_heads]      long-term block, row h * B + b; exactly 0 past sl[b]
 *   att1[num_heads * B, 1 + Sn, d / num_heads]  short-term block, position 0 = the bridge; exactly 0 past 1 + sl_new[b]
 * One deviation, for a sample with sl[b] == 0 (an empty history: never in the reference's data, build_dataset.py:49-52):
 * the reference's softmax over a fully masked row (model.py:384-386) is the uniform 1 / Ls; here that row of att0 is 0.1
 * at every position when the window is held in registers (Ls <= 10) and exactly 0 when it is streamed (Ls > 10).  No
 * other output depends on it: the long-term vector of such a sample is 0 in the reference and here. */
int tlsan_forward_att(const tlsan_dims* dims, const tlsan_params* p, const tlsan_batch* b,
                      float* logits_i, float* logits_j, float* u_t, float* att0, float* att1,
                      void* ws, size_t ws_bytes, void* stream);

/* One optimisation step -- replaces `sess.run([self.loss, self.train_op])`
 * (model.py:208-234): forward, BCE + L2 loss, backward, global-norm clip, SGD update of
 * every trainable, deterministic (bitwise reproducible) scatter-add of the embedding
 * gradients.
 * TLSAN_L2_LAZY, how the update is issued (round 6; an implementation note, the results are those of the
 * reference's update either way): where it was measured to win the used rows are summed AND updated by one launch beside
 * the gradient finalize, with clip coefficient 1 -- clip_by_global_norm's coefficient whenever the norm does not exceed
 * the clip --, and a second launch commits the table scale, updates the dense weights and, after a clipped or non-finite
 * step only, corrects the rows (w_spec - (s_true - s_spec) g = w_old - s_true g up to one rounding: one ulp of the
 * SPECULATIVE value, which is why bf16 tables that the caches hold keep the form that waits).  Category rows that several
 * workgroups share (a few, large categories) are updated by the second launch, with the true coefficient.  Unclipped steps
 * are bit-equal to the form that waits for the coefficient.  `state` grew by 64 sum-of-squares records for it
 * (tlsan_state_bytes); nothing else in the ABI changed (TLSAN_ABI_VERSION stays 14). */
int tlsan_train_step(const tlsan_dims* dims, const tlsan_params* p, const tlsan_batch* b,
                     const tlsan_hparams* hp, const tlsan_step_out* out,
                     void* state, void* ws, size_t ws_bytes, void* stream);

/* The reference's other optimizers (model.py:188-193: adadelta | adam | rmsprop, TF-1.8 defaults).
 * Their per-parameter accumulators ("slots") are two more sets of tables with the shapes of
 * `tlsan_params` (fp32, own row strides; item_cate / dense_KT / scale / table_dtype are ignored):
 *   ADAM     (beta1 0.9, beta2 0.999, epsilon 1e-8):  slot1 = m, slot2 = v, zero-initialised;
 *            `step` = number of this update counted from 1 (the beta powers);
 *   RMSPROP  (beta1 = decay 0.9, beta2 = momentum 0.0, epsilon 1e-10): slot1 = rms (initialised
 *            to ONE as TF does), slot2 = momentum (zero);
 *   ADADELTA (beta1 = rho 0.95, epsilon 1e-8): slot1 = accum, slot2 = accum_update (zero).
 * Arithmetic of TF 1.8's training ops (core/kernels/training_ops.cc ApplyAdam / ApplyRMSProp /
 * ApplyAdadelta and their Sparse* forms).  The embedding gradients reach the optimizer as
 * IndexedSlices covering every row (gathers + the dense L2 term), so every row of the four
 * regularised tables is updated every step; item_b (not regularised) is updated where used
 * (RMSProp, Adadelta) or everywhere (Adam, whose sparse form decays m and v of all rows).
 * Needs hp->l2_mode == TLSAN_L2_DENSE and fp32 tables.  kind == TLSAN_OPT_SGD (or opt == NULL) is
 * tlsan_train_step. */
enum { TLSAN_OPT_SGD = 0, TLSAN_OPT_ADAM = 1, TLSAN_OPT_RMSPROP = 2, TLSAN_OPT_ADADELTA = 3 };
typedef struct {
  int32_t kind;
  int32_t step;
  float beta1, beta2, epsilon;
  const tlsan_params* slot1;
  const tlsan_params* slot2;
} tlsan_optimizer;
int tlsan_train_step_opt(const tlsan_dims* dims, const tlsan_params* p, const tlsan_batch* b,
                         const tlsan_hparams* hp, const tlsan_optimizer* opt, const tlsan_step_out* out,
                         void* state, void* ws, size_t ws_bytes, void* stream);

/* Device-resident batcher.  A sample set packed as CSR, every pointer a DEVICE
 * pointer (counterpart of the python lists `dataset.pkl` holds, build_dataset.py:58-59,71):
 * sample s has history hist[hist_off[s] .. hist_off[s+1]) with weights hist_t (same indexing),
 * current session sess[sess_off[s] .. sess_off[s+1]), user u[s], category cate[s],
 * target[s] = the candidate (train) / positive (test) item, second[s] = the 0/1 label (train) /
 * the negative item (test). */
typedef struct {
  int32_t n;
  const int32_t* u; const int32_t* cate;
  const int32_t* hist_off; const int32_t* hist; const float* hist_t;
  const int32_t* sess_off; const int32_t* sess;
  const int32_t* target; const int32_t* second;
} tlsan_packed;

/* tlsan_batch_pack -- replaces DataInput.__next__ / DataInputTest.__next__ (TLSAN/input.py:17-54,
 * 70-107) without touching the host: assemble samples order[lo .. lo + out->B) of `set` into the
 * caller-allocated device arrays of `out` (u, i, y (train) or j (test), hist_i [B,Ls],
 * hist_i_new [B,Sn], hist_t [B,Ls], sl, sl_new, u_cate).  out->Sn must be >= the longest session
 * of the batch (the reference pads to exactly the longest; pass that for identical shapes).
 * Bit-identical to the reference's arrays on the committed fixtures. */
int tlsan_batch_pack(const tlsan_packed* set, const int32_t* order, int32_t lo, const tlsan_batch* out,
                     int32_t Ls, int32_t is_test, void* stream);

/* Build the destination index of batch `b` into index slot `slot` (0 .. 2) of the state: the two
 * launches a step otherwise starts with.  Independent of the parameters, so an input pipeline can
 * run it for the next batches on another stream while the current step computes; the caller orders
 * it after the last step that used the same slot and before the step that consumes it.
 * slot | TLSAN_INDEX_FOR_LAZY_SGD: the index will be consumed by a lazy-L2 SGD train step only, which reaches the
 * user table's offsets through the batch's ids and the used-row records -- they are then written for the used rows
 * only (10 M users: 80 MB less per step).  An index built that way must not feed tlsan_grads or a dense-L2 step.
 * item_cate: params->item_cate of the tables the step will run on (tables with thousands of categories count item
 * uses per category as well; may be NULL below TLSAN_CSEG_MIN categories -- the only parameter-side input, and it is
 * not a trainable). */
#define TLSAN_INDEX_FOR_LAZY_SGD 0x100
int tlsan_batch_index(const tlsan_dims* dims, const tlsan_batch* b, const int32_t* item_cate, void* state, int32_t slot,
                      void* stream);

/* Gradients only (no update) -- what `tf.gradients(self.loss, trainables)` (model.py:198)
 * returns, with duplicate ids summed and reg*W added for the four regularised tables.
 * Used by the parity tests.  Each output has the shape of the parameter it mirrors. */
typedef struct {
  float* item_emb; float* item_b; float* user_emb; float* usert_emb; float* cate_emb;
  float* dense;    /* [n_dense] */
  /* row strides in floats of the four row outputs; 0 = packed (d_item, 1, d_item, Ls) */
  int32_t ld_item, ld_itemb, ld_user, ld_usert;
  /* != 0: rows that received no gradient are not written (the caller zeroed the buffers).  With
   * strides this lets item and user gradients land in ONE fused [rows, W] buffer when a row is
   * only ever used as one kind (the compact per-step table of the sharded path).
   * 2 (with hp->reg == 0 and the tf18 norm, the pure row-sum path): the four row outputs ARE one fused
   * table of rows [item_emb | item_b | pad] / [user_emb | usert_emb | pad] of width ld_item == ld_user;
   * a written row is written in full (its unowned tail cleared), so only rows that received no
   * gradient keep the caller's content -- the buffer need not be zeroed when every row sent on is used. */
  int32_t sparse;
} tlsan_grads_out;
int tlsan_grads(const tlsan_dims* dims, const tlsan_params* p, const tlsan_batch* b,
                const tlsan_hparams* hp, const tlsan_grads_out* g, const tlsan_step_out* out,
                void* state, void* ws, size_t ws_bytes, void* stream);

/* All-items scoring -- replaces the metric update ops on `eval_logits`
 * (model.py:140-156): for each row, the rank of `labels[b]` inside
 * u_t[b] . [item_emb || cate_emb[item_cate]]^T + item_b under tf.nn.top_k's order
 * (higher score first, ties -> lower item id first).  hit@k == (rank < k).
 * The [B, I] score matrix is never materialised. */
int tlsan_eval_ranks(const tlsan_dims* dims, const tlsan_params* p, const float* u_t,
                     const int32_t* labels, int32_t B, int32_t* ranks,
                     void* ws, size_t ws_bytes, void* stream);

/* Item-sharded all-items scoring (the evaluation half of the multi-GPU path):
 *   tlsan_eval_label_scores: scores[b] = u_t[b] . all_emb[labels[b]] + item_b[labels[b]] on a table that
 *     holds the label rows (the rank that owns the users: its compact per-step table);
 *   tlsan_eval_counts_shard: for every row of u_t (the users of ALL ranks, all-gathered) the number
 *     of items of THIS rank's shard ranked ahead of the label under tf.nn.top_k's order; local item n
 *     is global item n * id_mul + id_add (ties -> lower GLOBAL id first).  Summing the counts over
 *     the ranks gives tlsan_eval_ranks' result on the unsharded table. */
int tlsan_eval_label_scores(const tlsan_dims* dims, const tlsan_params* p, const float* u_t, const int32_t* labels,
                            int32_t B, float* scores, void* ws, size_t ws_bytes, void* stream);
int tlsan_eval_counts_shard(const tlsan_dims* dims, const tlsan_params* p, const float* u_t, const float* label_scores,
                            const int32_t* labels_global, int32_t B, int32_t id_mul, int32_t id_add, int32_t* counts,
                            void* ws, size_t ws_bytes, void* stream);

/* Deterministic scatter-apply on one row table (the owner-side half of the multi-GPU step, and
 * the stand-alone form of the embedding update of model.py:198-205):
 *   for every row r < nrows:  g = gscale * sum_{k: dest[k]==r} grows[k] (+ reg * W[r] on the
 *   first reg_cols columns);   W[r] -= (*step_dev) * g
 * Rows without contributions still decay (dense L2, as the reference).  The sum is exact
 * (order independent) -> bitwise reproducible.  width % 4 == 0, width <= 192.
 * sumsq_out (nullable, device double) receives sum over the first reg_cols columns of W_new^2.
 * step_dev is a DEVICE float (lr * clip coefficient) so no host sync is needed. */
size_t tlsan_rows_apply_workspace(int32_t nrows, int32_t n);
int tlsan_rows_apply(float* W, int32_t ld, int32_t nrows, int32_t width, int32_t reg_cols,
                     const float* grows, int32_t ldg, const int32_t* dest, int32_t n,
                     float gscale, const float* step_dev, float reg, double* sumsq_out,
                     void* ws, size_t ws_bytes, void* stream);

/* ---- device side of the row-sharded multi-GPU step (tlsan_amd/dist.py; no reference counterpart:
 * the reference is single-GPU, train.py:53,146).  Key space: G owners x R rows, key = owner*R + row.
 *
 * tlsan_route_plan: distinct rows a batch touches, grouped by owner (= all-to-all send order):
 *   keys [n_keys] (duplicates fine) -> rank[G*R] compact index of every key, uniq[<=n_keys]
 *   distinct keys ascending, n_uniq[1], cate_c[n_uniq] = cate_by_key[uniq], comp[n_keys] = rank[keys],
 *   sendbuf [G][1 + cap] = per owner {count, row numbers inside the owner's shard ...}: the payload of
 *   ONE equal-split all-to-all.  cap must be the same on every rank (the exchange is equal-split);
 *   when it is smaller than min(R, n_keys), what this batch may need, no rows are written and every
 *   count is -min(R, n_keys): all ranks then see the same negative counts after the exchange, agree on a
 *   larger cap and repeat the plan.  cate_c entries [n_uniq, cate_pad) are set
 *   to -1 (a compact table padded to a fixed row count).  flags[G*R] is scratch that must be zero
 *   on entry and is zero again on exit.  counts_out (optional, [G]): the per-owner counts once more,
 *   written by the kernel -- pass device-visible pinned host memory to have the exchange sizes on
 *   the host without a copy. */
int tlsan_route_plan(const int32_t* keys, int32_t n_keys, int32_t R, int32_t G, const int32_t* cate_by_key,
                     int32_t* flags, int32_t* rank, int32_t* uniq, int32_t* n_uniq, int32_t* sendbuf, int32_t cap,
                     int32_t* cate_c, int32_t cate_pad, int32_t* comp, int32_t* counts_out, void* stream);

/* tlsan_shard_gather: owner side of the row fetch.  recvbuf [G][1 + cap] as received from the G
 * ranks; n_recv = sum of the counts (host value).  rows_out [n_recv, W] = the requested rows of
 * shard [R, ld] in source-rank order (the all-to-all send order), recv_rows [n_recv] = their row
 * numbers (input of tlsan_shard_apply). */
int tlsan_shard_gather(const float* shard, int32_t ld, int32_t R, int32_t W, const int32_t* recvbuf, int32_t cap,
                       int32_t G, int32_t n_recv, float* rows_out, int32_t* recv_rows, void* stream);

/* tlsan_shard_summary: after the all-reduce (sum over G ranks) of
 *   flat = [dense grads n_dense | cate grads n_cate | mean BCE | per-use row squares | table squares | pad]:
 * global norm (tf18 rule), clip coefficient (model.py:201), loss (model.py:164-172), the device
 * step size lr*coef (step_dev[0]; step_dev[1] = coef), and the SGD update of the replicated dense
 * parameters (+ K^T copy). */
int tlsan_shard_summary(const float* flat, int32_t n_dense, int32_t n_cate, int32_t G, float lr, float reg, float clip,
                        const double* S_cate, float* dense, float* dense_KT, const tlsan_dims* dims,
                        float* step_dev, float* loss_out, float* gnorm_out, void* stream);

/* The other optimizers (tlsan_optimizer above: same kinds, defaults and arithmetic) on the sharded step:
 * accumulators laid out like what they belong to -- the shard rows [R, ld], cate_emb [C, dc] and the
 * dense parameters (the last two replicated: every rank applies the same update).  kind SGD / NULL =
 * the plain functions.  item_b (column reg_item of an item row) is touched where a gradient arrived
 * (sparse RMSProp / Adadelta) or everywhere (Adam), as on one GPU. */
typedef struct {
  int32_t kind, step;
  float beta1, beta2, epsilon;
  float* shard_s1; float* shard_s2;
  float* cate_s1; float* cate_s2;
  float* dense_s1; float* dense_s2;
  /* lazy L2 on the sharded step (kind SGD only; the accumulators may then be NULL): the tables are
   * scale[0] * stored, one scale that every rank advances alike (tlsan_shard_apply_lazy commits it);
   * step_dev then has 4 floats: {lr coef, coef, lr coef / P_new, P_new}. */
  float* scale;
} tlsan_shard_optimizer;
int tlsan_shard_summary_opt(const float* flat, int32_t n_dense, int32_t n_cate, int32_t G, float lr, float reg, float clip,
                            const double* S_cate, float* dense, float* dense_KT, const tlsan_dims* dims,
                            float* step_dev, float* loss_out, float* gnorm_out, const tlsan_shard_optimizer* opt,
                            void* stream);

/* tlsan_shard_apply: owner-side update of the fused shard table [R, W] (rows [0,cI) items with
 * reg_item regularised columns, rows [cI,R) users with reg_user) from the row gradients received
 * from every rank (vals[n_recv, ldv] for local rows `rows`, concatenated in source-rank order,
 * src_off[G+1] host array of the per-source offsets; rows of one source are distinct), and of the
 * replicated category table from g_cate [C, dc]:  W -= step * (gscale * sum + reg * W) on every
 * row (dense L2, as the reference).  Fixed summation order -> bitwise reproducible.
 * slots: int32 [R*G] device scratch that must be zero on entry and is zero again on exit.
 * sumsq_out[2] (device doubles): sums of squares of the regularised columns of shard / cate_emb
 * after the update; sumsq_f32 (nullable) receives (float)sumsq_out[0]. */
size_t tlsan_shard_apply_workspace(int32_t R, int32_t C);
int tlsan_shard_apply(float* shard, int32_t ld, int32_t cI, int32_t R, int32_t W, int32_t reg_item, int32_t reg_user,
                      const float* vals, int32_t ldv, const int32_t* rows, int32_t n_recv, const int32_t* src_off,
                      int32_t G, int32_t* slots, float gscale, const float* step_dev, float reg,
                      float* cate_emb, int32_t C, int32_t dc, const float* g_cate,
                      double* sumsq_out, float* sumsq_f32, void* ws, size_t ws_bytes, void* stream);
int tlsan_shard_apply_opt(float* shard, int32_t ld, int32_t cI, int32_t R, int32_t W, int32_t reg_item, int32_t reg_user,
                          const float* vals, int32_t ldv, const int32_t* rows, int32_t n_recv, const int32_t* src_off,
                          int32_t G, int32_t* slots, float gscale, const float* step_dev, float reg,
                          float* cate_emb, int32_t C, int32_t dc, const float* g_cate,
                          double* sumsq_out, float* sumsq_f32, const tlsan_shard_optimizer* opt, float lr,
                          void* ws, size_t ws_bytes, void* stream);

/* tlsan_shard_apply_lazy: the owner-side update in its lazy-L2 form,
 *   W_stored -= (lr coef / P_new) * gscale * sum   for the rows whose gradients arrived (and every row of the
 * replicated category table), item_b (column reg_item of item rows, not regularised, not scaled) -= lr coef * ...
 * -- the same update as tlsan_shard_apply's dense sweep up to fp32 rounding, touching n_recv rows instead
 * of R.  slots64: uint64 [R*G] scratch, any content (entries carry `stamp`, which must differ from step
 * to step and never be 0 for a zero-initialised buffer).  sumsq_out[0] += the change of the stored shard
 * rows' sum of squares (regularised columns), sumsq_out[1] = that of cate_emb (stored values).
 * ws: tlsan_shard_apply_lazy_workspace(n_recv, C) bytes. */
size_t tlsan_shard_apply_lazy_workspace(int32_t n_recv, int32_t C);
int tlsan_shard_apply_lazy(float* shard, int32_t ld, int32_t cI, int32_t R, int32_t W, int32_t reg_item, int32_t reg_user,
                           const float* vals, int32_t ldv, const int32_t* rows, int32_t n_recv, const int32_t* src_off,
                           int32_t G, uint64_t* slots64, uint32_t stamp, float gscale, const float* step_dev,
                           float* cate_emb, int32_t C, int32_t dc, const float* g_cate,
                           double* sumsq_out, float* sumsq_f32, float* scale, void* ws, size_t ws_bytes, void* stream);

/* ---- static-shape forms of the sharded step (tlsan_amd/dist.py, ShardedModel(static_rows=True)).
 * Every (source, owner) pair exchanges exactly `cap` row slots in both directions, so no size ever has to reach
 * the host: the ids, the rows and the gradients travel in equal-split all-to-alls of fixed size, every kernel
 * argument is a constant of the shape, and a step can be recorded in a HIP graph.  The step's compact table is
 * numbered by slot: the j-th distinct row of owner g is compact row g * cap + j (at one rank: the plain compact
 * numbering); unused slots are in no category and never referenced.
 *
 * tlsan_route_plan_static: as tlsan_route_plan with that numbering; cate_c has G * cap entries.
 *   counts_out (optional [G]): the TRUE per-owner counts; the headers of sendbuf are clamped to cap.
 *   status [1] (device int32, zero-initialised by the caller): atomicMax of every count that exceeded cap.  A
 *   batch that needs more than cap rows of one owner is truncated -- that step is wrong; the host checks status
 *   when it next synchronises and must report it (cap >= min(R, n_keys) can never overflow).
 * tlsan_shard_gather_static: rows_out [G * cap, W], recv_rows [G * cap] (-1 = empty slot); when slots64 is
 *   given, the lazy apply's slot marks are written here (saves a launch): stamp is a DEVICE uint32 (never 0).
 * tlsan_shard_apply_lazy_static: tlsan_shard_apply_lazy over the G * cap slots (rows[e] < 0: skipped); the
 *   device-side stamp is advanced for the next step.  marked != 0: the slot marks were written by the gather. */
int tlsan_route_plan_static(const int32_t* keys, int32_t n_keys, int32_t R, int32_t G, const int32_t* cate_by_key,
                            int32_t* flags, int32_t* rank, int32_t* uniq, int32_t* n_uniq, int32_t* sendbuf, int32_t cap,
                            int32_t* cate_c, int32_t* comp, int32_t* counts_out, int32_t* status, void* stream);
int tlsan_shard_gather_static(const float* shard, int32_t ld, int32_t R, int32_t W, const int32_t* recvbuf, int32_t cap,
                              int32_t G, float* rows_out, int32_t* recv_rows, uint64_t* slots64, const uint32_t* stamp,
                              void* stream);
/* tlsan_shard_gather_wire_bf16: tlsan_shard_gather_static with bf16 rows on the wire.  Owners keep fp32 rows; slot e
 * of rows_out (pitch bytes per slot, a multiple of 16) receives
 *   [ the first d_emb floats of the row as bf16, round to nearest even | the next `tail` floats as fp32 | pad ]
 * -- for the fused [item_emb | item_b] / [user_emb | usert_emb] rows of tlsan_amd/dist.py: tail = max(1, Ls).  The
 * consumer points its tlsan_params, table_dtype = TLSAN_TABLE_BF16, into the slots: item_emb / user_emb at the slot base
 * with ld = pitch / 2, item_b / usert_emb at base + 2 * d_emb bytes with ld = pitch / 4. */
int tlsan_shard_gather_wire_bf16(const float* shard, int32_t ld, int32_t R, int32_t d_emb, int32_t tail,
                                 const int32_t* recvbuf, int32_t cap, int32_t G, void* rows_out, int32_t pitch,
                                 int32_t* recv_rows, uint64_t* slots64, const uint32_t* stamp, void* stream);
int tlsan_shard_apply_lazy_static(float* shard, int32_t ld, int32_t cI, int32_t R, int32_t W, int32_t reg_item, int32_t reg_user,
                                  const float* vals, int32_t ldv, const int32_t* rows, int32_t cap, int32_t G,
                                  uint64_t* slots64, uint32_t* stamp, int32_t marked, float gscale, const float* step_dev,
                                  float* cate_emb, int32_t C, int32_t dc, const float* g_cate,
                                  double* sumsq_out, float* sumsq_f32, float* scale,
                                  void* ws, size_t ws_bytes, void* stream);

/* ---- the static-shape step in as few host calls as there are collectives (round 4) -------------------------------
 * tlsan_amd/dist.py issued the step as ~15 Python-level calls (ctypes with 10-27 arguments each, torch events, a
 * stream-scoped copy): at one rank the HOST took 100 us per step for 77 us of kernels -- every rank of a multi-GPU run
 * would have been bound by its Python thread.  The two entry points below run the same launches from argument blocks
 * that are built once per (plan slot, batch): one call per segment between two collectives.
 *
 * tlsan_shard_step_static(s, phases, stream): the main stream's launches, `phases` = any contiguous run of
 *   TLSAN_PHASE_GATHER  tlsan_shard_gather_static / _wire_bf16 (s->wire)       -- then the row all-to-all
 *   TLSAN_PHASE_GRADS   tlsan_grads on the compact table                        -- then the all-reduce of s->flat
 *   TLSAN_PHASE_SUMMARY tlsan_shard_summary_opt                                 -- then the gradient all-to-all
 *   TLSAN_PHASE_APPLY   tlsan_shard_apply_lazy_static
 *   (one rank: all four in one call).  `plans` (optional) are issued behind the call's last launch -- they run on their own
 *   streams, and their slots were last used by earlier steps: with s->out.started set (a pinned host word the fused
 *   kernel stores s->out.started_value into when it begins to run; values count up from step to step) the call first
 *   waits, on the HOST, until the word has reached s->plans_after -- "step plans_after has started" = every step before
 *   it is complete -- instead of ordering the side streams behind an event recorded on the main stream (a barrier
 *   packet in the main queue: ~6 us of idle GPU per step).  Without the word the plans carry ev_fork.
 * tlsan_shard_plan_static(p): a batch's routing plan, category index and destination index on p->stream / p->stream2
 *   (tlsan_route_plan_static -- its closing kernel stores an overflow count into the pinned word --, tlsan_state_recategorize, tlsan_batch_index),
 *   ordered by the caller's HIP events (raw hipEvent_t / hipStream_t handles; NULL events are skipped):
 *     stream waits ev_fork; ev_planned recorded behind the route plan; stream2 waits ev_planned, builds the index,
 *     records ev_done1; ev_done0 recorded on stream last when record_done0 != 0 (the caller records it itself when it
 *     puts the id all-to-all there).  stream2 == NULL: the index follows on stream. */
#define TLSAN_PHASE_GATHER 1
#define TLSAN_PHASE_GRADS 2
#define TLSAN_PHASE_SUMMARY 4
#define TLSAN_PHASE_APPLY 8
#define TLSAN_PLAN_ASYNC 256   /* `plans` are issued by the library's launch thread (see tlsan_shard_plans_flush) */
typedef struct {
  const int32_t* keys; int32_t n_keys, R, G; const int32_t* cate_by_key;
  int32_t *flags, *rank, *uniq, *n_uniq, *sendbuf; int32_t cap;
  int32_t *cate_c, *comp, *status;
  int32_t* status_host;                  /* optional PINNED host word: receives the count of an overflowing owner (a system-scope store of the plan's kernel) */
  const tlsan_dims* dims; const tlsan_params* cp; const tlsan_batch* cb; void* state;
  void *stream, *stream2;                /* hipStream_t */
  void *ev_fork, *ev_planned, *ev_done0, *ev_done1;   /* hipEvent_t or NULL */
  int32_t record_done0;
} tlsan_static_plan;
typedef struct {
  /* gather */
  const float* shard; int32_t ld, R, W; const int32_t* recvbuf; int32_t cap, G; void* rows_out; int32_t* recv_rows;
  uint64_t* slots64; uint32_t* stamp;
  int32_t wire, d_emb, tail, pitch;      /* wire != 0: tlsan_shard_gather_wire_bf16(d_emb, tail, pitch) */
  /* grads */
  const tlsan_dims* dims; const tlsan_params* cp; const tlsan_batch* cb; tlsan_hparams hp; tlsan_grads_out go; tlsan_step_out out;
  void* state; void* ws; size_t ws_bytes;
  /* summary */
  float* flat; int32_t n_dense, n_cate; float lr, reg, clip; const double* S_cate; float* dense; float* dense_KT;
  const tlsan_dims* dims_full; float* step_dev; float* loss_out; float* gnorm_out; const tlsan_shard_optimizer* opt;
  /* apply */
  int32_t cI, reg_item, reg_user; const float* vals; int32_t ldv, marked; float gscale; float* cate_emb; int32_t C, dc;
  const float* g_cate; double* sumsq_out; float* sumsq_f32; float* scale; void* lws; size_t lws_bytes;
  /* plans: issued once the pinned word out.started has reached this value (steps count up; see below) */
  uint32_t plans_after;
} tlsan_static_step;
int tlsan_shard_plan_static(const tlsan_static_plan* p);
/* With TLSAN_PLAN_ASYNC in `phases` (needs s->out.started) the plans are copied and handed to ONE launch thread of the
 * library's own, which waits for the pinned word and issues them while the caller goes on with the main stream: a plan
 * is seven launches + a copy + four event operations, the step six launches, and one host thread issuing both was the
 * step's bound (85 us of HIP runtime calls per step for 77 us of kernels).  The only threading in the library (the header's
 * "single host thread per GPU" holds for every other entry point).  tlsan_shard_plans_flush() returns when the thread has
 * issued everything handed to it, with its first error if any: call it before waiting on a plan's events, before
 * touching what a queued plan writes (discarding a slot), before a stream capture, before tearing the streams down. */
int tlsan_shard_plans_flush(void);
int tlsan_shard_step_static(const tlsan_static_step* s, int32_t phases, const tlsan_static_plan* const* plans, int32_t n_plans,
                            void* stream);

/* Exclusive prefix sum + compaction of a device int32 array (the id-routing step of the sharded
 * path): prefix[k] = sum(cnt[0..k)), uniq = ascending list of k with cnt[k] > 0 and
 * prefix_nz[k]... see below.  For 0/1 flags, prefix[k] is the compact index of k.
 *   prefix  [n]  exclusive prefix of cnt
 *   uniq    [n]  (nullable) indices with cnt > 0, ascending
 *   n_uniq  [1]  (nullable) how many */
int tlsan_scan_compact(const int32_t* cnt, int32_t n, int32_t* prefix, int32_t* uniq, int32_t* n_uniq, void* stream);

/* Profiling hooks (measurement only; no reference counterpart).  level 0 = off (default),
 * 1 = HIP events around the fused forward/backward kernel of every train step,
 * 2 = events at every kernel boundary of the step.  Events are recorded on the step's stream.
 * tlsan_profile_collect waits for the recorded events and writes, per recorded step, 5 floats
 * of milliseconds {index build, fwd+bwd kernel, dK partial, dense finalize, row apply}
 * (level 1 fills only [1]); returns the number of steps written and clears the ring. */
#define TLSAN_PROF_SEGMENTS 5
int tlsan_profile_enable(int level);
/* record only every `every`-th step (default 1): timing events perturb a latency-bound pipeline */
int tlsan_profile_stride(int every);
int tlsan_profile_collect(float* host_ms, int max_steps);

/* Diagnostic: device buffer of s_memtime stamps (NULL = off, the default): entries
 * [blocks*8 waves][16] written by the fused kernel at its phase boundaries and, from entry
 * 2^20 on, [workgroups][8] written by k_apply (the buffer must hold 2^20 + 8*grid entries). */
int tlsan_debug_stamps(void* device_buf);

#ifdef __cplusplus
}
#endif
#endif /* TLSAN_H_ */
