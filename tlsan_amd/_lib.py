"""ctypes binding of libtlsan_hip.so (include/tlsan.h).  The product path has no CPU
fallback: if the HIP library is missing or fails to load, importing users get a loud error."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (TLSAN_LIB_PATH: a diagnostic build of the same library, e.g. the -DTLSAN_STAMPS=1 variant of scripts/stamps.py)
LIB_PATH = os.environ.get("TLSAN_LIB_PATH") or os.path.join(HERE, "libtlsan_hip.so")

ABI_VERSION = 14
NORM_TF18, NORM_DEDUP = 0, 1
TABLE_F32, TABLE_BF16 = 0, 1
MATRIX_F32, MATRIX_BF16 = 0, 1
L2_DENSE, L2_LAZY = 0, 1
INDEX_FOR_LAZY_SGD = 0x100   # TLSAN_INDEX_FOR_LAZY_SGD (include/tlsan.h)
INDEX_SLOTS = 3   # TLSAN_INDEX_SLOTS (csrc/tlsan_update.h): destination-index slots of the state
SN_CAP = 96     # TLSAN_SN_CAP (csrc/tlsan_common.h): longest session of a training batch

EXPORTS = [
    "tlsan_abi_version", "tlsan_last_error", "tlsan_dense_layout_of", "tlsan_workspace_bytes",
    "tlsan_state_bytes", "tlsan_state_init", "tlsan_state_reindex", "tlsan_state_recategorize", "tlsan_state_scale",
    "tlsan_state_renorm", "tlsan_sync_derived", "tlsan_forward", "tlsan_forward_att",
    "tlsan_train_step", "tlsan_train_step_opt", "tlsan_batch_pack", "tlsan_batch_index", "tlsan_grads", "tlsan_eval_ranks", "tlsan_eval_label_scores", "tlsan_eval_counts_shard", "tlsan_profile_enable", "tlsan_profile_stride",
    "tlsan_profile_collect", "tlsan_debug_stamps", "tlsan_rows_apply_workspace", "tlsan_rows_apply", "tlsan_scan_compact",
    "tlsan_route_plan", "tlsan_shard_gather", "tlsan_shard_summary", "tlsan_shard_apply_workspace", "tlsan_shard_apply",
    "tlsan_shard_summary_opt", "tlsan_shard_apply_opt", "tlsan_shard_apply_lazy_workspace", "tlsan_shard_apply_lazy",
    "tlsan_route_plan_static", "tlsan_shard_gather_static", "tlsan_shard_apply_lazy_static", "tlsan_shard_gather_wire_bf16",
    "tlsan_shard_plan_static", "tlsan_shard_step_static", "tlsan_shard_plans_flush",
]
PROF_SEGMENTS = ("index_build", "fwd_bwd", "dk_partial", "dense_finalize", "apply_rows")


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("user_count", "item_count", "cate_count", "d", "d_item", "d_cate", "num_heads", "Ls")]


class Params(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("item_emb", "item_b", "user_emb", "usert_emb", "cate_emb", "dense", "dense_KT", "item_cate")] + \
               [(n, C.c_int32) for n in ("ld_item", "ld_itemb", "ld_user", "ld_usert")] + [("scale", C.c_void_p), ("table_dtype", C.c_int32), ("matrix_dtype", C.c_int32)]


class DenseLayout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n_dense", "f1_W1", "f1_b1", "f1_W2", "f1_b2", "K", "k0", "f2_W1", "f2_b1", "f2_W2",
                 "f2_b2", "gamma")]


class Batch(C.Structure):
    _fields_ = [("B", C.c_int32), ("Sn", C.c_int32)] + [(n, C.c_void_p) for n in
                ("u", "i", "j", "y", "hist_i", "hist_i_new", "hist_t", "sl", "sl_new", "u_cate")]


class Packed(C.Structure):
    _fields_ = [("n", C.c_int32)] + [(n, C.c_void_p) for n in
                ("u", "cate", "hist_off", "hist", "hist_t", "sess_off", "sess", "target", "second")]


class HParams(C.Structure):
    _fields_ = [("lr", C.c_float), ("reg", C.c_float), ("clip", C.c_float),
                ("norm_mode", C.c_int32), ("l2_mode", C.c_int32), ("index_slot", C.c_int32), ("index_prebuilt", C.c_int32),
                ("dropout", C.c_float), ("dropout_seed", C.c_uint32), ("dropout_sample0", C.c_uint32)]


class Optimizer(C.Structure):
    _fields_ = [("kind", C.c_int32), ("step", C.c_int32), ("beta1", C.c_float), ("beta2", C.c_float),
                ("epsilon", C.c_float), ("slot1", C.c_void_p), ("slot2", C.c_void_p)]


OPT_SGD, OPT_ADAM, OPT_RMSPROP, OPT_ADADELTA = 0, 1, 2, 3


class ShardOptimizer(C.Structure):
    _fields_ = [("kind", C.c_int32), ("step", C.c_int32), ("beta1", C.c_float), ("beta2", C.c_float),
                ("epsilon", C.c_float)] + [(n, C.c_void_p) for n in ("shard_s1", "shard_s2", "cate_s1", "cate_s2",
                                                                      "dense_s1", "dense_s2", "scale")]


class StepOut(C.Structure):
    _fields_ = [("loss", C.c_void_p), ("gnorm", C.c_void_p), ("logits", C.c_void_p), ("sq_rows", C.c_void_p),
                ("started", C.c_void_p), ("started_value", C.c_uint32)]   # (host-visible word + value: include/tlsan.h)


class GradsOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("item_emb", "item_b", "user_emb", "usert_emb", "cate_emb", "dense")] + \
               [(n, C.c_int32) for n in ("ld_item", "ld_itemb", "ld_user", "ld_usert", "sparse")]


class StaticPlan(C.Structure):      # tlsan_static_plan (include/tlsan.h)
    _fields_ = [("keys", C.c_void_p), ("n_keys", C.c_int32), ("R", C.c_int32), ("G", C.c_int32), ("cate_by_key", C.c_void_p),
                ("flags", C.c_void_p), ("rank", C.c_void_p), ("uniq", C.c_void_p), ("n_uniq", C.c_void_p), ("sendbuf", C.c_void_p),
                ("cap", C.c_int32), ("cate_c", C.c_void_p), ("comp", C.c_void_p), ("status", C.c_void_p), ("status_host", C.c_void_p),
                ("dims", C.POINTER(Dims)), ("cp", C.POINTER(Params)), ("cb", C.POINTER(Batch)), ("state", C.c_void_p),
                ("stream", C.c_void_p), ("stream2", C.c_void_p),
                ("ev_fork", C.c_void_p), ("ev_planned", C.c_void_p), ("ev_done0", C.c_void_p), ("ev_done1", C.c_void_p),
                ("record_done0", C.c_int32)]


class StaticStep(C.Structure):      # tlsan_static_step (include/tlsan.h)
    _fields_ = [("shard", C.c_void_p), ("ld", C.c_int32), ("R", C.c_int32), ("W", C.c_int32), ("recvbuf", C.c_void_p),
                ("cap", C.c_int32), ("G", C.c_int32), ("rows_out", C.c_void_p), ("recv_rows", C.c_void_p),
                ("slots64", C.c_void_p), ("stamp", C.c_void_p),
                ("wire", C.c_int32), ("d_emb", C.c_int32), ("tail", C.c_int32), ("pitch", C.c_int32),
                ("dims", C.POINTER(Dims)), ("cp", C.POINTER(Params)), ("cb", C.POINTER(Batch)), ("hp", HParams), ("go", GradsOut),
                ("out", StepOut), ("state", C.c_void_p), ("ws", C.c_void_p), ("ws_bytes", C.c_size_t),
                ("flat", C.c_void_p), ("n_dense", C.c_int32), ("n_cate", C.c_int32), ("lr", C.c_float), ("reg", C.c_float),
                ("clip", C.c_float), ("S_cate", C.c_void_p), ("dense", C.c_void_p), ("dense_KT", C.c_void_p),
                ("dims_full", C.POINTER(Dims)), ("step_dev", C.c_void_p), ("loss_out", C.c_void_p), ("gnorm_out", C.c_void_p),
                ("opt", C.POINTER(ShardOptimizer)),
                ("cI", C.c_int32), ("reg_item", C.c_int32), ("reg_user", C.c_int32), ("vals", C.c_void_p), ("ldv", C.c_int32),
                ("marked", C.c_int32), ("gscale", C.c_float), ("cate_emb", C.c_void_p), ("C", C.c_int32), ("dc", C.c_int32),
                ("g_cate", C.c_void_p), ("sumsq_out", C.c_void_p), ("sumsq_f32", C.c_void_p), ("scale", C.c_void_p),
                ("lws", C.c_void_p), ("lws_bytes", C.c_size_t), ("plans_after", C.c_uint32)]


PHASE_GATHER, PHASE_GRADS, PHASE_SUMMARY, PHASE_APPLY, PLAN_ASYNC = 1, 2, 4, 8, 256

_lib = None


def load():
    """Load the shared library (building nothing).  Raises RuntimeError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own HIP runtime; import it first so this process has exactly one
    # libamdhip64 (loading ours first leaves two runtimes and no visible device in one of them)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "tlsan_amd: %s not found -- build it with `python -m tlsan_amd.build` "
            "(hipcc, --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    lib.tlsan_abi_version.restype = C.c_int
    lib.tlsan_last_error.restype = C.c_char_p
    lib.tlsan_dense_layout_of.argtypes = [P(Dims), P(DenseLayout)]
    lib.tlsan_workspace_bytes.argtypes = [P(Dims), C.c_int32, C.c_int32]
    lib.tlsan_workspace_bytes.restype = C.c_size_t
    lib.tlsan_state_bytes.argtypes = [P(Dims)]
    lib.tlsan_state_bytes.restype = C.c_size_t
    lib.tlsan_state_init.argtypes = [P(Dims), P(Params), C.c_void_p, C.c_void_p]
    lib.tlsan_sync_derived.argtypes = [P(Dims), P(Params), C.c_void_p]
    lib.tlsan_state_reindex.argtypes = [P(Dims), P(Params), C.c_void_p, C.c_void_p]
    lib.tlsan_state_reindex.restype = C.c_int
    lib.tlsan_state_recategorize.argtypes = [P(Dims), P(Params), C.c_void_p, C.c_void_p]
    lib.tlsan_state_recategorize.restype = C.c_int
    lib.tlsan_state_scale.argtypes = [C.c_void_p]
    lib.tlsan_state_scale.restype = C.c_void_p
    lib.tlsan_state_renorm.argtypes = [P(Dims), P(Params), C.c_void_p, C.c_void_p]
    lib.tlsan_state_renorm.restype = C.c_int
    lib.tlsan_forward.argtypes = [P(Dims), P(Params), P(Batch), C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.c_void_p]
    lib.tlsan_forward_att.argtypes = [P(Dims), P(Params), P(Batch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p]
    lib.tlsan_train_step.argtypes = [P(Dims), P(Params), P(Batch), P(HParams), P(StepOut),
                                     C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.tlsan_train_step_opt.argtypes = [P(Dims), P(Params), P(Batch), P(HParams), P(Optimizer), P(StepOut),
                                         C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.tlsan_train_step_opt.restype = C.c_int
    lib.tlsan_batch_pack.argtypes = [P(Packed), C.c_void_p, C.c_int32, P(Batch), C.c_int32, C.c_int32, C.c_void_p]
    lib.tlsan_batch_pack.restype = C.c_int
    lib.tlsan_batch_index.argtypes = [P(Dims), P(Batch), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.tlsan_batch_index.restype = C.c_int
    lib.tlsan_grads.argtypes = [P(Dims), P(Params), P(Batch), P(HParams), P(GradsOut), P(StepOut),
                                C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.tlsan_eval_ranks.argtypes = [P(Dims), P(Params), C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.tlsan_eval_label_scores.argtypes = [P(Dims), P(Params), C.c_void_p, C.c_void_p, C.c_int32,
                                            C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.tlsan_eval_label_scores.restype = C.c_int
    lib.tlsan_eval_counts_shard.argtypes = [P(Dims), P(Params), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.tlsan_eval_counts_shard.restype = C.c_int
    lib.tlsan_profile_enable.argtypes = [C.c_int]
    lib.tlsan_profile_enable.restype = C.c_int
    lib.tlsan_profile_stride.argtypes = [C.c_int]
    lib.tlsan_profile_stride.restype = C.c_int
    lib.tlsan_profile_collect.argtypes = [C.c_void_p, C.c_int]
    lib.tlsan_profile_collect.restype = C.c_int
    lib.tlsan_rows_apply_workspace.argtypes = [C.c_int32, C.c_int32]
    lib.tlsan_rows_apply_workspace.restype = C.c_size_t
    lib.tlsan_rows_apply.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                     C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_float, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]
    lib.tlsan_rows_apply.restype = C.c_int
    lib.tlsan_scan_compact.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.tlsan_scan_compact.restype = C.c_int
    lib.tlsan_route_plan.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 6 + \
                                    [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.tlsan_shard_gather.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.tlsan_shard_gather.restype = C.c_int
    lib.tlsan_route_plan.restype = C.c_int
    lib.tlsan_shard_summary.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float,
                                        C.c_void_p, C.c_void_p, C.c_void_p, P(Dims), C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]
    lib.tlsan_shard_summary.restype = C.c_int
    lib.tlsan_shard_summary_opt.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float,
                                            C.c_void_p, C.c_void_p, C.c_void_p, P(Dims), C.c_void_p, C.c_void_p, C.c_void_p,
                                            P(ShardOptimizer), C.c_void_p]
    lib.tlsan_shard_summary_opt.restype = C.c_int
    lib.tlsan_shard_apply_opt.argtypes = [C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                          P(C.c_int32), C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_float,
                                          C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                          P(ShardOptimizer), C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.tlsan_shard_apply_opt.restype = C.c_int
    lib.tlsan_shard_apply_lazy_workspace.argtypes = [C.c_int32, C.c_int32]
    lib.tlsan_shard_apply_lazy_workspace.restype = C.c_size_t
    lib.tlsan_shard_apply_lazy.argtypes = [C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                           P(C.c_int32), C.c_int32, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p,
                                           C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.tlsan_shard_apply_lazy.restype = C.c_int
    lib.tlsan_route_plan_static.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 6 + \
                                           [C.c_int32] + [C.c_void_p] * 5
    lib.tlsan_route_plan_static.restype = C.c_int
    lib.tlsan_shard_gather_static.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.tlsan_shard_gather_static.restype = C.c_int
    lib.tlsan_shard_gather_wire_bf16.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                                 C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.tlsan_shard_gather_wire_bf16.restype = C.c_int
    lib.tlsan_shard_apply_lazy_static.argtypes = [C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                                  C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p,
                                                  C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.tlsan_shard_apply_lazy_static.restype = C.c_int
    lib.tlsan_shard_plan_static.argtypes = [P(StaticPlan)]
    lib.tlsan_shard_plan_static.restype = C.c_int
    lib.tlsan_shard_step_static.argtypes = [P(StaticStep), C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    lib.tlsan_shard_step_static.restype = C.c_int
    lib.tlsan_shard_plans_flush.argtypes = []
    lib.tlsan_shard_plans_flush.restype = C.c_int
    lib.tlsan_shard_apply_workspace.argtypes = [C.c_int32, C.c_int32]
    lib.tlsan_shard_apply_workspace.restype = C.c_size_t
    lib.tlsan_shard_apply.argtypes = [C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                      P(C.c_int32), C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_float,
                                      C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p]
    lib.tlsan_shard_apply.restype = C.c_int
    lib.tlsan_debug_stamps.argtypes = [C.c_void_p]
    lib.tlsan_debug_stamps.restype = C.c_int
    for name in ("tlsan_dense_layout_of", "tlsan_state_init", "tlsan_sync_derived", "tlsan_forward", "tlsan_forward_att",
                 "tlsan_train_step", "tlsan_grads", "tlsan_eval_ranks"):
        getattr(lib, name).restype = C.c_int
    if lib.tlsan_abi_version() != ABI_VERSION:
        raise RuntimeError("tlsan_amd: ABI mismatch (library %d, binding %d)"
                           % (lib.tlsan_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


class TlsanError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        msg = load().tlsan_last_error()
        raise TlsanError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))
