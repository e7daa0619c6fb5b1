"""`Model`: the reference's ``TLSAN/model.py`` call surface over the HIP library.

Same constructor and methods as the reference class (model.py:13-313) so the reference's
``train.py`` loop drives it unchanged apart from the TensorFlow session object:

    Model(config, item_cate_list)
    .train(sess, batch, lr, add_summary=False) -> float     model.py:208-234
    .eval_auc(sess, batch) -> float                         model.py:237-263
    .eval_prec(sess, batch) / .eval_recall(sess, batch)     model.py:265-299
    .save(sess) / .restore(sess, path)                      model.py:302-313
    .global_step / .global_epoch_step / .global_epoch_step_op  (objects with .eval())
    .prec_1 ... .prec_50, .recall_1 ... .recall_50             (objects with .eval())
    .train_writer / .eval_writer                               (objects with .add_summary())

``sess`` is accepted and ignored (the reference passes a tf.Session).  ``batch`` is the 9-tuple
of ``input.py``.  All arithmetic runs in libtlsan_hip.so on the GPU; torch is used only to own
device memory and the stream.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

from . import _lib as L

KS = (1, 10, 20, 30, 40, 50)  # model.py:144-156


class _Var:
    """Stand-in for the tf.Variable / tensor handles train.py calls .eval() on."""

    def __init__(self, getter):
        self._get = getter

    def eval(self, session=None):  # noqa: A003 - name fixed by the reference
        return self._get()


class _Writer:
    """tf.summary.FileWriter stand-in (model.py:18-19; train.py:91-118): scalar summaries as rows
    `step,tag,value` appended to <logdir>/scalars.csv (histogram summaries of model.py:174-181 are kept
    as their count / min / max / mean / std).  Also keeps the rows in memory (`.rows`)."""

    def __init__(self, path):
        self.path = path
        self.rows = []
        self._pending = []

    def add_summary(self, summary=None, global_step=None):
        self.rows.append((global_step, summary))
        items = summary if isinstance(summary, list) else [summary]
        for it in items:
            if isinstance(it, tuple) and len(it) == 2:
                self._pending.append((global_step, it[0], it[1]))
        if len(self._pending) >= 64:
            self.flush()

    def flush(self):
        if not self._pending:
            return
        os.makedirs(self.path, exist_ok=True)
        with open(os.path.join(self.path, "scalars.csv"), "a") as f:
            for step, tag, val in self._pending:
                f.write("%s,%s,%.9g\n" % ("" if step is None else int(step), tag, float(val)))
        self._pending = []


def glorot_uniform(rng, shape):
    """TF-1.x default initializer of tf.get_variable (model.py:62-64,70-72,79-81,446)."""
    limit = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


DENSE_KEYS = ("fwa1_W1", "fwa1_b1", "fwa1_W2", "fwa1_b2", "dense_K", "dense_b",
              "fwa2_W1", "fwa2_b1", "fwa2_W2", "fwa2_b2", "gamma")
TABLE_KEYS = ("item_emb", "item_b", "user_emb", "usert_emb", "cate_emb")
# optimizer -> (TLSAN_OPT_*, beta1 | decay | rho, beta2 | momentum, epsilon): TF 1.8's constructor
# defaults, which model.py:188-193 keeps (only learning_rate is passed)
OPTIMIZERS = {"sgd": (L.OPT_SGD, 0.0, 0.0, 0.0), "adam": (L.OPT_ADAM, 0.9, 0.999, 1e-8),
              "rmsprop": (L.OPT_RMSPROP, 0.9, 0.0, 1e-10), "adadelta": (L.OPT_ADADELTA, 0.95, 0.0, 1e-8)}


_STREAM_CACHE = {}
_STARTED_WORDS = []   # pinned words that kernels of queued steps write (Model._started): never freed


def concurrent_streams(device, want=1, pool=6, **stream_kw):
    """`want` new streams whose work can run WHILE the current stream is busy, and while each other is.

    HIP multiplexes streams onto a few hardware queues (four by default), round-robin over every stream the process
    has used -- torch's, RCCL's, ours.  Two streams that land on one queue execute in launch order like a single
    stream: a "side" stream that shares the main stream's queue overlaps nothing (measured on the sharded step, one
    rank: 143 us with the plan streams colliding with the main stream, 122 with one of them, 107 with neither).
    Which queue a new stream gets cannot be asked for, but it can be observed: a spin kernel on stream a, a tiny kernel
    launched after it on stream b -- b finishes early exactly when the two are on different queues."""
    dev = torch.device(device)
    main = torch.cuda.current_stream(dev)
    # one set per (device, current stream) and process: every new stream is one more tenant of the four queues, and a
    # process that builds several models (bench.py's variants) would end up with side streams sharing queues again
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), main.cuda_stream, tuple(sorted(stream_kw.items())))
    have = _STREAM_CACHE.setdefault(key, [])
    if len(have) >= want:       # (bench.py's third model, with streams of its own: 98-113 us/step instead of 57)
        return have[:want]
    cands = [torch.cuda.Stream(dev, **stream_kw) for _ in range(max(pool, want))]
    x = torch.zeros(64, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def overlaps(a, b):
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(a):
            ev[0].record(a)
            torch.cuda._sleep(400000)
            ev[1].record(a)
        with torch.cuda.stream(b):
            x.add_(1.0)
            ev[2].record(b)
        torch.cuda.synchronize(dev)
        return ev[0].elapsed_time(ev[2]) < 0.5 * ev[0].elapsed_time(ev[1])

    for c in cands:                     # (a queue exists from the first use of its stream)
        with torch.cuda.stream(c):
            x.add_(1.0)
    chosen = []
    for c in cands:
        if len(chosen) < want:
            if overlaps(main, c) and all(overlaps(o, c) for o in chosen):
                chosen.append(c)
    for c in cands:                     # fewer independent queues than asked for: take what there is
        if len(chosen) < want and c not in chosen:
            chosen.append(c)
    torch.cuda.synchronize(dev)
    have[:] = chosen
    return chosen


class DeviceBatch:
    """The placeholders of model.py:27-53 as int32 / fp32 device tensors + the C struct."""

    def __init__(self, batch, device, is_test=False, Ls=None):
        u, i, yj, hist_i, hist_i_new, hist_t, sl, new_sl, c = batch
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(np.asarray(a)), dtype=dt).to(device, non_blocking=True)
        self.B = int(len(u))
        if self.B < 1:
            raise ValueError("empty batch")
        hist_i = np.asarray(hist_i)
        hist_i_new = np.asarray(hist_i_new).reshape(self.B, -1)
        if Ls is not None and hist_i.shape != (self.B, Ls):
            raise ValueError("hist_i must be [B, Ls=%d], got %s" % (Ls, hist_i.shape))
        self.Sn = int(hist_i_new.shape[1])
        self.u = t(u, torch.int32)
        self.i = t(i, torch.int32)
        self.hist_i = t(hist_i, torch.int32)
        self.hist_i_new = t(hist_i_new, torch.int32) if self.Sn > 0 else torch.zeros(1, dtype=torch.int32, device=device)
        self.hist_t = t(hist_t, torch.float32)
        self.sl = t(sl, torch.int32)
        self.sl_new = t(new_sl, torch.int32)
        self.u_cate = t(c, torch.int32)
        self.j = t(yj, torch.int32) if is_test else None
        self.y = None if is_test else t(yj, torch.float32)
        self.c = self.struct()

    @classmethod
    def allocate(cls, B, Sn, Ls, device, is_test=False):
        """Uninitialised device arrays of a [B, Ls] / [B, Sn] batch (filled by tlsan_batch_pack)."""
        self = cls.__new__(cls)
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=device)
        self.B, self.Sn = int(B), int(Sn)
        self.u, self.i = e(B, torch.int32), e(B, torch.int32)
        self.hist_i = e((B, Ls), torch.int32)
        self.hist_i_new = e((B, Sn), torch.int32) if Sn > 0 else torch.zeros(1, dtype=torch.int32, device=device)
        self.hist_t = e((B, Ls), torch.float32)
        self.sl, self.sl_new, self.u_cate = e(B, torch.int32), e(B, torch.int32), e(B, torch.int32)
        self.j = e(B, torch.int32) if is_test else None
        self.y = None if is_test else e(B, torch.float32)
        self.c = self.struct()
        return self

    def to_host(self):
        """The reference's 9-tuple (input.py:54 / :107) as numpy arrays."""
        n = lambda t: t.cpu().numpy()
        third = n(self.j).astype(np.int64) if self.j is not None else n(self.y).astype(np.int64)
        hin = n(self.hist_i_new).astype(np.int64).reshape(self.B, self.Sn) if self.Sn > 0 else np.zeros((self.B, 0), np.int64)
        return (n(self.u).astype(np.int64), n(self.i).astype(np.int64), third, n(self.hist_i).astype(np.int64), hin,
                n(self.hist_t), n(self.sl).astype(np.int64), n(self.sl_new).astype(np.int64), n(self.u_cate).astype(np.int64))

    def __len__(self):
        return self.B

    def struct(self, use_j=True):
        p = lambda x: None if x is None else x.data_ptr()
        return L.Batch(self.B, self.Sn, p(self.u), p(self.i), p(self.j) if use_j else None, p(self.y),
                       p(self.hist_i), p(self.hist_i_new), p(self.hist_t), p(self.sl), p(self.sl_new),
                       p(self.u_cate))


class Model(object):
    def __init__(self, config, item_cate_list, device="cuda:0", seed=1234, norm_mode="tf18", l2_mode="dense",
                 table_dtype="f32", init="numpy", matrix_dtype="f32"):
        """table_dtype: "f32" (the reference's precision) or "bf16" -- item_emb / user_emb / cate_emb stored
        as bfloat16 (BASELINE.json configs[2]), arithmetic in fp32, updates written back with
        deterministic stochastic rounding; usert_emb, item_b and the dense weights stay fp32.
        init: "numpy" -- the variables' initial values drawn on the host (init_params; reproducible across
        devices); "device" -- the same distributions drawn in HBM (tables of 10^7 rows: BASELINE.json
        configs[4]; no host copy of the tables is ever made)."""
        if init not in ("numpy", "device"):
            raise ValueError("init must be 'numpy' or 'device'")
        # matrix_dtype: arithmetic of the fused kernel's matrix products -- "f32": exact fp32 MFMA (the reference's
        # precision); "bf16": operands rounded to bfloat16, fp32 products and sums (include/tlsan.h,
        # tlsan_params.matrix_dtype).  An extension for BASELINE.json configs[2]; any window the kernels take (in registers or streamed), with or without dropout.
        if matrix_dtype not in ("f32", "bf16"):
            raise ValueError("matrix_dtype must be 'f32' or 'bf16'")
        self.matrix_dtype = matrix_dtype
        if table_dtype not in ("f32", "bf16"):
            raise ValueError("table_dtype must be 'f32' or 'bf16'")
        self.table_dtype = table_dtype
        self.config = config
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise RuntimeError("tlsan_amd.Model needs a GPU (no CPU fallback)")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        if config.get("num_blocks", 1) != 1:
            raise ValueError("num_blocks != 1 is degenerate in the reference (model.py:330-364) and unsupported")
        self.dropout = float(config.get("dropout", 0.0))           # model.py:116-118, 428-431
        if not 0.0 <= self.dropout < 1.0:
            raise ValueError("dropout must be in [0, 1)")
        self._seed = int(seed)
        self.optimizer = config.get("optimizer", "sgd")           # model.py:188-195
        if self.optimizer not in OPTIMIZERS:
            raise ValueError("optimizer must be one of %s" % (sorted(OPTIMIZERS),))
        if self.optimizer != "sgd":
            # adam / rmsprop / adadelta see every row of the regularised tables every step (the L2 term
            # makes the reference's gradients dense), so the lazy-L2 form does not apply
            if l2_mode != "dense":
                raise NotImplementedError("optimizer=%r needs l2_mode='dense'" % self.optimizer)
        self.train_writer = _Writer(os.path.join(config.get("model_dir", "."), "train"))
        self.eval_writer = _Writer(os.path.join(config.get("model_dir", "."), "eval"))
        d = config["hidden_units"]
        self.dims = L.Dims(config["user_count"], config["item_count"], config["cate_count"], d,
                           config["itemid_embedding_size"], config["cateid_embedding_size"],
                           config["num_heads"], config["Ls"])
        if config["userid_embedding_size"] != config["itemid_embedding_size"]:
            raise ValueError("userid_embedding_size must equal itemid_embedding_size (both + cate = hidden_units)")
        self.lay = L.DenseLayout()
        L.check(self.lib.tlsan_dense_layout_of(C.byref(self.dims), C.byref(self.lay)), "tlsan_dense_layout_of")
        nbytes = self.lib.tlsan_state_bytes(C.byref(self.dims))
        if nbytes == 0:
            raise L.TlsanError("unsupported configuration: %s" % self.lib.tlsan_last_error().decode())
        self.norm_mode = {"tf18": L.NORM_TF18, "dedup": L.NORM_DEDUP}[norm_mode]
        # l2_mode: "dense" decays every row every step like the reference's dense L2 gradient
        # (model.py:164-172); "lazy" is the same update kept as W = P * W_stored (one global scale),
        # touching only rows that received a gradient -- identical up to fp32 rounding.
        self.l2_mode = {"dense": L.L2_DENSE, "lazy": L.L2_LAZY}[l2_mode]
        if self.l2_mode == L.L2_LAZY and self.norm_mode != L.NORM_TF18:
            raise NotImplementedError("l2_mode='lazy' supports norm_mode='tf18' only")
        icl = np.asarray(item_cate_list, np.int32)
        if icl.shape != (config["item_count"],):
            raise ValueError("item_cate_list must be [item_count]")
        self.item_cate = torch.as_tensor(icl).to(self.device)
        self._alloc_params()
        if init == "numpy":
            self.set_params(self.init_params(config, seed))
        else:
            self._init_on_device(config, seed)
        self._alloc_slots()
        self.state = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        if self.l2_mode == L.L2_LAZY:
            self.cparams.scale = self.lib.tlsan_state_scale(self.state.data_ptr())
        self._ws = None
        self._ws_key = (0, 0)
        self.renorm_every = 4096   # lazy L2: fold the table scale into the tables every so many steps (0 = never)
        # two destination-index slots: the current step's and the one being built for the next batch
        self._idx_slot = 0
        self._idx_ready = [None] * L.INDEX_SLOTS     # batch whose destination index sits (or is being built) in the slot
        self._idx_event = [torch.cuda.Event() for _ in range(L.INDEX_SLOTS)]
        self._pre_event = torch.cuda.Event()
        # host-visible word the first kernel of a step writes its sequence number into (train_async)
        self._started = torch.zeros(16, dtype=torch.int32).pin_memory()
        # (a step still queued when the Model is dropped writes this word when it starts: the pinned block must not be
        #  recycled under it -- the word outlives the Model, 64 bytes per Model ever built)
        _STARTED_WORDS.append(self._started)
        self._started_addr = self._started.data_ptr()
        self._started_word = C.c_uint32.from_address(self._started_addr)
        self._start_seq = 0
        self.poll_seconds = 0.0     # time train_async spent polling the `started` word (the host waiting for the GPU)
        self.started_at = None      # a list: train_async appends the host time at which it saw each step's first kernel start
        self._side = None
        self._step = 0
        self._epoch = 0
        self.global_step = _Var(lambda: self._step)
        self.global_epoch_step = _Var(lambda: self._epoch)
        self.global_epoch_step_op = _Var(self._inc_epoch)
        self._out = torch.zeros(4, dtype=torch.float32, device=self.device)  # loss, gnorm, sq_rows
        self._hits_p = np.zeros(len(KS), np.int64)
        self._hits_r = np.zeros(len(KS), np.int64)
        self._n_p = 0
        self._n_r = 0
        for idx, k in enumerate(KS):
            setattr(self, "prec_%d" % k, _Var(lambda idx=idx, k=k: self._hits_p[idx] / max(1, k * self._n_p)))
            setattr(self, "recall_%d" % k, _Var(lambda idx=idx: self._hits_r[idx] / max(1, self._n_r)))
        self._sync_state()

    # ------------------------------------------------------------------ parameters
    @staticmethod
    def init_params(config, seed=1234):
        """Variables of model.py:58-81, :443-450, :347 with the reference's initial values."""
        rng = np.random.RandomState(seed)
        I, U, Cc = config["item_count"], config["user_count"], config["cate_count"]
        di, dc = config["itemid_embedding_size"], config["cateid_embedding_size"]
        d, H, Ls = config["hidden_units"], config["num_heads"], config["Ls"]
        dh = d // H
        p = {
            "gamma": np.ones((), np.float32),
            "item_emb": glorot_uniform(rng, (I, di)),
            "item_b": np.zeros(I, np.float32),
            "user_emb": glorot_uniform(rng, (U, di)),
            "usert_emb": -np.ones((U, Ls), np.float32),
            "cate_emb": glorot_uniform(rng, (Cc, dc)),
            "dense_K": glorot_uniform(rng, (d, d)),
            "dense_b": np.zeros(d, np.float32),
        }
        for blk in ("fwa1", "fwa2"):
            p[blk + "_W1"] = glorot_uniform(rng, (dh, dh))
            p[blk + "_b1"] = np.zeros(dh, np.float32)
            p[blk + "_W2"] = glorot_uniform(rng, (dh, dh))
            p[blk + "_b2"] = np.zeros(dh, np.float32)
        return p

    def _init_on_device(self, config, seed):
        """init_params' distributions (glorot-uniform tables, usert_emb = -1, item_b = 0) drawn on the device;
        the small dense weights still come from the host stream."""
        g = torch.Generator(device=self.device)
        g.manual_seed(int(seed))
        for t in (self.item_emb, self.user_emb, self.cate_emb):
            limit = float(np.sqrt(6.0 / (t.shape[0] + t.shape[1])))
            if t.dtype == torch.float32:
                t.uniform_(-limit, limit, generator=g)
            else:
                t.copy_(torch.empty(t.shape, dtype=torch.float32, device=self.device).uniform_(-limit, limit, generator=g))
        self.item_b.zero_()
        self.usert_emb.fill_(-1.0)
        small = dict(config, item_count=1, user_count=1, cate_count=1)
        self.dense.copy_(torch.as_tensor(self.pack_dense(self.init_params(small, seed))))

    def _alloc_params(self):
        cfg, dev = self.config, self.device
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        tdt = torch.bfloat16 if self.table_dtype == "bf16" else torch.float32
        zt = lambda *s: torch.zeros(*s, dtype=tdt, device=dev)
        self.item_emb = zt(cfg["item_count"], cfg["itemid_embedding_size"])
        self.item_b = z(cfg["item_count"])
        self.user_emb = zt(cfg["user_count"], cfg["itemid_embedding_size"])
        self.usert_emb = z(cfg["user_count"], cfg["Ls"])
        self.cate_emb = zt(cfg["cate_count"], cfg["cateid_embedding_size"])
        self.dense = z(self.lay.n_dense)
        self.dense_KT = z(cfg["hidden_units"], cfg["hidden_units"])
        self.cparams = L.Params(self.item_emb.data_ptr(), self.item_b.data_ptr(), self.user_emb.data_ptr(),
                                self.usert_emb.data_ptr(), self.cate_emb.data_ptr(), self.dense.data_ptr(),
                                self.dense_KT.data_ptr(), self.item_cate.data_ptr(), 0, 0, 0, 0, None,
                                L.TABLE_BF16 if self.table_dtype == "bf16" else L.TABLE_F32,
                                L.MATRIX_BF16 if self.matrix_dtype == "bf16" else L.MATRIX_F32)

    def _alloc_slots(self):
        """Accumulators of adam / rmsprop / adadelta (tlsan_optimizer in include/tlsan.h): two sets of
        tables shaped like the parameters; RMSProp's first slot starts at one as in TF 1.8."""
        self.slots = None
        self._copt = None
        if self.optimizer == "sgd":
            return
        self.slots, self._cslots = [], []
        for which in range(2):
            fill = 1.0 if (self.optimizer == "rmsprop" and which == 0) else 0.0
            t = {k: torch.full_like(getattr(self, k), fill, dtype=torch.float32) for k in TABLE_KEYS}
            t["dense"] = torch.full_like(self.dense, fill)
            self.slots.append(t)
            self._cslots.append(L.Params(t["item_emb"].data_ptr(), t["item_b"].data_ptr(), t["user_emb"].data_ptr(),
                                         t["usert_emb"].data_ptr(), t["cate_emb"].data_ptr(), t["dense"].data_ptr(),
                                         None, None, 0, 0, 0, 0, None, L.TABLE_F32))
        kind, b1, b2, eps = OPTIMIZERS[self.optimizer]
        self._copt = L.Optimizer(kind, 0, b1, b2, eps, C.addressof(self._cslots[0]), C.addressof(self._cslots[1]))

    def _train_call(self, db, hp, out, ws):
        if self._copt is None:
            L.check(self.lib.tlsan_train_step(C.byref(self.dims), C.byref(self.cparams), C.byref(db.c), C.byref(hp),
                                              C.byref(out), self.state.data_ptr(), ws.data_ptr(), ws.numel(),
                                              self._stream()), "tlsan_train_step")
        else:
            self._copt.step = self._step + 1      # Adam's beta powers: updates applied so far + 1
            L.check(self.lib.tlsan_train_step_opt(C.byref(self.dims), C.byref(self.cparams), C.byref(db.c), C.byref(hp),
                                                  C.byref(self._copt), C.byref(out), self.state.data_ptr(),
                                                  ws.data_ptr(), ws.numel(), self._stream()), "tlsan_train_step_opt")

    def get_slots(self):
        """The optimizer's accumulators as two dicts of numpy arrays named like the parameters (None for sgd)."""
        if self.slots is None:
            return None
        out = []
        for t in self.slots:
            d = {k: t[k].cpu().numpy().copy() for k in TABLE_KEYS}
            d.update(self.unpack_dense(t["dense"].cpu().numpy()))
            out.append(d)
        return out

    def set_slots(self, slots):
        for t, src in zip(self.slots, slots):
            for k in TABLE_KEYS:
                t[k].copy_(torch.as_tensor(np.asarray(src[k], np.float32)))
            t["dense"].copy_(torch.as_tensor(self.pack_dense(src)))

    def _dense_slices(self):
        lay, d = self.lay, self.config["hidden_units"]
        dh = d // self.config["num_heads"]
        return {
            "fwa1_W1": (lay.f1_W1, (dh, dh)), "fwa1_b1": (lay.f1_b1, (dh,)),
            "fwa1_W2": (lay.f1_W2, (dh, dh)), "fwa1_b2": (lay.f1_b2, (dh,)),
            "dense_K": (lay.K, (d, d)), "dense_b": (lay.k0, (d,)),
            "fwa2_W1": (lay.f2_W1, (dh, dh)), "fwa2_b1": (lay.f2_b1, (dh,)),
            "fwa2_W2": (lay.f2_W2, (dh, dh)), "fwa2_b2": (lay.f2_b2, (dh,)),
            "gamma": (lay.gamma, ()),
        }

    def pack_dense(self, p):
        out = np.zeros(self.lay.n_dense, np.float32)
        for k, (off, shape) in self._dense_slices().items():
            n = int(np.prod(shape)) if shape else 1
            out[off:off + n] = np.asarray(p[k], np.float32).reshape(-1)
        return out

    def unpack_dense(self, flat):
        flat = np.asarray(flat)
        out = {}
        for k, (off, shape) in self._dense_slices().items():
            n = int(np.prod(shape)) if shape else 1
            out[k] = flat[off:off + n].reshape(shape).copy()
        return out

    def set_params(self, p):
        """Load a dict of numpy arrays (names as in oracle / checkpoint) into device memory."""
        for k in TABLE_KEYS:
            t = getattr(self, k)
            a = np.asarray(p[k], np.float32)
            if tuple(a.shape) != tuple(t.shape):
                raise ValueError("%s: shape %s != %s" % (k, a.shape, tuple(t.shape)))
            t.copy_(torch.as_tensor(a))      # (bf16 tables: round to nearest even on load)
        self.dense.copy_(torch.as_tensor(self.pack_dense(p)))
        if hasattr(self, "state"):
            self._sync_state()

    def fold_scale(self):
        """lazy L2: fold the table scale P into the stored tables (P = 1 afterwards)."""
        if self.l2_mode == L.L2_LAZY:
            L.check(self.lib.tlsan_state_renorm(C.byref(self.dims), C.byref(self.cparams), self.state.data_ptr(),
                                                self._stream()), "tlsan_state_renorm")

    def table_scale(self):
        return float(self.state[:4].view(torch.float32).item())

    def get_params(self):
        self.fold_scale()
        out = {k: getattr(self, k).detach().float().cpu().numpy().copy() for k in TABLE_KEYS}
        out.update(self.unpack_dense(self.dense.detach().cpu().numpy()))
        return out

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _sync_state(self):
        # tlsan_state_init clears the whole state, both destination-index slots included: an index that was
        # prefetched for an announced successor (train_async(next_batch=)) is gone with it -- wait for the side
        # stream to be done with the slot, then forget the announcement (the next step builds its index inline)
        for k in range(L.INDEX_SLOTS):
            if self._idx_ready[k] is not None:
                self._idx_event[k].synchronize()
        self._idx_ready = [None] * L.INDEX_SLOTS
        L.check(self.lib.tlsan_state_init(C.byref(self.dims), C.byref(self.cparams), self.state.data_ptr(),
                                          self._stream()), "tlsan_state_init")

    def _workspace(self, B, Sn):
        kB, kS = self._ws_key
        if self._ws is None or B > kB or Sn > kS:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the workspace cannot grow during capture (capture_step sizes it first)")
            kB, kS = max(B, kB), max(Sn, kS)
            n = self.lib.tlsan_workspace_bytes(C.byref(self.dims), kB, kS)
            if n == 0:
                raise L.TlsanError("tlsan_workspace_bytes: %s" % self.lib.tlsan_last_error().decode())
            torch.cuda.synchronize(self.device)
            self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._ws_key = (kB, kS)
        return self._ws

    def _inc_epoch(self):
        self._epoch += 1
        return self._epoch

    def dropout_seed(self, step=None):
        """Seed of the keep / drop pattern of update number `step` (default: the next one): a fixed
        function of the model's seed and the step, so that runs are reproducible."""
        step = self._step if step is None else step
        return ((self._seed * 0x9E3779B1) ^ ((step + 1) * 0x85EBCA77)) & 0xFFFFFFFF

    def hparams(self, lr, index_slot=0, index_prebuilt=0):
        return L.HParams(float(lr), float(self.config["regulation_rate"]), float(self.config["max_gradient_norm"]),
                         self.norm_mode, self.l2_mode, index_slot, index_prebuilt,
                         self.dropout, self.dropout_seed() if self.dropout > 0.0 else 0)

    # ------------------------------------------------------------------ training
    def device_batch(self, batch, is_test=False):
        return batch if isinstance(batch, DeviceBatch) else DeviceBatch(batch, self.device, is_test, self.config["Ls"])

    def train_async(self, batch, lr, logits=None, next_batch=None, after_next=None):
        """Enqueue one step (model.py:208-234) without reading the loss back.

        next_batch (optional, what an input pipeline knows anyway): its destination index (use
        counts, segment offsets -- a function of the ids only) is built on a second stream while
        this step computes, so the next step starts directly with the fused kernel.  The two
        streams are ordered by HOST waits on events that are complete by the time they are
        needed (device-side event waits between queues cost more than the work they would hide).

        after_next (optional): the batch after next_batch.  Its index is built TWO steps ahead.  With one batch ahead
        the index of batch t+1 can only run once k_fwd_bwd of step t has left the GPU (that kernel fills every CU), and
        step t+1 is launched when the host has seen it finish: kernel -> index (12 us) -> host wake-up -> launch is a
        path as long as the step's own kernels, so shortening either alone changes nothing.  Two ahead, the index a
        step waits for was finished during the previous step and the main stream never idles."""
        db = self.device_batch(batch)
        ws = self._workspace(db.B, db.Sn)
        out = L.StepOut(self._out.data_ptr(), self._out.data_ptr() + 4, None if logits is None else logits.data_ptr(), None)
        if torch.cuda.is_current_stream_capturing():   # hipGraph capture: one self-contained step
            if self.dropout > 0.0:
                raise NotImplementedError("the dropout seed is a launch argument: capture is not supported")
            if self.optimizer == "adam":
                raise NotImplementedError("Adam's step count is a launch argument: capture is not supported")
            hp = self.hparams(lr)
            self._train_call(db, hp, out, ws)
            self._step += 1
            return db
        if self.l2_mode == L.L2_LAZY and self.renorm_every and self._step and self._step % self.renorm_every == 0:
            # keep the table scale P = prod(1 - lr c reg) away from fp32 underflow in very long runs
            # (a fixed schedule, so runs stay bitwise reproducible); one sweep of the tables
            self.fold_scale()
        k = self._idx_slot
        if self._idx_ready[k] is not None and self._idx_ready[k] is not db:
            raise RuntimeError("train_async: the batch announced as next_batch must be the next one trained "
                               "(its destination index is already counted into the state)")
        pre = self._idx_ready[k] is db
        # streams are ordered by HOST waits on events that are complete by the time they are needed (a device-side
        # cross-queue wait costs ~8 us on this stack: measured, worse); under stream capture the host cannot wait
        dev_wait = torch.cuda.is_current_stream_capturing()
        if pre:
            if dev_wait:
                torch.cuda.current_stream(self.device).wait_event(self._idx_event[k])
            else:
                self._idx_event[k].synchronize()       # the side stream finished this batch's index
        self._idx_ready[k] = None
        main = torch.cuda.current_stream(self.device)
        NS = L.INDEX_SLOTS
        ahead = []      # (batch, slot) whose index is not there yet
        for j, nb in enumerate((next_batch, after_next)):
            if nb is None:
                continue
            kk = (k + 1 + j) % NS
            ndb = self.device_batch(nb)
            if self._idx_ready[kk] is ndb:
                continue
            if self._idx_ready[kk] is not None:
                raise RuntimeError("train_async: the batches announced ahead must be trained in that order "
                                   "(the destination index of another batch is already counted into the state)")
            ahead.append((ndb, kk))
        # everything queued so far -- the previous steps (last users of the free index slots) and whatever produced the
        # announced batches' arrays -- must be done before the side stream reads them.  An event recorded here would be
        # a barrier packet between the previous step's last kernel and this step's first one (measured: the fused
        # kernel starts 6.6 us after its predecessor ends instead of < 2); instead the step's first kernel stores a
        # sequence number into a pinned host word when it begins to run (tlsan_step_out.started), and the host polls it.
        # (under stream capture nothing runs, so the word would never change: events are the capturable path)
        flag_wait = bool(ahead) and not dev_wait
        if ahead and not flag_wait:
            self._pre_event.record(main)
        if flag_wait:
            self._start_seq = (self._start_seq + 1) & 0x7FFFFFFF
            out.started = self._started_addr
            out.started_value = self._start_seq
        hp = self.hparams(lr, k, 1 if pre else 0)
        self._train_call(db, hp, out, ws)
        if ahead:
            if self._side is None:
                # (high priority: the index kernels are short and the NEXT step cannot start without them; left at
                #  the default they trail behind the 2400 workgroups of the row-sum / update launches they share
                #  the GPU with -- measured: no difference either way)
                self._side = concurrent_streams(self.device, 1, priority=-1)[0]
            if dev_wait:
                self._side.wait_event(self._pre_event)
            else:
                word, want, t0, polls = self._started_word, self._start_seq, None, 0
                tp = time.perf_counter()
                while word.value != want:
                    polls += 1
                    if polls & 255:
                        continue
                    time.sleep(0)            # (every 256 polls: let another Python thread have the GIL)
                    if t0 is None:
                        t0 = time.perf_counter()
                    elif time.perf_counter() - t0 > 30.0:
                        raise RuntimeError("train_async: the step's first kernel did not start within 30 s")
                tq = time.perf_counter()
                self.poll_seconds += tq - tp     # (waiting for the GPU, not host work: bench.py subtracts it)
                if self.started_at is not None:  # (when the host SAW this step's first kernel start: bench.py's spread of step times)
                    self.started_at.append(tq)
            for ndb, kk in ahead:
                flag = L.INDEX_FOR_LAZY_SGD if (self.l2_mode == L.L2_LAZY and self.optimizer == "sgd") else 0
                L.check(self.lib.tlsan_batch_index(C.byref(self.dims), C.byref(ndb.c), self.cparams.item_cate, self.state.data_ptr(), kk | flag,
                                                   C.c_void_p(self._side.cuda_stream)), "tlsan_batch_index")
                self._idx_event[kk].record(self._side)
                self._idx_ready[kk] = ndb
        self._idx_slot = (k + 1) % L.INDEX_SLOTS
        self._step += 1
        return db

    def capture_step(self, batch, lr):
        """Capture one training step on `batch` into a hipGraph (fork/join of the index-build
        side stream included) and return the graph: `g.replay()` re-runs the step on the same
        device buffers.  lr is baked in (re-capture when it changes, train.py:232-233)."""
        db = self.device_batch(batch)
        # the graph bakes the workspace pointer in: size it for the worst case of this batch size once (the
        # longest session the kernels take, TLSAN_SN_CAP), so that a later, longer batch cannot make
        # _workspace() reallocate it under graphs captured earlier
        self._workspace(db.B, max(db.Sn, L.SN_CAP))
        self.train_async(db, lr)          # warm: lazy one-time initialisation happens outside capture
        self._step -= 1
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.train_async(db, lr)
        self._step -= 1
        g._tlsan_batch = db               # keep the captured buffers alive
        g._tlsan_ws = self._ws            # ... and the workspace the graph writes to
        return g

    def replay(self, g):
        if g._tlsan_ws is not self._ws:
            # a larger batch made the workspace grow after the capture: the graph's block stays alive (pinned
            # above), but it is no longer the model's -- recapture rather than run on a stale pointer
            raise RuntimeError("replay: the workspace was reallocated after this graph was captured; recapture the step")
        g.replay()
        self._step += 1

    def train(self, sess, batch, lr, add_summary=False):
        self.train_async(batch, lr)
        loss = float(self._out[0].item())
        if add_summary:
            self.train_writer.add_summary(self.train_summary(loss), global_step=self._step)
        return loss

    def train_summary(self, loss=None):
        """model.py:174-183's merged summary as (tag, value) pairs: the two scalars, and for every
        histogram its count / min / max / mean / std (`attention_output`, the histogram of u_t, needs
        a forward pass of the batch and is not kept)."""
        P = float(self.state[:4].view(torch.float32).item()) if self.l2_mode == L.L2_LAZY else 1.0
        out = []
        l2 = 0.0
        for tag, t in (("embedding/1_item_emb", self.item_emb), ("embedding/2_user_emb", self.user_emb),
                       ("embedding/3_cate_emb", self.cate_emb), ("embedding/4_usert_emb", self.usert_emb)):
            v = t.float() * P                              # true parameter = P * stored (lazy L2)
            l2 += 0.5 * float(v.double().pow(2).sum().item())     # tf.nn.l2_loss (model.py:164-169)
            for name, x in (("count", v.numel()), ("min", v.min().item()), ("max", v.max().item()),
                            ("mean", v.mean().item()), ("std", v.std().item())):
                out.append(("%s/%s" % (tag, name), float(x)))
        out.append(("gamma", float(self.dense[self.lay.gamma].item())))
        out.append(("L2_norm_user_item", l2))
        out.append(("Training Loss", float(self._out[0].item()) if loss is None else float(loss)))
        return out

    def last_gnorm(self):
        return float(self._out[1].item())

    def grads(self, batch, lr=1.0):
        """tf.gradients(loss, trainables) (model.py:198) as numpy arrays; test/diagnostic API."""
        db = self.device_batch(batch)
        ws = self._workspace(db.B, db.Sn)
        g = {k: torch.zeros_like(getattr(self, k), dtype=torch.float32) for k in TABLE_KEYS}
        gd = torch.zeros_like(self.dense)
        logits = torch.zeros(db.B, dtype=torch.float32, device=self.device)
        go = L.GradsOut(g["item_emb"].data_ptr(), g["item_b"].data_ptr(), g["user_emb"].data_ptr(),
                        g["usert_emb"].data_ptr(), g["cate_emb"].data_ptr(), gd.data_ptr())
        out = L.StepOut(self._out.data_ptr(), self._out.data_ptr() + 4, logits.data_ptr(), self._out.data_ptr() + 8)
        # (an index slot at rest: the other one may hold the index prefetched for an announced next batch)
        free = [k for k in range(L.INDEX_SLOTS) if self._idx_ready[k] is None]
        if not free:
            raise RuntimeError("grads: every index slot holds a prefetched index")
        slot = free[0]
        hp = self.hparams(lr, slot, 0)
        L.check(self.lib.tlsan_grads(C.byref(self.dims), C.byref(self.cparams), C.byref(db.c), C.byref(hp),
                                     C.byref(go), C.byref(out), self.state.data_ptr(), ws.data_ptr(), ws.numel(),
                                     self._stream()), "tlsan_grads")
        res = {k: v.cpu().numpy() for k, v in g.items()}
        res.update(self.unpack_dense(gd.cpu().numpy()))
        o = self._out.cpu().numpy()
        return dict(grads=res, loss=float(o[0]), gnorm=float(o[1]), logits=logits.cpu().numpy())

    # ------------------------------------------------------------------ evaluation
    def forward(self, batch, is_test=True, want_u_t=False, want_att=False):
        """logits for candidate i (and j when the batch has one) -- `sess.run(self.logits)`.
        want_att: also leave the reference's two attention-weight tensors (model.py:122, 386-394) on the model:
        self.att0 [H*B, Ls, d/H] and self.att1 [H*B, 1+Sn, d/H], row h*B + b (`sess.run([model.att0, model.att1])`)."""
        db = self.device_batch(batch, is_test)
        li = torch.empty(db.B, dtype=torch.float32, device=self.device)
        lj = torch.empty(db.B, dtype=torch.float32, device=self.device) if db.j is not None else None
        ut = torch.empty(db.B, self.config["hidden_units"], dtype=torch.float32, device=self.device) if want_u_t else None
        a0 = a1 = None
        if want_att:
            H, dh = self.config["num_heads"], self.config["hidden_units"] // self.config["num_heads"]
            a0 = torch.empty(H * db.B, self.config["Ls"], dh, dtype=torch.float32, device=self.device)
            a1 = torch.empty(H * db.B, 1 + db.Sn, dh, dtype=torch.float32, device=self.device)
        L.check(self.lib.tlsan_forward_att(C.byref(self.dims), C.byref(self.cparams), C.byref(db.c), li.data_ptr(),
                                           None if lj is None else lj.data_ptr(), None if ut is None else ut.data_ptr(),
                                           None if a0 is None else a0.data_ptr(), None if a1 is None else a1.data_ptr(),
                                           None, 0, self._stream()), "tlsan_forward")
        if want_att:
            self.att0, self.att1 = a0, a1
        return li, lj, ut, db

    def eval_auc(self, sess, batch):
        """mean(logit(pos) - logit(neg) > 0), ties wrong (model.py:237-263)."""
        li, lj, _, _ = self.forward(batch, is_test=True)
        return float(((li - lj) > 0).float().mean().item())

    def pairs_ranked_right(self, batch):
        """Per test row: logit(pos) - logit(neg) > 0 (bool tensor on the device) -- what eval_auc averages.  A row's
        value does not depend on which other rows share its batch, so the driver evaluates in large launches and forms
        the reference's per-batch means from slices (train.eval_auc)."""
        li, lj, _, _ = self.forward(batch, is_test=True)
        return (li - lj) > 0

    def label_ranks(self, batch):
        """rank of the positive item among all items for each test row (model.py:140-156)."""
        li, lj, ut, db = self.forward(batch, is_test=True, want_u_t=True)
        ws = self._workspace(db.B, db.Sn)
        ranks = torch.empty(db.B, dtype=torch.int32, device=self.device)
        L.check(self.lib.tlsan_eval_ranks(C.byref(self.dims), C.byref(self.cparams), ut.data_ptr(), db.i.data_ptr(),
                                          db.B, ranks.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()),
                "tlsan_eval_ranks")
        return ranks

    def _hits(self, batch, ranks=None):
        r = self.label_ranks(batch).cpu().numpy() if ranks is None else np.asarray(ranks)
        return np.array([(r < k).sum() for k in KS], np.int64), len(r)

    def eval_prec(self, sess, batch, ranks=None):
        """Streaming precision_at_k update ops (model.py:265-281); counters are cumulative over
        every call, like the reference's never-reset local variables (train.py:75-76,82).
        ranks (optional): the label ranks of this batch's rows, already computed (label_ranks on a larger launch)."""
        h, n = self._hits(batch, ranks)
        self._hits_p += h
        self._n_p += n
        return [self._hits_p[i] / (k * self._n_p) for i, k in enumerate(KS)]

    def eval_recall(self, sess, batch, ranks=None):
        h, n = self._hits(batch, ranks)
        self._hits_r += h
        self._n_r += n
        return [self._hits_r[i] / self._n_r for i in range(len(KS))]

    def reset_metrics(self):
        """Not in the reference (its counters are never reset); provided for per-round metrics."""
        self._hits_p[:] = 0
        self._hits_r[:] = 0
        self._n_p = self._n_r = 0

    # ------------------------------------------------------------------ checkpoint
    def save(self, sess=None):
        """model.py:302-307: parameters + config JSON next to them."""
        os.makedirs(self.config["model_dir"], exist_ok=True)
        base = os.path.join(self.config["model_dir"], "TLSAN")
        path = "%s-%d.npz" % (base, self._step)
        extra = {}
        if self.slots is not None:        # tf.train.Saver keeps the optimizer's slot variables too
            for n, sl in enumerate(self.get_slots()):
                extra.update({"slot%d/%s" % (n + 1, k): v for k, v in sl.items()})
        np.savez(path, global_step=self._step, global_epoch_step=self._epoch, **self.get_params(), **extra)
        json.dump(self.config, open("%s-%d.json" % (base, self._step), "w"), indent=2)
        if not self.config.get("quiet"):
            print("model saved at %s" % path, flush=True)
        return path

    def restore(self, sess, path):
        """model.py:310-313."""
        z = np.load(path)
        self.set_params({k: z[k] for k in TABLE_KEYS + DENSE_KEYS})
        self._step = int(z["global_step"])
        self._epoch = int(z["global_epoch_step"])
        if self.slots is not None and "slot1/item_emb" in z:
            self.set_slots([{k: z["slot%d/%s" % (n, k)] for k in TABLE_KEYS + DENSE_KEYS} for n in (1, 2)])
        if not self.config.get("quiet"):
            print("model restored from %s" % path, flush=True)
