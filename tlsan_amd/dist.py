"""Multi-GPU TLSAN step: one process per GPU, embedding tables row-sharded, RCCL over xGMI.

The reference is single-GPU (train.py:53,146); this is the MI355X-native scale-out of
SURVEY.md section 8e.  Samples are independent units: every rank trains its own batch
(data parallel, weak scaling).  State placement:

  * ``user_emb`` + ``usert_emb`` rows: sharded by ``user_id % world``  (fused rows
    ``[user_emb | usert_emb | pad]``);
  * ``item_emb`` + ``item_b`` rows: sharded by ``item_id % world`` (fused ``[item_emb | item_b | pad]``;
    mod-G spreads the Zipf-hot items);
  * ``cate_emb`` (<= 5 MB), ``item_cate_list`` and the dense attention weights: replicated.

One step on the main stream, in this order on every rank (four collectives, all on one
communicator in program order):

  plan, stage 1 (tlsan_route_plan; normally queued one step AHEAD for the next batch): mark the
      rows the batch touches in an owner-major key space, compact the marks with one HIP scan
      (distinct rows already in all-to-all order + the id -> compact-row map), and tell every
      owner which of its rows are wanted with ONE equal-split all-to-all that carries counts and
      row numbers together; the counts are copied to pinned host memory behind an event;
  plan, stage 2: the host reads the counts (the only host wait of a step; free when stage 1 ran a
      step ahead);
  (1) owners gather the wanted rows (tlsan_shard_gather), all-to-all of the rows into a compact
      per-step table the HIP kernels run on in place (row strides, tlsan_params.ld_*);
  (2) fused forward/backward + exact per-row gradient sums written straight into the fused
      [rows, W] layout (tlsan_grads with strides / sparse rows);
  (3) ONE all-reduce of [dense grads | cate grads | loss | norm terms];
  (4) tlsan_shard_summary: global norm, clip coefficient, loss, dense update (one launch);
  (5) all-to-all of the per-row gradients back to the owners;
  (6) tlsan_shard_apply: the owners file each received row under (row, source rank) -- rows of one
      source are distinct, so no counting sort -- and apply the update with dense L2 decay to every
      local row in fixed source order (bitwise reproducible), category table included.

Items and users share one fused shard table and one exchange each way.

``KeyRouter`` / ``RowExchange`` are device-agnostic torch + torch.distributed plumbing (run on
CPU/gloo in the tests); all arithmetic on rows is in libtlsan_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib as L
from .model import DENSE_KEYS, TABLE_KEYS, DeviceBatch, Model, _Var, _Writer, concurrent_streams


_STATE_HDR_BYTES = 256   # sizeof(StateHdr), csrc/tlsan_update.h: what tlsan_state_reindex keeps
_STATIC_SLOTS = 4     # routing plans of the static-shape step: the current batch and up to two announced successors


class ModPartition:
    """Row r of a table with n rows lives on rank r % world at local row r // world."""

    def __init__(self, n, world):
        self.n, self.world = int(n), int(world)

    def local_count(self, rank):
        return (self.n - rank + self.world - 1) // self.world

    def owner(self, ids):
        return ids % self.world

    def local_row(self, ids):
        return torch.div(ids, self.world, rounding_mode="floor")

    def global_ids(self, rank, device=None):
        return torch.arange(rank, self.n, self.world, device=device)


# the announced batches' plans are issued by a launch thread of the library (tlsan_shard_step_static, TLSAN_PLAN_ASYNC);
# TLSAN_PLAN_THREAD=0 keeps them on the calling thread.  The thread has only ever run on one-GPU boxes, so it is the default
# at ONE rank only (where the step is bound by its host thread: 93.6 -> 83.9 us); over several ranks it is opt-in
# (TLSAN_PLAN_THREAD=1) until a node has exercised it -- there the exchanges, not the host, bound the step.
_PLAN_THREAD_ENV = os.environ.get("TLSAN_PLAN_THREAD")
_PLAN_THREAD = _PLAN_THREAD_ENV != "0"


def _staged(group):
    """gloo has no device all-to-all: stage CUDA tensors through the host (used only by the
    single-GPU multi-process tests; RCCL moves device buffers directly)."""
    return dist.get_backend(group) == "gloo"


def a2a(out, inp, out_splits, in_splits, group=None):
    if _staged(group) and out.is_cuda:
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu().contiguous(), out_splits, in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=group)
    return out


def allreduce_sum(t, group=None):
    if _staged(group) and t.is_cuda:
        c = t.cpu()
        dist.all_reduce(c, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, group=group)
    return t


def allgather_rows(t, group=None):
    """[n, ...] per rank -> [world * n, ...] in rank order (equal n on every rank)."""
    world = dist.get_world_size(group)
    if world == 1:
        return t
    if _staged(group) and t.is_cuda:
        c = t.cpu().contiguous()
        o = torch.empty((world * c.shape[0],) + tuple(c.shape[1:]), dtype=c.dtype)
        dist.all_gather_into_tensor(o, c, group=group)
        return o.to(t.device)
    o = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(o, t.contiguous(), group=group)
    return o


class ExchangePlan:
    __slots__ = ("order", "send_counts", "recv_counts", "recv_rows", "n")


class RowExchange:
    """Fetch rows of a row-sharded table by global id, and route per-row values back."""

    def __init__(self, part, group=None):
        self.part = part
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        assert self.world == part.world

    def plan(self, uniq_ids):
        """uniq_ids: 1-D int64 tensor of distinct global row ids this rank needs."""
        p = ExchangePlan()
        p.n = int(uniq_ids.numel())
        owner = self.part.owner(uniq_ids)
        p.order = torch.argsort(owner, stable=True)
        ids_sorted = uniq_ids[p.order]
        sc = torch.bincount(owner, minlength=self.world)
        if self.world == 1:
            p.send_counts = [p.n]
            p.recv_counts = [p.n]
            p.recv_rows = self.part.local_row(ids_sorted)
            return p
        rc = torch.empty_like(sc)
        a2a(rc, sc, None, None, self.group)
        p.send_counts = [int(x) for x in sc.tolist()]
        p.recv_counts = [int(x) for x in rc.tolist()]
        recv_ids = torch.empty(sum(p.recv_counts), dtype=uniq_ids.dtype, device=uniq_ids.device)
        a2a(recv_ids, ids_sorted, p.recv_counts, p.send_counts, self.group)
        p.recv_rows = self.part.local_row(recv_ids)
        return p

    def fetch(self, plan, shard):
        """rows of `shard` (this rank's [n_local, width] slice) for every id of the plan, in the
        order of the `uniq_ids` given to plan()."""
        rows = shard[plan.recv_rows]
        if self.world == 1:
            got = rows
        else:
            got = torch.empty((plan.n, shard.shape[1]), dtype=shard.dtype, device=shard.device)
            a2a(got, rows, plan.send_counts, plan.recv_counts, self.group)
        out = torch.empty_like(got)
        out[plan.order] = got
        return out

    def push(self, plan, values):
        """Send one value row per id of the plan back to its owner.  Returns (local_rows, rows):
        contributions concatenated in source-rank order (deterministic)."""
        v = values[plan.order].contiguous()
        if self.world == 1:
            return plan.recv_rows, v
        got = torch.empty((sum(plan.recv_counts), values.shape[1]), dtype=values.dtype, device=values.device)
        a2a(got, v, plan.recv_counts, plan.send_counts, self.group)
        return plan.recv_rows, got


def _ru4(x):
    return (x + 3) // 4 * 4


class KeyRouter:
    """Owner-major key space for the fused shard table of one rank group.

    Every rank owns R = cI + cU rows: its items (id % G == rank) at local rows [0, cI) and its
    users at [cI, R).  A global id maps to key = owner * R + local_row, so
      * marking the keys a batch touches and compacting the marks (one scan) yields the distinct
        rows it needs ALREADY grouped by owner, i.e. in all-to-all send order, and the exclusive
        prefix is the id -> compact-row map of the per-step table;
      * what is sent to an owner are its local row numbers (key - owner * R).
    One plan / fetch / push serves both tables: one collective each way instead of two, and no
    sort / unique / bincount kernels."""

    def __init__(self, n_items, n_users, world, rank, group=None):
        self.G, self.rank, self.group = int(world), int(rank), group
        self.cI = (int(n_items) + self.G - 1) // self.G
        self.cU = (int(n_users) + self.G - 1) // self.G
        self.R = self.cI + self.cU
        self.nkeys = self.G * self.R

    def item_keys(self, ids):
        return (ids % self.G) * self.R + torch.div(ids, self.G, rounding_mode="floor")

    def user_keys(self, ids):
        return (ids % self.G) * self.R + self.cI + torch.div(ids, self.G, rounding_mode="floor")

    def plan(self, keys, scan):
        """keys: int64 tensor of every key the batch touches (duplicates fine).  `scan(flags)` ->
        (prefix, uniq, n_uniq_tensor) is the exclusive-scan + compaction primitive
        (tlsan_scan_compact on the GPU).  Returns a dict describing the exchange."""
        dev = keys.device
        flags = torch.zeros(self.nkeys, dtype=torch.int32, device=dev)
        flags[keys] = 1
        prefix, uniq, n_uniq = scan(flags)
        bnd = torch.arange(0, self.nkeys, self.R, device=dev)
        starts = torch.cat([prefix[bnd], n_uniq.reshape(1)])
        sc = (starts[1:] - starts[:-1]).to(torch.int64)
        if self.G > 1:
            rc = torch.empty_like(sc)
            a2a(rc, sc, None, None, self.group)
            both = torch.stack([sc, rc]).cpu()          # the step's single host sync
            send_counts, recv_counts = both[0].tolist(), both[1].tolist()
        else:
            send_counts = recv_counts = sc.cpu().tolist()
        n = int(sum(send_counts))
        uniq = uniq[:n]
        local_rows = uniq % self.R                      # int32 row numbers inside the owner's shard
        if self.G > 1:
            recv_rows = torch.empty(sum(recv_counts), dtype=torch.int32, device=dev)
            a2a(recv_rows, local_rows, recv_counts, send_counts, self.group)
        else:
            recv_rows = local_rows
        return dict(prefix=prefix, uniq=uniq, n=n, send_counts=send_counts, recv_counts=recv_counts,
                    recv_rows=recv_rows)

    def fetch(self, plan, shard):
        """compact per-step table: row k = the shard row of plan['uniq'][k]"""
        rows = shard[plan["recv_rows"].long()]
        if self.G == 1:
            return rows
        out = torch.empty((plan["n"], shard.shape[1]), dtype=shard.dtype, device=shard.device)
        a2a(out, rows, plan["send_counts"], plan["recv_counts"], self.group)
        return out

    def push(self, plan, values):
        """one value row per compact row back to its owner: (local_rows, rows) in source-rank order"""
        if self.G == 1:
            return plan["recv_rows"], values
        got = torch.empty((sum(plan["recv_counts"]), values.shape[1]), dtype=values.dtype, device=values.device)
        a2a(got, values, plan["recv_counts"], plan["send_counts"], self.group)
        return plan["recv_rows"], got


def torch_scan(flags):
    """scan + compaction with torch ops (CPU tests of the routing; the GPU path uses the HIP scan)"""
    inc = torch.cumsum(flags, 0, dtype=torch.int32)
    prefix = inc - flags
    uniq = torch.nonzero(flags, as_tuple=False).reshape(-1).to(torch.int32)
    pad = torch.zeros(flags.numel() - uniq.numel(), dtype=torch.int32, device=flags.device)
    return prefix, torch.cat([uniq, pad]), inc[-1:].clone()


class ShardedModel:
    """Model surface (train / eval_auc / ...) over row-sharded tables; see module docstring."""

    def __init__(self, config, item_cate_list, device="cuda:0", seed=1234, group=None, l2_mode="dense", static_rows=False,
                 wire_dtype="f32", init="numpy", deferred_ids=False, coalesce=False):
        """l2_mode: "dense" -- every owner decays every one of its rows every step, as the reference's dense L2
        gradient does; "lazy" (sgd) -- the same update kept as W = P * W_stored with one scale P that all ranks
        advance alike, so an owner touches only the rows whose gradients arrived (tlsan_shard_apply_lazy).
          wire_dtype (static_rows only): "f32", or "bf16" -- rows cross the wire with bf16 embedding values (fp32 weights
        stay with their owners; see below).
          static_rows (lazy only): False -- exchange sizes follow the batch (the host reads them once per step);
        True or an int -- every (source, owner) pair exchanges a FIXED number of row slots (the int, or 1.5 x what
        the first batch needs, agreed over the ranks), no size reaches the host, and a step can be recorded in a HIP
        graph (capture_step / replay).  A batch that needs more slots than that raises at the next host check.
          deferred_ids (static_rows): plans built ahead never carry their own id all-to-all -- the step that uses a plan
        issues it, on the main stream -- which is what runs by default over RCCL (no second communicator); True forces
        the same under the gloo exchange of the tests, whose default keeps a side group.
          coalesce (static_rows): the all-reduce of the dense gradients and the all-to-all of the row gradients, which do
        not depend on each other, are issued as ONE RCCL group (one launch, one latency) ahead of the summary -- the row
        gradients then travel BEFORE the clip coefficient exists (unscaled; the coefficient reaches the owners through the
        summary).  Under the gloo exchange of the tests the group is two staged calls back to back: the ORDER of the
        phases of the step, which is what differs from the default, is covered by the two-process tests
        (tests/test_gpu_shard_static.py); the RCCL group itself only where there are two GPUs (tests/test_gpu_configs.py).
        Off by default: no hardware with more than one GPU has ever run it."""
        if l2_mode not in ("dense", "lazy"):
            raise ValueError("l2_mode must be 'dense' or 'lazy'")
        self.lazy = l2_mode == "lazy"
        if static_rows and not self.lazy:
            raise NotImplementedError("static_rows is the lazy-L2 step's form (l2_mode='lazy')")
        if wire_dtype not in ("f32", "bf16"):
            raise ValueError("wire_dtype must be 'f32' or 'bf16'")
        if wire_dtype == "bf16" and not static_rows:
            raise NotImplementedError("wire_dtype='bf16' is built for the static-shape step (static_rows)")
        # wire_dtype="bf16": the owners keep (and update) fp32 rows; the copies that travel to the ranks using them, and
        # that the kernels gather from, carry the embedding values as bf16 (round to nearest even) -- 176 instead of 304
        # bytes per row at d = 128, Ls = 10.  The category table (replicated) is read through a bf16 shadow refreshed
        # every step.  Gradients travel back in fp32.
        self.wire_dtype = wire_dtype
        self.static_rows = static_rows
        self.deferred_ids = bool(deferred_ids)
        self.coalesce = bool(coalesce)
        self._st = None            # static-shape buffers (made at the first training batch)
        if not dist.is_initialized():
            raise RuntimeError("ShardedModel needs torch.distributed to be initialised (one process per GPU)")
        if self.coalesce and not static_rows:
            raise NotImplementedError("coalesce=True is the static-shape step's option")
        from .model import OPTIMIZERS
        if config.get("num_blocks", 1) != 1:
            raise NotImplementedError("num_blocks != 1 (see tlsan_amd.model.Model)")
        self.optimizer = config.get("optimizer", "sgd")
        if self.optimizer not in OPTIMIZERS:
            raise ValueError("optimizer must be one of %s" % (sorted(OPTIMIZERS),))
        self.dropout = float(config.get("dropout", 0.0))
        if not 0.0 <= self.dropout < 1.0:
            raise ValueError("dropout must be in [0, 1)")
        self._seed = int(seed)
        if self.lazy and self.optimizer != "sgd":
            raise NotImplementedError("l2_mode='lazy' is the SGD update's form; optimizer=%r sweeps every row" % self.optimizer)
        self.config = config
        self.lib = L.load()
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        U, I, Cc = config["user_count"], config["item_count"], config["cate_count"]
        self.U, self.I, self.C = U, I, Cc
        self.di, self.dc, self.Ls = config["itemid_embedding_size"], config["cateid_embedding_size"], config["Ls"]
        self.d, self.H = config["hidden_units"], config["num_heads"]
        # fused rows: items [item_emb | item_b | pad], users [user_emb | usert_emb | pad], one width
        self.W = max(self.di + 4, _ru4(self.di + self.Ls))
        self.router = KeyRouter(I, U, self.world, self.rank, group)
        self.cI, self.cU = self.router.cI, self.router.cU
        dev = self.device
        self.shard = torch.zeros(self.router.R, self.W, dtype=torch.float32, device=dev)
        self.cate_emb = torch.zeros(Cc, self.dc, dtype=torch.float32, device=dev)
        icl = np.asarray(item_cate_list, np.int32)
        # item -> category in key space (user / padding keys: -1 = in no category), so the compact
        # table's map is one gather
        ck = np.full(self.router.nkeys, -1, np.int32)
        ids = np.arange(I)
        ck[(ids % self.world) * self.router.R + ids // self.world] = icl
        self.cate_by_key = torch.as_tensor(ck).to(dev)
        dims_full = L.Dims(U, I, Cc, self.d, self.di, self.dc, self.H, self.Ls)
        self.dims_full = dims_full
        self.lay = L.DenseLayout()
        L.check(self.lib.tlsan_dense_layout_of(C.byref(dims_full), C.byref(self.lay)), "tlsan_dense_layout_of")
        self.dense = torch.zeros(self.lay.n_dense, dtype=torch.float32, device=dev)
        self.dense_KT = torch.zeros(self.d, self.d, dtype=torch.float32, device=dev)
        self.reg = float(config["regulation_rate"])
        self.clip = float(config["max_gradient_norm"])
        if self.W > 256 or self.world > 16:
            raise NotImplementedError("fused shard rows up to 256 floats, up to 16 ranks")
        self._sq = torch.zeros(2, dtype=torch.float64, device=dev)   # sums of squares: local shard rows, cate_emb
        # all-reduced vector: dense grads | cate grads | mean BCE | per-use squares | local table squares | pad
        self._flat = torch.zeros(self.lay.n_dense + Cc * self.dc + 4, dtype=torch.float32, device=dev)
        self._gn_local = torch.zeros(1, dtype=torch.float32, device=dev)
        self._step_dev = torch.zeros(4, dtype=torch.float32, device=dev)     # lr * coef, coef, (lazy:) lr * coef / P_new, P_new
        self.renorm_every = 4096
        self._P = torch.ones(1, dtype=torch.float32, device=dev)            # lazy L2: tables = P * stored
        if self.lazy:
            self._slots64 = torch.zeros(self.router.R * self.world, dtype=torch.int64, device=dev)
            self._lws = None
            self._sopt_lazy = L.ShardOptimizer(L.OPT_SGD, 0, 0.0, 0.0, 0.0, None, None, None, None, None, None,
                                               self._P.data_ptr())
        # adam / rmsprop / adadelta: accumulators laid out like what they belong to (RMSProp's first starts at one)
        self._sopt = None
        if self.optimizer != "sgd":
            kind, b1, b2, eps = OPTIMIZERS[self.optimizer]
            one = 1.0 if self.optimizer == "rmsprop" else 0.0
            self.slots = dict(shard_s1=torch.full_like(self.shard, one), shard_s2=torch.zeros_like(self.shard),
                              cate_s1=torch.full_like(self.cate_emb, one), cate_s2=torch.zeros_like(self.cate_emb),
                              dense_s1=torch.full_like(self.dense, one), dense_s2=torch.zeros_like(self.dense))
            self._sopt = L.ShardOptimizer(kind, 0, b1, b2, eps, *[self.slots[k].data_ptr() for k in
                                          ("shard_s1", "shard_s2", "cate_s1", "cate_s2", "dense_s1", "dense_s2")])
        self.last_loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.last_gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._ws = None
        self._flags = torch.zeros(self.router.nkeys, dtype=torch.int32, device=dev)   # zero at rest
        self._rows_pad = 4096     # compact-table rows are padded (grow-only): the kernels' state layout is stable
        self._pcap = 65536        # shared per-owner capacity of the id exchange (see _slot)
        self._side = None
        self._sizes = {}      # (compact rows, categories, B, Sn) -> (state bytes, workspace bytes)
        self._ews = None
        self._hits_p = np.zeros(6, np.int64)
        self._hits_r = np.zeros(6, np.int64)
        self._n_p = self._n_r = 0
        loc = np.arange(self.rank, I, self.world)
        self._icl_local = torch.as_tensor(np.concatenate([icl[loc], np.zeros(1, np.int32)])).to(dev)   # category of local item n
        self._slots = [None, None, None]    # routing plans: current / prefetched / forward-only (evaluation)
        self._next_slot = 0
        self._slots_buf = torch.zeros(self.router.R * self.world, dtype=torch.int32, device=dev)  # zero at rest
        self._aws = torch.empty(int(self.lib.tlsan_shard_apply_workspace(self.router.R, Cc)), dtype=torch.uint8, device=dev)
        self._step = 0
        self._epoch = 0
        self.global_step = _Var(lambda: self._step)
        self.global_epoch_step = _Var(lambda: self._epoch)
        self.train_writer = _Writer(os.path.join(config.get("model_dir", "."), "train"))
        self.eval_writer = _Writer(os.path.join(config.get("model_dir", "."), "eval"))
        if init == "device":
            # the same distributions drawn on the device, shard by shard (tables of 10^7 rows: the host draw of the whole
            # model on every rank takes minutes and tens of GB); NOT the values of Model(init="numpy") on one GPU
            self._init_on_device(config, seed)
        elif init == "numpy":
            self.set_params(Model.init_params(config, seed))   # identical on every rank (numpy, seeded)
        else:
            raise ValueError("init must be 'numpy' or 'device'")

    def _init_on_device(self, config, seed):
        di, Ls, cI = self.di, self.Ls, self.cI
        g = torch.Generator(device=self.device)
        g.manual_seed(int(seed) * 1000003 + self.rank)
        n_i = len(range(self.rank, self.I, self.world))
        n_u = len(range(self.rank, self.U, self.world))
        self.shard.zero_()
        li, lu = float(np.sqrt(6.0 / (self.I + di))), float(np.sqrt(6.0 / (self.U + di)))
        self.shard[:n_i, :di].uniform_(-li, li, generator=g)
        self.shard[cI:cI + n_u, :di].uniform_(-lu, lu, generator=g)
        self.shard[cI:cI + n_u, di:di + Ls] = -1.0
        small = dict(config, item_count=1, user_count=1)
        p = Model.init_params(small, seed)          # cate_emb and the dense weights: the host stream, the same on every rank
        self.cate_emb.copy_(torch.as_tensor(np.asarray(p["cate_emb"], np.float32)))
        self._P.fill_(1.0)
        self._pack_dense(p)
        self._refresh_squares()
        self._reset_step_state()

    # ------------------------------------------------------------------ helpers
    def _pack_dense(self, p):
        d, lay = self.d, self.lay
        self.dense.copy_(torch.as_tensor(self._dense_flat(p)))
        self.dense_KT.copy_(self.dense[lay.K:lay.K + d * d].view(d, d).t())

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def device_batch(self, batch, is_test=False):
        """Upload the batch and map its ids into the router's key space (elementwise, once)."""
        db = batch if isinstance(batch, DeviceBatch) else DeviceBatch(batch, self.device, is_test, self.Ls)
        if not hasattr(db, "keys"):
            r = self.router
            # Padded slots of the windows (zeros past sl / sl_new, input.py:41-51) must not put item 0 into the plan:
            # a row that only padding refers to receives no gradient, and the per-row gradient buffer of the step is
            # not zero-filled (every compact row is expected to be written by a use).  They take the key of the
            # sample's own candidate instead -- always a real use; the kernels weigh padded slots with exactly 0.
            cand = db.i.long()
            ar = torch.arange(self.Ls, device=db.i.device)[None, :]
            hist = torch.where(ar < db.sl.long()[:, None], db.hist_i.long(), cand[:, None])
            parts = [r.item_keys(cand), r.item_keys(hist.reshape(-1))]
            if db.Sn > 0:
                ar = torch.arange(db.Sn, device=db.i.device)[None, :]
                new = torch.where(ar < db.sl_new.long()[:, None], db.hist_i_new.reshape(db.B, db.Sn).long(), cand[:, None])
                parts.append(r.item_keys(new.reshape(-1)))
            if db.j is not None:
                parts.append(r.item_keys(db.j.long()))
            parts.append(r.user_keys(db.u.long()))
            db.keys = torch.cat(parts).to(torch.int32)
        return db

    # ------------------------------------------------------------------ routing plan (two stages)
    def _slot(self, k, n_keys):
        """Per-slot routing buffers (two slots: the plan of the next batch is built while the
        current one is in use)."""
        sl = self._slots[k]
        # rows one owner can be asked for by one rank: the id exchange is equal-split, so this capacity
        # must be the SAME on every rank -- it is a value all ranks share (self._pcap: a constant at first,
        # raised in lockstep by the overflow protocol of _plan_stage2), never this rank's batch size
        pcap = min(self.router.R, self._pcap)
        if sl is None or sl["cap"] < n_keys or sl["pcap"] != pcap:
            dev, G, nk = self.device, self.world, self.router.nkeys
            cap = max(int(n_keys * 1.25) + 16, sl["cap"] if sl is not None else 0)
            sl = dict(k=k, cap=cap, pcap=pcap, rank=torch.empty(nk, dtype=torch.int32, device=dev),
                      uniq=torch.empty(nk, dtype=torch.int32, device=dev),
                      n_uniq=torch.zeros(1, dtype=torch.int32, device=dev),
                      # per peer {count, local row numbers ...}: ONE equal-split all-to-all carries both
                      sendbuf=torch.zeros(G, 1 + pcap, dtype=torch.int32, device=dev),
                      recvbuf=torch.zeros(G, 1 + pcap, dtype=torch.int32, device=dev),
                      cate_c=torch.empty(max(cap, self._rows_pad), dtype=torch.int32, device=dev),
                      comp=torch.empty(cap, dtype=torch.int32, device=dev),
                      host=torch.zeros(2, G, dtype=torch.int32).pin_memory(),
                      event=torch.cuda.Event(), state=None)
            self._slots[k] = sl
        return sl

    def _plan_stage1(self, db, k):
        """Device-only half: distinct rows grouped by owner, compact ids, and ONE all-to-all that
        tells every owner which of its rows are wanted (counts + row numbers); the copy of the
        counts to the host is queued, nothing waits."""
        nk = int(db.keys.numel())
        sl = self._slot(k, nk)
        if sl["cate_c"].numel() < self._rows_pad:    # the padded table grew since the slot was made
            sl["cate_c"] = torch.empty(self._rows_pad, dtype=torch.int32, device=self.device)
        sl["cate_pad"] = self._rows_pad               # category map entries [n_uniq, cate_pad) are set to -1
        r = self.router
        L.check(self.lib.tlsan_route_plan(db.keys.data_ptr(), nk, r.R, r.G, self.cate_by_key.data_ptr(),
                                          self._flags.data_ptr(), sl["rank"].data_ptr(), sl["uniq"].data_ptr(),
                                          sl["n_uniq"].data_ptr(), sl["sendbuf"].data_ptr(), sl["pcap"],
                                          sl["cate_c"].data_ptr(), sl["cate_pad"], sl["comp"].data_ptr(),
                                          sl["host"].data_ptr(), self._stream()),     # send counts -> host[0]
                "tlsan_route_plan")
        if self.world > 1:
            a2a(sl["recvbuf"].view(-1), sl["sendbuf"].view(-1), None, None, self.group)
            rb = sl["recvbuf"]
        else:
            rb = sl["sendbuf"]
        sl["rb"] = rb
        if self.world > 1:      # what the peers ask of this rank (at one rank: the same numbers, see stage 2)
            sl["host"][1].copy_(rb[:, 0], non_blocking=True)
        sl["event"].record(torch.cuda.current_stream(self.device))
        sl["db"] = db
        sl["prepared"] = False
        return sl

    def _plan_stage2(self, sl):
        """Host half: read the exchange sizes (the step's only host wait; free when stage 1 was
        queued a step ahead)."""
        if sl.get("stage2") is sl["db"]:
            return sl
        sl["event"].synchronize()
        h = sl["host"]
        send = h[0].tolist()
        recv = h[1].tolist() if self.world > 1 else send
        over = [-x for x in send + recv if x < 0]
        if over:
            # some rank's batch may ask one owner for more rows than the shared capacity holds: it sent
            # -need in every header and no rows.  Every rank sees the same negative headers, so all of
            # them raise the capacity to the same value and repeat the plan (rare: the capacity only grows)
            self._pcap = int(max(over) * 1.25) + 16
            sl["stage2"] = None
            return self._plan_stage2(self._plan_stage1(sl["db"], sl["k"]))
        n, n_recv = int(sum(send)), int(sum(recv))
        recv_rows = torch.empty(max(n_recv, 1), dtype=torch.int32, device=self.device)
        off = [0]
        for c in recv:
            off.append(off[-1] + int(c))
        sl.update(send=send, recv=recv, n=n, n_recv=n_recv, recv_rows=recv_rows,
                  src_off=(C.c_int32 * (self.world + 1))(*off), stage2=sl["db"])
        return sl

    def _plan(self, db):
        """The plan of a TRAINING batch: prefetched (train_async(..., next_batch=)) or built now."""
        k = self._next_slot
        sl = self._slots[k]
        if sl is None or sl.get("db") is not db:
            if sl is not None and sl.get("prepared"):
                raise RuntimeError("train_async: the batch announced as next_batch must be the next one trained "
                                   "(its destination index is already counted into the slot's state)")
            sl = self._plan_stage1(db, k)
        self._next_slot = 1 - k
        return self._plan_stage2(sl)

    def _plan_eval(self, db):
        """The plan of a forward-only batch, in a slot of its own: evaluation may run between a training
        step and the successor it has announced (whose plan and indices are waiting in the other slots)."""
        if self._st is not None:   # static-shape training: plans announced ahead may be running on the side streams
            main = torch.cuda.current_stream(self.device)     # (they use the same mark scratch as this plan)
            self._flush_plans()
            main.wait_stream(self._st["side"])
            main.wait_stream(self._st["side2"])
        return self._plan_stage2(self._plan_stage1(db, 2))

    def _fetch(self, sl):
        """compact per-step table: row k = the owner's shard row of the k-th distinct key"""
        if self._rows_pad < sl["n"]:        # compact-table rows, padded (grow-only) so the state layout is stable
            self._rows_pad = (sl["n"] + sl["n"] // 16 + 4095) // 4096 * 4096
        if sl["cate_pad"] < self._rows_pad:  # (rare) the table grew after this plan's category map was written
            cc = torch.full((self._rows_pad,), -1, dtype=torch.int32, device=self.device)
            cc[:sl["n"]] = sl["cate_c"][:sl["n"]]
            sl["cate_c"], sl["cate_pad"] = cc, self._rows_pad
        rows = torch.empty((max(sl["n_recv"], self._rows_pad if self.world == 1 else 1), self.W), dtype=torch.float32,
                           device=self.device)
        L.check(self.lib.tlsan_shard_gather(self.shard.data_ptr(), self.W, self.router.R, self.W, sl["rb"].data_ptr(),
                                            sl["pcap"], self.world, sl["n_recv"], rows.data_ptr(),
                                            sl["recv_rows"].data_ptr(), self._stream()), "tlsan_shard_gather")
        if self.world == 1:
            return rows
        table = torch.empty((self._rows_pad, self.W), dtype=torch.float32, device=self.device)
        a2a(table[:sl["n"]], rows[:sl["n_recv"]], sl["send"], sl["recv"], self.group)
        return table

    def _compact(self, db, sl, table):
        """ctypes views of the step on the compact table (cached per slot: building the three structs
        costs the host more than a kernel launch, and their pointers only change when a buffer does)"""
        key = (id(db), sl["comp"].data_ptr(), sl["cate_c"].data_ptr(), None if table is None else table.data_ptr(),
               self._rows_pad)
        cache = sl.setdefault("_compact", {})
        hit = cache.get(key)
        if hit is not None and hit[0] is db:
            return hit[1]
        if len(cache) >= 8:
            cache.clear()
        out = self._compact_build(db, sl, table)
        cache[key] = (db, out)
        return out

    def _compact_build(self, db, sl, table):
        B, Ls, Sn = db.B, self.Ls, db.Sn
        comp = sl["comp"]
        esz = 4
        base_c = comp.data_ptr()
        o = 0
        p_i = base_c + esz * o; o += B
        p_hist = base_c + esz * o; o += B * Ls
        p_new = base_c + esz * o if Sn > 0 else db.hist_i_new.data_ptr(); o += B * Sn
        p_j = None
        if db.j is not None:
            p_j = base_c + esz * o; o += B
        p_u = base_c + esz * o
        ptr = lambda t: None if t is None else t.data_ptr()
        cb = L.Batch(B, Sn, p_u, p_i, p_j, ptr(db.y), p_hist, p_new, ptr(db.hist_t),
                     ptr(db.sl), ptr(db.sl_new), ptr(db.u_cate))
        base = (self.shard if table is None else table).data_ptr()   # (None: only item_cate is looked at)
        cp = L.Params(base, base + 4 * self.di, base, base + 4 * self.di, self.cate_emb.data_ptr(),
                      self.dense.data_ptr(), self.dense_KT.data_ptr(), sl["cate_c"].data_ptr(),
                      self.W, self.W, self.W, self.W, self._P.data_ptr() if self.lazy else None)
        # rows [n, pad) of the padded table are never referenced (no use, category -1)
        n = self._rows_pad
        dims = L.Dims(n, n, self.C, self.d, self.di, self.dc, self.H, self.Ls)
        return dims, cp, cb

    def _prepare_side(self, ndb, nsl):
        """While the current step runs: on a second stream, rebuild the category index of the next
        step's compact table and build its destination index (both functions of the routing plan
        only).  Ordered by host waits on events that are already complete; no collective here."""
        st8 = nsl.get("state")
        if st8 is None or nsl["n"] > self._rows_pad or nsl.get("dims_key") != (self._rows_pad, self.C) or self._ws is None:
            return   # first use of the slot or the padded shape grows: the step does it inline
        dims, cp, cb = self._compact(ndb, nsl, None)
        if self._side is None:
            self._side = concurrent_streams(self.device, 1)[0]
            self._side_event = [torch.cuda.Event(), torch.cuda.Event()]
        sst = C.c_void_p(self._side.cuda_stream)
        L.check(self.lib.tlsan_state_recategorize(C.byref(dims), C.byref(cp), st8.data_ptr(), sst), "tlsan_state_recategorize")
        L.check(self.lib.tlsan_batch_index(C.byref(dims), C.byref(cb), cp.item_cate, st8.data_ptr(), 0, sst), "tlsan_batch_index")
        ev = self._side_event[self._next_slot]
        ev.record(self._side)
        nsl["prep_event"] = ev
        nsl["prepared"] = True

    def _buffers(self, sl, dims, cp, B, Sn, prepared=False):
        skey = (dims.item_count, dims.cate_count, B, Sn)
        if skey not in self._sizes:
            nst = self.lib.tlsan_state_bytes(C.byref(dims))
            nws = self.lib.tlsan_workspace_bytes(C.byref(dims), B, Sn)
            if nst == 0 or nws == 0:
                raise L.TlsanError(self.lib.tlsan_last_error().decode())
            self._sizes[skey] = (nst, nws)
        nst, nws = self._sizes[skey]
        fresh = sl["state"] is None or sl["state"].numel() < nst
        if fresh:
            sl["state"] = torch.zeros(int(nst * 1.5), dtype=torch.uint8, device=self.device)
            sl["state"][:4].view(torch.float32).fill_(1.0)   # table scale P = 1 (the owners apply the decay)
        if self._ws is None or self._ws.numel() < nws:
            self._ws = torch.empty(int(nws * 1.25), dtype=torch.uint8, device=self.device)
        # the compact table's item -> category map changes every step: rebuild the category -> items
        # index for it; the use counters are zero between steps unless the (padded) shape changed
        key = (dims.item_count, dims.cate_count)
        if prepared and not fresh and sl.get("dims_key") == key:
            return
        if fresh or sl.get("dims_key") != key:
            sl["dims_key"] = key
            L.check(self.lib.tlsan_state_reindex(C.byref(dims), C.byref(cp), sl["state"].data_ptr(), self._stream()),
                    "tlsan_state_reindex")
        else:
            L.check(self.lib.tlsan_state_recategorize(C.byref(dims), C.byref(cp), sl["state"].data_ptr(), self._stream()),
                    "tlsan_state_recategorize")

    # ------------------------------------------------------------------ training
    def dropout_seed(self, step=None):
        """As tlsan_amd.model.Model.dropout_seed: the same pattern on every rank (ranks differ by sample offset)."""
        step = self._step if step is None else step
        return ((self._seed * 0x9E3779B1) ^ ((step + 1) * 0x85EBCA77)) & 0xFFFFFFFF

    def train_async(self, batch, lr, next_batch=None, weight=1.0, sample0=0, after_next=None):
        """One step.  sample0: position of this rank's first sample in the global batch (dropout pattern).
          `weight`: this rank's share of the global mean when the ranks' batches differ in
        size, B_rank * world / B_global (1 when they are equal; 0 for a rank that only holds a
        placeholder row of a global batch smaller than the world).
          `next_batch` (optional): its routing plan is queued before this step's heavy
        kernels, so that the next step's host wait for the exchange sizes costs nothing.
          `after_next` (optional, static_rows): the batch after that; its plan is built two steps ahead, beside the
        second half of this step and all of the next, and is off the critical path altogether."""
        db = self.device_batch(batch)
        G = self.world
        if self.static_rows:
            return self._train_static(db, lr, next_batch, weight, sample0, after_next)
        sl = self._plan(db)
        ndb = None
        if next_batch is not None:
            ndb = self.device_batch(next_batch)
            self._plan_stage1(ndb, self._next_slot)
        prepared = bool(sl.get("prepared")) and sl["n"] <= self._rows_pad
        sl["prepared"] = False
        if prepared:
            sl["prep_event"].synchronize()      # the side stream finished this batch's indices
        table = self._fetch(sl)
        dims, cp, cb = self._compact(db, sl, table)
        self._buffers(sl, dims, cp, db.B, db.Sn, prepared)
        n, Cc, di, Ls, W = dims.item_count, self.C, self.di, self.Ls, self.W
        dev = self.device
        n_dense, n_cate = self.lay.n_dense, Cc * self.dc
        flat = self._flat
        fp = flat.data_ptr()
        # per-row gradients land directly in the fused layout (a compact row is an item or a user)
        # (every compact row below sl["n"] is used by this batch and written in full: no zero fill; the rows
        #  past it are padding that is never sent on)
        gf = torch.empty(n, W, dtype=torch.float32, device=dev)
        g0 = gf.data_ptr()
        go = L.GradsOut(g0, g0 + 4 * di, g0, g0 + 4 * di, fp + 4 * n_dense, fp, W, W, W, W, 2)
        tail = fp + 4 * (n_dense + n_cate)
        out = L.StepOut(tail, self._gn_local.data_ptr(), None, tail + 4)
        hp = L.HParams(float(lr), 0.0, self.clip, L.NORM_TF18, L.L2_DENSE, 0, 1 if prepared else 0,   # reg: applied by the owners
                       self.dropout, self.dropout_seed() if self.dropout > 0.0 else 0, int(sample0))
        st = self._stream()
        L.check(self.lib.tlsan_grads(C.byref(dims), C.byref(cp), C.byref(cb), C.byref(hp), C.byref(go), C.byref(out),
                                     sl["state"].data_ptr(), self._ws.data_ptr(), self._ws.numel(), st), "tlsan_grads")
        if weight != 1.0:      # uneven split of the global batch: the local means enter with their share
            k = n_dense + n_cate
            flat[:k + 1].mul_(float(weight))
            flat[k + 1:k + 2].mul_(float(weight) ** 2)
            gf.mul_(float(weight))
        # ---- one all-reduce: dense grads | cate grads | loss | per-use squares | local table squares
        if G > 1:
            allreduce_sum(flat, self.group)
        if self._sopt is not None:
            self._sopt.step = self._step + 1
        sopt = None if self._sopt is None else C.byref(self._sopt)
        if self.lazy:
            sopt = C.byref(self._sopt_lazy)
        L.check(self.lib.tlsan_shard_summary_opt(fp, n_dense, n_cate, G, float(lr), self.reg, self.clip,
                                                 self._sq.data_ptr() + 8, self.dense.data_ptr(), self.dense_KT.data_ptr(),
                                                 C.byref(self.dims_full), self._step_dev.data_ptr(),
                                                 self.last_loss.data_ptr(), self.last_gnorm.data_ptr(), sopt, st),
                "tlsan_shard_summary")
        # ---- row gradients back to the owners, deterministic apply with dense L2 decay of every row
        if G > 1:
            vals = torch.empty((max(sl["n_recv"], 1), W), dtype=torch.float32, device=dev)
            a2a(vals[:sl["n_recv"]], gf[:sl["n"]], sl["recv"], sl["send"], self.group)
        else:
            vals = gf
        if self.lazy:
            nws = int(self.lib.tlsan_shard_apply_lazy_workspace(sl["n_recv"], Cc))
            if self._lws is None or self._lws.numel() < nws:
                self._lws = torch.empty(int(nws * 1.5) + 256, dtype=torch.uint8, device=dev)
            L.check(self.lib.tlsan_shard_apply_lazy(self.shard.data_ptr(), W, self.cI, self.router.R, W, di, di + Ls,
                                                    vals.data_ptr(), W, sl["recv_rows"].data_ptr(), sl["n_recv"],
                                                    sl["src_off"], G, self._slots64.data_ptr(), self._lazy_stamp(),
                                                    1.0 / G, self._step_dev.data_ptr(), self.cate_emb.data_ptr(), Cc, self.dc,
                                                    fp + 4 * n_dense, self._sq.data_ptr(), tail + 8, self._P.data_ptr(),
                                                    self._lws.data_ptr(), self._lws.numel(), st), "tlsan_shard_apply_lazy")
            self._step += 1
            self._keep = (table, gf, vals, sl["recv_rows"])
            if self.renorm_every and self._step % self.renorm_every == 0:
                self.fold_scale()
            if ndb is not None:
                nsl = self._plan_stage2(self._slots[self._next_slot])
                self._prepare_side(ndb, nsl)
            return db
        L.check(self.lib.tlsan_shard_apply_opt(self.shard.data_ptr(), W, self.cI, self.router.R, W, di, di + Ls,
                                               vals.data_ptr(), W, sl["recv_rows"].data_ptr(), sl["n_recv"], sl["src_off"],
                                               G, self._slots_buf.data_ptr(), 1.0 / G, self._step_dev.data_ptr(), self.reg,
                                               self.cate_emb.data_ptr(), Cc, self.dc, fp + 4 * n_dense,
                                               self._sq.data_ptr(), tail + 8, sopt, float(lr),
                                               self._aws.data_ptr(), self._aws.numel(), st),
                "tlsan_shard_apply")
        self._step += 1
        self._keep = (table, gf, vals, sl["recv_rows"])
        if ndb is not None:
            nsl = self._plan_stage2(self._slots[self._next_slot])   # its counts arrived long ago
            self._prepare_side(ndb, nsl)
        return db

    # ------------------------------------------------------------------ static-shape step (static_rows)
    def _static_setup(self, db):
        """Buffers of the static-shape step, sized once: `cap` row slots per (source, owner) pair."""
        r, G, dev, W = self.router, self.world, self.device, self.W
        if self.static_rows is True:
            # what this batch needs of its busiest owner (one host read, at set-up only), with headroom; the
            # all-to-alls are equal-split, so every rank must use the same number
            flags = torch.zeros(r.nkeys, dtype=torch.int32, device=dev)
            flags[db.keys.long()] = 1
            need = int(flags.view(G, r.R).sum(1).max().item())
            t = torch.tensor([need], dtype=torch.int64, device=dev)
            if G > 1:
                if _staged(self.group):
                    c = t.cpu(); dist.all_reduce(c, op=dist.ReduceOp.MAX, group=self.group); t = c
                else:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            cap = (int(t.item()) * 3 // 2 + 64 + 255) // 256 * 256
        else:
            cap = int(self.static_rows)
        cap = max(1, min(cap, r.R))
        n = G * cap
        kcap = db.B * (self.Ls + L.SN_CAP + 3)      # ids of the largest batch the kernels take
        side, side2 = concurrent_streams(self.device, 2)     # on hardware queues of their own (see there)
        st = dict(cap=cap, n=n, kcap=kcap, B=db.B,
                  status=torch.zeros(1, dtype=torch.int32, device=dev),
                  status_host=torch.zeros(1, dtype=torch.int32).pin_memory(), status_step=0,   # mirror of `status`, copied behind every plan
                  stamp=torch.ones(1, dtype=torch.int32, device=dev),           # uint32 on the device side; never 0
                  gf=torch.zeros(n, W, dtype=torch.float32, device=dev),
                  vals=torch.zeros(n, W, dtype=torch.float32, device=dev) if G > 1 else None,
                  lws=torch.empty(int(self.lib.tlsan_shard_apply_lazy_workspace(n, self.C)) + 256, dtype=torch.uint8, device=dev),
                  slots=[None] * _STATIC_SLOTS, next=0, side=side, side2=side2,
                  fork=torch.cuda.Event(),
                  side_group=None, checked=0, graphs=0, warm=False)
        st["fork"].record()
        from .model import _STARTED_WORDS
        st["started"] = torch.zeros(16, dtype=torch.int32).pin_memory()
        _STARTED_WORDS.append(st["started"])          # (a queued step writes it when it starts: it outlives the model)
        st["started_word"] = C.c_uint32.from_address(st["started"].data_ptr())
        st["start_seq"] = 0
        if G > 1 and not self.deferred_ids and (_staged(self.group) or os.environ.get("TLSAN_SIDE_COMM", "0") == "1"):
            # the next batch's id exchange runs on the side stream while the main stream's collectives are in
            # flight: a communicator of its own keeps the two sequences independent.  Two RCCL communicators with
            # kernels in flight on one device are only safe if every rank's GPU schedules them in a compatible
            # order, which has never been exercised on hardware (the pool's boxes have one GPU): over RCCL this is
            # opt-in (TLSAN_SIDE_COMM=1); by default the id exchange of a plan built ahead is issued by the step that
            # uses it, on the main stream (one more small all-to-all on the critical path, no second communicator).
            # The host-staged gloo exchange of the tests is serial either way and keeps the side group.
            ranks = list(range(G)) if self.group is None else dist.get_process_group_ranks(self.group)
            st["side_group"] = dist.new_group(ranks, backend=dist.get_backend(self.group))
            # communicators are set up at their first collective: do that here, on every rank at the same point, not in
            # the middle of the first step with the main communicator's collectives in flight
            warm = torch.zeros(G, dtype=torch.int32, device=dev)
            a2a(torch.empty_like(warm), warm, None, None, st["side_group"])
            torch.cuda.synchronize(dev)
        dims = L.Dims(n, n, self.C, self.d, self.di, self.dc, self.H, self.Ls)
        nst = self.lib.tlsan_state_bytes(C.byref(dims))
        nws = self.lib.tlsan_workspace_bytes(C.byref(dims), db.B, L.SN_CAP)
        if nst == 0 or nws == 0:
            raise L.TlsanError(self.lib.tlsan_last_error().decode())
        st["dims"] = dims
        if self._ws is None or self._ws.numel() < nws:
            self._ws = torch.empty(int(nws), dtype=torch.uint8, device=dev)
        wire = self.wire_dtype == "bf16"
        di = self.di
        tail = max(1, self.Ls)                              # fp32 floats after the embedding: item_b / the position weights
        pitch = (2 * di + 4 * tail + 15) // 16 * 16 if wire else 4 * W      # bytes per slot of the compact table
        st.update(wire=wire, pitch=pitch, tail=tail)
        if wire:
            if di % 4:
                raise NotImplementedError("wire_dtype='bf16': embedding widths must be multiples of 4")
            st["cate_bf16"] = torch.zeros(self.C, self.dc, dtype=torch.bfloat16, device=dev)
        for k in range(_STATIC_SLOTS):
            sl = dict(k=k, rank=torch.empty(r.nkeys, dtype=torch.int32, device=dev),
                      uniq=torch.empty(r.nkeys, dtype=torch.int32, device=dev),
                      n_uniq=torch.zeros(1, dtype=torch.int32, device=dev),
                      sendbuf=torch.zeros(G, 1 + cap, dtype=torch.int32, device=dev),
                      recvbuf=torch.zeros(G, 1 + cap, dtype=torch.int32, device=dev) if G > 1 else None,
                      cate_c=torch.full((n,), -1, dtype=torch.int32, device=dev),
                      comp=torch.zeros(kcap, dtype=torch.int32, device=dev),
                      rows=torch.zeros(n, pitch, dtype=torch.uint8, device=dev),
                      table=torch.zeros(n, pitch, dtype=torch.uint8, device=dev) if G > 1 else None,
                      recv_rows=torch.full((n,), -1, dtype=torch.int32, device=dev),
                      state=torch.zeros(int(nst), dtype=torch.uint8, device=dev), db=None, views={},
                      done=(torch.cuda.Event(), torch.cuda.Event()), planned=torch.cuda.Event(), pending=False, fresh=False)
            sl["state"][:4].view(torch.float32).fill_(1.0)     # table scale P = 1 (the owners apply the decay)
            sl["plan_args"], sl["step_args"] = {}, {}
            for ev in sl["done"] + (sl["planned"],):            # (a torch event gets its handle at its first record: the
                ev.record()                                     #  library's argument blocks hold the raw handles)
            st["slots"][k] = sl
        self._st = st
        return st

    def _static_views(self, sl, db):
        """ctypes views of a batch on a slot's compact table (cached: the pointers are constants of the pair)"""
        hit = sl["views"].get(id(db))
        if hit is not None and hit[0] is db:
            return hit[1]
        if len(sl["views"]) >= 16:
            sl["views"].clear()
        st = self._st
        if int(db.keys.numel()) > st["kcap"]:
            raise ValueError("static_rows: this batch holds more ids than the buffers were sized for (batch size grew?)")
        B, Ls, Sn = db.B, self.Ls, db.Sn
        base_c, o = sl["comp"].data_ptr(), 0
        p_i = base_c + 4 * o; o += B
        p_hist = base_c + 4 * o; o += B * Ls
        p_new = base_c + 4 * o if Sn > 0 else db.hist_i_new.data_ptr(); o += B * Sn
        p_j = None
        if db.j is not None:
            p_j = base_c + 4 * o; o += B
        p_u = base_c + 4 * o
        ptr = lambda t: None if t is None else t.data_ptr()
        cb = L.Batch(B, Sn, p_u, p_i, p_j, ptr(db.y), p_hist, p_new, ptr(db.hist_t), ptr(db.sl), ptr(db.sl_new), ptr(db.u_cate))
        table = sl["rows"] if self.world == 1 else sl["table"]
        base = table.data_ptr()
        if st["wire"]:      # bf16 embedding values + fp32 tail in one slot: strides in elements of each pointer's type
            pitch = st["pitch"]
            cp = L.Params(base, base + 2 * self.di, base, base + 2 * self.di, st["cate_bf16"].data_ptr(),
                          self.dense.data_ptr(), self.dense_KT.data_ptr(), sl["cate_c"].data_ptr(),
                          pitch // 2, pitch // 4, pitch // 2, pitch // 4, self._P.data_ptr(), L.TABLE_BF16)
        else:
            cp = L.Params(base, base + 4 * self.di, base, base + 4 * self.di, self.cate_emb.data_ptr(),
                          self.dense.data_ptr(), self.dense_KT.data_ptr(), sl["cate_c"].data_ptr(),
                          self.W, self.W, self.W, self.W, self._P.data_ptr())
        out = (cp, cb)
        sl["views"][id(db)] = (db, out)
        return out

    def _static_step_args(self, sl, db):
        """tlsan_static_step of (slot, batch): every pointer of the step's four launches (cached; lr, the dropout seed and
        the sample offset are filled in per step)."""
        hit = sl["step_args"].get(id(db))
        if hit is not None and hit[0] is db and hit[2] is self._ws:     # (the workspace may have been re-allocated)
            return hit[1]
        if len(sl["step_args"]) >= 16:
            sl["step_args"].clear()
        st = self._st
        G, W, di, Ls, Cc = self.world, self.W, self.di, self.Ls, self.C
        cp, cb = self._static_views(sl, db)
        n_dense, n_cate = self.lay.n_dense, Cc * self.dc
        flat, gf = self._flat, st["gf"]
        fp, g0 = flat.data_ptr(), gf.data_ptr()
        tail = fp + 4 * (n_dense + n_cate)
        rb = (sl["recvbuf"] if G > 1 else sl["sendbuf"]).data_ptr()
        ss = L.StaticStep()
        ss.shard, ss.ld, ss.R, ss.W, ss.recvbuf, ss.cap, ss.G = self.shard.data_ptr(), W, self.router.R, W, rb, st["cap"], G
        ss.rows_out, ss.recv_rows = sl["rows"].data_ptr(), sl["recv_rows"].data_ptr()
        ss.slots64, ss.stamp = self._slots64.data_ptr(), st["stamp"].data_ptr()
        ss.wire, ss.d_emb, ss.tail, ss.pitch = (1 if st["wire"] else 0), di, st["tail"], st["pitch"]
        ss.dims, ss.cp, ss.cb = C.pointer(st["dims"]), C.pointer(cp), C.pointer(cb)
        ss.hp = L.HParams(1.0, 0.0, self.clip, L.NORM_TF18, L.L2_DENSE, 0, 1, self.dropout, 0, 0)
        ss.go = L.GradsOut(g0, g0 + 4 * di, g0, g0 + 4 * di, fp + 4 * n_dense, fp, W, W, W, W, 2)
        ss.out = L.StepOut(tail, self._gn_local.data_ptr(), None, tail + 4)
        ss.state, ss.ws, ss.ws_bytes = sl["state"].data_ptr(), self._ws.data_ptr(), self._ws.numel()
        ss.flat, ss.n_dense, ss.n_cate, ss.lr, ss.reg, ss.clip = fp, n_dense, n_cate, 1.0, self.reg, self.clip
        ss.S_cate, ss.dense, ss.dense_KT = self._sq.data_ptr() + 8, self.dense.data_ptr(), self.dense_KT.data_ptr()
        ss.dims_full, ss.step_dev = C.pointer(self.dims_full), self._step_dev.data_ptr()
        ss.loss_out, ss.gnorm_out, ss.opt = self.last_loss.data_ptr(), self.last_gnorm.data_ptr(), C.pointer(self._sopt_lazy)
        ss.cI, ss.reg_item, ss.reg_user = self.cI, di, di + Ls
        ss.vals, ss.ldv, ss.marked, ss.gscale = (st["vals"] if G > 1 else gf).data_ptr(), W, 1, 1.0 / G
        ss.cate_emb, ss.C, ss.dc, ss.g_cate = self.cate_emb.data_ptr(), Cc, self.dc, fp + 4 * n_dense
        ss.sumsq_out, ss.sumsq_f32, ss.scale = self._sq.data_ptr(), tail + 8, self._P.data_ptr()
        ss.lws, ss.lws_bytes = st["lws"].data_ptr(), st["lws"].numel()
        sl["step_args"][id(db)] = (db, ss, self._ws)
        return ss

    def __del__(self):
        # plans still queued with the library's launch thread hold raw pointers into this model's buffers: let the
        # thread issue them before the buffers go (the streams then order their kernels before the frees)
        try:
            if getattr(self, "_st", None) is not None:
                self.lib.tlsan_shard_plans_flush()
        except Exception:
            pass

    def _flush_plans(self):
        """Plans handed to the library's launch thread (tlsan_shard_step_static, TLSAN_PLAN_ASYNC) are issued when that
        thread gets to them: wait until it has, before anything on this thread that must come BEHIND a plan's launches
        (waiting on its events or streams, clearing what it writes, a stream capture)."""
        L.check(self.lib.tlsan_shard_plans_flush(), "tlsan_shard_plans_flush")

    def _static_discard(self, sl, stream):
        """A plan that was built into the slot and never trained (an announcement that was abandoned, a restore, a
        capture over a slot an eager step had planned): its destination index is still counted into the slot's
        state -- tlsan_batch_index only ADDS to the use counters, the step that consumes a plan counts them back to
        zero -- so a second plan on top would double every count and the row sums would run past their segments.
        Clear everything behind the state's header (the category index is rebuilt by every plan anyway), after
        whatever the abandoned plan still has in flight."""
        if not torch.cuda.is_current_stream_capturing():
            self._flush_plans()
        if sl["pending"] and not torch.cuda.is_current_stream_capturing():
            stream.wait_event(sl["done"][0])
            stream.wait_event(sl["done"][1])
        with torch.cuda.stream(stream):
            sl["state"][_STATE_HDR_BYTES:].zero_()
        sl["pending"] = sl["fresh"] = False
        sl["db"] = None

    def _static_plan_args(self, sl, db, stream, group, stream2):
        """tlsan_static_plan of (slot, batch, streams): the pointers are constants of the combination (cached)."""
        st, r = self._st, self.router
        key = (id(db), stream.cuda_stream, None if stream2 is None else stream2.cuda_stream)
        hit = sl["plan_args"].get(key)
        if hit is not None and hit[0] is db:
            return hit[1]
        if len(sl["plan_args"]) >= 16:
            sl["plan_args"].clear()
        cp, cb = self._static_views(sl, db)
        ids_here = self._static_ids_in_plan(stream, group)
        pa = L.StaticPlan(db.keys.data_ptr(), int(db.keys.numel()), r.R, r.G, self.cate_by_key.data_ptr(),
                          self._flags.data_ptr(), sl["rank"].data_ptr(), sl["uniq"].data_ptr(), sl["n_uniq"].data_ptr(),
                          sl["sendbuf"].data_ptr(), st["cap"], sl["cate_c"].data_ptr(), sl["comp"].data_ptr(),
                          st["status"].data_ptr(), st["status_host"].data_ptr(),
                          C.pointer(st["dims"]), C.pointer(cp), C.pointer(cb), sl["state"].data_ptr(),
                          stream.cuda_stream, None if stream2 is None else stream2.cuda_stream,
                          None, sl["planned"].cuda_event, sl["done"][0].cuda_event if stream2 is not None else None,
                          sl["done"][1].cuda_event if stream2 is not None else None,
                          1 if (stream2 is not None and not ids_here) else 0)
        sl["plan_args"][key] = (db, pa)
        return pa

    def _static_ids_in_plan(self, stream, group):
        """Does the plan carry its own id all-to-all (on `stream`)?  Over several ranks: yes when it is built in line on
        the main stream, or ahead on a side stream with a communicator of its own (or over the host-staged gloo exchange
        of the tests, which is serial anyway); a plan built ahead WITHOUT one (the default over RCCL, see _static_setup;
        deferred_ids=True forces it anywhere) leaves the exchange to the step that uses it, in the main communicator's
        program order."""
        if self.world == 1:
            return False
        if stream == torch.cuda.current_stream(self.device):
            return True
        if self.deferred_ids:
            return False
        return group is not None or _staged(self.group)

    def _static_plan_tail(self, sl, stream, group, stream2):
        """What follows the plan's launches on the host: the id exchange, when the plan carries it, and the event behind it."""
        ids_here = self._static_ids_in_plan(stream, group)
        sl["ids_sent"] = True
        if self.world > 1:
            if not ids_here:
                sl["ids_sent"] = False
            else:
                with torch.cuda.stream(stream):
                    a2a(sl["recvbuf"].view(-1), sl["sendbuf"].view(-1), None, None, group if group is not None else self.group)
                if stream2 is not None:
                    sl["done"][0].record(stream)

    def _static_plan(self, db, k, stream, group, stream2=None, defer=False):
        """Routing plan of `db` into slot k, destination index included: device work only, queued on `stream`
        (the destination index on `stream2` when given: it needs the plan's compact ids, the category index needs its
        category map -- two independent tails; the slot's `done` events then mark their ends).  One library call
        (tlsan_shard_plan_static); defer=True returns its argument block instead of issuing it (the step's own call
        issues it behind its forward/backward kernel) -- only for plans whose id exchange is not part of the plan."""
        st, r = self._st, self.router
        sl = st["slots"][k]
        if sl["fresh"]:          # an unconsumed plan sits in the slot: take its index out first
            self._static_discard(sl, stream)
        pa = self._static_plan_args(sl, db, stream, group, stream2)
        # (the overflow word travels to pinned host memory behind the plan that may have raised it, on the plan's own
        #  stream; the next steps look at the copy without synchronising.  Not part of a recording.)
        pa.status_host = None if torch.cuda.is_current_stream_capturing() else st["status_host"].data_ptr()
        if not defer:
            L.check(self.lib.tlsan_shard_plan_static(C.byref(pa)), "tlsan_shard_plan_static")
            self._static_plan_tail(sl, stream, group, stream2)
        sl["pending"] = stream2 is not None
        sl["db"] = db
        sl["fresh"] = True      # (a step consumes its plan: the destination index counts down to zero)
        if defer:
            sl["ids_sent"] = self.world == 1
            return pa
        return None

    def check_static_overflow(self):
        """Host check of the static exchange (synchronises): raises when a batch needed more row slots of one owner
        than the exchange holds -- the steps since then are wrong."""
        st = self._st
        if st is None:
            return
        self._flush_plans()
        need = int(st["status"].item())
        if need > st["cap"]:
            raise RuntimeError("static_rows: a batch needed %d rows of one owner, the exchange holds %d per pair; "
                               "rebuild the model with static_rows >= %d" % (need, st["cap"], need))

    def _train_static(self, db, lr, next_batch, weight, sample0, after_next=None):
        st = self._st if self._st is not None else self._static_setup(db)
        G, W, di, Ls, Cc = self.world, self.W, self.di, self.Ls, self.C
        NS = _STATIC_SLOTS
        main = torch.cuda.current_stream(self.device)
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            # an exchange that overflowed (a plan needed more rows of one owner than `cap`) truncates the step on the
            # device: seen here one or two steps later through the pinned copy -- not at the next 1024-step check
            need = int(st["status_host"][0])
            if need > st["cap"]:
                raise RuntimeError("static_rows: a batch planned around step %d needed %d rows of one owner, the exchange "
                                   "holds %d per pair: the steps since then are wrong; rebuild the model with static_rows >= %d"
                                   % (self._step, need, st["cap"], need))
        k = st["next"]
        sl = st["slots"][k]
        planned_inline = False
        if sl["db"] is not db or not sl["fresh"]:      # not announced by an earlier step: plan it now, in line
            planned_inline = True
            if not capturing:
                self._flush_plans()
                main.wait_stream(st["side"])     # (whatever an abandoned announcement left running in the slots)
                main.wait_stream(st["side2"])
            self._static_plan(db, k, main, self.group)
        elif sl["pending"]:
            if not capturing:                    # (recorded steps join their side work at their own end)
                # a HOST wait: announced two batches ahead the plan is long done, and a device-side wait is a barrier packet
                # in the main queue (~6 us of idle GPU each; the single-GPU Model waits on the host for the same reason)
                self._flush_plans()                  # (the launch thread has recorded the events by then)
                sl["done"][0].synchronize()
                sl["done"][1].synchronize()
            sl["pending"] = False
        sl["fresh"] = False
        if G > 1 and not sl.get("ids_sent", True):     # (plan built ahead without a side communicator: see _static_plan)
            a2a(sl["recvbuf"].view(-1), sl["sendbuf"].view(-1), None, None, self.group)
            sl["ids_sent"] = True
        st["next"] = (k + 1) % NS
        ahead = [(self.device_batch(b), (k + 1 + j) % NS) for j, b in enumerate((next_batch, after_next)) if b is not None]
        ahead = [(b, kk) for b, kk in ahead if not (st["slots"][kk]["db"] is b and st["slots"][kk]["fresh"])]
        # EVERY eager step stamps the pinned word when its fused kernel starts (not only the steps that announce a batch:
        # a plan waits for "step t - 1 has started", which only names step t - 1 if that step carried a stamp -- ADVICE r4).
        # The word paces a plan only when the step before this one was such a stamped eager step; behind a graph replay, a
        # capture or the first step of a model the plans are ordered by an event on the main stream instead.
        stamp = not capturing
        # (... nor when this step has just planned its own batch in line, on the main stream: the plans share their mark
        #  scratch, and "the step before has started" says nothing about a plan queued behind it -- the event does)
        use_flag = bool(ahead) and stamp and st.get("prev_stamped", False) and not planned_inline     # (see tlsan_shard_step_static: no event on the main stream)
        if ahead and not use_flag:
            st["fork"].record(main)      # everything before this step: the slots the new plans go to are free from here

        # The step is issued in as few library calls as there are collectives in it (tlsan_shard_step_static: one at one
        # rank) from an argument block that is a constant of (slot, batch) -- as ~15 Python-level calls the HOST took
        # 100 us per step for 77 us of kernels.  Where the plans of the announced batches go in the launch order (the
        # device-side dependencies are the same either way): behind the forward/backward.  That kernel fills the GPU,
        # so a plan can only run beside the small kernels around it; queued ahead of it, the plan delays it (one rank,
        # same box: graph replay 113 us with the plan queued after the kernel, 153 us before).
        ss = self._static_step_args(sl, db)
        ss.hp.lr = float(lr); ss.lr = float(lr)
        ss.hp.dropout_seed = self.dropout_seed() if self.dropout > 0.0 else 0
        ss.hp.dropout_sample0 = int(sample0)
        sp = C.c_void_p(main.cuda_stream)
        n_dense, n_cate = self.lay.n_dense, Cc * self.dc
        flat, gf = self._flat, st["gf"]
        plans, late = [], []          # argument blocks the step's own call issues / plans whose id exchange Python issues
        for b_, kk in ahead:
            if self._static_ids_in_plan(st["side"], st["side_group"]):
                late.append((b_, kk))
            else:
                pa = self._static_plan(b_, kk, st["side"], st["side_group"], st["side2"], defer=True)
                pa.ev_fork = None if use_flag else st["fork"].cuda_event
                plans.append(pa)
        parr = (C.POINTER(L.StaticPlan) * max(1, len(plans)))(*[C.pointer(x) for x in plans]) if plans else None

        def run(phases, with_plans=False):
            # (plans: issued by the library's launch thread while this one goes on with the main stream -- not under capture)
            if with_plans and plans and use_flag and _PLAN_THREAD and (G == 1 or _PLAN_THREAD_ENV == "1"):
                phases |= L.PLAN_ASYNC
            L.check(self.lib.tlsan_shard_step_static(C.byref(ss), phases, parr if with_plans else None,
                                                     len(plans) if with_plans else 0, sp), "tlsan_shard_step_static")

        if stamp:         # the fused kernel stores the step's number into a pinned word when it begins to run
            st["start_seq"] = (st["start_seq"] + 1) & 0xFFFFFFFF       # (the full 32 bits: the library compares (int32)(word - after))
            ss.out.started, ss.out.started_value = st["started"].data_ptr(), st["start_seq"]
            # the announced batches' plans go to slots that steps t - 2 and t - 3 were the last to use: "step t - 1 has
            # started" is all they wait for -- long true when this step is queued, so the host does not block
            ss.plans_after = (st["start_seq"] - 1) & 0xFFFFFFFF
        else:
            ss.out.started, ss.out.started_value = None, 0

        def plan_late():
            if use_flag:
                word, want, t0, polls = st["started_word"], (st["start_seq"] - 1) & 0xFFFFFFFF, None, 0
                while ((word.value - want) & 0xFFFFFFFF) >= 0x80000000:     # (not yet reached: the library's (int32)(word - after) < 0)
                    polls += 1
                    if polls & 255:
                        continue
                    time.sleep(0)
                    if t0 is None:
                        t0 = time.perf_counter()
                    elif time.perf_counter() - t0 > 30.0:
                        raise RuntimeError("train_async: the step's first kernel did not start within 30 s")
            else:
                st["side"].wait_event(st["fork"])
            for b_, kk in late:
                self._static_plan(b_, kk, st["side"], st["side_group"], st["side2"])

        if st["wire"]:
            st["cate_bf16"].copy_(self.cate_emb)      # (round to nearest even; the table is small and replicated)
        if G == 1 and weight == 1.0 and not late:
            run(L.PHASE_GATHER | L.PHASE_GRADS | L.PHASE_SUMMARY | L.PHASE_APPLY, True)
        else:
            run(L.PHASE_GATHER)
            if G > 1:
                a2a(sl["table"].view(-1), sl["rows"].view(-1), None, None, self.group)
            run(L.PHASE_GRADS, True)
            if late:
                plan_late()
            if weight != 1.0:
                kk = n_dense + n_cate
                flat[:kk + 1].mul_(float(weight))
                flat[kk + 1:kk + 2].mul_(float(weight) ** 2)
                gf.mul_(float(weight))
            if G > 1 and self.coalesce:
                # the two exchanges behind the kernels are independent of each other (the row sums travel unscaled: the
                # clip coefficient reaches the owners through the summary): one RCCL group instead of two collectives
                if _staged(self.group):      # (gloo: no groups of collectives -- the same two exchanges, back to back, in the same place)
                    allreduce_sum(flat, self.group)
                    a2a(st["vals"].view(-1), gf.view(-1), None, None, self.group)
                else:
                    with dist._coalescing_manager(group=self.group, device=torch.device(self.device), async_ops=False):
                        dist.all_reduce(flat, group=self.group)
                        dist.all_to_all_single(st["vals"].view(-1), gf.view(-1), group=self.group)
                run(L.PHASE_SUMMARY | L.PHASE_APPLY)
            else:
                if G > 1:
                    allreduce_sum(flat, self.group)
                run(L.PHASE_SUMMARY)
                if G > 1:
                    a2a(st["vals"].view(-1), gf.view(-1), None, None, self.group)
                run(L.PHASE_APPLY)
        if ahead and capturing:              # a recording must join its forks; eager steps wait where a plan is used
            main.wait_stream(st["side"])
            main.wait_stream(st["side2"])
            for b, kk in ahead:
                st["slots"][kk]["pending"] = False
        self._step += 1
        st["prev_stamped"] = stamp          # (a recorded step never stamps: the step behind a capture or replay takes the event path)
        if not capturing:
            st["warm"] = True
            if self.renorm_every and self._step % self.renorm_every == 0:
                self.fold_scale()
            if self._step - st["checked"] >= 1024:
                st["checked"] = self._step
                self.check_static_overflow()
        return db

    def capture_step(self, batch, next_batch, lr):
        """Record one static-shape step (this batch in the slot it is planned in, the next batch's plan on the side
        streams) in a HIP graph.  Replays must follow the order of capture: the step of `batch` expects its plan
        where the previous step left it.  Capture and replay alternately along a cycle whose length is a multiple of
        the number of plan slots (4):  g0 = capture(b0, b1); replay(g0); g1 = capture(b1, b2); replay(g1); ...;
        g3 = capture(b3, b0); replay(g3) -- then replay the graphs in that order.  lr is baked in."""
        if not self.static_rows:
            raise RuntimeError("capture_step needs static_rows (the exchange sizes of the dynamic step pass through the host)")
        db, ndb = self.device_batch(batch), self.device_batch(next_batch)
        st = self._st
        if st is None or not st["warm"]:
            raise RuntimeError("capture_step: run one eager step first (one-time initialisation cannot be recorded)")
        main = torch.cuda.current_stream(self.device)
        self._flush_plans()
        main.wait_stream(st["side"])                          # nothing of the eager steps is left in flight
        main.wait_stream(st["side2"])
        k0, step0 = st["next"], self._step
        k1 = (k0 + 1) % _STATIC_SLOTS
        for sl in st["slots"]:
            sl["pending"] = False
        if st["slots"][k0]["db"] is not db or not st["slots"][k0]["fresh"]:   # the recorded step expects its plan in place
            self._static_plan(db, k0, main, self.group)
        if self.world > 1 and not self._static_ids_in_plan(st["side"], st["side_group"]):
            # plans built ahead leave their id exchange to the step that uses them: EVERY replay of this step consumes a
            # plan the previous graph built, so the recording must carry the exchange itself -- whatever the eager plan
            # above has already sent (round 3 recorded g0 without it and replays consumed the first exchange's result)
            st["slots"][k0]["ids_sent"] = False
        if st["slots"][k1]["fresh"]:                          # (an eager step announced a batch into k1: the graph builds the
            self._static_discard(st["slots"][k1], main)       #  next plan itself, every replay -- not on top of that one)
        st["slots"][k1]["fresh"] = False
        self._static_views(st["slots"][k0], db)               # ctypes structs are built outside the capture
        self._static_views(st["slots"][k1], ndb)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._train_static(db, lr, ndb, 1.0, 0)
        # recording queues nothing: take the host-side bookkeeping of the step back
        st["next"], self._step = k0, step0
        st["slots"][k0]["fresh"] = True
        st["slots"][k1]["fresh"] = False
        g._tlsan = (db, ndb, k0, self._ws)
        st["graphs"] += 1
        return g

    def replay(self, g):
        db, ndb, k0, ws = g._tlsan
        st = self._st
        if ws is not self._ws:
            raise RuntimeError("replay: the workspace was reallocated after this graph was captured; recapture the step")
        if st["next"] != k0 or st["slots"][k0]["db"] is not db or not st["slots"][k0]["fresh"] or st["slots"][k0]["pending"]:
            raise RuntimeError("replay: this step expects its batch planned in slot %d by the step before it "
                               "(replay the graphs in the order they were captured)" % k0)
        g.replay()
        k1 = (k0 + 1) % _STATIC_SLOTS
        st["slots"][k0]["fresh"] = False
        st["slots"][k1]["db"] = ndb
        st["slots"][k1]["fresh"] = True
        st["slots"][k1]["pending"] = False
        st["next"] = k1
        st["prev_stamped"] = False      # (a recorded step carries no stamp: the next eager step orders its plans by an event)
        self._step += 1

    def _lazy_stamp(self):
        """Stamp of this step's entries in _slots64 (tlsan_shard_apply_lazy): 1 .. 2^32-2, never 0 (the value
        of a cleared slot); the slots are cleared whenever the sequence restarts."""
        s = self._step % 0xFFFFFFFE
        if s == 0 and self._step > 0:
            self._slots64.zero_()
        return s + 1

    def train(self, sess, batch, lr, add_summary=False):
        self.train_async(batch, lr)
        return float(self.last_loss.item())

    # ------------------------------------------------------------------ evaluation
    def forward(self, batch, is_test=True, want_ranks=False):
        db = self.device_batch(batch, is_test)
        sl = self._plan_eval(db)
        table = self._fetch(sl)
        dims, cp, cb = self._compact(db, sl, table)
        li = torch.empty(db.B, dtype=torch.float32, device=self.device)
        lj = torch.empty(db.B, dtype=torch.float32, device=self.device) if db.j is not None else None
        ut = torch.empty(db.B, self.d, dtype=torch.float32, device=self.device) if want_ranks else None
        L.check(self.lib.tlsan_forward(C.byref(dims), C.byref(cp), C.byref(cb), li.data_ptr(),
                                       None if lj is None else lj.data_ptr(), None if ut is None else ut.data_ptr(),
                                       None, 0, self._stream()), "tlsan_forward")
        ranks = self._ranks(db, dims, cp, cb, ut) if want_ranks else None
        torch.cuda.current_stream(self.device).synchronize()   # `table` stays alive until done
        return (li, lj, ranks) if want_ranks else (li, lj)

    def _ranks(self, db, dims, cp, cb, ut):
        """All-items ranking with the items sharded: the label's own score comes from the
        compact table of the rank that holds the user; u_t, label scores and label ids are
        all-gathered; every rank counts the items of ITS shard that rank ahead of each label
        (ties -> lower global id, tf.nn.top_k's order); one all-reduce of the counts gives the
        ranks.  Same kernels as the single-GPU tlsan_eval_ranks."""
        B, st = db.B, self._stream()
        nws = self.lib.tlsan_workspace_bytes(C.byref(dims), B, 0)
        if self._ws is None or self._ws.numel() < nws:
            self._ws = torch.empty(int(nws * 1.25), dtype=torch.uint8, device=self.device)
        s_lab = torch.empty(B, dtype=torch.float32, device=self.device)
        L.check(self.lib.tlsan_eval_label_scores(C.byref(dims), C.byref(cp), ut.data_ptr(), cb.i, B, s_lab.data_ptr(),
                                                 self._ws.data_ptr(), self._ws.numel(), st), "tlsan_eval_label_scores")
        ut_all, s_all, lab_all = allgather_rows(ut, self.group), allgather_rows(s_lab, self.group), allgather_rows(db.i, self.group)
        Bt = int(ut_all.shape[0])
        nloc = ModPartition(self.I, self.world).local_count(self.rank)
        ldims = L.Dims(self.U, max(nloc, 1), self.C, self.d, self.di, self.dc, self.H, self.Ls)
        base = self.shard.data_ptr()
        lp = L.Params(base, base + 4 * self.di, base, base + 4 * self.di, self.cate_emb.data_ptr(), self.dense.data_ptr(),
                      self.dense_KT.data_ptr(), self._icl_local.data_ptr(), self.W, self.W, self.W, self.W,
                      self._P.data_ptr() if self.lazy else None)
        nws = self.lib.tlsan_workspace_bytes(C.byref(ldims), Bt, 0)
        if self._ews is None or self._ews.numel() < nws:
            self._ews = torch.empty(int(nws * 1.25), dtype=torch.uint8, device=self.device)
        counts = torch.zeros(Bt, dtype=torch.int32, device=self.device)
        if nloc > 0:
            L.check(self.lib.tlsan_eval_counts_shard(C.byref(ldims), C.byref(lp), ut_all.data_ptr(), s_all.data_ptr(),
                                                     lab_all.data_ptr(), Bt, self.world, self.rank, counts.data_ptr(),
                                                     self._ews.data_ptr(), self._ews.numel(), st), "tlsan_eval_counts_shard")
        if self.world > 1:
            allreduce_sum(counts, self.group)
        return counts[self.rank * B:(self.rank + 1) * B]

    def _hits(self, batch, n_valid=None):
        from .model import KS
        r = self.label_ranks(batch)
        if n_valid is not None:        # rows past n_valid only pad this rank's share to the common size
            r = r[:n_valid]
        h = torch.stack([(r < k).sum() for k in KS] + [torch.tensor(r.numel(), device=r.device)]).to(torch.int64)
        if self.world > 1:
            allreduce_sum(h, self.group)        # hits and rows of the GLOBAL test batch
        h = h.cpu().numpy()
        return h[:-1], int(h[-1])

    def eval_prec(self, sess, batch, n_valid=None):
        """Streaming precision_at_k over the global batch (model.py:265-281); cumulative like the reference's
        never-reset local variables (train.py:75-76,82).  Identical on every rank.  Every rank must pass
        the same number of rows (the all-gather is equal-sized); n_valid marks how many of them count."""
        from .model import KS
        h, n = self._hits(batch, n_valid)
        self._hits_p += h
        self._n_p += n
        return [self._hits_p[i] / (k * self._n_p) for i, k in enumerate(KS)]

    def eval_recall(self, sess, batch, n_valid=None):
        from .model import KS
        h, n = self._hits(batch, n_valid)
        self._hits_r += h
        self._n_r += n
        return [self._hits_r[i] / self._n_r for i in range(len(KS))]

    def label_ranks(self, batch):
        """rank of the positive item among ALL items for each test row of this rank's batch (model.py:140-156)"""
        return self.forward(batch, is_test=True, want_ranks=True)[2]

    def eval_auc(self, sess, batch):
        li, lj = self.forward(batch, is_test=True)
        return float(((li - lj) > 0).float().mean().item())

    # ------------------------------------------------------------------ inspection
    def _table_views(self):
        return self.shard[:self.cI], self.shard[self.cI:]

    def fold_scale(self):
        """lazy L2: multiply the scale into the stored tables (P -> 1); what is trained does not change.
        Done on a fixed schedule (renorm_every) so that P stays away from fp32 underflow, and before the
        parameters are read."""
        if not self.lazy:
            return
        P = self._P
        di, Ls = self.di, self.Ls
        self.shard[:self.cI, :di].mul_(P)
        self.shard[self.cI:, :di + Ls].mul_(P)
        self.cate_emb.mul_(P)
        self._P.fill_(1.0)
        self._refresh_squares()

    def _gather_tables(self, shard):
        """A fused [item | user] shard tensor of every rank -> full tables (numpy), on every rank."""
        if self.world > 1:
            # (RCCL gathers device buffers; gloo, used by the single-GPU multi-process tests, host ones)
            src = shard.cpu() if _staged(self.group) else shard
            outs = [torch.empty_like(src) for _ in range(self.world)]
            dist.all_gather(outs, src, group=self.group)
            outs = [o.cpu() for o in outs]
        else:
            outs = [shard.cpu()]
        I, U, G, di, Ls = self.I, self.U, self.world, self.di, self.Ls
        item = np.zeros((I, self.W), np.float32)
        user = np.zeros((U, self.W), np.float32)
        for r in range(G):
            t = outs[r].numpy()
            ni, nu = len(range(r, I, G)), len(range(r, U, G))
            item[r::G] = t[:ni]
            user[r::G] = t[self.cI:self.cI + nu]
        return dict(item_emb=item[:, :di].copy(), item_b=item[:, di].copy(),
                    user_emb=user[:, :di].copy(), usert_emb=user[:, di:di + Ls].copy())

    def _unpack_dense(self, flat):
        d, dh, lay = self.d, self.d // self.H, self.lay
        out = {}
        for k, off, shape in (("fwa1_W1", lay.f1_W1, (dh, dh)), ("fwa1_b1", lay.f1_b1, (dh,)),
                              ("fwa1_W2", lay.f1_W2, (dh, dh)), ("fwa1_b2", lay.f1_b2, (dh,)),
                              ("dense_K", lay.K, (d, d)), ("dense_b", lay.k0, (d,)),
                              ("fwa2_W1", lay.f2_W1, (dh, dh)), ("fwa2_b1", lay.f2_b1, (dh,)),
                              ("fwa2_W2", lay.f2_W2, (dh, dh)), ("fwa2_b2", lay.f2_b2, (dh,)), ("gamma", lay.gamma, ())):
            n = int(np.prod(shape)) if shape else 1
            out[k] = flat[off:off + n].reshape(shape).copy()
        return out

    def gather_params(self):
        """Full (un-sharded) parameters on every rank, as numpy (tests / checkpoints)."""
        self.fold_scale()
        out = self._gather_tables(self.shard)
        out["cate_emb"] = self.cate_emb.cpu().numpy()
        out.update(self._unpack_dense(self.dense.cpu().numpy()))
        return out

    # ------------------------------------------------------------------ checkpoints (model.py:302-313)
    def save(self, sess=None, sharded=True):
        """Sharded checkpoint: every rank writes its own rows (TLSAN-<step>.shard<r>of<G>.npz: the fused
        [item | user] shard as it sits in HBM), rank 0 adds the replicated parts (cate_emb, dense weights,
        counters) and the config JSON -- nothing is gathered, so tables that exceed one host's memory
        can be saved.  sharded=False gathers everything to rank 0 and writes the single-GPU format of
        tlsan_amd.model.Model.save (restores into either model, any world size).  Returns the path prefix."""
        import json
        self.check_static_overflow()       # never write down the result of a truncated exchange
        self.fold_scale()
        os.makedirs(self.config["model_dir"], exist_ok=True)
        base = os.path.join(self.config["model_dir"], "TLSAN-%d" % self._step)
        if not sharded:
            full = self.gather_params()                      # (a collective: every rank takes part)
            slots = self.gather_slots()                      # tf.train.Saver keeps the optimizer's slot variables too
            extra = {} if slots is None else {"slot%d/%s" % (n, k): v for n, sl_ in enumerate(slots, 1) for k, v in sl_.items()}
            if self.rank == 0:
                np.savez(base + ".npz", global_step=self._step, global_epoch_step=self._epoch, **full, **extra)
        else:
            sl = {} if self._sopt is None else {k: v.cpu().numpy() for k, v in self.slots.items()}
            np.savez("%s.shard%dof%d.npz" % (base, self.rank, self.world), shard=self.shard.cpu().numpy(),
                     cI=self.cI, W=self.W, item_count=self.I, user_count=self.U,
                     **{k: v for k, v in sl.items() if k.startswith("shard_")})
            if self.rank == 0:
                np.savez(base + ".replicated.npz", cate_emb=self.cate_emb.cpu().numpy(), dense=self.dense.cpu().numpy(),
                         global_step=self._step, global_epoch_step=self._epoch, world=self.world,
                         **{k: v for k, v in sl.items() if not k.startswith("shard_")})
        if self.rank == 0:
            json.dump(self.config, open(base + ".json", "w"), indent=2)
        if self.world > 1:
            dist.barrier(group=self.group)                   # the files exist when any rank returns
        return base

    def restore(self, sess, path):
        """`path`: the prefix save() returned (sharded checkpoint of the SAME world size), or a
        single-file .npz checkpoint of Model.save / save(sharded=False) -- then every rank keeps its rows."""
        if path.endswith(".npz"):
            z = np.load(path)
            self.set_params({k: z[k] for k in TABLE_KEYS + DENSE_KEYS})
            if self._sopt is not None and "slot1/item_emb" in z.files:   # the optimizer's accumulators (Model.save)
                self._set_slots([{k: z["slot%d/%s" % (n, k)] for k in TABLE_KEYS + DENSE_KEYS} for n in (1, 2)])
        else:
            rep_ = np.load(path + ".replicated.npz")
            if int(rep_["world"]) != self.world:
                raise ValueError("sharded checkpoint of %d ranks cannot be restored on %d (use sharded=False)"
                                 % (int(rep_["world"]), self.world))
            zs = np.load("%s.shard%dof%d.npz" % (path, self.rank, self.world))
            if tuple(zs["shard"].shape) != tuple(self.shard.shape) or int(zs["cI"]) != self.cI:
                raise ValueError("shard shape %s does not match this model" % (zs["shard"].shape,))
            self.shard.copy_(torch.as_tensor(zs["shard"]))
            self._P.fill_(1.0)
            self.cate_emb.copy_(torch.as_tensor(rep_["cate_emb"]))
            self.dense.copy_(torch.as_tensor(rep_["dense"]))
            K = self.dense[self.lay.K:self.lay.K + self.d * self.d].view(self.d, self.d)
            self.dense_KT.copy_(K.t())
            self._refresh_squares()
            if self._sopt is not None and "shard_s1" in zs.files:      # the optimizer's accumulators
                for k in self.slots:
                    self.slots[k].copy_(torch.as_tensor((zs if k.startswith("shard_") else rep_)[k]))
            z = rep_
        self._step = int(z["global_step"])
        self._epoch = int(z["global_epoch_step"])
        self._reset_step_state()

    def _reset_step_state(self):
        """After the parameters or the step counter were replaced: forget everything that was keyed by the old
        step sequence (the stamped slots of the lazy owner update) or built for an announced successor."""
        if self.lazy:
            self._slots64.zero_()
        self._slots = [None, None, None]
        self._next_slot = 0
        if self._st is not None:       # static-shape step: no plan is pending, the stamps restart
            self._st["stamp"].fill_(1)
            self._st["next"] = 0
            main = torch.cuda.current_stream(self.device)
            self._flush_plans()
            main.wait_stream(self._st["side"])
            main.wait_stream(self._st["side2"])
            for sl in self._st["slots"]:
                if sl["fresh"]:            # a plan nobody will train: its counters must not meet the next plan's
                    self._static_discard(sl, main)
                sl["db"] = None
                sl["fresh"] = sl["pending"] = False

    def _shard_layout(self, p):
        """Full tables (dict of numpy arrays named like the parameters) -> this rank's fused [item | user] rows."""
        G, r, di, Ls = self.world, self.rank, self.di, self.Ls
        gi = np.arange(r, self.I, G)
        gu = np.arange(r, self.U, G)
        t = np.zeros(tuple(self.shard.shape), np.float32)
        t[:len(gi), :di] = np.asarray(p["item_emb"], np.float32)[gi]
        t[:len(gi), di] = np.asarray(p["item_b"], np.float32)[gi]
        t[self.cI:self.cI + len(gu), :di] = np.asarray(p["user_emb"], np.float32)[gu]
        t[self.cI:self.cI + len(gu), di:di + Ls] = np.asarray(p["usert_emb"], np.float32)[gu]
        return t

    def _dense_flat(self, p):
        lay = self.lay
        flat = np.zeros(lay.n_dense, np.float32)
        for k, off in (("fwa1_W1", lay.f1_W1), ("fwa1_b1", lay.f1_b1), ("fwa1_W2", lay.f1_W2), ("fwa1_b2", lay.f1_b2),
                       ("dense_K", lay.K), ("dense_b", lay.k0), ("fwa2_W1", lay.f2_W1), ("fwa2_b1", lay.f2_b1),
                       ("fwa2_W2", lay.f2_W2), ("fwa2_b2", lay.f2_b2), ("gamma", lay.gamma)):
            a = np.asarray(p[k], np.float32).reshape(-1)
            flat[off:off + a.size] = a
        return flat

    def _set_slots(self, slots):
        """The two accumulator sets of adam / rmsprop / adadelta from full tables (Model.get_slots' format)."""
        for n, src in enumerate(slots, 1):
            self.slots["shard_s%d" % n].copy_(torch.as_tensor(self._shard_layout(src)))
            self.slots["cate_s%d" % n].copy_(torch.as_tensor(np.asarray(src["cate_emb"], np.float32)))
            self.slots["dense_s%d" % n].copy_(torch.as_tensor(self._dense_flat(src)))

    def gather_slots(self):
        """Full (un-sharded) optimizer accumulators on every rank, in Model.get_slots' format (None for sgd)."""
        if self._sopt is None:
            return None
        out = []
        for n in (1, 2):
            full = self._gather_tables(self.slots["shard_s%d" % n])
            full["cate_emb"] = self.slots["cate_s%d" % n].cpu().numpy()
            full.update(self._unpack_dense(self.slots["dense_s%d" % n].cpu().numpy()))
            out.append(full)
        return out

    def _refresh_squares(self):
        it, us = self._table_views()
        di, Ls = self.di, self.Ls
        # running sums of squares of the regularised tables (tf.nn.l2_loss terms, model.py:164-169)
        self._sq[0] = it[:, :di].double().pow(2).sum() + us[:, :di + Ls].double().pow(2).sum()
        self._sq[1] = self.cate_emb.double().pow(2).sum()
        self._flat[self.lay.n_dense + self.C * self.dc + 2] = self._sq[0].float()   # rides in the all-reduce

    def set_params(self, p):
        """Load full parameters (dict of numpy arrays); every rank keeps its own rows."""
        self.shard.copy_(torch.as_tensor(self._shard_layout(p)))
        self.cate_emb.copy_(torch.as_tensor(np.asarray(p["cate_emb"], np.float32)))
        self._P.fill_(1.0)
        self._pack_dense(p)
        self._refresh_squares()
        self._reset_step_state()
