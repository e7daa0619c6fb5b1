"""Multi-GPU TLSAN step: one process per GPU, embedding tables row-sharded, RCCL over xGMI.

The reference is single-GPU (train.py:53,146); this is the MI355X-native scale-out of
SURVEY.md section 8e.  Samples are independent units: every rank trains its own batch
(data parallel, weak scaling).  State placement:

  * ``user_emb`` + ``usert_emb`` rows: sharded by ``user_id % world``  (fused rows
    ``[user_emb | usert_emb | pad]``);
  * ``item_emb`` + ``item_b`` rows: sharded by ``item_id % world`` (fused ``[item_emb | item_b | pad]``;
    mod-G spreads the Zipf-hot items);
  * ``cate_emb`` (<= 5 MB), ``item_cate_list`` and the dense attention weights: replicated.

One step = (1) mark the rows the local batch touches in an owner-major key space and compact the
marks (one HIP scan: distinct rows already in all-to-all order + the id -> compact-row map),
(2) all-to-all of the per-owner counts and local row numbers, (3) all-to-all of the rows back
into a compact per-step table the HIP kernels run on in place (row strides,
tlsan_params.ld_*), (4) fused forward/backward + exact per-row gradient sums (tlsan_grads),
(5) ONE all-reduce of [dense grads | cate grads | loss | norm terms], (6) all-to-all of the
per-row gradients back to the owners, (7) owners apply them with the deterministic
tlsan_rows_apply (dense L2 decay of every local row, as the reference).  Items and users share
one fused shard table and one exchange each way; one host sync per step (split sizes).

``KeyRouter`` / ``RowExchange`` are device-agnostic torch + torch.distributed plumbing (run on
CPU/gloo in the tests); all arithmetic on rows is in libtlsan_hip.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib as L
from .model import DENSE_KEYS, TABLE_KEYS, DeviceBatch, Model, _Var, _Writer


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

    def __init__(self, config, item_cate_list, device="cuda:0", seed=1234, group=None):
        if not dist.is_initialized():
            raise RuntimeError("ShardedModel needs torch.distributed to be initialised (one process per GPU)")
        if config.get("optimizer", "sgd") != "sgd" or config.get("dropout", 0.0) != 0.0 or config.get("num_blocks", 1) != 1:
            raise NotImplementedError("only optimizer='sgd', dropout=0, num_blocks=1 (see tlsan_amd.model.Model)")
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
        self._sq = torch.zeros(3, dtype=torch.float64, device=dev)   # rows_apply sumsq outputs
        self._out = torch.zeros(4, dtype=torch.float32, device=dev)  # loss, gnorm, sq_rows (local)
        self._step_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self.last_loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.last_gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._state = self._ws = self._rws = None
        self._scan_prefix = torch.empty(self.router.nkeys, dtype=torch.int32, device=dev)
        self._scan_uniq = torch.empty(self.router.nkeys, dtype=torch.int32, device=dev)
        self._scan_n = torch.zeros(1, dtype=torch.int32, device=dev)
        self._step = 0
        self._epoch = 0
        self.global_step = _Var(lambda: self._step)
        self.global_epoch_step = _Var(lambda: self._epoch)
        self.train_writer, self.eval_writer = _Writer("train"), _Writer("eval")
        self._cate_ids = torch.arange(Cc, dtype=torch.int32, device=dev)
        self.set_params(Model.init_params(config, seed))   # identical on every rank (numpy, seeded)

    # ------------------------------------------------------------------ helpers
    def _pack_dense(self, p):
        d = self.d
        lay = self.lay
        flat = np.zeros(lay.n_dense, np.float32)
        for k, off in (("fwa1_W1", lay.f1_W1), ("fwa1_b1", lay.f1_b1), ("fwa1_W2", lay.f1_W2), ("fwa1_b2", lay.f1_b2),
                       ("dense_K", lay.K), ("dense_b", lay.k0), ("fwa2_W1", lay.f2_W1), ("fwa2_b1", lay.f2_b1),
                       ("fwa2_W2", lay.f2_W2), ("fwa2_b2", lay.f2_b2), ("gamma", lay.gamma)):
            a = np.asarray(p[k], np.float32).reshape(-1)
            flat[off:off + a.size] = a
        self.dense.copy_(torch.as_tensor(flat))
        self.dense_KT.copy_(self.dense[lay.K:lay.K + d * d].view(d, d).t())

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _scan(self, flags):
        L.check(self.lib.tlsan_scan_compact(flags.data_ptr(), flags.numel(), self._scan_prefix.data_ptr(),
                                            self._scan_uniq.data_ptr(), self._scan_n.data_ptr(), self._stream()),
                "tlsan_scan_compact")
        return self._scan_prefix, self._scan_uniq, self._scan_n

    def device_batch(self, batch, is_test=False):
        """Upload the batch and map its ids into the router's key space (elementwise, once)."""
        db = batch if isinstance(batch, DeviceBatch) else DeviceBatch(batch, self.device, is_test, self.Ls)
        if not hasattr(db, "keys"):
            r = self.router
            parts = [r.item_keys(db.i.long()), r.item_keys(db.hist_i.reshape(-1).long())]
            if db.Sn > 0:
                parts.append(r.item_keys(db.hist_i_new.reshape(-1).long()))
            if db.j is not None:
                parts.append(r.item_keys(db.j.long()))
            parts.append(r.user_keys(db.u.long()))
            db.keys = torch.cat(parts)
        return db

    def _buffers(self, dims, cp, B, Sn, n_rows_max):
        nst = self.lib.tlsan_state_bytes(C.byref(dims))
        nws = self.lib.tlsan_workspace_bytes(C.byref(dims), B, Sn)
        if nst == 0 or nws == 0:
            raise L.TlsanError(self.lib.tlsan_last_error().decode())
        if self._state is None or self._state.numel() < nst:
            self._state = torch.zeros(int(nst * 1.5), dtype=torch.uint8, device=self.device)
        if self._ws is None or self._ws.numel() < nws:
            self._ws = torch.empty(int(nws * 1.25), dtype=torch.uint8, device=self.device)
        # the compact table (and its item -> category map) changes every step: clear the use
        # counters and rebuild the category -> items index for it; P = 1 (owners apply the decay)
        self._state[:4].view(torch.float32).fill_(1.0)
        L.check(self.lib.tlsan_state_reindex(C.byref(dims), C.byref(cp), self._state.data_ptr(), self._stream()),
                "tlsan_state_reindex")
        nr = self.lib.tlsan_rows_apply_workspace(max(self.router.R, self.C), max(n_rows_max, 1))
        if self._rws is None or self._rws.numel() < nr:
            self._rws = torch.empty(int(nr * 1.25), dtype=torch.uint8, device=self.device)

    def _compact(self, db):
        """Route: distinct rows the batch touches -> compact per-step table + remapped batch."""
        B, Ls, Sn = db.B, self.Ls, db.Sn
        plan = self.router.plan(db.keys, self._scan)
        table = self.router.fetch(plan, self.shard)              # [n, W], rows in key order
        comp = plan["prefix"][db.keys]                            # compact row of every id of the batch
        o = 0
        i_c = comp[o:o + B]; o += B
        hist_c = comp[o:o + B * Ls]; o += B * Ls
        new_c = comp[o:o + B * Sn] if Sn > 0 else db.hist_i_new; o += B * Sn
        j_c = None
        if db.j is not None:
            j_c = comp[o:o + B]; o += B
        u_c = comp[o:o + B]
        cate_c = self.cate_by_key[plan["uniq"].long()]
        keep = (table, comp, cate_c)
        ptr = lambda t: None if t is None else t.data_ptr()
        cb = L.Batch(B, Sn, ptr(u_c), ptr(i_c), ptr(j_c), ptr(db.y), ptr(hist_c), ptr(new_c), ptr(db.hist_t),
                     ptr(db.sl), ptr(db.sl_new), ptr(db.u_cate))
        base = table.data_ptr()
        cp = L.Params(base, base + 4 * self.di, base, base + 4 * self.di, self.cate_emb.data_ptr(),
                      self.dense.data_ptr(), self.dense_KT.data_ptr(), cate_c.data_ptr(),
                      self.W, self.W, self.W, self.W, None)
        n = max(plan["n"], 1)
        dims = L.Dims(n, n, self.C, self.d, self.di, self.dc, self.H, self.Ls)
        return dims, cp, cb, plan, keep

    # ------------------------------------------------------------------ training
    def train_async(self, batch, lr):
        db = self.device_batch(batch)
        G = self.world
        dims, cp, cb, plan, keep = self._compact(db)
        n, Cc, di, Ls, W = dims.item_count, self.C, self.di, self.Ls, self.W
        self._buffers(dims, cp, db.B, db.Sn, max(int(plan["recv_rows"].numel()), Cc))
        dev = self.device
        g_item = torch.empty(n, di, dtype=torch.float32, device=dev)
        g_itemb = torch.empty(n, dtype=torch.float32, device=dev)
        g_user = torch.empty(n, di, dtype=torch.float32, device=dev)
        g_usert = torch.empty(n, Ls, dtype=torch.float32, device=dev)
        n_dense = self.lay.n_dense
        flat = torch.zeros(n_dense + Cc * self.dc + 4, dtype=torch.float32, device=dev)
        go = L.GradsOut(g_item.data_ptr(), g_itemb.data_ptr(), g_user.data_ptr(), g_usert.data_ptr(),
                        flat.data_ptr() + 4 * n_dense, flat.data_ptr())
        out = L.StepOut(self._out.data_ptr(), self._out.data_ptr() + 4, None, self._out.data_ptr() + 8)
        hp = L.HParams(float(lr), 0.0, self.clip, L.NORM_TF18, L.L2_DENSE)   # reg is applied by the owners
        st = self._stream()
        L.check(self.lib.tlsan_grads(C.byref(dims), C.byref(cp), C.byref(cb), C.byref(hp), C.byref(go), C.byref(out),
                                     self._state.data_ptr(), self._ws.data_ptr(), self._ws.numel(), st), "tlsan_grads")
        # ---- one all-reduce: dense grads | cate grads | loss | per-use squares | local table squares
        tail = flat[n_dense + Cc * self.dc:]
        tail[0] = self._out[0]
        tail[1] = self._out[2]
        tail[2] = self.S_local[0].float()
        if G > 1:
            allreduce_sum(flat, self.group)
        inv_g = 1.0 / G
        gd = flat[:n_dense] * inv_g
        S_tot = tail[2].double() + self.S_cate[0]
        sq = (tail[1].double() * (inv_g * inv_g) + (self.reg * self.reg) * S_tot + gd.double().pow(2).sum())
        norm = sq.sqrt().float()
        coef = self.clip / torch.clamp(norm, min=self.clip)          # clip_by_global_norm (model.py:201)
        self._step_dev.copy_((coef * float(lr)).reshape(1))
        self.last_gnorm.copy_(norm.reshape(1))
        self.last_loss.copy_((tail[0] * inv_g + self.reg * 0.5 * S_tot.float()).reshape(1))
        # ---- per-row gradients in the fused layout (rows of the other kind are exactly zero)
        gf = torch.zeros(n, W, dtype=torch.float32, device=dev)
        gf[:, :di] = g_item + g_user
        gf[:, di:di + Ls] = g_usert
        gf[:, di] += g_itemb
        rows32, vals = self.router.push(plan, gf)
        # owners: deterministic scatter-apply with dense L2 decay of every local row; destinations
        # outside a view's range (the other table's rows) are ignored by tlsan_rows_apply
        self._rows_apply(self.shard[:self.cI], di, vals, rows32, inv_g, 0)
        self._rows_apply(self.shard[self.cI:], di + Ls, vals, rows32 - self.cI, inv_g, 1)
        g_cate = flat[n_dense:n_dense + Cc * self.dc].view(Cc, self.dc)
        self._rows_apply(self.cate_emb, self.dc, g_cate, self._cate_ids, inv_g, 2)
        self.S_local = (self._sq[0] + self._sq[1]).reshape(1).clone()
        self.S_cate = self._sq[2].reshape(1).clone()
        # dense attention weights: replicated, identical update on every rank
        self.dense.sub_(gd * self._step_dev)
        L.check(self.lib.tlsan_sync_derived(C.byref(self.dims_full), C.byref(cp), st), "tlsan_sync_derived")
        self._step += 1
        self._keep = (keep, flat, gf, vals, rows32, g_item, g_itemb, g_user, g_usert)
        return db

    def _rows_apply(self, Wt, reg_cols, vals, rows32, gscale, slot):
        n = int(rows32.numel())
        rows32 = rows32.contiguous()
        vals = vals.contiguous()
        L.check(self.lib.tlsan_rows_apply(Wt.data_ptr(), Wt.stride(0), Wt.shape[0], Wt.shape[1], reg_cols,
                                          vals.data_ptr() if n else None, vals.stride(0) if n else Wt.shape[1],
                                          rows32.data_ptr() if n else None, n, float(gscale),
                                          self._step_dev.data_ptr(), self.reg, self._sq.data_ptr() + 8 * slot,
                                          self._rws.data_ptr(), self._rws.numel(), self._stream()), "tlsan_rows_apply")
        self._keep_rows = getattr(self, "_keep_rows", [])[-6:] + [(rows32, vals)]

    def train(self, sess, batch, lr, add_summary=False):
        self.train_async(batch, lr)
        return float(self.last_loss.item())

    # ------------------------------------------------------------------ evaluation
    def forward(self, batch, is_test=True):
        db = self.device_batch(batch, is_test)
        dims, cp, cb, _, keep = self._compact(db)
        li = torch.empty(db.B, dtype=torch.float32, device=self.device)
        lj = torch.empty(db.B, dtype=torch.float32, device=self.device) if db.j is not None else None
        L.check(self.lib.tlsan_forward(C.byref(dims), C.byref(cp), C.byref(cb), li.data_ptr(),
                                       None if lj is None else lj.data_ptr(), None, None, 0, self._stream()),
                "tlsan_forward")
        torch.cuda.current_stream(self.device).synchronize()   # `keep` tensors stay alive until done
        return li, lj

    def eval_auc(self, sess, batch):
        li, lj = self.forward(batch, is_test=True)
        return float(((li - lj) > 0).float().mean().item())

    # ------------------------------------------------------------------ inspection
    def _table_views(self):
        return self.shard[:self.cI], self.shard[self.cI:]

    def gather_params(self):
        """Full (un-sharded) parameters on every rank, as numpy (tests / checkpoints)."""
        host = self.shard.cpu()
        if self.world > 1:
            outs = [torch.empty_like(host) for _ in range(self.world)]
            dist.all_gather(outs, host, group=self.group)
        else:
            outs = [host]
        I, U, G, di, Ls = self.I, self.U, self.world, self.di, self.Ls
        item = np.zeros((I, self.W), np.float32)
        user = np.zeros((U, self.W), np.float32)
        for r in range(G):
            t = outs[r].numpy()
            ni, nu = len(range(r, I, G)), len(range(r, U, G))
            item[r::G] = t[:ni]
            user[r::G] = t[self.cI:self.cI + nu]
        out = dict(item_emb=item[:, :di].copy(), item_b=item[:, di].copy(),
                   user_emb=user[:, :di].copy(), usert_emb=user[:, di:di + Ls].copy(),
                   cate_emb=self.cate_emb.cpu().numpy())
        flat = self.dense.cpu().numpy()
        d, dh, lay = self.d, self.d // self.H, self.lay
        for k, off, shape in (("fwa1_W1", lay.f1_W1, (dh, dh)), ("fwa1_b1", lay.f1_b1, (dh,)),
                              ("fwa1_W2", lay.f1_W2, (dh, dh)), ("fwa1_b2", lay.f1_b2, (dh,)),
                              ("dense_K", lay.K, (d, d)), ("dense_b", lay.k0, (d,)),
                              ("fwa2_W1", lay.f2_W1, (dh, dh)), ("fwa2_b1", lay.f2_b1, (dh,)),
                              ("fwa2_W2", lay.f2_W2, (dh, dh)), ("fwa2_b2", lay.f2_b2, (dh,)), ("gamma", lay.gamma, ())):
            n = int(np.prod(shape)) if shape else 1
            out[k] = flat[off:off + n].reshape(shape).copy()
        return out

    def set_params(self, p):
        """Load full parameters (dict of numpy arrays); every rank keeps its own rows."""
        G, r, di, Ls = self.world, self.rank, self.di, self.Ls
        gi = np.arange(r, self.I, G)
        gu = np.arange(r, self.U, G)
        t = np.zeros(tuple(self.shard.shape), np.float32)
        t[:len(gi), :di] = np.asarray(p["item_emb"], np.float32)[gi]
        t[:len(gi), di] = np.asarray(p["item_b"], np.float32)[gi]
        t[self.cI:self.cI + len(gu), :di] = np.asarray(p["user_emb"], np.float32)[gu]
        t[self.cI:self.cI + len(gu), di:di + Ls] = np.asarray(p["usert_emb"], np.float32)[gu]
        self.shard.copy_(torch.as_tensor(t))
        self.cate_emb.copy_(torch.as_tensor(np.asarray(p["cate_emb"], np.float32)))
        self._pack_dense(p)
        it, us = self._table_views()
        # running sums of squares of the regularised tables (tf.nn.l2_loss terms, model.py:164-169)
        self.S_local = (it[:, :di].double().pow(2).sum() + us[:, :di + Ls].double().pow(2).sum()).reshape(1)
        self.S_cate = self.cate_emb.double().pow(2).sum().reshape(1)
