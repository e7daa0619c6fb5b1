"""Multi-GPU TLSAN step: one process per GPU, embedding tables row-sharded, RCCL over xGMI.

The reference is single-GPU (train.py:53,146); this is the MI355X-native scale-out of
SURVEY.md section 8e.  Samples are independent units: every rank trains its own batch
(data parallel, weak scaling).  State placement:

  * ``user_emb`` + ``usert_emb`` rows: sharded by ``user_id % world``  (fused rows
    ``[user_emb | usert_emb | pad]``);
  * ``item_emb`` + ``item_b`` rows: sharded by ``item_id % world`` (fused ``[item_emb | item_b | pad]``;
    mod-G spreads the Zipf-hot items);
  * ``cate_emb`` (<= 5 MB), ``item_cate_list`` and the dense attention weights: replicated.

One step = (1) de-duplicate the ids the local batch touches, (2) all-to-all ids -> owners,
(3) all-to-all rows back into a compact per-step table the HIP kernels run on in place (row
strides, tlsan_params.ld_*), (4) fused forward/backward + exact per-row gradient sums
(tlsan_grads), (5) ONE all-reduce of [dense grads | cate grads | loss | norm terms],
(6) all-to-all of the per-row gradients back to the owners, (7) owners apply them with the
deterministic tlsan_rows_apply (dense L2 decay of every local row, as the reference).

``RowExchange`` is device-agnostic torch + torch.distributed plumbing (runs on CPU/gloo in the
tests); all arithmetic on rows is in libtlsan_hip.so.
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
        self.di, self.dc, self.Ls = config["itemid_embedding_size"], config["cateid_embedding_size"], config["Ls"]
        self.d, self.H = config["hidden_units"], config["num_heads"]
        self.WI = self.di + 4                    # [item_emb | item_b | pad x3]
        self.WU = _ru4(self.di + self.Ls)        # [user_emb | usert_emb | pad]
        self.part_i, self.part_u = ModPartition(I, self.world), ModPartition(U, self.world)
        self.xi, self.xu = RowExchange(self.part_i, group), RowExchange(self.part_u, group)
        p = Model.init_params(config, seed)      # identical on every rank (numpy, seeded)
        dev = self.device
        gi = np.arange(self.rank, I, self.world)
        gu = np.arange(self.rank, U, self.world)
        item = np.zeros((len(gi), self.WI), np.float32)
        item[:, :self.di] = p["item_emb"][gi]
        item[:, self.di] = p["item_b"][gi]
        user = np.zeros((len(gu), self.WU), np.float32)
        user[:, :self.di] = p["user_emb"][gu]
        user[:, self.di:self.di + self.Ls] = p["usert_emb"][gu]
        self.item_shard = torch.as_tensor(item).to(dev)
        self.user_shard = torch.as_tensor(user).to(dev)
        self.cate_emb = torch.as_tensor(p["cate_emb"]).to(dev)
        self.item_cate = torch.as_tensor(np.asarray(item_cate_list, np.int32)).to(dev)
        # dims of the full problem only to obtain the dense layout
        dims_full = L.Dims(U, I, Cc, self.d, self.di, self.dc, self.H, self.Ls)
        self.lay = L.DenseLayout()
        L.check(self.lib.tlsan_dense_layout_of(C.byref(dims_full), C.byref(self.lay)), "tlsan_dense_layout_of")
        self.dense = torch.zeros(self.lay.n_dense, dtype=torch.float32, device=dev)
        self.dense_KT = torch.zeros(self.d, self.d, dtype=torch.float32, device=dev)
        self._pack_dense(p)
        self.reg = float(config["regulation_rate"])
        self.clip = float(config["max_gradient_norm"])
        # running sums of squares of the regularised tables (tf.nn.l2_loss terms, model.py:164-169)
        self.S_local = (self.item_shard[:, :self.di].double().pow(2).sum()
                        + self.user_shard[:, :self.di + self.Ls].double().pow(2).sum()).reshape(1)
        self.S_cate = self.cate_emb.double().pow(2).sum().reshape(1)
        self._sq = torch.zeros(3, dtype=torch.float64, device=dev)   # rows_apply sumsq outputs
        self._out = torch.zeros(4, dtype=torch.float32, device=dev)  # loss, gnorm, sq_rows (local)
        self._step_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self.last_loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.last_gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._state = None
        self._ws = None
        self._rws = None
        self._step = 0
        self._epoch = 0
        self.global_step = _Var(lambda: self._step)
        self.global_epoch_step = _Var(lambda: self._epoch)
        self.train_writer, self.eval_writer = _Writer("train"), _Writer("eval")
        self._cate_ids = torch.arange(Cc, dtype=torch.int32, device=dev)

    # ------------------------------------------------------------------ helpers
    def _pack_dense(self, p):
        d, dh = self.d, self.d // self.H
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

    def device_batch(self, batch, is_test=False):
        return batch if isinstance(batch, DeviceBatch) else DeviceBatch(batch, self.device, is_test, self.Ls)

    def _buffers(self, dims, cp, B, Sn, n_rows_max):
        nst = self.lib.tlsan_state_bytes(C.byref(dims))
        nws = self.lib.tlsan_workspace_bytes(C.byref(dims), B, Sn)
        if nst == 0 or nws == 0:
            raise L.TlsanError(self.lib.tlsan_last_error().decode())
        if self._state is None or self._state.numel() < nst:
            self._state = torch.zeros(int(nst * 1.5), dtype=torch.uint8, device=self.device)
        # the compact item table (and its item -> category map) changes every step: clear the
        # use counters and rebuild the category -> items index for it
        L.check(self.lib.tlsan_state_reindex(C.byref(dims), C.byref(cp), self._state.data_ptr(), self._stream()),
                "tlsan_state_reindex")
        if self._ws is None or self._ws.numel() < nws:
            self._ws = torch.empty(int(nws * 1.25), dtype=torch.uint8, device=self.device)
        nr = self.lib.tlsan_rows_apply_workspace(max(self.item_shard.shape[0], self.user_shard.shape[0],
                                                     self.cate_emb.shape[0]), n_rows_max)
        if self._rws is None or self._rws.numel() < nr:
            self._rws = torch.empty(int(nr * 1.25), dtype=torch.uint8, device=self.device)

    def _compact(self, db, with_j):
        """De-duplicate ids, fetch their rows, build the compact batch + parameter structs."""
        B, Ls, Sn = db.B, self.Ls, db.Sn
        parts = [db.i.long(), db.hist_i.reshape(-1).long()]
        if Sn > 0:
            parts.append(db.hist_i_new.reshape(-1).long())
        if with_j and db.j is not None:
            parts.append(db.j.long())
        uniq_i, inv_i = torch.unique(torch.cat(parts), sorted=True, return_inverse=True)
        uniq_u, inv_u = torch.unique(db.u.long(), sorted=True, return_inverse=True)
        plan_i, plan_u = self.xi.plan(uniq_i), self.xu.plan(uniq_u)
        item_c = self.xi.fetch(plan_i, self.item_shard)     # [n_i, WI]
        user_c = self.xu.fetch(plan_u, self.user_shard)     # [n_u, WU]
        inv_i = inv_i.int()
        o = 0
        i_c = inv_i[o:o + B].contiguous(); o += B
        hist_c = inv_i[o:o + B * Ls].contiguous(); o += B * Ls
        new_c = inv_i[o:o + B * Sn].contiguous() if Sn > 0 else db.hist_i_new; o += B * Sn
        j_c = inv_i[o:o + B].contiguous() if (with_j and db.j is not None) else None
        u_c = inv_u.int().contiguous()
        cate_c = self.item_cate[uniq_i].contiguous()
        keep = (item_c, user_c, i_c, hist_c, new_c, j_c, u_c, cate_c)
        ptr = lambda t: None if t is None else t.data_ptr()
        cb = L.Batch(B, Sn, ptr(u_c), ptr(i_c), ptr(j_c), ptr(db.y), ptr(hist_c), ptr(new_c), ptr(db.hist_t),
                     ptr(db.sl), ptr(db.sl_new), ptr(db.u_cate))
        cp = L.Params(item_c.data_ptr(), item_c.data_ptr() + 4 * self.di, user_c.data_ptr(),
                      user_c.data_ptr() + 4 * self.di, self.cate_emb.data_ptr(), self.dense.data_ptr(),
                      self.dense_KT.data_ptr(), cate_c.data_ptr(), self.WI, self.WI, self.WU, self.WU, None)
        dims = L.Dims(int(uniq_u.numel()), int(uniq_i.numel()), self.config["cate_count"], self.d, self.di, self.dc,
                      self.H, self.Ls)
        return dims, cp, cb, plan_i, plan_u, keep

    # ------------------------------------------------------------------ training
    def train_async(self, batch, lr):
        db = self.device_batch(batch)
        G = self.world
        dims, cp, cb, plan_i, plan_u, keep = self._compact(db, with_j=False)
        n_i, n_u, Cc = dims.item_count, dims.user_count, dims.cate_count
        self._buffers(dims, cp, db.B, db.Sn, max(sum(plan_i.recv_counts), sum(plan_u.recv_counts), Cc))
        dev = self.device
        g_item = torch.empty(n_i, self.di, dtype=torch.float32, device=dev)
        g_itemb = torch.empty(n_i, dtype=torch.float32, device=dev)
        g_user = torch.empty(n_u, self.di, dtype=torch.float32, device=dev)
        g_usert = torch.empty(n_u, self.Ls, dtype=torch.float32, device=dev)
        n_dense = self.lay.n_dense
        flat = torch.zeros(n_dense + Cc * self.dc + 4, dtype=torch.float32, device=dev)
        go = L.GradsOut(g_item.data_ptr(), g_itemb.data_ptr(), g_user.data_ptr(), g_usert.data_ptr(),
                        flat.data_ptr() + 4 * n_dense, flat.data_ptr())
        out = L.StepOut(self._out.data_ptr(), self._out.data_ptr() + 4, None, self._out.data_ptr() + 8)
        hp = L.HParams(float(lr), 0.0, self.clip, L.NORM_TF18, L.L2_DENSE)   # reg applied by the owners
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
        # ---- route the per-row gradients to their owners and apply (deterministic, dense L2)
        gi_f = torch.zeros(n_i, self.WI, dtype=torch.float32, device=dev)
        gi_f[:, :self.di] = g_item
        gi_f[:, self.di] = g_itemb
        gu_f = torch.zeros(n_u, self.WU, dtype=torch.float32, device=dev)
        gu_f[:, :self.di] = g_user
        gu_f[:, self.di:self.di + self.Ls] = g_usert
        rows_i, vals_i = self.xi.push(plan_i, gi_f)
        rows_u, vals_u = self.xu.push(plan_u, gu_f)
        self._rows_apply(self.item_shard, self.WI, self.di, vals_i, rows_i, inv_g, 0)
        self._rows_apply(self.user_shard, self.WU, self.di + self.Ls, vals_u, rows_u, inv_g, 1)
        g_cate = flat[n_dense:n_dense + Cc * self.dc].view(Cc, self.dc)
        self._rows_apply(self.cate_emb, self.dc, self.dc, g_cate, self._cate_ids, inv_g, 2)
        self.S_local = (self._sq[0] + self._sq[1]).reshape(1).clone()
        self.S_cate = self._sq[2].reshape(1).clone()
        # dense attention weights: replicated, identical update on every rank
        self.dense.sub_(gd * self._step_dev)
        dims_full = L.Dims(self.config["user_count"], self.config["item_count"], Cc, self.d, self.di, self.dc, self.H, self.Ls)
        L.check(self.lib.tlsan_sync_derived(C.byref(dims_full), C.byref(cp), st), "tlsan_sync_derived")
        self._step += 1
        self._keep = (keep, flat, gi_f, gu_f, vals_i, vals_u, rows_i, rows_u, g_item, g_itemb, g_user, g_usert)
        return db

    def _rows_apply(self, W, width, reg_cols, vals, rows, gscale, slot):
        n = int(rows.numel())
        rows32 = rows.int().contiguous() if rows.dtype != torch.int32 else rows
        vals = vals.contiguous()
        L.check(self.lib.tlsan_rows_apply(W.data_ptr(), W.shape[1], W.shape[0], width, reg_cols,
                                          vals.data_ptr() if n else None, vals.shape[1] if n else width,
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
        dims, cp, cb, _, _, keep = self._compact(db, with_j=True)
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
    def gather_params(self):
        """Full (un-sharded) parameters on every rank, as numpy (tests / checkpoints)."""
        def allgather_rows(shard, part):
            n_max = part.local_count(0)
            pad = torch.zeros(n_max, shard.shape[1], dtype=shard.dtype, device=shard.device)
            pad[:shard.shape[0]] = shard
            if self.world > 1:
                host = pad.cpu()
                outs = [torch.empty_like(host) for _ in range(self.world)]
                dist.all_gather(outs, host, group=self.group)
                outs = [o.to(shard.device) for o in outs]
            else:
                outs = [pad]
            full = torch.zeros(part.n, shard.shape[1], dtype=shard.dtype, device=shard.device)
            for r in range(self.world):
                full[r::self.world] = outs[r][:part.local_count(r)]
            return full.cpu().numpy()
        item = allgather_rows(self.item_shard, self.part_i)
        user = allgather_rows(self.user_shard, self.part_u)
        out = dict(item_emb=item[:, :self.di].copy(), item_b=item[:, self.di].copy(),
                   user_emb=user[:, :self.di].copy(), usert_emb=user[:, self.di:self.di + self.Ls].copy(),
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
        gi = np.arange(self.rank, self.part_i.n, self.world)
        gu = np.arange(self.rank, self.part_u.n, self.world)
        item = np.zeros(tuple(self.item_shard.shape), np.float32)
        item[:, :self.di] = np.asarray(p["item_emb"], np.float32)[gi]
        item[:, self.di] = np.asarray(p["item_b"], np.float32)[gi]
        user = np.zeros(tuple(self.user_shard.shape), np.float32)
        user[:, :self.di] = np.asarray(p["user_emb"], np.float32)[gu]
        user[:, self.di:self.di + self.Ls] = np.asarray(p["usert_emb"], np.float32)[gu]
        self.item_shard.copy_(torch.as_tensor(item))
        self.user_shard.copy_(torch.as_tensor(user))
        self.cate_emb.copy_(torch.as_tensor(np.asarray(p["cate_emb"], np.float32)))
        self._pack_dense(p)
        self.S_local = (self.item_shard[:, :self.di].double().pow(2).sum()
                        + self.user_shard[:, :self.di + self.Ls].double().pow(2).sum()).reshape(1)
        self.S_cate = self.cate_emb.double().pow(2).sum().reshape(1)
