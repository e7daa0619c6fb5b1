"""GPU tests at the shapes BASELINE.json's `configs` name, all through the C ABI:

  C3  Electronics (39991 / 22048 / 673), d = 128, batch 4096 -- the bench's own tables and inputs through one forward
      pass and one train step (lazy and dense L2; fp32 and bf16 table storage) against the fp64 oracle AT SIZE;

  C2  Digital-Music (1659 / 1583 / 53), d = 128 (64/64/64), fp32, batch 1024 -- the real fixture batch
      `DataInput_bs1024_k10_b0` (captured from the reference's input.py) through one train step and an
      evaluation, against the fp64 oracle;
  C4  Movies-TV sizes (35896 / 28589 / 15), window of 90 positions: one train step at batch 4096 against the fp64
      oracle AT SIZE (round 6, id compaction); tables row-sharded over 2 ranks (two processes on cuda:0, gloo):
      bitwise determinism and sharded == single-GPU `Model` on the concatenated batch;
  C5  10 M users / 5 M items / 10 k categories, d = 256 (128/128/128), window 90, batch 4096 on one GPU: one train step
      against the fp64 oracle AT SIZE (round 6: the oracle steps the rows the batch touches, the other rows enter through
      their sum of squares); bitwise determinism, lazy L2 == dense L2 (the reference's update) on every row,
      untouched rows not written, the item_b checksum.
Plus the driver's resume flow (train.py:71-76,124-127) and the captured-graph workspace.
"""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import tlsan_oracle as orc
from tests.helpers import fixture_batch, make_config

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-4
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# ------------------------------------------------------------------------------------------- C3
@pytest.mark.parametrize("table_dtype", ["f32", "bf16"])
def test_c3_electronics_one_step_matches_oracle(table_dtype):
    """BASELINE.json configs[2] at its own size -- Electronics (39991 / 22048 / 673), d = 128 (64/64/64), batch 4096, the
    bench's synthetic inputs -- against the fp64 oracle (TLSAN/model.py:56-205 restated; one oracle step at this size is
    ~8 s of numpy): logits and u_t of the forward pass, then one train step in lazy and in dense L2 mode -- loss, clip
    norm, every parameter.  Second case: the precision the config names, bf16 table storage -- the oracle runs on the
    tables as stored (rounded to bf16), fp32 parameters are held to the same bound, and every stored bf16 element must
    be a bf16 neighbour of the oracle's exact update (stochastic rounding on the write-back)."""
    from tlsan_amd import synth
    from tlsan_amd.model import Model
    cfg = synth.make_config("electronics")
    icl = synth.item_cate_list(cfg)
    batch = synth.make_batches(cfg, 1, 4096, seed=1234)[0]
    p = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in orc.init_params(cfg, seed=1234, dtype=np.float32).items()}
    bf = ("item_emb", "user_emb", "cate_emb")
    if table_dtype == "bf16":
        for k in bf:
            u32 = np.asarray(p[k], np.float32).view(np.uint32).astype(np.uint64)
            u32 = ((u32 + 0x7FFF + ((u32 >> 16) & 1)) >> 16) << 16          # round to nearest even
            p[k] = u32.astype(np.uint32).view(np.float32).reshape(p[k].shape).astype(np.float64)
    b = orc.as_batch(batch)
    ref = orc.forward(p, icl, b, 8)
    loss, newp, info = orc.train_step(p, icl, b, 8, cfg["regulation_rate"], lr=1.0)
    for l2 in ("lazy", "dense"):
        m = Model(cfg, icl, l2_mode=l2, table_dtype=table_dtype)
        m.set_params({k: np.asarray(v, np.float32) for k, v in p.items()})
        li, _, ut, _ = m.forward(batch, is_test=False, want_u_t=True)
        assert np.abs(li.cpu().numpy() - ref["logits"]).max() < LOGIT_TOL, l2
        assert np.abs(ut.cpu().numpy() - ref["u_t"]).max() < LOGIT_TOL, l2
        got_loss = m.train(None, batch, 1.0)
        assert abs(got_loss - loss) < 1e-4 * max(1.0, abs(loss)), (l2, got_loss, loss)
        assert abs(m.last_gnorm() - info["norm"]) < 3e-4 * info["norm"], l2
        got = m.get_params()
        for k in newp:
            a, r = np.asarray(got[k], np.float64).reshape(p[k].shape), newp[k]
            if table_dtype == "bf16" and k in bf:
                # (the ulp of the larger of the two: an element next to a power of two, or one that the second rounding of the
                #  lazy read -- the table scale folded in -- carries across one, has neighbours in two binades)
                ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.maximum(np.abs(r), np.abs(a)), 1e-30))) - 7)
                nround = 2 if l2 == "lazy" else 1     # (lazy: reading the parameters folds the table scale in, a second rounding)
                dev = np.abs(a - r) / ulp
                assert dev.max() <= nround + 1e-3, (l2, k, dev.max())
                assert abs(((a - r) / ulp).mean()) < 0.02, (l2, k)       # unbiased
            else:
                du, dr = a - p[k], r - p[k]
                assert np.abs(du - dr).max() < 3e-4 * (np.abs(dr).max() + 1e-9) + 2e-7, (l2, k)


# ------------------------------------------------------------------------------------------- C2
def test_c2_digital_music_batch_1024_d128():
    from tlsan_amd.model import Model
    batch, (U, I, C), cat = fixture_batch("digital_music", "DataInput", 1024, 10, 0)
    assert len(batch[0]) == 1024
    cfg = make_config(U=U, I=I, C=C, d=128)                    # 64 / 64 / 64, the reference's other defaults
    p = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in orc.init_params(cfg, seed=1234, dtype=np.float32).items()}
    b = orc.as_batch(batch)
    ref = orc.forward(p, cat, b, 8)
    loss, newp, info = orc.train_step(p, cat, b, 8, cfg["regulation_rate"], lr=1.0)
    for l2 in ("dense", "lazy"):
        m = Model(cfg, cat, l2_mode=l2)
        m.set_params({k: np.asarray(v, np.float32) for k, v in p.items()})
        li, _, ut, _ = m.forward(batch, is_test=False, want_u_t=True)
        assert np.abs(li.cpu().numpy() - ref["logits"]).max() < LOGIT_TOL
        assert np.abs(ut.cpu().numpy() - ref["u_t"]).max() < LOGIT_TOL
        got_loss = m.train(None, batch, 1.0)
        assert abs(got_loss - loss) < 1e-4 * max(1.0, abs(loss)), l2
        assert abs(m.last_gnorm() - info["norm"]) < 3e-4 * info["norm"], l2
        got = m.get_params()
        for k in newp:
            du = np.asarray(got[k], np.float64).reshape(p[k].shape) - p[k]
            dr = newp[k] - p[k]
            assert np.abs(du - dr).max() < 3e-4 * (np.abs(dr).max() + 1e-9) + 2e-7, (l2, k)
    # evaluation on the test fixture batch of the same dataset (test batch 128, train.py:45) with the trained weights
    tb, _, _ = fixture_batch("digital_music", "DataInputTest", 128, 10, 0)
    tbo = orc.as_batch(tb, is_test=True)
    q = {k: np.asarray(v, np.float64) for k, v in got.items()}
    ri = orc.forward(q, cat, dict(tbo, y=np.zeros(128)), 8)["logits"]
    rj = orc.forward(q, cat, dict(tbo, i=tbo["j"], y=np.zeros(128)), 8)["logits"]
    li, lj, _, _ = m.forward(tb, is_test=True)
    assert np.abs(li.cpu().numpy() - ri).max() < LOGIT_TOL and np.abs(lj.cpu().numpy() - rj).max() < LOGIT_TOL
    clear = np.abs(ri - rj) > 2e-4
    auc = m.eval_auc(None, tb)
    assert abs(auc - float(((ri - rj) > 0).mean())) <= (~clear).sum() / 128.0 + 1e-9


# ------------------------------------------------------------------------------------------- C5
def test_c5_ten_million_users_d256_window_90():
    from tlsan_amd import synth
    from tlsan_amd.model import Model
    cfg = synth.make_config("electronics", Ls=90, hidden_units=256, itemid_embedding_size=128, userid_embedding_size=128,
                            cateid_embedding_size=128, user_count=10_000_000, item_count=5_000_000, cate_count=10_000)
    icl = synth.item_cate_list(cfg)
    batches = synth.make_batches(cfg, 2, 4096, seed=55)
    touched_u = np.unique(np.concatenate([np.asarray(b[0]) for b in batches]))
    runs = {}
    for key, mode in (("lazy", "lazy"), ("lazy2", "lazy"), ("dense", "dense")):
        m = Model(cfg, icl, l2_mode=mode, init="device", seed=7)
        if key == "lazy":
            before = m.user_emb[torch.as_tensor(touched_u, device=m.device)].clone()
            probe = torch.arange(0, cfg["user_count"], 9973, device=m.device)
            probe = probe[~torch.isin(probe, torch.as_tensor(touched_u, device=m.device))]
            before_probe = m.user_emb[probe].clone()
        losses = [m.train(None, b, 1.0) for b in batches]
        norm = m.last_gnorm()
        if key == "lazy":    # (before the fold: in lazy mode only rows that received a gradient are written)
            assert torch.equal(m.user_emb[probe], before_probe)
            assert not torch.equal(m.user_emb[torch.as_tensor(touched_u, device=m.device)], before)
        m.fold_scale()
        torch.cuda.synchronize()
        runs[key] = dict(m=m, losses=losses, norm=norm)
    a, b2, dn = runs["lazy"], runs["lazy2"], runs["dense"]
    assert a["losses"] == b2["losses"] and a["norm"] == b2["norm"]
    for k in ("item_emb", "item_b", "user_emb", "usert_emb", "cate_emb", "dense"):
        ta, tb, td = getattr(a["m"], k), getattr(b2["m"], k), getattr(dn["m"], k)
        assert torch.equal(ta, tb), k                                              # bitwise reproducible
        scale = float(td.abs().max().item())
        assert float((ta - td).abs().max().item()) <= 3e-6 * scale + 1e-9, k       # lazy == the reference's dense update
    assert np.allclose(a["losses"], dn["losses"], rtol=3e-6, atol=0)
    assert abs(a["norm"] - dn["norm"]) <= 2e-5 * dn["norm"]
    assert np.isfinite(a["losses"]).all() and a["losses"][0] > 100.0               # the L2 term of 10^7 usert rows at -1
    # checksum: item_b moves by -lr * coef * sum_b d loss / d logit_b
    del runs, b2, dn
    m = a["m"]
    bt = batches[0]
    li, _, _, _ = m.forward(bt, is_test=False)
    logit = li.cpu().numpy().astype(np.float64)
    dl_sum = ((1.0 / (1.0 + np.exp(-logit))) - np.asarray(bt[2], np.float64)).sum() / len(logit)
    ib0 = m.item_b.double().sum().item()
    m.train(None, bt, 1.0)
    coef = min(1.0, cfg["max_gradient_norm"] / m.last_gnorm())
    ib1 = m.item_b.double().sum().item()
    assert abs((ib0 - ib1) - coef * dl_sum) < 1e-5 * max(1.0, abs(dl_sum)) + 1e-6


# ------------------------------------------------------------------------------------------- C4 / C5 against the oracle, at size
def _sumsq64(t, chunk=1 << 26):
    """sum of squares of a device tensor in fp64 without a full-size fp64 copy (10^7-row tables)"""
    flat, tot = t.reshape(-1), 0.0
    for lo in range(0, flat.numel(), chunk):
        tot += float(flat[lo:lo + chunk].double().pow(2).sum().item())
    return tot


def _one_step_against_oracle_at_size(cfg, icl, batch, seed, n_probe=4096):
    """One train step of a FULL-SIZE device model against the fp64 oracle (TLSAN/model.py:56-205 restated) by id compaction
    (tests/helpers.compact_problem; pinned on a case numpy can hold by tests/test_oracle.py): the rows the batch touches
    are extracted from the device tables into small tables with renumbered ids, the rest of the regularised tables enters
    through its sum of squares (chunked fp64 on the device) -- the L2 term of the loss and of clip_by_global_norm's norm --
    and the oracle steps that problem.  Compared: logits and u_t of the forward pass, loss, clip norm, EVERY touched row
    of the five tables, every dense parameter, and a sample of untouched rows, which must have decayed by the dense L2
    gradient (lazy L2: after the table scale is folded in).  Lazy and dense L2 from the same initial values."""
    from tests.helpers import compact_problem
    from tlsan_amd.model import Model
    cp = compact_problem(batch, icl)
    sel = dict(item_emb="items", item_b="items", user_emb="users", usert_emb="users", cate_emb="cates")
    counts = dict(items=cfg["item_count"], users=cfg["user_count"], cates=cfg["cate_count"])
    rng = np.random.default_rng(seed)
    probe = {}
    for s_, n in counts.items():     # rows the batch does not touch
        cand = np.unique(rng.integers(0, n, min(n_probe, n)))
        probe[s_] = cand[~np.isin(cand, cp[s_])]
    lr, reg, clip = 1.0, cfg["regulation_rate"], cfg["max_gradient_norm"]
    ref = q0 = None
    for l2 in ("lazy", "dense"):
        m = Model(cfg, icl, l2_mode=l2, init="device", seed=seed)
        ids = {s_: torch.as_tensor(cp[s_], device=m.device) for s_ in counts}
        pids = {s_: torch.as_tensor(probe[s_], device=m.device) for s_ in counts}
        q = {k: getattr(m, k)[ids[s_]].double().cpu().numpy() for k, s_ in sel.items()}
        q.update({k: np.asarray(v, np.float64) for k, v in m.unpack_dense(m.dense.cpu().numpy()).items()})
        old_probe = {k: getattr(m, k)[pids[s_]].double().cpu().numpy() for k, s_ in sel.items()}
        if ref is None:
            q0 = q
            extra = sum(_sumsq64(getattr(m, k)) - float((q[k] ** 2).sum()) for k in orc.REG_TABLES)
            ob = orc.as_batch(cp["batch"])
            fwd = orc.forward(q, cp["item_cate"], ob, cfg["num_heads"])
            ref = (fwd,) + orc.train_step(q, cp["item_cate"], ob, cfg["num_heads"], reg, lr=lr, clip=clip, l2_extra=extra)
        else:
            for k in q:      # (the same seed draws the same tables on the device: one oracle step serves both modes)
                assert np.array_equal(q[k], q0[k]), k
        fwd, loss, newp, info = ref
        li, _, ut, _ = m.forward(batch, is_test=False, want_u_t=True)
        assert np.abs(li.cpu().numpy() - fwd["logits"]).max() < LOGIT_TOL, l2
        assert np.abs(ut.cpu().numpy() - fwd["u_t"]).max() < LOGIT_TOL, l2
        got_loss = m.train(None, batch, lr)
        assert abs(got_loss - loss) < 3e-6 * abs(loss) + 1e-4, (l2, got_loss, loss)
        assert abs(m.last_gnorm() - info["norm"]) < 3e-4 * info["norm"], (l2, m.last_gnorm(), info["norm"])
        m.fold_scale()
        got = {k: getattr(m, k)[ids[s_]].double().cpu().numpy() for k, s_ in sel.items()}
        got.update({k: np.asarray(v, np.float64) for k, v in m.unpack_dense(m.dense.cpu().numpy()).items()})
        for k in newp:
            du, dr = got[k].reshape(q0[k].shape) - q0[k], newp[k] - q0[k]
            assert np.abs(du - dr).max() < 3e-4 * (np.abs(dr).max() + 1e-9) + 2e-7, (l2, k, float(np.abs(du - dr).max()), float(np.abs(dr).max()))
        decay = 1.0 - lr * info["coef"] * reg
        for k, s_ in sel.items():
            if len(probe[s_]) == 0:      # (Movies-TV: a batch touches all 15 categories)
                continue
            now = getattr(m, k)[pids[s_]].double().cpu().numpy()
            want = old_probe[k] * (decay if k in orc.REG_TABLES else 1.0)      # item_b is not regularised (model.py:164-169)
            assert np.abs(now - want).max() <= 3e-7 * np.abs(want).max() + 1e-12, (l2, k)
        del m
        torch.cuda.empty_cache()
    return cp


def test_c5_one_step_matches_oracle_at_size():
    """BASELINE.json configs[4] at its own size -- 10 M users / 5 M items / 10 k categories, d = 256 (128/128/128), window
    90, batch 4096 -- against the oracle, by id compaction (one oracle step on the ~70 k touched item rows: ~1 min of numpy)."""
    from tlsan_amd import synth
    cfg = synth.make_config("electronics", Ls=90, hidden_units=256, itemid_embedding_size=128, userid_embedding_size=128,
                            cateid_embedding_size=128, user_count=10_000_000, item_count=5_000_000, cate_count=10_000)
    icl = synth.item_cate_list(cfg)
    cp = _one_step_against_oracle_at_size(cfg, icl, synth.make_batches(cfg, 1, 4096, seed=55)[0], seed=7)
    assert len(cp["items"]) > 50_000 and len(cp["users"]) > 4000          # (tens of thousands of rows, spread over tables of 10^7)


def test_c4_one_step_matches_oracle_at_size():
    """BASELINE.json configs[3]'s shape -- Movies-TV (35896 / 28589 / 15), d = 128, window 90, batch 4096 -- on one GPU
    against the oracle at size (the sharded step is held to this single-GPU step by
    test_c4_movies_tv_window_90_sharded_over_two_ranks)."""
    from tlsan_amd import synth
    cfg = _c4_cfg()
    icl = synth.item_cate_list(cfg)
    _one_step_against_oracle_at_size(cfg, icl, synth.make_batches(cfg, 1, 4096, seed=300)[0], seed=9, n_probe=20000)


# ------------------------------------------------------------------------------------------- C4
def _c4_cfg():
    from tlsan_amd import synth
    return synth.make_config("movies_tv", Ls=90)


def _c4_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tlsan_amd import synth
        from tlsan_amd.dist import ShardedModel
        from tlsan_amd.model import Model
        cfg = _c4_cfg()
        icl = synth.item_cate_list(cfg)
        B = 1024
        per_step = [[synth.make_batches(cfg, 1, B, seed=300 + 10 * s + r)[0] for r in range(world)] for s in range(2)]

        def run():
            m = ShardedModel(cfg, icl, device="cuda:0", l2_mode="lazy")
            losses = []
            for per in per_step:
                m.train_async(per[rank], 1.0)
                losses.append(float(m.last_loss.item()))
            return m, losses, m.gather_params()
        m1, l1, p1 = run()
        m2, l2, p2 = run()
        assert l1 == l2
        for k in p1:
            assert np.array_equal(p1[k], p2[k]), k                  # bitwise reproducible across runs

        # the configuration as BASELINE.json names it: static-shape exchanges with bf16 rows on the wire (fp32 weights at
        # the owners) -- reproducible bit for bit, and close to the fp32 exchange (the forward sees bf16-rounded tables)
        def run_wire():
            m = ShardedModel(cfg, icl, device="cuda:0", l2_mode="lazy", static_rows=True, wire_dtype="bf16")
            dbs = [m.device_batch(per[rank]) for per in per_step]
            losses = []
            for s, db in enumerate(dbs):
                m.train_async(db, 1.0, next_batch=dbs[s + 1] if s + 1 < len(dbs) else None)
                losses.append(float(m.last_loss.item()))
            m.check_static_overflow()
            return losses, m.gather_params()
        lw1, pw1 = run_wire()
        lw2, pw2 = run_wire()
        assert lw1 == lw2
        for k in pw1:
            assert np.array_equal(pw1[k], pw2[k]), k
        assert np.allclose(lw1, l1, rtol=2e-3), (lw1, l1)
        p00 = Model.init_params(cfg, 1234)
        for k in pw1:
            du = np.asarray(pw1[k], np.float64) - np.asarray(p00[k], np.float64).reshape(np.shape(pw1[k]))
            dr = np.asarray(p1[k], np.float64) - np.asarray(p00[k], np.float64).reshape(np.shape(p1[k]))
            assert np.abs(du - dr).max() < 0.15 * (np.abs(dr).max() + 1e-12) + 1e-6, (k, float(np.abs(du - dr).max()), float(np.abs(dr).max()))   # (sanity; the semantics are pinned by test_bf16_rows_on_the_wire)
        if rank == 0:
            # the same two global batches (2048 sequences) through the single-GPU model
            def cat2(per):
                Sn = max(np.asarray(b[4]).shape[1] for b in per)
                cols = []
                for c in range(9):
                    parts = [np.asarray(b[c]) for b in per]
                    if c == 4:
                        parts = [np.pad(x, ((0, 0), (0, Sn - x.shape[1]))) for x in parts]
                    cols.append(np.concatenate(parts, 0))
                return tuple(cols)
            ms = Model(cfg, icl, l2_mode="lazy")
            ls = [ms.train(None, cat2(per), 1.0) for per in per_step]
            ps = ms.get_params()
            assert np.allclose(l1, ls, rtol=3e-6, atol=0), (l1, ls)
            p0 = Model.init_params(cfg, 1234)
            # ... and directly against the fp64 oracle on the concatenated batches (round 6: the oracle holds Movies-TV's tables)
            q = {k: np.asarray(v, np.float64) for k, v in p0.items()}
            lo = []
            for per in per_step:
                l, q, _ = orc.train_step(q, icl, orc.as_batch(cat2(per)), 8, cfg["regulation_rate"], lr=1.0)
                lo.append(l)
            assert np.allclose(l1, lo, rtol=2e-5, atol=0), (l1, lo)
            for k in q:
                du = np.asarray(p1[k], np.float64).reshape(q[k].shape) - np.asarray(p0[k], np.float64)
                dr = q[k] - np.asarray(p0[k], np.float64)
                assert np.abs(du - dr).max() < 5e-4 * (np.abs(dr).max() + 1e-12) + 5e-7, ("oracle", k, float(np.abs(du - dr).max()), float(np.abs(dr).max()))
            for k in ps:
                du = np.asarray(p1[k], np.float64).reshape(np.shape(ps[k])) - np.asarray(p0[k], np.float64)
                dr = np.asarray(ps[k], np.float64) - np.asarray(p0[k], np.float64)
                assert np.abs(du - dr).max() < 5e-4 * (np.abs(dr).max() + 1e-12) + 5e-7, (k, float(np.abs(du - dr).max()), float(np.abs(dr).max()))
        ret[rank] = "ok"
    except Exception:
        import traceback
        ret[rank] = "FAIL: " + traceback.format_exc()
    finally:
        dist.destroy_process_group()


def test_c4_movies_tv_window_90_sharded_over_two_ranks():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_c4_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    assert all(ret.get(r) == "ok" for r in range(world)), dict(ret)


# ------------------------------------------------------------------------------------------- C5, sharded
def _c5_sharded_worker(rank, world, port, ret):
    import gc
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tlsan_amd import synth
        from tlsan_amd.dist import ShardedModel
        cfg = synth.make_config("electronics", Ls=90, hidden_units=256, itemid_embedding_size=128, userid_embedding_size=128,
                                cateid_embedding_size=128, user_count=10_000_000, item_count=5_000_000, cate_count=10_000)
        icl = synth.item_cate_list(cfg)
        B = 1024
        per_step = [[synth.make_batches(cfg, 1, B, seed=500 + 10 * s + r)[0] for r in range(world)] for s in range(3)]

        def run(static):
            m = ShardedModel(cfg, icl, device="cuda:0", l2_mode="lazy", static_rows=static, init="device", seed=11)
            dbs = [m.device_batch(per[rank]) for per in per_step]
            rows0 = m.shard[:4096].clone()
            losses = []
            for s, db in enumerate(dbs):
                kw = dict(after_next=dbs[s + 2] if s + 2 < len(dbs) else None) if static else {}
                m.train_async(db, 1.0, next_batch=dbs[s + 1] if s + 1 < len(dbs) else None, **kw)
                losses.append(float(m.last_loss.item()))
            if static:
                m.check_static_overflow()
            changed = int((m.shard[:4096] != rows0).any(dim=1).sum().item())   # (before the fold, which rescales every row)
            m.fold_scale()
            torch.cuda.synchronize()
            # (the shard stays on the device: 7.5 M rows of 220 floats per rank -- compare digests, keep a slice)
            digest = [float(m.shard.double().sum().item()), float(m.shard.double().pow(2).sum().item()),
                      float(m.cate_emb.double().sum().item()), float(m.dense.double().sum().item())]
            head = m.shard[:4096].cpu().numpy().copy()
            del m, dbs
            gc.collect()
            torch.cuda.empty_cache()
            return losses, digest, head, changed
        l1, d1, h1, c1 = run(False)
        l2, d2, h2, c2 = run(False)
        assert l1 == l2 and d1 == d2 and np.array_equal(h1, h2)          # bitwise reproducible
        l3, d3, h3, c3 = run(True)
        assert l3 == l1 and d3 == d1 and np.array_equal(h3, h1)          # static-shape exchanges: the same step, bit for bit
        assert np.isfinite(l1).all() and l1[0] > 100.0                   # the L2 term of 10^7 usert rows at -1
        assert 0 < c1 < 4096                                             # lazy L2: only rows that received a gradient moved
        ret[rank] = "ok"
    except Exception:
        import traceback
        ret[rank] = "FAIL: " + traceback.format_exc()
    finally:
        dist.destroy_process_group()


def _c5_sharded_oracle_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.helpers import compact_problem
        from tlsan_amd import synth
        from tlsan_amd.dist import ShardedModel
        cfg = synth.make_config("electronics", Ls=90, hidden_units=256, itemid_embedding_size=128, userid_embedding_size=128,
                                cateid_embedding_size=128, user_count=10_000_000, item_count=5_000_000, cate_count=10_000)
        icl = synth.item_cate_list(cfg)
        B, di, Ls, G = 1024, 128, 90, world
        per = [synth.make_batches(cfg, 1, B, seed=700 + r)[0] for r in range(world)]      # (every rank can build every part)
        Sn = max(np.asarray(b[4]).shape[1] for b in per)
        gb = tuple(np.concatenate([np.pad(np.asarray(b[c]), ((0, 0), (0, Sn - np.asarray(b[c]).shape[1]))) if c == 4 else np.asarray(b[c])
                                   for b in per], 0) for c in range(9))
        cp = compact_problem(gb, icl)
        m = ShardedModel(cfg, icl, device="cuda:0", l2_mode="lazy", init="device", seed=13)
        dev = m.device

        def mine(ids, base):      # the touched rows this rank owns (ModPartition: id % G, local row id // G), as fused shard rows
            own = ids[ids % G == rank]
            return own, m.shard[torch.as_tensor(base + own // G, device=dev)].double().cpu().numpy()

        def collect():
            parts = [None] * world
            dist.all_gather_object(parts, (mine(cp["items"], 0), mine(cp["users"], m.cI)))
            it = np.zeros((len(cp["items"]), m.W)); us = np.zeros((len(cp["users"]), m.W))
            for (ii, ir), (ui, ur) in parts:
                it[np.searchsorted(cp["items"], ii)] = ir
                us[np.searchsorted(cp["users"], ui)] = ur
            q = dict(item_emb=it[:, :di].copy(), item_b=it[:, di].copy(), user_emb=us[:, :di].copy(), usert_emb=us[:, di:di + Ls].copy(),
                     cate_emb=m.cate_emb[torch.as_tensor(cp["cates"], device=dev)].double().cpu().numpy())
            q.update({k: np.asarray(v, np.float64) for k, v in m._unpack_dense(m.dense.cpu().numpy()).items()})
            return q

        q0 = collect()
        # sum of squares of the regularised tables over ALL ranks (chunked fp64 on the device), minus the compact part
        ssq = torch.tensor([_sumsq64(m.shard[:m.cI, :di]) + _sumsq64(m.shard[m.cI:, :di + Ls])], dtype=torch.float64)
        dist.all_reduce(ssq)
        extra = float(ssq.item()) + _sumsq64(m.cate_emb) - sum(float((q0[k] ** 2).sum()) for k in orc.REG_TABLES)
        m.train_async(per[rank], 1.0)
        loss, norm = float(m.last_loss.item()), float(m.last_gnorm.item())
        m.fold_scale()
        q1 = collect()
        if rank == 0:
            lo, newp, info = orc.train_step(q0, cp["item_cate"], orc.as_batch(cp["batch"]), 8, cfg["regulation_rate"], lr=1.0, l2_extra=extra)
            assert abs(loss - lo) < 3e-6 * abs(lo) + 1e-4, (loss, lo)
            assert abs(norm - info["norm"]) < 3e-4 * info["norm"], (norm, info["norm"])
            for k in newp:
                du, dr = q1[k].reshape(q0[k].shape) - q0[k], newp[k] - q0[k]
                assert np.abs(du - dr).max() < 5e-4 * (np.abs(dr).max() + 1e-9) + 5e-7, (k, float(np.abs(du - dr).max()), float(np.abs(dr).max()))
        ret[rank] = "ok"
    except Exception:
        import traceback
        ret[rank] = "FAIL: " + traceback.format_exc()
    finally:
        dist.destroy_process_group()


def test_c5_sharded_step_matches_oracle_at_size():
    """BASELINE.json configs[4] through the SHARDED step against the fp64 oracle (round 6): two ranks (two processes on
    cuda:0, gloo), tables of 10 M users / 5 M items row-sharded by id % 2, one lazy step of 2 x 1024 sequences.  The rows
    the global batch touches are collected from their owners' shards (before and after the step), the other rows enter the
    oracle through their sum of squares -- the id compaction of `_one_step_against_oracle_at_size`."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_c5_sharded_oracle_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(1500)
    assert all(ret.get(r) == "ok" for r in range(world)), dict(ret)


def test_c5_tables_sharded_over_two_ranks():
    """BASELINE.json configs[4] (10 M users / 5 M items / 10 k categories, d = 256, window 90) through the SHARDED step: two
    ranks (two processes on cuda:0, gloo), each holding half of the rows (7.5 M fused rows of 220 floats), 1024
    sequences per rank and step.  Properties (the oracle cannot run this size): bitwise reproducible; the static-shape
    step (fixed-size exchanges, plans two batches ahead) equals the sized one bit for bit; lazy L2 leaves rows without
    a gradient alone."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_c5_sharded_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(1500)
    assert all(ret.get(r) == "ok" for r in range(world)), dict(ret)


# ------------------------------------------------------------------------------------------- multi-GPU over RCCL
def _nccl_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:%d" % rank))
    try:
        from tests.helpers import random_batch, random_params
        from tlsan_amd.dist import ShardedModel
        cfg = make_config(U=61, I=83, C=9, d=128, regulation_rate=1e-3)
        p = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in random_params(cfg, seed=17).items()}
        _, cat = random_batch(cfg, B=4, Sn=2, seed=0)
        tup = lambda b: (b["u"], b["i"], b["y"], b["hist_i"], b["hist_i_new"], b["hist_t"], b["sl"], b["sl_new"], b["u_cate"])
        steps = [[random_batch(cfg, B=24, Sn=3, seed=1000 + 10 * s + r)[0] for r in range(world)] for s in range(3)]
        # static: fixed-size exchanges, nothing through the host; coalesce: its two exchanges behind the kernels as one RCCL group
        for lazy, static, coalesce in ((False, False, False), (True, False, False), (True, True, False), (True, True, True)):
            m = ShardedModel(cfg, cat, device="cuda:%d" % rank, l2_mode="lazy" if lazy else "dense", static_rows=static,
                             coalesce=coalesce)
            m.set_params({k: np.asarray(v, np.float32) for k, v in p.items()})
            dbs = [m.device_batch(tup(per[rank])) for per in steps]
            losses = []
            for k, db in enumerate(dbs):
                m.train_async(db, 0.8, next_batch=dbs[k + 1] if k + 1 < len(dbs) else None)
                losses.append(float(m.last_loss.item()))
            got = m.gather_params()
            if rank == 0:
                q, ref = dict(p), []
                for per in steps:
                    Sn = max(b["hist_i_new"].shape[1] for b in per)
                    g = {k: np.concatenate([np.pad(b[k], ((0, 0), (0, Sn - b[k].shape[1]))) if k == "hist_i_new" else b[k]
                                            for b in per], 0) for k in per[0]}
                    l, q, _ = orc.train_step(q, cat, g, 8, cfg["regulation_rate"], lr=0.8)
                    ref.append(l)
                assert np.allclose(losses, ref, rtol=2e-4, atol=1e-5), (losses, ref)
                for k in q:
                    du = np.asarray(got[k], np.float64).reshape(q[k].shape) - p[k]
                    dr = q[k] - p[k]
                    assert np.abs(du - dr).max() < 5e-4 * (np.abs(dr).max() + 1e-9) + 5e-7, (lazy, static, coalesce, k)
        ret[rank] = "ok"
    except Exception:
        import traceback
        ret[rank] = "FAIL: " + traceback.format_exc()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_step_over_rccl(world):
    """The production transport: one process per GPU, all-to-all / all-reduce on device buffers over RCCL.
    Needs `world` GPUs in the box (the 1-GPU boxes of the test pool skip it; the 8-GPU node runs it)."""
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs, found %d" % (world, torch.cuda.device_count()))
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
    assert all(ret.get(r) == "ok" for r in range(world)), dict(ret)


# ------------------------------------------------------------------------------------------- driver + graph
def test_driver_resume_and_from_scratch(tmp_path):
    """train.py:124-127 (`from_scratch` wipes model_dir) and :71-76 (otherwise the latest checkpoint is
    reloaded: parameters, global_step, epoch counter) through tlsan_amd.train."""
    from tlsan_amd import train as T
    ds = os.path.join(GOLDEN, "packed_clothing.npz")
    d = str(tmp_path / "ckpt")
    base = ["--dataset", ds, "--eval_freq", "100", "--quiet", "--model_dir", d, "--eval_topk", "0"]
    a = T.train(T.parse(base + ["--max_steps", "200"]))
    assert a["steps"] == 200 and os.path.exists(os.path.join(d, "TLSAN-200.npz"))
    open(os.path.join(d, "stale.txt"), "w").write("x")
    b = T.train(T.parse(base + ["--max_steps", "300", "--from_scratch", "false"]))
    # resumed: starts from the saved parameters (its initial AUC is the first run's final one), counts on from 200
    assert b["init_auc"] == pytest.approx(a["final_auc"], abs=1e-9) and b["steps"] == 300
    assert os.path.exists(os.path.join(d, "stale.txt")) and os.path.exists(os.path.join(d, "TLSAN-300.npz"))
    rows = [l.split(",") for l in open(os.path.join(d, "eval", "scalars.csv"))]
    assert [int(r[0]) for r in rows if r[1] == "AUC"][:2] == [0, 100]              # both runs' rows: appended
    c = T.train(T.parse(base + ["--max_steps", "100"]))                              # from_scratch (default): wiped
    assert not os.path.exists(os.path.join(d, "stale.txt")) and not os.path.exists(os.path.join(d, "TLSAN-300.npz"))
    assert c["init_auc"] == pytest.approx(a["init_auc"], abs=1e-9) and c["steps"] == 100
    # the same data order in both fresh runs (the reference's shuffle stream): identical trajectories
    assert c["history"][0][2] == pytest.approx(a["history"][0][2], abs=1e-12)


def test_captured_graphs_survive_longer_sessions():
    """a graph bakes the workspace pointer in; capturing batches of growing session length must not
    leave earlier graphs writing into a freed block."""
    from tests.helpers import random_batch, random_params
    from tlsan_amd.model import Model
    cfg = make_config(U=80, I=120, C=9, d=128)
    p = {k: np.asarray(v, np.float32) for k, v in random_params(cfg, seed=3).items()}
    tup = lambda b: (b["u"], b["i"], b["y"], b["hist_i"], b["hist_i_new"], b["hist_t"], b["sl"], b["sl_new"], b["u_cate"])
    bs = [random_batch(cfg, B=64, Sn=sn, seed=40 + sn)[0] for sn in (1, 4, 9, 17)]
    cat = random_batch(cfg, B=4, Sn=1, seed=0)[1]
    eager = Model(cfg, cat, l2_mode="lazy"); eager.set_params(p)
    for b in bs + bs:
        eager.train_async(tup(b), 0.7)
    m = Model(cfg, cat, l2_mode="lazy"); m.set_params(p)
    graphs = [m.capture_step(tup(b), 0.7) for b in bs]      # Sn grows from capture to capture
    assert all(g._tlsan_ws is m._ws for g in graphs)
    m.set_params(p)                                         # (capture_step ran one warm step per batch)
    for g in graphs + graphs:
        m.replay(g)
    torch.cuda.synchronize()
    a, b_ = eager.get_params(), m.get_params()
    for k in a:
        assert np.array_equal(a[k], b_[k]), k
    # a batch larger than anything sized for makes the workspace grow: stale graphs refuse to replay
    big = random_batch(cfg, B=4096, Sn=2, seed=99)[0]
    m.train_async(tup(big), 0.7)
    with pytest.raises(RuntimeError):
        m.replay(graphs[0])


# ------------------------------------------------------------------------------------------- README band
def test_readme_band_clothing(tmp_path):
    """The reference's only published float result (README.md:29-41) on the dataset BASELINE.json configs[0] names:
    the full protocol (TLSAN/train.py:26-49 defaults, 20 epochs = 6180 steps, "Best test_auc") on the real Clothing
    samples, at the published regulation_rate and at 5e-6, against `tests/golden/readme_band.json` (means of three
    initialisations each, profiles/r03_readme_band.md) and against the README's 0.9363 -- a change of the kernels' or
    the oracle's reading of the L2 term / the update shows up here as a shift of either number."""
    import json
    from tlsan_amd import train as T
    gold = json.load(open(os.path.join(GOLDEN, "readme_band.json")))["clothing"]
    ds = os.path.join(GOLDEN, "packed_clothing.npz")
    for tag, reg in (("5e-5", "0.00005"), ("5e-6", "0.000005")):
        res = T.train(T.parse(["--dataset", ds, "--quiet", "--eval_topk", "0", "--regulation_rate", reg, "--seed", "1234",
                               "--model_dir", str(tmp_path / tag)]))
        g = gold[tag]
        assert res["steps"] == g["steps"] == 6180
        # one initialisation against the mean of three: three run-to-run sigmas (at least 0.003) + the evaluation noise
        assert abs(res["best_auc"] - g["mean"]) <= 3.0 * max(g["sigma"], 0.003) + 0.002, (tag, res["best_auc"], g)
        # ... and the README itself, within two sigmas of the test set's sampling error and the run-to-run spread
        assert abs(res["best_auc"] - g["readme"]) <= 2.0 * float(np.hypot(g["sampling_sigma"], max(g["sigma"], 0.003))), (tag, res["best_auc"])
