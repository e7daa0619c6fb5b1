"""GPU tests of the sharded multi-GPU step (tlsan_amd.dist.ShardedModel) against the oracle.
Only one GPU is available to the tests, so world_size 2 runs as two processes that share
cuda:0 and talk over gloo (host-staged all-to-all); the production path is RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tlsan_oracle as orc
from tests.helpers import make_config, random_batch, random_params

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tuple(b):
    return (b["u"], b["i"], b["y"], b["hist_i"], b["hist_i_new"], b["hist_t"], b["sl"], b["sl_new"], b["u_cate"])


def _case(d=128, clip=5.0, C=9, **extra):
    cfg = make_config(U=61, I=83, C=C, d=d, regulation_rate=1e-3, max_gradient_norm=clip, **extra)
    p = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in random_params(cfg, seed=17).items()}
    _, cat = random_batch(cfg, B=4, Sn=2, seed=0)
    return cfg, p, cat


def _split_batches(cfg, world, steps, B, uneven=False):
    out = []
    for s in range(steps):
        per = [random_batch(cfg, B=B - (7 * r if uneven else 0), Sn=3, seed=1000 + 10 * s + r)[0] for r in range(world)]
        out.append(per)
    return out


def _concat(per):
    """the global batch all ranks train on together (pad session columns to a common width)"""
    Sn = max(b["hist_i_new"].shape[1] for b in per)
    out = {}
    for k in per[0]:
        if k == "hist_i_new":
            out[k] = np.concatenate([np.pad(b[k], ((0, 0), (0, Sn - b[k].shape[1]))) for b in per], 0)
        else:
            out[k] = np.concatenate([b[k] for b in per], 0)
    return out


def _worker(rank, world, port, ret, d, prefetch, ckpt, uneven=False, hotcat=False, lazy=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tlsan_amd.dist import ShardedModel
        clip = 0.05 if uneven else 5.0          # (uneven shares: with the clip active, so that the weighted squares matter)
        # (hotcat: 2 categories and 150 samples per rank -> >1000 uses per category: the row-sum pass of
        #  tlsan_grads splits every category over several workgroups)
        cfg, p, cat = _case(d, clip, C=2 if hotcat else 9)
        m = ShardedModel(cfg, cat, device="cuda:0", l2_mode="lazy" if lazy else "dense")
        m.set_params({k: np.asarray(v, np.float32) for k, v in p.items()})
        if uneven:
            m._pcap = 5     # a shared exchange capacity far too small: the overflow protocol must raise it in lockstep
        steps = _split_batches(cfg, world, 4, B=150 if hotcat else 24, uneven=uneven)
        wgt = lambda per: len(per[rank]["u"]) * world / sum(len(q_["u"]) for q_ in per)
        losses = []
        if prefetch:   # every step is told its successor: routing plan + indices are built a step ahead
            dbs = [m.device_batch(_tuple(per[rank])) for per in steps]
            for k, db in enumerate(dbs):
                m.train_async(db, 0.8, next_batch=dbs[k + 1] if k + 1 < len(dbs) else None, weight=wgt(steps[k]))
                losses.append(float(m.last_loss.item()))
        else:
            for per in steps:
                m.train_async(_tuple(per[rank]), 0.8, weight=wgt(per))
                losses.append(float(m.last_loss.item()))
        if uneven:
            assert m._pcap > 5 and all(sl is None or sl["pcap"] == min(m.router.R, m._pcap) for sl in m._slots[:2])
        got = m.gather_params()
        auc = m.eval_auc(None, tuple(list(_tuple(steps[0][rank]))[:2] + [steps[0][rank]["i"][::-1].copy()] + list(_tuple(steps[0][rank]))[3:]))
        if rank == 0:
            q = dict(p)
            ref = []
            for per in steps:
                l, q, info = orc.train_step(q, cat, _concat(per), 8, cfg["regulation_rate"], lr=0.8, clip=clip)
                assert (info["coef"] < 1.0) == uneven
                ref.append(l)
            assert np.allclose(losses, ref, rtol=2e-4, atol=1e-5), (losses, ref)
            for k in q:
                g = np.asarray(got[k], np.float64).reshape(q[k].shape)
                du, dr = g - p[k], q[k] - p[k]
                assert np.abs(du - dr).max() < 5e-4 * (np.abs(dr).max() + 1e-9) + 5e-7, (k, float(np.abs(du - dr).max()), float(np.abs(dr).max()))
        assert 0.0 <= auc <= 1.0
        if uneven:     # (the ranking's all-gather is equal-sized: tlsan_amd.train pads the shares, see _equal_share)
            ret[rank] = "ok"
            return
        # all-items ranking with the items sharded == the oracle's ranks on the gathered parameters
        tb = steps[0][rank]
        ranks = m.label_ranks(tuple(list(_tuple(tb))[:2] + [tb["i"][::-1].copy()] + list(_tuple(tb))[3:])).cpu().numpy()
        full = {k: np.asarray(v, np.float64) for k, v in got.items()}
        out = orc.forward(full, cat, tb, 8)
        sc = orc.all_item_scores(full, cat, out["u_t"]) if hasattr(orc, "all_item_scores") else None
        if sc is None:
            all_emb = np.concatenate([full["item_emb"], full["cate_emb"][np.asarray(cat)]], 1)
            sc = out["u_t"] @ all_emb.T + full["item_b"][None, :]
        lab = np.asarray(tb["i"])
        own = sc[np.arange(len(lab)), lab]
        ids = np.arange(sc.shape[1])[None, :]
        ref_rank = ((sc > own[:, None]) | ((sc == own[:, None]) & (ids < lab[:, None]))).sum(1)
        margin = np.abs(sc - own[:, None]); margin[np.arange(len(lab)), lab] = 1.0
        clear = margin.min(1) > 1e-4            # rows whose rank does not hinge on an fp32-level near tie
        assert clear.sum() >= len(lab) // 2 and np.array_equal(ranks[clear], ref_rank[clear]), (ranks, ref_rank)
        rec = m.eval_recall(None, tuple(list(_tuple(tb))[:2] + [tb["i"][::-1].copy()] + list(_tuple(tb))[3:]))
        assert len(rec) == 6 and all(0.0 <= x <= 1.0 for x in rec) and rec == sorted(rec)   # R@1 <= ... <= R@50, global batch
        # checkpoints: per-rank shard files and the gathered single-file format both restore the exact state
        m.config["model_dir"] = ckpt
        prefix = m.save()
        one = m.save(sharded=False)
        for path in (prefix, one + ".npz"):
            m2 = ShardedModel(cfg, cat, device="cuda:0", seed=99, l2_mode="lazy" if lazy else "dense")   # different initial values
            m2.restore(None, path)
            assert m2.global_step.eval() == m.global_step.eval() == 4
            back = m2.gather_params()
            for k in got:
                assert np.array_equal(back[k], got[k]), (path, k)
            # (the running sums of squares are recomputed from the restored tables: equal up to rounding)
            assert torch.allclose(m2._sq, m._sq, rtol=1e-6, atol=0) and torch.equal(m2.dense_KT, m.dense_KT)
        ret[rank] = "ok"
    except Exception:
        import traceback
        ret[rank] = "FAIL: " + traceback.format_exc()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,d,prefetch,uneven,hotcat,lazy", [
    (1, 128, False, False, False, False), (2, 128, False, False, False, False), (2, 64, False, False, False, False),
    (1, 128, True, False, False, False), (2, 128, True, False, False, False), (2, 128, True, True, False, False),
    (2, 128, True, False, True, False),
    (1, 128, False, False, False, True), (2, 128, True, False, False, True), (2, 64, True, True, False, True)])
def test_sharded_model_matches_oracle(world, d, prefetch, uneven, hotcat, lazy, tmp_path):
    """(uneven: the ranks hold 24 and 17 samples of each global batch and enter with their shares;
    hotcat: two categories with > 1000 uses each; lazy: the owners' update in its lazy-L2 form)"""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, d, prefetch, str(tmp_path), uneven, hotcat, lazy), nprocs=world, join=True)
    assert all(v == "ok" for v in dict(ret).values()) and len(ret) == world, dict(ret)


def test_rows_apply_matches_numpy_and_is_deterministic():
    import ctypes as C
    from tlsan_amd import _lib as L
    lib = L.load()
    rng = np.random.RandomState(3)
    nrows, width, ld, n = 300, 36, 40, 5000
    W0 = rng.randn(nrows, ld).astype(np.float32)
    G = rng.randn(n, width).astype(np.float32)
    dest = np.minimum(rng.zipf(1.3, n) - 1, nrows - 1).astype(np.int32)   # skewed: long lists
    step, reg, gscale, reg_cols = 0.37, 1e-2, 0.5, 32
    outs = []
    for rep in range(2):
        W = torch.as_tensor(W0).cuda()
        Gd, dd = torch.as_tensor(G).cuda(), torch.as_tensor(dest).cuda()
        sd = torch.tensor([step], dtype=torch.float32, device="cuda")
        ss = torch.zeros(1, dtype=torch.float64, device="cuda")
        nb = lib.tlsan_rows_apply_workspace(nrows, n)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        L.check(lib.tlsan_rows_apply(W.data_ptr(), ld, nrows, width, reg_cols, Gd.data_ptr(), width, dd.data_ptr(), n,
                                     gscale, sd.data_ptr(), reg, ss.data_ptr(), ws.data_ptr(), nb,
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), "tlsan_rows_apply")
        outs.append((W.cpu().numpy(), float(ss.item())))
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
    acc = np.zeros((nrows, width), np.float64)
    np.add.at(acc, dest, G.astype(np.float64))
    ref = W0.astype(np.float64).copy()
    g = gscale * acc
    g[:, :reg_cols] += reg * ref[:, :reg_cols]
    ref[:, :width] -= step * g
    assert np.abs(outs[0][0] - ref).max() < 2e-5
    assert np.array_equal(outs[0][0][:, width:], W0[:, width:])           # padding columns untouched
    assert abs(outs[0][1] - (ref[:, :reg_cols] ** 2).sum()) < 1e-3 * (ref[:, :reg_cols] ** 2).sum()


@pytest.mark.parametrize("n_items,n_users,n_uses", [(203, 117, 700), (260_000, 90_000, 5000)])
def test_route_plan_matches_key_router(n_items, n_users, n_uses):
    """tlsan_route_plan (the GPU routing of the sharded step) against KeyRouter.plan with the torch
    scan (what tests/test_dist_cpu.py checks over gloo): same distinct rows, same per-owner counts,
    same local row numbers in all-to-all order, same compact ids, same category map.  The large case
    has a key space of 86 chunks of 4096 -> the scan runs in its two-launch form with chunk sums."""
    import ctypes as C
    from tlsan_amd import _lib as L
    from tlsan_amd.dist import KeyRouter
    lib = L.load()
    G = 4
    r = KeyRouter(n_items, n_users, G, 0)
    rng = np.random.RandomState(11)
    items, users = rng.randint(0, n_items, n_uses), rng.randint(0, n_users, 64)
    keys = torch.cat([r.item_keys(torch.as_tensor(items)), r.user_keys(torch.as_tensor(users))]).to(torch.int32).cuda()
    cbk = torch.full((r.nkeys,), -1, dtype=torch.int32)
    ids = np.arange(n_items)
    cbk[(ids % G) * r.R + ids // G] = torch.as_tensor(rng.randint(0, 9, n_items).astype(np.int32))
    cbk = cbk.cuda()
    nk, cap, pad = int(keys.numel()), r.R, 8192
    z = lambda n: torch.zeros(n, dtype=torch.int32, device="cuda")
    flags, rank, uniq, n_uniq, sendbuf, cate_c, comp = z(r.nkeys), z(r.nkeys), z(r.nkeys), z(1), z(G * (1 + cap)), z(pad), z(nk)
    hostc = torch.zeros(G, dtype=torch.int32).pin_memory()      # the kernel writes the counts straight to the host
    L.check(lib.tlsan_route_plan(keys.data_ptr(), nk, r.R, G, cbk.data_ptr(), flags.data_ptr(), rank.data_ptr(), uniq.data_ptr(),
                                 n_uniq.data_ptr(), sendbuf.data_ptr(), cap, cate_c.data_ptr(), pad, comp.data_ptr(),
                                 hostc.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "tlsan_route_plan")
    torch.cuda.synchronize()
    # reference: the same quantities KeyRouter.plan derives (checked over gloo in tests/test_dist_cpu.py)
    kn = keys.cpu().numpy().astype(np.int64)
    u_ref = np.unique(kn)                                   # distinct keys, ascending = grouped by owner
    n = len(u_ref)
    counts = [int(((u_ref >= g * r.R) & (u_ref < (g + 1) * r.R)).sum()) for g in range(G)]
    assert int(n_uniq.item()) == n
    assert np.array_equal(uniq[:n].cpu().numpy(), u_ref)
    sb = sendbuf.view(G, 1 + cap).cpu().numpy()
    assert sb[:, 0].tolist() == counts == hostc.tolist()
    rows = np.concatenate([sb[g, 1:1 + counts[g]] for g in range(G)])
    assert np.array_equal(rows, u_ref % r.R)                # local row numbers in all-to-all send order
    assert np.array_equal(comp.cpu().numpy(), np.searchsorted(u_ref, kn))
    assert np.array_equal(cate_c[:n].cpu().numpy(), cbk.cpu().numpy()[u_ref])
    assert (cate_c[n:pad] == -1).all() and int(flags.abs().sum().item()) == 0     # pads marked, marks cleared
    # and the python router agrees on the key space
    assert np.array_equal(r.item_keys(torch.as_tensor(items)).numpy(), (items % G) * r.R + items // G)


@pytest.mark.parametrize("n_items,n_users,n_uses,cap", [(203, 117, 700, 96), (260_000, 90_000, 5000, 1536), (203, 117, 700, 40)])
def test_static_route_plan_and_gather(n_items, n_users, n_uses, cap):
    """tlsan_route_plan_static / tlsan_shard_gather_static against numpy: slot numbering (the j-th distinct row of
    owner g is compact row g * cap + j), per-owner request lists, category map with -1 in the empty slots, and on the
    owner's side the gathered rows, `recv_rows` (-1 = empty) and the slot marks.  cap = 40 is too small for this
    batch: counts are clamped and the true maximum lands in `status`."""
    import ctypes as C
    from tlsan_amd import _lib as L
    from tlsan_amd.dist import KeyRouter
    lib = L.load()
    G = 4
    r = KeyRouter(n_items, n_users, G, 0)
    rng = np.random.RandomState(12)
    items, users = rng.randint(0, n_items, n_uses), rng.randint(0, n_users, 64)
    keys = torch.cat([r.item_keys(torch.as_tensor(items)), r.user_keys(torch.as_tensor(users))]).to(torch.int32).cuda()
    cbk = torch.full((r.nkeys,), -1, dtype=torch.int32)
    ids = np.arange(n_items)
    cbk[(ids % G) * r.R + ids // G] = torch.as_tensor(rng.randint(0, 9, n_items).astype(np.int32))
    cbk = cbk.cuda()
    nk = int(keys.numel())
    z = lambda n: torch.zeros(n, dtype=torch.int32, device="cuda")
    flags, rank, uniq, n_uniq, sendbuf, comp, status = z(r.nkeys), z(r.nkeys), z(r.nkeys), z(1), z(G * (1 + cap)), z(nk), z(1)
    cate_c = torch.full((G * cap,), 7777, dtype=torch.int32, device="cuda")
    counts_out = z(G)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.tlsan_route_plan_static(keys.data_ptr(), nk, r.R, G, cbk.data_ptr(), flags.data_ptr(), rank.data_ptr(),
                                        uniq.data_ptr(), n_uniq.data_ptr(), sendbuf.data_ptr(), cap, cate_c.data_ptr(),
                                        comp.data_ptr(), counts_out.data_ptr(), status.data_ptr(), st), "tlsan_route_plan_static")
    torch.cuda.synchronize()
    kn = keys.cpu().numpy().astype(np.int64)
    u_ref = np.unique(kn)
    per = [u_ref[(u_ref >= g * r.R) & (u_ref < (g + 1) * r.R)] for g in range(G)]
    true_counts = [len(x) for x in per]
    assert counts_out.cpu().tolist() == true_counts
    assert int(flags.abs().sum().item()) == 0                                      # marks cleared
    sb = sendbuf.view(G, 1 + cap).cpu().numpy()
    cc = cate_c.cpu().numpy().reshape(G, cap)
    cbk_h = cbk.cpu().numpy()
    fits = max(true_counts) <= cap
    assert int(status.item()) == (0 if fits else max(true_counts))
    for g in range(G):
        c = min(true_counts[g], cap)
        assert sb[g, 0] == c
        assert np.array_equal(sb[g, 1:1 + c], per[g][:c] - g * r.R)                # row numbers inside the owner's shard
        assert np.array_equal(cc[g, :c], cbk_h[per[g][:c]]) and (cc[g, c:] == -1).all()
    if fits:
        where = {int(k): g * cap + j for g in range(G) for j, k in enumerate(per[g])}
        assert np.array_equal(comp.cpu().numpy(), np.array([where[int(k)] for k in kn]))
    # owner side (as if this rank were asked by all G ranks for what it asked of owner g: reuse sendbuf as recvbuf)
    W = 12
    shard = torch.arange(r.R * W, dtype=torch.float32, device="cuda").view(r.R, W)
    rows = torch.full((G * cap, W), -5.0, device="cuda")
    recv_rows = z(G * cap)
    slots = torch.zeros(r.R * G, dtype=torch.int64, device="cuda")
    stamp = torch.full((1,), 9, dtype=torch.int32, device="cuda")
    L.check(lib.tlsan_shard_gather_static(shard.data_ptr(), W, r.R, W, sendbuf.data_ptr(), cap, G, rows.data_ptr(),
                                          recv_rows.data_ptr(), slots.data_ptr(), stamp.data_ptr(), st), "tlsan_shard_gather_static")
    torch.cuda.synchronize()
    rr = recv_rows.cpu().numpy().reshape(G, cap)
    rows_h, sl = rows.cpu().numpy().reshape(G, cap, W), slots.cpu().numpy().reshape(r.R, G)
    for g in range(G):
        c = sb[g, 0]
        assert np.array_equal(rr[g, :c], sb[g, 1:1 + c]) and (rr[g, c:] == -1).all()
        assert np.array_equal(rows_h[g, :c], shard.cpu().numpy()[sb[g, 1:1 + c]])
        assert (rows_h[g, c:] == -5.0).all()                                        # empty slots are not touched
        for j in range(c):
            assert sl[sb[g, 1 + j], g] == (9 << 32) | (g * cap + j + 1)
    assert np.count_nonzero(sl) == int(sb[:, 0].sum())


def _driver_worker(rank, world, port, ret, ckpt, extra):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tlsan_amd import train as T
        ds = os.path.join(os.path.dirname(__file__), "golden", "packed_clothing.npz")
        # batch 33 over 2 ranks: uneven shares (16 / 17) -> the weighted means; test batches of 128 split 64 / 64,
        # the last one (2010 % 128 = 90) 45 / 45
        argv = ["--dataset", ds, "--max_steps", "40", "--eval_freq", "20", "--quiet", "--train_batch_size", "33",
                "--model_dir", os.path.join(ckpt, "r%d" % rank), "--device_input", "0"] + list(extra)
        res = T.train_sharded(T.parse(argv + ["--sharded", "1"]))
        if rank == 0:
            one = T.train(T.parse(argv))
            assert res["steps"] == one["steps"] == 40 and res["world"] == world
            assert abs(res["init_auc"] - one["init_auc"]) < 1e-9, (res["init_auc"], one["init_auc"])
            assert abs(res["final_auc"] - one["final_auc"]) < 2e-3, (res["final_auc"], one["final_auc"])
            for a, b in zip(res["history"], one["history"]):
                assert a[0] == b[0] and abs(a[2] - b[2]) < 2e-3, (a, b)
            assert np.allclose(res["recall"], one["recall"], atol=2e-3), (res["recall"], one["recall"])
            assert np.allclose(res["prec"], one["prec"], atol=2e-3), (res["prec"], one["prec"])
        ret[rank] = "ok"
    except Exception:
        import traceback
        ret[rank] = "FAIL: " + traceback.format_exc()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("extra", [(), ("--optimizer", "adam", "--learning_rate", "0.01", "--dropout", "0.1"),
                                   ("--l2_mode", "lazy"), ("--l2_mode", "lazy", "--static_rows", "1")])
def test_sharded_train_driver_matches_single_gpu(extra, tmp_path):
    """python -m tlsan_amd.train --sharded: the reference's train.py flow over 2 ranks (global batches
    split over the ranks, unevenly here; evaluation over split test batches) against the single-GPU driver
    on the same data and shuffle -- with the defaults, and with adam + dropout."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_driver_worker, args=(2, _free_port(), ret, str(tmp_path), extra), nprocs=2, join=True)
    assert all(v == "ok" for v in dict(ret).values()) and len(ret) == 2, dict(ret)


def _opt_worker(rank, world, port, ret, optimizer, dropout):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tlsan_amd.dist import ShardedModel
        clip, lr = 0.05, {"sgd": 0.8, "adam": 0.05, "rmsprop": 0.02, "adadelta": 1.0}[optimizer]
        cfg, p, cat = _case(128, clip, optimizer=optimizer, dropout=dropout)
        m = ShardedModel(cfg, cat, device="cuda:0")
        m.set_params({k: np.asarray(v, np.float32) for k, v in p.items()})
        steps = _split_batches(cfg, world, 4, B=24)
        q = dict(p)
        st = orc.init_opt_state(p, optimizer) if optimizer != "sgd" else None
        for per in steps:
            seed = m.dropout_seed()
            m.train_async(_tuple(per[rank]), lr, sample0=24 * rank)
            loss = float(m.last_loss.item())
            ref, q, info = orc.train_step(q, cat, _concat(per), 8, cfg["regulation_rate"], lr=lr, clip=clip,
                                          optimizer=optimizer, opt_state=st,
                                          dropout=(dropout, seed) if dropout > 0 else None)
            assert info["coef"] < 1.0
            assert abs(loss - ref) < 2e-4 * max(1.0, abs(ref)), (loss, ref)
        got = m.gather_params()
        if rank == 0:
            for k in q:
                if k.endswith("_b2") and optimizer == "adam":
                    continue        # zero-gradient parameter: Adam turns rounding noise into +-lr (see the single-GPU test)
                g = np.asarray(got[k], np.float64).reshape(q[k].shape)
                du, dr = g - p[k], q[k] - p[k]
                assert np.abs(du - dr).max() < 3e-3 * (np.abs(dr).max() + 1e-9) + 1e-6, (k, float(np.abs(du - dr).max()), float(np.abs(dr).max()))
        ret[rank] = "ok"
    except Exception:
        import traceback
        ret[rank] = "FAIL: " + traceback.format_exc()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,optimizer,dropout", [(2, "adam", 0.0), (2, "rmsprop", 0.0), (1, "adadelta", 0.0),
                                                     (2, "sgd", 0.3), (2, "adam", 0.2)])
def test_sharded_optimizers_and_dropout(world, optimizer, dropout):
    """The other optimizers (accumulators sharded like the rows they belong to) and dropout (the pattern
    indexed by the sample's position in the GLOBAL batch) on the sharded step, against the oracle on the
    concatenated batch, clip active."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_opt_worker, args=(world, _free_port(), ret, optimizer, dropout), nprocs=world, join=True)
    assert all(v == "ok" for v in dict(ret).values()) and len(ret) == world, dict(ret)
