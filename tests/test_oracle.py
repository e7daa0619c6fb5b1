"""The oracle checks itself: hand-written numpy backward (oracle/tlsan_oracle.py) vs. the
independent op-for-op torch-autograd restatement (oracle/tlsan_torch_ref.py) vs. finite
differences.  float64; tolerances written here."""
import numpy as np
import pytest
import torch

from oracle import tlsan_oracle as orc
from oracle import tlsan_torch_ref as tref
from tests.helpers import fixture_batch, make_config, random_batch, random_params

TOL = 1e-10


@pytest.mark.parametrize("d,H,Ls,Sn", [(64, 8, 10, 3), (128, 8, 10, 5), (32, 4, 6, 0), (256, 8, 4, 2)])
def test_numpy_vs_torch_forward_and_grads(d, H, Ls, Sn):
    cfg = make_config(d=d, H=H, Ls=Ls)
    p = random_params(cfg, seed=1)
    b, cat = random_batch(cfg, B=9, Sn=Sn, seed=2)
    reg = 5e-3
    loss, logits, g, sparse = orc.backward(p, cat, b, H, reg)
    tp = tref.params_to_torch(p)
    tb = tref.batch_to_torch(b)
    tl, tlog, tg = tref.grads(tp, cat, tb, H, reg)
    assert abs(loss - float(tl)) < TOL
    assert np.abs(logits - tlog.numpy()).max() < TOL
    for k in p:
        assert np.abs(g[k] - tg[k].numpy()).max() < TOL, k
    # dedup norm == norm of autograd's summed gradient
    n1 = orc.global_norm(p, g, sparse, reg, "dedup")
    n2 = float(torch.sqrt(sum((v ** 2).sum() for v in tg.values())))
    assert abs(n1 - n2) < 1e-9
    # tf18 norm >= per construction differs from dedup when ids repeat
    assert orc.global_norm(p, g, sparse, reg, "tf18") > 0


def test_finite_differences():
    cfg = make_config(U=6, I=9, C=3, d=16, H=2, Ls=4)
    p = random_params(cfg, seed=3)
    b, cat = random_batch(cfg, B=5, Sn=2, seed=4)
    reg = 1e-2
    _, _, g, _ = orc.backward(p, cat, b, 2, reg)
    rng = np.random.RandomState(0)
    eps = 1e-6
    for k in p:
        flat = p[k].reshape(-1)
        for idx in rng.choice(flat.size, size=min(6, flat.size), replace=False):
            old = flat[idx]
            flat[idx] = old + eps
            lp = orc.loss_fn(p, cat, b, 2, reg)
            flat[idx] = old - eps
            lm = orc.loss_fn(p, cat, b, 2, reg)
            flat[idx] = old
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - g[k].reshape(-1)[idx]) < 1e-7, (k, idx, fd, g[k].reshape(-1)[idx])


def test_train_step_matches_torch_dedup():
    cfg = make_config(d=64)
    p = random_params(cfg, seed=5)
    b, cat = random_batch(cfg, B=16, Sn=4, seed=6)
    loss, newp, info = orc.train_step(p, cat, b, 8, 5e-5, lr=1.0, clip=0.05, norm_mode="dedup")
    assert info["coef"] < 1.0  # clip active
    tp = tref.params_to_torch(p)
    tl, tn = tref.train_step_(tp, cat, tref.batch_to_torch(b), 8, 5e-5, 1.0, clip=0.05)
    assert abs(tl - loss) < TOL and abs(tn - info["norm"]) < 1e-9
    for k in p:
        assert np.abs(newp[k] - tp[k].detach().numpy()).max() < TOL, k


def test_masked_positions_are_inert():
    """Padded slots must contribute exactly nothing: changing the padded ids /
    time weights must not change logits or gradients."""
    cfg = make_config(d=64)
    p = random_params(cfg, seed=7)
    b, cat = random_batch(cfg, B=8, Sn=4, seed=8)
    _, logits, g, _ = orc.backward(p, cat, b, 8, 5e-5)
    b2 = {k: v.copy() for k, v in b.items()}
    ar = np.arange(cfg["Ls"])[None, :]
    pad = ar >= b["sl"][:, None]
    b2["hist_i"][pad] = 5
    pad2 = np.arange(4)[None, :] >= b["sl_new"][:, None]
    b2["hist_i_new"][pad2] = 7
    _, logits2, g2, _ = orc.backward(p, cat, b2, 8, 5e-5)
    assert np.array_equal(logits, logits2)
    for k in g:
        assert np.array_equal(g[k], g2[k]), k


def test_real_fixture_batch_runs_and_tf18_norm():
    batch, (U, I, C), icl = fixture_batch("clothing")
    cfg = make_config(U=U, I=I, C=C, d=64)
    p = orc.init_params(cfg, seed=1234)
    b = orc.as_batch(batch)
    loss, newp, info = orc.train_step(p, icl, b, 8, 5e-5, 1.0)
    # initial loss = ln2-ish BCE + 5e-5 * l2 (usert=-1 -> U*Ls/2 dominates)
    assert 0.5 < loss < 2.0
    assert info["coef"] == 1.0
    assert np.isfinite(info["logits"]).all()


def test_eval_helpers():
    rng = np.random.RandomState(0)
    s = rng.randn(5, 30)
    s[0, 3] = s[0, 7]  # tie: lower index first
    lab = np.array([7, 1, 2, 3, 4])
    r = orc.label_ranks(s, lab)
    top = orc.topk_ids(s, 30)
    for row in range(5):
        assert top[row, r[row]] == lab[row]
    assert orc.hits_at_k(s, lab).shape == (6,)


@pytest.mark.parametrize("optimizer", ["adam", "rmsprop", "adadelta"])
def test_optimizer_formulas_against_torch(optimizer):
    """apply_optimizer (TF 1.8 training_ops arithmetic) against torch.optim on a toy problem.  The two
    families differ only in where epsilon sits (TF: sqrt(v) + eps before the bias correction for Adam,
    sqrt(ms + eps) for RMSProp) and in RMSProp's initial accumulator (TF: ones), so with gradients far
    above epsilon and the accumulator preset they agree to ~1e-7."""
    import torch
    rng = np.random.default_rng(5)
    p = {"a": rng.normal(size=(7, 3)), "item_b": rng.normal(size=9)}
    st = orc.init_opt_state(p, optimizer)
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    lr = {"adam": 0.05, "rmsprop": 0.02, "adadelta": 1.0}[optimizer]
    hp = orc.OPT_DEFAULTS[optimizer]
    if optimizer == "adam":
        opt = torch.optim.Adam(tp.values(), lr=lr, betas=(hp["beta1"], hp["beta2"]), eps=hp["epsilon"])
    elif optimizer == "rmsprop":
        opt = torch.optim.RMSprop(tp.values(), lr=lr, alpha=hp["decay"], eps=hp["epsilon"], momentum=0.0)
        for t in tp.values():           # create the state, then preset it to TF's initial value (ones)
            t.grad = torch.zeros_like(t)
        opt.step()
        for t in tp.values():
            opt.state[t]["square_avg"].fill_(1.0)
    else:
        opt = torch.optim.Adadelta(tp.values(), lr=lr, rho=hp["rho"], eps=hp["epsilon"])
    q = dict(p)
    for step in range(5):
        g = {k: rng.normal(size=v.shape) * 0.3 + 0.5 for k, v in p.items()}
        q = orc.apply_optimizer(q, g, lr, optimizer, st)
        for k, t in tp.items():
            t.grad = torch.tensor(g[k])
        opt.step()
        for k in p:
            assert np.abs(q[k] - tp[k].detach().numpy()).max() < 1e-6, (optimizer, step, k)
    assert st["t"] == 5


def test_sparse_optimizers_touch_only_gathered_item_b_rows():
    """item_b reaches the optimizer as IndexedSlices of the candidate rows only: the sparse RMSProp /
    Adadelta kernels leave every other row and its accumulators alone, Adam's sparse form does not."""
    rng = np.random.default_rng(6)
    p = {"item_b": rng.normal(size=6)}
    g = {"item_b": np.array([0.3, 0.0, 0.0, -0.2, 0.0, 0.0])}
    used = g["item_b"] != 0
    for optimizer in ("rmsprop", "adadelta"):
        st = orc.init_opt_state(p, optimizer)
        q = orc.apply_optimizer(p, g, 0.1, optimizer, st, used)
        assert np.array_equal(q["item_b"][~used], p["item_b"][~used])
        assert np.array_equal(st["slot1"]["item_b"][~used], orc.init_opt_state(p, optimizer)["slot1"]["item_b"][~used])
        assert np.all(q["item_b"][used] != p["item_b"][used])
    st = orc.init_opt_state(p, "adam")
    q = orc.apply_optimizer(p, g, 0.1, "adam", st, used)
    st["slot1"]["item_b"][:] = 0.5   # a row with momentum keeps moving without a gradient
    q2 = orc.apply_optimizer(q, {"item_b": np.zeros(6)}, 0.1, "adam", st, np.zeros(6, bool))
    assert np.all(q2["item_b"] != q["item_b"])


@pytest.mark.parametrize("d,H,Ls,Sn", [(64, 8, 10, 3), (128, 8, 6, 0)])
def test_dropout_backward_against_autograd(d, H, Ls, Sn):
    """dropout > 0 (model.py:428-431): with the same keep / drop pattern, the oracle's manual gradients
    equal autograd's of the torch restatement; the pattern keeps about keep_prob of the elements,
    scales the kept ones by 1 / keep_prob and is a pure function of (seed, sample, position, channel)."""
    cfg = make_config(d=d, H=H, Ls=Ls)
    p = random_params(cfg, seed=5)
    b, cat = random_batch(cfg, B=7, Sn=Sn, seed=6)
    rate, seed, reg = 0.3, 0xC0FFEE, 5e-3
    loss, logits, g, _ = orc.backward(p, cat, b, H, reg, dropout=(rate, seed))
    ks = [orc.dropout_scale(rate, seed, 7, Ls, d, 0, 0), orc.dropout_scale(rate, seed, 7, Ls, d, 0, 1),
          orc.dropout_scale(rate, seed, 7, Sn + 1, d, 1, 0), orc.dropout_scale(rate, seed, 7, Sn + 1, d, 1, 1)]
    tl, tlog, tg = tref.grads(tref.params_to_torch(p), cat, tref.batch_to_torch(b), H, reg, [torch.tensor(k) for k in ks])
    assert abs(loss - float(tl)) < TOL and np.abs(logits - tlog.numpy()).max() < TOL
    for k in p:
        assert np.abs(g[k] - tg[k].numpy()).max() < TOL, k
    l0 = orc.backward(p, cat, b, H, reg)[0]
    assert abs(l0 - loss) > 1e-6                                   # it does something
    big = orc.dropout_scale(rate, seed, 64, 16, 128, 0, 0)
    assert set(np.unique(big)) == {0.0, float(1.0 / np.float32(0.7))}
    assert abs((big > 0).mean() - 0.7) < 0.01
    assert np.array_equal(big[5:9], orc.dropout_scale(rate, seed, 4, 16, 128, 0, 0, sample0=5))
    assert not np.array_equal(big, orc.dropout_scale(rate, seed + 1, 64, 16, 128, 0, 0))


def test_oracle_on_the_touched_rows_is_the_oracle_on_the_full_tables():
    """Id compaction (tests/helpers.compact_problem; what lets the C4 / C5 GPU tests meet the oracle at sizes numpy
    cannot hold): one train step on the rows a batch touches, with the rest of the regularised tables entering through
    their sum of squares only (`l2_extra`), reproduces the step on the full tables -- loss, clip norm, every touched row,
    every dense parameter; rows the batch does not touch decay by the dense L2 gradient (model.py:164-172, 198-205)."""
    from tests.helpers import compact_problem
    cfg = make_config(U=400, I=900, C=700, d=64, Ls=14, regulation_rate=2e-3, max_gradient_norm=0.05)
    p = random_params(cfg, seed=21)
    b, cat = random_batch(cfg, B=24, Sn=3, seed=22)
    tup = (b["u"], b["i"], b["y"], b["hist_i"], b["hist_i_new"], b["hist_t"], b["sl"], b["sl_new"], b["u_cate"])
    for mode in ("tf18", "dedup"):
        loss, newp, info = orc.train_step(p, cat, b, 8, cfg["regulation_rate"], lr=0.9, clip=0.05, norm_mode=mode)
        assert info["coef"] < 1.0
        cp = compact_problem(tup, cat)
        sel = dict(item_emb=cp["items"], item_b=cp["items"], user_emb=cp["users"], usert_emb=cp["users"], cate_emb=cp["cates"])
        q = {k: (p[k][sel[k]] if k in sel else p[k]) for k in p}
        extra = sum(float((p[k] ** 2).sum() - (q[k] ** 2).sum()) for k in orc.REG_TABLES)
        closs, cnew, cinfo = orc.train_step(q, cp["item_cate"], orc.as_batch(cp["batch"]), 8, cfg["regulation_rate"], lr=0.9,
                                            clip=0.05, norm_mode=mode, l2_extra=extra)
        assert abs(closs - loss) < 1e-12 * max(1.0, abs(loss)) and abs(cinfo["norm"] - info["norm"]) < 1e-12 * info["norm"]
        for k in p:
            ref = newp[k][sel[k]] if k in sel else newp[k]
            assert np.abs(cnew[k] - ref).max() < 1e-13, (mode, k)
        decay = 1.0 - 0.9 * info["coef"] * cfg["regulation_rate"]
        for k in orc.REG_TABLES:
            rest = np.setdiff1d(np.arange(p[k].shape[0]), sel[k])
            assert len(rest) > 0 and np.abs(newp[k][rest] - decay * p[k][rest]).max() < 1e-15, k
        rest = np.setdiff1d(np.arange(cfg["item_count"]), cp["items"])
        assert np.array_equal(newp["item_b"][rest], p["item_b"][rest])          # item_b is not regularised (model.py:164-169)
