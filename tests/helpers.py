"""Shared test helpers: seeded configs, random batches, fixture batches."""
import os

import numpy as np

from oracle import tlsan_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def make_config(U=50, I=40, C=7, d=64, H=8, Ls=10, di=None, **kw):
    di = d // 2 if di is None else di
    cfg = dict(user_count=U, item_count=I, cate_count=C, hidden_units=d, num_heads=H, Ls=Ls,
               itemid_embedding_size=di, userid_embedding_size=di, cateid_embedding_size=d - di,
               num_blocks=1, dropout=0.0, regulation_rate=5e-5, optimizer="sgd",
               max_gradient_norm=5.0, model_dir="/tmp/tlsan_test_model")
    cfg.update(kw)
    return cfg


def random_params(cfg, seed=0, scale_bias=True, dtype=np.float64):
    """Oracle params with every bias / gamma / usert perturbed so no gradient path is
    trivially zero (the reference's init has zero biases and usert=-1)."""
    p = orc.init_params(cfg, seed=seed, dtype=np.float64)
    rng = np.random.RandomState(seed + 99)
    if scale_bias:
        for k in p:
            if k.endswith("_b1") or k.endswith("_b2") or k in ("dense_b", "item_b"):
                p[k] = rng.uniform(-0.3, 0.3, p[k].shape)
        p["gamma"] = np.array(1.3)
        p["usert_emb"] = rng.uniform(-1.5, -0.5, p["usert_emb"].shape)
        # larger tables so attention is not degenerate
        for k in ("item_emb", "user_emb", "cate_emb"):
            p[k] = rng.uniform(-0.8, 0.8, p[k].shape)
        for k in p:
            if k.endswith("_W1") or k.endswith("_W2"):
                p[k] = rng.uniform(-0.7, 0.7, p[k].shape)
    return {k: np.asarray(v, dtype) for k, v in p.items()}


def random_batch(cfg, B, Sn, seed=0, test=False, full=False):
    """Random batch with the padding conventions of TLSAN/input.py (zeros past the lengths)."""
    rng = np.random.RandomState(seed)
    U, I, C, Ls = cfg["user_count"], cfg["item_count"], cfg["cate_count"], cfg["Ls"]
    sl = rng.randint(1, Ls + 1, B)
    sl_new = rng.randint(0 if not full else Sn, Sn + 1, B)
    if Sn > 0 and sl_new.max() < Sn:
        sl_new[rng.randint(B)] = Sn
    if full:
        sl[:] = Ls
    hist_i = rng.randint(0, I, (B, Ls))
    hist_t = (1.0 / rng.randint(1, 13, (B, Ls))).astype(np.float32)
    hist_i_new = rng.randint(0, I, (B, Sn))
    ar = np.arange(Ls)[None, :]
    hist_i = np.where(ar < sl[:, None], hist_i, 0)
    hist_t = np.where(ar < sl[:, None], hist_t, 0).astype(np.float32)
    hist_i_new = np.where(np.arange(Sn)[None, :] < sl_new[:, None], hist_i_new, 0)
    b = dict(u=rng.randint(0, U, B), i=rng.randint(0, I, B), hist_i=hist_i, hist_i_new=hist_i_new,
             hist_t=hist_t, sl=sl, sl_new=sl_new, u_cate=rng.randint(0, C, B))
    if test:
        b["j"] = rng.randint(0, I, B)
    else:
        b["y"] = rng.randint(0, 2, B).astype(np.float32)
    b = {k: (np.asarray(v, np.int64) if np.asarray(v).dtype.kind in "iu" else v) for k, v in b.items()}
    cat = rng.randint(0, C, I).astype(np.int32)
    return b, cat


def fixture_batch(name, cls_name="DataInput", bs=32, k=10, bi=0):
    fx = np.load(os.path.join(GOLDEN, "batches_%s.npz" % name))
    pre = "%s_bs%d_k%d_b%d_" % (cls_name, bs, k, bi)
    keys = ["u", "i", "yj", "hist_i", "hist_i_new", "hist_t", "sl", "new_sl", "c"]
    batch = tuple(fx[pre + key] for key in keys)
    pk = np.load(os.path.join(GOLDEN, "packed_%s.npz" % name))
    counts = tuple(int(x) for x in pk["counts"])
    return batch, counts, pk["item_cate_list"].astype(np.int32)


def compact_problem(batch, item_cate_list):
    """Id compaction: the model depends on its tables only through the rows a batch gathers plus the tables' sum of
    squares (the L2 term of TLSAN/model.py:164-172 and its gradient reg * W).  Returns the ids the batch touches --
    `items` (candidates, valid window and session entries, and id 0: padded slots gather row 0, input.py:47-51), `users`,
    `cates` (of those items and the u_cate column), all sorted -- the batch with its ids renumbered into those lists,
    and the renumbered item -> category map.  Running the oracle on tables restricted to these rows, with the sum of
    squares of the rows left out as `l2_extra`, is the oracle on the full tables (tests/test_oracle.py checks that on a
    case numpy can hold); the C4 / C5 tests use it where it cannot."""
    u, i, y, hist_i, hist_i_new, hist_t, sl, sl_new, c = [np.asarray(x) for x in batch]
    B, Ls = hist_i.shape
    hist_i_new = hist_i_new.reshape(B, -1)
    vl = np.arange(Ls)[None, :] < sl[:, None]
    vs = np.arange(hist_i_new.shape[1])[None, :] < sl_new[:, None]
    assert (hist_i[~vl] == 0).all() and (hist_i_new[~vs] == 0).all(), "padded slots hold id 0"
    icl = np.asarray(item_cate_list, np.int64)
    items = np.unique(np.concatenate([i.ravel(), hist_i[vl], hist_i_new[vs], [0]])).astype(np.int64)
    users = np.unique(u).astype(np.int64)
    cates = np.unique(np.concatenate([icl[items], c.ravel()])).astype(np.int64)
    ri = lambda x: np.searchsorted(items, x)
    cb = (np.searchsorted(users, u), ri(i), y, ri(hist_i), ri(hist_i_new), hist_t, sl, sl_new, np.searchsorted(cates, c))
    return dict(items=items, users=users, cates=cates, batch=cb, item_cate=np.searchsorted(cates, icl[items]).astype(np.int32))
