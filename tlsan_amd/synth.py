"""Synthetic Amazon-shaped workloads, seed 1234 everywhere.

The three large datasets named by BASELINE.json (Electronics, Movies-TV, CDs) are not shipped
with the reference (Data/ holds 7 of 10), so throughput configs use synthetic sequences with
the measured statistics of the shipped ones:
  u ~ U[0,U); item ids Zipf-like p(rank) ~ rank^-0.5 over a fixed random permutation of [0,I);
  item_cate_list ~ U[0,C); u_cate ~ U[0,C); y alternating 1/0 (train sets are exactly 50/50,
  build_dataset.py:58-59); history length n_pre = min(90, ceil(LogNormal(2.2, 1.0))),
  sl = min(n_pre, Ls) with input.py:41-49's alignment; sl_new = min(18, Geometric(0.82));
  hist_t[l] = 1/k_l with k_l in 1..12 non-increasing in l (build_dataset.py:18-21).
"""
from __future__ import annotations

import numpy as np

CONFIGS = {
    # name: (users, items, cates, hidden, batch)  -- BASELINE.json `configs`, README.md:15-27
    "clothing": (2010, 1723, 226, 64, 32),
    "digital_music": (1659, 1583, 53, 128, 1024),
    "electronics": (39991, 22048, 673, 128, 4096),
    "movies_tv": (35896, 28589, 15, 128, 4096),
}


def make_config(name, Ls=10, **over):
    U, I, C, d, B = CONFIGS[name]
    cfg = dict(user_count=U, item_count=I, cate_count=C, hidden_units=d, num_heads=8, Ls=Ls,
               itemid_embedding_size=d // 2, userid_embedding_size=d // 2, cateid_embedding_size=d // 2,
               num_blocks=1, dropout=0.0, regulation_rate=5e-5, optimizer="sgd", max_gradient_norm=5.0,
               learning_rate=1.0, train_batch_size=B, test_batch_size=128, model_dir="save_path")
    cfg.update(over)
    return cfg


class ItemSampler:
    def __init__(self, I, rng, alpha=0.5):
        p = np.arange(1, I + 1, dtype=np.float64) ** (-alpha)
        self.cdf = np.cumsum(p / p.sum())
        self.perm = rng.permutation(I)

    def draw(self, rng, size):
        r = np.searchsorted(self.cdf, rng.random(size), side="right")
        return self.perm[np.minimum(r, len(self.perm) - 1)]


def item_cate_list(cfg, seed=1234):
    rng = np.random.default_rng(seed)
    return rng.integers(0, cfg["cate_count"], cfg["item_count"]).astype(np.int32)


# session lengths of the Digital-Music training set (tests/golden/packed_digital_music.npz, 37 970 samples):
# counts of lengths 1 .. 12 (12 = "12 or more")
_AMAZON_SESSION_HIST = np.array([33124, 3366, 848, 324, 150, 50, 44, 26, 16, 8, 0, 14], np.float64)


def make_batches(cfg, n_batches, batch_size=None, seed=1234, test=False, sessions="geometric"):
    """List of 9-tuples in the layout of TLSAN/input.py:54 (train) / :107 (test).
    sessions: "geometric" (the bench's distribution since round 1), "amazon" (the empirical, longer-tailed one) or
    "capN" (diagnostic: geometric, cut at N entries)."""
    rng = np.random.default_rng(seed + (1 if test else 0))
    U, I, C, Ls = cfg["user_count"], cfg["item_count"], cfg["cate_count"], cfg["Ls"]
    B = batch_size or cfg["train_batch_size"]
    sampler = ItemSampler(I, np.random.default_rng(seed))
    out = []
    for _ in range(n_batches):
        u = rng.integers(0, U, B)
        n_pre = np.minimum(90, np.ceil(rng.lognormal(2.2, 1.0, B))).astype(np.int64)
        sl = np.minimum(n_pre, Ls)
        if sessions == "amazon":
            sl_new = 1 + rng.choice(12, B, p=_AMAZON_SESSION_HIST / _AMAZON_SESSION_HIST.sum())
        else:
            sl_new = np.minimum(18, rng.geometric(0.82, B))
            if sessions.startswith("cap"):   # (diagnostic: the geometric lengths cut at N -- what the long sessions cost a launch)
                sl_new = np.minimum(sl_new, int(sessions[3:]))
        ar = np.arange(Ls)[None, :]
        valid = ar < sl[:, None]
        hist_i = np.where(valid, sampler.draw(rng, (B, Ls)), 0).astype(np.int64)
        k = np.sort(rng.integers(1, 13, (B, Ls)), axis=1)[:, ::-1]          # non-increasing
        # left-aligned valid slots get the LAST sl entries' ordering (oldest first)
        hist_t = np.where(valid, (1.0 / k).astype(np.float32), np.float32(0)).astype(np.float32)
        Sn = int(sl_new.max())
        valid2 = np.arange(Sn)[None, :] < sl_new[:, None]
        hist_i_new = np.where(valid2, sampler.draw(rng, (B, Sn)), 0).astype(np.int64)
        c = rng.integers(0, C, B)
        i = sampler.draw(rng, B)
        if test:
            yj = rng.integers(0, I, B)
        else:
            yj = (np.arange(B) % 2 == 0).astype(np.int64)                    # 1,0,1,0,...
        out.append((u, i, yj, hist_i, hist_i_new, hist_t, sl, sl_new, c))
    return out


def algorithmic_bytes(cfg, batch, elem_bytes=4):
    """logical bytes per batch, every gathered/scattered row counted once per use,
    valid positions only.  Returns dict(S_rows, S_idx, fwd, fwd_bwd_kernel, train_step)."""
    u, i, yj, hist_i, hist_i_new, hist_t, sl, sl_new, c = batch
    di, dc, Ls = cfg["itemid_embedding_size"], cfg["cateid_embedding_size"], cfg["Ls"]
    R = np.asarray(sl, np.int64) + np.asarray(sl_new, np.int64) + 1
    S_rows = (R * di * elem_bytes + (R + 1) * dc * elem_bytes + di * elem_bytes + Ls * 4 + 4).sum()
    S_idx = (4 * R + 8 * Ls + 4 * np.asarray(sl_new, np.int64) + 24).sum()
    return dict(S_rows=int(S_rows), S_idx=int(S_idx), fwd=int(S_rows + S_idx),
                fwd_bwd_kernel=int(2 * S_rows + S_idx), train_step=int(3 * S_rows + S_idx))


def algorithmic_flops(cfg, batch):
    """train FLOPs of the fused kernel = 3 * (4*d*dh*P + 2*d^2) per sequence with
    P = n_l + n_s + 1 positions (two dh x dh maps per head per position + the d x d bridge; x3
    for forward, dX and dW), valid positions only, no recomputation counted."""
    u, i, yj, hist_i, hist_i_new, hist_t, sl, sl_new, c = batch
    d, dh = cfg["hidden_units"], cfg["hidden_units"] // cfg["num_heads"]
    Pn = np.asarray(sl, np.int64) + np.asarray(sl_new, np.int64) + 1
    return int((3 * (4 * d * dh * Pn + 2 * d * d)).sum())


def dense_sweep_bytes(cfg, elem_bytes=4):
    """The reference's full-table read+write per step in l2_mode=dense."""
    U, I, C = cfg["user_count"], cfg["item_count"], cfg["cate_count"]
    di, dc, Ls = cfg["itemid_embedding_size"], cfg["cateid_embedding_size"], cfg["Ls"]
    return 2 * elem_bytes * (U * di + I * di + C * dc) + 2 * 4 * U * Ls
