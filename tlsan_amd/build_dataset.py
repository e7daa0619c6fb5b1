"""Sample builder: counterpart of the reference's ``TLSAN/build_dataset.py``.

Input: the remapped review log, one row per review -- (reviewerID, asin, unixReviewTime in days) --
plus ``item_cate_list``.  Output: the samples ``dataset.pkl`` holds, **identical to the reference's,
sample for sample, in its shuffled order** (tests/test_build_dataset.py: seven datasets).

The reference walks every user's reviews in nested Python loops (7-60 s per dataset, most of it in
per-item ``meta_df`` scans and ``pd.value_counts`` calls).  Here the whole log is sessionised at once
with array operations and the samples come out as flat CSR arrays (:class:`tlsan_amd.input.PackedSet`,
what the batchers consume):

* **sessions** (:39-47): a user's distinct days in ascending order with their multiplicities; session
  ``k`` is the positional slice ``[lo_k, hi_k)`` of the user's rows (``lo`` = running sum of the
  multiplicities) -- one lexsort + run-length encoding for all users together;
* **roles** (:49-72): with ``VL = min(len, 90)``, a session is a *training* session while
  ``hi_k < VL - 1`` (k >= 1), and the first later one is the user's *test* session; ``hi`` grows with
  ``k``, so the test session is number ``max(1, #{k : hi_k < VL - 1})`` -- a per-user count, no walk;
* **samples**: a training session gives a positive and a negative sample (target = the row right
  after the session / the negative drawn for that row) whose history is every row before the session;
  histories, sessions and time weights ``1 / #{g in 2,4,..,4096 : days + 1 >= g}`` (:16-21) are
  gathered with repeat/arange index arithmetic; the "current category" (:54) is the mode of the
  history's categories, found from one sort of (sample, category) pairs -- exact ties (rare) are left
  to pandas' own ``value_counts`` so that the result is the reference's on the same pandas.

What cannot be array code is the pseudo-random stream: negatives (:29-34, a rejection loop per row),
the held-out item of a multi-item test session (:66) and the two final shuffles (:75-76) all draw
from Python's ``random`` seeded with 1234 (:8).  They are drawn here from a private
``random.Random(1234)`` in the same order -- one light pass over the users, then two index shuffles.

    python -m tlsan_amd.build_dataset reviews.npz dataset.pkl      # npz: reviewerID, asin, unixReviewTime,
                                                                   #      item_cate_list, counts
"""
from __future__ import annotations

import pickle
import random
import sys

import numpy as np

from .input import PackedSet

MAX_LENGTH = 90  # build_dataset.py:7
GAP = np.array([2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096])  # :14


def proc_time_emb(hist_t, cur_t):
    """build_dataset.py:16-21 for one history (numpy float64 results, as there)."""
    return list(time_weights(np.asarray(hist_t, np.int64), cur_t))


def time_weights(days, cur):
    """1 / #{g in GAP : cur - day + 1 >= g}, elementwise (float64)."""
    delta = np.asarray(cur, np.int64) - days + 1
    with np.errstate(divide="ignore"):
        return 1.0 / (delta[..., None] >= GAP).sum(-1)


def _ragged(starts, lengths):
    """Flat indices of the ranges [starts[k], starts[k] + lengths[k]) laid end to end, and the offsets
    (CSR) of the ranges in that flat array."""
    lengths = np.asarray(lengths, np.int64)
    off = np.zeros(len(lengths) + 1, np.int64)
    np.cumsum(lengths, out=off[1:])
    owner = np.repeat(np.arange(len(lengths), dtype=np.int64), lengths)
    idx = np.arange(off[-1], dtype=np.int64) - off[owner] + np.asarray(starts, np.int64)[owner]
    return idx, off, owner


def _modes(owner, values, n_groups, ordered_lists):
    """Most frequent value of every group (``pd.value_counts(x).index[0]``, :54).  ordered_lists(g) gives
    group g's values in their original order: only called for an exact tie, which pandas' sort decides."""
    if n_groups == 0:
        return np.zeros(0, np.int64)
    vmax = int(values.max()) + 1 if len(values) else 1
    key = owner * vmax + values
    uk, cnt = np.unique(key, return_counts=True)
    g, v = uk // vmax, uk % vmax
    best = np.zeros(n_groups, np.int64)
    np.maximum.at(best, g, cnt)
    top = cnt == best[g]
    n_top = np.bincount(g[top], minlength=n_groups)
    out = np.full(n_groups, -1, np.int64)
    out[g[top]] = v[top]                      # (unique maximum: the one writer)
    ties = np.flatnonzero(n_top > 1)
    if len(ties):
        import pandas as pd
        for gi in ties:
            out[gi] = pd.Series(ordered_lists(int(gi))).value_counts().index[0]
    return out


def build_packed(reviewer, asin, when, item_cate_list, item_count, seed=1234, max_length=MAX_LENGTH):
    """-> (train PackedSet, test PackedSet) in the reference's (shuffled) sample order.  Rows must be in
    the reference's DataFrame order (its groupby keeps the order inside a user)."""
    rnd = random.Random(seed)
    reviewer = np.asarray(reviewer, np.int64)
    cate_of = np.asarray(item_cate_list, np.int64)
    order = np.argsort(reviewer, kind="stable")                 # groupby('reviewerID'): keys ascending, rows in order
    uid_row, item, day = reviewer[order], np.asarray(asin, np.int64)[order], np.asarray(when, np.int64)[order]
    n = len(uid_row)
    uoff = np.concatenate([[0], np.flatnonzero(uid_row[1:] != uid_row[:-1]) + 1, [n]]).astype(np.int64)
    nU = len(uoff) - 1
    ulen = np.diff(uoff)
    urow = np.repeat(np.arange(nU, dtype=np.int64), ulen)       # user index of every row
    # ---- sessions of all users: distinct (user, day), days ascending, with multiplicities (:39-43)
    by_day = np.lexsort((day, urow))
    su, sd = urow[by_day], day[by_day]
    first = np.flatnonzero(np.concatenate([[True], (su[1:] != su[:-1]) | (sd[1:] != sd[:-1])]))
    s_user, s_day = su[first], sd[first]
    s_cnt = np.diff(np.concatenate([first, [n]]))
    nS = len(first)
    s0 = np.concatenate([[0], np.flatnonzero(s_user[1:] != s_user[:-1]) + 1]).astype(np.int64)   # first session of a user
    run = np.cumsum(s_cnt) - s_cnt
    s_lo = run - run[s0][s_user]                                # positional slice inside the user (:42-44)
    s_hi = s_lo + s_cnt
    s_k = np.arange(nS, dtype=np.int64) - s0[s_user]
    n_sess = np.bincount(s_user, minlength=nU)
    # ---- roles (:49-72)
    VL = np.minimum(ulen, max_length)
    is_early = s_hi < VL[s_user] - 1
    k_test = np.maximum(np.bincount(s_user, weights=is_early, minlength=nU).astype(np.int64), 1)
    if np.any(k_test >= n_sess):
        raise ValueError("a user has no test session (the reference asserts len(test_set) == user_count, :78)")
    tr = np.flatnonzero((s_k >= 1) & (s_k < k_test[s_user]))    # training sessions, in (user, session) order
    te = s0 + k_test                                            # the test session of every user
    # ---- the pseudo-random stream, in the reference's order: per user its negatives (:29-34), then the
    #      held-out item of a multi-item test session (:66)
    neg = np.empty(n, np.int64)
    t_pos = np.empty(nU, np.int64)          # test target ...
    t_neg = np.empty(nU, np.int64)          # ... its negative ...
    t_drop = np.full(nU, -1, np.int64)      # ... and which row of the test session it leaves (-1: none, :65-67)
    items_l, te_lo, te_cnt = item.tolist(), (uoff[:-1] + s_lo[te]).tolist(), s_cnt[te].tolist()
    randint, hi = rnd.randint, item_count - 1
    for u in range(nU):
        a, b = int(uoff[u]), int(uoff[u + 1])
        pos_list = items_l[a:b]
        seen = set(pos_list)
        first_item = pos_list[0]
        out = []
        for _ in pos_list:
            x = first_item
            while x in seen:
                x = randint(0, hi)
            out.append(x)
        neg[a:b] = out
        lo, cnt = te_lo[u], te_cnt[u]
        pos_item = items_l[lo]
        if cnt > 1:
            sess = items_l[lo:lo + cnt]
            pos_item = rnd.choice(sess)
            t_drop[u] = lo + sess.index(pos_item)               # list.remove: the first occurrence
        t_pos[u] = pos_item
        t_neg[u] = out[pos_list.index(pos_item)]                # :68-69
    # ---- training samples: two per training session (label 1, then 0), history = every row before it
    def assemble(sessions, cur_day):
        u_s = s_user[sessions]
        base = uoff[u_s]
        h_idx, h_off, h_owner = _ragged(base, s_lo[sessions])
        n_idx, n_off, _ = _ragged(base + s_lo[sessions], s_cnt[sessions])
        w = time_weights(day[h_idx], cur_day[h_owner]) if len(h_idx) else np.zeros(0)
        cat = _modes(h_owner, cate_of[item[h_idx]], len(sessions),
                     lambda g: cate_of[item[h_idx[h_off[g]:h_off[g + 1]]]].tolist())
        return u_s, h_idx, h_off, n_idx, n_off, w, cat

    u_s, h_idx, h_off, n_idx, n_off, w, cat = assemble(tr, day[uoff[s_user[tr]] + s_lo[tr]])     # tim_list[i] (:56)
    nxt = uoff[u_s] + s_hi[tr]                                  # pos_list[i + count] / neg_list[i + count]
    two = np.repeat(np.arange(len(tr), dtype=np.int64), 2)      # the pair shares history, session and category
    hl, nl = np.diff(h_off)[two], np.diff(n_off)[two]
    hh, hoff2, _ = _ragged(h_off[:-1][two], hl)
    nn, noff2, _ = _ragged(n_off[:-1][two], nl)
    target = np.stack([item[nxt], neg[nxt]], 1).reshape(-1)
    label = np.tile(np.array([1, 0], np.int64), len(tr))
    train = (uid_row[uoff[u_s]][two], hoff2, item[h_idx][hh], w[hh], noff2, item[n_idx][nn], cat[two], target, label)
    # ---- test samples: one per user (:63-72); the held-out item leaves a multi-item session
    u_t, h_idx, h_off, n_idx, n_off, w, cat = assemble(te, s_day[te])                              # t (:70)
    keep = ~np.isin(n_idx, t_drop[t_drop >= 0])
    n_off = np.concatenate([[0], np.cumsum(np.bincount(np.repeat(np.arange(nU), np.diff(n_off))[keep], minlength=nU))])
    test = (uid_row[uoff[u_t]], h_off, item[h_idx], w, n_off, item[n_idx[keep]], cat, t_pos, t_neg)
    # ---- the two final shuffles (:75-76), as index shuffles
    def shuffled(n_samples):
        p = list(range(n_samples))
        rnd.shuffle(p)
        return np.asarray(p, np.int64)

    return _take(train, shuffled(len(train[0])), False), _take(test, shuffled(nU), True)


def _take(cols, perm, is_test):
    """Reorder CSR samples by `perm` -> PackedSet."""
    u, h_off, hist, w, n_off, sess, cat, a, b = cols
    hi, ho, _ = _ragged(h_off[:-1][perm], np.diff(h_off)[perm])
    ni, no, _ = _ragged(n_off[:-1][perm], np.diff(n_off)[perm])
    # (time weights are 1/k, k = 1..12, in float64; the batcher stores them into a float32 array, input.py:35,43)
    kw = dict(pos=a[perm], neg=b[perm]) if is_test else dict(target=a[perm], label=b[perm])
    return PackedSet(u[perm], ho, hist[hi], w[hi].astype(np.float32), no, sess[ni], cat[perm], **kw)


def to_samples(ps):
    """A PackedSet as the python list of tuples ``dataset.pkl`` holds (:58-59 train, :71 test)."""
    out = []
    ho, so = ps.hist_off.tolist(), ps.sess_off.tolist()
    hist, sess = ps.hist.tolist(), ps.sess.tolist()
    w = (1.0 / np.rint(1.0 / ps.hist_t.astype(np.float64))).tolist()     # back to the float64 1/k the reference stores
    u, c = ps.u.tolist(), ps.cate.tolist()
    for k in range(len(u)):
        h, s, t = hist[ho[k]:ho[k + 1]], sess[so[k]:so[k + 1]], w[ho[k]:ho[k + 1]]
        if ps.is_test:
            out.append((u[k], h, s, t, (int(ps.pos[k]), int(ps.neg[k])), c[k]))
        else:
            out.append((u[k], h, s, t, int(ps.target[k]), int(ps.label[k]), c[k]))
    return out


def build_dataset(reviewer, asin, when, item_cate_list, item_count, seed=1234, max_length=MAX_LENGTH):
    """-> (train_set, test_set) as lists of the reference's tuples."""
    train, test = build_packed(reviewer, asin, when, item_cate_list, item_count, seed, max_length)
    return to_samples(train), to_samples(test)


def save_packed(path, train, test, counts, item_cate_list):
    """``packed_<name>.npz``, the export ``tlsan_amd.input.load_packed`` / ``train.py --dataset`` read."""
    cols = {}
    for prefix, ps in (("train_", train), ("test_", test)):
        for k in ("u", "hist_off", "hist", "hist_t", "sess_off", "sess", "cate"):
            cols[prefix + k] = getattr(ps, k)
        for k in (("pos", "neg") if ps.is_test else ("target", "label")):
            cols[prefix + k] = getattr(ps, k)
    np.savez_compressed(path, counts=np.asarray(counts, np.int64), item_cate_list=np.asarray(item_cate_list, np.int32), **cols)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        raise SystemExit(__doc__)
    z = np.load(argv[0])
    counts = tuple(int(x) for x in z["counts"][:3])
    train, test = build_packed(z["reviewerID"], z["asin"], z["unixReviewTime"], z["item_cate_list"], counts[1])
    assert len(test) == counts[0]                     # build_dataset.py:78
    if argv[1].endswith(".npz"):
        save_packed(argv[1], train, test, counts, z["item_cate_list"])
    else:
        with open(argv[1], "wb") as f:                # :80-84
            pickle.dump(to_samples(train), f, pickle.HIGHEST_PROTOCOL)
            pickle.dump(to_samples(test), f, pickle.HIGHEST_PROTOCOL)
            pickle.dump(counts, f, pickle.HIGHEST_PROTOCOL)
            pickle.dump(np.asarray(z["item_cate_list"]), f, pickle.HIGHEST_PROTOCOL)
    print("%d train / %d test samples -> %s" % (len(train), len(test), argv[1]))


if __name__ == "__main__":
    main()
