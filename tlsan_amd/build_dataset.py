"""Sample builder: counterpart of the reference's ``TLSAN/build_dataset.py``.

From the remapped review log -- one row per review: (reviewerID, asin, unixReviewTime in days),
sorted by user then time -- it produces the sample tuples ``dataset.pkl`` holds
(build_dataset.py:58-59 train, :71 test), **identical to the reference's, tuple for tuple**:
sessions are the runs of reviews of one day (:39-47), every session after the first yields a
positive and a negative training sample whose history is everything before it (:55-62), the
first session that reaches the end of the (length-capped) history becomes the user's test sample
(:63-72), time weights are ``1 / #{g in 2,4,...,4096 : days + 1 >= g}`` (:16-21), the user's
current category is the most frequent category of the history (:54).

Identity needs the same pseudo-random stream: negatives (:29-34), the held-out item of a
multi-item last session (:66) and the two final shuffles (:75-76) all draw from Python's
``random`` seeded with 1234 (:8), so they are drawn here in the same order from a private
``random.Random(1234)``.  ``pd.value_counts`` decides ties of the category count (:54) by its
sort; ties are delegated to pandas itself, everything else is plain Python.  The reference spends
most of its 19 s in per-item ``meta_df`` scans and ``pd.value_counts`` calls; this takes ~1 s.

    python -m tlsan_amd.build_dataset reviews.npz dataset.pkl      # npz: reviewerID, asin, unixReviewTime,
                                                                   #      item_cate_list, counts
"""
from __future__ import annotations

import pickle
import random
import sys
from collections import Counter

import numpy as np

MAX_LENGTH = 90  # build_dataset.py:7
GAP = np.array([2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096])  # :14


def proc_time_emb(hist_t, cur_t):
    """build_dataset.py:16-21 (numpy float64 results, as there)."""
    return [1 / np.sum((cur_t - h + 1) >= GAP) for h in hist_t]


def _now_cate(pre_cates, counter):
    """``pd.value_counts(pre_cates).index[0]`` (:54).  A unique maximum needs no pandas; ties are
    broken by pandas' own sort so that the result is the reference's on the same pandas."""
    best = max(counter.values())
    top = [c for c, n in counter.items() if n == best]
    if len(top) == 1:
        return top[0]
    import pandas as pd
    return pd.Series(pre_cates).value_counts().index[0]


def build_dataset(reviewer, asin, when, item_cate_list, item_count, seed=1234, max_length=MAX_LENGTH):
    """-> (train_set, test_set): lists of the reference's tuples.  Rows must be in the reference's
    DataFrame order (the groupby keeps the order inside a user)."""
    rnd = random.Random(seed)
    reviewer = np.asarray(reviewer, np.int64)
    asin = np.asarray(asin, np.int64)
    when = np.asarray(when, np.int64)
    cate = np.asarray(item_cate_list, np.int64)
    order = np.argsort(reviewer, kind="stable")       # groupby('reviewerID'): keys ascending, rows in order
    bounds = np.flatnonzero(np.diff(reviewer[order])) + 1
    starts = np.concatenate([[0], bounds, [len(order)]])
    train_set, test_set = [], []
    for g in range(len(starts) - 1):
        idx = order[starts[g]:starts[g + 1]]
        uid = int(reviewer[idx[0]])
        pos_list = asin[idx].tolist()
        tim_list = when[idx].tolist()
        pos_set = set(pos_list)
        neg_list = []
        for _ in pos_list:                            # :29-34
            neg = pos_list[0]
            while neg in pos_set:
                neg = rnd.randint(0, item_count - 1)
            neg_list.append(neg)
        length = len(pos_list)
        valid_length = min(length, max_length)
        i = 0
        tcount = Counter(tim_list)
        sessions = sorted(tcount)
        pre_session, pre_time, pre_cates = [], [], []
        ccount = Counter()
        for t in sessions:
            count = tcount[t]                         # tim_list.count(t)
            new_session = pos_list[i:i + count]
            new_time = tim_list[i:i + count]
            new_cates = [int(cate[item]) for item in new_session]   # meta_df lookup (:47)
            if t == sessions[0]:
                pre_session.extend(new_session)
                pre_time.extend(new_time)
                pre_cates.extend(new_cates)
                ccount.update(new_cates)
            else:
                now_cate = _now_cate(pre_cates, ccount)
                if i + count < valid_length - 1:
                    pre_time_emb = proc_time_emb(pre_time, tim_list[i])
                    pre_session_copy = list(pre_session)
                    train_set.append((uid, pre_session_copy, new_session, pre_time_emb, pos_list[i + count], 1, now_cate))
                    train_set.append((uid, pre_session_copy, new_session, pre_time_emb, neg_list[i + count], 0, now_cate))
                    pre_session.extend(new_session)
                    pre_time.extend(new_time)
                    pre_cates.extend(new_cates)
                    ccount.update(new_cates)
                else:
                    pos_item = pos_list[i]
                    if count > 1:                     # the target stays in the session otherwise (a quirk kept)
                        pos_item = rnd.choice(new_session)
                        new_session.remove(pos_item)
                    neg_index = pos_list.index(pos_item)
                    pos_neg = (pos_item, neg_list[neg_index])
                    pre_time_emb = proc_time_emb(pre_time, t)
                    test_set.append((uid, pre_session, new_session, pre_time_emb, pos_neg, now_cate))
                    break
            i += count
    rnd.shuffle(train_set)                            # :75-76
    rnd.shuffle(test_set)
    return train_set, test_set


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        raise SystemExit(__doc__)
    z = np.load(argv[0])
    counts = tuple(int(x) for x in z["counts"][:3])
    train_set, test_set = build_dataset(z["reviewerID"], z["asin"], z["unixReviewTime"], z["item_cate_list"], counts[1])
    assert len(test_set) == counts[0]                 # build_dataset.py:78
    with open(argv[1], "wb") as f:                    # :80-84
        pickle.dump(train_set, f, pickle.HIGHEST_PROTOCOL)
        pickle.dump(test_set, f, pickle.HIGHEST_PROTOCOL)
        pickle.dump(counts, f, pickle.HIGHEST_PROTOCOL)
        pickle.dump(np.asarray(z["item_cate_list"]), f, pickle.HIGHEST_PROTOCOL)
    print("%d train / %d test samples -> %s" % (len(train_set), len(test_set), argv[1]))


if __name__ == "__main__":
    main()
