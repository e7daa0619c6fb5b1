"""f4: tlsan_amd.build_dataset against what the reference's real TLSAN/build_dataset.py
built from the same review logs (tests/golden/make_fixtures.py, run in the build container):
  * Clothing and Digital-Music: every train and test sample, in the shuffled order, bit for bit
    (tests/golden/packed_<name>.npz);
  * the other five datasets the reference ships (README.md:36,38-41): sha256 of every array of the CSR
    export, sizes and the first samples (tests/golden/digest_<name>.json)."""
import hashlib
import json
import os

import numpy as np
import pytest

from tlsan_amd.build_dataset import build_dataset, build_packed, proc_time_emb, to_samples
from tlsan_amd.input import PackedSet, load_packed

TRAIN = ["u", "hist_off", "hist", "sess_off", "sess", "cate", "target", "label"]
TEST = ["u", "hist_off", "hist", "sess_off", "sess", "cate", "pos", "neg"]


def _build(golden_dir, name):
    z = np.load(os.path.join(golden_dir, "reviews_%s.npz" % name))
    U, I, C = (int(x) for x in z["counts"][:3])
    train, test = build_packed(z["reviewerID"], z["asin"], z["unixReviewTime"], z["item_cate_list"], I)
    assert len(test) == U                                   # build_dataset.py:78
    return z, (U, I, C), train, test


@pytest.mark.parametrize("name", ["clothing", "digital_music"])
def test_samples_identical_to_reference(golden_dir, name):
    z, (U, I, C), train, test = _build(golden_dir, name)
    ref_train, ref_test, counts, icl = load_packed(os.path.join(golden_dir, "packed_%s.npz" % name))
    assert counts == (U, I, C) and np.array_equal(icl, z["item_cate_list"])
    for got, ref in ((train, ref_train), (test, ref_test)):
        assert len(got) == len(ref)
        for k in (TEST if ref.is_test else TRAIN):
            assert np.array_equal(getattr(got, k), getattr(ref, k)), k
        assert np.array_equal(got.hist_t.view(np.uint32), ref.hist_t.view(np.uint32))


@pytest.mark.parametrize("name", ["beauty", "home_kitchen", "office", "toys", "video_games"])
def test_digest_of_reference_build(golden_dir, name):
    path = os.path.join(golden_dir, "digest_%s.json" % name)
    if not os.path.exists(path):
        pytest.skip("digest not generated")
    want = json.load(open(path))
    z, counts, train, test = _build(golden_dir, name)
    assert list(counts) == want["counts"]
    assert hashlib.sha256(np.asarray(z["item_cate_list"], np.int32).tobytes()).hexdigest() == want["item_cate_list"]
    for prefix, ps, names in (("train_", train, TRAIN), ("test_", test, TEST)):
        assert len(ps) == want[prefix + "n"]
        for k in names + ["hist_t"]:
            a = np.ascontiguousarray(getattr(ps, k))
            assert a.dtype == (np.float32 if k == "hist_t" else np.int64)
            assert hashlib.sha256(a.tobytes()).hexdigest() == want[prefix + k], (prefix, k)
    for k, v in want["train_head"].items():
        assert getattr(train, k)[:8].tolist() == v
    for k, v in want["test_head"].items():
        assert getattr(test, k)[:8].tolist() == v


def test_tuple_form_round_trips(golden_dir):
    """`dataset.pkl`'s python tuples (build_dataset.py:58-59, :71) from the CSR form and back."""
    z, (U, I, C), train, test = _build(golden_dir, "clothing")
    tr, te = build_dataset(z["reviewerID"], z["asin"], z["unixReviewTime"], z["item_cate_list"], I)
    assert len(tr[0]) == 7 and len(te[0]) == 6 and tr[0][5] in (0, 1) and isinstance(te[0][4], tuple)
    assert all(float(w) in [1.0 / k for k in range(1, 13)] for w in tr[0][3])
    for ps, samples in ((train, tr), (test, te)):
        back = PackedSet.from_samples(samples)
        for k in (TEST if ps.is_test else TRAIN):
            assert np.array_equal(getattr(back, k), getattr(ps, k)), k
        assert np.array_equal(back.hist_t, ps.hist_t)
        assert to_samples(back)[:50] == samples[:50]


def test_time_weights():
    # 1 / #{g in 2,4,..,4096 : days + 1 >= g}  (build_dataset.py:16-21); gaps of 0, 1, 3, 7, 100, 5010 days
    w = proc_time_emb([10, 9, 7, 3, -90, -5000], 10)[1:]    # (a same-day gap never occurs: sessions are whole days)
    assert [round(1 / float(x)) for x in w] == [1, 2, 3, 6, 12]
