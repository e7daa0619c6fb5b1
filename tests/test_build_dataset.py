"""f4: tlsan_amd.build_dataset against the tuples the reference's real
TLSAN/build_dataset.py built from the same review log (tests/golden/packed_<name>.npz, made by
tests/golden/make_fixtures.py): every train and test tuple, in the shuffled order, bit for bit."""
import os

import numpy as np
import pytest

from tlsan_amd.build_dataset import build_dataset, proc_time_emb
from tlsan_amd.input import PackedSet, load_packed


@pytest.mark.parametrize("name", ["clothing", "digital_music"])
def test_tuples_identical_to_reference(golden_dir, name):
    z = np.load(os.path.join(golden_dir, "reviews_%s.npz" % name))
    U, I, C = (int(x) for x in z["counts"][:3])
    train, test = build_dataset(z["reviewerID"], z["asin"], z["unixReviewTime"], z["item_cate_list"], I)
    assert len(test) == U                                   # build_dataset.py:78
    ref_train, ref_test, counts, icl = load_packed(os.path.join(golden_dir, "packed_%s.npz" % name))
    assert counts == (U, I, C) and np.array_equal(icl, z["item_cate_list"])
    for got, ref in ((PackedSet.from_samples(train), ref_train), (PackedSet.from_samples(test), ref_test)):
        assert len(got) == len(ref)
        names = ["u", "hist_off", "hist", "sess_off", "sess", "cate"] + (["pos", "neg"] if ref.is_test else ["target", "label"])
        for k in names:
            assert np.array_equal(getattr(got, k), getattr(ref, k)), k
        assert np.array_equal(got.hist_t.view(np.uint32), ref.hist_t.view(np.uint32))


def test_time_weights():
    # 1 / #{g in 2,4,..,4096 : days + 1 >= g}  (build_dataset.py:16-21); gaps of 0, 1, 3, 7, 100, 5010 days
    w = proc_time_emb([10, 9, 7, 3, -90, -5000], 10)[1:]    # (a same-day gap never occurs: sessions are whole days)
    assert [round(1 / float(x)) for x in w] == [1, 2, 3, 6, 12]
