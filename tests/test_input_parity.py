"""a1: our batcher vs. what the reference's real TLSAN/input.py emitted
(fixtures captured by tests/golden/make_fixtures.py).  Integer/mask work: bit-exact."""
import os

import numpy as np
import pytest

from tlsan_amd.input import DataInput, DataInputTest, PackedSet, load_packed

NAMES = ["clothing", "digital_music"]
CASES = [("DataInput", 32, 10), ("DataInputTest", 128, 10), ("DataInput", 64, 4),
         ("DataInputTest", 50, 3), ("DataInput", 1024, 10)]


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("cls_name,bs,k", CASES)
def test_batches_bit_exact(golden_dir, name, cls_name, bs, k):
    train, test, counts, icl = load_packed(os.path.join(golden_dir, "packed_%s.npz" % name))
    fx = np.load(os.path.join(golden_dir, "batches_%s.npz" % name))
    data = train if cls_name == "DataInput" else test
    cls = DataInput if cls_name == "DataInput" else DataInputTest
    it = cls(data, bs, k)
    assert it.epoch_size == int(fx["%s_bs%d_k%d_nbatches" % (cls_name, bs, k)])
    seen = 0
    for step, batch in it:
        pre = "%s_bs%d_k%d_b%d_" % (cls_name, bs, k, step - 1)
        if pre + "u" not in fx:
            continue
        seen += 1
        keys = ["u", "i", "yj", "hist_i", "hist_i_new", "hist_t", "sl", "new_sl", "c"]
        for key, got in zip(keys, batch):
            want = fx[pre + key]
            got = np.asarray(got)
            assert got.shape == want.shape, (pre, key)
            if key == "hist_t":
                assert got.dtype == np.float32
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (pre, key)
            else:
                assert np.array_equal(got, want), (pre, key)
            if key in ("hist_i", "hist_i_new"):
                assert got.dtype == np.int64
    assert seen >= 2


def test_from_samples_roundtrip():
    # ragged + edge cases: history longer/shorter than k, empty session
    train = [
        (3, [1, 2, 3, 4, 5, 6], [7], [0.5, 0.5, 1.0, 1.0, 1.0, 1.0], 9, 1, 2),
        (4, [8], [], [1.0], 2, 0, 1),
        (5, [1, 2, 3], [4, 5, 6, 7], [1 / 3, 0.5, 1.0], 3, 1, 0),
    ]
    ps = PackedSet.from_samples(train)
    _, b = next(iter(DataInput(ps, 8, 4)))
    u, i, y, hist_i, hist_i_new, hist_t, sl, new_sl, c = b
    assert hist_i.tolist() == [[3, 4, 5, 6], [8, 0, 0, 0], [1, 2, 3, 0]]
    assert sl.tolist() == [4, 1, 3] and new_sl.tolist() == [1, 0, 4]
    assert hist_i_new.tolist() == [[7, 0, 0, 0], [0, 0, 0, 0], [4, 5, 6, 7]]
    assert np.allclose(hist_t[2], [np.float32(1 / 3), 0.5, 1.0, 0.0])
    assert y.tolist() == [1, 0, 1] and c.tolist() == [2, 1, 0]
    test = [(1, [1, 2], [3], [1.0, 1.0], (4, 5), 0)]
    _, tb = next(iter(DataInputTest(test, 2, 10)))
    assert tb[1].tolist() == [4] and tb[2].tolist() == [5]
    with pytest.raises(ValueError):
        DataInput(test, 2, 10)
    with pytest.raises(ValueError):
        PackedSet.from_samples([])
