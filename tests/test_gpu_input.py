"""a1 / f1: the device-resident batcher (tlsan_batch_pack) against what the reference's
real TLSAN/input.py emitted (fixtures captured by tests/golden/make_fixtures.py) and against the
host batcher.  Integer / mask work: bit-exact."""
import os

import numpy as np
import pytest

from tlsan_amd.input import DataInput, DataInputTest, PackedSet, load_packed

pytestmark = pytest.mark.gpu
CASES = [("DataInput", 32, 10), ("DataInputTest", 128, 10), ("DataInput", 64, 4),
         ("DataInputTest", 50, 3), ("DataInput", 1024, 10)]
KEYS = ["u", "i", "yj", "hist_i", "hist_i_new", "hist_t", "sl", "new_sl", "c"]


def _same(got, want, what):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, what
    if got.dtype == np.float32:
        assert np.array_equal(got.view(np.uint32), np.asarray(want, np.float32).view(np.uint32)), what
    else:
        assert np.array_equal(got, want), what


@pytest.mark.parametrize("name", ["clothing", "digital_music"])
@pytest.mark.parametrize("cls_name,bs,k", CASES)
def test_device_batches_bit_exact(golden_dir, name, cls_name, bs, k):
    from tlsan_amd.device_input import DeviceDataInput, DeviceDataInputTest
    train, test, counts, icl = load_packed(os.path.join(golden_dir, "packed_%s.npz" % name))
    fx = np.load(os.path.join(golden_dir, "batches_%s.npz" % name))
    data = train if cls_name == "DataInput" else test
    it = (DeviceDataInput if cls_name == "DataInput" else DeviceDataInputTest)(data, bs, k)
    assert it.epoch_size == int(fx["%s_bs%d_k%d_nbatches" % (cls_name, bs, k)])
    seen = 0
    for step, db in it:
        pre = "%s_bs%d_k%d_b%d_" % (cls_name, bs, k, step - 1)
        if pre + "u" not in fx:
            continue
        seen += 1
        for key, got in zip(KEYS, db.to_host()):
            _same(got, fx[pre + key], (pre, key))
    assert seen >= 2


def test_device_batcher_matches_host_batcher_after_shuffle():
    """ragged edge cases (history longer / shorter than k, empty session, short last batch) and a
    shuffled epoch: every batch equals the host batcher's."""
    from tlsan_amd.device_input import DeviceDataInput, DevicePackedSet
    rng = np.random.RandomState(5)
    samples = []
    for s in range(203):
        nh, ns = rng.randint(1, 25), rng.randint(0, 7)
        samples.append((int(rng.randint(50)), [int(x) for x in rng.randint(0, 90, nh)], [int(x) for x in rng.randint(0, 90, ns)],
                        [float(1.0 / x) for x in rng.randint(1, 13, nh)], int(rng.randint(90)), int(rng.randint(2)), int(rng.randint(9))))
    host, dev = PackedSet.from_samples(samples), DevicePackedSet(PackedSet.from_samples(samples))
    for epoch in range(2):
        host.shuffle(np.random.RandomState(epoch))
        dev.shuffle(np.random.RandomState(epoch))
        for (s1, hb), (s2, db) in zip(DataInput(host, 32, 10), DeviceDataInput(dev, 32, 10)):
            assert s1 == s2
            for key, a, b in zip(KEYS, hb, db.to_host()):
                _same(b, a, (epoch, s1, key))


def test_training_on_device_batches_equals_host_batches(golden_dir):
    """Model.train accepts the DeviceBatch directly; same parameters as with host tuples."""
    from tlsan_amd.device_input import DeviceDataInput
    from tlsan_amd.model import Model
    from tests.helpers import make_config
    train, test, (U, I, Cc), icl = load_packed(os.path.join(golden_dir, "packed_clothing.npz"))
    cfg = make_config(U=U, I=I, C=Cc, d=64)
    runs = []
    for cls in (DataInput, DeviceDataInput):
        m = Model(cfg, icl, l2_mode="lazy")
        for step, batch in cls(train, 64, 10):
            m.train_async(batch, 1.0)
            if step == 6:
                break
        runs.append(m.get_params())
    for k in runs[0]:
        assert np.array_equal(runs[0][k], runs[1][k]), k
