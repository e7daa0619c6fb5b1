"""Driver fidelity that needs no GPU (reference TLSAN/train.py): the epoch shuffle stream (:15,191) and the
model_dir handling of --from_scratch (:124-127) / reload of the latest checkpoint (:71-76)."""
import os

import numpy as np
import pytest

from tlsan_amd.input import DataInput, load_packed
from tlsan_amd.train import epoch_rng, prepare_model_dir

KEYS = ["u", "i", "yj", "hist_i", "hist_i_new", "hist_t", "sl", "new_sl", "c"]


@pytest.mark.parametrize("name", ["clothing", "digital_music"])
def test_epoch_order_is_the_reference_s(golden_dir, name):
    """Fixture: `random.seed(1234)` + `random.shuffle(train_set)` per epoch + the real input.py's DataInput
    (tests/golden/make_fixtures.py --epoch-order).  Ours: PackedSet.shuffle(epoch_rng()) + our DataInput."""
    z = np.load(os.path.join(golden_dir, "epoch_order_%s.npz" % name))
    train = load_packed(os.path.join(golden_dir, "packed_%s.npz" % name))[0]
    rng = epoch_rng()
    for epoch in (1, 2):
        train.shuffle(rng)
        nb = int(z["e%d_DataInput_bs32_k10_nbatches" % epoch])
        seen = 0
        for step, batch in DataInput(train, 32, 10):
            bi = step - 1
            if bi not in (0, 1, nb - 1):
                continue
            seen += 1
            for k, got in zip(KEYS, batch):
                want = z["e%d_DataInput_bs32_k10_b%d_%s" % (epoch, bi, k)]
                assert np.array_equal(np.asarray(got), want), (epoch, bi, k)
                if k == "hist_t":
                    assert np.asarray(got).dtype == np.float32
        assert seen == 3 and step == nb


def test_model_dir_from_scratch_and_resume(tmp_path):
    d = str(tmp_path / "save_path")
    assert prepare_model_dir(d, False) is None and os.path.isdir(d)          # nothing to reload yet
    for step in (1000, 12000, 3000):
        open(os.path.join(d, "TLSAN-%d.npz" % step), "w").close()
        open(os.path.join(d, "TLSAN-%d.json" % step), "w").close()
    os.makedirs(os.path.join(d, "eval"))
    open(os.path.join(d, "eval", "scalars.csv"), "w").write("0,AUC,0.5\n")
    assert prepare_model_dir(d, False) == os.path.join(d, "TLSAN-12000.npz")  # the latest save (train.py:71-76)
    # a sharded checkpoint is named by its prefix (ShardedModel.restore)
    open(os.path.join(d, "TLSAN-20000.replicated.npz"), "w").close()
    open(os.path.join(d, "TLSAN-20000.shard0of2.npz"), "w").close()
    assert prepare_model_dir(d, False) == os.path.join(d, "TLSAN-20000")
    assert prepare_model_dir(d, True) is None                                  # train.py:124-127: wiped, recreated
    assert os.path.isdir(d) and os.listdir(d) == []
