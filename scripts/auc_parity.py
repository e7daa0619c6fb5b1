#!/usr/bin/env python3
"""End-to-end check on a real dataset: train the HIP path and the fp64 oracle from the same
initial weights on the same batch order (reference hyper-parameters: d=64, B=32, lr=1, L2 5e-5,
clip 5) and compare test AUC.  Also prints the README sanity band value.
    python scripts/auc_parity.py clothing 1500"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tlsan_oracle as orc
from tlsan_amd.input import DataInput, DataInputTest, load_packed
from tlsan_amd.model import Model
from tlsan_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "clothing"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
with_oracle = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
README = {"clothing": 0.9363, "digital_music": 0.9753}
train_set, test_set, (U, I, C), icl = load_packed(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "packed_%s.npz" % name))
cfg = synth.make_config("clothing", hidden_units=64, itemid_embedding_size=32, userid_embedding_size=32,
                        cateid_embedding_size=32, user_count=U, item_count=I, cate_count=C)
m = Model(cfg, icl)
p = {k: v.astype(np.float64) for k, v in m.get_params().items()}

def auc_gpu():
    s = 0.0
    for _, tb in DataInputTest(test_set, 128, 10):
        s += m.eval_auc(None, tb) * len(tb[0])
    return s / len(test_set)

def auc_orc(q):
    s = 0.0
    for _, tb in DataInputTest(test_set, 128, 10):
        a, _, _ = orc.eval_auc_batch(q, icl, orc.as_batch(tb, True), 8)
        s += a * len(tb[0])
    return s / len(test_set)

rng = np.random.RandomState(1234)
step, t0, rows = 0, time.time(), []
print("init AUC gpu %.4f" % auc_gpu() + (" oracle %.4f" % auc_orc(p) if with_oracle else ""), flush=True)
while step < nsteps:
    train_set.shuffle(rng)
    for _, batch in DataInput(train_set, 32, 10):
        lg = m.train(None, batch, 1.0)
        if with_oracle:
            lo, p, _ = orc.train_step(p, icl, orc.as_batch(batch), 8, 5e-5, 1.0)
        step += 1
        if step % 500 == 0 or step == nsteps:
            ag = auc_gpu()
            ao = auc_orc(p) if with_oracle else float("nan")
            rows.append((step, ag, ao))
            print("step %5d loss gpu %.5f%s | AUC gpu %.4f oracle %.4f diff %+.4f  (%.0fs)" %
                  (step, lg, (" oracle %.5f" % lo) if with_oracle else "", ag, ao, ag - ao, time.time() - t0), flush=True)
        if step >= nsteps:
            break
print(json.dumps(dict(dataset=name, steps=step, readme_auc=README.get(name), rows=rows)))
