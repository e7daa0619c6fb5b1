import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tlsan_amd import synth
from tlsan_amd.model import Model
cfg = synth.make_config("electronics")
m = Model(cfg, synth.item_cate_list(cfg), l2_mode="lazy")
dbs = [m.device_batch(b) for b in synth.make_batches(cfg, 16, 4096, seed=1234)]
def step(s):
    m.train_async(dbs[s % 16], 1.0, next_batch=dbs[(s + 1) % 16], after_next=dbs[(s + 2) % 16])
for rep in range(4):
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    for s in range(W): step(s)
    torch.cuda.synchronize()
    m.started_at = []
    t0 = time.perf_counter()
    for s in range(W, W + 20): step(s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    st = np.asarray(m.started_at); m.started_at = None
    d = np.diff(st) * 1e6
    print("window %.1f us/step | first start after t0 %.1f us | last start -> sync end %.1f us | host loop done at %.1f us | intervals: %s" % (
        (t2 - t0) / 20 * 1e6, (st[0] - t0) * 1e6, (t2 - st[-1]) * 1e6, (t1 - t0) * 1e6, " ".join("%.0f" % x for x in d)))
    time.sleep(0.5)
