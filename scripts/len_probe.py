#!/usr/bin/env python3
"""How much of the long-window step is length imbalance?  Movies-TV shape, Ls = 90: the synthetic window lengths
(log-normal, mean 14, 1 % at 90) against batches whose windows all have ONE length (timing only: the ids stay).
    python scripts/len_probe.py [d=128] [U=.. I=.. C=..]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlsan_amd import synth
from tlsan_amd.model import Model
kw = dict(a.split("=") for a in sys.argv[1:])
d = int(kw.get("d", 128))
cfg = synth.make_config("electronics", Ls=90, hidden_units=d, itemid_embedding_size=d // 2, userid_embedding_size=d // 2,
                        cateid_embedding_size=d // 2, user_count=int(kw.get("U", 35896)), item_count=int(kw.get("I", 28589)),
                        cate_count=int(kw.get("C", 15)))
host = synth.make_batches(cfg, 4, 4096, seed=1234)
def run(tag, fix=None):
    m = Model(cfg, synth.item_cate_list(cfg), l2_mode="lazy", matrix_dtype=kw.get("mm", "f32"), table_dtype=kw.get("td", "f32"))
    dbs = [m.device_batch(b) for b in host]
    if fix is not None:
        g = torch.Generator(device="cuda").manual_seed(5)
        for db in dbs:     # (ids of the synthetic popularity law in every slot, so that the longer windows are not all item 0)
            src = db.hist_i[db.hist_i > 0]
            db.hist_i.copy_(src[torch.randint(0, src.numel(), tuple(db.hist_i.shape), device="cuda", generator=g)])
            db.hist_t.fill_(0.5)
            db.sl.fill_(fix)
    def step(s):
        m.train_async(dbs[s % 4], 1.0, next_batch=dbs[(s + 1) % 4], after_next=dbs[(s + 2) % 4])
    for s in range(8):
        step(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(8, 68):
        step(s)
    torch.cuda.synchronize()
    print("%-34s %.1f us/step" % (tag, (time.perf_counter() - t0) / 60 * 1e6), flush=True)
import numpy as np
sl = np.asarray(host[0][6])
print("window lengths: mean %.1f, p50 %d, p90 %d, p99 %d, max %d, share at 90: %.3f" % (sl.mean(), np.percentile(sl, 50), np.percentile(sl, 90), np.percentile(sl, 99), sl.max(), (sl == 90).mean()))
run("synthetic lengths")
for L in (14, 30, 45, 90):
    run("every window %d long" % L, L)
