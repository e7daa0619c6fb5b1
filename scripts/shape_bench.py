#!/usr/bin/env python3
"""Train-step time at other shapes of BASELINE.json's configs (single GPU, resident batches):
python scripts/shape_bench.py d=256 Ls=10 B=4096 [U=.. I=.. C=..] [sess=amazon] [td=bf16] [mm=bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlsan_amd import synth
from tlsan_amd.model import Model
kw = dict(a.split("=") for a in sys.argv[1:])
d, Ls, B = int(kw.get("d", 256)), int(kw.get("Ls", 10)), int(kw.get("B", 4096))
cfg = synth.make_config("electronics", Ls=Ls, hidden_units=d, itemid_embedding_size=d // 2, userid_embedding_size=d // 2,
                        cateid_embedding_size=d // 2, user_count=int(kw.get("U", 39991)), item_count=int(kw.get("I", 22048)),
                        cate_count=int(kw.get("C", 673)))
m = Model(cfg, synth.item_cate_list(cfg), l2_mode="lazy", table_dtype=kw.get("td", "f32"), matrix_dtype=kw.get("mm", "f32"))
dbs = [m.device_batch(b) for b in synth.make_batches(cfg, 4, B, seed=1234, sessions=kw.get("sess", "geometric"))]
AH = int(kw.get("ahead", 2))      # destination indices built this many steps ahead (0: inside the step)
def step(s):
    m.train_async(dbs[s % 4], 1.0, next_batch=dbs[(s + 1) % 4] if AH >= 1 else None, after_next=dbs[(s + 2) % 4] if AH >= 2 else None)
for s in range(10):
    step(s)
torch.cuda.synchronize()
N = 100
t0 = time.perf_counter()
for s in range(10, 10 + N):
    step(s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print("d=%d Ls=%d B=%d%s: %.1f us/step, %.2f M seq/s, loss %.4f" % (d, Ls, B, "".join(" %s=%s" % (k, kw[k]) for k in ("sess", "td", "mm") if k in kw), dt * 1e6, B / dt / 1e6, float(m._out[0].item())))
