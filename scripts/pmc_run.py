#!/usr/bin/env python3
"""Small driver for rocprofv3 --pmc passes: a handful of train steps at the bench shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlsan_amd import synth
from tlsan_amd.model import Model
# python scripts/pmc_run.py [B] [d=128 Ls=90 U=.. I=.. C=.. td=.. mm=..]   (the shape arguments of scripts/shape_bench.py)
pos = [a for a in sys.argv[1:] if "=" not in a]
kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
B = int(pos[0]) if pos else int(kw.get("B", 4096))
if kw:
    d = int(kw.get("d", 128))
    cfg = synth.make_config("electronics", Ls=int(kw.get("Ls", 10)), hidden_units=d, itemid_embedding_size=d // 2, userid_embedding_size=d // 2,
                            cateid_embedding_size=d // 2, user_count=int(kw.get("U", 39991)), item_count=int(kw.get("I", 22048)),
                            cate_count=int(kw.get("C", 673)))
else:
    cfg = synth.make_config("electronics")
m = Model(cfg, synth.item_cate_list(cfg), l2_mode="lazy", table_dtype=kw.get("td", "f32"), matrix_dtype=kw.get("mm", "f32"))   # the bench default
dbs = [m.device_batch(b) for b in synth.make_batches(cfg, 4, B, seed=1234)]
for s in range(12):
    m.train_async(dbs[s % 4], 1.0)
torch.cuda.synchronize()
