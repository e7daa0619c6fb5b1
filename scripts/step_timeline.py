#!/usr/bin/env python3
"""Where do the extra microseconds of a 20-step timed region go?  GPU-side time between consecutive steps (events
recorded after every step) and host-side enqueue times, for a short run after a fence."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlsan_amd import synth
from tlsan_amd.model import Model
cfg = synth.make_config("electronics")
m = Model(cfg, synth.item_cate_list(cfg), l2_mode="lazy")
dbs = [m.device_batch(b) for b in synth.make_batches(cfg, 16, 4096, seed=1234)]
K, W = int(os.environ.get("K", 20)), int(os.environ.get("W", 5))
def step(s):
    m.train_async(dbs[s % 16], 1.0, next_batch=dbs[(s + 1) % 16], after_next=dbs[(s + 2) % 16])
n = 0
for rep in range(3):
    for s in range(W):
        step(n); n += 1
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    host = []
    t0 = time.perf_counter()
    ev[0].record()
    for s in range(K):
        step(n); n += 1
        ev[s + 1].record()
        host.append((time.perf_counter() - t0) * 1e6)
    torch.cuda.synchronize()
    t1 = (time.perf_counter() - t0) * 1e6
    gaps = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(K)]
    print("rep %d: wall %.0f us = %.1f us/step; GPU per step: %s" % (rep, t1, t1 / K, " ".join("%.0f" % g for g in gaps)))
    print("        host returned from step i at (us): %s" % " ".join("%.0f" % h for h in host))
