#!/usr/bin/env python3
"""Host-side cost of enqueueing one step (is the eager loop host-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlsan_amd import synth
from tlsan_amd.model import Model
cfg = synth.make_config("electronics")
m = Model(cfg, synth.item_cate_list(cfg), l2_mode="lazy")
dbs = [m.device_batch(b) for b in synth.make_batches(cfg, 4, 4096, seed=1234)]
for pf in (0, 1):
    for s in range(20):
        m.train_async(dbs[s % 4], 1.0, next_batch=dbs[(s + 1) % 4] if pf else None)
    torch.cuda.synchronize()
    N = 300
    t0 = time.perf_counter()
    for s in range(N):
        m.train_async(dbs[s % 4], 1.0, next_batch=dbs[(s + 1) % 4] if pf else None)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("prefetch=%d: enqueue %.1f us/step, total %.1f us/step" % (pf, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
    if pf:  # consume the dangling announced batch
        m.train_async(dbs[N % 4], 1.0)
        torch.cuda.synchronize()
# pure python/ctypes overhead of the call with a trivially small batch (GPU work negligible)
sm = [m.device_batch(b) for b in synth.make_batches(cfg, 4, 16, seed=1)]
for s in range(20):
    m.train_async(sm[s % 4], 1.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(300):
    m.train_async(sm[s % 4], 1.0)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("B=16: enqueue %.1f us/step, total %.1f" % ((t1 - t0) / 300 * 1e6, (time.perf_counter() - t0) / 300 * 1e6))
