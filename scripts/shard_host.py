#!/usr/bin/env python3
"""Static-shape sharded step at world 1: host enqueue time vs wall time per step (is the Python side the bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from tlsan_amd import synth
from tlsan_amd.dist import ShardedModel
cfg = synth.make_config("electronics")
m = ShardedModel(cfg, synth.item_cate_list(cfg), l2_mode="lazy", static_rows=True)
dbs = [m.device_batch(b) for b in synth.make_batches(cfg, 4, 4096, seed=1)]
for ahead in (2, 1, 0):
    def step(s):
        m.train_async(dbs[s % 4], 1.0, next_batch=dbs[(s + 1) % 4] if ahead >= 1 else None, after_next=dbs[(s + 2) % 4] if ahead >= 2 else None)
    for s in range(12):
        step(s)
    torch.cuda.synchronize()
    N = 200
    t0 = time.perf_counter()
    for s in range(N):
        step(s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("ahead=%d: host enqueue %.1f us/step, total %.1f us/step" % (ahead, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6), flush=True)
dist.destroy_process_group()
