#!/usr/bin/env python3
"""Sharded step at world 1: where does the HOST time of train_async go (cProfile, prefetching loop)?"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29546")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from tlsan_amd import synth
from tlsan_amd.dist import ShardedModel
cfg = synth.make_config("electronics")
m = ShardedModel(cfg, synth.item_cate_list(cfg), l2_mode="lazy", static_rows=True)
dbs = [m.device_batch(b) for b in synth.make_batches(cfg, 4, 4096, seed=1)]
for s in range(12):
    m.train_async(dbs[s % 4], 1.0, next_batch=dbs[(s + 1) % 4], after_next=dbs[(s + 2) % 4])
torch.cuda.synchronize()
N = 300
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for s in range(N):
    m.train_async(dbs[s % 4], 1.0, next_batch=dbs[(s + 1) % 4], after_next=dbs[(s + 2) % 4])
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("enqueue %.1f us/step (under cProfile), total %.1f" % ((t1 - t0) / N * 1e6, (time.perf_counter() - t0) / N * 1e6))
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
dist.destroy_process_group()
