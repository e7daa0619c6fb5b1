#!/usr/bin/env python3
"""Where does the sharded step spend its time?  (world=1, torch profiler summary)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from tlsan_amd import synth
from tlsan_amd.dist import ShardedModel
cfg = synth.make_config("electronics")
m = ShardedModel(cfg, synth.item_cate_list(cfg), l2_mode=os.environ.get("SHARD_L2", "dense"))
dbs = [m.device_batch(b) for b in synth.make_batches(cfg, 4, 4096, seed=1)]
for s in range(12):
    m.train_async(dbs[s % 4], 1.0, next_batch=dbs[(s + 1) % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(52):
    m.train_async(dbs[s % 4], 1.0, next_batch=dbs[(s + 1) % 4])
torch.cuda.synchronize()
print("step %.1f us" % ((time.perf_counter() - t0) / 52 * 1e6))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for s in range(12):
        m.train_async(dbs[s % 4], 1.0, next_batch=dbs[(s + 1) % 4])
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=22, max_name_column_width=60))
dist.destroy_process_group()
