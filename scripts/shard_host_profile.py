#!/usr/bin/env python3
"""Static-shape sharded step at world 1 under cProfile: where the host's time per step goes (python scripts/shard_host_profile.py)."""
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
def step(s):
    m.train_async(dbs[s % 4], 1.0, next_batch=dbs[(s + 1) % 4], after_next=dbs[(s + 2) % 4])
for s in range(40):
    step(s)
torch.cuda.synchronize()
N = 400
t0 = time.perf_counter()
for s in range(N):
    step(s)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("plain: host %.1f us/step, total %.1f us/step" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6), flush=True)
pr = cProfile.Profile()
pr.enable()
for s in range(N):
    step(s)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
dist.destroy_process_group()
