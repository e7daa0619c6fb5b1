#!/usr/bin/env python3
"""Static-shape sharded step at one rank under a profiler: eager steps only (MODE=graph: graph replays)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29548")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from tlsan_amd import synth
from tlsan_amd.dist import ShardedModel
cfg = synth.make_config("electronics")
m = ShardedModel(cfg, synth.item_cate_list(cfg), l2_mode="lazy", static_rows=True)
NB = int(os.environ.get("NB", 4))
AHEAD = int(os.environ.get("AHEAD", 1))
if os.environ.get("OTHER"):      # a second model in the process (its streams, its buffers)
    m_other = ShardedModel(cfg, synth.item_cate_list(cfg), l2_mode="lazy")
    ob = [m_other.device_batch(b) for b in synth.make_batches(cfg, 2, 4096, seed=1)]
    for s in range(4):
        m_other.train_async(ob[s % 2], 1.0, next_batch=ob[(s + 1) % 2])
dbs = [m.device_batch(b) for b in synth.make_batches(cfg, NB, 4096, seed=1)]
def step(s):
    m.train_async(dbs[s % NB], 1.0, next_batch=dbs[(s + 1) % NB], after_next=dbs[(s + 2) % NB] if AHEAD > 1 else None)
for s in range(2 * NB):
    step(s)
graphs = []
if os.environ.get("MODE") == "graph":
    for i in range(4):
        g = m.capture_step(dbs[i], dbs[(i + 1) % 4], 1.0)
        m.replay(g)
        graphs.append(g)
torch.cuda.synchronize()
N = int(os.environ.get("N", 200))
t0 = time.perf_counter()
for s in range(N):
    if graphs:
        m.replay(graphs[s % 4])
    else:
        step(s)
torch.cuda.synchronize()
print("%s: %.1f us/step" % (os.environ.get("MODE", "eager"), (time.perf_counter() - t0) / N * 1e6))
dist.destroy_process_group()
