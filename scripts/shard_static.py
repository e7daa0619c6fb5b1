#!/usr/bin/env python3
"""Static-shape sharded step at one rank: parity with the dynamic step, eager and graph-replay step times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from tlsan_amd import synth
from tlsan_amd.dist import ShardedModel
cfg = synth.make_config("electronics")
icl = synth.item_cate_list(cfg)
batches = synth.make_batches(cfg, 4, 4096, seed=1)
def run(static, steps=12):
    m = ShardedModel(cfg, icl, l2_mode="lazy", static_rows=static)
    dbs = [m.device_batch(b) for b in batches]
    losses = []
    for s in range(steps):
        m.train_async(dbs[s % 4], 1.0, next_batch=dbs[(s + 1) % 4])
        losses.append(float(m.last_loss.item()))
    return m, dbs, losses
m0, dbs0, l0 = run(False)
m1, dbs, l1 = run(True)
print("losses dynamic:", ["%.6f" % x for x in l0[-4:]])
print("losses static :", ["%.6f" % x for x in l1[-4:]])
print("max |loss diff| %.3g; shard max diff %.3g; cap %d" % (max(abs(a - b) for a, b in zip(l0, l1)), (m0.shard * m0._P - m1.shard * m1._P).abs().max().item(), m1._st["cap"]))
m1.check_static_overflow()
def timeit(f, n=200, reps=4):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(n): f(s)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best
print("dynamic eager: %.1f us/step" % timeit(lambda s: m0.train_async(dbs0[s % 4], 1.0, next_batch=dbs0[(s + 1) % 4])))
print("static eager : %.1f us/step" % timeit(lambda s: m1.train_async(dbs[s % 4], 1.0, next_batch=dbs[(s + 1) % 4])))
print("static eager, two ahead: %.1f us/step" % timeit(lambda s: m1.train_async(dbs[s % 4], 1.0, next_batch=dbs[(s + 1) % 4], after_next=dbs[(s + 2) % 4])))
l2a = float(m1.last_loss.item())
# host side alone: enqueue time per step (no device wait inside the loop)
import time as _t
torch.cuda.synchronize(); t0 = _t.perf_counter()
for s in range(100): m1.train_async(dbs[s % 4], 1.0, next_batch=dbs[(s + 1) % 4], after_next=dbs[(s + 2) % 4])
t1 = _t.perf_counter(); torch.cuda.synchronize(); t2 = _t.perf_counter()
print("static eager, two ahead: host enqueue %.1f us/step, drained %.1f us/step" % ((t1 - t0) / 100 * 1e6, (t2 - t0) / 100 * 1e6))
nb16 = [m1.device_batch(b) for b in synth.make_batches(cfg, 16, 4096, seed=1234)]
for s in range(16): m1.train_async(nb16[s % 16], 1.0, next_batch=nb16[(s + 1) % 16], after_next=nb16[(s + 2) % 16])
print("static eager, two ahead, 16 batches: %.1f us/step" % timeit(lambda s: m1.train_async(nb16[s % 16], 1.0, next_batch=nb16[(s + 1) % 16], after_next=nb16[(s + 2) % 16]), n=208))
m1.check_static_overflow()
graphs = []
for i in range(4):
    g = m1.capture_step(dbs[i], dbs[(i + 1) % 4], 1.0)
    m1.replay(g)
    graphs.append(g)
print("static graph : %.1f us/step" % timeit(lambda s: m1.replay(graphs[s % 4])))
print("loss after replays %.6f" % float(m1.last_loss.item()))
m1.check_static_overflow()
dist.destroy_process_group()
