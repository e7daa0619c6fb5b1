#!/usr/bin/env python3
"""Probe: does index-building work on a second stream overlap with the train step's kernels?
Main stream: eager train steps.  Side stream: one tlsan_route_plan-sized job (memset + mark +
scan + finish, about the cost of k_count + k_index_scan) per step, no dependency on the step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from tlsan_amd import _lib as L, synth
from tlsan_amd.model import Model
cfg = synth.make_config("electronics")
m = Model(cfg, synth.item_cate_list(cfg), l2_mode="lazy")
lib = L.load()
dbs = [m.device_batch(b) for b in synth.make_batches(cfg, 4, 4096, seed=1234)]
dev = "cuda:0"
R, G = 62039, 1
nk = 90000
keys = torch.randint(0, R, (nk,), dtype=torch.int32, device=dev)
cbk = torch.zeros(R, dtype=torch.int32, device=dev)
bufs = [torch.zeros(R, dtype=torch.int32, device=dev) for _ in range(3)]
n_uniq = torch.zeros(1, dtype=torch.int32, device=dev)
sendbuf = torch.zeros(1 + R, dtype=torch.int32, device=dev)
cate_c = torch.zeros(nk, dtype=torch.int32, device=dev)
comp = torch.zeros(nk, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
def side_job():
    L.check(lib.tlsan_route_plan(keys.data_ptr(), nk, R, G, cbk.data_ptr(), bufs[0].data_ptr(), bufs[1].data_ptr(),
                                 bufs[2].data_ptr(), n_uniq.data_ptr(), sendbuf.data_ptr(), R, cate_c.data_ptr(),
                                 comp.data_ptr(), C.c_void_p(side.cuda_stream)), "route_plan")
def run(n, with_side, sync_events):
    ev_prev = None
    for s in range(n):
        if with_side:
            if sync_events and ev_prev is not None:
                side.wait_event(ev_prev)          # side job may start once the previous step is done
            side_job()
            if sync_events:
                e = torch.cuda.Event(); e.record(side)
                torch.cuda.current_stream().wait_event(e)   # consumed by the NEXT step in the real scheme: emulate worst case
        m.train_async(dbs[s % 4], 1.0)
        if sync_events:
            ev_prev = torch.cuda.Event(); ev_prev.record(torch.cuda.current_stream())
for name, ws, se in (("main only", False, False), ("main + free-running side", True, False), ("main + side with event waits", True, True)):
    run(20, ws, se); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(200, ws, se); torch.cuda.synchronize()
    print("%-32s %.1f us/step" % (name, (time.perf_counter() - t0) / 200 * 1e6))
# the side job alone
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): side_job()
torch.cuda.synchronize(); print("side job alone %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
