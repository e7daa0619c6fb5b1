#!/usr/bin/env python3
"""Kernel-level experiments: per-stage event timings of the train step vs batch size, and the
forward-only kernel.  Usage: python scripts/kbench.py [B ...]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlsan_amd import _lib as L, synth
from tlsan_amd.model import Model

cfg = synth.make_config(os.environ.get("WORKLOAD", "electronics"))
icl = synth.item_cate_list(cfg)
lib = L.load()
Bs = [int(x) for x in sys.argv[1:]] or [256, 1024, 4096, 16384]
m = Model(cfg, icl)
for B in Bs:
    hb = synth.make_batches(cfg, 4, B, seed=7)
    dbs = [m.device_batch(b) for b in hb]
    for lvl in (2, 1):
        for _ in range(10):
            m.train_async(dbs[0], 1.0)
        torch.cuda.synchronize()
        lib.tlsan_profile_enable(lvl)
        t0 = time.perf_counter()
        n = 50
        for s in range(n):
            m.train_async(dbs[s % 4], 1.0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        buf = (ctypes.c_float * (n * 5))()
        k = lib.tlsan_profile_collect(buf, n)
        lib.tlsan_profile_enable(0)
        seg = np.frombuffer(buf, dtype=np.float32)[: k * 5].reshape(k, 5).mean(0) * 1e3
        print("B=%6d lvl%d step %.1f us | idx %.1f fwd_bwd %.1f dk %.1f fin %.1f apply %.1f | Sn=%d" % (B, lvl, dt * 1e6, *seg, dbs[0].Sn), flush=True)
    # forward-only kernel
    tb = synth.make_batches(cfg, 1, B, seed=9, test=True)[0]
    db = m.device_batch(tb, is_test=True)
    for _ in range(5):
        m.forward(db)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        m.forward(db)
    e1.record()
    torch.cuda.synchronize()
    print("B=%6d forward-only %.1f us" % (B, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
