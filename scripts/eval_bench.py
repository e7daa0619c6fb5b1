#!/usr/bin/env python3
"""Evaluation-side throughput at the Electronics-scale shape (SURVEY 8d asks for it beside the
train step): eval_auc (two logits per test user, model.py:237-263) and the all-items ranking
(model.py:140-156: u_t . all_emb^T over I items) on the GPU, and the CPU port of eval_auc."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tlsan_amd import synth
from tlsan_amd.model import Model
cfg = synth.make_config("electronics")
icl = synth.item_cate_list(cfg)
m = Model(cfg, icl, l2_mode="lazy")
for B in (128, 4096):
    tb = [m.device_batch(b, is_test=True) for b in synth.make_batches(cfg, 4, B, seed=9, test=True)]
    for name, fn in (("eval_auc forward (2 logits/user)", lambda db: m.forward(db)), ("all-items ranks (I=%d)" % cfg["item_count"], lambda db: m.label_ranks(db))):
        for s in range(5):
            fn(tb[s % 4])
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for s in range(n):
            fn(tb[s % 4])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        extra = ""
        if "ranks" in name:
            fl = 2.0 * B * cfg["item_count"] * cfg["hidden_units"]
            extra = "  (%.1f TFLOP/s on the fp32 matrix pipe incl. the forward)" % (fl / dt / 1e12)
        print("GPU B=%5d %-36s %8.1f us/batch  %10.0f users/s%s" % (B, name, dt * 1e6, B / dt, extra), flush=True)
if "--cpu" in sys.argv:
    from oracle import tlsan_oracle as orc, tlsan_torch_ref as tref
    p = tref.params_to_torch(orc.init_params(cfg, seed=1234, dtype=np.float32), dtype=torch.float32)
    for B in (128,):
        hb = synth.make_batches(cfg, 1, B, seed=9, test=True)[0]
        b = tref.batch_to_torch(orc.as_batch(hb, test=True) if "test" in orc.as_batch.__code__.co_varnames else orc.as_batch(hb), dtype=torch.float32)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 5:
            tref.forward(p, icl, b, cfg["num_heads"]); n += 1
        print("CPU port forward B=%d: %.0f users/s (%d threads)" % (B, n * B / (time.perf_counter() - t0), torch.get_num_threads()))
