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
# a whole evaluation pass over an Electronics-size test set (39 991 users: one test row per user), the way the driver
# runs it (tlsan_amd.train: launches of EVAL_CHUNK = 4096 rows, one host copy at the end) and the way the reference
# feeds it (test batch 128, a host read per batch)
from tlsan_amd.train import EVAL_CHUNK
N = cfg["user_count"]
for chunk, sync_each in ((EVAL_CHUNK, False), (128, True)):
    tb = m.device_batch(synth.make_batches(cfg, 1, chunk, seed=9, test=True)[0], is_test=True)
    for kind, fn in (("eval_auc", m.pairs_ranked_right), ("P@k / R@k ranks", m.label_ranks)):
        def one_pass():
            parts = []
            for lo in range(0, N, chunk):
                r = fn(tb)
                parts.append(r.cpu() if sync_each else r)
            if not sync_each:
                torch.cat(parts).cpu()
        one_pass()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            one_pass()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        extra = ""
        if "ranks" in kind:
            nl = (N + chunk - 1) // chunk
            extra = "  (%.1f TFLOP/s on the fp32 matrix pipe)" % (2.0 * nl * chunk * cfg["item_count"] * cfg["hidden_units"] / dt / 1e12)
        print("full pass, %5d users, %4d rows per launch%s: %-16s %8.2f ms%s" % (N, chunk, ", host read per launch" if sync_each else "", kind, dt * 1e3, extra), flush=True)
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
