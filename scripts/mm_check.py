#!/usr/bin/env python3
"""bf16 matrix products (Model(matrix_dtype="bf16")): error against the fp64 oracle and step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tlsan_oracle as orc
from tests.helpers import make_config, random_batch, random_params
from tlsan_amd import synth
from tlsan_amd.model import Model
tup = lambda b: (b["u"], b["i"], b["y"], b["hist_i"], b["hist_i_new"], b["hist_t"], b["sl"], b["sl_new"], b["u_cate"])
for d in (64, 128, 256):
    cfg = make_config(U=300, I=400, C=17, d=d, regulation_rate=1e-3)
    p = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in random_params(cfg, seed=7).items()}
    b, cat = random_batch(cfg, B=96, Sn=4, seed=8)
    ref = orc.forward(p, cat, b, 8)
    g = orc.backward(p, cat, b, 8, cfg["regulation_rate"])
    for tdt in ("f32", "bf16"):
        for mdt in ("f32", "bf16"):
            m = Model(cfg, cat, table_dtype=tdt, matrix_dtype=mdt)
            m.set_params({k: np.asarray(v, np.float32) for k, v in p.items()})
            q = {k: np.asarray(v, np.float64) for k, v in m.get_params().items()}    # the stored (rounded) tables
            r = orc.forward(q, cat, b, 8)
            out = m.grads(tup(b))
            el = np.abs(out["logits"] - r["logits"]).max()
            gq = orc.backward(q, cat, b, 8, cfg["regulation_rate"])
            gq_loss = gq[0]
            gq = gq[2]
            errs, l2 = {}, {}
            for k, v in out["grads"].items():
                if k in gq and not k.endswith("_b2"):      # (b2 of both blocks: mathematically zero gradient)
                    ref_g = np.asarray(gq[k], np.float64).reshape(np.shape(v))
                    errs[k] = float(np.abs(v - ref_g).max() / (np.abs(ref_g).max() + 1e-12))
                    l2[k] = float(np.linalg.norm(v - ref_g) / (np.linalg.norm(ref_g) + 1e-30))
            print("d=%d tables=%s matrix=%s: max|dlogit| %.2e (logit scale %.2f)  worst rel grad err %.2e (%s)  worst L2-rel %.2e (%s) loss err %.2e" %
                  (d, tdt, mdt, el, np.abs(r["logits"]).max(), max(errs.values()) if errs else -1, max(errs, key=errs.get) if errs else "",
                   max(l2.values()), max(l2, key=l2.get), abs(out["loss"] - gq_loss)), flush=True)
cfg = synth.make_config("electronics")
icl = synth.item_cate_list(cfg)
hb = synth.make_batches(cfg, 8, 4096, seed=1234)
for tdt, mdt in (("f32", "f32"), ("bf16", "f32"), ("f32", "bf16"), ("bf16", "bf16")):
    m = Model(cfg, icl, l2_mode="lazy", table_dtype=tdt, matrix_dtype=mdt)
    dbs = [m.device_batch(b) for b in hb]
    for s in range(20):
        m.train_async(dbs[s % 8], 1.0, next_batch=dbs[(s + 1) % 8])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(20, 220):
        m.train_async(dbs[s % 8], 1.0, next_batch=dbs[(s + 1) % 8])
    torch.cuda.synchronize()
    print("electronics B=4096 tables=%s matrix=%s: %.1f us/step, loss %.5f" % (tdt, mdt, (time.perf_counter() - t0) / 200 * 1e6, float(m._out[0].item())), flush=True)
