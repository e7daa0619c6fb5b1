#!/usr/bin/env python3
"""Is the sharded step bitwise reproducible?  Two fresh ShardedModels (world 1, gloo) on the same batches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
dist.init_process_group("gloo", rank=0, world_size=1)
from tlsan_amd import synth
from tlsan_amd.dist import ShardedModel
from tlsan_amd.model import Model
for name, Ls, B, l2 in (("movies_tv", 90, 1024, "lazy"), ("movies_tv", 10, 1024, "lazy"), ("movies_tv", 90, 1024, "dense"),
                        ("electronics", 10, 1024, "lazy"), ("electronics", 90, 256, "lazy")):
    cfg = synth.make_config(name, Ls=Ls)
    icl = synth.item_cate_list(cfg)
    bs = synth.make_batches(cfg, 3, B, seed=300)
    outs = []
    for rep in range(2):
        m = ShardedModel(cfg, icl, device="cuda:0", l2_mode=l2)
        ls = []
        for b in bs:
            m.train_async(b, 1.0)
            ls.append(float(m.last_loss.item()))
        outs.append((ls, m.gather_params()))
    same = outs[0][0] == outs[1][0]
    diffs = {k: float(np.abs(outs[0][1][k].astype(np.float64) - outs[1][1][k]).max()) for k in outs[0][1]}
    print(name, Ls, B, l2, "losses equal:", same, outs[0][0], outs[1][0], {k: v for k, v in diffs.items() if v > 0}, flush=True)
    # single-GPU Model, same thing
    outs = []
    for rep in range(2):
        m = Model(cfg, icl, l2_mode=l2)
        ls = [m.train(None, b, 1.0) for b in bs]
        outs.append((ls, m.get_params()))
    diffs = {k: float(np.abs(outs[0][1][k].astype(np.float64) - outs[1][1][k]).max()) for k in outs[0][1]}
    print("   Model:", outs[0][0] == outs[1][0], {k: v for k, v in diffs.items() if v > 0}, flush=True)
dist.destroy_process_group()
