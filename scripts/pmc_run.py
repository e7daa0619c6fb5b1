#!/usr/bin/env python3
"""Small driver for rocprofv3 --pmc passes: a handful of train steps at the bench shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlsan_amd import synth
from tlsan_amd.model import Model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = synth.make_config("electronics")
m = Model(cfg, synth.item_cate_list(cfg), l2_mode="lazy")   # the bench default
dbs = [m.device_batch(b) for b in synth.make_batches(cfg, 4, B, seed=1234)]
for s in range(12):
    m.train_async(dbs[s % 4], 1.0)
torch.cuda.synchronize()
