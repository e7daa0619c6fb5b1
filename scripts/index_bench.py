#!/usr/bin/env python3
"""The destination-index build of a batch ALONE (nothing else on the GPU), per kernel under rocprofv3:
python scripts/index_bench.py d=128 Ls=10 B=4096 U=10000000 I=5000000 C=10000   (the shape arguments of shape_bench.py)
Counters are not consumed between the calls (no training step runs): timings only."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlsan_amd import _lib as L, synth
from tlsan_amd.model import Model
kw = dict(a.split("=") for a in sys.argv[1:])
d, Ls, B = int(kw.get("d", 128)), int(kw.get("Ls", 10)), int(kw.get("B", 4096))
cfg = synth.make_config("electronics", Ls=Ls, hidden_units=d, itemid_embedding_size=d // 2, userid_embedding_size=d // 2,
                        cateid_embedding_size=d // 2, user_count=int(kw.get("U", 39991)), item_count=int(kw.get("I", 22048)),
                        cate_count=int(kw.get("C", 673)))
m = Model(cfg, synth.item_cate_list(cfg), l2_mode="lazy")
dbs = [m.device_batch(b) for b in synth.make_batches(cfg, 4, B, seed=1234)]
flag = L.INDEX_FOR_LAZY_SGD
st = torch.cuda.current_stream().cuda_stream
def build(s):
    L.check(m.lib.tlsan_batch_index(C.byref(m.dims), C.byref(dbs[s % 4].c), m.cparams.item_cate, m.state.data_ptr(), (s % 3) | flag, C.c_void_p(st)), "tlsan_batch_index")
for s in range(6):
    build(s)
torch.cuda.synchronize()
N = 60
t0 = time.perf_counter()
for s in range(N):
    build(s)
torch.cuda.synchronize()
print("index build alone: %.1f us per batch (B=%d, %d use slots)" % ((time.perf_counter() - t0) / N * 1e6, B, B * (Ls + dbs[0].Sn + 1)))
