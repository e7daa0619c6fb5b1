#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of k_fwd_bwd from in-kernel s_memtime stamps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the production kernel carries no stamp code: build the variant first (scripts/mkvariants.sh stamps:"-DTLSAN_STAMPS=1")
os.environ.setdefault("TLSAN_LIB_PATH", os.path.join(ROOT, "ab_run", "stamps.so"))
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlsan_amd import _lib as L, synth
from tlsan_amd.model import Model
# python scripts/stamps.py [B] [d=128 Ls=10 U=.. I=.. C=..]   (the shape arguments of scripts/shape_bench.py)
pos = [a for a in sys.argv[1:] if "=" not in a]
kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
B = int(pos[0]) if pos else int(kw.get("B", 4096))
if kw:
    d = int(kw.get("d", 128))
    cfg = synth.make_config("electronics", Ls=int(kw.get("Ls", 10)), hidden_units=d, itemid_embedding_size=d // 2, userid_embedding_size=d // 2,
                            cateid_embedding_size=d // 2, user_count=int(kw.get("U", 39991)), item_count=int(kw.get("I", 22048)),
                            cate_count=int(kw.get("C", 673)))
else:
    cfg = synth.make_config("electronics")
m = Model(cfg, synth.item_cate_list(cfg), l2_mode="lazy", matrix_dtype=os.environ.get("MM", "f32"), table_dtype=os.environ.get("TD", "f32"))
lib = L.load()
db_host = synth.make_batches(cfg, 1, B, seed=7)[0]
db = m.device_batch(db_host)
for _ in range(5):
    m.train_async(db, 1.0)
# (TLSAN_NW4=2 with d = 128 and the window in registers: 4-wavefront workgroups of 8 samples)
NW4 = os.environ.get("TLSAN_NW4", "1") == "2" and cfg["hidden_units"] == 128 and cfg["Ls"] <= 10
NSB, NWV = (8, 4) if NW4 else (16, 4 if cfg["hidden_units"] == 64 else 8)   # samples / wavefronts per workgroup
nblk = (B + NSB - 1) // NSB
st = torch.zeros((1 << 20) + 8 * 8192, dtype=torch.int64, device="cuda")   # k_apply stamps live from entry 2^20 on
lib.tlsan_debug_stamps(st.data_ptr())
m.train_async(db, 1.0)
torch.cuda.synchronize()
lib.tlsan_debug_stamps(None)
raw = st.cpu().numpy()[:nblk * NWV * 32].reshape(nblk, NWV, 32).astype(np.float64)
s = raw[:, :, :12]
d = np.diff(s, axis=2)
names = ["P1 gather+fwa1", "bar1", "P2 bridge", "bar2", "P3 fwd+logit", "P3 bwd", "bar3", "P4+reduce", "bar4", "P5 bwd long", "bar5+reduce"]
tot = s[:, :, 11] - s[:, :, 0]
print("ticks/wave (s_memtime, 100 MHz on gfx950?): mean %.0f max %.0f" % (tot.mean(), tot.max()))
for i, n in enumerate(names):
    print("%-16s mean %8.0f  p50 %8.0f  max %8.0f  (%.1f%%)" % (n, d[:, :, i].mean(), np.median(d[:, :, i]), d[:, :, i].max(), 100 * d[:, :, i].mean() / tot.mean()))
f = raw[:, :, 12:16]
fd = np.diff(f, axis=2)
ok = (f[:, :, 0] > 0)
for i, n in enumerate(["P5 pos1: maps+exp", "P5 pos1: bwd_compute", "P5 pos1: bwd_dw"]):
    sel = ok & (f[:, :, i + 1] > 0)      # (the pipelined long backward has no such sub-stamps: they stay 0)
    if sel.any():
        print("%-24s mean %8.0f p50 %8.0f max %8.0f" % (n, fd[:, :, i][sel].mean(), np.median(fd[:, :, i][sel]), fd[:, :, i][sel].max()))
ok = raw[:, :, 16] > 0                   # waves that ran position 1 of the long backward
full = (raw[:, :, 9] > 0) & (raw[:, :, 18] > 0)     # waves whose window is full (position 9 ran)
def seg(nm, lo, hi, sel=None):
    # (a stamp a wavefront never reached stays 0: such waves are left out of the segment, not read as a time)
    use = (ok if sel is None else sel) & (raw[:, :, hi] > 0) & (raw[:, :, lo] > 0)
    if not use.any():
        print("%-44s (no wavefront reached both stamps)" % nm)
        return
    dd = (raw[:, :, hi] - raw[:, :, lo])[use]
    print("%-44s mean %8.0f p50 %8.0f max %8.0f" % (nm, dd.mean(), np.median(dd), dd.max()))
if "Ls" in kw and int(kw["Ls"]) > 10:   # streamed windows (the list form)
    anyw = raw[:, :, 0] > 0
    seg("list: ids -> LDS, cursor draws, barrier", 0, 21, anyw)
    seg("list: forward loop", 21, 22, anyw)
    seg("list: barrier, merge of the runs, statistics", 22, 23, anyw)
    seg("list backward: entry -> loop", 9, 16, anyw)
    seg("list backward: loop", 16, 18, anyw)
    seg("list backward: drain, padding, staging", 18, 10, anyw)
else:                                   # (windows in registers: these stamps mean something else in the list form)
    seg("P1: loads issued -> rows in registers", 0, 21)
    seg("P1: weight fragments from the LDS", 21, 22)
    seg("P1: long forward (maps, softmax)", 22, 23)
seg("P1: store long, issue P3's row loads + bridge B", 23, 1)
seg("P3: frags + first session loads", 4, 24)
seg("P3: forward positions + normalise", 24, 25)
seg("P3: logit, BCE, candidate / user stores", 25, 26)
seg("P3: backward positions", 26, 27)
seg("P3: stage accumulators + dlong B loads", 27, 6)
seg("P4: reduction of the short block's staged accumulators", 7, 30, raw[:, :, 7] > 0)
seg("P4: dlong GEMM", 30, 31, raw[:, :, 7] > 0)
seg("P4: dlong GEMM -> barrier 4 (reset of the dK ticket)", 31, 8, raw[:, :, 7] > 0)
if not ("Ls" in kw and int(kw["Ls"]) > 10):
    seg("P5: entry -> start of position 1 (frags, pos 0)", 9, 16)
    seg("P5 pos1: after dW -> start of position 2 (stores)", 15, 17)
    seg("P5: position 1 in all", 16, 17)
    seg("P5: positions 2..8 (7 positions)", 17, 18, full)
    seg("P5: position 9 + last dW", 18, 19, full)
    seg("P5: dsp reductions + Gu stores", 19, 20)
    seg("P5: stage accumulators + dK tile groups drawn by ticket (to stamp 10)", 20, 10)
# the workgroup's critical path: a barrier releases when its slowest wavefront arrives
arr = [1, 3, 6, 8, 10]   # stamps taken on arrival at the five barriers
rel0 = s[:, :, 0].min(1)
prev = rel0
names_cp = ["P1 (to barrier 1)", "P2 bridge (to barrier 2)", "P3 (to barrier 3)", "P4 dlong (to barrier 4)", "P5 (to barrier 5)"]
tot_cp = 0
for nm, k in zip(names_cp, arr):
    rel = s[:, :, k].max(1)
    print("critical path  %-26s mean %8.0f  p50 %8.0f  max %8.0f" % (nm, (rel - prev).mean(), np.median(rel - prev), (rel - prev).max()))
    tot_cp += (rel - prev).mean()
    prev = rel
endw = s[:, :, 11].max(1)
print("critical path  %-26s mean %8.0f" % ("final reduce", (endw - prev).mean()))
print("critical path  total per workgroup: mean %.0f  max %.0f ticks" % ((endw - rel0).mean(), (endw - rel0).max()))
# s_memtime counters differ between CUs; stamps 28 / 29 are s_memrealtime (100 MHz, device-wide) at the start / end
rt0 = raw[:, :, 28].min(1)
rt1 = raw[:, :, 29].max(1)
t00 = rt0.min()
print("device clock: workgroup starts after the first (us): p50 %.2f p90 %.2f max %.2f;  ends: p10 %.2f p50 %.2f p90 %.2f max %.2f"
      % (tuple(np.percentile(rt0 - t00, [50, 90, 100]) / 100) + tuple(np.percentile(rt1 - t00, [10, 50, 90, 100]) / 100)))
dur = (rt1 - rt0) / 100
print("workgroup durations (us): p10 %.2f p50 %.2f p90 %.2f max %.2f;  cycles per 10 ns tick: %.1f" % (tuple(np.percentile(dur, [10, 50, 90, 100])) + (np.median((s[:, :, 11].max(1) - s[:, :, 0].min(1)) / (rt1 - rt0)),)))
order = np.argsort(rt1)[::-1][:8]
print("last workgroups to end: " + ", ".join("#%d start %.2f dur %.2f" % (b, (rt0[b] - t00) / 100, dur[b]) for b in order))
cp = []
prev = s[:, :, 0].min(1)
for k in arr + [11]:
    rel = s[:, :, k].max(1)
    cp.append(rel - prev)
    prev = rel
cp = np.stack(cp, 1)
print("critical path of the median workgroup (P1 P2 P3 P4 P5 end):", np.median(cp, 0).astype(int))
sl_new = np.asarray(db_host[7])
sl_long = np.asarray(db_host[6])
for b in order[:6]:
    extra = "  sessions " + str(sorted(sl_new[NSB * b:NSB * b + NSB].tolist(), reverse=True)[:6])
    extra += "  windows " + str(sorted(sl_long[NSB * b:NSB * b + NSB].tolist(), reverse=True)[:6])
    print("  workgroup #%d:" % b, cp[b].astype(int), extra)
p1 = d[:, :, 0]
print("P1 percentiles:", np.percentile(p1, [1, 10, 25, 50, 75, 90, 99, 100]).astype(int))
print("P1 mean by XCD (block%8):", [int(p1[x::8].mean()) for x in range(8)])
print("P1 mean by wave:", [int(p1[:, w].mean()) for w in range(NWV)])
blk = p1.mean(1)
print("P1 block-mean percentiles:", np.percentile(blk, [0, 10, 50, 90, 100]).astype(int))
start = s[:, :, 0] - s[:, :, 0].min()
print("wave start-time percentiles (ticks after first):", np.percentile(start, [0, 10, 50, 90, 100]).astype(int))
end = s[:, :, 11] - s[:, :, 0].min()
print("wave end-time percentiles:", np.percentile(end, [0, 10, 50, 90, 100]).astype(int))
