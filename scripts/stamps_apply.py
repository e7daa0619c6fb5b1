#!/usr/bin/env python3
"""Diagnostic: per-workgroup s_memtime stamps of k_apply (start times, phase durations per block type)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tlsan_amd import _lib as L, synth
from tlsan_amd.model import Model
B = int(sys.argv[1]) if len(sys.argv) > 1 and "=" not in sys.argv[1] else 4096
kw = dict(x.split("=") for x in sys.argv[1:] if "=" in x)     # e.g. U=35896 I=28589 C=15 (Movies-TV shape)
dh = int(kw.get("d", 128))
cfg = synth.make_config("electronics", user_count=int(kw.get("U", 39991)), item_count=int(kw.get("I", 22048)),
                        cate_count=int(kw.get("C", 673)), Ls=int(kw.get("Ls", 10)), hidden_units=dh, itemid_embedding_size=dh // 2,
                        userid_embedding_size=dh // 2, cateid_embedding_size=dh // 2)
m = Model(cfg, synth.item_cate_list(cfg), l2_mode="lazy")
lib = L.load()
hb = synth.make_batches(cfg, 1, B, seed=7)[0]
db = m.device_batch(hb)
for _ in range(5):
    m.train_async(db, 1.0)
OFF = 1 << 20
NBLK = 65536
st = torch.zeros(OFF + NBLK * 8, dtype=torch.int64, device="cuda")
lib.tlsan_debug_stamps(st.data_ptr())
m.train_async(db, 1.0)
torch.cuda.synchronize()
lib.tlsan_debug_stamps(None)
s = st.cpu().numpy()[OFF:].reshape(NBLK, 8).astype(np.float64)
used = s[:, 0] > 0
n = int(np.flatnonzero(used).max()) + 1   # (lazy row blocks past the used rows leave no stamps)
s = s[:n]
t0 = s[used[:n], 0].min()
C = cfg["cate_count"]
Sn = db.Sn
uses = -(-(B * (cfg["Ls"] + db.Sn + 2)) // C)
ni = min(cfg["item_count"], B * (cfg["Ls"] + Sn + 1))
nbI = (ni + 15) // 16
nbU = (min(B, cfg["user_count"]) + 15) // 16
# workgroups of the category part (category_split, tlsan_api.hip; TLSAN_CSPLIT_FINE)
nsh = min(64, uses // 128) if uses > 512 else 1
if uses > 96 and os.environ.get("TLSAN_CSPLIT_FINE", "1") != "0":
    nsh = max(nsh, min(64, uses // 48, (1280 - (206 + nbI + nbU)) // C))
C = C * max(nsh, 1)
if cfg["cate_count"] >= int(os.environ.get("TLSAN_CSEG_MIN", 2048)):      # category segments: 16 categories per workgroup
    C = (cfg["cate_count"] + 15) // 16
print("blocks %d: cate %d, item %d, user %d, dense %d; span %.0f ticks (100 MHz -> %.1f us)" % (n, C, nbI, nbU, n - C - nbI - nbU, s[:, 6].max() - t0, (s[:, 6].max() - t0) / 100))
# the one-pass row launch places the user-row workgroups ahead of the item-row workgroups where rows are wide (tlsan_api.hip:
# ApplyArgs.ufirst, and the shared-category form's own rule), and launches at most SPEC_ITEM_BLOCKS item-row workgroups
WU = (cfg["hidden_units"] // 2 + cfg["Ls"] + 3) // 4 * 4
ufirst = (cfg["hidden_units"] // 2 > 64 or WU > 128) and os.environ.get("TLSAN_LAZY_ONE_PASS", "1") != "0"
cap = int(os.environ.get("TLSAN_SPEC_ITEM_BLOCKS", 2048))
if cap > 0 and nbI > cap and nsh <= 1 and os.environ.get("TLSAN_LAZY_ONE_PASS", "1") != "0":
    nbI = cap
ROLES = [("cate", 0, C), ("user", C, C + nbU), ("item", C + nbU, C + nbU + nbI)] if ufirst else [("cate", 0, C), ("item", C, C + nbI), ("user", C + nbI, C + nbI + nbU)]
def show(name, lo, hi):
    x = s[lo:hi]
    x = x[x[:, 0] > 0]
    if len(x) == 0:
        return
    print("%-6s start p0/50/100 %5.0f %5.0f %5.0f | end p50/100 %5.0f %5.0f | dur p50 %5.0f max %5.0f" % (
        name, *(np.percentile(x[:, 0] - t0, [0, 50, 100])), *(np.percentile(x[:, 6] - t0, [50, 100])),
        np.median(x[:, 6] - x[:, 0]), (x[:, 6] - x[:, 0]).max()))
    cols = [0, 1, 2, 3, 6]
    y = x[:, cols]
    ok = (y > 0).all(1)
    if ok.any():
        d = np.diff(y[ok], axis=1)
        print("        phases (median / p90 ticks): " + " ".join("%d-%d: %.0f/%.0f" % (cols[i], cols[i + 1], np.median(d[:, i]), np.percentile(d[:, i], 90)) for i in range(4)))
for nm_, lo_, hi_ in ROLES:
    show(nm_, lo_, hi_)
show("dense/fin", C + nbI + nbU, n)

# timeline on the device-wide 100 MHz clock (slots 4/5)
r0 = s[s[:, 4] > 0, 4].min()
print("timeline (us after the first workgroup start; 100 MHz s_memrealtime): kernel span %.2f us" % ((s[:, 5].max() - r0) / 100))
for nm, lo, hi in ROLES + [("dense/fin", C + nbI + nbU, n)]:
    x = s[lo:hi]
    x = x[x[:, 4] > 0]
    if len(x):
        st, en = (x[:, 4] - r0) / 100, (x[:, 5] - r0) / 100
        print("  %-9s start p0/10/50/90/100 %s | end p50/90/100 %s | dur p50 %.2f" % (
            nm, " ".join("%5.2f" % v for v in np.percentile(st, [0, 10, 50, 90, 100])),
            " ".join("%5.2f" % v for v in np.percentile(en, [50, 90, 100])), np.median(en - st)))
# how many workgroups run at once (the launch's residency): running blocks sampled every 0.5 us
x = s[s[:, 4] > 0]
st, en = (x[:, 4] - r0) / 100, (x[:, 5] - r0) / 100
print("resident workgroups at t (us): " + "  ".join("%.1f:%d" % (t, int(((st <= t) & (en > t)).sum())) for t in np.arange(0.5, en.max(), 1.0)))
# the category blocks that end last, with their use counts (item uses of the category's items + its u_cate uses)
if C == cfg["cate_count"]:
    icl = synth.item_cate_list(cfg)
    u_, i_, y_, hi_, hn_, ht_, sl_, sn_, c_ = hb
    ids = [np.asarray(i_)]
    ids += [np.asarray(hi_)[k, :int(sl_[k])] for k in range(B)]
    ids += [np.asarray(hn_)[k, :int(sn_[k])] for k in range(B)]
    ids = np.concatenate(ids)
    uses = np.bincount(icl[ids], minlength=C) + np.bincount(np.asarray(c_), minlength=C)
    x = s[:C]
    en = (x[:, 5] - r0) / 100
    order = np.argsort(-en)[:8]
    print("category blocks by end time: " + "  ".join("c%d end %.2f dur %.2f uses %d items %d" % (k, en[k], (x[k, 5] - x[k, 4]) / 100, uses[k], int((icl == k).sum())) for k in order))
    print("uses per category: p50 %d p90 %d max %d;  duration vs uses correlation %.2f" % (np.median(uses), np.percentile(uses, 90), uses.max(), np.corrcoef(uses, (x[:, 5] - x[:, 4]))[0, 1]))
# the finalize workgroups by role (they are listed last, in launch order: dK entry blocks, small-parameter blocks, the
# block that folds the sum-of-squares records); the last one to arrive also writes the step summary
D = cfg["hidden_units"]
nbK = D * D // 256
x = s[C + nbI + nbU:]
x = x[x[:, 4] > 0]
if len(x) > nbK + 1:
    en = (x[:, 5] - r0) / 100
    stt = (x[:, 4] - r0) / 100
    for nm, lo, hi in [("dK entries", 0, nbK), ("small params", nbK, len(x) - 1), ("fold", len(x) - 1, len(x))]:
        e, d_ = en[lo:hi], (en - stt)[lo:hi]
        print("  finalize %-12s n %3d | end p50/90/100 %5.2f %5.2f %5.2f | dur p50/90/100 %5.2f %5.2f %5.2f" % (
            nm, hi - lo, *np.percentile(e, [50, 90, 100]), *np.percentile(d_, [50, 90, 100])))
    k = int(np.argmax(en))
    print("  last to end: finalize block %d (end %.2f, dur %.2f); second %.2f" % (k, en[k], en[k] - stt[k], np.sort(en)[-2]))
