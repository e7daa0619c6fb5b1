#!/usr/bin/env python3
"""Compile the d=128 training kernel with -save-temps and print, per stamped phase, the sequence
of loads (L), atomics (A), waits ([v(n)]/[k(n)]), MFMAs (M), LDS ops (d), branches (|), exps (e),
stores (S) and runs of other VALU (.n.).  Usage: python scripts/isa_seq.py [phase ...]"""
import os, re, subprocess, sys, tempfile, glob
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + root + "/include", "-I" + root + "/tlsan_amd/csrc",
                "-fPIC", "-c", root + "/tlsan_amd/csrc/tlsan_attn_d128.hip", "-save-temps", "-o", "x.o"], cwd=tmp,
               stderr=subprocess.DEVNULL)
src = open(glob.glob(tmp + "/*gfx950*.s")[0]).read().split("\n")
a = next(i for i, l in enumerate(src) if l.startswith("_Z9k_fwd_bwdILi128ELi16ELb1ELb0ELi0ELb0ELi0EEv7FwdArgs:"))
b = next(i for i in range(a, len(src)) if "s_endpgm" in src[i])
lines = src[a:b]
marks = [i for i, l in enumerate(lines) if "s_memtime" in l]
print("instructions:", sum(1 for l in lines if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")), "stamps at", marks)
def seq(a, b):
    out = []
    for l in lines[a:b]:
        t = l.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op.startswith("global_load") or op.startswith("scratch_load"): out.append("L")
        elif op.startswith("global_atomic"): out.append("A")
        elif op.startswith("global_store") or op.startswith("scratch_store"): out.append("S")
        elif op.startswith("s_waitcnt"): out.append("[" + t.split(None, 1)[1].replace("vmcnt", "v").replace("lgkmcnt", "k").replace(" ", "") + "]")
        elif op.startswith("v_mfma"): out.append("M")
        elif op.startswith("ds_"): out.append("d")
        elif op.startswith("s_cbranch"): out.append("|")
        elif op.startswith("s_barrier"): out.append("#")
        elif op.startswith("v_exp"): out.append("e")
        elif op.startswith("v_"): out.append(".")
    s = "".join(out)
    return re.sub(r"\.{4,}", lambda m: ".%d." % len(m.group()), s)
want = [int(x) for x in sys.argv[1:]] or range(len(marks) - 1)
for k in want:
    print("--- phase", k, "(lines %d)" % (marks[k + 1] - marks[k]))
    print(seq(marks[k], marks[k + 1]))
