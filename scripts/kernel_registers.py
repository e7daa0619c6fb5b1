#!/usr/bin/env python3
"""Register / scratch report of every k_fwd_bwd instantiation: compiles the fused kernel's translation units for gfx950
with -Rpass-analysis=kernel-resource-usage (no GPU needed) and prints one line per variant.
  python scripts/kernel_registers.py > profiles/rNN_kernel_registers.txt"""
import os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tlsan_amd", "csrc")
UNITS = ("tlsan_attn_d64", "tlsan_attn_d128", "tlsan_attn_d128w4", "tlsan_attn_d256", "tlsan_attn_d256s")
sys.path.insert(0, ROOT)
from tlsan_amd.build import SOURCE_FLAGS   # (per-source compiler flags of the product build)
def run(u):
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                            "-I" + CSRC] + SOURCE_FLAGS.get(u + ".hip", []) + ["--cuda-device-only", "-c", os.path.join(CSRC, u + ".hip"), "-o", os.path.join(td, "o.o"),
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        return r.stderr
with ThreadPoolExecutor(4) as ex:
    texts = list(ex.map(run, UNITS))
out = []
for txt in texts:
    for n, v, a, sc, occ, ss, vs in re.findall(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)", txt, re.S):
        m = re.match(r"_Z9k_fwd_bwdILi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELi(\d)ELb(\d)ELi(\d)ELb(\d)ELi(\d)E", n)
        if m:
            D, DH, tr, ls, dt, dr, mm, cs, nw = map(int, m.groups())
            out.append((D, tr, ls, dt, dr, mm, cs, nw, int(v), int(a), int(sc), int(occ), int(ss), int(vs)))
out.sort()
print("# k_fwd_bwd<D, D/8, TRAIN, LSTREAM, table bf16, DROP, matrix bf16, CSEG, NW> -- hipcc -Rpass-analysis=kernel-resource-usage, gfx950")
print("%4s %5s %7s %4s %4s %4s %4s %3s | %5s %5s %12s %4s %10s %10s" % ("D", "TRAIN", "LSTREAM", "tbf", "DROP", "mbf", "CSEG", "NW", "VGPR", "AGPR", "scratch B/ln", "occ", "SGPR spill", "VGPR spill"))
for r in out:
    print("%4d %5d %7d %4d %4d %4d %4d %3d | %5d %5d %12d %4d %10d %10d" % r)
