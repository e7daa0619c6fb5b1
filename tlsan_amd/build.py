"""Build libtlsan_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake: six translation
units (the C ABI with every kernel but the fused one; k_fwd_bwd for d = 64 / 128 / 128 as 8-sample workgroups / 256 with
the window in registers / 256 streamed)
compiled in parallel, one link.  `python -m tlsan_amd.build` or `build()`."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libtlsan_hip.so")
SOURCES = ["tlsan_api.hip", "tlsan_attn_d64.hip", "tlsan_attn_d128.hip", "tlsan_attn_d128w4.hip", "tlsan_attn_d256.hip", "tlsan_attn_d256s.hip"]
# per-source extra flags (see the source's header comment)
# -fno-honor-nans on the d <= 128 units: fmaxf on an MFMA result otherwise gets a canonicalising v_max x, x, x in front of it
# (cdna_hip_programming.md, pitfalls): 78 vector instructions of the bf16-operand kernel, 18 of the fp32 one; same results on
# finite data (round 5 A/B, profiles/r05_isa_budget.md)
# -fno-signed-zeros on the same units: 2 722 -> 2 692 vector instructions (bf16 operands 3 019 -> 2 930); step 59.3 -> 58.5 us
# (bf16 50.65 -> 50.2), four interleaved rounds, same loss to the last digit (profiles/r05_ab_fpflags.txt); -ffast-math
# gains no more and reassociates
_NONAN = ["-fno-honor-nans", "-fno-signed-zeros"]   # (no canonicalising v_max before fmaxf; x + 0 / 0 - x folds: the sign of a zero is never looked at)
SOURCE_FLAGS = {"tlsan_attn_d64.hip": _NONAN, "tlsan_attn_d128.hip": _NONAN, "tlsan_attn_d128w4.hip": _NONAN,
                "tlsan_attn_d256.hip": ["-mllvm", "-sink-insts-to-avoid-spills"],
                # (the two FP switches on the streamed d = 256 unit: C5 305-307 -> 302-304 us/step, three interleaved rounds,
                #  profiles/r05_ab_fpflags_d256.txt; nothing on the Ls = 10 unit, which keeps its flags)
                "tlsan_attn_d256s.hip": ["-mllvm", "-disable-machine-licm", "-mllvm", "-sink-insts-to-avoid-spills"] + _NONAN}
if os.environ.get("TLSAN_SOURCE_FLAGS"):   # (experiments: JSON {source: [flags]}, replaces the entries it names)
    import json
    SOURCE_FLAGS.update(json.loads(os.environ["TLSAN_SOURCE_FLAGS"]))
HEADERS = ["tlsan_common.h", "tlsan_attn.h", "tlsan_attn_inst.h", "tlsan_update.h", "tlsan_eval.h", "tlsan_rows.h", "tlsan_shard.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC] + \
        os.environ.get("TLSAN_HIPCC_EXTRA", "").split()   # (experiments: extra compiler flags)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(INCLUDE, "tlsan.h")]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def run(job):
        s, o = job
        cmd = [hipcc] + FLAGS + SOURCE_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr[-4000:]))
        return o

    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
