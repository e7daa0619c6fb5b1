#!/usr/bin/env python3
"""Compare two device-assembly dumps (hipcc --cuda-device-only -S) kernel by kernel: instruction streams with
comments and debug directives stripped.  Used to check that a source clean-up changed no generated code.
Usage: asm_diff.py old.s new.s"""
import re, sys

def kernels(path):
    out, cur, name = {}, None, None
    for ln in open(path, errors="replace"):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if cur is None:
            continue
        if ln.startswith(".Lfunc_end"):
            cur = None
            continue
        t = ln.split(";")[0].strip()
        if not t or t.startswith(".") and not t.startswith(".LBB"):
            continue
        cur.append(re.sub(r"\.L(BB|tmp|func_\w+?)\d+_", r".L\1_", t))   # (labels carry the function's ordinal in the unit)
    return out

a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
same = diff = 0
for k in sorted(set(a) | set(b)):
    if k not in a or k not in b:
        print("only in %s: %s" % ("old" if k in a else "new", k)); diff += 1
    elif a[k] != b[k]:
        n = sum(1 for x, y in zip(a[k], b[k]) if x != y) + abs(len(a[k]) - len(b[k]))
        print("DIFF %s: %d vs %d instructions, %d differing lines" % (k, len(a[k]), len(b[k]), n)); diff += 1
    else:
        same += 1
print("%d kernels identical, %d differ" % (same, diff))
