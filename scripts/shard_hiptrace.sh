#!/bin/bash
# On the GPU box: which runtime calls (memsets, copies, launches) one eager static-shape sharded step makes at one rank --
# a HIP API + kernel + memory-copy trace of scripts/shard_static_prof.py, summarised per call name (and per memset / copy size).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sht
N=${N:-100} AHEAD=${AHEAD:-2} rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d /tmp/sht -- python3 $R/scripts/shard_static_prof.py > /tmp/sht.log 2>&1
tail -1 /tmp/sht.log
python3 - <<'PY'
import csv, glob, collections
steps = 100 + 8
for f in glob.glob('/tmp/sht/**/*hip_api_trace.csv', recursive=True):
    c = collections.Counter(r['Function'] for r in csv.DictReader(open(f)))
    print('HIP API calls per step:')
    for k, v in c.most_common(25):
        print('  %-40s %7.2f' % (k, v / steps))
for f in glob.glob('/tmp/sht/**/*memory_copy_trace.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    c = collections.Counter((r.get('Direction', '?'), ) for r in rows)
    print('memory copies per step:', {k: round(v / steps, 2) for k, v in c.items()})
    print(rows[len(rows) // 2] if rows else None)
PY
python3 - <<'PY'
import csv, glob
for f in glob.glob('/tmp/sht/**/*hip_api_trace.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(list(rows[0].keys()))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    skip = ('hipGetDevice', 'hipGetLastError', '__hipPushCallConfiguration', '__hipPopCallConfiguration', 'hipSetDevice', 'hipThreadExchangeStreamCaptureMode', 'hipDevicePrimaryCtxGetState', 'hipStreamIsCapturing')
    seq = [(r['Function'], r.get('Thread_Id', '?')) for r in rows if r['Function'] not in skip and not r['Function'].startswith('__hipRegister')]
    mid = len(seq) * 2 // 3
    tids = sorted(set(t for _, t in seq))
    for fn, t in seq[mid:mid + 90]:
        print('  T%d %s' % (tids.index(t), fn))
PY
