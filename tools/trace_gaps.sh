#!/bin/bash
# usage (GPU box): bash tools/trace_gaps.sh <python script> [args]  -> per-kernel durations and idle gaps of the last 40 launches
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tg
rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o p -- python3 "$@" > /tmp/tg.log 2>&1
f=$(find /tmp/tg -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
rows = rows[-int(__import__("os").environ.get("TG_ROWS", "60")):-10]
prev_end = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print('%-60s dur %7.1f us  gap-before %6.1f us  grid %s' % (r['Kernel_Name'][:60], (e - s) / 1e3, gap, r.get('Grid_Size', r.get('Grid_Size_X', '?'))))
    prev_end = e
PY
