#!/bin/bash
# usage (GPU box): bash tools/prof_any.sh <python script> [args]  -> rocprofv3 kernel stats (top 12)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pa
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o p -- python3 "$@" > /tmp/pa.log 2>&1
grep -v "^[EW]2026" /tmp/pa.log | tail -4
f=$(find /tmp/pa -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print('%-72s calls %5s avg %9.1f us  min %9.1f  max %9.1f  %5s%%' % (r['Name'][:72], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, r['Percentage'][:5]))
PY
