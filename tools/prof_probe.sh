#!/bin/bash
# usage (GPU box): bash tools/prof_probe.sh d,B,T   -> rocprofv3 kernel stats of tools/perf_probe.py at that shape
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python3 $R/tools/perf_probe.py "$@" > /tmp/pp.log 2>&1
tail -6 /tmp/pp.log
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print('%-70s calls %4s avg %10.1f us  %5s%%' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage'][:5]))
PY
