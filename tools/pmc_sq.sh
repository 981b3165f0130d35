#!/bin/bash
# usage: tools/pmc_sq.sh <kernel-substring> <python driver> [driver args]   (run on the GPU box)
R=$GRAFT_REPO_ROOT; K=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p2
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/p1 -o p -- python3 "$@" > /tmp/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p2 -o p -- python3 "$@" > /tmp/p2.log 2>&1
python3 - "$K" <<'PY'
import csv, collections, sys, glob
K = sys.argv[1]
for d in ("/tmp/p1", "/tmp/p2"):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        print(d, "no counter file"); print(open(d + ".log").read()[-800:]); continue
    rows = list(csv.DictReader(open(files[0])))
    acc = collections.defaultdict(list)
    for r in rows:
        if K in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%-24s %.4g" % (k, sum(v) / len(v)))
    for r in rows:
        if K in r["Kernel_Name"]:
            print("VGPR", r["VGPR_Count"], "LDS", r["LDS_Block_Size"], "grid", r["Grid_Size"], "wg", r["Workgroup_Size"],
                  "dur_us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            break
PY
