#!/bin/bash
# Cycle table of the fused rollout kernels (GPU box): the DYNAMIC instruction-class mix from the SQ class counters
# (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32/F64, INT32, INT64, CVT; separate --pmc passes, --kernel-trace only), priced with
# the per-class issue costs measured by tools/micro/valu_rates on the same chip, against the MEASURED VALU-active cycles
# (SQ_ACTIVE_INST_VALU, quad-cycles x 4).  Optional ablation builds (tools/ablate.sh build) split the cycles by piece.
# usage: bash tools/cycle_table.sh d,T,B > gpurun_out/<tag>/cycle_table_d<d>.txt
R=$GRAFT_REPO_ROOT; SH=${1:-21,15,65536}; V=${MFG_VARIANT_DIR:-$R/discrete_mean_field_game_amd/csrc/variants}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_BUSY_CYCLES"
P2="SQ_WAVES SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"
P3="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"
run() {  # $1 = out dir, $2 = counters, $3 = library ('' = shipped)
  rm -rf $1
  if [ -n "$3" ]; then export MFG_HIP_LIB=$3; else unset MFG_HIP_LIB; fi
  rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $1 -o p -- python3 $R/tools/pmc_rollout.py $SH > $1.log 2>&1
}
run /tmp/ct1 "$P1" ""; run /tmp/ct2 "$P2" ""; run /tmp/ct3 "$P3" ""
# the kernel's WARM duration (the three launches of a PMC pass run on a device that is still ramping its clock, ~20 % slower):
# a kernel-trace-only pass with enough launches, fastest launch
rm -rf /tmp/ct0; unset MFG_HIP_LIB
PMC_LAUNCHES=$([ "${SH%%,*}" -le 64 ] && echo 40 || echo 4) rocprofv3 --kernel-trace --output-format csv -d /tmp/ct0 -o p -- python3 $R/tools/pmc_rollout.py $SH > /tmp/ct0.log 2>&1
DUR0=$(python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/ct0/**/*kernel_trace.csv', recursive=True)
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f[0])) if 'k_core_' in r['Kernel_Name']] if f else []
print(min(d) if d else 0.0)
PY
)
export CT_DUR0=$DUR0
ABL=""
for a in PHILOX BM SETUP TRY HTAB LNY EPI COLT TSUM V; do
  if [ -f $V/libabl_$a.so ]; then run /tmp/cta_$a "$P1" $V/libabl_$a.so; ABL="$ABL $a"; fi
done
python3 - "$SH" $ABL <<'PY'
import csv, glob, collections, sys
shape = sys.argv[1]; abl = sys.argv[2:]
d, T, B = (int(x) for x in shape.split(','))
def load(dirn):
    f = glob.glob(dirn + '/**/*counter_collection.csv', recursive=True)
    acc = collections.defaultdict(list); dur = []
    if not f: return {}, 0.0
    for r in csv.DictReader(open(f[0])):
        if 'k_core_' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    seen = set()
    for r in csv.DictReader(open(f[0])):
        if 'k_core_' in r['Kernel_Name'] and r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id']); dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return {k: sum(v) / len(v) for k, v in acc.items()}, (sum(dur) / len(dur) if dur else 0.0)
m = {}
durs = []
for dn in ('/tmp/ct1', '/tmp/ct2', '/tmp/ct3'):
    c, du = load(dn); m.update(c); durs.append(du)
# issue cost per wave64 instruction per SIMD, cycles (tools/micro/valu_rates.hip, MI355X, loop glue subtracted)
COST = collections.OrderedDict([('FMA_F32', 2.2), ('ADD_F32', 2.2), ('MUL_F32', 2.2), ('INT32', 2.2), ('CVT', 2.2), ('TRANS_F32', 8.0),
                                ('INT64', 6.4), ('FMA_F64', 4.6), ('ADD_F64', 4.6), ('MUL_F64', 4.6), ('TRANS_F64', 16.0)])
W = m.get('SQ_WAVES', 1.0)
tot_i = m.get('SQ_INSTS_VALU', 0.0); act = 4.0 * m.get('SQ_ACTIVE_INST_VALU', 0.0)
elems = (d * T) if d <= 64 else ((d * d // 64) * T)   # matrix elements per lane per launch
GHZ, NSIMD = 2.4, 1024
import os
dur = float(os.environ.get('CT_DUR0', '0')) or min(x for x in durs if x > 0)
budget = dur * 1e-6 * GHZ * 1e9 * NSIMD / W              # SIMD cycles available per wave over the kernel's duration
print('shape d,T,B = %s   kernel time %.1f us without counters (%.1f / %.1f / %.1f us in the PMC passes)   waves %.0f   matrix elements per lane per launch %d' % (shape, dur, *durs, W, elems))
print('VALU instructions per wave %.0f (%.1f per element)' % (tot_i / W, tot_i / W / elems))
print('%-34s %12s %10s %8s %12s %8s' % ('class (SQ_INSTS_VALU_*)', 'instr/wave', 'per elem', 'cost', 'cycles/wave', 'share'))
rows = []; known_i = 0.0
for k, c in COST.items():
    v = m.get('SQ_INSTS_VALU_' + k, 0.0)
    rows.append((k, v, c)); known_i += v
rest = tot_i - known_i
rows.append(('not in a class counter (logic, moves, selects, compares, DPP, readlane: plain 32-bit)', rest, 2.2))
priced = sum(v * c for _, v, c in rows)
for k, v, c in rows:
    print('%-34s %12.0f %10.2f %8.1f %12.0f %7.1f%%' % (k[:34], v / W, v / W / elems, c, v * c / W, 100 * v * c / priced))
    if len(k) > 34: print('    (%s)' % k)
print('%-34s %12.0f %10.2f %8.2f %12.0f %7.1f%%' % ('TOTAL priced with measured costs', tot_i / W, tot_i / W / elems, priced / tot_i, priced / W, 100.0))
print('SIMD cycles available per wave (kernel time x %.1f GHz x %d SIMDs / waves): %.0f  =>  the priced VALU issue work fills %.0f %% of them'
      % (GHZ, NSIMD, budget, 100 * priced / W / budget))
trans = m.get('SQ_INSTS_VALU_TRANS_F32', 0.0) + m.get('SQ_INSTS_VALU_TRANS_F64', 0.0)
print('SQ_ACTIVE_INST_VALU x 4 = %.0f per wave = %.2f per instruction.  NOT an execution time: the counter books one quad-cycle (4 cycles) per '
      'VALU instruction issued and two per transcendental, whatever the pipe does afterwards -- 4 + 4 x (transcendental share %.4f) = %.2f; '
      'the pieces below that contain no transcendental (fp32-only HTAB, fp64-only COLREW / V, Philox with its 6.4-cycle v_mad_u64_u32) all '
      'come out at exactly 4.00.' % (act / W, act / max(tot_i, 1), trans / max(tot_i, 1), 4 + 4 * trans / max(tot_i, 1)))
if abl:
    print('\nby piece (shipped build minus the build with that piece replaced by a 1-4 instruction stand-in, tools/ablate.sh):')
    print('%-10s %14s %14s %10s %12s' % ('piece', 'instr/elem', 'counter/elem', 'ctr/instr', 'share of instr'))
    ti = tc = 0.0
    for a in abl:
        c, _ = load('/tmp/cta_' + a)
        if not c: continue
        di = (tot_i - c.get('SQ_INSTS_VALU', 0.0) * W / max(c.get('SQ_WAVES', W), 1)) / W / elems
        dc = (act - 4.0 * c.get('SQ_ACTIVE_INST_VALU', 0.0) * W / max(c.get('SQ_WAVES', W), 1)) / W / elems
        ti += di; tc += dc
        print('%-10s %14.2f %14.1f %10.2f %11.1f%%' % (a, di, dc, dc / di if di else 0.0, 100 * di * W * elems / max(tot_i, 1)))
    print('(the rest = what no stand-in can replace without changing the trajectories: field extraction of the Philox block, score terms,'
          ' quad sums, fp64 folds, pointers -- split statically, piece by piece, in profiles/rNN_loop_table_d21.txt (tools/loop_table.py) --'
          ' plus per-step E / F staging, state loads / stores, trajectory outputs)')
    print('%-10s %14.2f %14.1f %10.2f %11.1f%%' % ('the rest', tot_i / W / elems - ti, act / W / elems - tc,
                                                 (act / W / elems - tc) / max(tot_i / W / elems - ti, 1e-9), 100 * (1 - ti * W * elems / max(tot_i, 1))))
PY
