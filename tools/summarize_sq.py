"""Turn the two rocprofv3 --pmc SQ passes of tools/profile_round.sh into a per-kernel issue summary.

SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md, PMC units table), summed over
all SIMDs of the device; GRBM_GUI_ACTIVE counts shader cycles, summed over the 8 XCDs.  Derived, per kernel:
  simd_cycles          = n_simd * GRBM_GUI_ACTIVE / n_xcd            SIMD-cycles the launch lasted
  valu_insts_per_wave  = SQ_INSTS_VALU / SQ_WAVES
  valu_insts_per_simd_cycle = SQ_INSTS_VALU / simd_cycles            an IPC: instructions issued per SIMD per cycle
  valu_insts_x4_frac   = 4 * SQ_INSTS_VALU / simd_cycles             the same, priced at 4 cycles per wave64 instruction
  valu_busy            = 4 * SQ_ACTIVE_INST_VALU / simd_cycles       rocprof's VALUBusy definition
Both "fractions" price an instruction at one quad-cycle = 4 cycles (the counter books one per VALU instruction, two per
transcendental).  That is NOT what the pipe needs on gfx950: tools/micro/valu_rates (profiles/rNN_valu_rates.txt) measures
2.2 cycles per plain fp32 / integer wave64 instruction, 4.6 per fp64, 6.4 per v_mad_u64_u32, 8 per transcendental.  A kernel
whose mix is mostly plain fp32 (the wave-per-trajectory sampling kernels: 72 % of the instructions) can therefore issue MORE
than one instruction per quad-cycle and both numbers exceed 1 (1.19 / 1.21 at d = 128 / 256): they are upper bounds of the issue
time, not fractions.  `valu_issue_bound_frac` = min(1, .): what may be quoted as "share of the launch the vector unit was issuing";
the priced-cycle tables (tools/cycle_table.sh -> profiles/rNN_cycle_table_d*.txt) give the share with the measured per-class costs.
usage: summarize_sq.py <pass1_csv> <pass2_csv> <round_tag> <d,T,B> [kernel-substring ...]
"""
import csv, json, sys

p1, p2, tag, shape = sys.argv[1:5]
want = sys.argv[5:] or ['k_core_', 'k_grad_', 'k_step_']
d, T, B = (int(x) for x in shape.split(','))
N_SIMD, N_XCD = 1024, 8


def collect(path):
    acc, meta = {}, {}
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if not any(w in name for w in want):
            continue
        acc.setdefault(name, {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        m = meta.setdefault(name, {'vgpr': r.get('VGPR_Count'), 'lds_block_bytes': r.get('LDS_Block_Size'),
                                   'grid': r.get('Grid_Size'), 'workgroup': r.get('Workgroup_Size'), 'dur_us': []})
        m['dur_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return acc, meta


out = {'round': tag, 'shape': {'d': d, 'T': T, 'B': B}, 'units': 'SQ_* in quad-cycles or instructions, device totals per launch',
       'kernels': {}}
for path in (p1, p2):
    acc, meta = collect(path)
    for k, c in acc.items():
        e = out['kernels'].setdefault(k, {'counters': {}})
        for cn, v in c.items():
            e['counters'][cn] = sum(v) / len(v)
        m = meta[k]
        e.update(vgpr=m['vgpr'], lds_block_bytes=m['lds_block_bytes'], grid=m['grid'], workgroup=m['workgroup'],
                 dur_us_under_pmc=sum(m['dur_us']) / len(m['dur_us']))
for k, e in out['kernels'].items():
    c = e['counters']
    if c.get('GRBM_GUI_ACTIVE'):
        simd_cycles = N_SIMD * c['GRBM_GUI_ACTIVE'] / N_XCD
        e['simd_cycles'] = simd_cycles
        if 'SQ_ACTIVE_INST_VALU' in c:
            e['valu_busy'] = 4.0 * c['SQ_ACTIVE_INST_VALU'] / simd_cycles
        if 'SQ_INSTS_VALU' in c:
            e['valu_insts_per_simd_cycle'] = c['SQ_INSTS_VALU'] / simd_cycles
            e['valu_insts_x4_frac'] = 4.0 * c['SQ_INSTS_VALU'] / simd_cycles
            e['valu_issue_bound_frac'] = min(1.0, e['valu_insts_x4_frac'])
            if e['valu_insts_x4_frac'] > 1.0:
                e['note'] = ('more than one VALU instruction per quad-cycle and SIMD: plain fp32 wave64 instructions issue in ~2.2 cycles '
                             'on gfx950 (profiles/*_valu_rates.txt), the x4 pricing over-counts -- an upper bound, not a fraction')
    if 'SQ_INSTS_VALU' in c and c.get('SQ_WAVES'):
        e['valu_insts_per_wave'] = c['SQ_INSTS_VALU'] / c['SQ_WAVES']
        e['lds_insts_per_wave'] = c.get('SQ_INSTS_LDS', 0.0) / c['SQ_WAVES']
    if 'SQ_WAIT_INST_ANY' in c and c.get('SQ_WAVE_CYCLES'):
        e['wait_inst_any_frac_of_wave_cycles'] = c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']
json.dump(out, sys.stdout, indent=1)
