"""Static cycle table of a stretch of gfx950 ISA (developer tool): classify every instruction of the given line ranges of
an assembly listing (hipcc -S --cuda-device-only) and price each class with the issue cost measured by
tools/micro/valu_rates.hip on the same chip.  usage: isa_cycles.py file.s name:lo-hi[,lo-hi...] [name:...]"""
import re, sys, collections

# cycles per wave64 instruction per SIMD (tools/micro/valu_rates.hip on MI355X, 8 waves/SIMD; see profiles/r03_valu_rates.txt)
COST = {'mad_u64': 8.0, 'trans': 8.0, 'f64': 4.0, 'cvt64': 4.0, 'f32': 2.0, 'int': 2.0, 'cmp': 2.0, 'cndmask': 2.0, 'dpp': 2.0,
        'mov': 2.0, 'u64': 4.0, 'lds': 0.0, 'vmem': 0.0, 'salu': 0.0, 'other': 2.0}


def classify(op):
    if op.startswith('v_mad_u64_u32') or op.startswith('v_mad_i64_i32'):
        return 'mad_u64'
    if re.match(r'v_(log|exp|rcp|rsq|sqrt|sin|cos)_f(32|16)', op):
        return 'trans'
    if re.match(r'v_(rcp|rsq|sqrt)_f64', op):
        return 'trans64'
    if re.match(r'v_cvt_(f64_f32|f32_f64|f64_i32|i32_f64|f64_u32|u32_f64)', op):
        return 'cvt64'
    if re.match(r'v_(fma|add|mul|max|min|ldexp|frexp_mant|frexp_exp_i32|fract|trig_preop|div_fmas|div_fixup|div_scale|cmp\w*)_f64', op) or op.endswith('_f64_e32') or op.endswith('_f64_e64') or op.endswith('_f64'):
        return 'f64'
    if re.match(r'v_(lshl_add_u64|lshlrev_b64|lshrrev_b64|ashrrev_i64|add_co|addc_co|sub_co|subb_co|mul_lo_u32|mul_hi_u32)', op):
        return 'u64'
    if op.startswith('v_cmp') or op.startswith('v_cmpx'):
        return 'cmp'
    if op.startswith('v_cndmask'):
        return 'cndmask'
    if op.startswith('v_mov') or op.startswith('v_accvgpr') or op.startswith('v_readlane') or op.startswith('v_readfirstlane') or op.startswith('v_writelane'):
        return 'mov'
    if re.match(r'v_(add|sub|subrev|mul|fma|fmac|fmaak|fmamk|mad|mac|max|min|med3|fract|floor|rndne|ldexp|cvt_\w+|frexp\w*)_?', op) and ('f32' in op or 'f16' in op):
        return 'f32'
    if op.startswith('v_'):
        return 'int'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'):
        return 'vmem'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    lines = open(sys.argv[1]).read().split('\n')
    cost = dict(COST)
    cost['trans64'] = 16.0
    grand = collections.Counter()
    for spec in sys.argv[2:]:
        name, rng = spec.split(':')
        mult = 1.0
        if '*' in name:
            name, m = name.split('*')
            mult = float(m)
        cnt = collections.Counter()
        for part in rng.split(','):
            lo, hi = (int(x) for x in part.split('-'))
            for ln in lines[lo - 1:hi]:
                ln = ln.split(';')[0].strip()
                if not ln or ln.endswith(':') or ln.startswith('.') or ln.startswith(';'):
                    continue
                op = ln.split()[0]
                c = classify(op)
                if 'dpp' in ln or 'row_' in ln or 'quad_perm' in ln:
                    c = 'dpp' if c not in ('salu',) else c
                cnt[c] += 1
        tot_i = sum(v for k, v in cnt.items() if k not in ('lds', 'vmem', 'salu'))
        tot_c = sum(v * cost[k] for k, v in cnt.items())
        print('%s (x%g): %d VALU instructions, %.0f cycles (%.2f cycles/instruction)' % (name, mult, tot_i, tot_c, tot_c / max(tot_i, 1)))
        for k, v in sorted(cnt.items(), key=lambda kv: -kv[1] * cost[kv[0]]):
            print('    %-8s %4d instr  x %4.1f = %6.0f cycles  (%4.1f %%)' % (k, v, cost[k], v * cost[k], 100.0 * v * cost[k] / max(tot_c, 1)))
            grand[k] += v * mult
    if len(sys.argv) > 3:
        tot_c = sum(v * cost[k] for k, v in grand.items())
        tot_i = sum(v for k, v in grand.items() if k not in ('lds', 'vmem', 'salu'))
        print('TOTAL: %.0f VALU instructions, %.0f cycles (%.2f cycles/instruction)' % (tot_i, tot_c, tot_c / tot_i))
        for k, v in sorted(grand.items(), key=lambda kv: -kv[1] * cost[kv[0]]):
            print('    %-8s %7.0f instr  x %4.1f = %8.0f cycles  (%4.1f %%)' % (k, v, cost[k], v * cost[k], 100.0 * v * cost[k] / tot_c))


if __name__ == '__main__':
    main()
