"""Static instruction table of the sampling (quad) loop of the packed rollout kernel, by piece (developer tool; no GPU).

The loop body runs exactly floor(d/4) times per matrix row and env step, so its VALU count per quad is an exact dynamic
figure.  The split by piece below is a hand-made ledger (piece -> opcodes it compiles to, read off the listing): the
script compiles the shipped sources, extracts the hot path of the loop from the assembly and CHECKS that the ledger
reproduces the opcode histogram exactly -- when the kernel changes the check fails and the ledger has to be redone.
It covers what the ablation builds of tools/cycle_table.sh cannot separate without changing the trajectories (field
extraction, quad sums, fp64 folds, tile stores / pointers, the score terms); pieces outside the loop (per-step E / F
staging, per-row epilogue, column pass, per-trajectory sums, value) are in the ablation table.
usage: python tools/loop_table.py  > profiles/rNN_loop_table_d21.txt"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'discrete_mean_field_game_amd', 'csrc')
KERNEL = '_ZN3mfg12k_core_smallILb1ELb1ELb1ELi21ELb0EEEvNS_8CoreArgsE'
COST = {'mad64': 6.4, 'trans': 8.0, 'f64': 4.6, 'plain': 2.2}          # cycles per wave64 instruction (profiles/r03_valu_rates.txt)

# piece -> opcode multiset PER QUAD (4 matrix elements per lane)
LEDGER = collections.OrderedDict([
    ('Philox4x32-10 block (one per quad; rounds 1-2 partly loop invariant) + element id',
     {'v_mad_u64_u32': 18, 'v_xor_b32': 34, 'v_add_u32': 1}),
    ('fields of the block: 2 radius uniforms (shift, convert, fma), 2 angles (half-word convert, fma), 4 acceptance integers',
     {'v_lshrrev_b32': 2, 'v_cvt_f32_u32': 8, 'v_fmamk_f32': 4, 'v_and_b32': 2}),
    ('Box-Muller, two pairs: log2 u, sqrt(-.) (the factor 2 ln 2 is folded into c and the squeeze slope); sin, cos; 4 products',
     {'v_log_f32': 2, 'v_mul_f32': 4, 'v_sqrt_f32': 2, 'v_sin_f32': 2, 'v_cos_f32': 2}),
    ('concentration x4: x = pi_j - (pi_i + s), e = E_j F_i, u = 1 + e, log2 u, 1/u, log1p correction (u - 1, e - (u - 1), scale, fma)',
     {'v_sub_f32': 8, 'v_mul_f32': 8, 'v_add_f32': 8, 'v_log_f32': 4, 'v_rcp_f32': 4, 'v_fmac_f32': 4}),
    ('gamma set-up x4: d = alpha scale - 1/3, 9 d, c = rsq(9 d)',
     {'v_fmaak_f32': 8, 'v_rsq_f32': 4}),
    ('Marsaglia-Tsang x4: t = c x, q = x t, q^2, threshold fma, three compares (mask arithmetic is scalar), v = 1 + t(3 + t(3 + t)), y = d v',
     {'v_mul_f32': 16, 'v_fmamk_f32': 4, 'v_cmp_nle_f32': 8, 'v_cmp_gt_f32': 4, 'v_add_f32': 4, 'v_fmaak_f32': 4, 'v_fma_f32': 4}),
    ('h(z) table x4: interval coordinate fma, clamp, convert, 64-bit address, fraction, cubic (3 fma)',
     {'v_fmaak_f32': 4, 'v_med3_f32': 4, 'v_cvt_u32_f32': 4, 'v_lshl_add_u64': 4, 'v_fract_f32': 4, 'v_fma_f32': 8, 'v_fmac_f32': 4}),
    ('score terms x4: sigmoid = e / u, alpha\' = x sigmoid, -x h, log2 y, fma',
     {'v_mul_f32': 12, 'v_log_f32': 4, 'v_fmac_f32': 4}),
    ('quad sums in fp32 (S, A, D, g: three adds each)', {'v_add_f32': 12}),
    ('fold of the quad sums into the fp64 row sums (4 conversions + 4 adds)', {'v_cvt_f64_f32': 4, 'v_add_f64': 4}),
    ('running LDS pointers (tile row, state / E line)', {'v_add_u32': 2}),
    ('merge of the first pair\'s hot / exact-path values (two moves)', {'v_mov_b32': 2}),
])


def klass(op):
    if op.startswith('v_mad_u64') or op.startswith('v_lshl_add_u64'):
        return 'mad64'
    if re.match(r'v_(log|exp|rcp|rsq|sqrt|sin|cos)_f32', op):
        return 'trans'
    if 'f64' in op:
        return 'f64'
    return 'plain'


def main():
    asm = os.path.join(tempfile.mkdtemp(), 'k.s')
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=on', '-Wno-unused-function',
           '-Wno-pass-failed', '-fno-slp-vectorize', '-mllvm', '-amdgpu-sched-strategy=iterative-ilp', '-S', '--cuda-device-only', '-o', asm,
           os.path.join(CSRC, 'mfg_core_small.hip')]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(asm).read().split('\n')
    start = next(i for i, l in enumerate(text) if l.startswith(KERNEL + ':'))
    end = next(i for i in range(start, len(text)) if text[i].startswith('.Lfunc_end'))
    body = text[start:end]
    hdr = next(i for i, l in enumerate(body) if 'This Loop Header: Depth=3' in l)
    for i in range(hdr, 0, -1):
        m = re.match(r'^(\.LBB\d+_\d+):', body[i])
        if m:
            lab, hdr = m.group(1), i
            break
    back = next(i for i in range(hdr, len(body)) if re.match(r'\s*s_branch\s+' + re.escape(lab) + r'\b', body[i]))
    hist = collections.Counter()
    for l in body[hdr:back + 1]:
        m = re.match(r'^\s+([a-z_0-9]+)\s', l + ' ')
        if m and not l.lstrip().startswith((';', '.')):
            hist[re.sub(r'_e32$|_e64$|_sdwa$', '', m.group(1))] += 1
    ledger = collections.Counter()
    for ops in LEDGER.values():
        ledger.update(ops)
    valu = {k: v for k, v in hist.items() if k.startswith('v_')}
    if dict(ledger) != valu:
        diff = {k: (valu.get(k, 0), ledger.get(k, 0)) for k in set(valu) | set(ledger) if valu.get(k, 0) != ledger.get(k, 0)}
        sys.exit('the ledger no longer matches the kernel (opcode: listing, ledger): %r' % diff)
    nv = sum(valu.values())
    print('k_core_small<SAMPLE, TD, MIXED, 21>: hot path of the sampling loop, one iteration = one quad = 4 matrix elements per lane.')
    print('Static count from the listing of the shipped sources; the ledger below reproduces the opcode histogram exactly (checked).')
    print('%-150s %5s %9s %7s %6s' % ('piece', 'VALU', 'per elem', 'cycles', 'share'))
    totc = sum(COST[klass(k)] * v for k, v in valu.items())
    for p, ops in LEDGER.items():
        n = sum(ops.values())
        c = sum(COST[klass(k)] * v for k, v in ops.items())
        print('%-150s %5d %9.2f %7.0f %5.1f%%' % (p, n, n / 4.0, c, 100.0 * c / totc))
    print('%-150s %5d %9.2f %7.0f' % ('TOTAL per quad', nv, nv / 4.0, totc))
    print('besides, per quad: %d scalar / branch instructions (loop control, the exact-path masks), %d LDS instructions (4 reads, 2 writes), %d h-table loads'
          % (sum(v for k, v in hist.items() if k.startswith('s_')), sum(v for k, v in hist.items() if k.startswith('ds_')),
             sum(v for k, v in hist.items() if k.startswith('global_'))))
    print('per matrix row of d = 21: 5 quads = %d VALU instructions here, + the trailing single element (~140 instructions on even steps, ~75 on odd ones: its Box-Muller pair is keyed by the even step, the partner normal carried -- sample_tail1), + the per-row and'
          ' per-step pieces of the ablation table (E / F staging 22 per row, epilogue, column pass, sums, value).' % (5 * nv))
    print('round 3 (same method, listing of commit 33c6160): 273 VALU per quad.')


if __name__ == '__main__':
    main()
