"""Static instruction ledger of the wave-per-trajectory rollout kernel k_core_large<R, SAMPLE, TD, MIXED, FULL> at the BASELINE
shapes d = 256 (R = 4, config C5) and d = 128 (R = 2, config C3): the hot path of the row loop (one iteration = one Philox
block = 4 matrix elements per lane), the pair merge of the transposed row sums, and the per-batch phase (row totals,
reciprocal, normalise + fold into the column sums).  Developer tool, no GPU.

Like tools/loop_table.py (d = 21) the split by piece is a hand-made ledger (piece -> the opcodes it compiles to, read off the
listing) that the script CHECKS against the opcode histogram of the compiled listing: when the kernel changes the check fails.
Together the three blocks are 70.7 (d = 256) / 72.4 (d = 128) of the 71.8 / 74.5 VALU instructions per matrix element that the SQ
counters measure (profiles/rNN_cycle_table_d256.txt / _d128.txt): what round 4's tables carried as ONE bucket of 35 instructions
per element ("the rest", 49 %) is every piece below that has no ablation stand-in.
usage: python tools/loop_table_large.py [--dump]  > profiles/rNN_loop_table_large.txt"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'discrete_mean_field_game_amd', 'csrc')
COST = {'mad64': 6.4, 'trans': 8.0, 'f64': 4.6, 'plain': 2.2}          # cycles per wave64 instruction (profiles/r03_valu_rates.txt)

QUAD = collections.OrderedDict([    # the sampling of one quad: the pieces of tools/loop_table.py, same opcodes
    ('Philox4x32-10 block (one per quad; rounds 1-2 partly loop invariant) + element ids',
     {'v_mad_u64_u32': 18, 'v_xor_b32': 34}),
    ('fields of the block: 2 radius uniforms, 2 angles, 4 acceptance integers',
     {'v_lshrrev_b32': 2, 'v_cvt_f32_u32': 8, 'v_fmamk_f32': 4, 'v_and_b32': 2}),
    ('Box-Muller, two pairs: log2 u, sqrt, sin, cos, 4 products',
     {'v_log_f32': 2, 'v_mul_f32': 4, 'v_sqrt_f32': 2, 'v_sin_f32': 2, 'v_cos_f32': 2}),
    ('concentration x4: x = pi_j - (pi_i + s), e = E_j F_i, u = 1 + e, log2 u, 1/u, log1p correction',
     {'v_sub_f32': 8, 'v_mul_f32': 8, 'v_add_f32': 8, 'v_log_f32': 4, 'v_rcp_f32': 4, 'v_fmac_f32': 4}),
    ('gamma set-up x4: d = alpha scale - 1/3, 9 d, c = rsq(9 d)',
     {'v_fmaak_f32': 8, 'v_rsq_f32': 4}),
    ('Marsaglia-Tsang x4: t, q, q^2, threshold fma, three compares (mask arithmetic is scalar), v, y = d v',
     {'v_mul_f32': 16, 'v_fmamk_f32': 4, 'v_cmp_nle_f32': 8, 'v_cmp_gt_f32': 4, 'v_add_f32': 4, 'v_fmaak_f32': 4, 'v_fma_f32': 4}),
    ('h(z) table x4: interval coordinate, clamp, convert, 64-bit address, fraction, cubic',
     {'v_fmaak_f32': 4, 'v_med3_f32': 4, 'v_cvt_u32_f32': 4, 'v_lshl_add_u64': 4, 'v_fract_f32': 4, 'v_fma_f32': 8, 'v_fmac_f32': 4}),
    ('score terms x4: sigmoid = e / u, alpha\' = x sigmoid, -x h, log2 y, fma',
     {'v_mul_f32': 12, 'v_log_f32': 4, 'v_fmac_f32': 4}),
])

# kernel -> (mangled name, d, elements per lane per row iteration, rows per batch, what differs from the common quad)
KERNELS = collections.OrderedDict([
    ('d = 256 (R = 4: a row iteration = ONE row, its four columns c, c+64, c+128, c+192 of the lane)', dict(
        name='_ZN3mfg12k_core_largeILi4ELb1ELb1ELb1ELb1EEEvNS_8CoreArgsE', d=256, rows_per_iter=1, batch_rows=8, per_lane_row=4,
        row_extra=collections.OrderedDict([
            ('element ids of the quad (row base + lane, +64, +128, +192)', {'v_add_u32': 3, 'v_lshl_add_u32': 1}),
            ('row sums of the quad in fp32 (S, A, D, g: three adds each) + the running fp32 score sum of the batch', {'v_add_f32': 14}),
            ('hot / exact-path merges and the parked row of a pair (moves)', {'v_mov_b32': 5}),
        ]),
        merge=collections.OrderedDict([
            ('pair merge of the transposed row sums, every second row: 3 sums x (row_ror:8 add + bank-masked add + 2 butterfly steps) + deposit',
             {'v_add_f32_dpp': 12, 'v_cndmask_b32': 3}),
        ]),
        batch=collections.OrderedDict([
            ('normalise + fold 8 rows x 4 columns into the column sums: p = y * (1/S) (fp32), convert, u = p pi_i, pi\' += u, s1 += u p, s2 += u^2',
             {'v_mul_f32': 32, 'v_cvt_f64_f32': 32, 'v_mul_f64': 32, 'v_add_f64': 32, 'v_fmac_f64': 64}),
            ('state entries of the batch as fp64 (8), score sum of the batch into the fp64 accumulator (convert + add)', {'v_cvt_f64_f32': 9, 'v_add_f64': 1}),
            ('batch totals of S, A, D in one packed register: last DPP / permlane steps, publish, reciprocal + Newton step, 8 read-lanes',
             {'v_add_f32_dpp': 3, 'v_add_f32': 3, 'v_permlane16_swap_b32': 2, 'v_permlane32_swap_b32': 1, 'v_rcp_f32': 1, 'v_fma_f32': 1,
              'v_fmac_f32': 1, 'v_readlane_b32': 8, 'v_mov_b32': 2, 'v_or_b32': 1, 'v_mad_u64_u32': 1, 'v_lshl_add_u32': 1}),
        ]))),
    ('d = 128 (R = 2: a row iteration = TWO rows i, i+1, columns c, c+64 of the lane; the pair is merged in line)', dict(
        name='_ZN3mfg12k_core_largeILi2ELb1ELb1ELb1ELb1EEEvNS_8CoreArgsE', d=128, rows_per_iter=2, batch_rows=8, per_lane_row=2,
        row_extra=collections.OrderedDict([
            ('element ids of the quad (two row bases + lane, +64)', {'v_add_u32': 3, 'v_lshl_add_u32': 1}),
            ('row sums of the two rows in fp32 (S, A, D per row, g of the quad) + the running fp32 score sum of the batch', {'v_add_f32': 12}),
            ('pair merge of the transposed row sums (every iteration): 3 sums x (row_ror:8 add + bank-masked add + 2 butterfly steps) + deposit',
             {'v_add_f32_dpp': 12, 'v_cndmask_b32': 3}),
            ('hot / exact-path merges (moves)', {'v_mov_b32': 2}),
        ]),
        merge=collections.OrderedDict(),
        batch=collections.OrderedDict([
            ('normalise + fold 8 rows x 2 columns into the column sums: p = y * (1/S) (fp32), convert, u = p pi_i, pi\' += u, s1 += u p, s2 += u^2',
             {'v_mul_f32': 16, 'v_cvt_f64_f32': 16, 'v_mul_f64': 16, 'v_add_f64': 16, 'v_fmac_f64': 32}),
            ('state entries of the batch as fp64 (8), score sum of the batch into the fp64 accumulator (convert + add)', {'v_cvt_f64_f32': 9, 'v_add_f64': 1}),
            ('batch totals of S, A, D in one packed register: last DPP / permlane steps, publish, reciprocal + Newton step, 8 read-lanes',
             {'v_add_f32_dpp': 3, 'v_add_f32': 3, 'v_permlane16_swap_b32': 2, 'v_permlane32_swap_b32': 1, 'v_rcp_f32': 1, 'v_fma_f32': 1,
              'v_fmac_f32': 1, 'v_readlane_b32': 8, 'v_mov_b32': 2, 'v_or_b32': 1, 'v_mad_u64_u32': 1, 'v_lshl_add_u32': 1}),
        ]))),
])


def klass(op):
    if op.startswith('v_mad_u64') or op.startswith('v_lshl_add_u64'):
        return 'mad64'
    if re.match(r'v_(log|exp|rcp|rsq|sqrt|sin|cos)_f32', op):
        return 'trans'
    if 'f64' in op:
        return 'f64'
    return 'plain'


def listing():
    asm = os.path.join(tempfile.mkdtemp(), 'k.s')
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=on', '-Wno-unused-function',
           '-Wno-pass-failed', '-fno-slp-vectorize', '-mllvm', '-amdgpu-sched-strategy=iterative-ilp', '-S', '--cuda-device-only', '-o', asm,
           os.path.join(CSRC, 'mfg_core_large_mixed_ilp.hip')]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(asm).read().split('\n')


def hist(lines):
    h = collections.Counter()
    for l in lines:
        m = re.match(r'^\s+([a-z_0-9]+)\s', l + ' ')
        if m and not l.lstrip().startswith((';', '.')):
            h[re.sub(r'_e32$|_e64$|_sdwa$', '', m.group(1))] += 1
    return h


def blocks(text, name):
    """The three hot blocks of the kernel's listing: (row path, pair merge, per-batch phase) as line lists.  Layout relied on (and
    asserted): depth-3 loop = row batches, depth-4 loop = the rows of a batch (`#pragma unroll 1`), depth-5 loops = the cold
    exact-acceptance continuations behind exec-masked branches AFTER the back edge of the row loop."""
    start = next(i for i, l in enumerate(text) if l.startswith(name + ':'))
    end = next(i for i in range(start, len(text)) if text[i].startswith('.Lfunc_end'))
    b = text[start:end]
    lab_at = {m.group(1): i for i, l in enumerate(b) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    h4 = next(i for i, l in enumerate(b) if 'This Loop Header: Depth=4' in l) - 3
    while not re.match(r'^\.LBB\d+_\d+:', b[h4]):
        h4 += 1
    hdr4 = re.match(r'^(\.LBB\d+_\d+):', b[h4]).group(1)
    h3 = max(i for i in lab_at.values() if i < h4 and 'This Loop Header: Depth=3' in ''.join(b[i:i + 4]))
    # blocks of the row loop that sit in front of its header (R = 4: the pair merge of every second row and the increment /
    # exit test; R = 2 merges every iteration, in line: none)
    pre = sorted(i for l, i in lab_at.items() if h3 < i < h4 and ('Header=' + hdr4[2:]) in b[i])
    assert len(pre) in (0, 2), pre
    if pre:
        merge, incr = b[pre[0]:pre[1]], b[pre[1]:h4]
        assert any('dpp' in l or 'row_ror' in l for l in merge)
        back_to = [l for l, i in lab_at.items() if i == pre[1]][0]
    else:
        merge, incr, back_to = [], [], hdr4
    # hot path of a row iteration: header .. the back edge
    back = next(i for i in range(h4, len(b)) if re.match(r'\s*s_branch\s+' + re.escape(back_to) + r'\b', b[i]))
    row = incr + b[h4:back + 1]
    exit_lab = [re.match(r'\s*s_cbranch_scc1\s+(\.LBB\d+_\d+)', l).group(1) for l in row if re.match(r'\s*s_cbranch_scc1', l)]
    exit_lab = exit_lab[0] if pre else exit_lab[-1]
    # per-batch phase: depth-3 header .. the row loop; exit block .. the variant WITHOUT the action write .. batch increment
    ex = lab_at[exit_lab]
    first_execz = next(i for i in range(ex, len(b)) if re.match(r'\s*s_cbranch_execz\s+(\.LBB\d+_\d+)', b[i]))
    nowrite = lab_at[re.match(r'\s*s_cbranch_execz\s+(\.LBB\d+_\d+)', b[first_execz]).group(1)]
    back3 = next(i for i in range(nowrite, len(b)) if re.match(r'\s*s_branch\s+\.LBB', b[i]))
    inc3 = lab_at[re.match(r'\s*s_branch\s+(\.LBB\d+_\d+)', b[back3]).group(1)]
    into4 = [i for i in range(h3, h4) if re.match(r'\s*s_branch\s+' + re.escape(hdr4) + r'\b', b[i])]
    into4 = into4[0] if into4 else (pre[0] - 1 if pre else h4 - 1)
    batch = b[inc3:into4 + 1] + b[ex:first_execz + 1] + b[nowrite:back3 + 1]
    return row, merge, batch


def check(label, lines, ledger):
    valu = {k: v for k, v in hist(lines).items() if k.startswith('v_')}
    want = collections.Counter()
    for ops in ledger.values():
        want.update(ops)
    if dict(want) != valu:
        diff = {k: (valu.get(k, 0), want.get(k, 0)) for k in set(valu) | set(want) if valu.get(k, 0) != want.get(k, 0)}
        sys.exit('%s: the ledger no longer matches the kernel (opcode: listing, ledger): %r' % (label, diff))
    return sum(valu.values()), sum(COST[klass(k)] * v for k, v in valu.items())


def main():
    text = listing()
    for title, K in KERNELS.items():
        row, merge, batch = blocks(text, K['name'])
        if '--dump' in sys.argv:
            for nm, ls in (('row', row), ('merge', merge), ('batch', batch)):
                print(nm, sorted(((k, c) for k, c in hist(ls).items() if k.startswith('v_')), key=lambda x: -x[1]))
            continue
        row_ledger = collections.OrderedDict(list(QUAD.items()) + list(K['row_extra'].items()))
        n_row, c_row = check(title + ' / row', row, row_ledger)
        n_mrg, c_mrg = check(title + ' / merge', merge, K['merge']) if K['merge'] else (0, 0.0)
        n_bat, c_bat = check(title + ' / batch', batch, K['batch'])
        epi = 4.0 / K['rows_per_iter'] if False else 4.0                 # elements per lane per row iteration
        el_batch = K['batch_rows'] * K['per_lane_row']                     # elements per lane per batch
        merges_per_batch = K['batch_rows'] // 2 if K['merge'] else 0
        per_elem = n_row / epi + n_mrg * merges_per_batch / el_batch + n_bat / el_batch
        cyc_elem = c_row / epi + c_mrg * merges_per_batch / el_batch + c_bat / el_batch
        print('k_core_large<SAMPLE, TD, MIXED>, %s' % title)
        print('Static count from the listing of the shipped sources; the ledger reproduces the opcode histogram of each block exactly (checked).')
        print('%-176s %5s %9s %7s' % ('piece', 'VALU', 'per elem', 'cycles'))
        for sect, ledger, div in (('row iteration (4 elements per lane)', row_ledger, epi),
                                  ('pair merge (every second row)', K['merge'], el_batch / max(merges_per_batch, 1)),
                                  ('per batch of %d rows (%d elements per lane)' % (K['batch_rows'], el_batch), K['batch'], el_batch)):
            if not ledger:
                continue
            print('  -- %s' % sect)
            for p, ops in ledger.items():
                n = sum(ops.values())
                c = sum(COST[klass(k)] * v for k, v in ops.items())
                print('%-176s %5d %9.2f %7.0f' % ('  ' + p, n, n / div, c))
        print('%-176s %5s %9.2f %7.1f  (priced cycles per element)' % ('TOTAL of the three blocks, per matrix element and lane', '', per_elem, cyc_elem))
        hr = hist(row)
        dyn = {256: '71.8 (profiles/r04_cycle_table_d256.txt)', 128: '74.5 (profiles/r04_cycle_table_d128.txt)'}[K['d']]
        print('measured (SQ counters, whole kernel incl. per-step staging, per-row epilogue after the row loop, the ~2 %% of pairs on the exact path): %s VALU per element' % dyn)
        print('besides, per row iteration: %d scalar / branch instructions, %d LDS instructions (the variates of the row into the lane-private stash), %d h-table loads'
              % (sum(v for k, v in hr.items() if k.startswith('s_')), sum(v for k, v in hr.items() if k.startswith('ds_')),
                 sum(v for k, v in hr.items() if k.startswith('global_'))))
        print()


if __name__ == '__main__':
    main()
