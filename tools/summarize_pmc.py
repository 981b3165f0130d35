"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into a per-launch HBM traffic summary.

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB; on gfx950
FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced streaming read, so the read side
is doubled for kernels whose loads are 16 B/lane streams (k_step_small / k_step_large<4,*>); WRITE_SIZE is
used as reported (uncalibrated in the guide; here it matches the algorithmic byte count within 1 %).
usage: summarize_pmc.py <fetch_csv> <write_csv> <round_tag> <d,T,B>
"""
import csv, json, sys

fetch_csv, write_csv, tag, shape = sys.argv[1:5]
d, T, B = (int(x) for x in shape.split(','))
WIDE = ('k_step_small', 'k_step_wave', 'k_step_large', 'k_step_rows')


def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        acc.setdefault(name, []).append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


f, nf = per_kernel(fetch_csv, 'FETCH_SIZE')
w, nw = per_kernel(write_csv, 'WRITE_SIZE')
N = B * T
out = {'round': tag, 'shape': {'d': d, 'T': T, 'B': B, 'transitions_per_launch': N},
       'algorithmic_bytes_per_step': 4 * (d * d + 2 * d + 1), 'kernels': {}}
for k in sorted(set(f) | set(w)):
    if not k.startswith('k_'):
        continue
    wide = any(k.startswith(x) for x in WIDE)
    fb = f.get(k, 0.0) * 1024 * (2 if wide else 1)
    wb = w.get(k, 0.0) * 1024
    out['kernels'][k] = {'launches_sampled': nf.get(k, 0), 'FETCH_SIZE_KiB_raw': f.get(k), 'WRITE_SIZE_KiB_raw': w.get(k),
                         'fetch_correction': 'x2 (gfx950 wide coalesced stream)' if wide else 'x1',
                         'hbm_read_bytes': fb, 'hbm_write_bytes': wb, 'hbm_bytes_per_launch': fb + wb}
k0 = [k for k in out['kernels'] if k.startswith('k_step_')]
if k0:
    e = out['kernels'][k0[0]]
    e['algorithmic_bytes_per_launch'] = N * out['algorithmic_bytes_per_step']
    e['traffic_over_algorithmic'] = e['hbm_bytes_per_launch'] / e['algorithmic_bytes_per_launch']
json.dump(out, sys.stdout, indent=1)
