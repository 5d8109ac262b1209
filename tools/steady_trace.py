"""Steady-state summary of tools/steady_trace.sh: the launches of the LAST THREE steps of a single-stream kernel trace (cut at the optimizer kernel
`k_clip_adamw`, one per step), so start-up work (model upload copies, per-convolution weight packs of the first steps) is not in the table.
Writes gpurun_out/TAG_steady_summary.md: kernel families, a duration histogram (how much of the step is launches shorter than 20 us), the largest
symbols, and the short launches by symbol with their typical grid (the level a launch belongs to shows in its grid size)."""
import collections, csv, glob, os, re, sys
tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
f = glob.glob(os.path.join(root, f'{tag}_steady', '**', '*kernel_trace.csv'), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
opt = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_clip_adamw')]
opt = [i for j, i in enumerate(opt) if i - (opt[j - 1] if j else -1) > 100]         # whole steps only (bench.py times the optimizer kernels alone afterwards)
NS = 3
assert len(opt) >= NS + 1, f'{len(opt)} optimizer launches in the trace'
rows = rows[opt[-NS - 1] + 1: opt[-1] + 1]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3      # us
fams = [('first layers: conv + BatchNorm, conv output recomputed (k_c3_bn*)', r'k_c3_bn'), ('BatchNorm (k_bn*)', r'k_bn'), ('pointwise fused backward (k_pw_bwd)', r'k_pw_bwd'),
        ('pointwise fwd/dgrad (k_pw_fwd*)', r'k_pw_fwd'), ('conv32 fwd/dgrad (k_conv32_mfma, k_conv32_fwd*_stream, chains)', r'k_conv32_mfma|k_conv32_fwd|k_conv32_chain'),
        ('conv32 wgrad (k_conv32_wgrad*)', r'k_conv32_wgrad'), ('pointwise wgrad (k_pw_wgrad*)', r'k_pw_wgrad'), ('depthwise (k_dw*)', r'k_dw'),
        ('bilinear', r'k_bilinear'), ('LayerNorm / token mixer', r'k_ln_'), ('softmax-Dice', r'k_dice|k_updice'), ('elementwise (k_map*, residual, concat)', r'k_map|k_residual|k_concat|k_split'),
        ('torch kernels', r'at::native'), ('memset/copy (rocclr)', r'rocclr'), ('loss side (gumbel, colsoftmax, fpl, normadd)', r'k_gumbel|k_col|k_fs_|k_fpl|k_normadd|k_invnorm|k_l2norm|k_label|k_mse')]
acc = collections.OrderedDict((n, [0.0, 0]) for n, _ in fams)
acc['other (pool, pack, optimizer, heads, ...)'] = [0.0, 0]
sym = collections.defaultdict(lambda: [0.0, 0, collections.Counter()])
buckets = [(0, 5), (5, 10), (10, 20), (20, 50), (50, 100), (100, 1e9)]
hist = [[0.0, 0] for _ in buckets]
for r in rows:
    d = dur(r)
    name = r['Kernel_Name']
    for n, pat in fams:
        if re.search(pat, name):
            acc[n][0] += d; acc[n][1] += 1
            break
    else:
        acc['other (pool, pack, optimizer, heads, ...)'][0] += d; acc['other (pool, pack, optimizer, heads, ...)'][1] += 1
    s = sym[name.split('(')[0]]
    s[0] += d; s[1] += 1
    if d < 20:
        s[2][int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1)] += 1
    for b, (lo, hi) in zip(hist, buckets):
        if lo <= d < hi:
            b[0] += d; b[1] += 1
tot = sum(dur(r) for r in rows)
span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3
L = [f'# {tag}: steady-state single-stream kernel trace, last {NS} of 6 steps (`tools/steady_trace.sh`)\n',
     f'{len(rows) / NS:.0f} launches per step, kernel time {tot / NS / 1e3:.2f} ms per step (span of the traced steps under the profiler: {span / NS / 1e3:.2f} ms per step).\n',
     '| family | ms/step | launches/step | % |\n|---|---|---|---|']
for n, (t, c) in sorted(acc.items(), key=lambda x: -x[1][0]):
    L.append(f'| {n} | {t / NS / 1e3:.3f} | {c / NS:.0f} | {100 * t / tot:.1f} |')
L.append('\n| launch duration | launches/step | ms/step |\n|---|---|---|')
for (lo, hi), (t, c) in zip(buckets, hist):
    L.append(f'| {lo}-{hi if hi < 1e9 else "inf"} us | {c / NS:.0f} | {t / NS / 1e3:.3f} |')
short = sum(b[0] for b in hist[:3]) / NS / 1e3
L.append(f'\nlaunches shorter than 20 us: {sum(b[1] for b in hist[:3]) / NS:.0f} per step, {short:.3f} ms per step single-stream.\n')
L.append('| kernel | calls/step | ms/step | avg us |\n|---|---|---|---|')
for k, (t, c, _) in sorted(sym.items(), key=lambda x: -x[1][0])[:40]:
    L.append(f'| `{k[:90]}` | {c / NS:.1f} | {t / NS / 1e3:.3f} | {t / c:.1f} |')
L.append('\n## launches shorter than 20 us by symbol (grid sizes of those launches)\n\n| kernel | short calls/step | grids |\n|---|---|---|')
for k, (t, c, g) in sorted(sym.items(), key=lambda x: -sum(x[1][2].values()))[:30]:
    if g:
        L.append(f'| `{k[:90]}` | {sum(g.values()) / NS:.1f} | {dict(sorted(g.items())[:6])} |')
with open(os.path.join(root, f'{tag}_steady_symbols.csv'), 'w') as fcsv:          # symbol -> calls / step, ms / step (joined with the PMC traffic by tools/profile_summary.py)
    fcsv.write('symbol,calls_per_step,ms_per_step\n')
    for k, (t, c, _) in sorted(sym.items(), key=lambda x: -x[1][0]):
        fcsv.write('"%s",%.3f,%.5f\n' % (k.replace('"', "'"), c / NS, t / NS / 1e3))
out = os.path.join(root, f'{tag}_steady_summary.md')
open(out, 'w').write('\n'.join(L) + '\n')
print('\n'.join(L[:28]))
