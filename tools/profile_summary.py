"""Turn the files written by tools/collect_profiles.sh (gpurun_out/TAG_*) into the committed profiles/TAG_* summaries."""
import csv, collections, json, re, shutil, sys, os
tag = sys.argv[1]
G, P = 'gpurun_out', 'profiles'
step = list(csv.DictReader(open(f'{G}/{tag}_step/step_kernel_stats.csv')))
roof = list(csv.DictReader(open(f'{G}/{tag}_roof/roof_kernel_stats.csv')))
shutil.copy(f'{G}/{tag}_step/step_kernel_stats.csv', f'{P}/{tag}_kernel_stats.csv')
shutil.copy(f'{G}/{tag}_roof/roof_kernel_stats.csv', f'{P}/{tag}_roofline_only_kernel_stats.csv')
for n in ('bench', 'bench_reg', 'bench_fullloss', 'bench_fp32'):
    if os.path.exists(f'{G}/{tag}_{n}.json'):
        shutil.copy(f'{G}/{tag}_{n}.json', f'{P}/{tag}_{n}.json')
shutil.copy(f'{G}/{tag}_infer.txt', f'{P}/{tag}_infer.txt')
nsteps = 3
tot = sum(float(r['TotalDurationNs']) for r in step)
fams = [('first layers: conv + train-mode BatchNorm, conv output recomputed (k_c3_bn*)', r'k_c3_bn'), ('BatchNorm (k_bn*)', r'k_bn'), ('pointwise fused backward (k_pw_bwd)', r'k_pw_bwd'), ('pointwise fwd/dgrad (k_pw_fwd, k_pw_fwd2)', r'k_pw_fwd'),
        ('conv32 fwd/dgrad (k_conv32_mfma, k_conv32_fwd*_stream)', r'k_conv32_mfma|k_conv32_fwd'), ('conv32 fused backward (k_conv32_bwd33)', r'k_conv32_bwd33'),
        ('conv32 wgrad (k_conv32_wgrad)', r'k_conv32_wgrad'), ('pointwise wgrad (k_pw_wgrad*)', r'k_pw_wgrad'), ('depthwise (k_dw*)', r'k_dw'),
        ('bilinear', r'k_bilinear'), ('LayerNorm', r'k_ln_'), ('softmax-Dice', r'k_dice'), ('elementwise (k_map*, residual, concat)', r'k_map|k_residual|k_concat|k_split'),
        ('torch (autograd grad accumulation adds etc.)', r'at::native'), ('memset/copy (rocclr)', r'rocclr')]
acc = collections.OrderedDict((n, 0.0) for n, _ in fams)
other = 0.0
for r in step:
    t = float(r['TotalDurationNs'])
    for n, pat in fams:
        if re.search(pat, r['Name']):
            acc[n] += t
            break
    else:
        other += t
acc['other (pool, im2col, pack, optimizer, ...)'] = other
b = json.load(open(f'{G}/{tag}_bench.json'))
bf = json.load(open(f'{G}/{tag}_bench_fullloss.json'))
b32 = json.load(open(f'{G}/{tag}_bench_fp32.json'))
br = json.load(open(f'{G}/{tag}_bench_reg.json')) if os.path.exists(f'{G}/{tag}_bench_reg.json') else None
pm = {}
for kind in ('fetch', 'write'):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f'{G}/{tag}_{kind}/{kind}_counter_collection.csv')):
        d[r['Kernel_Name']].append(float(r['Counter_Value']))
    pm[kind] = {k: sum(v) / len(v) for k, v in d.items()}
L = []
L.append(f'# {tag}: MI355X, bf16, bs=8 1x800x1100\n')
L.append(f'Un-profiled bench lines: `profiles/{tag}_bench.json` ({b["value"]} B-scans/s, {b["ms_per_step"]} ms/step, `--los=di`); '
         + (f'`--los=di+reg` (BASELINE configs[2]): {br["value"]} B-scans/s, {br["ms_per_step"]} ms/step (`profiles/{tag}_bench_reg.json`); ' if br else '') +
         f'`--los=di+reg+fpl` (configs[3]): {bf["value"]} B-scans/s, {bf["ms_per_step"]} ms/step; fp32 parity mode: {b32["value"]} B-scans/s; '
         f'inference (`tools/infer_bench.py`): `profiles/{tag}_infer.txt`.\n')
# ---- the whole step, STEADY STATE (round 5): tools/steady_trace.sh -- the last three of six steps, cut at the optimizer kernel, so the model upload copies and the first
# steps' per-convolution weight packs are not in the table (the --stats CSV of the 3-step run above still is profiles/TAG_kernel_stats.csv)
st = f'{G}/{tag}_steady_summary.md'
if os.path.exists(st):
    shutil.copy(st, f'{P}/{tag}_steady_summary.md')
    L.append(f'## whole step, steady state — `bash tools/steady_trace.sh {tag}` (`TCCT_STREAMS=0`, one stream so that kernel durations add up; the bench lines above run the CNN / ViT encoders and the weight gradients on side streams)\n')
    body = open(st).read().splitlines()
    L.extend(body[2:])
else:
    L.append(f'(no steady-state trace collected; start-up-inclusive totals: {tot / 1e6 / nsteps:.2f} ms/step over {nsteps} steps)')
L.append(f'\n## roofline kernels alone — `rocprofv3 --kernel-trace --stats -- python3 bench.py --roofline-only` and, in separate passes, `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`\n')
L.append('HBM bytes = 2 x FETCH_SIZE (gfx950 wide-read correction, MI355X_MICROARCH.md) + WRITE_SIZE, counters in KiB, per launch.\n')
L.append('| kernel | calls | avg us (rocprof) | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM MB / launch | algorithmic MB | ratio |\n|---|---|---|---|---|---|---|---|')
alg = {'k_conv32_chain33': 1356.5952, 'k_conv32_wgradk_stream': 904.3968, 'k_conv32_mfma': 904.3968, 'k_conv32_fwd33_stream': 904.3968, 'k_conv32_wgrad33_stream': 904.3968, 'k_conv32_wgrad33_roll': 904.3968, 'k_conv32_wgrad': 904.3968, 'k_bn_bwd_reduce': 904.3968, 'k_pw_fwd': 452.1984, 'k_pw_fwd2': 452.1984, 'k_pw_bwd': 678.2976}
for r in roof:
    for key, a in alg.items():
        if key + '<' in r['Name'] or key + '(' in r['Name']:
            f = pm['fetch'].get(r['Name']); w = pm['write'].get(r['Name'])
            hbm = (2 * f + w) * 1024 / 1e6
            L.append(f'| `{r["Name"].split("(")[0]}` | {r["Calls"]} | {float(r["AverageNs"]) / 1e3:.1f} | {f:.1f} | {w:.1f} | {hbm:.1f} | {a:.1f} | {hbm / a:.2f} |')
def pmc_table(kind):
    """kernel -> counter -> mean per launch, from a rocprofv3 --pmc pass (None if the pass is missing)"""
    path = f'{G}/{tag}_{kind}/{kind}_counter_collection.csv'
    if not os.path.exists(path):
        return None
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        d[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in d.items()}
mf, sq = pmc_table('mfma'), pmc_table('sq')
if mf:
    L.append('\n### matrix-pipe occupancy of the same loops — `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` and `--pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY`\n')
    L.append('MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs): the gfx94x `MfmaUtil` formula (ROCm 7.2 ships no gfx950 derived '
             'metrics) with GRBM_GUI_ACTIVE divided by 8 because rocprofv3 reports it summed over the 8 XCDs (value / 8 = kernel duration x ~1.6 GHz). Cross-check for the 3x3 '
             'convolution: 130 GFLOP per launch = 3.97 M `v_mfma_f32_32x32x16_bf16` x 32 cycles = 1.27e8 busy cycles (the counter), and 538 TFLOP/s is 21.5 % of the 2.5 PFLOP/s dense bf16 peak '
             '(which assumes 2.4 GHz). These kernels are HBM-bound (SURVEY 8(d)): the matrix pipes idle most of the time by construction; the wave-cycle split shows where waves wait.\n')
    L.append('| kernel | MFMA busy cycles | GUI active cycles | MFMA busy % | wave cycles: waiting (s_waitcnt/barrier) % | issue-stalled % | issuing % |\n|---|---|---|---|---|---|---|')
    for r in roof:
        if any(key + '<' in r['Name'] or key + '(' in r['Name'] for key in alg):
            m = mf.get(r['Name'])
            if not m:
                continue
            busy, act = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), m.get('GRBM_GUI_ACTIVE', 0.0)
            q = (sq or {}).get(r['Name'], {})
            wc = q.get('SQ_WAVE_CYCLES', 0.0)
            pct = lambda v: f'{100 * v / wc:.1f}' if wc else '-'
            L.append(f'| `{r["Name"].split("(")[0]}` | {busy:.3g} | {act:.3g} | {100 * busy / (act / 8 * 1024) if act else 0:.1f} | {pct(q.get("SQ_WAIT_ANY", 0))} | {pct(q.get("SQ_WAIT_INST_ANY", 0))} | {pct(q.get("SQ_ACTIVE_INST_ANY", 0))} |')
# ---- loss-side kernels (BASELINE configs[2..3]): single-stream trace of the --los=di+reg+fpl step + the roofline loops of the same kernels
if os.path.exists(f'{G}/{tag}_stepfl/stepfl_kernel_stats.csv'):
    shutil.copy(f'{G}/{tag}_stepfl/stepfl_kernel_stats.csv', f'{P}/{tag}_fullloss_kernel_stats.csv')
    sfl = list(csv.DictReader(open(f'{G}/{tag}_stepfl/stepfl_kernel_stats.csv')))
    base = {r['Name']: float(r['TotalDurationNs']) for r in step}
    tfl = sum(float(r['TotalDurationNs']) for r in sfl)
    L.append(f'\n## `--los=di+reg+fpl` (BASELINE configs[3]) — single-stream kernel trace, {tfl / 1e6 / nsteps:.2f} ms/step (+{(tfl - tot) / 1e6 / nsteps:.2f} over `--los=di`); the kernels that only this configuration runs\n')
    L.append('| kernel | calls / step | ms / step |\n|---|---|---|')
    extra = sorted(((float(r['TotalDurationNs']) - base.get(r['Name'], 0.0), r) for r in sfl), key=lambda t: -t[0])
    for d, r in extra[:16]:
        if d > 0.02e6 * nsteps:
            L.append(f'| `{r["Name"].split("(")[0][:80]}` | {int(r["Calls"]) // nsteps} | {d / 1e6 / nsteps:.3f} |')
    L.append(f'\nrocPRIM symbols in this trace: {sum(1 for r in sfl if "rocprim" in r["Name"])} (round 4: the library sort is gone from `libtcct_hip.so` altogether).')
if os.path.exists(f'{G}/{tag}_lroof/lroof_kernel_stats.csv'):
    lroof = list(csv.DictReader(open(f'{G}/{tag}_lroof/lroof_kernel_stats.csv')))
    lp = {}
    for kind in ('lfetch', 'lwrite'):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f'{G}/{tag}_{kind}/{kind}_counter_collection.csv')):
            d[r['Kernel_Name']].append(float(r['Counter_Value']))
        lp[kind] = {k: sum(v) / len(v) for k, v in d.items()}
    L.append('\n### loss-side kernels alone — `bench.py --roofline-only --los=di+reg+fpl` under `--kernel-trace --stats`, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`\n')
    L.append('| kernel | calls | avg us | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM MB / launch (2 x fetch + write) |\n|---|---|---|---|---|---|')
    for r in lroof:
        if re.search(r'k_gumbel|k_normadd|k_invnorm|k_fs_', r['Name']):
            f, w = lp['lfetch'].get(r['Name'], 0.0), lp['lwrite'].get(r['Name'], 0.0)
            L.append(f'| `{r["Name"].split("(")[0][:60]}` | {r["Calls"]} | {float(r["AverageNs"]) / 1e3:.1f} | {f:.1f} | {w:.1f} | {(2 * f + w) * 1024 / 1e6:.1f} |')
    L.append(f'\nbench.py `roofline.others` of the full-loss line: {[(o["kernel"][:40], o["ms_per_launch"], o["frac"]) for o in bf.get("roofline", {}).get("others", [])]}')
L.append(f'\nbench.py\'s own HIP-event timing of the same loops (un-profiled run): `roofline.ms_per_launch` = {b["roofline"]["ms_per_launch"]} ms '
         f'({b["roofline"]["achieved"]} GB/s algorithmic, frac {b["roofline"]["frac"]}), second = {b["roofline"]["second"]["ms_per_launch"]} ms, '
         f'others = {[(o["kernel"], o["ms_per_launch"]) for o in b["roofline"].get("others", [])]}.')
# ---- per symbol: steady-state time joined with the PMC traffic of the same single-stream step -> TB/s on REAL bytes (round 5)
sc, tc = f'{G}/{tag}_steady_symbols.csv', f'{G}/{tag}_traffic_symbols.csv'
if os.path.exists(sc) and os.path.exists(tc):
    tm = {r['symbol']: (float(r['calls_per_step']), float(r['ms_per_step'])) for r in csv.DictReader(open(sc))}
    tr = {r['symbol']: float(r['mb_per_step']) for r in csv.DictReader(open(tc))}
    L.append('\n## every large symbol of the step: time (steady-state trace) x bytes (PMC, 2 x FETCH_SIZE + WRITE_SIZE) -> TB/s on the bytes it REALLY moves\n')
    L.append('| kernel | calls/step | ms/step | HBM MB/step | TB/s |\n|---|---|---|---|---|')
    tot_ms = tot_mb = 0.0
    for k, (c, ms) in sorted(tm.items(), key=lambda x: -x[1][1])[:45]:
        mb = tr.get(k)
        if mb is None:
            continue
        tot_ms += ms; tot_mb += mb
        L.append(f'| `{k[:90]}` | {c:.0f} | {ms:.3f} | {mb:.0f} | {mb / ms / 1e3:.2f} |')
    L.append(f'| (these rows) | | {tot_ms:.2f} | {tot_mb:.0f} | {tot_mb / tot_ms / 1e3:.2f} |')
# ---- HBM traffic of the WHOLE step (tools/step_traffic.sh: PMC FETCH_SIZE / WRITE_SIZE over every kernel of the single-stream step)
if os.path.exists(f'{G}/{tag}_traffic.txt'):
    shutil.copy(f'{G}/{tag}_traffic.txt', f'{P}/{tag}_step_traffic.txt')
    tl = open(f'{G}/{tag}_traffic.txt').read().splitlines()
    head = [x for x in tl if x.startswith('total HBM traffic')]
    if head:
        gb = float(head[0].split(':')[1].split('GB')[0])
        L.append(f'\n## HBM traffic of the whole step (`tools/step_traffic.sh`, PMC over every kernel, 2 x FETCH_SIZE + WRITE_SIZE)\n')
        L.append(f'{head[0]}.  At the un-profiled {b["ms_per_step"]} ms/step that is **{gb / b["ms_per_step"]:.2f} TB/s on average** '
                 f'(copy ceiling of this box: {b["roofline"].get("copy_ceiling", {}).get("GBs", "?")} GB/s; SURVEY 8(d) model bytes: '
                 f'{b["roofline"].get("step_model_GBs", "?")} GB/s): the step as a whole moves its REAL bytes close to the achievable rate, '
                 f'so what is left to gain is passes removed, not kernels tuned.  Per kernel: `profiles/{tag}_step_traffic.txt`.')
# ---- round 6: the step by level and branch (tools/attrib_trace.sh), the level timeline of the multi-stream step, the allocator plateau
for src, dst in ((f'{tag}_attrib_summary.md', f'{tag}_attrib_summary.md'), (f'{tag}_fl_attrib_summary.md', f'{tag}_attrib_fullloss_summary.md'),
                 (f'{tag}_inf_attrib_summary.md', f'{tag}_attrib_infer_summary.md'), (f'{tag}_attrib_calls.csv', f'{tag}_attrib_calls.csv'),
                 (f'{tag}_level_timeline.txt', f'{tag}_level_timeline.txt'), (f'{tag}_level_timeline_fullloss.txt', f'{tag}_level_timeline_fullloss.txt'),
                 (f'{tag}_memgrow.txt', f'{tag}_memgrow.txt')):
    if os.path.exists(f'{G}/{src}'):
        shutil.copy(f'{G}/{src}', f'{P}/{dst}')
if os.path.exists(f'{P}/{tag}_attrib_infer_summary.md'):
    t_ = open(f'{P}/{tag}_attrib_infer_summary.md').read().replace('the training step by level and branch', 'KiteSeg.predict (eval forward + argmax mask) by level and branch')
    open(f'{P}/{tag}_attrib_infer_summary.md', 'w').write(t_)
if os.path.exists(f'{P}/{tag}_attrib_summary.md'):
    L.append(f'\n## the step by level and branch (`tools/attrib_trace.sh`): `profiles/{tag}_attrib_summary.md` (`--los=di`), `profiles/{tag}_attrib_fullloss_summary.md`, '
             f'`profiles/{tag}_attrib_infer_summary.md` (predict); per C-ABI call: `profiles/{tag}_attrib_calls.csv`; level timeline of the multi-stream step (HIP events, no profiler): '
             f'`profiles/{tag}_level_timeline.txt`\n')
    body = open(f'{P}/{tag}_attrib_summary.md').read().split('## levels 3-4 of the CNN encoder')[0].splitlines()
    L.extend(body[3:])
open(f'{P}/{tag}_summary.md', 'w').write('\n'.join(L) + '\n')
print('\n'.join(L)[:3000])
# HBM traffic of the roofline kernels + the hash of the sources they were collected on -> profiles/TAG_pmc.json (bench.py's roofline.traffic)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_json  # noqa: E402
sys.argv = [sys.argv[0], tag]
pmc_json.main()
