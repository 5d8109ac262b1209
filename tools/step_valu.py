"""Per-kernel VALU picture of the training step from tools/step_valu.sh: which kernels spend their wave cycles issuing VALU instructions (as opposed to waiting for memory)?

    python tools/step_valu.py TAG  ->  table on stdout: kernel, dispatches, VALU instructions per dispatch (M), share of wave cycles with a VALU instruction in flight"""
import collections, csv, glob, os, sys
tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
f = glob.glob(os.path.join(root, f'{tag}_svalu', '**', '*counter_collection.csv'), recursive=True)
rows = list(csv.DictReader(open(f[0])))
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name'].split('(')[0]
    per[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
        cnt[k] += 1
out = []
for k, c in per.items():
    wc = c.get('SQ_WAVE_CYCLES', 0.0)
    if wc <= 0:
        continue
    out.append((c.get('SQ_ACTIVE_INST_VALU', 0.0), k, cnt[k], c.get('SQ_INSTS_VALU', 0.0), c.get('SQ_ACTIVE_INST_VALU', 0.0) / wc, c.get('SQ_ACTIVE_INST_ANY', 0.0) / wc, wc))
out.sort(reverse=True)
print('# kernels of 3 steps (1 warm-up + 2), single stream; sorted by cycles with a VALU instruction executing (SQ_ACTIVE_INST_VALU, summed over SIMDs)')
print('| kernel | dispatches | VALU instr per dispatch (M, per wave) | VALU-active / wave cycles | any-instruction-active / wave cycles |\n|---|---|---|---|---|')
for a, k, n, iv, fv, fa, wc in out[:60]:
    print(f'| `{k[:90]}` | {n} | {iv / max(n, 1) / 1e6:.2f} | {fv:.3f} | {fa:.3f} |')
