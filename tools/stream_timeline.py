"""Multi-stream timeline of ONE steady-state step from a per-launch kernel trace taken WITH the side streams on (bash tools/stream_timeline.sh TAG):
per HIP queue busy time, how much of the step has 1 / 2 / 3 kernels in flight, and which kernels run ALONE (nothing else in flight) -- the exposed ones."""
import collections, csv, glob, os, sys
tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
f = glob.glob(os.path.join(root, f'{tag}_mstream', '**', '*kernel_trace.csv'), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
opt = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_clip_adamw')]
opt = [i for j, i in enumerate(opt) if i - (opt[j - 1] if j else -1) > 100]
rows = rows[opt[-2] + 1: opt[-1] + 1]
t0 = min(int(r['Start_Timestamp']) for r in rows); t1 = max(int(r['End_Timestamp']) for r in rows)
ev = []
for i, r in enumerate(rows):
    ev.append((int(r['Start_Timestamp']), 1, i)); ev.append((int(r['End_Timestamp']), -1, i))
ev.sort()
conc = collections.Counter(); alone = collections.defaultdict(float); live = set(); last = t0
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        conc[len(live)] += dt
        if len(live) == 1:
            alone[rows[next(iter(live))]['Kernel_Name'].split('(')[0]] += dt
    last = t
    if d > 0: live.add(i)
    else: live.discard(i)
q = collections.defaultdict(float)
for r in rows: q[r['Queue_Id']] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
L = [f'# {tag}: one steady-state step with the side streams ON ({len(rows)} launches, wall {(t1 - t0) / 1e6:.2f} ms under the profiler)\n',
     '| kernels in flight | ms | % of the step |\n|---|---|---|']
for k in sorted(conc): L.append(f'| {k} | {conc[k] / 1e6:.2f} | {100 * conc[k] / (t1 - t0):.1f} |')
L.append('\n| HIP queue | kernel time (ms) |\n|---|---|')
for k, v in sorted(q.items(), key=lambda x: -x[1]): L.append(f'| {k} | {v / 1e6:.2f} |')
L.append('\n| kernel running ALONE (nothing else in flight) | ms |\n|---|---|')
for k, v in sorted(alone.items(), key=lambda x: -x[1])[:40]: L.append(f'| `{k[:100]}` | {v / 1e6:.3f} |')
open(os.path.join(root, f'{tag}_mstream_summary.md'), 'w').write('\n'.join(L) + '\n')
print('\n'.join(L))
