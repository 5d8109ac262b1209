"""Per-kernel totals of the LAST training step in a rocprofv3 --kernel-trace CSV (tools/trace_step.sh):  python tools/trace_top.py CSV [pattern ...]
With patterns: every launch of the matching kernels (duration us, blocks, grid y)."""
import collections
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_clip_adamw' in r['Kernel_Name']]
# (bench.py times clip + AdamW alone at the end: 20 back-to-back k_clip_adamw launches -- the training step is the LONGEST segment between two of them)
segs = [rows[a + 1: b + 1] for a, b in zip(idx[:-1], idx[1:])]
step = max(segs, key=len) if segs else rows
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
short = lambda r: r['Kernel_Name'].split('(')[0].replace('void ', '').replace('__hip_bfloat16', 'bf16')
print(f'{len(step)} launches, {sum(dur(r) for r in step) / 1e3:.2f} ms of kernel time in the last step')
if len(sys.argv) > 2:
    for pat in sys.argv[2:]:
        print(pat, [(round(dur(r)), int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y'])) for r in step if pat in short(r)])
else:
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in step:
        agg[short(r)[:64]][0] += 1
        agg[short(r)[:64]][1] += dur(r)
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f'{n:66s} {c:4d} {t / 1e3:6.2f} ms  avg {t / c:6.1f} us')
