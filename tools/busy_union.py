"""GPU busy time (union of kernel intervals over all streams) vs wall time between the first and last kernel of the traced steps.
usage: python tools/busy_union.py <kernel_trace.csv> [last_ms]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
t_first, t_last = iv[0][0], iv[-1][1]
cut = t_last - last_ms * 1e6                       # steady state only: the last `last_ms` milliseconds of the trace
iv = [(a, b) for a, b in iv if a >= cut]
busy = 0
cs, ce = iv[0]
gaps = []
for a, b in iv[1:]:
    if a > ce:
        busy += ce - cs
        gaps.append(a - ce)
        cs, ce = a, b
    else:
        ce = max(ce, b)
busy += ce - cs
wall = iv[-1][1] - iv[0][0]
print(f'kernels {len(iv)}  wall {wall / 1e6:.2f} ms  busy(union) {busy / 1e6:.2f} ms  idle {100 * (1 - busy / wall):.1f}%  sum of durations {sum(b - a for a, b in iv) / 1e6:.2f} ms')
gaps.sort(reverse=True)
print('largest gaps (us):', [round(g / 1e3, 1) for g in gaps[:12]], ' gaps > 20us:', sum(1 for g in gaps if g > 20000), ' total gap ms:', round(sum(gaps) / 1e6, 2))
