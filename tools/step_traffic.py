"""Sum the PMC traffic of tools/step_traffic.sh per kernel symbol: HBM MB per step = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 wide-read
correction of MI355X_MICROARCH.md), 3 traced steps (1 warm-up + 2)."""
import collections, csv, glob, os, sys
tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
def load(kind):
    f = glob.glob(os.path.join(root, f'{tag}_{kind}', '**', '*counter_collection.csv'), recursive=True)[0]
    d = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name']] += float(r['Counter_Value']); n[r['Kernel_Name']] += 1
    return d, n
fe, n = load('sfetch'); wr, _ = load('swrite')
rows = []
for k in set(fe) | set(wr):
    mb = (2 * fe.get(k, 0) + wr.get(k, 0)) * 1024 / 1e6 / 3
    rows.append((mb, k, n[k] / 3, 2 * fe.get(k, 0) * 1024 / 1e6 / 3, wr.get(k, 0) * 1024 / 1e6 / 3))
rows.sort(reverse=True)
with open(os.path.join(root, f'{tag}_traffic_symbols.csv'), 'w') as fcsv:         # symbol (as steady_trace.py names it) -> MB / step
    fcsv.write('symbol,calls_per_step,mb_per_step\n')
    for mb, k, c, r_, w_ in rows:
        fcsv.write('"%s",%.3f,%.3f\n' % (k.split('(')[0].replace('"', "'"), c, mb))
tot = sum(r[0] for r in rows)
print(f'total HBM traffic per step: {tot / 1e3:.2f} GB (reads {sum(r[3] for r in rows) / 1e3:.2f}, writes {sum(r[4] for r in rows) / 1e3:.2f})')
for mb, k, c, r_, w_ in rows[:45]:
    print(f'{mb:9.1f} MB  {c:5.1f} calls  r {r_:8.1f} w {w_:8.1f}  {k[:100]}')
