"""Which kernels surround the __amd_rocclr_copyBuffer dispatches of a step?  (reads a rocprofv3 --kernel-trace CSV)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0][:60] for r in rows]
pairs = collections.Counter()
for i, n in enumerate(names):
    if 'copyBuffer' in n:
        prev = next((names[j] for j in range(i - 1, -1, -1) if 'copyBuffer' not in names[j]), '-')
        nxt = next((names[j] for j in range(i + 1, len(names)) if 'copyBuffer' not in names[j]), '-')
        pairs[(prev, nxt)] += 1
for (p, n), c in pairs.most_common(25):
    print(c, '|', p, '->', n)
print('total copies', sum(pairs.values()), 'of', len(names), 'dispatches')
