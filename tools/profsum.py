"""Summarise a rocprofv3 rocpd database (kernel-trace) by kernel family: python tools/profsum.py gpurun_out/prof_x/x_results.db STEPS"""
import sqlite3, re, collections, sys
c = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
rows = c.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f'total kernel time {tot / 1e6 / steps:.2f} ms/step over {steps} steps')
fam = collections.Counter()
for n, cnt, t, a in rows:
    k = re.sub(r'<.*', '', n); k = re.sub(r'\(.*', '', k).replace('void ', '')
    fam[k] += t
for k, t in fam.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 28):
    print(f'{t / 1e6 / steps:7.2f} ms  {k[:90]}')
if len(sys.argv) > 4:
    for n, cnt, t, a in rows[:int(sys.argv[4])]:
        print(f'{t / 1e6 / steps:7.2f} ms {cnt / steps:6.1f}x {a / 1e3:8.1f} us  {n[:120]}')
