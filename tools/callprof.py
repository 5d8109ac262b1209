"""Per-call-site GPU time of one training step: every libtcct_hip call is bracketed by synchronize() and accumulated by
(entry point, integer arguments).  Shows which shapes of which kernels the step time sits in.  usage: python tools/callprof.py [bench flags]"""
import sys, os, time, collections, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tcct_amd import _lib
sys.argv = ['bench.py'] + sys.argv[1:]
a = bench.parse()
k, ds, args = bench.build_trainer(a, 1)
k.model.train()
img, lab, _, _ = ds.parse(ds.make_batch(a.bs, 2023))
for _ in range(3):
    k.train_step(img, lab)
torch.cuda.synchronize()

acc = collections.defaultdict(lambda: [0, 0.0])
L = _lib.lib
orig = {n: f for n, f in L.__dict__.items() if callable(f) and getattr(f, '__name__', '').startswith('tcct_')}
gap = [0.0, None]


def wrap(name, f):
    sig = L.protos['tcct_' + name][1]

    def timed(*args):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if gap[1] is not None:
            gap[0] += t0 - gap[1]
        r = f(*args)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        gap[1] = t1
        key = (name, tuple(int(v) for v, (ct, nm) in zip(args, sig) if ct is not ctypes.c_void_p and nm not in ('dtype',) and isinstance(v, int)))
        e = acc[key]
        e[0] += 1
        e[1] += t1 - t0
        return r
    return timed


for n, f in orig.items():
    L.__dict__[n] = wrap(n, f)
t0 = time.perf_counter()
k.train_step(img, lab)
torch.cuda.synchronize()
t1 = time.perf_counter()
tot = sum(e[1] for e in acc.values())
print(f'step {1e3 * (t1 - t0):.1f} ms with sync; in tcct calls {1e3 * tot:.1f} ms; between calls (torch ops, python) {1e3 * gap[0]:.1f} ms')
fam = collections.defaultdict(float)
for (n, ar), e in acc.items():
    fam[n] += e[1]
print('--- by entry point')
for n, t in sorted(fam.items(), key=lambda x: -x[1])[:40]:
    print(f'{1e3 * t:8.2f} ms  {n}')
print('--- by call site shape (top 70)')
for (n, ar), e in sorted(acc.items(), key=lambda x: -x[1][1])[:70]:
    print(f'{1e3 * e[1]:8.3f} ms {e[0]:3d}x  {n} {ar}')
