"""Per-module GPU time of one training step (forward and backward), measured with synchronising hooks.
usage: python tools/modprof.py [bench flags] -- coarse table of where the step goes, by sub-module of RegNet(stc_tt)."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sys.argv = ['bench.py'] + sys.argv[1:]
a = bench.parse()
k, ds, args = bench.build_trainer(a, 1)
k.model.train()
img, lab, _, _ = ds.parse(ds.make_batch(a.bs, 2023))
for _ in range(3):
    k.train_step(img, lab)
torch.cuda.synchronize()

DEPTH = int(os.environ.get('MODPROF_DEPTH', '3'))
PREFIX = os.environ.get('MODPROF_PREFIX', '')
names = [n for n, m in k.model.named_modules() if n and n.startswith(PREFIX) and n.count('.') < DEPTH and sum(1 for _ in m.parameters()) > 0]
# keep only "leaf-most" selected names (a parent is dropped when any child is selected)
sel = [n for n in names if not any(o.startswith(n + '.') for o in names)]
mods = dict(k.model.named_modules())
fwd = collections.OrderedDict((n, 0.0) for n in sel)
bwd = collections.OrderedDict((n, 0.0) for n in sel)
marks = []          # (time, event) in execution order


def now():
    torch.cuda.synchronize()
    return time.perf_counter()


def first_tensor(o):
    if torch.is_tensor(o):
        return o
    if isinstance(o, (list, tuple)):
        for e in o:
            t = first_tensor(e)
            if t is not None:
                return t
    return None


for n in sel:
    m = mods[n]

    def pre(mod, inp, n=n):
        marks.append((now(), 'f0', n))
        t = first_tensor(inp)
        if t is not None and t.requires_grad:
            t.register_hook(lambda g, n=n: marks.append((now(), 'b1', n)))

    def post(mod, inp, out, n=n):
        marks.append((now(), 'f1', n))
        t = first_tensor(out)
        if t is not None and t.requires_grad:
            t.register_hook(lambda g, n=n: marks.append((now(), 'b0', n)))

    m.register_forward_pre_hook(pre)
    m.register_forward_hook(post)

t_start = now()
k.train_step(img, lab)
t_end = now()
# attribute every interval between consecutive marks to the module of the *later* mark when it closes a span (f1/b1),
# otherwise to "other" (glue between modules: losses, FTC adds, optimizer ...)
other_f = other_b = 0.0
prev = t_start
phase = 'f'
for t, kind, n in marks:
    dt = (t - prev) * 1e3
    if kind == 'f1':
        fwd[n] += dt
    elif kind == 'b1':
        bwd[n] += dt
    elif kind == 'f0':
        other_f += dt
    else:
        other_b += dt
    prev = t
tail = (t_end - prev) * 1e3
print(f'step (with sync hooks) {1e3 * (t_end - t_start):.1f} ms; glue fwd {other_f:.2f} ms, glue bwd(+loss) {other_b:.2f} ms, tail(optimizer etc) {tail:.2f} ms')
print(f'{"module":60s} {"fwd ms":>8s} {"bwd ms":>8s}')
for n in sel:
    if fwd[n] or bwd[n]:
        print(f'{n:60s} {fwd[n]:8.2f} {bwd[n]:8.2f}')
print(f'{"sum":60s} {sum(fwd.values()):8.2f} {sum(bwd.values()):8.2f}')
