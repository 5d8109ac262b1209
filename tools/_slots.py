import sys, os, torch, collections
sys.path.insert(0, '/root/repo')
import bench
sys.argv = ['bench.py', '--bs=2', '--height=64', '--width=96']
a = bench.parse()
k, ds, args = bench.build_trainer(a, 1)
k.model.train()
img, lab, _, _ = ds.parse(ds.make_batch(a.bs, 2023))
for _ in range(2):
    k.train_step(img, lab)
# one more step by hand to look at the grads before optimizer.step
from tcct_amd import ops
opt = k.optimG
opt.zero_grad(set_to_none=True)
ops.begin_step(k.device)
loss, _ = k.calc_loss(k.cuda(img), k.cuda(lab), want_log=False)
loss.backward()
ops.end_step()
named = dict(k.model.named_parameters())
cnt = collections.Counter()
ex = {}
for n, p in named.items():
    slot = getattr(p, '_grad_slot', None)
    if p.grad is None and slot is not None: cnt['in-place (grad None)'] += 1
    elif p.grad is None: cnt['no grad'] += 1
    elif slot is not None and p.grad.data_ptr() == slot.data_ptr(): cnt['alias'] += 1
    else:
        cnt['copied'] += 1; ex.setdefault(n.split('.')[-2] + '.' + n.split('.')[-1] if '.' in n else n, []).append((n, tuple(p.shape)))
print(cnt)
for kx, v in ex.items(): print(kx, len(v), v[:2])
