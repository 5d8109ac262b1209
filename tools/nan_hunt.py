"""Train with the full loss and report the first non-finite quantity (loss part, gradient norm, parameter)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tcct_amd import ops
a = bench.parse()
k, ds, args = bench.build_trainer(a, 1)
k.model.train()
img, lab, _, _ = ds.parse(ds.make_batch(a.bs, 2023))
img, lab = img.contiguous(), lab.contiguous()
for it in range(a.steps):
    k.optimG.zero_grad(set_to_none=True)
    ops.begin_step(k.device)
    try:
        tot, log = k.calc_loss(img, lab, want_log=True)
        tot.backward()
    finally:
        ops.end_step()
    k.optimG.step()
    gn = float(k.optimG.last_total_norm)
    lr = k.optimG.param_groups[0]['lr']
    bad = [n for n, p in k.model.named_parameters() if not torch.isfinite(p).all()]
    if it % 10 == 0 or not (gn == gn) or bad or 'nan' in log:
        print(it, log, 'gnorm %.4g lr %.3g' % (gn, lr), 'bad params:', bad[:4], flush=True)
    if bad or not (gn == gn):
        g = k.optimG._flat['g']
        print('nonfinite grads:', int((~torch.isfinite(g)).sum()))
        off = 0
        for p_, (n, _) in zip(k.optimG._flat['plist'], [(None, None)] * 10**6):
            pass
        names = {id(p): n for n, p in k.model.named_parameters()}
        off = 0
        for p_ in k.optimG._flat['plist']:
            kk = p_.numel()
            if not torch.isfinite(g[off:off + kk]).all():
                print('  grad of', names.get(id(p_)), 'non-finite')
            off += kk
        break
