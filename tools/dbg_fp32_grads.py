"""debug: per-tensor gradient deviation of the fp32 HIP path from the fp64 oracle on a golden fixture, with and without the fp32 MFMA kernels"""
import os, sys, json, argparse
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import tcct_oracle as O
from tcct_amd import ops
name = sys.argv[1] if len(sys.argv) > 1 else 'full_2x32x32'
GOLD = os.path.join(ROOT, 'tests', 'golden')
fx = dict(np.load(os.path.join(GOLD, name + '.npz')))
keys = [(k, tuple(s)) for k, s in json.load(open(os.path.join(GOLD, 'state_dict_keys.json')))]
names = [str(n) for n in fx['grad_names']]
udh, reg = bool(fx['flags'][0]), bool(fx['flags'][1])
lab = torch.tensor(fx['lab']).long()


def oracle(dt):
    sd = {kk: (v.to(dt) if v.is_floating_point() else v.clone()) for kk, v in O.formula_state_dict(keys).items()}
    for n in names:
        sd[n].requires_grad_(True)
    oh = torch.nn.functional.one_hot(lab, 5).permute(0, 3, 1, 2)
    dm = [torch.tensor(m).to(dt) for m in fx['dp_masks']] if 'dp_masks' in fx else None
    nz = tuple(torch.tensor(fx[f'noise{i}']).to(dt) for i in range(4)) if reg else None
    t, _, _, _ = O.total_loss(sd, torch.tensor(fx['img']).to(dt).repeat(1, 3, 1, 1), oh, udh=udh, reg=reg, dp_masks=dm, noise=nz)
    t.backward()
    return {n: sd[n].grad.double() for n in names}


def hip(mfma):
    from test_model_gpu import build, make_kite, run_losses
    ops.F32_MFMA = mfma
    model, _ = build(torch.float32)
    k = make_kite(model, '/tmp/dbg_fp32', udh, reg)
    out, parts, total = run_losses(k, fx, torch.tensor(fx['img']).cuda(), lab.cuda())
    k.optimG.zero_grad(set_to_none=True)
    total.backward()
    return {n: p.grad.double().cpu() for n, p in model.named_parameters() if p.grad is not None}


g64, g32 = oracle(torch.float64), oracle(torch.float32)
for mfma, pw in ((True, [False, True, True]), (True, [True, True, True])):
    ops.F32_PW[:] = pw
    print('F32_PW', pw)
    g = hip(mfma)
    rows = sorted(((g[n] - g64[n]).norm().item() / max(g64[n].norm().item(), 1e-30), (g32[n] - g64[n]).norm().item() / max(g64[n].norm().item(), 1e-30), n)
                  for n in names if g64[n].norm().item() > 1e-3 * max(v.norm().item() for v in g64.values()))
    e = np.array([r[0] for r in rows])
    print(f'F32_MFMA={mfma}: median {np.median(e):.2e} p90 {np.percentile(e, 90):.2e} max {e.max():.2e}; torch fp32 median {np.median([r[1] for r in rows]):.2e}')
    top = sorted(names, key=lambda n: -g64[n].norm().item())[:14]
    tn = lambda d: sum((d[n] ** 2).sum() for n in names).sqrt().item()      # noqa: E731
    print('   total norm hip %.2f  fp64 %.2f  torch32 %.2f' % (tn(g), tn(g64), tn(g32)))
    for n in top:
        cos = (g[n] * g64[n]).sum().item() / (g[n].norm().item() * g64[n].norm().item())
        print('   |g64| %.1f  |hip|/|g64| %.4f  cos %.5f  %s' % (g64[n].norm().item(), g[n].norm().item() / g64[n].norm().item(), cos, n))
