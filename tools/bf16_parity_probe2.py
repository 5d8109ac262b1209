"""Well-conditioned bf16 parity probe: seeded DEFAULT-initialised weights (not the formula weights of the fixtures, whose train-mode
network amplifies any rounding by ~1e3), 2 x 128 x 128, full loss.  HIP bf16 vs the rounding-point oracle vs the fp32 oracle."""
import argparse, contextlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import tcct_oracle as O
from tcct_amd.nets import stc_tt, RegNet
from tcct_amd.kite import KiteSeg

H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
udh = reg = (len(sys.argv) <= 2 or sys.argv[2] == 'full')


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / max(1e-30, b.abs().max().item())).item()


torch.manual_seed(0)
ref = RegNet(stc_tt(5), con='cos', out_channels=5)
sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
img, lab = O.synth_batch(2, H, H, seed=31)
oh = torch.nn.functional.one_hot(lab, 5).permute(0, 3, 1, 2)
g = torch.Generator().manual_seed(5)
noise = (torch.rand(2, 4, H, H, generator=g), torch.rand(2, 4, H, H, generator=g), torch.rand(1, 1, H, 1, generator=g), torch.rand(1, 1, H, 1, generator=g))


def hip(dt):
    model = RegNet(stc_tt(5, compute_dtype=dt), con='cos', out_channels=5)
    model.load_state_dict(sd0)
    class DS: out_channels = 5
    args = argparse.Namespace(los='di', lr=1e-2, gpu='0', pl=False, bs=2, coff_ds=1, udh=udh, reg=reg, epl=False, coff_udh=1, coff_reg=.1, coff_epl=.1, bug=True)
    k = KiteSeg(model=model, dataset=DS(), root='/tmp/probe2', args=args)
    model.train(); model.base.base_vit.drop_probs = [0.0] * 4
    out = model(img[:, :1].cuda())
    parts = {'dice': k.grad_calc(out, lab.cuda(), ds=True, criterion=k.criterion)}
    if udh:
        parts['udh'] = model.regular_udh(out[0], lab.cuda())
    if reg:
        parts['reg'] = model.regular_reg(out[0], lab.cuda(), noise=noise) * 0.1
    tot = sum(parts.values())
    tot.backward()
    return tot.item(), [o.detach().float().cpu() for o in out], {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}


def orc(mode):
    sd = {k: v.clone() for k, v in sd0.items()}
    for n, v in sd.items():
        if v.is_floating_point() and not n.endswith(('running_mean', 'running_var')) and not n.startswith('fcp.'):
            v.requires_grad_(True)
    ctx = O.rounding_points('bf16') if mode == 'bf16' else contextlib.nullcontext()
    with ctx:
        tot, parts, outs, feats = O.total_loss(sd, img, oh, udh=udh, reg=reg, noise=noise if reg else None)
        tot.backward()
    return tot.item(), [o.detach() for o in outs], {n: v.grad for n, v in sd.items() if getattr(v, 'grad', None) is not None}


h32, hb = hip(torch.float32), hip(torch.bfloat16)
o32, ob = orc('fp32'), orc('bf16')
print(f'H={H} udh/reg={udh}')
print('loss: hip32', h32[0], 'hipbf16', hb[0], 'o32', o32[0], 'obf16', ob[0])
for lab_, a, b in (('HIP32 vs o32', h32, o32), ('HIPbf16 vs o-bf16', hb, ob), ('HIPbf16 vs o32', hb, o32), ('o-bf16 vs o32', ob, o32)):
    print(f'{lab_:20s} loss {abs(a[0] - b[0]) / abs(b[0]):.2e}  out', [f'{rel(x, y):.2e}' for x, y in zip(a[1], b[1])])
gn = {n: o32[2][n].norm().item() for n in o32[2]}
gmax = max(gn.values())
names = [n for n in sorted(gn) if gn[n] > 1e-3 * gmax and n in hb[2]]
for lab_, a, b in (('HIP32 vs o32', h32, o32), ('HIPbf16 vs o-bf16', hb, ob), ('HIPbf16 vs o32', hb, o32), ('o-bf16 vs o32', ob, o32)):
    e = np.array([(a[2][n] - b[2][n]).norm().item() / max(b[2][n].norm().item(), 1e-30) for n in names])
    print(f'grads {lab_:20s} rel-L2 median {np.median(e):.3e} p90 {np.percentile(e, 90):.3e} max {e.max():.3e} ({names[int(e.argmax())]})')
tn = lambda d: torch.sqrt(sum((g_.double() ** 2).sum() for g_ in d.values())).item()
print('total grad norm hip32', tn(h32[2]), 'hipbf16', tn(hb[2]), 'o32', tn(o32[2]), 'obf16', tn(ob[2]))
