"""diagnostic: per-parameter gradient error of stc_tt(att=...) vs the oracle on one small batch (fp32)"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import tcct_oracle as O
from tcct_amd.nets import stc_tt, RegNet
from test_model_gpu import make_kite

for att, (H, W) in (('pool', (32, 64)), ('factor', (32, 64)), ('pool', (64, 128)), ('factor', (64, 128))):
    img, lab = O.synth_batch(2, H, W, seed=5)
    model = RegNet(stc_tt(5, att=att), con='cos', out_channels=5)
    sd = O.formula_state_dict([(k, tuple(v.shape)) for k, v in model.state_dict().items()])
    model.load_state_dict(sd, strict=True)
    model.base.base_vit.drop_probs = [0.0] * 4
    k = make_kite(model.cuda().train(), '/tmp/fa_diag', False, False)
    loss, _ = k.calc_loss(img.cuda(), lab.cuda())
    loss.backward()
    torch.cuda.synchronize()
    for dt in (torch.float32, torch.float64):
        osd = {kk: v.clone().to(dt) if v.is_floating_point() else v.clone() for kk, v in sd.items()}
        for kk, v in osd.items():
            if v.is_floating_point() and not kk.endswith(('running_mean', 'running_var')) and not kk.startswith('fcp.'):
                v.requires_grad_(True)
        oh = torch.nn.functional.one_hot(lab, 5).permute(0, 3, 1, 2)
        tot, *_ = O.total_loss(osd, img.to(dt), oh, udh=False, reg=False)
        tot.backward()
        rows, n_h, n_o = [], 0.0, 0.0
        for name, p in model.named_parameters():
            g = osd[O.canonical_key(name)].grad if O.canonical_key(name) in osd else None
            if p.grad is None or g is None:
                continue
            a, b = p.grad.double().cpu(), g.double()
            n_h += float((a ** 2).sum()); n_o += float((b ** 2).sum())
            rows.append((float((a - b).norm() / (b.norm() + 1e-30)), float(b.norm()), name))
        rows.sort(reverse=True)
        print(f'== att={att} {H}x{W} oracle {dt}: loss hip {loss.item():.6f} oracle {tot.item():.6f}; |g| hip {n_h ** .5:.2f} oracle {n_o ** .5:.2f}')
        for r in rows[:6]:
            print('   relerr %.3e  |g| %.3e  %s' % r)
        big = sorted(rows, key=lambda r: -r[1])[:4]
        for r in big:
            print('   (largest) relerr %.3e  |g| %.3e  %s' % r)
