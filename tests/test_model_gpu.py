"""Whole-path parity on the GPU: HIP model + losses + optimizer step vs the golden fixtures generated from the
reference (tests/golden, made by oracle/make_golden.py) and vs the oracle on fresh seeded inputs.  fp32 tolerance 1e-3
(BASELINE.json north_star); bf16 is reported and only loosely bounded."""
import argparse
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', 'oracle'))
GOLD = os.path.join(HERE, 'golden')


def keys():
    return [(k, tuple(s)) for k, s in json.load(open(os.path.join(GOLD, 'state_dict_keys.json')))]


def build(dtype=torch.float32):
    import tcct_oracle as O
    from tcct_amd.nets import stc_tt, RegNet
    model = RegNet(stc_tt(5, compute_dtype=dtype), con='cos', out_channels=5)
    sd = O.formula_state_dict(keys())
    model.load_state_dict(sd, strict=True)
    return model.cuda().train(), sd


def make_kite(model, tmp_path, udh, reg, lr=1e-2):
    from tcct_amd.kite import KiteSeg

    class DS:
        out_channels = 5
    args = argparse.Namespace(los='di', lr=lr, gpu='0', pl=False, bs=2, coff_ds=1, udh=udh, reg=reg, epl=False,
                              coff_udh=1, coff_reg=.1, coff_epl=.1, bug=True)
    return KiteSeg(model=model, dataset=DS(), root=str(tmp_path), args=args)


def relerr(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / max(1.0, b.abs().max().item())).item()


def run_losses(k, fx, img, lab):
    model = k.model
    udh, reg = bool(fx['flags'][0]), bool(fx['flags'][1])
    if 'dp_masks' in fx:
        model.base.base_vit.forced_dp_masks = [torch.tensor(m, dtype=torch.float32) for m in fx['dp_masks']]
    else:
        model.base.base_vit.drop_probs = [0.0] * 4
    out = model(img)
    parts = {'dice': k.grad_calc(out, lab, ds=True, criterion=k.criterion)}
    if udh:
        parts['udh'] = model.regular_udh(out[0], lab) * 1.0
    if reg:
        noise = tuple(torch.tensor(fx[f'noise{i}']) for i in range(4))
        parts['reg'] = model.regular_reg(out[0], lab, noise=noise) * 0.1
    total = sum(parts.values())
    return out, parts, total


@pytest.mark.parametrize('name', ['full_2x32x32', 'full_2x64x64', 'di_2x64x64'])
def test_fp32_matches_reference_fixture(name, tmp_path):
    fx = dict(np.load(os.path.join(GOLD, name + '.npz')))
    model, sd0 = build(torch.float32)
    udh, reg = bool(fx['flags'][0]), bool(fx['flags'][1])
    k = make_kite(model, tmp_path, udh, reg)
    img = torch.tensor(fx['img']).cuda()                      # [B,1,H,W] -> replicated to 3 channels in-kernel
    lab = torch.tensor(fx['lab']).long().cuda()
    out, parts, total = run_losses(k, fx, img, lab)
    H = img.shape[2]
    sub = (slice(None), slice(None), slice(None, None, 4), slice(None, None, 4)) if H > 32 else (Ellipsis,)
    errs = {'out0': relerr(out[0], fx['out0'])}
    for i in (1, 2, 3):
        errs[f'out{i}'] = relerr(out[i][sub], fx[f'out{i}'])
    errs['feats'] = relerr(model.base.feats[0][sub], fx['feats'])
    errs['loss_dice'] = relerr(parts['dice'], fx['loss_dice'])
    if udh:
        errs['loss_udh'] = relerr(parts['udh'], fx['loss_udh'])
        errs['emb'] = relerr(torch.stack(model.emb_list, 0), fx['emb'])
    if reg:
        errs['loss_reg'] = relerr(parts['reg'], fx['loss_reg'])
        errs['edge_pred'] = relerr(model.edge_pred.view(-1), fx['edge_pred'].reshape(-1))
        errs['edge_true'] = relerr(model.edge_true.view(-1), fx['edge_true'].reshape(-1))
    errs['loss_total'] = relerr(total, fx['loss_total'])
    print(name, 'forward errs', {a: f'{b:.2e}' for a, b in errs.items()})
    for a, b in errs.items():
        assert b < 1e-3, (a, b)
    # masks + Dice metric of the 1e-3 criterion
    from tcct_amd.kite.losses import MDiceLoss, MIouLoss
    model.eval()
    # (train-mode logits were checked above; the metric kernels are checked on the reference's own train-mode mask)
    import tcct_oracle as O
    mask = O.predict_mask(torch.tensor(fx['out0']))
    f1 = MDiceLoss.scorem(mask.cuda(), lab, start_idx=1).item()
    io = MIouLoss.scorem(mask.cuda(), lab, start_idx=1).item()
    assert abs(f1 - float(fx['dice_scorem'])) < 1e-5 and abs(io - float(fx['iou_scorem'])) < 1e-5
    model.train()
    # backward
    k.optimG.zero_grad(set_to_none=True)
    total.backward()
    named = dict(model.named_parameters())
    names = [str(n) for n in fx['grad_names']]
    have = sorted(n for n, p in named.items() if p.grad is not None)
    assert have == sorted(names), (set(have) ^ set(names))
    gmax = max(float(v) for v in fx['grad_l2'])
    worst = 0.0
    for n, l2 in zip(names, fx['grad_l2']):
        g = named[n].grad.double().norm().item()
        # tensors whose true gradient is ~0 (conv bias in front of a train-mode BN) carry only rounding noise
        if n.endswith('.bias') and float(l2) < 2e-2:
            continue
        e = abs(g - float(l2)) / max(float(l2), 1e-3 * gmax)
        worst = max(worst, e)
        assert e < 2e-3, (n, g, float(l2))
    for key in fx:
        if key.startswith('grad:'):
            n = key[5:]
            e = relerr(named[n].grad, fx[key])
            assert e < 1e-3 * max(1.0, 1.0), (n, e)
    print(name, 'worst grad-norm rel err', f'{worst:.2e}')
    # optimizer step: clip(12) + AdamW at the reference's lr (CyclicLR start 1e-6)
    before = {n: named[n].detach().clone() for n in names}
    k.optimG.step()
    assert abs(k.optimG.last_total_norm.item() - float(fx['grad_total_norm'])) / float(fx['grad_total_norm']) < 1e-3
    lr = float(fx['lr'])
    assert abs(k.optimG.param_groups[0]['lr'] - lr) < 1e-12
    for key in fx:
        if key.startswith('step:'):
            n = key[5:]
            if n.endswith('.bias') and float(fx['grad_l2'][names.index(n)]) < 2e-2:
                continue
            d = (named[n].detach().double() - before[n].double()).cpu().numpy() / lr
            ref = fx[key]
            # step 1 of Adam is ~ -sign(g): compare where the reference gradient is not at noise level
            gref = fx['grad:' + n]
            big = np.abs(gref) > 1e-3 * np.abs(gref).max()
            assert np.abs(d - ref)[big].max() < 5e-2, (n, np.abs(d - ref)[big].max())
    # BN running statistics after one train-mode forward (lap_map's BN is applied twice per step)
    sd = model.state_dict()
    for key in fx:
        if key.startswith('buf:'):
            assert relerr(sd[key[4:]].float(), fx[key].astype(np.float32)) < 1e-4, key


def test_bf16_close_to_fixture(tmp_path):
    fx = dict(np.load(os.path.join(GOLD, 'full_2x64x64.npz')))
    model, _ = build(torch.bfloat16)
    k = make_kite(model, tmp_path, True, True)
    img = torch.tensor(fx['img']).cuda()
    lab = torch.tensor(fx['lab']).long().cuda()
    out, parts, total = run_losses(k, fx, img, lab)
    e = relerr(out[0], fx['out0'])
    print('bf16 logits rel err', e, 'loss', total.item(), 'ref', float(fx['loss_total']))
    assert e < 0.15
    assert abs(total.item() - float(fx['loss_total'])) / float(fx['loss_total']) < 2e-2
    total.backward()
    k.optimG.step()
    assert torch.isfinite(k.optimG.last_total_norm).item()
