"""GPU-box probe behind the bf16 parity bounds of tests/test_model_gpu.py: HIP bf16 vs the oracle with the same rounding points
(oracle/tcct_oracle.py rounding_points('bf16')) vs the fp32 oracle, on the golden fixtures; and the bf16-vs-fp32 validation Dice after
30 training steps.  Prints measurements only (the tests assert)."""
import argparse, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import tcct_oracle as O
from tcct_amd.nets import stc_tt, RegNet
from tcct_amd.kite import KiteSeg

GOLD = os.path.join(ROOT, 'tests', 'golden')
keys = [(k, tuple(s)) for k, s in json.load(open(os.path.join(GOLD, 'state_dict_keys.json')))]


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / max(1.0, b.abs().max().item())).item()


def kite(model, udh, reg, root):
    class DS: out_channels = 5
    args = argparse.Namespace(los='di', lr=1e-2, gpu='0', pl=False, bs=2, coff_ds=1, udh=udh, reg=reg, epl=False, coff_udh=1, coff_reg=.1, coff_epl=.1, bug=True)
    return KiteSeg(model=model, dataset=DS(), root=root, args=args)


def oracle_run(fx, mode, udh, reg):
    sd = O.formula_state_dict(keys)
    names = [str(n) for n in fx['grad_names']]
    for n in names:
        sd[n].requires_grad_(True)
    img = torch.tensor(fx['img']).repeat(1, 3, 1, 1)
    oh = torch.nn.functional.one_hot(torch.tensor(fx['lab']).long(), 5).permute(0, 3, 1, 2)
    dm = [torch.tensor(m).float() for m in fx['dp_masks']] if 'dp_masks' in fx else None
    nz = tuple(torch.tensor(fx[f'noise{i}']) for i in range(4)) if reg else None
    import contextlib
    ctx = O.rounding_points('bf16') if mode == 'bf16' else contextlib.nullcontext()
    want = {}
    with ctx:
        tot, parts, outs, feats = O.total_loss(sd, img, oh, udh=udh, reg=reg, dp_masks=dm, noise=nz, want=want)
        tot.backward()
    return tot.detach(), {k: v.detach() for k, v in parts.items()}, [o.detach() for o in outs], feats.detach(), {n: sd[n].grad for n in names}, want


for name in sys.argv[1:] or ['full_2x64x64', 'reg_2x64x64', 'full_2x128x128', 'full_2x32x32']:
    fx = dict(np.load(os.path.join(GOLD, name + '.npz')))
    udh, reg = bool(fx['flags'][0]), bool(fx['flags'][1])
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        model = RegNet(stc_tt(5, compute_dtype=dt), con='cos', out_channels=5)
        model.load_state_dict(O.formula_state_dict(keys), strict=True)
        model = model.cuda().train()
        k = kite(model, udh, reg, '/tmp/probe_root')
        if 'dp_masks' in fx:
            model.base.base_vit.forced_dp_masks = [torch.tensor(m, dtype=torch.float32) for m in fx['dp_masks']]
        else:
            model.base.base_vit.drop_probs = [0.0] * 4
        img = torch.tensor(fx['img']).cuda(); lab = torch.tensor(fx['lab']).long().cuda()
        out = model(img)
        parts = {'dice': k.grad_calc(out, lab, ds=True, criterion=k.criterion)}
        if udh:
            parts['udh'] = model.regular_udh(out[0], lab) * 1.0
        if reg:
            noise = tuple(torch.tensor(fx[f'noise{i}']) for i in range(4))
            parts['reg'] = model.regular_reg(out[0], lab, noise=noise) * 0.1
        total = sum(parts.values())
        feats = model.base.feats[0] if udh else None
        total.backward()
        res[dt] = (total.detach().cpu(), {a: b.detach().cpu() for a, b in parts.items()}, [o.detach().float().cpu() for o in out],
                   feats.detach().float().cpu() if feats is not None else None, {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None},
                   (model.edge_pred.cpu(), model.edge_true.cpu()) if reg else None)
    t0 = time.time()
    o32 = oracle_run(fx, 'fp32', udh, reg)
    ob = oracle_run(fx, 'bf16', udh, reg)
    print(f'== {name}: oracle runs {time.time() - t0:.1f}s')
    h32, hb = res[torch.float32], res[torch.bfloat16]
    print('fp32 HIP vs oracle32: loss', rel(h32[0], o32[0]), 'out', [f'{rel(a, b):.2e}' for a, b in zip(h32[2], o32[2])], 'feats', rel(h32[3], o32[3]) if udh else None)
    print('bf16 HIP vs oracle-bf16: loss', f'{rel(hb[0], ob[0]):.2e}', {a: f'{rel(hb[1][a], ob[1][a]):.2e}' for a in hb[1]}, 'out', [f'{rel(a, b):.2e}' for a, b in zip(hb[2], ob[2])], 'feats', f'{rel(hb[3], ob[3]):.2e}' if udh else None)
    print('bf16 HIP vs oracle32  : loss', f'{rel(hb[0], o32[0]):.2e}', 'out', [f'{rel(a, b):.2e}' for a, b in zip(hb[2], o32[2])])
    print('oracle-bf16 vs oracle32: loss', f'{rel(ob[0], o32[0]):.2e}', 'out', [f'{rel(a, b):.2e}' for a, b in zip(ob[2], o32[2])])
    if reg:
        print('edge: HIPbf16 vs oracle-bf16', rel(hb[5][0].view(-1), ob[5]['edge_pred'].reshape(-1)), 'HIP32 vs o32', rel(h32[5][0].view(-1), o32[5]['edge_pred'].reshape(-1)))
    # gradients: per-tensor relative L2 error, for tensors above the noise floor
    gn = {n: o32[4][n].norm().item() for n in o32[4]}
    gmax = max(gn.values())
    rows = []
    for n in sorted(gn):
        if gn[n] < 1e-3 * gmax:
            continue
        e_hb_ob = (hb[4][n] - ob[4][n]).norm().item() / max(ob[4][n].norm().item(), 1e-30)
        e_hb_o32 = (hb[4][n] - o32[4][n]).norm().item() / gn[n]
        e_ob_o32 = (ob[4][n] - o32[4][n]).norm().item() / gn[n]
        e_h32 = (h32[4][n] - o32[4][n]).norm().item() / gn[n]
        rows.append((n, e_hb_ob, e_hb_o32, e_ob_o32, e_h32))
    a = np.array([r[1:] for r in rows])
    print(f'grads ({len(rows)} tensors above 1e-3 of the largest norm): rel-L2 median / p90 / max')
    for j, lab_ in enumerate(('HIPbf16 vs oracle-bf16', 'HIPbf16 vs oracle32', 'oracle-bf16 vs oracle32', 'HIP32 vs oracle32')):
        print(f'   {lab_:28s} {np.median(a[:, j]):.3e} {np.percentile(a[:, j], 90):.3e} {a[:, j].max():.3e}   worst: {rows[int(a[:, j].argmax())][0]}')
    tot_hb = torch.sqrt(sum((g.double() ** 2).sum() for g in hb[4].values())).item()
    tot_ob = torch.sqrt(sum((g.double() ** 2).sum() for g in ob[4].values())).item()
    tot_32 = torch.sqrt(sum((g.double() ** 2).sum() for g in o32[4].values())).item()
    print('total grad norm: HIPbf16', tot_hb, 'oracle-bf16', tot_ob, 'oracle32', tot_32)
