"""CPU: the oracle (oracle/tcct_oracle.py) reproduces the fixtures generated from the real reference (tests/golden/*.npz,
made by oracle/make_golden.py in the build container).  This pins the checker itself."""
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', 'oracle'))
GOLD = os.path.join(HERE, 'golden')
import tcct_oracle as O  # noqa: E402


def keys():
    return [(k, tuple(s)) for k, s in json.load(open(os.path.join(GOLD, 'state_dict_keys.json')))]


@pytest.mark.parametrize('name', ['full_2x32x32', 'full_2x64x64', 'di_2x64x64', 'reg_2x64x64', 'full_2x128x128'])
def test_oracle_reproduces_reference_fixture(name):
    torch.set_num_threads(4)
    fx = dict(np.load(os.path.join(GOLD, name + '.npz')))
    sd = O.formula_state_dict(keys())
    names = [str(n) for n in fx['grad_names']]
    for n in names:
        sd[n].requires_grad_(True)
    img = torch.tensor(fx['img']).repeat(1, 3, 1, 1)
    lab = torch.tensor(fx['lab']).long()
    oh = torch.nn.functional.one_hot(lab, 5).permute(0, 3, 1, 2)
    udh, reg = bool(fx['flags'][0]), bool(fx['flags'][1])
    masks = [torch.tensor(m, dtype=torch.float32) for m in fx['dp_masks']] if 'dp_masks' in fx else None
    noise = tuple(torch.tensor(fx[f'noise{i}']) for i in range(4)) if reg else None
    want = {}
    tot, parts, outs, feats = O.total_loss(sd, img, oh, udh=udh, reg=reg, dp_masks=masks, noise=noise, want=want)
    H = img.shape[2]
    sub = (slice(None), slice(None), slice(None, None, 4), slice(None, None, 4)) if H > 32 else (Ellipsis,)

    def close(a, b, tol=2e-5):
        a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
        assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())
    close(outs[0], fx['out0'])
    for i in (1, 2, 3):
        close(outs[i][sub], fx[f'out{i}'])
    close(feats[sub], fx['feats'])
    close(tot, fx['loss_total'])
    close(parts['dice'], fx['loss_dice'])
    if udh:
        close(parts['udh'], fx['loss_udh'])
        close(want['emb'], fx['emb'], 1e-4)
    if reg:
        close(parts['reg'], fx['loss_reg'])
        close(want['edge_pred'], fx['edge_pred'])
        close(want['edge_true'], fx['edge_true'])
    tot.backward()
    have = sorted(n for n in sd if getattr(sd[n], 'grad', None) is not None)
    assert have == sorted(names)
    for n, l2, sm in zip(names, fx['grad_l2'], fx['grad_sum']):
        if n.endswith('.bias') and l2 < 2e-2:        # mathematically-zero gradients (bias in front of a train-mode BN)
            continue
        g = sd[n].grad.double()
        assert abs(g.norm().item() - l2) <= 1e-4 * max(1.0, l2), n
    for key in fx:
        if key.startswith('grad:'):
            close(sd[key[5:]].grad, fx[key], 5e-5)
        if key.startswith('buf:'):
            close(sd[key[4:]].float(), fx[key].astype(np.float32), 1e-5)
    mask = O.predict_mask(torch.tensor(fx['out0']))
    close(O.dice_scorem(mask, oh, 1), fx['dice_scorem'], 1e-6)
    close(O.iou_scorem(mask, oh, 1), fx['iou_scorem'], 1e-6)


def test_oracle_reproduces_trained_weights_train_fixture():
    """tests/golden/duke_train_2x160x160.npz (oracle/make_golden_duke_train.py): the reference's real trained 9-class checkpoint in train
    mode, full loss, backward -- the well-conditioned reference-held fixture.  The oracle must reproduce outputs, losses and gradients."""
    torch.set_num_threads(4)
    fx = np.load(os.path.join(GOLD, 'duke_train_2x160x160.npz'))
    ck = np.load(os.path.join(GOLD, 'ckpt_duke.npz'))
    sd = {}
    for k in ck.files:
        if k.startswith('w::'):
            sd[k[3:]] = torch.from_numpy(ck[k].view(np.int16).copy()).view(torch.bfloat16).float()
        elif k.startswith('i::'):
            sd[k[3:]] = torch.from_numpy(ck[k].copy())
    names = [str(n) for n in fx['grad_names']]
    for n in names:
        sd[n].requires_grad_(True)
    C = int(fx['n_class'])
    img = torch.from_numpy(fx['crops_u8']).permute(0, 3, 1, 2).float() / 255
    oh = torch.nn.functional.one_hot(torch.from_numpy(fx['lab']).long(), C).permute(0, 3, 1, 2)
    masks = [torch.tensor(m, dtype=torch.float32) for m in fx['dp_masks']]
    noise = tuple(torch.tensor(fx[f'noise{i}']) for i in range(4))
    want = {}
    tot, parts, outs, feats = O.total_loss(sd, img, oh, udh=True, reg=True, dp_masks=masks, noise=noise, want=want)
    sub = (slice(None), slice(None), slice(None, None, 4), slice(None, None, 4))

    def close(a, b, tol=1e-4):
        a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
        assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())
    close(outs[0], fx['out0'])
    for i in (1, 2, 3):
        close(outs[i][sub], fx[f'out{i}'])
    close(feats[sub], fx['feats'])
    for nm in ('dice', 'udh', 'reg'):
        close(parts[nm], fx['loss_' + nm], 2e-5)
    close(tot, fx['loss_total'], 2e-5)
    close(want['edge_pred'], fx['edge_pred'])
    close(want['edge_true'], fx['edge_true'])
    tot.backward()
    gmax = float(fx['grad_max'].max())
    n_full = 0
    for key in fx.files:
        if key.startswith('grad:'):
            ref = torch.from_numpy(fx[key]).double()
            if ref.abs().max().item() < 1e-4 * gmax:
                continue
            e = (sd[key[5:]].grad.double() - ref).norm().item() / ref.norm().item()
            assert e < 1e-3, (key, e)
            n_full += 1
    assert n_full >= 20


@pytest.mark.parametrize('name', ['di_trained_2x64x64', 'reg_trained_2x64x64', 'full_trained_2x64x64'])
def test_oracle_reproduces_trained5_fixture(name):
    """tests/golden/*_trained_2x64x64.npz (oracle/make_golden_trained5.py): the REAL reference trained by itself for 300 CPU steps
    (5 classes, weights bf16-rounded in ckpt_trained5.npz), one recorded train step per loss configuration of BASELINE cfg1 / cfg3 /
    cfg4.  Well conditioned (`cond_*`: the reference's fp32 result sits <= 1e-6 on the heads and <= 2.1e-4 per gradient tensor from an
    fp64 evaluation of the same graph), so everything is pinned tightly."""
    torch.set_num_threads(4)
    fx = np.load(os.path.join(GOLD, name + '.npz'))
    assert fx['cond_heads'].max() < 1e-5 and float(fx['cond_grad_max']) < 1e-3
    ck = np.load(os.path.join(GOLD, 'ckpt_trained5.npz'))
    sd = {}
    for k in ck.files:
        if k.startswith('w::'):
            sd[k[3:]] = torch.from_numpy(ck[k].view(np.int16).copy()).view(torch.bfloat16).float()
        elif k.startswith('i::'):
            sd[k[3:]] = torch.from_numpy(np.asarray(ck[k]).copy())
    names = [str(n) for n in fx['grad_names']]
    for n in names:
        sd[n].requires_grad_(True)
    udh, reg = bool(fx['flags'][0]), bool(fx['flags'][1])
    img = torch.tensor(fx['img']).repeat(1, 3, 1, 1)
    oh = torch.nn.functional.one_hot(torch.from_numpy(fx['lab']).long(), 5).permute(0, 3, 1, 2)
    masks = [torch.tensor(m, dtype=torch.float32) for m in fx['dp_masks']]
    noise = tuple(torch.tensor(fx[f'noise{i}']) for i in range(4)) if reg else None
    want = {}
    tot, parts, outs, feats = O.total_loss(sd, img, oh, udh=udh, reg=reg, dp_masks=masks, noise=noise, want=want)
    sub = (slice(None), slice(None), slice(None, None, 4), slice(None, None, 4))

    def close(a, b, tol=2e-5):
        a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
        assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item())
    close(outs[0], fx['out0'])
    for i in (1, 2, 3):
        close(outs[i][sub], fx[f'out{i}'])
    close(feats[sub], fx['feats'])
    close(tot, fx['loss_total'])
    close(parts['dice'], fx['loss_dice'])
    if udh:
        close(parts['udh'], fx['loss_udh'])
    if reg:
        close(parts['reg'], fx['loss_reg'])
        close(want['edge_pred'], fx['edge_pred'])
        close(want['edge_true'], fx['edge_true'])
    tot.backward()
    assert sorted(n for n in sd if getattr(sd[n], 'grad', None) is not None) == sorted(names)
    gmax = float(fx['grad_max'].max())
    n_full = 0
    for key in fx.files:
        if key.startswith('grad:'):
            ref = torch.from_numpy(fx[key]).double()
            if ref.abs().max().item() < 1e-4 * gmax:
                continue
            e = (sd[key[5:]].grad.double() - ref).norm().item() / ref.norm().item()
            assert e < 1e-3, (key, e)
            n_full += 1
    assert n_full >= 20
    mask = O.predict_mask(torch.tensor(fx['out0']))
    assert (mask.argmax(1).numpy() == fx['mask0']).all()
    close(O.dice_scorem(mask, oh, 1), fx['dice_scorem'], 1e-6)
    close(O.iou_scorem(mask, oh, 1), fx['iou_scorem'], 1e-6)


@pytest.mark.parametrize('name', ['traj5_di', 'traj5_reg', 'traj5_full'])
def test_oracle_follows_the_reference_for_five_steps(name):
    """tests/golden/traj5_*.npz (oracle/make_golden_traj5.py): FIVE consecutive steps of the real reference's own loop from ckpt_trained5 (its
    calc_loss, backward, clip_grad_norm_(12), AdamW(wd 2e-4); lr 1e-3 for three steps, then its CyclicLR's 2.575e-5; kite/loop_seg.py:108-142,
    kite/loopback.py:102-128).  The oracle runs the five steps FREELY -- its own weights, Adam moments and BatchNorm buffers, never re-synchronised --
    and must stay on the reference's trajectory: per-step loss parts and gradient norms, the weights' displacement after step 5, the moments, every
    BatchNorm running statistic and num_batches_tracked (lap_map's BatchNorm runs twice per step)."""
    torch.set_num_threads(4)
    fx = np.load(os.path.join(GOLD, name + '.npz'))
    ck = np.load(os.path.join(GOLD, 'ckpt_trained5.npz'))
    sd = {}
    for k in ck.files:
        if k.startswith('w::'):
            sd[k[3:]] = torch.from_numpy(ck[k].view(np.int16).copy()).view(torch.bfloat16).float()
        elif k.startswith('i::'):
            sd[k[3:]] = torch.from_numpy(np.asarray(ck[k]).copy())
    sd0 = {k: v.clone() for k, v in sd.items()}
    names = [str(n) for n in fx['names']]
    udh, reg = bool(fx['flags'][0]), bool(fx['flags'][1])
    m = {n: torch.zeros_like(sd[n]) for n in names}
    v = {n: torch.zeros_like(sd[n]) for n in names}
    for t in range(int(fx['n_steps'])):
        for n in names:
            sd[n] = sd[n].detach().requires_grad_(True)
        img = torch.tensor(fx['img'][t]).repeat(1, 3, 1, 1)
        oh = torch.nn.functional.one_hot(torch.from_numpy(fx['lab'][t]).long(), 5).permute(0, 3, 1, 2)
        masks = [torch.tensor(x, dtype=torch.float32) for x in fx['dp_masks'][t]]
        noise = tuple(torch.tensor(fx[f'noise{t}_{j}']) for j in range(4)) if reg else None
        tot, parts, _, _ = O.total_loss(sd, img, oh, udh=udh, reg=reg, dp_masks=masks, noise=noise)
        tot.backward()
        assert sorted(n for n in sd if getattr(sd[n], 'grad', None) is not None) == sorted(names)
        total = O.clip_adamw_step([sd[n].data for n in names], [sd[n].grad for n in names], [m[n] for n in names], [v[n] for n in names],
                                  t + 1, float(fx['lr'][t]))
        for n in names:
            sd[n] = sd[n].detach()
        assert abs(tot.item() - float(fx['loss_total'][t])) <= 1e-4 * max(1.0, abs(float(fx['loss_total'][t]))), (t, tot.item())
        assert abs(parts['dice'].item() - float(fx['loss_dice'][t])) <= 1e-4
        if udh:
            assert abs(parts['udh'].item() - float(fx['loss_udh'][t])) <= 1e-4
        if reg:
            assert abs(parts['reg'].item() - float(fx['loss_reg'][t])) <= 1e-4
        assert abs(total.item() - float(fx['grad_total_norm'][t])) <= 1e-3 * float(fx['grad_total_norm'][t]), (t, total.item())
    signal = dict(zip(names, fx['signal']))
    n_w = 0
    for key in fx.files:
        if key[:2] not in ('w:', 'm:', 'v:'):
            continue
        n = key[2:]
        if not signal[n]:               # exact-zero gradient (a bias in front of a train-mode BatchNorm): rounding noise through Adam, also in the reference
            continue
        ref = torch.from_numpy(fx[key]).double()
        if key[0] == 'w':
            d_ref, d = ref - sd0[n].double(), sd[n].double() - sd0[n].double()
            assert (d - d_ref).norm().item() <= 2e-2 * d_ref.norm().item(), (key, (d - d_ref).norm().item() / d_ref.norm().item())
            n_w += 1
        else:
            got = (m if key[0] == 'm' else v)[n].double()
            assert (got - ref).norm().item() <= 5e-3 * ref.norm().item(), (key, (got - ref).norm().item() / ref.norm().item())
    assert n_w >= 25
    lr_sum = float(fx['lr'].sum())
    for n, l2, dl2 in zip(names, fx['w_l2'], fx['disp_l2']):
        disp = (sd[n].double() - sd0[n].double()).norm().item()
        if signal[n]:
            assert abs(sd[n].double().norm().item() - l2) <= 1e-4 * max(l2, 1e-3), n
            assert abs(disp - dl2) <= 2e-2 * dl2, n
        else:       # Adam turns rounding-level gradients into steps of at most lr per element and step, in the reference (dl2) as here
            bound = 1.02 * lr_sum * sd[n].numel() ** 0.5 + 1e-3 * sd0[n].double().norm().item() * lr_sum
            assert disp <= bound and dl2 <= bound, (n, disp, dl2, bound)
    n_buf = 0
    for key in fx.files:
        if key.startswith('buf:'):
            ref, got = torch.from_numpy(np.asarray(fx[key])), sd[key[4:]]
            if key.endswith('num_batches_tracked'):
                assert int(got) == int(ref), key
            else:
                # a running MEAN also carries the convolution bias in front of its BatchNorm, whose gradient is exactly zero: Adam walks it by +-lr of
                # random sign per step in the reference and here alike, and the 0.1-momentum average picks up to (1 - 0.9^5) of that difference up
                slack = 0.5 * lr_sum if key.endswith('running_mean') else 0.0
                assert (got.double() - ref.double()).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item()) + slack, key
            n_buf += 1
    assert n_buf >= 150 and int(sd['lap_map.1.num_batches_tracked']) - int(sd0['lap_map.1.num_batches_tracked']) == (10 if reg else 0)


def test_oracle_known_answers():
    """known-answer checks that need no reference"""
    # MetaPool == 3x3 box over (token, channel) with valid-count divisor, minus identity
    t = torch.arange(2 * 5 * 8, dtype=torch.float32).reshape(2, 5, 8)
    y = O.metapool(t)
    n, c = 2, 3
    box = t[0, n - 1:n + 2, c - 1:c + 2].mean()
    assert abs(y[0, n, c].item() - (box - t[0, n, c]).item()) < 1e-5
    assert abs(y[1, 0, 0].item() - (t[1, 0:2, 0:2].mean() - t[1, 0, 0]).item()) < 1e-5
    # Dice of a perfect (one-hot-certain) prediction is 0 per class; uniform softmax at init gives ~0.8 per class and head
    lab = torch.randint(0, 5, (2, 16, 16), generator=torch.Generator().manual_seed(0))
    oh = torch.nn.functional.one_hot(lab, 5).permute(0, 3, 1, 2)
    assert O.dice_multi(oh.float() * 100, oh).item() < 1e-4
    z = torch.zeros(2, 5, 16, 16)
    assert abs(O.deep_supervision([z, z, z, z], oh).item() - 16.0) < 0.3
    # prob_true has exactly k ones in a column with k label boundaries
    _, lab = O.synth_batch(1, 64, 16, seed=3)
    oh = torch.nn.functional.one_hot(lab, 5).permute(0, 3, 1, 2)
    true = oh[:, 1:].float()
    pt = torch.nn.functional.pad((true[:, :, 1:] - true[:, :, :-1]).abs(), (0, 0, 1, 0)).sum(1).clamp_max(1)
    nb = (lab[:, 1:] != lab[:, :-1]).sum(1)
    assert torch.equal(pt.sum(1).long(), nb)
    # AdamW+clip restatement vs torch.optim.AdamW
    g = torch.Generator().manual_seed(1)
    ps = [torch.nn.Parameter(torch.randn(7, 5, generator=g)), torch.nn.Parameter(torch.randn(11, generator=g))]
    qs = [p.detach().clone() for p in ps]
    M, V = [torch.zeros_like(q) for q in qs], [torch.zeros_like(q) for q in qs]
    opt = torch.optim.AdamW(ps, lr=1e-3, weight_decay=2e-4)
    for step in range(1, 4):
        gr = [torch.randn(p.shape, generator=g) * 10 for p in ps]
        for p, x in zip(ps, gr):
            p.grad = x.clone()
        tn = torch.nn.utils.clip_grad_norm_(ps, 12)
        opt.step()
        on = O.clip_adamw_step(qs, [x.clone() for x in gr], M, V, step, 1e-3)
        assert abs(on.item() - tn.item()) < 1e-4
        for p, q in zip(ps, qs):
            assert (p.detach() - q).abs().max().item() < 1e-6


def test_goals_preprocessing_oracle_known_answers():
    """numpy restatement of the GOALS image preprocessing (parity unpinned: OpenCV / albumentations are absent): known answers of
    the nearest-neighbour rule and the round trip prep -> post"""
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'oracle'))
    import goals_oracle as G
    assert list(G.nn_index(8, 4)) == [0, 0, 1, 1, 2, 2, 3, 3]                      # 2x upsample repeats
    assert list(G.nn_index(4, 8)) == [0, 2, 4, 6]                                  # 2x downsample takes every other sample
    assert list(G.nn_index(5, 5)) == [0, 1, 2, 3, 4]                               # identity
    assert G.nn_index(512, 1100)[-1] == int(np.floor(511 * (1.0 / (512 / 1100)))) and G.nn_index(1100, 512).max() == 511
    g = np.random.default_rng(0)
    img = g.integers(0, 256, size=(2, 800, 1100, 3), dtype=np.uint8)
    cls = np.sort(g.integers(0, 5, size=(2, 800, 1100)), axis=1).astype(np.uint8)
    im2, lab2 = G.goals_prep(img, cls * 30)
    assert im2.shape == (2, 608, 512, 3) and lab2.shape == (2, 608, 512) and lab2.max() <= 4
    assert np.array_equal(lab2[0, :, 0], cls[0, :608, 0]) and np.array_equal(im2[1, 5, 0], img[1, 5, 0])
    back = G.goals_post(lab2)
    assert back.shape == (2, 800, 1100) and (back[:, 608:] == 0).all() and set(np.unique(back)) <= {0, 30, 60, 90, 120}
    c = G.crop_flip(img, True, 3, 5, 256, 256, True, False)
    assert c.shape == (2, 256, 256, 3) and np.array_equal(c[0, 0, 0], img[0, 3, 5 + 255])
    # secondary pin: an independent implementation of the same rule.  torch documents F.interpolate(mode='nearest') as
    # the mode that "matches OpenCV's INTER_NEAREST" (as opposed to 'nearest-exact' = PIL / scikit-image); at the exact GOALS sizes
    # (608 x 1100 -> 608 x 512 -> 608 x 1100) it picks the same source pixel as the restatement for every destination pixel
    import torch.nn.functional as F
    for dn, sn in ((512, 1100), (1100, 512), (608, 608)):
        ref = F.interpolate(torch.arange(sn, dtype=torch.float64).view(1, 1, 1, sn), size=(1, dn), mode='nearest').view(-1).long().numpy()
        assert np.array_equal(G.nn_index(dn, sn), ref), (dn, sn)
    t_img = torch.from_numpy(img[:, :608]).permute(0, 3, 1, 2).double()
    assert np.array_equal(F.interpolate(t_img, size=(608, 512), mode='nearest').permute(0, 2, 3, 1).numpy().astype(np.uint8), im2)
    t_lab = torch.from_numpy(lab2.astype(np.float64) * 30)[:, None]
    assert np.array_equal(F.interpolate(t_lab, size=(608, 1100), mode='nearest')[:, 0].numpy().astype(np.uint8), back[:, :608])


@pytest.mark.parametrize('tag', ['fa64', 'fa96'])
def test_factor_attention_oracle_matches_reference_fixture(tag):
    """tests/golden/factoratt.npz holds the REAL reference classes' forward / backward (FactorAtt_ConvRelPosEnc + ConvRelPosEnc,
    nets/tcct.py:219-341; oracle/make_golden_factoratt.py): the restatement must reproduce output and every gradient"""
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'oracle'))
    import tcct_oracle as O
    fx = {k[len(tag) + 1:]: torch.tensor(v) for k, v in np.load(os.path.join(os.path.dirname(__file__), 'golden', 'factoratt.npz')).items()
          if k.startswith(tag + '.')}
    H, W = (int(v) for v in fx['size'])
    x = fx['x'].clone().requires_grad_(True)
    ps = {k[2:]: v.clone().requires_grad_(True) for k, v in fx.items() if k.startswith('p.')}
    wb = [(ps[f'crpe.conv_list.{i}.weight'], ps[f'crpe.conv_list.{i}.bias']) for i in range(3)]
    y = O.factor_att(x, ps['qkv.weight'], ps['qkv.bias'], ps['proj.weight'], ps['proj.bias'], wb, (H, W), int(fx['heads']))
    y.backward(fx['gout'])
    assert torch.allclose(y, fx['y'], rtol=1e-5, atol=1e-5 * float(fx['y'].abs().max()))
    assert torch.allclose(x.grad, fx['dx'], rtol=1e-5, atol=1e-5 * float(fx['dx'].abs().max()))
    for k, p in ps.items():
        g = fx['g.' + k]
        assert torch.allclose(p.grad, g, rtol=1e-4, atol=1e-5 * float(g.abs().max())), k
    # known answer: with q = 0 the mixer output is exactly the projection bias (both terms are linear in q)
    z = O.factor_att(x.detach(), torch.zeros_like(ps['qkv.weight']), None, ps['proj.weight'].detach(), ps['proj.bias'].detach(),
                     [(w.detach(), b.detach()) for w, b in wb], (H, W), int(fx['heads']))
    assert torch.allclose(z, ps['proj.bias'].detach().expand_as(z), atol=1e-6)
