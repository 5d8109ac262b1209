"""Full-size (BASELINE.json: bs 8, 1x800x1100 -> 3x800x1104) checks through size-independent properties: the oracle cannot run
8 full B-scans in seconds, so the HIP kernels are checked against each other (MFMA vs the fp32 VALU formulation), against
closed-form answers and through linearity / statistics identities."""
import pytest
import torch

pytestmark = pytest.mark.gpu
B, H, W = 8, 800, 1104


def _x(c=32, seed=0, dt=torch.bfloat16):
    g = torch.Generator(device='cuda').manual_seed(seed)
    return torch.randn((B, H, W, c), device='cuda', generator=g).to(dt)


@pytest.mark.parametrize('k', [(3, 3), (1, 13), (13, 1)])
def test_conv32_mfma_equals_valu_formulation_fullsize(k):
    from tcct_amd._lib import lib
    kh, kw = k
    x = _x()
    g = torch.Generator(device='cuda').manual_seed(1)
    w = torch.randn((32, 32, kh, kw), device='cuda', generator=g) / (32 * kh * kw) ** 0.5
    b = torch.randn(32, device='cuda', generator=g)
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    wp = torch.empty(kh * kw * 1024, device='cuda', dtype=torch.bfloat16)
    lib.conv32_pack_weights(w, wp, kh, kw, 0)
    lib.conv32_fwd(x, wp, b, y1, B, H, W, kh, kw, (kh - 1) // 2, (kw - 1) // 2)
    lib.conv2d_fwd(x, w.to(torch.bfloat16).float(), b, y2, B, H, W, 32, 32, 32, kh, kw, 1, (kh - 1) // 2, (kw - 1) // 2, 1, 1)
    d = (y1.float() - y2.float()).abs().max().item()
    assert d <= 2 ** -6 * max(1.0, y2.float().abs().max().item()), d          # both round the same fp32 sums to bf16
    # weight gradient: MFMA (transposing LDS reads) vs VALU, and linearity in dy
    dy1, dy2 = _x(seed=2), _x(seed=3)
    dw1, dw2, dw12, dwv = (torch.empty_like(w) for _ in range(4))
    db = torch.empty(32, device='cuda')
    lib.conv32_wgrad(x, dy1, dw1, db, B, H, W, kh, kw, (kh - 1) // 2, (kw - 1) // 2)
    lib.conv32_wgrad(x, dy2, dw2, None, B, H, W, kh, kw, (kh - 1) // 2, (kw - 1) // 2)
    dys = (dy1.float() + dy2.float())
    lib.conv2d_wgrad(x, dy1, dwv, None, B, H, W, 32, 32, 32, kh, kw, 1, (kh - 1) // 2, (kw - 1) // 2, 1, 1)
    scale = dwv.abs().max().item()
    assert (dw1 - dwv).abs().max().item() < 2e-3 * scale
    torch.testing.assert_close(db, dy1.float().sum((0, 1, 2)), rtol=2e-3, atol=2e-3 * dy1.float().sum((0, 1, 2)).abs().max().item())
    lib.conv2d_wgrad(x.float(), dys, dw12, None, B, H, W, 32, 32, 32, kh, kw, 1, (kh - 1) // 2, (kw - 1) // 2, 0, 0)
    assert (dw1 + dw2 - dw12).abs().max().item() < 3e-3 * dw12.abs().max().item()


def test_batchnorm_statistics_fullsize():
    from tcct_amd import ops
    x = (_x(seed=4).float() * 1.7 + 0.4).to(torch.bfloat16)
    gamma = torch.linspace(0.5, 1.5, 32, device='cuda')
    beta = torch.linspace(-0.3, 0.3, 32, device='cuda')
    rm, rv = torch.zeros(32, device='cuda'), torch.ones(32, device='cuda')
    nbt = torch.zeros((), device='cuda', dtype=torch.int64)
    y = ops.batchnorm(x, gamma, beta, rm, rv, nbt, training=True).float()
    m, v = y.mean((0, 1, 2)), y.var((0, 1, 2), unbiased=False)
    assert (m - beta).abs().max().item() < 5e-3 and (v - gamma ** 2).abs().max().item() < 2e-2
    assert (rm - 0.1 * 0.4).abs().max().item() < 2e-3 and (rv - (0.9 + 0.1 * 1.7 ** 2)).abs().max().item() < 2e-2


def test_dice_closed_forms_fullsize():
    from tcct_amd import ops
    lab = torch.randint(0, 5, (B, H, W), device='cuda', dtype=torch.uint8)
    onehot = torch.nn.functional.one_hot(lab.long(), 5).float()
    perfect = ops.softmax_dice((onehot * 60).contiguous(), lab)
    assert perfect.item() < 1e-4                                   # DiceLoss.dice(g, g) == 1 for every class
    uniform = ops.softmax_dice(torch.zeros((B, H, W, 5), device='cuda'), lab)
    # p = 1/5 everywhere: 1 - (1 + 2 n_c/5) / (1 + M/5 + n_c) summed over classes
    M = B * H * W
    nc = torch.bincount(lab.flatten().long(), minlength=5).double()
    ref = (1 - (1 + 2 * nc / 5) / (1 + M / 5 + nc)).sum().item()
    assert abs(uniform.item() - ref) < 1e-4


def test_full_step_runs_at_bench_shape(tmp_path):
    """one full training step at the bench configuration (bf16, full loss): finite loss ~16 at random init, finite clip norm,
    every trained parameter receives a gradient, BN buffers move"""
    import argparse
    from tcct_amd.nets import stc_tt, RegNet
    from tcct_amd.kite import KiteSeg
    from tcct_amd.data import SynthOCT
    torch.manual_seed(0)
    ds = SynthOCT(device='cuda')
    model = RegNet(stc_tt(5, compute_dtype=torch.bfloat16), con='cos', out_channels=5)
    args = argparse.Namespace(los='di', lr=1e-2, gpu='0', pl=False, bs=8, coff_ds=1, udh=True, reg=True, epl=False, coff_udh=1,
                              coff_reg=.1, coff_epl=.1, bug=True)
    k = KiteSeg(model=model, dataset=ds, root=str(tmp_path), args=args)
    k.model.train()
    img, lab, _, _ = ds.parse(ds.make_batch(8, 7))
    assert img.shape == (8, 1, 800, 1104)
    l1 = k.train_step(img, lab).item()
    l2 = k.train_step(img, lab).item()
    assert 10.0 < l1 < 25.0 and 10.0 < l2 < 25.0
    n = k.optimG.last_total_norm.item()
    assert n == n and 0 < n < 1e7
    assert k.optimG.flat_numel == 802298                       # elements that receive gradients with reg on
    assert model.base.base_cnn.cnn[1].num_batches_tracked.item() == 2 and model.lap_map[1].num_batches_tracked.item() == 4
    # eval path on the same weights: masks + metrics
    k.model.eval()
    from tcct_amd.kite.losses import MDiceLoss
    mask = k.predict(img[:1])
    d = MDiceLoss.scorem(mask, lab[:1], start_idx=1).item()
    assert 0.0 <= d <= 1.0


def test_cfg3_reg_only_step_at_bench_shape(tmp_path):
    """BASELINE.json configs[2]: `stc_tt --los=di --reg=true` (boundary regression on, feature polarization OFF), bs 8, 1x800x1100, bf16.
    With udh off nothing reads `model.feats`: the lazy norm_add side output must never be materialised (reference kite/loop_seg.py:
    158-165 with args.udh False), yet the trained-parameter set is the 802 298 elements of SURVEY 8(a21) (the reg loss adds lap_reg /
    lap_map), the loss is finite and its reg part is positive."""
    import argparse
    from tcct_amd.nets import stc_tt, RegNet
    from tcct_amd.kite import KiteSeg
    from tcct_amd.data import SynthOCT
    from tcct_amd import ops
    torch.manual_seed(0)
    ds = SynthOCT(device='cuda')
    model = RegNet(stc_tt(5, compute_dtype=torch.bfloat16), con='cos', out_channels=5)
    args = argparse.Namespace(los='di', lr=1e-2, gpu='0', pl=False, bs=8, coff_ds=1, udh=False, reg=True, epl=False, coff_udh=1,
                              coff_reg=.1, coff_epl=.1, bug=True)
    k = KiteSeg(model=model, dataset=ds, root=str(tmp_path), args=args)
    k.model.train()
    img, lab, _, _ = ds.parse(ds.make_batch(8, 11))
    calls = []
    real = ops.norm_add3
    ops.norm_add3 = lambda *a, **kw: (calls.append(1), real(*a, **kw))[1]
    try:
        tot, log = k.calc_loss(img, lab)
        assert 'reg=' in log and 'udh=' not in log
        reg_part = float(log.split('reg=')[1].split(',')[0])
        assert reg_part > 0 and 10.0 < tot.item() < 25.0
        l1 = k.train_step(img, lab).item()
        l2 = k.train_step(img, lab).item()
    finally:
        ops.norm_add3 = real
    assert not calls and model.base._feats is None, 'feats was materialised although nothing reads it'
    assert 10.0 < l1 < 25.0 and 10.0 < l2 < 25.0
    n = k.optimG.last_total_norm.item()
    assert n == n and 0 < n < 1e7
    assert k.optimG.flat_numel == 802298
    assert model.lap_map[1].num_batches_tracked.item() == 6          # applied to pred and true in each of the three forward passes


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_fullsize_forward_and_losses_match_the_oracle(dt, tmp_path):
    """bs 2 at the FULL bench resolution (3 x 800 x 1104: 34.5 tiles of 32 columns per row, 50 of 16 rows -- every tile-edge case of the
    MFMA convolutions, all five levels down to 50 x 69) against the oracle on the CPU, full loss (Dice + reg + fpl), seeded default
    weights.  fp32: the four heads, every loss scalar and the boundary coordinates at the 1e-3 contract (heads 1e-4).  bf16: the same
    against the oracle's rounding-point mode, as in tests/test_model_gpu.py::test_bf16_matches_rounding_point_oracle."""
    import argparse
    import contextlib
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
    import tcct_oracle as O
    from tcct_amd.nets import stc_tt, RegNet
    from tcct_amd.kite import KiteSeg
    H, W = 800, 1104
    torch.manual_seed(0)
    sd0 = {k: v.clone() for k, v in RegNet(stc_tt(5), con='cos', out_channels=5).state_dict().items()}
    img3, lab = O.synth_batch(2, H, W, seed=77)
    g = torch.Generator().manual_seed(9)
    noise = (torch.rand(2, 4, H, W, generator=g), torch.rand(2, 4, H, W, generator=g), torch.rand(1, 1, H, 1, generator=g), torch.rand(1, 1, H, 1, generator=g))
    model = RegNet(stc_tt(5, compute_dtype=dt), con='cos', out_channels=5)
    model.load_state_dict(sd0)

    class DS:
        out_channels = 5
    args = argparse.Namespace(los='di', lr=1e-2, gpu='0', pl=False, bs=2, coff_ds=1, udh=True, reg=True, epl=False, coff_udh=1, coff_reg=.1,
                              coff_epl=.1, bug=True)
    k = KiteSeg(model=model, dataset=DS(), root=str(tmp_path), args=args)
    model.train()
    model.base.base_vit.drop_probs = [0.0] * 4
    with torch.no_grad():
        out = model(img3[:, :1].cuda())
        parts = {'dice': k.grad_calc(out, lab.cuda(), ds=True, criterion=k.criterion).item(),
                 'udh': model.regular_udh(out[0], lab.cuda()).item(),
                 'reg': model.regular_reg(out[0], lab.cuda(), noise=noise).item() * 0.1}
    outs = [o.float().cpu() for o in out]
    edge = model.edge_pred.float().cpu().reshape(-1)
    oh = torch.nn.functional.one_hot(lab, 5).permute(0, 3, 1, 2)
    torch.set_num_threads(min(32, os.cpu_count() or 8))

    def oracle(mode):
        want = {}
        with torch.no_grad(), (O.rounding_points('bf16') if mode == 'bf16' else contextlib.nullcontext()):
            tot, p, o, _ = O.total_loss({kk: v.clone() for kk, v in sd0.items()}, img3, oh, udh=True, reg=True, noise=noise, want=want)
        return {a: b.item() for a, b in p.items()}, o, want['edge_pred'].reshape(-1)

    def rel(a, b):
        return ((a.double() - b.double()).abs().max() / max(1e-30, b.double().abs().max().item())).item()
    p32, o32, e32 = oracle('fp32')
    if dt == torch.float32:
        errs = [rel(a, b) for a, b in zip(outs, o32)]
        print('full-size fp32: heads', errs, 'losses', parts, p32, 'edge', rel(edge, e32))
        assert max(errs) < 1e-4
        for n in parts:
            assert abs(parts[n] - p32[n]) <= 1e-4 * max(1.0, abs(p32[n])), (n, parts[n], p32[n])
        assert rel(edge, e32) < 1e-3
        return
    pb, ob, eb = oracle('bf16')
    rep = []
    for i in range(4):
        model_err = rel(ob[i], o32[i])
        rep.append((i, model_err, rel(outs[i], ob[i]), rel(outs[i], o32[i])))
    print('full-size bf16: (head, model error, HIP vs model, HIP vs fp32)', rep, 'losses', parts, pb, p32)
    for i, model_err, d_model, d_fp32 in rep:
        assert d_model <= 0.6 * model_err and d_fp32 <= 1.3 * model_err + 1e-3, rep
    for n in parts:
        assert abs(parts[n] - pb[n]) <= 1e-3 * max(1.0, abs(pb[n])), (n, parts[n], pb[n], p32[n])
    assert rel(edge, eb) <= 0.6 * rel(eb, e32) + 1e-3


def test_fullsize_backward_matches_the_oracle(tmp_path):
    """ONE B-scan at the full bench resolution (3 x 800 x 1104), fp32, full loss (Dice + reg + fpl), seeded default weights: the BACKWARD
    pass against the oracle's CPU backward.  At this size every weight-gradient kernel accumulates over many tiles and blocks (level 4 is
    50 x 69, level 0 is 1 725 tiles of 16 x 32) -- the multi-tile accumulation and the per-block atomics the small fixtures never reach.
    Per-tensor relative L2 over all trained tensors above noise level and over a named list spanning CNN L0-L4, the ViT stages, the
    decoder and the loss modules: median <= 1e-3, max <= 1e-2; total norm within 2e-3."""
    import argparse
    import os
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
    import tcct_oracle as O
    from tcct_amd.nets import stc_tt, RegNet
    from tcct_amd.kite import KiteSeg
    H, W = 800, 1104
    torch.manual_seed(0)
    sd0 = {k: v.clone() for k, v in RegNet(stc_tt(5), con='cos', out_channels=5).state_dict().items()}
    img3, lab = O.synth_batch(1, H, W, seed=78)
    g = torch.Generator().manual_seed(10)
    noise = (torch.rand(1, 4, H, W, generator=g), torch.rand(1, 4, H, W, generator=g), torch.rand(1, 1, H, 1, generator=g), torch.rand(1, 1, H, 1, generator=g))
    model = RegNet(stc_tt(5, compute_dtype=torch.float32), con='cos', out_channels=5)
    model.load_state_dict(sd0)

    class DS:
        out_channels = 5
    args = argparse.Namespace(los='di', lr=1e-2, gpu='0', pl=False, bs=1, coff_ds=1, udh=True, reg=True, epl=False, coff_udh=1, coff_reg=.1,
                              coff_epl=.1, bug=True)
    k = KiteSeg(model=model, dataset=DS(), root=str(tmp_path), args=args)
    model.train()
    model.base.base_vit.drop_probs = [0.0] * 4
    out = model(img3[:, :1].cuda())
    tot = (k.grad_calc(out, lab.cuda(), ds=True, criterion=k.criterion) + model.regular_udh(out[0], lab.cuda())
           + model.regular_reg(out[0], lab.cuda(), noise=noise) * 0.1)
    k.optimG.zero_grad(set_to_none=True)
    tot.backward()
    gh = {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None}
    # oracle backward on the CPU
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    sd = {kk: v.clone() for kk, v in sd0.items()}
    for n, v in sd.items():
        if v.is_floating_point() and not n.endswith(('running_mean', 'running_var')) and not n.startswith('fcp.'):
            v.requires_grad_(True)
    oh = torch.nn.functional.one_hot(lab, 5).permute(0, 3, 1, 2)
    to, _, _, _ = O.total_loss(sd, img3, oh, udh=True, reg=True, noise=noise)
    to.backward()
    go = {n: v.grad.double() for n, v in sd.items() if getattr(v, 'grad', None) is not None}
    assert set(gh) == set(go)
    assert abs(tot.item() - to.item()) <= 1e-4 * abs(to.item())
    gmaxn = max(v.norm().item() for v in go.values())
    big = [n for n in sorted(go) if go[n].norm().item() > 1e-3 * gmaxn]
    e = {n: (gh[n] - go[n]).norm().item() / go[n].norm().item() for n in big}
    named = ['base.base_cnn.cnn.0.weight', 'base.base_cnn.path_estan.0.block12.0.weight', 'base.base_cnn.path_estan.0.block34.0.weight',
             'base.base_cnn.path_estan.0.block34.1.weight', 'base.base_cnn.path_estan.0.block5.0.weight', 'base.base_cnn.path_estan.1.block12.1.weight',
             'base.base_cnn.path_estan.2.block34.2.weight', 'base.base_cnn.path_estan.3.block5.0.weight', 'base.base_cnn.path_estan.4.block12.0.weight',
             'base.base_vit.stem.0.conv.weight', 'base.base_vit.stem.1.conv.weight',
             'base.base_vit.patch_embed_stages.0.patch_embeds.0.patch_conv.dwconv.weight', 'base.base_vit.mhca_stages.0.InvRes.conv1.conv.weight',
             'base.base_vit.mhca_stages.0.aggregate.conv.weight', 'base.base_vit.mhca_stages.1.mhca_blks.0.MHCA_layers.0.mlp.fc1.weight',
             'base.base_vit.mhca_stages.2.InvRes.dwconv.weight', 'base.base_vit.mhca_stages.3.mhca_blks.0.cpe.proj.weight',
             'base.tran_vit0.0.weight', 'base.tran_cnn3.0.weight', 'base.head.0.weight', 'base.dec1.prep.0.weight', 'base.dec4.post.0.weight',
             'base.t324.weight', 'base.aux0.weight', 'base.aux4.weight', 'lap_reg.0.weight', 'lap_map.0.weight']
    # (the boundary-loss modules lap_reg / lap_map see gradients ~1e-9 of the largest tensor's at these weights -- 0.1 x a loss that barely
    # depends on them: fp32 noise level, where torch's own fp32 result is as far from an fp64 evaluation; they get 5e-2)
    en = {n: (gh[n] - go[n]).norm().item() / max(go[n].norm().item(), 1e-30) for n in named if not n.startswith('lap_')}
    for n in named:
        if n.startswith('lap_'):
            assert (gh[n] - go[n]).norm().item() <= 5e-2 * go[n].norm().item(), n
    v = np.array(list(e.values()))
    vn = np.array(list(en.values()))
    print(f'full-size fp32 backward: {len(big)} tensors rel-L2 median {np.median(v):.2e} p90 {np.percentile(v, 90):.2e} max {v.max():.2e} '
          f'({max(e, key=e.get)}); named {len(named)}: median {np.median(vn):.2e} max {vn.max():.2e} ({max(en, key=en.get)})')
    assert len(en) >= 20 and np.median(vn) <= 1e-3 and vn.max() <= 1e-2, en
    assert np.median(v) <= 1e-3 and v.max() <= 1e-2, sorted(e.items(), key=lambda t: -t[1])[:5]
    tn = lambda d: sum((t ** 2).sum() for t in d.values()).sqrt().item()      # noqa: E731
    assert abs(tn(gh) - tn(go)) <= 2e-3 * tn(go), (tn(gh), tn(go))


def test_allocator_pool_stays_bounded_over_steps_fullsize(tmp_path):
    """bench shape, 140 training steps without a host sync: the multi-stream step (round 6: four streams -- the two encoders, the transformer half of every ViT stage,
    the weight gradients) must not make the caching allocator grow step after step.  Gradients that cross streams inside the backward pass are record_stream()ed by
    autograd itself and return to the pool a little later, so the pool needs ~60-80 steps to reach its plateau now (tools/memgrow.py, no syncs: 15 -> 43.3 GB reserved
    at step 80, 43.3 at step 180, 44.6 at step 400: five more segments in 320 steps, against 3.5 GB per STEP when every activation was record_stream()ed in round 2):
    compared are steps 100 and 139."""
    import argparse
    from tcct_amd.nets import stc_tt, RegNet
    from tcct_amd.kite import KiteSeg
    from tcct_amd.data import SynthOCT

    ds = SynthOCT(height=800, width=1100, device='cuda')
    net = RegNet(stc_tt(ds.out_channels, compute_dtype=torch.bfloat16), con='cos', out_channels=ds.out_channels).cuda()
    args = argparse.Namespace(los='di', lr=1e-3, gpu='0', pl=False, bs=B, coff_ds=1, udh=False, reg=False, epl=False, coff_udh=1, coff_reg=.1,
                              coff_epl=.1, bug=True)
    k = KiteSeg(model=net, dataset=ds, root=str(tmp_path), args=args)
    k.model.train()
    img, lab, _, _ = ds.parse(ds.make_batch(B, seed=5))
    stats = []
    for it in range(140):
        loss = k.train_step(img, lab)
        if it in (100, 139):
            ms = torch.cuda.memory_stats()
            stats.append((ms['num_device_alloc'], ms['reserved_bytes.all.current']))
    assert torch.isfinite(loss).item()
    (seg0, res0), (seg1, res1) = stats
    assert seg1 - seg0 <= 12 and res1 <= res0 * 1.2, stats
    del k, net
    torch.cuda.empty_cache()


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_factor_attention_identities_fullsize(dt):
    """factorised attention (reference nets/tcct.py:311-331) at the size of MPViT stage 0 of the bench batch (8 x 400 x 552 tokens, C = 64, 8
    heads), through identities that need no oracle:
      * softmax over the tokens sums to 1 per channel: with v == 1 and zero crpe weights M is all ones, so out[n, h, :] = scale * sum_k q[n, h, k]
      * with a crpe bias of 1 (zero taps) the second term adds q itself
      * the output gradient w.r.t. v of sum(out) obeys sum_n dv[n, h, v] = scale * sum_n sum_k q[n, h, k] (columns of P sum to 1)
      * linearity in q: out(q1 + q2) = out(q1) + out(q2)"""
    from tcct_amd import ops
    Bt, Ht, Wt, C, heads = 8, 400, 552, 64, 8
    N, Ch, scale = Ht * Wt, 8, 8 ** -0.5
    g = torch.Generator(device='cuda').manual_seed(7)
    qkv = torch.randn((Bt, N, 3 * C), device='cuda', generator=g).to(dt)
    qkv[..., 2 * C:] = 1.0
    convs = []
    for k, split in ((3, 2), (5, 3), (7, 3)):
        m = torch.nn.Conv2d(split * Ch, split * Ch, k, padding=k // 2, groups=split * Ch).cuda()
        m.weight.data.zero_()
        m.bias.data.zero_()
        convs.append(m)
    tolr = 2e-3 if dt == torch.float32 else 2e-2
    q = qkv[..., :C].float().view(Bt, N, heads, Ch)
    want = (scale * q.sum(-1, keepdim=True)).expand(Bt, N, heads, Ch).reshape(Bt, N, C)
    qkv.requires_grad_(True)
    out = ops.factor_att(qkv, (Ht, Wt), heads, scale, convs)
    assert (out.float() - want).abs().max().item() < tolr * max(1.0, want.abs().max().item())
    out.float().sum().backward()
    dv = qkv.grad[..., 2 * C:].float().view(Bt, N, heads, Ch).sum(1)                    # [B, heads, Ch]
    want_dv = (scale * q.sum((1, 3))).unsqueeze(-1).expand(Bt, heads, Ch)
    assert (dv - want_dv).abs().max().item() < 2e-2 * max(1.0, want_dv.abs().max().item())
    with torch.no_grad():
        for m in convs:
            m.bias.data.fill_(1.0)
        out1 = ops.factor_att(qkv.detach(), (Ht, Wt), heads, scale, convs)
        assert (out1.float() - (want + qkv.detach()[..., :C].float())).abs().max().item() < tolr * max(1.0, want.abs().max().item())
        if dt == torch.float32:
            a = qkv.detach().clone()
            b2 = qkv.detach().clone()
            b2[..., :C] = torch.randn((Bt, N, C), device='cuda', generator=g)
            s = a.clone()
            s[..., :C] = a[..., :C] + b2[..., :C]
            oa, ob, os_ = (ops.factor_att(t, (Ht, Wt), heads, scale, convs) for t in (a, b2, s))
            assert (oa + ob - os_).abs().max().item() < 1e-3 * max(1.0, os_.abs().max().item())


# ---- round 6: the round-5 kernels at the bench size, inside pytest (they were reached at 8 x 800 x 1104 only by property tests and tools/stress_fs.py)
CENSUS = {'chain33': 0, 'wgradk_stream': 1, 'wgrad33_stream': 2, 'fwd33_stream': 3}


def _census(reset=False):
    from tcct_amd._lib import lib
    return {k: int(lib.kernel_census(i, 1 if reset else 0)) for k, i in CENSUS.items()}


@pytest.mark.parametrize('mode', ['plain', 'stats', 'res'])
def test_conv3x3_chain_is_bit_identical_to_two_launches_at_bench_size(mode):
    """k_conv32_chain33<0|2|3> at 8 x 800 x 1104 (the level-0 shape of the bench step: 37 strips of 30 pixels, 19 strip pairs, runs of 24+ rows) against
    tcct_conv32_fwd twice: intermediate and output bit for bit, one run, nothing else on the GPU (tools/stress_fs.py repeats it beside a second stream)"""
    from tcct_amd._lib import lib
    x, res = _x(seed=20), _x(seed=21)
    g = torch.Generator(device='cuda').manual_seed(22)
    packs = []
    for _ in range(2):
        w = torch.randn((32, 32, 3, 3), device='cuda', generator=g) / 17
        wp = torch.empty(9 * 1024, device='cuda', dtype=torch.bfloat16)
        lib.conv32_pack_weights(w, wp, 3, 3, 0)
        packs.append((wp, torch.randn(32, device='cuda', generator=g)))
    s_a, s_b = (torch.zeros(64, device='cuda', dtype=torch.float64) for _ in range(2))
    mid_a, y_a = torch.empty_like(x), torch.empty_like(x)
    mid_b, y_b = torch.full_like(x, 777.0), torch.full_like(x, 555.0)
    prev = lib.conv32_fwd_mode(1)              # the comparison arm on the TILED kernel: a different load / store structure, the same MFMA order
    try:
        lib.conv32_fwd(x, packs[0][0], packs[0][1], mid_a, B, H, W, 3, 3, 1, 1)
        if mode == 'stats':
            lib.conv32_fwd_bnstats(mid_a, packs[1][0], packs[1][1], y_a, B, H, W, 3, 3, 1, 1, s_a, 1)
        elif mode == 'res':
            lib.conv32_fwd_add(mid_a, packs[1][0], packs[1][1], res, y_a, B, H, W, 3, 3, 1, 1)
        else:
            lib.conv32_fwd(mid_a, packs[1][0], packs[1][1], y_a, B, H, W, 3, 3, 1, 1)
    finally:
        lib.conv32_fwd_mode(prev)
    _census(reset=True)
    lib.conv32_chain33(x, packs[0][0], packs[0][1], mid_b, packs[1][0], packs[1][1], y_b, res if mode == 'res' else None, B, H, W, s_b if mode == 'stats' else None)
    torch.cuda.synchronize()
    assert _census()['chain33'] == 1
    assert int((mid_a != mid_b).any(dim=3).sum()) == 0 and int((y_a != y_b).any(dim=3).sum()) == 0
    if mode == 'stats':
        torch.testing.assert_close(s_b, s_a, rtol=1e-5, atol=1e-5 * s_a.abs().max().item())


@pytest.mark.parametrize('k', [(1, 13), (13, 1)])
def test_cross_conv_weight_gradient_row_streams_at_bench_size(k):
    """k_conv32_wgradk_stream<13, VERT> at 8 x 800 x 1104 -- the shape the entry point routes to it by default (asserted through the launch census) -- against the
    shifted-line kernel (mode 4) and the generic register-staged one (mode 1): the same products summed in another order"""
    from tcct_amd._lib import lib
    kh, kw = k
    x, dy = _x(seed=23), _x(seed=24)
    outs = {}
    prev = lib.conv32_wgrad_mode(-1)
    try:
        for m in (0, 4, 1):
            lib.conv32_wgrad_mode(m)
            dw = torch.full((32, 32, kh, kw), 7.0, device='cuda')
            db = torch.full((32,), 7.0, device='cuda')
            _census(reset=True)
            lib.conv32_wgrad(x, dy, dw, db, B, H, W, kh, kw, kh // 2, kw // 2)
            torch.cuda.synchronize()
            assert _census()['wgradk_stream'] == (1 if m == 0 else 0), m
            outs[m] = (dw, db)
    finally:
        lib.conv32_wgrad_mode(prev)
    sc = outs[1][0].abs().max().item()
    for m in (0, 4):
        assert (outs[m][0] - outs[1][0]).abs().max().item() <= 1e-4 * sc, m
        torch.testing.assert_close(outs[m][1], outs[1][1], rtol=1e-4, atol=1e-4 * outs[1][1].abs().max().item())


def test_fullsize_bf16_train_step_matches_the_rounding_point_oracle(tmp_path):
    """The BENCHMARKED path -- bf16, gradients enabled, the pooled step of KiteSeg.train_step (pack-all launch, gradient slots in the flat buffer, weight gradients on
    the side stream) -- at the bench RESOLUTION (bs 2 at 3 x 800 x 1104: 1.77 M pixels per level-0 tensor, so ops.conv3x3_chain_ok is true at level 0 and the
    13-tap weight gradients take the row streams), full loss (Dice + reg + fpl), seeded default weights, against the oracle's rounding-point mode
    (`tcct_oracle.rounding_points('bf16')`: fp32 arithmetic, bf16 where the HIP path stores) forward AND backward on the CPU.  The launch census proves which
    kernels served the step.  Bounds for heads and loss parts as in tests/test_model_gpu.py::test_bf16_matches_rounding_point_oracle; named gradient tensors
    spanning CNN L0-L4, the ViT stages and the decoder: direction (cosine >= 0.99; 0.985 for the CNN encoder, measured >= 0.9922) and norm against the oracle's backward."""
    import argparse
    import contextlib
    import os
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
    import tcct_oracle as O
    from tcct_amd import ops
    from tcct_amd.nets import stc_tt, RegNet
    from tcct_amd.kite import KiteSeg
    Hh, Ww = 800, 1104
    torch.manual_seed(0)
    sd0 = {k: v.clone() for k, v in RegNet(stc_tt(5), con='cos', out_channels=5).state_dict().items()}
    img3, lab = O.synth_batch(2, Hh, Ww, seed=79)
    g = torch.Generator().manual_seed(11)
    noise = (torch.rand(2, 4, Hh, Ww, generator=g), torch.rand(2, 4, Hh, Ww, generator=g), torch.rand(1, 1, Hh, 1, generator=g), torch.rand(1, 1, Hh, 1, generator=g))
    model = RegNet(stc_tt(5, compute_dtype=torch.bfloat16), con='cos', out_channels=5)
    model.load_state_dict(sd0)

    class DS:
        out_channels = 5
    args = argparse.Namespace(los='di', lr=0.0, gpu='0', pl=False, bs=2, coff_ds=1, udh=True, reg=True, epl=False, coff_udh=1, coff_reg=.1, coff_epl=.1, bug=True)
    k = KiteSeg(model=model, dataset=DS(), root=str(tmp_path), args=args)
    model.train()
    model.base.base_vit.drop_probs = [0.0] * 4
    for grp in k.optimG.param_groups:
        grp['lr'] = 0.0                     # the first step only lays out the flat gradient buffer (its AdamW update is exactly zero): weights stay sd0
    real_reg = model.regular_reg
    model.regular_reg = lambda o, l: real_reg(o, l, noise=noise)       # the recorded draws instead of rand_like (reference nets/reg.py:120,147-148)
    img, labd = img3[:, :1].cuda(), lab.cuda()
    k.train_step(img, labd)
    w_now = {n: p.detach().float().cpu() for n, p in model.named_parameters()}
    assert all(torch.equal(w_now[n], sd0[n]) for n in w_now), 'lr 0 must leave the weights untouched'
    # the measured step: exactly KiteSeg.train_step without the optimizer update
    _census(reset=True)
    k.optimG.zero_grad(set_to_none=True)
    ops.begin_step(k.device)
    try:
        tot, log = k.calc_loss(img, labd)
        tot.backward()
    finally:
        ops.end_step()
    torch.cuda.synchronize()
    cen = _census()
    print('launch census of the step:', cen)
    assert cen['chain33'] >= 2, cen                 # block12 of level 0 forward + its input-gradient chain
    assert cen['wgradk_stream'] >= 2, cen           # 1 x 13 and 13 x 1 weight gradients of level 0
    assert cen['wgrad33_stream'] >= 4 and cen['fwd33_stream'] >= 4, cen
    parts = {kv.split('=')[0]: float(kv.split('=')[1]) for kv in log.split(',')}
    gh = {}
    for n, p in model.named_parameters():
        gsrc = p.grad if p.grad is not None else getattr(p, '_grad_slot', None)
        if gsrc is not None:
            gh[n] = gsrc.detach().double().cpu().reshape(p.shape)
    edge = model.edge_pred.float().cpu().reshape(-1)
    # heads of the same weights, same mode (train-mode BatchNorm), no_grad: the forward kernels are the training step's except for the chain's `mid` store
    with torch.no_grad():
        heads = [o.float().cpu() for o in model(img)]
    # ---- oracle: fp32 (the error model's reference point, forward only) and rounding-point mode (forward + backward)
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    oh = torch.nn.functional.one_hot(lab, 5).permute(0, 3, 1, 2)
    with torch.no_grad():
        _, p32, o32, _ = O.total_loss({kk: v.clone() for kk, v in sd0.items()}, img3, oh, udh=True, reg=True, noise=noise)
    sd = {kk: v.clone() for kk, v in sd0.items()}
    for n, v in sd.items():
        if v.is_floating_point() and not n.endswith(('running_mean', 'running_var')) and not n.startswith('fcp.'):
            v.requires_grad_(True)
    want = {}
    with O.rounding_points('bf16'):
        tb, pb, ob, _ = O.total_loss(sd, img3, oh, udh=True, reg=True, noise=noise, want=want)
        tb.backward()
    go = {n: v.grad.double() for n, v in sd.items() if getattr(v, 'grad', None) is not None}
    # the fp32 oracle's backward: how far the rounding model ITSELF moves a gradient (one realisation of bf16 storage noise against none)
    sd32 = {kk: v.clone() for kk, v in sd0.items()}
    for n, v in sd32.items():
        if v.is_floating_point() and not n.endswith(('running_mean', 'running_var')) and not n.startswith('fcp.'):
            v.requires_grad_(True)
    t32, _, _, _ = O.total_loss(sd32, img3, oh, udh=True, reg=True, noise=noise)
    t32.backward()
    g32 = {n: v.grad.double() for n, v in sd32.items() if getattr(v, 'grad', None) is not None}

    def rel(a, b):
        return ((a.double() - b.double()).abs().max() / max(1e-30, b.double().abs().max().item())).item()
    rep = [(i, rel(ob[i].detach(), o32[i]), rel(heads[i], ob[i].detach()), rel(heads[i], o32[i])) for i in range(4)]
    print('full-size bf16 train step: (head, model error, HIP vs model, HIP vs fp32)', rep)
    for i, model_err, d_model, d_fp32 in rep:
        assert d_model <= 0.6 * model_err and d_fp32 <= 1.3 * model_err + 1e-3, rep
    got = {'dice': parts['los'], 'udh': parts['udh'], 'reg': parts['reg']}
    print('loss parts: HIP', got, 'rounding oracle', {a: b.item() for a, b in pb.items()}, 'fp32 oracle', {a: b.item() for a, b in p32.items()})
    for n in got:
        assert abs(got[n] - pb[n].item()) <= 1e-3 * max(1.0, abs(pb[n].item())), (n, got[n], pb[n].item())
    assert abs(tot.item() - tb.item()) <= 2e-4 * abs(tb.item()), (tot.item(), tb.item())
    eb = want['edge_pred'].detach().reshape(-1)
    assert rel(edge, eb) <= 1e-3
    assert set(gh) == set(go)
    named = ['base.base_cnn.cnn.0.weight', 'base.base_cnn.path_estan.0.block12.0.weight', 'base.base_cnn.path_estan.0.block12.1.weight',
             'base.base_cnn.path_estan.0.block34.0.weight', 'base.base_cnn.path_estan.0.block34.1.weight', 'base.base_cnn.path_estan.0.block34.2.weight',
             'base.base_cnn.path_estan.0.block5.0.weight', 'base.base_cnn.path_estan.1.block12.0.weight', 'base.base_cnn.path_estan.1.block34.0.weight',
             'base.base_cnn.path_estan.1.block34.1.weight', 'base.base_cnn.path_estan.2.block34.2.weight', 'base.base_cnn.path_estan.3.block5.0.weight',
             'base.base_cnn.path_estan.4.block12.0.weight', 'base.base_vit.stem.0.conv.weight', 'base.base_vit.stem.1.conv.weight',
             'base.base_vit.patch_embed_stages.0.patch_embeds.0.patch_conv.dwconv.weight', 'base.base_vit.mhca_stages.0.InvRes.conv1.conv.weight',
             'base.base_vit.mhca_stages.0.aggregate.conv.weight', 'base.base_vit.mhca_stages.1.mhca_blks.0.MHCA_layers.0.mlp.fc1.weight',
             'base.base_vit.mhca_stages.2.InvRes.dwconv.weight', 'base.base_vit.mhca_stages.3.mhca_blks.0.cpe.proj.weight',
             'base.tran_vit0.0.weight', 'base.tran_cnn3.0.weight', 'base.head.0.weight', 'base.dec1.prep.0.weight', 'base.dec4.post.0.weight',
             'base.t324.weight', 'base.aux0.weight', 'base.aux4.weight']
    cos = {n: (gh[n] * go[n]).sum().item() / max(gh[n].norm().item() * go[n].norm().item(), 1e-300) for n in named}
    nrm = {n: gh[n].norm().item() / max(go[n].norm().item(), 1e-300) for n in named}
    allcos = np.array([(gh[n] * go[n]).sum().item() / max(gh[n].norm().item() * go[n].norm().item(), 1e-300) for n in go if go[n].norm().item() > 1e-3 * max(v.norm().item() for v in go.values())])
    print('named gradients: cosine min %.4f (%s), norm ratio %.4f .. %.4f; all tensors above noise (%d): cosine median %.4f min %.4f' % (
        min(cos.values()), min(cos, key=cos.get), min(nrm.values()), max(nrm.values()), len(allcos), float(np.median(allcos)), float(allcos.min())))
    print({n.replace('base.', ''): (round(cos[n], 4), round(nrm[n], 4)) for n in named})
    dev_model = {n: abs(go[n].norm().item() / max(g32[n].norm().item(), 1e-300) - 1.0) for n in named}
    print('norm deviation of the rounding model from fp32 (named):', {n.replace('base.', ''): round(v, 4) for n, v in dev_model.items()})
    assert len(named) >= 20
    for n in named:
        assert cos[n] >= (0.985 if n.startswith('base.base_cnn') else 0.99), (n, cos[n])       # measured: CNN 0.9922-0.9997 (lowest: level 4's first 3x3), everything else >= 0.9993
        # norm within 3 % of the rounding oracle's; 5 % for the CNN encoder's levels 0-1.  HIP and the oracle round at the same points but sum in different orders: two
        # realisations of the same storage noise, which the junction's 1 / sigma of a BatchNorm amplifies for a whole branch at once (block34's three weights move
        # together).  tools/grad_bias_probe.py (profiles/r06_grad_noise_probe.txt) at 2 x 256 x 256: the rounding model ITSELF moves these norms by -15 ... +9 % against
        # fp32, HIP by -10 ... +5 %; the spread shrinks with 1 / sqrt(pixels) to the 2.3-3.2 % measured here over three runs (everything outside the CNN's levels 0-1: <= 1 %)
        wide = n.startswith(('base.base_cnn.cnn.0', 'base.base_cnn.path_estan.0', 'base.base_cnn.path_estan.1'))
        assert abs(nrm[n] - 1.0) <= (0.05 if wide else 0.03), (n, nrm[n], dev_model[n])
    tn = lambda d: sum((t ** 2).sum() for t in d.values()).sqrt().item()      # noqa: E731
    assert abs(tn(gh) - tn(go)) <= 3e-2 * tn(go), (tn(gh), tn(go))
