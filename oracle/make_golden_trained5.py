"""TEST INFRASTRUCTURE (runs only in the build container, needs /root/reference): WELL-CONDITIONED 5-class train-mode fixtures for
BASELINE cfg1 / cfg3 / cfg4 (`--los=di`, `+reg`, `+reg+fpl`) at the fixture size 2x64x64.

The formula-weight fixtures of make_golden.py are ill-conditioned in train mode (batch variances at fp32 rounding level at the 4x4 /
8x8 levels), and the reference's only trained current-layout checkpoint has 9 classes (make_golden_duke_train.py).  This script
TRAINS THE REAL REFERENCE -- `RegNet(stc_tt(5))`, default initialisation, the reference's own KiteSeg.calc_loss (Dice deep
supervision + udh + reg), backward, clip_grad_norm_(12), its own AdamW (kite/loop_seg.py:121-130,146-171) -- on the CPU for
N_TRAIN steps on a pool of synthetic 2x64x64 batches, rounds the trained state to bf16 (stored as bit patterns in
tests/golden/ckpt_trained5.npz, same layout as ckpt_duke.npz) and then runs ONE recorded train step of the real reference from that
state per loss configuration on a held-out batch:

    di_trained_2x64x64.npz    udh off, reg off     (BASELINE cfg1)
    reg_trained_2x64x64.npz   udh off, reg on      (BASELINE cfg3)
    full_trained_2x64x64.npz  udh on,  reg on      (BASELINE cfg4)

    cd /tmp/scratch && python /root/repo/oracle/make_golden_trained5.py

Stored (data only): input, labels, forced DropPath masks, the four rand_like draws, all four heads (aux heads subsampled), feats, loss
parts, boundary coordinates, per-tensor gradient norms of all trained tensors, ~35 full gradients, post-step deltas, BatchNorm
buffers, Dice / IoU scores of the train-mode mask.  The oracle is asserted equal to the reference on the way, and the conditioning of
the case is measured and stored (`cond_*`: distance of the reference's fp32 result from an fp64 evaluation of the same graph)."""
import argparse, contextlib, io, os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport
import tcct_oracle as O

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
N_TRAIN = 300
N_CLASS = 5
FULL = ['base.aux0.weight', 'base.aux1.weight', 'base.t324.weight', 'base.t321.bias', 'base.dec4.prep.0.weight', 'base.dec4.prep.1.weight',
        'base.dec1.post.0.weight', 'base.head.0.weight', 'base.tran_vit0.0.weight', 'base.tran_cnn3.1.bias',
        'base.base_cnn.cnn.0.weight', 'base.base_cnn.cnn.1.weight',
        'base.base_cnn.path_estan.0.block12.0.weight', 'base.base_cnn.path_estan.0.block34.0.weight', 'base.base_cnn.path_estan.0.block34.1.weight',
        'base.base_cnn.path_estan.0.block5.2.weight', 'base.base_cnn.path_estan.1.block12.1.weight', 'base.base_cnn.path_estan.2.block34.2.weight',
        'base.base_cnn.path_estan.3.block5.0.weight', 'base.base_cnn.path_estan.4.block12.3.bias',
        'base.base_vit.stem.0.conv.weight', 'base.base_vit.stem.1.bn.weight',
        'base.base_vit.patch_embed_stages.0.patch_embeds.0.patch_conv.dwconv.weight',
        'base.base_vit.mhca_stages.0.InvRes.conv1.conv.weight', 'base.base_vit.mhca_stages.0.InvRes.norm.weight',
        'base.base_vit.mhca_stages.0.aggregate.conv.weight', 'base.base_vit.mhca_stages.1.mhca_blks.0.MHCA_layers.0.mlp.fc1.weight',
        'base.base_vit.mhca_stages.1.InvRes.dwconv.weight', 'base.base_vit.mhca_stages.2.mhca_blks.0.MHCA_layers.0.norm1.weight',
        'base.base_vit.mhca_stages.2.InvRes.conv2.bn.weight', 'base.base_vit.mhca_stages.3.mhca_blks.0.cpe.proj.weight',
        'base.base_vit.mhca_stages.3.aggregate.bn.bias', 'lap_reg.0.weight', 'lap_map.1.weight', 'lap_map.2.bias']


class DS:
    out_channels = N_CLASS


def kite(KiteSeg, model, udh, reg):
    args = argparse.Namespace(los='di', lr=1e-2, gpu='0', pl=False, bs=2, coff_ds=1, udh=udh, reg=reg, epl=False, coff_udh=1, coff_reg=.1,
                              coff_epl=.1, bug=True)
    with contextlib.redirect_stdout(io.StringIO()):
        return KiteSeg(model=model, dataset=DS(), root='', args=args)


def train(nets, KiteSeg, setup_seed):
    """the reference trains itself; returns its state_dict rounded to bf16"""
    setup_seed(5)
    with contextlib.redirect_stdout(io.StringIO()):
        model = nets.RegNet(nets.stc_tt(N_CLASS), con='cos', out_channels=N_CLASS)
    k = kite(KiteSeg, model, True, True)
    for g in k.optimG.param_groups:
        g['lr'] = 2e-3
    k.model.train()
    pool = [O.synth_batch(2, 64, 64, seed=100 + i) for i in range(24)]
    for it in range(N_TRAIN):
        img, lab = pool[it % len(pool)]
        onehot = torch.nn.functional.one_hot(lab, N_CLASS).permute(0, 3, 1, 2)
        k.optimG.zero_grad()
        loss, log = k.calc_loss(img, onehot)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(k.model.parameters(), 12)
        k.optimG.step()
        if it % 25 == 0 or it == N_TRAIN - 1:
            print(f'  train step {it:4d}: loss {loss.item():.4f} ({log})', flush=True)
    sd = {}
    for kk, v in k.model.state_dict().items():
        sd[kk] = v.detach().to(torch.bfloat16).to(torch.float32) if v.is_floating_point() else v.detach().clone()
    return sd


def one_case(name, nets, KiteSeg, setup_seed, sd_round, udh, reg, seed):
    with contextlib.redirect_stdout(io.StringIO()):
        model = nets.RegNet(nets.stc_tt(N_CLASS), con='cos', out_channels=N_CLASS)
    model.load_state_dict(sd_round, strict=True)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    img, lab = O.synth_batch(2, 64, 64, seed=seed)
    onehot = torch.nn.functional.one_hot(lab, N_CLASS).permute(0, 3, 1, 2)
    k = kite(KiteSeg, model, udh, reg)
    k.model.train()
    setup_seed(seed)
    dp_masks = [torch.tensor(m, dtype=torch.float32) for m in ([1, 1], [1, 0], [1, 1], [0, 1], [1, 1], [1, 0])]
    _refimport.DropPath.forced = [m.clone() for m in dp_masks]
    draws, real = [], torch.rand_like

    def rec(t, **kw):
        r = real(t, **kw)
        draws.append(r.clone())
        return r
    torch.rand_like = rec
    try:
        k.optimG.zero_grad()
        loss, log = k.calc_loss(img, onehot)
    finally:
        torch.rand_like = real
    assert len(draws) == (4 if reg else 0)
    loss.backward()
    named = dict(k.model.named_parameters())
    grads = {n: p.grad.detach().clone() for n, p in named.items() if p.grad is not None}
    before = {n: named[n].detach().clone() for n in grads}
    lr = k.optimG.param_groups[0]['lr']
    gnorm = torch.nn.utils.clip_grad_norm_(k.model.parameters(), 12)
    k.optimG.step()
    after = {n: named[n].detach().clone() for n in grads}
    sd_after = {kk: v.detach().clone() for kk, v in k.model.state_dict().items()}

    # ---- oracle == reference on the same inputs, in fp32; and the fp64 evaluation of the same graph (conditioning of the case)
    def oracle(dt):
        sd = {kk: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for kk, v in sd0.items()}
        for kk in sd:
            if kk in named and named[kk].requires_grad:
                sd[kk].requires_grad_(True)
        want = {}
        noise = tuple(d.to(dt) for d in draws) if reg else None
        tot, parts, outs, feats = O.total_loss(sd, img.to(dt), onehot, udh=udh, reg=reg, dp_masks=[m.clone().to(dt) for m in dp_masks], noise=noise, want=want)
        tot.backward()
        return sd, tot, parts, outs, feats, want
    sd, tot, parts, outs, feats, want = oracle(torch.float32)
    assert abs(tot.item() - loss.item()) < 1e-5 * max(1.0, abs(loss.item())), (tot.item(), loss.item())
    gmax = max(x.abs().max().item() for x in grads.values())
    worst = 0.0
    for n, g in grads.items():
        if g.abs().max().item() < 1e-4 * gmax:
            continue
        e = (sd[n].grad.double() - g.double()).norm().item() / g.double().norm().item()
        worst = max(worst, e)
        assert e < 1e-3, (n, e)
    sd64, tot64, parts64, outs64, feats64, want64 = oracle(torch.float64)
    cond_out = [((a.double() - b).abs().max() / max(1.0, b.abs().max().item())).item() for a, b in zip(outs, outs64)]
    cond_grad = []
    for n, g in grads.items():
        if g.abs().max().item() < 1e-4 * gmax:
            continue
        cond_grad.append((g.double() - sd64[n].grad).norm().item() / sd64[n].grad.norm().item())
    print(f'[{name}] oracle == reference: loss {loss.item():.6f} ({log}); worst gradient rel-L2 {worst:.2e}; |g| {gnorm.item():.5f}; lr {lr:g}; '
          f'{len(grads)} tensors;  conditioning (reference fp32 vs fp64 graph): heads {[f"{c:.1e}" for c in cond_out]}, '
          f'gradients median {np.median(cond_grad):.1e} max {max(cond_grad):.1e}', flush=True)

    with torch.no_grad():
        k.model.load_state_dict(sd0, strict=True)
        k.model.train()
        _refimport.DropPath.forced = [m.clone() for m in dp_masks]
        routs = k.model(img)
        rfeats = k.model.base.feats[0]
        _refimport.DropPath.forced = None
    for a, b in zip(routs, outs):
        assert (a - b).abs().max().item() < 1e-4 * max(1.0, a.abs().max().item())
    mask = O.predict_mask(routs[0])
    from kite.losses.miou import MDiceLoss, MIouLoss
    f1 = MDiceLoss.scorem(mask, onehot, start_idx=1)
    iou = MIouLoss.scorem(mask, onehot, start_idx=1)
    sub = (slice(None), slice(None), slice(None, None, 4), slice(None, None, 4))
    names = sorted(grads)
    fx = dict(img=img[:, :1].numpy(), lab=lab.numpy().astype(np.uint8), n_class=np.int64(N_CLASS), flags=np.array([int(udh), int(reg)]),
              dp_masks=torch.stack(dp_masks, 0).numpy().astype(np.uint8),
              out0=routs[0].numpy(), out1=routs[1][sub].numpy(), out2=routs[2][sub].numpy(), out3=routs[3][sub].numpy(), feats=rfeats[sub].numpy(),
              mask0=mask.argmax(1).numpy().astype(np.uint8) if mask.ndim == 4 else mask.numpy().astype(np.uint8),
              dice_scorem=np.float32(f1.item()), iou_scorem=np.float32(iou.item()),
              loss_total=np.float32(loss.item()), loss_dice=np.float32(parts['dice'].item()),
              lr=np.float64(lr), grad_total_norm=np.float32(gnorm.item()), grad_names=np.array(names),
              grad_l2=np.array([grads[n].double().norm().item() for n in names]), grad_max=np.array([grads[n].abs().max().item() for n in names]),
              cond_heads=np.array(cond_out), cond_grad_median=np.float64(np.median(cond_grad)), cond_grad_max=np.float64(max(cond_grad)))
    if udh:
        fx['loss_udh'] = np.float32(parts['udh'].item())
        fx['emb'] = torch.stack(k.model.emb_list, 0).detach().numpy()
    if reg:
        fx['loss_reg'] = np.float32(parts['reg'].item())
        fx['edge_pred'] = want['edge_pred'].detach().numpy()
        fx['edge_true'] = want['edge_true'].detach().numpy()
        for i, d in enumerate(draws):
            fx[f'noise{i}'] = d.numpy()
    for n in FULL:
        if n in grads:
            fx['grad:' + n] = grads[n].numpy()
            fx['step:' + n] = ((after[n].double() - before[n].double()) / lr).numpy()
    for kk in sd_after:
        if kk.endswith(('running_mean', 'running_var', 'num_batches_tracked')) and (kk.startswith(('base.base_cnn.cnn.1', 'base.head.1', 'lap_map.1')) or 'InvRes.norm' in kk):
            fx['buf:' + kk] = sd_after[kk].numpy()
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **fx)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB', flush=True)


def main():
    torch.set_num_threads(8)
    with contextlib.redirect_stdout(io.StringIO()):
        nets, KiteSeg, setup_seed, _ = _refimport.load()
    ck = os.path.join(OUT, 'ckpt_trained5.npz')
    if '--reuse' in sys.argv and os.path.exists(ck):
        z = np.load(ck)
        sd = {}
        for k in z.files:
            if k.startswith('w::'):
                sd[k[3:]] = torch.from_numpy(z[k].view(np.int16).copy()).view(torch.bfloat16).float()
            elif k.startswith('i::'):
                sd[k[3:]] = torch.from_numpy(np.asarray(z[k]).copy())
    else:
        sd = train(nets, KiteSeg, setup_seed)
        arrays = {('w::' if v.is_floating_point() else 'i::') + k: (v.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16) if v.is_floating_point() else v.numpy())
                  for k, v in sd.items()}
        arrays['n_class'] = np.int64(N_CLASS)
        arrays['n_train_steps'] = np.int64(N_TRAIN)
        np.savez_compressed(ck, **arrays)
        print('wrote', ck, os.path.getsize(ck) // 1024, 'KiB')
    one_case('di_trained_2x64x64', nets, KiteSeg, setup_seed, sd, False, False, 3001)
    one_case('reg_trained_2x64x64', nets, KiteSeg, setup_seed, sd, False, True, 3002)
    one_case('full_trained_2x64x64', nets, KiteSeg, setup_seed, sd, True, True, 3003)


if __name__ == '__main__':
    main()
