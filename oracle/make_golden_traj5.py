"""TEST INFRASTRUCTURE (runs only in the build container, needs /root/reference): FIVE CONSECUTIVE train steps of the real reference per loss
configuration, so that what one step cannot show is pinned to the reference's own loop (kite/loop_seg.py:108-142, kite/loopback.py:102-128):
Adam's first / second moments and their bias correction at t >= 2, the weights after several updates, BatchNorm running statistics and
`num_batches_tracked` over several steps (lap_map's BatchNorm runs twice per step), and the CyclicLR scheduler driving the optimizer's lr.

Start state: tests/golden/ckpt_trained5.npz (the reference trained by itself, make_golden_trained5.py).  Per configuration the REAL reference
`RegNet(stc_tt(5))` + its own KiteSeg.calc_loss / backward / clip_grad_norm_(12) / AdamW(wd=2e-4) runs 5 steps on 5 different synthetic
2x64x64 batches with forced DropPath masks and recorded `rand_like` draws:
    steps 1-3 at lr = 1e-3 (set by hand: large enough that the weights move and later steps see it; the update of step 1 alone is lr * sign(g)),
    then `schedG.step()` (the reference steps CyclicLR(1e-6 .. 1e-4, 4 up / 60 down) once per epoch, kite/loop_seg.py:61 -> lr = 2.575e-5),
    steps 4-5 at the scheduler's lr.

    traj5_di.npz    udh off, reg off   (BASELINE cfg1 / cfg2)
    traj5_reg.npz   udh off, reg on    (cfg3)
    traj5_full.npz  udh on,  reg on    (cfg4)

    cd /tmp/scratch && python /root/repo/oracle/make_golden_traj5.py

Stored (data only): the five inputs / label maps / mask sets / noise draws; per step the loss parts, total loss, pre-clip total gradient norm, lr;
after step 5: ~35 full weight tensors with their Adam moments, the L2 norm of every trained tensor and of its displacement from the start,
every BatchNorm running mean / variance / num_batches_tracked.  The oracle (tcct_oracle + its own clip_adamw_step) runs the same five steps
FREELY (its own weights and moments, never re-synchronised) and must stay on the reference's trajectory -- asserted here, and again by
tests/test_oracle_golden.py."""
import argparse, contextlib, io, os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport
import tcct_oracle as O
from make_golden_trained5 import FULL, DS, N_CLASS, OUT

N_STEPS = 5
LR_HAND = 1e-3
SCHED_AFTER = 3         # schedG.step() after this many steps
DP_PATTERNS = ([1, 1], [1, 0], [0, 1], [1, 1], [1, 0], [0, 1], [1, 1])


def dp_masks_for(step):
    """six DropPath draws per forward (two per stage with p > 0, in call order): a fixed pattern that changes with the step"""
    return [torch.tensor(DP_PATTERNS[(step + i) % len(DP_PATTERNS)], dtype=torch.float32) for i in range(6)]


def load_ckpt():
    z = np.load(os.path.join(OUT, 'ckpt_trained5.npz'))
    sd = {}
    for k in z.files:
        if k.startswith('w::'):
            sd[k[3:]] = torch.from_numpy(z[k].view(np.int16).copy()).view(torch.bfloat16).float()
        elif k.startswith('i::'):
            sd[k[3:]] = torch.from_numpy(np.asarray(z[k]).copy())
    return sd


def oracle_trajectory(sd0, trained, inputs, udh, reg, lrs):
    """the oracle's own five steps: (per-step totals, per-step parts, per-step total norms, final state dict, moments)"""
    sd = {k: v.clone() for k, v in sd0.items()}
    m = {n: torch.zeros_like(sd[n]) for n in trained}
    v = {n: torch.zeros_like(sd[n]) for n in trained}
    totals, parts_all, norms = [], [], []
    for t, (img, onehot, masks, noise) in enumerate(inputs):
        for n in trained:
            sd[n] = sd[n].detach().requires_grad_(True)
        tot, parts, _, _ = O.total_loss(sd, img, onehot, udh=udh, reg=reg, dp_masks=[x.clone() for x in masks], noise=noise)
        tot.backward()
        ps = [sd[n] for n in trained]
        gs = [sd[n].grad for n in trained]
        with torch.no_grad():
            total = O.clip_adamw_step([p.data for p in ps], gs, [m[n] for n in trained], [v[n] for n in trained], t + 1, lrs[t])
        for n in trained:
            sd[n] = sd[n].detach()
        totals.append(tot.item())
        parts_all.append({k: x.item() for k, x in parts.items()})
        norms.append(total.item())
    return totals, parts_all, norms, sd, m, v


def one_traj(name, nets, KiteSeg, setup_seed, sd_round, udh, reg, seed0):
    with contextlib.redirect_stdout(io.StringIO()):
        model = nets.RegNet(nets.stc_tt(N_CLASS), con='cos', out_channels=N_CLASS)
    model.load_state_dict(sd_round, strict=True)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    args = argparse.Namespace(los='di', lr=1e-2, gpu='0', pl=False, bs=2, coff_ds=1, udh=udh, reg=reg, epl=False, coff_udh=1, coff_reg=.1,
                              coff_epl=.1, bug=True)
    with contextlib.redirect_stdout(io.StringIO()):
        k = KiteSeg(model=model, dataset=DS(), root='', args=args)
    k.model.train()
    for g in k.optimG.param_groups:
        g['lr'] = LR_HAND
    named = dict(k.model.named_parameters())
    inputs, lrs, totals, norms, gmaxes = [], [], [], [], []
    real = torch.rand_like
    for t in range(N_STEPS):
        img, lab = O.synth_batch(2, 64, 64, seed=seed0 + t)
        onehot = torch.nn.functional.one_hot(lab, N_CLASS).permute(0, 3, 1, 2)
        masks = dp_masks_for(t)
        setup_seed(seed0 + t)
        _refimport.DropPath.forced = [x.clone() for x in masks]
        draws = []

        def rec(x, **kw):
            r = real(x, **kw)
            draws.append(r.clone())
            return r
        torch.rand_like = rec
        try:
            k.optimG.zero_grad()
            loss, _ = k.calc_loss(img, onehot)
        finally:
            torch.rand_like = real
            _refimport.DropPath.forced = None
        assert len(draws) == (4 if reg else 0)
        loss.backward()
        lrs.append(float(k.optimG.param_groups[0]['lr']))
        gmaxes.append({n: p.grad.abs().max().item() for n, p in named.items() if p.grad is not None})
        gnorm = torch.nn.utils.clip_grad_norm_(k.model.parameters(), 12)
        k.optimG.step()
        if t + 1 == SCHED_AFTER:
            k.schedG.step()
        totals.append(loss.item())
        norms.append(gnorm.item())
        inputs.append((img, onehot, masks, tuple(draws) if reg else None))
        if t == 0:
            trained = sorted(n for n, p in named.items() if p.grad is not None)
    sd5 = {kk: v.detach().clone() for kk, v in k.model.state_dict().items()}
    state = k.optimG.state

    # ---- the oracle on its own: same start, same inputs, its own weights and moments for five steps
    o_tot, o_parts, o_norms, o_sd, o_m, o_v = oracle_trajectory(sd0, trained, inputs, udh, reg, lrs)
    # A bias in front of a train-mode BatchNorm has an exactly zero gradient; in fp32 it is rounding noise, which Adam's normalisation turns into
    # +-lr steps of random sign (in the reference as anywhere else).  `signal[n]`: the tensor's gradient stood above that level in every step.
    signal = {n: all(g[n] >= 1e-4 * max(g.values()) for g in gmaxes) for n in trained}
    worst_w, worst_name = 0.0, ''
    for n in trained:
        if not signal[n]:
            continue
        d_ref = (sd5[n].double() - sd0[n].double())
        d_orc = (o_sd[n].double() - sd0[n].double())
        if d_ref.norm().item() == 0:
            continue
        e = (d_orc - d_ref).norm().item() / d_ref.norm().item()
        if e > worst_w:
            worst_w, worst_name = e, n
    print(f'[{name}] reference losses {[f"{x:.6f}" for x in totals]}  norms {[f"{x:.4f}" for x in norms]}  lrs {lrs}')
    print(f'[{name}] oracle    losses {[f"{x:.6f}" for x in o_tot]}  norms {[f"{x:.4f}" for x in o_norms]}')
    print(f'[{name}] oracle vs reference after {N_STEPS} free steps: worst displacement rel-L2 {worst_w:.2e} ({worst_name}) over '
          f'{sum(signal.values())} of {len(trained)} tensors with a gradient above rounding level', flush=True)
    assert worst_w < 5e-2, (worst_w, worst_name)
    for a, b in zip(o_tot, totals):
        assert abs(a - b) <= 1e-3 * max(1.0, abs(b)), (a, b)
    for a, b in zip(o_norms, norms):
        assert abs(a - b) <= 1e-3 * b, (a, b)

    fx = dict(n_class=np.int64(N_CLASS), flags=np.array([int(udh), int(reg)]), n_steps=np.int64(N_STEPS), sched_after=np.int64(SCHED_AFTER),
              img=np.stack([i[0][:, :1].numpy() for i in inputs]), lab=np.stack([i[1].argmax(1).numpy().astype(np.uint8) for i in inputs]),
              dp_masks=np.stack([torch.stack(i[2], 0).numpy().astype(np.uint8) for i in inputs]),
              lr=np.array(lrs, dtype=np.float64), loss_total=np.array(totals, dtype=np.float32), grad_total_norm=np.array(norms, dtype=np.float32),
              loss_dice=np.array([p['dice'] for p in o_parts], dtype=np.float32), names=np.array(trained),
              w_l2=np.array([sd5[n].double().norm().item() for n in trained]),
              disp_l2=np.array([(sd5[n].double() - sd0[n].double()).norm().item() for n in trained]),
              signal=np.array([signal[n] for n in trained]), oracle_loss_total=np.array(o_tot, dtype=np.float32))
    # the parts come from the oracle evaluated AT THE REFERENCE'S STATE of each step (the reference returns only the total and a 4-decimal log string);
    # recomputed below so that they belong to the reference trajectory, not to the oracle's free run
    if udh:
        fx['loss_udh'] = np.zeros(N_STEPS, np.float32)
    if reg:
        fx['loss_reg'] = np.zeros(N_STEPS, np.float32)
        for t, i in enumerate(inputs):
            for j, d in enumerate(i[3]):
                fx[f'noise{t}_{j}'] = d.numpy()
    fx['_parts_pending'] = np.int64(1)
    for n in FULL:
        if n in trained:
            fx['w:' + n] = sd5[n].numpy()
            fx['m:' + n] = state[named[n]]['exp_avg'].detach().numpy()
            fx['v:' + n] = state[named[n]]['exp_avg_sq'].detach().numpy()
    for kk in sd5:
        if kk.endswith(('running_mean', 'running_var', 'num_batches_tracked')):
            fx['buf:' + kk] = sd5[kk].numpy()
    return fx, sd0, trained, inputs, lrs, named, k


def parts_on_reference_trajectory(fx, nets, KiteSeg, sd_round, udh, reg, inputs, lrs, trained):
    """second pass of the REAL reference over the same five steps; before each step its state is handed to the oracle, whose loss parts (asserted to add up
    to the reference's total of that step) are stored"""
    with contextlib.redirect_stdout(io.StringIO()):
        model = nets.RegNet(nets.stc_tt(N_CLASS), con='cos', out_channels=N_CLASS)
    model.load_state_dict(sd_round, strict=True)
    args = argparse.Namespace(los='di', lr=1e-2, gpu='0', pl=False, bs=2, coff_ds=1, udh=udh, reg=reg, epl=False, coff_udh=1, coff_reg=.1,
                              coff_epl=.1, bug=True)
    with contextlib.redirect_stdout(io.StringIO()):
        k = KiteSeg(model=model, dataset=DS(), root='', args=args)
    k.model.train()
    real = torch.rand_like
    for t, (img, onehot, masks, noise) in enumerate(inputs):
        for g in k.optimG.param_groups:
            g['lr'] = lrs[t]
        sd_t = {kk: v.detach().clone() for kk, v in k.model.state_dict().items()}
        with torch.no_grad():
            tot, parts, _, _ = O.total_loss(sd_t, img, onehot, udh=udh, reg=reg, dp_masks=[x.clone() for x in masks], noise=noise)
        assert abs(tot.item() - float(fx['loss_total'][t])) <= 2e-5 * max(1.0, abs(tot.item())), (t, tot.item(), float(fx['loss_total'][t]))
        fx['loss_dice'][t] = parts['dice'].item()
        if udh:
            fx['loss_udh'][t] = parts['udh'].item()
        if reg:
            fx['loss_reg'][t] = parts['reg'].item()
        _refimport.DropPath.forced = [x.clone() for x in masks]
        queue = list(noise) if reg else []
        torch.rand_like = (lambda x, **kw: queue.pop(0).clone()) if reg else real
        try:
            k.optimG.zero_grad()
            loss, _ = k.calc_loss(img, onehot)
        finally:
            torch.rand_like = real
            _refimport.DropPath.forced = None
        assert abs(loss.item() - float(fx['loss_total'][t])) <= 1e-6 * max(1.0, abs(loss.item())), 'the second pass left the recorded trajectory'
        loss.backward()
        torch.nn.utils.clip_grad_norm_(k.model.parameters(), 12)
        k.optimG.step()
    del fx['_parts_pending']


def main():
    torch.set_num_threads(8)
    with contextlib.redirect_stdout(io.StringIO()):
        nets, KiteSeg, setup_seed, _ = _refimport.load()
    sd = load_ckpt()
    for name, udh, reg, seed0 in (('traj5_di', False, False, 4100), ('traj5_reg', False, True, 4200), ('traj5_full', True, True, 4300)):
        fx, sd0, trained, inputs, lrs, named, k = one_traj(name, nets, KiteSeg, setup_seed, sd, udh, reg, seed0)
        parts_on_reference_trajectory(fx, nets, KiteSeg, sd, udh, reg, inputs, lrs, trained)
        path = os.path.join(OUT, name + '.npz')
        np.savez_compressed(path, **fx)
        print('wrote', path, os.path.getsize(path) // 1024, 'KiB', flush=True)


if __name__ == '__main__':
    main()
