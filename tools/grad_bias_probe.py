"""Diagnostic (round 6): per-tensor gradient norm ratio HIP bf16 / rounding-point oracle for the CNN encoder's weights, with switches, at a CPU-cheap size.
usage: python tools/grad_bias_probe.py [H W] [--set NAME=v,...]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import torch
import tcct_oracle as O
from tcct_amd import ops
from tcct_amd.nets import stc_tt, RegNet
from tcct_amd.kite import KiteSeg
p = argparse.ArgumentParser()
p.add_argument('hw', nargs='*', type=int, default=[256, 256])
p.add_argument('--set', default='')
p.add_argument('--bs', type=int, default=2)
a = p.parse_args()
for kv in filter(None, a.set.split(',')):
    k_, v_ = kv.split('=')
    cur_ = getattr(ops, k_)
    setattr(ops, k_, int(v_) if isinstance(cur_, int) and not isinstance(cur_, bool) else bool(int(v_)))
H, W = a.hw
torch.manual_seed(0)
sd0 = {k: v.clone() for k, v in RegNet(stc_tt(5), con='cos', out_channels=5).state_dict().items()}
img3, lab = O.synth_batch(a.bs, H, W, seed=79)
oh = torch.nn.functional.one_hot(lab, 5).permute(0, 3, 1, 2)
res = {}
for mode in ('hip', 'oracle_bf16', 'oracle_fp32'):
    if mode == 'hip':
        model = RegNet(stc_tt(5, compute_dtype=torch.bfloat16), con='cos', out_channels=5)
        model.load_state_dict(sd0)
        class DS: out_channels = 5
        args = argparse.Namespace(los='di', lr=0.0, gpu='0', pl=False, bs=a.bs, coff_ds=1, udh=False, reg=False, epl=False, coff_udh=1, coff_reg=.1, coff_epl=.1, bug=True)
        k = KiteSeg(model=model, dataset=DS(), root='/tmp/gbp', args=args)
        model.train(); model.base.base_vit.drop_probs = [0.0] * 4
        out = model(img3[:, :1].cuda())
        tot = k.grad_calc(out, lab.cuda(), ds=True, criterion=k.criterion)
        tot.backward()
        res[mode] = {n: p_.grad.detach().double().cpu() for n, p_ in model.named_parameters() if p_.grad is not None}
    else:
        import contextlib
        sd = {kk: v.clone() for kk, v in sd0.items()}
        for n, v in sd.items():
            if v.is_floating_point() and not n.endswith(('running_mean', 'running_var')) and not n.startswith('fcp.'):
                v.requires_grad_(True)
        with (O.rounding_points('bf16') if mode == 'oracle_bf16' else contextlib.nullcontext()):
            t, _, _, _ = O.total_loss(sd, img3, oh, udh=False, reg=False)
            t.backward()
        res[mode] = {n: v.grad.double() for n, v in sd.items() if getattr(v, 'grad', None) is not None}
print(f'# {H}x{W} bs {a.bs} --set {a.set or "-"}: |g_hip| / |g_oracle_bf16|, cosine; |g_oracle_bf16| / |g_oracle_fp32|')
for n in sorted(res['hip']):
    if 'base_cnn' in n and n.endswith('weight') and ('path_estan.0' in n or 'path_estan.1' in n or 'cnn.0' in n) and res['hip'][n].dim() == 4:
        gh, gb, g3 = res['hip'][n], res['oracle_bf16'][n], res['oracle_fp32'][n]
        print(f'{n[14:]:34s} {gh.norm() / gb.norm():.4f}  cos {(gh * gb).sum() / gh.norm() / gb.norm():.4f}   model {gb.norm() / g3.norm():.4f}   hip/fp32 {gh.norm() / g3.norm():.4f}')
