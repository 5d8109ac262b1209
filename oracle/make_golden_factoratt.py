"""TEST INFRASTRUCTURE (build container only, needs /root/reference): fixture for the factorised-attention token mixer the reference
defines but keeps commented out in MHCABlock (nets/tcct.py:219-341, 436-449).  The REAL reference classes
`FactorAtt_ConvRelPosEnc` + `ConvRelPosEnc` (8 heads, windows {3:2, 5:3, 7:3}, qkv_bias=True as MHCABlock would pass, tcct.py:424) run
forward and backward on formula inputs / formula weights; inputs, parameters, output and every gradient are committed as data, and the
oracle restatement (tcct_oracle.factor_att) is asserted against them here."""
import os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport
import tcct_oracle as O


def sinfill(name, shape, amp):
    n = int(np.prod(shape))
    i = torch.arange(n, dtype=torch.float64)
    return (amp * torch.sin(0.37 * i + O._crc(name)) * torch.cos(0.011 * i + 1.0)).float().reshape(shape)


def case(ref, dim, B, H, W, tag):
    heads = 8
    crpe = ref.ConvRelPosEnc(Ch=dim // heads, h=heads, window={3: 2, 5: 3, 7: 3})
    att = ref.FactorAtt_ConvRelPosEnc(dim, num_heads=heads, qkv_bias=True, shared_crpe=crpe)
    keys = [(k, tuple(v.shape)) for k, v in att.state_dict().items()]
    att.load_state_dict({k: O.formula_tensor(f'{tag}.{k}', s) for k, s in keys}, strict=True)
    with torch.no_grad():                   # balance the two terms of the mixer (formula weights add up coherently with formula inputs)
        for m in crpe.conv_list:
            m.weight.mul_(0.15)
    x = sinfill(f'{tag}.x', (B, H * W, dim), 1.0).requires_grad_(True)
    gout = sinfill(f'{tag}.gout', (B, H * W, dim), 5.0)
    att.train()
    y = att(x, (H, W))
    y.backward(gout)
    out = {'x': x.detach().numpy(), 'gout': gout.numpy(), 'y': y.detach().numpy(), 'dx': x.grad.numpy(), 'size': np.array([H, W]),
           'heads': np.array(heads)}
    for k, p in att.named_parameters():
        out['p.' + k] = p.detach().numpy()
        out['g.' + k] = p.grad.numpy()
    # the restatement against the real thing
    ps = {k: p.detach().clone().requires_grad_(True) for k, p in att.named_parameters()}
    xo = x.detach().clone().requires_grad_(True)
    wb = [(ps[f'crpe.conv_list.{i}.weight'], ps[f'crpe.conv_list.{i}.bias']) for i in range(3)]
    yo = O.factor_att(xo, ps['qkv.weight'], ps['qkv.bias'], ps['proj.weight'], ps['proj.bias'], wb, (H, W), heads)
    yo.backward(gout)
    assert torch.allclose(yo, y, rtol=1e-5, atol=1e-6), float((yo - y).abs().max())
    assert torch.allclose(xo.grad, x.grad, rtol=1e-5, atol=1e-6)
    for k, p in att.named_parameters():
        assert torch.allclose(ps[k].grad, p.grad, rtol=1e-4, atol=1e-5 * float(p.grad.abs().max())), k
    with torch.no_grad():                   # how much each term of the mixer contributes, and how peaked the softmax over tokens is
        wb0 = [(m.weight, m.bias) for m in crpe.conv_list]
        zero = [(torch.zeros_like(w), torch.zeros_like(b)) for w, b in wb0]
        a_only = O.factor_att_mix(x, att.qkv.weight, att.qkv.bias, zero, (H, W), heads)
        both = O.factor_att_mix(x, att.qkv.weight, att.qkv.bias, wb0, (H, W), heads)
        kk = torch.nn.functional.linear(x, att.qkv.weight, att.qkv.bias)[..., dim:2 * dim]
        print(tag, 'attention term max', float(a_only.abs().max()), 'crpe term max', float((both - a_only).abs().max()),
              'k range', float(kk.min()), float(kk.max()), 'max softmax prob', float(kk.softmax(1).max()))
    print(tag, 'y range', float(y.min()), float(y.max()), '|dx| max', float(x.grad.abs().max()), 'oracle == reference')
    return out


if __name__ == '__main__':
    _refimport.install()
    import nets  # noqa: F401
    ref = sys.modules['nets.tcct']      # (the package attribute `nets.tcct` is the factory alias of the same name)
    allv = {}
    for dim, B, H, W, tag in ((64, 2, 6, 10, 'fa64'), (96, 1, 5, 7, 'fa96')):
        for k, v in case(ref, dim, B, H, W, tag).items():
            allv[f'{tag}.{k}'] = v
    path = os.path.join(HERE, '..', 'tests', 'golden', 'factoratt.npz')
    np.savez_compressed(path, **allv)
    print(path, os.path.getsize(path) // 1024, 'KiB')
