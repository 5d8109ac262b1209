"""FeatConPolar — class prototypes on the unit hypersphere (reference nets/fcp.py:16-75).

With `init=False` (how RegNet builds it, reference nets/reg.py:57) the prototypes are just row-normalised U[0,1) draws
that live in the state_dict (`vec_grad` frozen parameter, `buf_grad`, `cos_dist` buffers).  `choice(pro, i)` of the
reference (row i repeated) is folded into the fused FPL loss kernel."""
import torch
from torch import nn


class FeatConPolar(nn.Module):
    def __init__(self, num_cls=8, num_emb=32, init=False):
        super().__init__()
        if init:
            raise NotImplementedError('FeatConPolar(init=True) (333-step Adam pre-optimisation, reference fcp.py:36-57) '
                                      'is not on the stc_tt path; RegNet uses init=False')
        self.num_cls = num_cls
        self.vec_grad = nn.Parameter(torch.rand(num_cls, num_emb), requires_grad=False)
        n = num_cls * (num_cls - 1) // 2
        self.register_buffer('cos_dist', torch.full((n,), -1.0 / (num_cls - 1)))
        self.register_buffer('buf_grad', nn.functional.normalize(self.vec_grad.detach(), p=2, dim=-1))

    def choice(self, pro, i):
        """row i of the prototype table repeated pro.shape[0] times (reference fcp.py:72-75); view, no kernel"""
        return self.buf_grad[i:i + 1].expand(pro.shape[0], -1)
