"""Test-infrastructure only: import the read-only reference (/root/reference/task1) in THIS container.

Never shipped to / used on the GPU box.  Neutralises the reference's missing third-party imports
(timm, cv2, kite.utils, kite.optims) exactly as SURVEY.md Appendix C describes.  DropPath is the one
piece of third-party arithmetic on the path (timm, version unpinned): restated here with timm>=0.4
semantics, and with an optional queue of forced masks so fixtures are deterministic.
"""
import sys, types, os, glob, random
import numpy as np
import torch
from torch import nn

REF = '/root/reference/task1'


class DropPath(nn.Module):
    forced = None  # list of [B] 0/1 mask tensors, consumed in call order when set

    def __init__(self, p=0.):
        super().__init__()
        self.drop_prob = p

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        if DropPath.forced is not None:
            m = DropPath.forced.pop(0).to(x.dtype).view((x.shape[0],) + (1,) * (x.ndim - 1))
        else:
            m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * m.div(keep)


def install():
    if 'nets' in sys.modules and hasattr(sys.modules['nets'], 'stc_tt'):
        return
    timm, data, models, layers = (types.ModuleType(n) for n in
                                  ('timm', 'timm.data', 'timm.models', 'timm.models.layers'))
    data.IMAGENET_DEFAULT_MEAN = data.IMAGENET_DEFAULT_STD = (0.5, 0.5, 0.5)
    layers.DropPath, layers.trunc_normal_ = DropPath, nn.init.trunc_normal_
    timm.data, timm.models, models.layers = data, models, layers
    sys.modules.update({'timm': timm, 'timm.data': data, 'timm.models': models,
                        'timm.models.layers': layers})
    sys.modules['cv2'] = types.ModuleType('cv2')
    u = types.ModuleType('kite.utils')
    u.random, u.np, u.os, u.glob = random, np, os, glob
    u.__all__ = ['random', 'np', 'os', 'glob']
    o = types.ModuleType('kite.optims')
    o.__all__ = []
    sys.modules.update({'kite.utils': u, 'kite.optims': o})
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use('Agg')


def load():
    """returns (nets module namespace dict, KiteSeg, setup_seed, get_loss)"""
    install()
    import nets
    from kite.loop_seg import KiteSeg
    from kite.loopback import setup_seed
    from kite.losses import get_loss
    return nets, KiteSeg, setup_seed, get_loss
