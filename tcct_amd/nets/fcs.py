"""FeatConSuper — host-side mirror of reference nets/fcs.py:52-96 (bin selection + un-normalised dot-product loss).

The arithmetic runs in the fused FPL kernels (tcct_amd/csrc/fpl_optim.hip) for all classes at once; this class keeps the
reference's constructor / attribute surface (`con`, `__name__`) so `RegNet(base, con=args.type_udh)` is unchanged."""
from torch import nn


class FeatConSuper(nn.Module):
    def __init__(self, con='cos', mode='bins', *args):
        super().__init__()
        self.__name__ = con
        self.con = con
        self.bins = 32          # reference fcs.py:35
