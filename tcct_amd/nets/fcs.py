"""FeatConSuper / points_selection_bins — mirror of reference nets/fcs.py:25-96 (bin selection + un-normalised dot-product loss).

`RegNet.regular_udh` does NOT go through these methods: it evaluates all classes at once in the fused FPL kernels
(`ops.fpl`: one key sort for the five classes, bin means, loss and its gradient).  The methods below give reference-side
callers (`model.fcs.select1(feat, pred_i, true_i)`, `model.fcs.foreach_loss(pros, tgts)`, `model.fcs(q, k)`) the same
names, arguments and results, for ONE class per call, on the same HIP kernels: radix multi-select of the bin boundaries + bin sums
(`tcct_fpl_select`), prototypes (`tcct_fpl_loss`), scatter of the bin gradients (`tcct_fpl_backward`).  The 32x32 dot product of
`cosinesim` is a single tiny matmul on the device."""
import torch
from torch import nn

from .._lib import lib, dtype_code, TcctError
from .. import ops

BINS = 32        # reference fcs.py:35


class _SelectBins(ops._FastFunction):
    """rows of feat [M,32] whose `true` > .5, ranked by `prob` (descending), averaged over 32 equal rank ranges -> [32,32]
    (reference fcs.py:25-50: the n % 32 lowest-probability rows are dropped; fewer than 32 rows give NaN like the reference's
    mean of an empty selection).  Gradient flows to `feat` only (the reference sorts detached probabilities, reg.py:89)."""

    @staticmethod
    def forward(ctx, feat, prob, true):
        M = feat.shape[0]
        dev = feat.device
        sel = (true.reshape(-1) > 0.5)
        if sel.numel() != M or prob.numel() != M:
            raise TcctError('points_selection_bins: feat, prob and true must describe the same pixels')
        lab = (~sel).to(torch.uint8).contiguous()                 # class 0 = selected rows, class 1 = the rest (ignored: C = 1)
        prob = prob.reshape(-1).to(torch.float32).contiguous()
        counts = torch.empty(16, device=dev, dtype=torch.int32)
        pro_sum = torch.empty((2, BINS, 32), device=dev, dtype=torch.float32)       # class 1 (the unselected rows) is binned too and ignored
        pro = torch.empty((2, BINS, 32), device=dev, dtype=torch.float32)
        scratch = torch.empty((2, BINS, 32), device=dev, dtype=torch.float32)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        binmap = torch.empty(M, device=dev, dtype=torch.uint8)
        zero_proto = torch.zeros((2, 32), device=dev, dtype=torch.float32)
        ws = torch.empty(int(lib.fpl_select_workspace_bytes()), device=dev, dtype=torch.uint8)
        lib.fpl_select(feat, lab, prob, M, 2, ws, counts, binmap, pro_sum, dtype_code(feat.dtype))
        lib.fpl_loss(pro_sum, counts, zero_proto, 2, pro, loss, scratch)
        binmap = torch.where(lab == 0, binmap, torch.full_like(binmap, 255))          # only the selected rows carry a gradient
        ctx.save_for_backward(lab, binmap, counts)
        ctx.cfg = (tuple(feat.shape), feat.dtype, M)
        return pro[0]

    @staticmethod
    def backward(ctx, g):
        lab, binmap, counts = ctx.saved_tensors
        shape, dt, M = ctx.cfg
        per_bin = torch.div(counts[0:1], BINS, rounding_mode='floor').to(torch.float32)
        d_over_n = (g.to(torch.float32) / per_bin).reshape(1, BINS, 32).contiguous()
        dfeat = torch.empty(shape, device=g.device, dtype=dt)
        lib.fpl_backward(lab, binmap, d_over_n, None, 1.0, M, dfeat, dtype_code(dt))
        return dfeat, None, None


def points_selection_bins(feat, prob, true, card=512, **args):
    """reference fcs.py:25-50.  feat [N,32] (bf16 or fp32, contiguous, on the GPU), prob / true: N values each.  -> [32,32] fp32"""
    assert len(feat.shape) == 2, 'feat should contains N*L two dims!'
    if not feat.is_cuda:
        raise TcctError('points_selection_bins needs GPU tensors (no CPU fallback)')
    if feat.shape[1] != 32:
        raise TcctError('points_selection_bins: the HIP kernels are built for 32-wide features (RegNet num_emb=32)')
    if feat.dtype not in (torch.float32, torch.bfloat16):
        feat = feat.float()
    return _SelectBins.apply(feat.contiguous(), prob.detach(), true)


class FeatConSuper(nn.Module):
    def __init__(self, con='cos', mode='bins', *args):
        super().__init__()
        self.__name__ = con
        self.con = con
        self.bins = BINS
        self.mse = nn.MSELoss(reduction='mean')
        self.func = points_selection_bins

    def cosinesim(self, q, k):
        """reference fcs.py:63-67: -mean(q k^T) / C, an un-normalised dot product"""
        return -torch.einsum('nc,kc->nk', [q, k]).mean() / q.shape[-1]

    def forward(self, q, k):          # reference fcs.py:60: self.forward = self.cosinesim
        return self.cosinesim(q, k)

    def foreach_loss(self, fts, gts):
        """reference fcs.py:69-80: matching class pairs only (i == j), summed"""
        losFor = 0
        for i, ft in enumerate(fts):
            for j, gt in enumerate(gts):
                if i == j:
                    losFor = losFor + self.forward(ft, gt)
        return losFor

    def select1(self, feat, pred, true, mask=None, ksize=5, card_select=16):
        """reference fcs.py:82-96: feat [B,32,H,W] (NCHW-shaped; a view of NHWC memory costs nothing), pred [B,1,H,W] = the class's
        (detached) softmax probability, true [B,1,H,W] = its one-hot plane -> [32,32] bin prototypes"""
        assert feat.shape[-2:] == true.shape[-2:], 'shape of feat & true donot match!'
        assert feat.shape[-2:] == pred.shape[-2:], 'shape of feat & pred donot match!'
        dim_latent = feat.shape[1]
        feat = feat.permute(0, 2, 3, 1).reshape(-1, dim_latent)
        true = true.float().round()
        return self.func(feat, pred, true, card=card_select * true.shape[0])
