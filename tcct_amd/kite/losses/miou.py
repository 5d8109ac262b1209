"""Validation metrics — mirror of the static scorers of reference kite/losses/miou.py:28-44,69-91.

`pr` is either the one-hot mask tensor [B,C,H,W] returned by `KiteSeg.predict` (a lazy `MaskOneHot`), or any tensor whose
argmax over dim 1 is the class; `gt` one-hot [B,C,H,W] or class indices [B,H,W].  Counting runs in one HIP kernel
(tcct_confusion_counts); the final 15-number arithmetic is done on the tiny count table."""
import torch
from torch import nn

from ..._lib import lib, TcctError
from ...nets.reg import as_label_index


def _counts(pr, gt):
    """-> fp32 [B, C, 3] of {intersection, |pred|, |label|}"""
    if isinstance(pr, MaskOneHot):
        pred, C = pr.index, pr.n_class
    else:
        if pr.dim() != 4:
            raise TcctError('prediction must be [B,C,H,W]')
        C = pr.shape[1]
        pred = as_label_index(pr if pr.dtype == torch.int64 else pr.round().long())
    lab = as_label_index(gt)
    B, H, W = lab.shape
    out = torch.empty((B, C, 3), device=lab.device, dtype=torch.float32)
    lib.confusion_counts(pred, lab, B, H * W, C, out)
    return out


class MaskOneHot:
    """Lazy one-hot mask: class-index uint8 [B,H,W] + n_class; `.dense()` materialises the reference's float [B,C,H,W]."""

    def __init__(self, index, n_class):
        self.index, self.n_class = index, n_class
        self.shape = (index.shape[0], n_class, index.shape[1], index.shape[2])

    def dense(self):
        return torch.nn.functional.one_hot(self.index.long(), self.n_class).permute(0, 3, 1, 2).float()

    def detach(self):
        return self

    def boundaries(self):
        """int32 [B, n_class-1, W]: row at which layer k = 1..n_class-1 starts in every column (number of pixels of the column
        labelled < k) -- the boundary coordinates of a layered OCT segmentation, straight from the class-index mask on the GPU."""
        B, H, W = self.index.shape
        out = torch.empty((B, self.n_class - 1, W), device=self.index.device, dtype=torch.int32)
        lib.mask_boundaries(self.index.contiguous(), out, B, H, W, self.n_class)
        return out


class MDiceLoss(nn.Module):
    @staticmethod
    def _per_class(pr, gt, smooth=1):
        c = _counts(pr, gt)
        return ((2 * c[..., 0] + smooth) / (c[..., 1] + c[..., 2] + smooth)).mean(0)      # [C]

    @staticmethod
    def scores(pr, gt):
        return [float(v) for v in MDiceLoss._per_class(pr, gt).cpu()]

    @staticmethod
    def scorem(pr, gt, start_idx=0):
        return MDiceLoss._per_class(pr, gt)[start_idx:].mean()


class MIouLoss(nn.Module):
    @staticmethod
    def scorem(pr, gt, start_idx=0, smooth=1):
        c = _counts(pr, gt)
        s = ((c[..., 0] + smooth) / (c[..., 1] + c[..., 2] - c[..., 0] + smooth)).mean(0)
        return s[start_idx:].mean()
