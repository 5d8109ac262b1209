"""Training criterion — mirror of reference kite/losses/loss.py:83-110 (`get_loss` -> `MultiLoss(DiceLoss)`).

`criterion(logits[B,C,H,W], onehot[B,C,H,W] | index[B,H,W]) -> scalar`: softmax over C, per class
1-(1+2*sum(p*g))/(1+sum(p)+sum(g)) with sums over the WHOLE batch, summed over classes.  One fused HIP kernel pair
(tcct_softmax_dice_fwd/bwd) instead of the reference's softmax + 5x3 reductions."""
from torch import nn

from ... import ops
from ...nets.reg import as_label_index, as_nhwc
from ..._lib import TcctError


class DiceLoss(nn.Module):
    __name__ = 'DiceLoss'

    def __init__(self, bi=False):
        super().__init__()
        if bi:
            raise TcctError('DiceLoss(bi=True) (dice2) is not on the stc_tt path')


class MultiLoss(nn.Module):
    __name__ = 'MultiLoss'

    def __init__(self, losses, weight=None):
        super().__init__()
        if not isinstance(losses, DiceLoss):
            raise TcctError('only MultiLoss(DiceLoss) (--los=di / dice) is implemented natively')
        if weight is not None:
            raise TcctError('per-class weights other than 1 are not on the reference path')
        self.losses = losses
        self.WEIGHT = [1, ] * 40

    def forward(self, pr, gt, **args):
        if isinstance(pr, ops.LowResLogits):        # deep-supervision head before its resize: fused resize + softmax + Dice
            return ops.softmax_dice_upsampled(pr, as_label_index(gt))
        return ops.softmax_dice(as_nhwc(pr), as_label_index(gt))


def get_loss(loss='di', **args):
    """reference kite/losses/loss.py:101-110; the MSE branch of the reference is not reachable from task1's recipe."""
    if loss in ('dice', 'di'):
        return MultiLoss(DiceLoss(bi=False))
    raise TcctError(f"--los={loss!r}: only 'di'/'dice' is implemented (the reference recipe, README.md:56-62)")
