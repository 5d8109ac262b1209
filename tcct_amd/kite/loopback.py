"""KiteBack — mirror of reference kite/loopback.py:16-139 (seed, optimizer + scheduler, checkpoints, deep supervision).

Same attributes/methods as the reference class; differences are the reference's own defects (imports of the absent
`.utils`/`.optims`, device hard-wired to cuda:0) and the native optimizer: AdamW + clip_grad_norm_ run as two HIP kernels
on one flat fp32 buffer (tcct_amd/optim.py)."""
import glob
import os
import random

import numpy as np
import torch
from torch.optim import lr_scheduler

from .. import dist as tdist
from .. import ops
from ..optim import FlatAdamW
from .losses import get_loss
from .losses.loss import MultiLoss, as_nhwc, as_label_index


def setup_seed(seed):
    """reference kite/loopback.py:16-26"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


class KiteBack(object):
    lossItem = 0
    device = torch.device('cpu')
    coff_ds = 0.5

    def __init__(self, model, dataset, root=None, **args):
        super().__init__()
        self.model = model
        self.root = root if root else 'exp_tcct-bp'         # reference loopback.py:36 always uses this folder
        os.makedirs(self.root, exist_ok=True)
        self.dataset = dataset
        if not self.weights_load('los', desc=False):
            pass                                            # fresh init (reference prints 'weights_init_kaiming')

    def cuda(self, m):
        return m.to(self.device)

    def islrLowerThan(self, thresh=1e-5):
        return self.optimG.param_groups[0]['lr'] < thresh

    def grad_dump(self, epoch):
        """reference loopback.py:56-59: params.tar = {'epoch','loss','lr'}"""
        torch.save({'epoch': epoch, 'loss': self.lossName, 'lr': self.optimG.param_groups[0]['lr']}, self.checkpoint_grad)

    def grad_calc(self, outs, true, ds=True, criterion=None):
        """deep supervision, reference loopback.py:62-73: sum_{i=3,2,1} coff_ds*crit(outs[i]) + crit(outs[0])"""
        losSum = 0
        if ds and isinstance(criterion, MultiLoss) and ops.deep_supervision_dice_ok(outs, self.args.coff_ds) and torch.is_grad_enabled():
            # the whole deep-supervision criterion as one node (same sum, same order, no scalar torch kernels between the Dice kernels)
            logits0 = as_nhwc(outs[0])
            if logits0.shape[1:3] == tuple(outs[1].size) and logits0.shape[-1] <= 8:
                return ops.deep_supervision_dice(logits0, as_label_index(true), list(outs[1:]), self.args.coff_ds)
        if isinstance(outs, (list, tuple)):
            if ds:
                for i in range(len(outs) - 1, 0, -1):
                    losSum = losSum + criterion(outs[i], true) * self.args.coff_ds
            outs = outs[0]
        return losSum + criterion(outs, true)

    def weights_load(self, mode, desc=True):
        path = mode if mode.endswith('.pt') else os.path.join(self.root, mode + '.pt')
        if not os.path.isfile(path):
            return False
        pt = torch.load(path, map_location=self.device, weights_only=True)
        self.model.load_state_dict(pt, strict=False)
        if desc:
            print('\nLoad weight:', path)
        return True

    def weights_desc(self, key='my'):
        for n, m in self.model.named_parameters():
            if key in n:
                print(n, m.detach().cpu().numpy())

    def remove_pths(self, flag_ignore='los'):
        for path in glob.glob(self.root + '/*.pt'):
            if flag_ignore not in path:
                os.remove(path)

    def set_superes(self, loss='ce', lr=0.01, wd=2e-4, **args):
        """reference loopback.py:102-128: resume epoch/lr from params.tar; AdamW(wd=2e-4) + CyclicLR(1e-6..1e-4, 4 up/60 down)
        stepped once per epoch (constructing it resets the lr to 1e-6, as in the reference)."""
        self.checkpoint_grad = os.path.join(self.root, 'params.tar')
        epoch = 0
        if os.path.isfile(self.checkpoint_grad):
            try:
                tar = torch.load(self.checkpoint_grad, weights_only=True)
                epoch, loss, lr = tar['epoch'], tar['loss'], tar['lr']
            except Exception:
                epoch = 0
        self.epoch = epoch
        self.lossName = loss
        params = [p for p in self.model.parameters() if p.requires_grad]
        self.optimG = tdist.attach(FlatAdamW(params, lr=lr, weight_decay=wd, max_norm=12.0), self.model)
        self.schedG = lr_scheduler.CyclicLR(self.optimG, base_lr=1e-6, max_lr=1e-4, cycle_momentum=False, step_size_up=4,
                                            step_size_down=60)

    def set_backend(self, gpu='0', parallel=False, **args):
        """reference loopback.py:130-139.  `parallel=True` (--pl) = one process per GPU under torchrun (tcct_amd/dist.py)."""
        if not torch.cuda.is_available():
            raise RuntimeError('tcct_amd needs an MI355X (HIP device); there is no CPU fallback for the training path')
        local = 0
        if parallel:
            _, _, local = tdist.init()
        self.device = torch.device('cuda', local)
        torch.cuda.set_device(self.device)
        self.model = self.model.to(self.device)
        if parallel:
            tdist.broadcast_params_(self.model)
        self.criterion = get_loss(self.args.los).to(self.device)
