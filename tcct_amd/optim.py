"""Flat-buffer AdamW with fused clip_grad_norm_ — the native counterpart of reference kite/loop_seg.py:128-130
(`clip_grad_norm_(params, 12)` + `AdamW.step()`, kite/loopback.py:126-127: lr from the scheduler, wd 2e-4, betas .9/.999).

All parameters that receive gradients live as views of ONE fp32 buffer; their gradients are gathered into one flat buffer
(3.2 MB for stc_tt), optionally all-reduced across ranks (RCCL via torch.distributed, see tcct_amd/dist.py), then two HIP
kernels do sum-of-squares and clip+AdamW.  Parameters whose `.grad is None` are skipped exactly like torch.optim.AdamW
does (so unused parameters — crpe, cls_head, fuse, lap_epl, tau — never decay)."""
import torch

from ._lib import lib, TcctError


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=2e-4, max_norm=12.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_norm = float(max_norm)
        self._flat = None            # (plist, flat_p, flat_g, m, v, sumsq, total_norm)
        self._step = 0
        self.world = 1
        self.allreduce = None        # callable(flat_g) -> None, set by tcct_amd.dist.attach
        self.buckets = None          # tcct_amd.dist.GradBuckets: bucketed all-reduce overlapped with the backward pass
        self.allreduce_mode = 'none'
        self.last_total_norm = None
        self.names = {}              # id(parameter) -> name, for the layout signature of state_dict()
        self._slots_live = False
        self.device_state = None     # [lr, steps taken] on the device once enable_device_state() was called (hipGraph-replayable step)

    def enable_device_state(self):
        """Keep the learning rate and the step count in device memory (tcct_clip_adamw_dev): the step then launches with constant
        arguments and can be captured into a hipGraph; `sync_lr()` pushes the scheduler's current lr before a replay."""
        if self._flat is None:
            raise TcctError('enable_device_state(): take one eager step first (the flat buffers are built at the first step)')
        if self.device_state is None:
            self.device_state = torch.tensor([float(self.param_groups[0]['lr']), float(self._step)], device=self._flat['p'].device,
                                             dtype=torch.float32)
            self._lr_pushed = float(self.param_groups[0]['lr'])
        return self

    def sync_lr(self):
        lr = float(self.param_groups[0]['lr'])
        if self.device_state is not None and lr != self._lr_pushed:
            self.device_state[0:1].fill_(lr)
            self._lr_pushed = lr

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=True)
        if self._flat is not None:
            self._flat['g'].zero_()     # one memset: the slots that backward kernels accumulate into
        if self.buckets is not None:
            self.buckets.begin_step(armed=self._slots_live)

    def state_dict(self):
        """torch's layout plus the flat AdamW state (exp_avg / exp_avg_sq / step live outside `self.state`): the reference never saves
        its optimizer (kite/loopback.py:56-59), so this is an addition, not a wire format"""
        sd = super().state_dict()
        if self._flat is not None:
            # `layout`: the order of the flat buffer as (parameter NAME, shape) in buffer order.  It depends on which parameters had a gradient
            # at the first step and, with TCCT_DP_OVERLAP=1, on the bucket sort: equal numel does not mean equal layout, and neither do equal
            # shapes (dozens of tensors share 32x32x3x3 / [32]): the names decide
            sd['flat'] = dict(step=self._step, m=self._flat['m'].clone(), v=self._flat['v'].clone(), numel=self._flat['n'], layout=self.layout())
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        flat = sd.pop('flat', None)
        super().load_state_dict(sd)
        if flat is not None:
            if self._flat is None:
                raise TcctError('FlatAdamW.load_state_dict(): take one step first (the flat buffers are laid out at the first step)')
            if int(flat['numel']) != self._flat['n']:
                raise TcctError(f"FlatAdamW.load_state_dict(): saved state has {flat['numel']} elements, this optimizer {self._flat['n']}")
            layout = flat.get('layout')
            if layout is not None and not self._same_layout(layout):
                raise TcctError('FlatAdamW.load_state_dict(): the saved moments are laid out in another parameter order (saved with a different '
                                'TCCT_DP_OVERLAP mode or another set of trained parameters); applying them would permute the AdamW state')
            self._flat['m'].copy_(flat['m'])
            self._flat['v'].copy_(flat['v'])
            self._step = int(flat['step'])
            if self.device_state is not None:
                self.device_state[1:2].fill_(float(self._step))

    def _same_layout(self, saved):
        """saved: a `layout` entry of state_dict().  Two formats exist: [(name, shape)] (round 4 on) and the older [shape].  Names are compared only
        where BOTH sides have real ones ('#pos' stands for "attached without a model"); shapes and positions always."""
        mine = self.layout()
        if len(saved) != len(mine):
            return False
        for entry, (name, shape) in zip(saved, mine):
            legacy = len(entry) == 0 or not isinstance(entry[0], str)          # a plain shape tuple / list / torch.Size
            sname, sshape = (None, tuple(entry)) if legacy else (str(entry[0]), tuple(entry[1]))
            if tuple(int(d) for d in sshape) != shape:
                return False
            if sname is not None and not sname.startswith('#') and not name.startswith('#') and sname != name:
                return False
        return True

    def layout(self):
        """[(name, shape)] in flat-buffer order.  Names come from `named=` (tcct_amd.dist.attach passes model.named_parameters()); a parameter
        without one is named by its position in the param groups, which is stable for a given model construction."""
        if self._flat is None:
            return []
        pos = {id(p): i for i, p in enumerate(q for g in self.param_groups for q in g['params'])}
        return [(self.names.get(id(p), f'#{pos[id(p)]}'), tuple(p.shape)) for p in self._flat['plist']]

    def name_parameters(self, named):
        """named: iterable of (name, parameter), e.g. model.named_parameters()"""
        self.names.update({id(p): str(n) for n, p in named})
        return self

    def _build(self):
        plist = [p for g in self.param_groups for p in g['params'] if p.grad is not None]
        if not plist:
            raise TcctError('FlatAdamW.step(): no parameter has a gradient')
        if self.buckets is not None:
            # data-parallel buckets are contiguous ranges of the flat buffer: order the parameters by bucket (stable inside a bucket).
            # Parameters whose gradient arrives through autograd (no in-place slot: used twice per step) are copied into the flat
            # buffer in step(), i.e. after the early buckets have left: they belong to the last bucket.
            last = self.buckets.n_buckets - 1
            key = lambda p: last if getattr(p, '_tcct_multi_use', False) else min(getattr(p, '_tcct_bucket', last), last)   # noqa: E731
            plist = sorted(plist, key=key)
            sizes = [0] * self.buckets.n_buckets
            for p in plist:
                sizes[key(p)] += p.numel()
        dev = plist[0].device
        if dev.type != 'cuda':
            raise TcctError('FlatAdamW needs parameters on the GPU (no CPU fallback)')
        n = sum(p.numel() for p in plist)
        flat_p = torch.empty(n, device=dev, dtype=torch.float32)
        off = 0
        for p in plist:
            k = p.numel()
            flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = flat_p[off:off + k].view_as(p.data)
            off += k
        z = lambda: torch.zeros(n, device=dev, dtype=torch.float32)   # noqa: E731
        flat_g = z()
        off = 0
        for p in plist:             # gradient slots: backward kernels write parameter gradients straight into the flat buffer
            k = p.numel()
            if not getattr(p, '_tcct_multi_use', False):     # parameters used twice per step accumulate through autograd
                p._grad_slot = flat_g[off:off + k].view_as(p)
            off += k
        self._flat = dict(plist=plist, p=flat_p, g=flat_g, m=z(), v=z(),
                          sumsq=torch.zeros((), device=dev, dtype=torch.float64),
                          norm=torch.zeros((), device=dev, dtype=torch.float32), n=n)
        if self.buckets is not None:
            self.buckets.bind(flat_g, sizes, plist)

    @property
    def flat_numel(self):
        return self._flat['n'] if self._flat else 0

    @torch.no_grad()
    def step(self, closure=None):
        if self._flat is None:
            self._build()
        f = self._flat
        if not self._slots_live:
            for p in f['plist']:
                if p.grad is None:
                    raise TcctError('a parameter that had a gradient at the first step has none now; the set of trained '
                                    'parameters must be static (rebuild the optimizer after changing loss flags)')
        if self._slots_live:
            off = 0
            for p in f['plist']:        # kernels wrote most gradients straight into their slots (p.grad stays None then)
                k = p.numel()
                slot = getattr(p, '_grad_slot', None)
                if p.grad is not None and (slot is None or p.grad.data_ptr() != slot.data_ptr()):
                    # produced through autograd (multi-use parameter, or a plain backward outside the pooled step): the
                    # kernels did not touch this slot, so the autograd result IS the gradient
                    if self.buckets is not None and self.buckets.is_launched(off):
                        raise TcctError('a gradient arrived through autograd for a parameter whose bucket has already been all-reduced '
                                        'during the backward pass (mark it _tcct_multi_use, or set TCCT_DP_OVERLAP=0)')
                    f['g'][off:off + k].copy_(p.grad.reshape(-1))
                if slot is not None:
                    p.grad = slot                   # keep the torch contract: p.grad holds the gradient after step()
                off += k
        else:
            torch.cat([p.grad.reshape(-1) for p in f['plist']], out=f['g'])
        self._slots_live = True
        if self.buckets is not None:
            self.buckets.finish()           # launches what the backward pass has not launched yet, then joins the comm stream
        elif self.allreduce is not None:
            self.allreduce(f['g'])
        self._step += 1
        g0 = self.param_groups[0]
        lib.grad_sumsq(f['g'], f['n'], f['sumsq'])
        if self.device_state is not None:
            if not torch.cuda.is_current_stream_capturing():
                self.sync_lr()              # an eager step after enable_device_state() must see the scheduler's current lr too
            lib.clip_adamw_dev(f['p'], f['g'], f['m'], f['v'], f['n'], f['sumsq'], self.max_norm, 1.0 / self.world, self.device_state,
                               float(g0['betas'][0]), float(g0['betas'][1]), float(g0['eps']), float(g0['weight_decay']), f['norm'])
        else:
            lib.clip_adamw(f['p'], f['g'], f['m'], f['v'], f['n'], f['sumsq'], self.max_norm, 1.0 / self.world, float(g0['lr']),
                           float(g0['betas'][0]), float(g0['betas'][1]), float(g0['eps']), float(g0['weight_decay']),
                           self._step, f['norm'])
        self.last_total_norm = f['norm']
