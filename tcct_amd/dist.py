"""Data-parallel replicas: one process per GPU, gradients summed with ONE all-reduce of the flat fp32 buffer (3.2 MB for
stc_tt) over RCCL/xGMI (torch.distributed backend "nccl" == RCCL on ROCm) — no reference counterpart (the
reference has no collective at all).  Per-replica semantics are those of torch-DDP over the reference: BN statistics and
the batch-global Dice sums are per replica; gradients are averaged (the 1/world factor is folded into the clip+AdamW
kernel).  On CPU (tests) the same code runs over gloo."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (world, rank, local_rank)."""
    world, rank, local = env_world()
    force = os.environ.get('TCCT_FORCE_DIST', '0') == '1'      # exercise the RCCL path on a single GPU (tests)
    if (world > 1 or (force and 'RANK' in os.environ)) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:         # TCCT_DIST_BACKEND=gloo: several ranks on ONE GPU (tests; RCCL refuses to share a device)
            backend = os.environ.get('TCCT_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():
            local = local % max(torch.cuda.device_count(), 1)
        dist.init_process_group(backend=backend, world_size=world, rank=rank)
    return world, rank, local


def world_rank():
    """(world size, rank) of the initialised process group, (1, 0) without one"""
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def shard_batch(n_global, world, rank):
    """contiguous slice of the global minibatch owned by `rank` (global batch 64 -> 8 x 8)"""
    if n_global % world:
        raise ValueError(f'global batch {n_global} is not divisible by world size {world}')
    per = n_global // world
    return slice(rank * per, (rank + 1) * per)


def allreduce_sum_(flat):
    """in-place sum all-reduce of a flat gradient buffer (no-op for a single process)"""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get('TCCT_FORCE_DIST', '0') == '1'):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def average_buffers_(model):
    """mean over the ranks of every floating-point buffer (BatchNorm running_mean / running_var); integer buffers (num_batches_tracked)
    are equal on all ranks already.  Called before validation so that all ranks evaluate one and the same model."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    bufs = [b for b in model.buffers() if b.is_floating_point()]
    if not bufs:
        return
    flat = torch.cat([b.detach().reshape(-1).float() for b in bufs])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    off = 0
    with torch.no_grad():
        for b in bufs:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()


N_BUCKETS = 3


def bucket_of(name):
    """Static gradient bucket of a parameter, by the order in which the backward pass finishes with its module:
       0  fusion + decoder + heads (FTC.tran_*, head, dec*, t32*, aux*): done when the backward pass reaches the encoder outputs
       1  the deep encoder levels: CrossResNet blocks 1-4, MPViT stages 1-3
       2  everything that finishes last: CrossResNet level 0 + first conv, MPViT stem + stage 0, and the loss-side RegNet modules
          (lap_reg / lap_map are applied twice per step: their gradients accumulate through autograd and are copied in at step())"""
    n = name[5:] if name.startswith('base.') else name
    if n.startswith('base_cnn.path_estan.'):
        return 2 if n.split('.')[2] == '0' else 1
    if n.startswith(('base_vit.mhca_stages.', 'base_vit.patch_embed_stages.')):
        return 2 if n.split('.')[2] == '0' else 1
    if n.startswith(('base_cnn.', 'base_vit.')) or not name.startswith('base.'):
        return 2
    return 0


class GradBuckets:
    """Bucketed all-reduce of the flat gradient, overlapped with the backward pass.  The model marks the tensor edges that leave a
    bucket's modules (ops.grad_mark); when the backward pass has crossed all edges of bucket b (and buckets < b have left), the
    bucket's slice of the flat gradient is all-reduced on a dedicated comm stream that waits for every compute stream of the step
    (events; no host sync).  FlatAdamW.step() launches whatever is left (always the last bucket) and joins the comm stream before the
    sum-of-squares kernel.  Launch order is 0,1,2 on every rank by construction, so the collectives match across ranks."""

    KEYS = {'dec': 0, 'deep': 1}

    def __init__(self):
        self.n_buckets = N_BUCKETS
        self.ranges = None
        self.flat = None
        self.comm = None
        self.works = []
        self.launched = 0
        self.ready = [False] * N_BUCKETS
        self.armed = False
        self.overlap = True
        self.launch_log = []            # (bucket, 'backward' | 'step') of the last step: tests and bench read it
        self.params = None
        self.fallbacks = 0              # steps whose early launches were called off (a gradient arrived through autograd)

    def bind(self, flat_g, sizes, params=None):
        """params: the optimizer's parameters in flat-buffer order.  With them, a bucket only leaves DURING the backward pass when every one of
        its gradients was written in place into the flat buffer (p.grad still None / already the slot): a gradient that arrived through
        autograd (fp32 parity mode, an op without an in-place slot) is copied in by step(), i.e. after an early launch would have sent stale
        data -- such a step falls back to launching everything in finish() (same decision on every rank: the code path is deterministic)"""
        self.flat = flat_g
        self.params = None
        if params is not None:
            self.params, i = [], 0
            for n in sizes:
                grp, got = [], 0
                while got < n:
                    grp.append(params[i]); got += params[i].numel(); i += 1
                self.params.append(grp)
        self.ranges, off = [], 0
        for n in sizes:
            self.ranges.append((off, off + n))
            off += n
        if flat_g.is_cuda:
            from . import ops
            self.comm = ops.fresh_stream(flat_g.device)

    def begin_step(self, armed):
        self.works, self.launched = [], 0
        self.ready = [False] * self.n_buckets
        self.armed = bool(armed) and self.overlap and self.ranges is not None
        self.launch_log = []

    def is_launched(self, offset):
        return self.ranges is not None and any(b < self.launched and s <= offset < e for b, (s, e) in enumerate(self.ranges))

    def on_mark(self, key):
        """ops.grad_mark listener (autograd's device thread): all edges of `key` crossed"""
        b = self.KEYS.get(key)
        if b is None or not self.armed:
            return
        self.ready[b] = True
        while self.armed and self.launched < self.n_buckets - 1 and self.ready[self.launched]:
            if self.params is not None and not all(p.grad is None or (getattr(p, '_grad_slot', None) is not None and p.grad.data_ptr() == p._grad_slot.data_ptr())
                                                   for p in self.params[self.launched]):
                self.armed = False          # not every gradient of this bucket sits in the flat buffer yet: everything leaves in finish()
                self.fallbacks += 1
                break
            self._launch(self.launched, 'backward')

    def _launch(self, b, where):
        s, e = self.ranges[b]
        self.launched = b + 1
        self.launch_log.append((b, where))
        if e <= s:
            return
        part = self.flat[s:e]
        if self.comm is None:
            self.works.append(dist.all_reduce(part, op=dist.ReduceOp.SUM, async_op=True))
            return
        from . import ops
        for st in ops.step_streams() + [torch.cuda.current_stream(self.flat.device)]:
            ev = torch.cuda.Event()
            ev.record(st)
            self.comm.wait_event(ev)
        with torch.cuda.stream(self.comm):
            self.works.append(dist.all_reduce(part, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        if self.ranges is None:
            raise RuntimeError('GradBuckets.finish() before bind()')
        while self.launched < self.n_buckets:
            self._launch(self.launched, 'step')
        for w in self.works:
            w.wait()                    # nccl: the current stream waits for the collective; gloo: the host does
        if self.comm is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.comm)
        self.works = []


def attach(optimizer, model=None):
    """make a FlatAdamW average gradients over the process group before the clip+AdamW kernel.

    Default: ONE all-reduce of the whole flat buffer (3.2 MB) right after the backward pass.  TCCT_DP_OVERLAP=1 (needs `model`):
    N_BUCKETS static buckets on a comm stream, launched while the backward pass is still running (GradBuckets).  Both are tested
    for gradient equivalence with the single-process mean (tests/test_model_gpu.py).  The overlapped form is NOT the default because
    it measures slower where it could be measured -- forced process group of one rank over RCCL on one MI355X, bench shape, same box:
    30.62 ms single all-reduce, 31.68 ms three buckets all launched in step(), 32.02 ms three buckets overlapped with backward
    (gpurun_out/r02i_dp_ab.log): every extra collective costs ~0.35 ms of stream hand-offs for a transfer that takes tens of
    microseconds, and there is no bandwidth to hide at this message size."""
    if model is not None:
        optimizer.name_parameters(model.named_parameters())         # layout signature of FlatAdamW.state_dict() (names, not shapes)
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get('TCCT_FORCE_DIST', '0') == '1'):
        optimizer.world = dist.get_world_size()
        optimizer.allreduce = allreduce_sum_
        optimizer.allreduce_mode = 'single blocking all-reduce after backward'
        if model is not None and os.environ.get('TCCT_DP_OVERLAP', '0') == '1':
            from . import ops
            for name, p in model.named_parameters():
                p._tcct_bucket = bucket_of(name)
            optimizer.buckets = GradBuckets()
            ops.set_grad_mark_listener(optimizer.buckets.on_mark)
            optimizer.allreduce_mode = f'{N_BUCKETS} buckets (decoder | deep encoder levels | level 0 + stems) on a comm stream, overlapped with backward'
    return optimizer


def broadcast_params_(model, src=0):
    """identical initial weights / buffers on every replica"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src)


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_strings(s):
    """one string per rank, in rank order (bench.py records each rank's device in its result line)"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, s)
        return out
    return [s]


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
