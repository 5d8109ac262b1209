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


def attach(optimizer):
    """make a FlatAdamW average gradients over the process group before the clip+AdamW kernel"""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get('TCCT_FORCE_DIST', '0') == '1'):
        optimizer.world = dist.get_world_size()
        optimizer.allreduce = allreduce_sum_
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


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
