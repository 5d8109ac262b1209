"""hipGraph replay of the inference forward.  At small batch the eval forward is launch-bound (~250 kernels, 4.2 ms of host enqueue
for 1.7 ms of GPU work at bs=1 -- the reference validates with bs=1, kite/loop_seg.py:66-106): the kernel sequence is captured once
per input shape with torch.cuda.CUDAGraph (hipGraph on ROCm; the library only launches on the stream it is given, so capture needs
nothing special) and replayed.  Weights and BatchNorm buffers are read through their storage at replay time, so in-place updates
(the fused optimizer, load_state_dict) are seen without re-capturing; a re-bound storage triggers a new capture."""
import os

import torch

from . import ops
from ._lib import TcctError


_POOL = []


def _shared_pool():
    if not _POOL:
        _POOL.append(torch.cuda.graph_pool_handle())
    return _POOL[0]


class GraphedPredict:
    """callable: image batch [B,3|1,H,W] (CUDA) -> (logits of head 0 [B,C,H,W] view, argmax class map uint8 [B,H,W]); both are
    STATIC buffers that the next call overwrites -- clone what must survive."""

    def __init__(self, model, warmup=2):
        self.model, self.warmup = model, warmup
        self._cache = {}
        ops.graphs_exclude_stage_fork('GraphedPredict')

    def _capture(self, img):
        from ._lib import lib
        from .nets.reg import as_nhwc
        model = self.model
        if model.training or torch.is_grad_enabled():
            raise TcctError('GraphedPredict captures the eval forward: call it under model.eval() and torch.no_grad()')
        static_in = img.clone()

        def run():
            out = model(static_in)
            out = out[0] if isinstance(out, (list, tuple)) else out
            lg = as_nhwc(out)
            B, H, W, C = lg.shape
            idx = torch.empty((B, H, W), device=lg.device, dtype=torch.uint8)
            lib.softmax_pick(lg, None, B * H * W, C, None, idx, 0 if lg.dtype == torch.float32 else 1)
            return out, idx
        side = ops.fresh_stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):               # warm-up off the capture: first-call attribute setting, allocator warm
            for _ in range(self.warmup):
                run()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            outs = run()
        return g, static_in, outs

    def __call__(self, img):
        if not img.is_cuda:
            raise TcctError('GraphedPredict needs a CUDA(HIP) input')
        # the graph holds raw pointers: a re-bound parameter / buffer storage (the fused optimizer moves the parameters into its flat
        # buffer at its first step; load_state_dict(assign=True)) must trigger a new capture, in-place updates must not
        ptrs = hash(tuple(t.data_ptr() for t in self.model.state_dict(keep_vars=True).values()))
        key = (tuple(img.shape), img.dtype, img.device.index, ops.INFER_FUSE, ops.PARALLEL_BRANCHES, ptrs)
        ent = self._cache.get(key)
        if ent is None:
            ent = self._cache[key] = self._capture(img.contiguous())
        g, static_in, outs = ent
        static_in.copy_(img)
        g.replay()
        return outs


class GraphedTrainStep:
    """hipGraph replay of the WHOLE training step (forward, losses, backward, clip + AdamW) for launch-bound batch shapes -- the
    reference trains on 256x256 crops (data/octgen.py:8-19), where 8 x 256 x 256 pixels keep the GPU busy for a fraction of the
    ~17 ms the host needs to enqueue the ~800 kernels of a step.

    Protocol: the first `warmup` calls run eagerly ON THE CAPTURE STREAM (they are real optimisation steps), the next call captures
    the step (capture does not execute it) and replays it, later calls copy the batch into the static input buffers and replay.
    Random draws (DropPath masks, the boundary loss's noise) come from torch's graph-safe generator and advance on every replay; the
    optimizer's step count and learning rate live in device memory (FlatAdamW.enable_device_state).  A step that ran on another
    stream earlier would leave autograd AccumulateGrad nodes bound to that stream (kept alive by `KiteSeg.udh_out`), which breaks
    capture -- the references are dropped before capturing."""

    def __init__(self, kite, warmup=3):
        self.k, self.warmup = kite, warmup
        self.calls = 0
        self.graph = None
        self.stream = ops.fresh_stream()          # never an alias of the library's side streams (pool of 32 streams, round-robin)
        ops.graphs_exclude_stage_fork('GraphedTrainStep')
        self.shape = None

    def __call__(self, img, lab):
        k = self.k
        if self.k.optimG.allreduce is not None:
            raise TcctError('GraphedTrainStep: single-process only (the RCCL all-reduce is not captured)')
        key = (tuple(img.shape), tuple(lab.shape), img.dtype, lab.dtype)
        if self.shape is not None and key != self.shape:
            raise TcctError(f'GraphedTrainStep was captured for {self.shape}, got {key}: batches must keep one shape')
        self.shape = key
        cur = torch.cuda.current_stream()
        if self.graph is None:
            if self.calls < self.warmup:
                self.calls += 1
                k.udh_out = None
                self.stream.wait_stream(cur)
                with torch.cuda.stream(self.stream):
                    loss = k.train_step(img, lab)
                cur.wait_stream(self.stream)
                loss.record_stream(cur)
                return loss
            k.optimG.enable_device_state()
            k.udh_out = None
            # a late capture in a long process has crashed inside hipGraphLaunch (cause open): drop every dead autograd
            # graph / cached block of earlier work before capturing, so the capture sees a quiet allocator
            import gc
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            self.s_img, self.s_lab = img.clone(), lab.clone()
            self.stream.wait_stream(cur)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.stream):
                self.s_loss = k.train_step(self.s_img, self.s_lab)
            k.udh_out = None
            k.optimG._step -= 1         # the Python side of step() ran once while capturing; the replay below is the real step
            self.graph = g
            ops.ZERO.frozen = True      # the graph holds pointers into the zero pool
            ops.packs_freeze()          # ... and into the weight-pack descriptor table and the pack buffers it names
        else:
            self.s_img.copy_(img)
            self.s_lab.copy_(lab)
        k.optimG.sync_lr()
        self.graph.replay()
        k.optimG._step += 1
        return self.s_loss
