"""torch.autograd.Function wrappers over the libtcct_hip.so C-ABI.

PyTorch here is plumbing only (device memory via the caching allocator, the current HIP stream, the autograd tape);
every arithmetic op of the hot path is a HIP kernel behind `tcct_amd._lib.lib`.  No CPU fallback exists: a CPU tensor
or a missing .so raises.  Activations are NHWC-contiguous (`[N,H,W,C]`, tokens `[B,N,C]` are the same memory).
"""
import os
import weakref

import torch

from ._lib import lib, dtype_code, TcctError, launch_on, F32, BF16  # noqa: F401

ACT = {None: 0, 'none': 0, 'lrelu': 1, 'hswish': 2, 'gelu': 3, 'sigmoid': 4, 'abs': 5}


class _ZeroPool:
    """One pre-zeroed buffer per training step for every accumulation output (BN/LN partial sums, parameter-gradient buffers that
    have no slot in the optimizer's flat gradient): ONE memset per step instead of ~270 hipMemsetAsync calls.  Active between
    `begin_step()` and `end_step()`; otherwise the entry points clear their outputs themselves."""

    def __init__(self):
        self.buf = None
        self.off = 0
        self.active = False
        self.frozen = False     # set once a hipGraph has captured a step that uses the pool (tcct_amd/graph.py)

    def begin(self, device, nbytes=8 << 20):
        if self.buf is None or self.buf.device != device or self.buf.numel() < nbytes:
            if self.frozen:     # a captured hipGraph holds pointers into the pool: replacing it would leave them dangling
                raise TcctError('the zero pool cannot be re-allocated after a training step was captured into a hipGraph '
                                f'(pool on {self.buf.device}, step on {device})')
            self.buf = torch.empty(nbytes, device=device, dtype=torch.uint8)
        self.buf.zero_()
        self.off = 0
        self.active = True
        lib.set_outputs_prezeroed(1)

    def end(self):
        if self.active:
            self.active = False
            lib.set_outputs_prezeroed(0)

    def get(self, shape, dtype, device):
        """zero-initialised tensor when the pool is active, uninitialised otherwise (the entry point clears it then)"""
        n = 1
        for d in shape:
            n *= d
        nbytes = n * dtype.itemsize
        if not self.active:
            return torch.empty(shape, device=device, dtype=dtype)
        start = (self.off + 255) // 256 * 256
        if self.buf.device != device or start + nbytes > self.buf.numel():
            return torch.zeros(shape, device=device, dtype=dtype)
        self.off = start + nbytes
        return self.buf[start:start + nbytes].view(dtype).view(shape)


ZERO = _ZeroPool()


# ---- one weight-pack launch per step (round 3) -----------------------------------------------------------------------------------------
# The bf16 [tap][co][ci] packs (+ the flipped / transposed input-gradient packs) of the ~40 dense 32 -> 32 convolutions used to be made by one
# small launch per convolution per step (39 x ~5 us on the critical stream).  Convolutions seen during a pooled step are remembered; from the
# next step on begin_step() re-packs ALL of them with one launch (the optimizer has just changed the weights) and the forward pass picks its
# pack out of the cache.  `PACK_ALL = False` (bench.py --set PACK_ALL=0) restores the per-call launches.
PACK_ALL = True
# ent: id(weight) -> [weakref(weight), pack buffer, (KH, KW), data_ptr, step last looked up].  Entries hold NO strong reference to the weight:
# a dead model's entries disappear at the next begin_step(), as do entries no convolution asked for during the previous step.
# frozen / keep: once a hipGraph has captured a step whose pack launch reads a descriptor table, that table and the pack buffers it names are kept
# alive for the life of the process (`keep`) -- a replay reads their addresses -- and a changed entry set builds a NEW table beside them.
_PACKS = {'ent': {}, 'desc': None, 'sig': None, 'fresh': False, 'step': 0, 'frozen': False, 'keep': []}


def packs_freeze():
    """called by tcct_amd.graph after a capture that contains the pack-all launch: pins the captured table and its pack buffers"""
    _PACKS['frozen'] = True
    if _PACKS['desc'] is not None:
        _PACKS['keep'].append((_PACKS['desc'], [e[1] for e in _PACKS['ent'].values()]))


def _pack_lookup(w, KH, KW):
    """(forward pack, input-gradient pack) of weight w made by this step's pack-all launch, or None"""
    if not (PACK_ALL and ZERO.active):
        return None
    ent = _PACKS['ent'].get(id(w))
    if ent is None or ent[0]() is not w or ent[3] != w.data_ptr() or ent[2] != (KH, KW):
        wp2 = torch.empty(2 * KH * KW * 1024, device=w.device, dtype=torch.bfloat16)
        _PACKS['ent'][id(w)] = [weakref.ref(w), wp2, (KH, KW), w.data_ptr(), _PACKS['step']]
        _PACKS['sig'] = None                    # the descriptor table is rebuilt at the next begin_step()
        return None
    ent[4] = _PACKS['step']
    if not _PACKS['fresh'] or _PACKS['sig'] is None or not ent[5:]:
        return None
    n = KH * KW * 1024
    return ent[1][:n], ent[1][n:]


def _pack_evict():
    """drop entries whose weight is gone, whose storage was re-bound (the flat optimizer buffer, load_state_dict) or that no convolution of the
    previous step looked up (another model's step in between, a changed loss configuration)"""
    step = _PACKS['step']
    for k in [k for k, e in _PACKS['ent'].items() if e[0]() is None or e[0]().data_ptr() != e[3] or e[4] < step]:
        del _PACKS['ent'][k]
    _PACKS['step'] = step + 1


def _pack_all(device):
    ents = [e for e in _PACKS['ent'].values() if e[0]().device == device]
    _PACKS['fresh'] = False
    if not (PACK_ALL and ents):
        return
    sig = tuple((e[3], e[1].data_ptr()) for e in ents)
    if sig != _PACKS['sig'] and torch.cuda.is_current_stream_capturing():
        return                      # the descriptor upload is a host-to-device copy: not inside a hipGraph capture (the convolutions pack per call then)
    if sig != _PACKS['sig']:
        rows = [[e[3], e[1].data_ptr(), e[2][0], e[2][1]] for e in ents]
        _PACKS['desc'] = torch.tensor(rows, dtype=torch.int64).to(device)       # a NEW tensor: tables named by captured graphs stay in `keep`
        _PACKS['sig'] = sig
    for e in _PACKS['ent'].values():
        del e[5:]
    for e in ents:
        e.append(True)              # packed by this launch
    lib.conv32_pack_weights_multi(_PACKS['desc'], len(ents))
    _PACKS['fresh'] = True


BILINEAR_BWD_X2 = True      # exact x2 resizes: the separable lane-exchange backward (tcct_bilinear_bwd_x2; round 6); False: the tiled gather kernel (A/B timing)
_BILINEAR_X2_SET = [True]


def begin_step(device):
    """call once per training step before the forward (KiteSeg.train_step does): arms the zero pool"""
    device = torch.device(device)
    if device.type == 'cuda' and device.index is None:          # torch.device('cuda') != torch.device('cuda', 0): never re-allocate the pool for that
        device = torch.device('cuda', torch.cuda.current_device())
    ZERO.begin(device)
    if BILINEAR_BWD_X2 != _BILINEAR_X2_SET[0]:          # kernel A/B (bench.py --set BILINEAR_BWD_X2=0): the library-wide switch follows the module constant
        lib.bilinear_bwd_x2(int(BILINEAR_BWD_X2))
        _BILINEAR_X2_SET[0] = BILINEAR_BWD_X2
    fpl_lazy_grad_reset()
    _pack_evict()
    _pack_all(device)
    _STEP['main'] = torch.cuda.current_stream(device)
    for v in _STEP['marks'].values():
        v[0] = v[1] = 0


# ---- gradient-readiness marks (data-parallel overlap, tcct_amd/dist.py) ------------------------------------------------------
# The optimizer of a data-parallel run all-reduces its flat gradient in buckets while the rest of the backward pass is still running
# (decoder/FTC -> deep encoder levels -> level 0).  A bucket is complete when the backward pass has crossed every
# tensor edge that leads out of its modules; the model marks those edges with grad_mark(x, key): identity in the forward, and its
# backward node reports "one more edge of `key` crossed".  Without a registered listener grad_mark is a no-op (single-GPU path).
_STEP = {'main': None, 'marks': {}, 'listener': None}


def set_grad_mark_listener(fn):
    """fn(key) is called (from autograd's device thread) when the backward pass has crossed ALL edges marked `key` in this step"""
    _STEP['listener'] = fn
    _STEP['marks'].clear()


def step_streams():
    """every stream this library may have launched kernels of the running step on: the step's main stream + the side streams"""
    out = [] if _STEP['main'] is None else [_STEP['main']]
    for v in _SIDE_STREAMS.values():
        out.append(v[0] if isinstance(v, tuple) else v)
    return out



class _FastFunction(torch.autograd.Function):
    """torch.autograd.Function whose `apply` goes straight to the C++ implementation (round 6).  `Function.apply` first checks for a `setup_context` override, asks whether
    functorch transforms are active and runs `unwrap_dead_wrappers` over the arguments -- 6-8 us per call, ~130 calls per forward pass; the forward pass of the bench step is
    HOST-bound (tools/host_profile.py: ~8 ms of enqueue for ~5.5 ms of kernels once the two encoders overlap), so this is step time, not bookkeeping.  None of these
    Functions defines `setup_context` or is used under functorch transforms (vmap / grad / jvp)."""

    @classmethod
    def apply(cls, *args):
        return super(torch.autograd.Function, cls).apply(*args)


class _GradMark(_FastFunction):
    @staticmethod
    def forward(ctx, x, key):
        ctx.key = key
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        cnt = _STEP['marks'].get(ctx.key)
        fn = _STEP['listener']
        if cnt is not None and fn is not None:
            cnt[1] += 1
            if cnt[1] == cnt[0]:
                fn(ctx.key)
        return dy, None


def grad_mark(x, key):
    if _STEP['listener'] is None or not ZERO.active or not (torch.is_grad_enabled() and x.requires_grad):
        return x
    _STEP['marks'].setdefault(key, [0, 0])[0] += 1
    return _GradMark.apply(x, key)


def end_step():
    ZERO.end()
    if _WGRAD_USED:             # weight gradients were issued on the side stream: the optimizer (current stream) must see them
        cur = torch.cuda.current_stream()
        for st in _WGRAD_USED.values():
            cur.wait_stream(st)
        _WGRAD_USED.clear()
    _WGRAD_KEEP.clear()         # after the join: blocks freed from here on are reused by work ordered behind the weight gradients


# Weight-gradient kernels only feed the optimizer, so inside a pooled step they are issued on a side stream and joined in
# end_step(): the input-gradient chain (the critical path of backward) continues without waiting for them and the
# latency-bound weight gradients of the coarse levels overlap it.  TCCT_STREAMS=0 disables.
_WGRAD_USED = {}
_WGRAD_KEEP = []
DW_WGRAD_SIDE = True
_WGRAD_FRESH_EVENT = False          # experiment on the late-capture crash
_WGRAD_RECORD_STREAM = False      # the old behaviour, kept for A/B measurements only


def fresh_stream(device=None, avoid=()):
    """A torch.cuda.Stream whose HIP stream is none of: the current stream, the library's side streams, `avoid`.  torch hands out streams
    from a pool of 32 per device, round-robin: in a long process the 33rd `torch.cuda.Stream()` IS the first one again, and a capture
    stream that aliased the weight-gradient / encoder side stream would turn their fork-join into a wait of a stream on itself.  (A guard:
    it was written while chasing the late-capture segfault of DESIGN 5b, which it did NOT cure.)"""
    taken = {torch.cuda.current_stream(device).cuda_stream}
    for v in _SIDE_STREAMS.values():
        taken.add((v[0] if isinstance(v, tuple) else v).cuda_stream)
    for st in avoid:
        taken.add(st.cuda_stream)
    for _ in range(64):
        st = torch.cuda.Stream(device=device)
        if st.cuda_stream not in taken:
            return st
    raise TcctError('fresh_stream: no stream of the pool is free of aliases')


class _wgrad_stream:
    def __init__(self, enable, *tensors):
        self.enable, self.tensors = enable, tensors

    def __enter__(self):
        if not self.enable:
            return self
        cur = torch.cuda.current_stream()
        key = ('wgrad', cur.device.index)
        ent = _SIDE_STREAMS.get(key)
        if ent is None:         # the side stream and ONE reusable event for the cur -> side dependency (a wait captures the state of the
            ent = _SIDE_STREAMS[key] = (fresh_stream(cur.device), torch.cuda.Event())   # event at the time it is issued)
        self.side, ev = ent
        if _WGRAD_FRESH_EVENT:
            ev = torch.cuda.Event()
        ev.record(cur)
        self.side.wait_event(ev)
        # the block only launches this library's kernels into pre-allocated gradient slots: name the stream for them instead of
        # re-binding torch's current stream (launch_on: ~1 us instead of ~25 us per weight gradient)
        self.ctx = launch_on(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if not self.enable:
            return False
        self.ctx.__exit__(*exc)
        # autograd may drop x / dy as soon as this node returns while the side stream still reads them.  They are kept alive until
        # end_step() has joined the side stream -- NOT tensor.record_stream(side): with every activation and gradient of the step
        # marked that way the caching allocator could not reuse a block until the side stream caught up, the host runs most of a
        # step ahead, and the reserved pool grew by ~3.5 GB per step (124 GB after 30 steps, then a multi-second free-and-retry stall)
        if _WGRAD_RECORD_STREAM:
            for t in self.tensors:
                t.record_stream(self.side)
        else:
            _WGRAD_KEEP.extend(self.tensors)
        _WGRAD_USED[id(self.side)] = self.side
        return False


def _slot_written(*params):
    return PARALLEL_BRANCHES and ZERO.active and all(p is None or getattr(p, '_grad_slot', None) is not None for p in params)


def _grad_out(param, shape=None):
    """destination of a parameter gradient: the parameter's slot in the optimizer's flat gradient buffer (pre-zeroed by
    zero_grad; written in place, so the optimizer needs no gather) or a pooled / fresh tensor"""
    slot = getattr(param, '_grad_slot', None)
    if slot is not None and ZERO.active:
        return slot.view(shape if shape is not None else slot.shape)
    return ZERO.get(tuple(shape if shape is not None else param.shape), torch.float32, param.device)


def _ret(t, param):
    """what a backward returns for a parameter gradient: None when the kernel already wrote it into the parameter's slot of the
    flat gradient buffer (autograd must not clone/accumulate it again; the optimizer reads the slot), else the tensor itself"""
    if t is None:
        return None
    slot = getattr(param, '_grad_slot', None)
    if slot is not None and t.data_ptr() == slot.data_ptr():
        return None
    return t


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise TcctError('tcct_amd ops need CUDA(HIP) tensors; there is no CPU fallback')
        if not t.is_contiguous():
            raise TcctError(f'non-contiguous tensor of shape {tuple(t.shape)} passed to a HIP op')


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _as(t, dtype):
    t = _c(t)
    return t if t.dtype == dtype else t.to(dtype)


# ------------------------------------------------------------------------------------------------- conv
def _mfma32_ok(in_dt, out_dt, Cin, Cin_w, Cout, KH, KW, stride, padh, padw):
    """the MFMA implicit-GEMM kernel covers bf16 32->32 stride-1 'same' convolutions (3x3, 1xk, kx1)"""
    return (in_dt == torch.bfloat16 and out_dt == torch.bfloat16 and Cin == 32 and Cin_w == 32 and Cout == 32 and stride == 1
            and 2 * padh == KH - 1 and 2 * padw == KW - 1 and KH * KW > 1 and (KH == 1 or KW == 1 or (KH == 3 and KW == 3)))


F32_MFMA = True        # False: the VALU convolution for the fp32 parity mode (A/B timing, bisecting)
# fp32 MFMA for the pointwise forward / input gradient / weight gradient.  The FORWARD stays on the sequential VALU kernel in the parity mode: on
# the formula-weight fixtures some BatchNorm channels at the 2x2 / 4x4 levels have a batch variance at fp32 rounding level, and the MFMA summation
# order (exact to 2-5e-7 against fp64, like the VALU kernel's) moved rstd enough to scale the whole CNN level-0 gradient by 0.93 -- outside the
# envelope the fixtures assert, which is built from oneDNN's sequential order (tests/test_model_gpu.py, profiles/r03_parity.md).  TCCT_F32_PW_FWD=1
# puts the forward on the matrix pipes as well (+~5 % fp32 throughput).
F32_PW = [os.environ.get('TCCT_F32_PW_FWD', '0') == '1', True, True]


def _mfma32f_ok(in_dt, out_dt, Cin, Cin_w, Cout, KH, KW, stride, padh, padw):
    """the fp32 MFMA kernels (conv_f32_mfma.hip) cover fp32 32->32 stride-1 'same' convolutions (3x3, 1xk, kx1, k <= 13) of the parity mode"""
    return (F32_MFMA and in_dt == torch.float32 and out_dt == torch.float32 and Cin == 32 and Cin_w == 32 and Cout == 32 and stride == 1
            and 2 * padh == KH - 1 and 2 * padw == KW - 1 and 1 < KH * KW <= 13 and (KH == 1 or KW == 1 or (KH == 3 and KW == 3)))


def _pwf_ok(in_dt, out_dt, Cin, Cin_w, Cout, KH, KW, stride, padh, padw):
    """fp32 MFMA pointwise kernels (conv_f32_mfma.hip): fp32 1x1 convolutions / Linear with channel counts that are multiples of 32"""
    return (F32_MFMA and in_dt == torch.float32 and out_dt == torch.float32 and KH == 1 and KW == 1 and stride == 1 and padh == 0 and padw == 0
            and Cin == Cin_w and Cin % 32 == 0 and Cout % 32 == 0)


def _mfma_slabs_ok(in_dt, out_dt, Cin, Cin_w, Cout, KH, KW, stride, padh, padw):
    """wider bf16 stride-1 'same' convolutions (3x3, 1xk, kx1 with channel counts that are multiples of 32) as 32x32 sub-GEMMs of the
    conv32 kernels: MPViT stem[1] (32->64) and the wide CNN encoder of stc_tb / gtc_tb (32-64-96-128-256, nets/tcct.py:861-864)"""
    return (in_dt == torch.bfloat16 and out_dt == torch.bfloat16 and Cin == Cin_w and Cin % 32 == 0 and Cout % 32 == 0
            and 32 <= Cin <= 256 and 32 <= Cout <= 256 and (Cin, Cout) != (32, 32) and stride == 1
            and 2 * padh == KH - 1 and 2 * padw == KW - 1 and KH * KW > 1 and max(KH, KW) <= 13
            and (KH == 1 or KW == 1 or (KH == 3 and KW == 3)))


def _mfma_slabs_f32_ok(in_dt, out_dt, Cin, Cin_w, Cout, KH, KW, stride, padh, padw):
    """wider fp32 stride-1 'same' convolutions as 32x32 sub-GEMMs of the fp32 MFMA kernels (MPViT stem[1] 32->64 is a third of the parity mode's
    convolution time on the VALU kernel; the wide CNN encoder of stc_tb / gtc_tb)"""
    return (F32_MFMA and in_dt == torch.float32 and out_dt == torch.float32 and Cin == Cin_w and Cin % 32 == 0 and Cout % 32 == 0
            and 32 <= Cin <= 256 and 32 <= Cout <= 256 and (Cin, Cout) != (32, 32) and stride == 1
            and 2 * padh == KH - 1 and 2 * padw == KW - 1 and 1 < KH * KW <= 13 and (KH == 1 or KW == 1 or (KH == 3 and KW == 3)))


def _conv_slabs_fwd_f32(x, w, bias, y, N, H, W, Cin, Cout, KH, KW, padh, padw, transposed):
    """fp32 counterpart of _conv_slabs_fwd (no fused statistics): transposed=True computes the input gradient (x = dy, w OIHW [Cin here][Cout here])"""
    for oh in range(Cout // 32):
        for ih in range(Cin // 32):
            wp = torch.empty(KH * KW * 1024, device=x.device, dtype=torch.float32)
            if not transposed:
                lib.conv32f_pack_weights_sub(w, wp, KH, KW, 0, Cin, 32 * oh, 32 * ih)
            else:
                lib.conv32f_pack_weights_sub(w, wp, KH, KW, 1, Cout, 32 * ih, 32 * oh)
            b = bias[32 * oh:32 * oh + 32] if (bias is not None and ih == 0) else None
            lib.conv32f_fwd_strided(x, wp, b, y, N, H, W, KH, KW, padh, padw, Cin, 32 * ih, Cout, 32 * oh, 1 if ih > 0 else 0)


def _conv_slabs_fwd(x, w, bias, y, N, H, W, Cin, Cout, KH, KW, padh, padw, transposed, stats=None, stat_pre=0):
    """y[..., Cout] = conv(x[..., Cin]) via 32x32 sub-GEMMs; transposed=True computes the input gradient (x=dy, w OIHW [Cin_orig=Cout
    here][...]) i.e. roles of the weight's O/I dims swap"""
    wp = torch.empty(KH * KW * 1024, device=x.device, dtype=torch.bfloat16)
    for oh in range(Cout // 32):
        for ih in range(Cin // 32):
            if not transposed:
                lib.conv32_pack_weights_sub(w, wp, KH, KW, 0, Cin, 32 * oh, 32 * ih)
            else:       # w is OIHW [Cin(this call's input = orig Cout)][Cout(this call's output = orig Cin)]
                lib.conv32_pack_weights_sub(w, wp, KH, KW, 1, Cout, 32 * ih, 32 * oh)
            b = bias[32 * oh:32 * oh + 32] if (bias is not None and ih == 0) else None
            if stats is not None and ih == Cin // 32 - 1:       # the last input slab: the accumulated output is what the BatchNorm statistics count
                lib.conv32_fwd_strided_bnstats(x, wp, b, y, N, H, W, KH, KW, padh, padw, Cin, 32 * ih, Cout, 32 * oh, 1 if ih > 0 else 0, stats, stat_pre)
            else:
                lib.conv32_fwd_strided(x, wp, b, y, N, H, W, KH, KW, padh, padw, Cin, 32 * ih, Cout, 32 * oh, 1 if ih > 0 else 0)
            if ih + 1 < Cin // 32 or oh + 1 < Cout // 32:
                wp = torch.empty_like(wp)


# Fused 3x3 conv32 backward (tcct_conv32_bwd3x3): correct and tested, but OFF by default -- alone it takes 0.49 ms against 0.23 + 0.31 ms for
# the separate input-gradient / weight-gradient kernels at level 0, and inside the step it gains nothing (29.3 vs 29.3 ms, same box): the
# separate weight gradient runs on the side stream beside the input-gradient chain, the fused kernel sits on the critical path and is
# bound by the latency of its one 8-wave block per CU (111 KB of LDS), not by HBM.  `bench.py --set FUSED_CONV_BWD=1` enables it.
FUSED_CONV_BWD = False
FUSED_PW_BWD = True        # False: separate input-gradient / weight-gradient kernels (A/B timing)


def _pw_bwd_ok(x, dy, Cin, Cin_w, Cout, KH, KW, stride, padh, padw):
    """shapes of the fused pointwise backward kernel (tcct_pw_bwd): bf16 1x1, channel counts in {32, 64, 96, 128}"""
    return (x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and KH == 1 and KW == 1 and stride == 1 and padh == 0 and padw == 0
            and Cin == Cin_w and Cin % 32 == 0 and Cout % 32 == 0 and 32 <= Cin <= 128 and 32 <= Cout <= 128
            and x.numel() * 2 < 2 ** 31 and dy.numel() * 2 < 2 ** 31)


def _pw_ok(in_dt, Cin, Cin_w, KH, KW, stride, padh, padw):
    """the MFMA pointwise kernels cover bf16 1x1 convs / Linear with Cin % 32 == 0"""
    return (in_dt == torch.bfloat16 and KH == 1 and KW == 1 and stride == 1 and padh == 0 and padw == 0 and Cin == Cin_w
            and Cin % 32 == 0)


class _Conv2d(_FastFunction):
    @staticmethod
    def forward(ctx, x, w, bias, stride, padh, padw, out_dtype, stats_box, fork=False):
        """fork: also return an alias of x for the OTHER consumers of x -- their gradient then arrives in backward() and is added
        inside the input-gradient kernel's epilogue (tcct_conv32_fwd_add) instead of by an autograd accumulation pass"""
        ctx.set_materialize_grads(False)      # an unused output must arrive as None in backward(), not as a zero-filled tensor
        _chk(x, w, bias)
        N, H, W, Cin = x.shape
        Cout, Cin_w, KH, KW = w.shape
        Ho = (H + 2 * padh - KH) // stride + 1
        Wo = (W + 2 * padw - KW) // stride + 1
        odt = out_dtype or x.dtype
        y = torch.empty((N, Ho, Wo, Cout), device=x.device, dtype=odt)
        mfma = _mfma32_ok(x.dtype, odt, Cin, Cin_w, Cout, KH, KW, stride, padh, padw)
        if _pw_ok(x.dtype, Cin, Cin_w, KH, KW, stride, padh, padw) and not mfma:
            if stats_box is not None and odt == torch.bfloat16 and Cout % 32 == 0 and Cout <= 160:
                sums = ZERO.get((2 * Cout,), torch.float64, x.device) if ZERO.active else torch.zeros(2 * Cout, device=x.device, dtype=torch.float64)
                lib.pw_fwd_bnstats(x, w, bias, y, N * H * W, Cin, Cout, sums, stats_box[0])
                stats_box[1] = sums
            else:
                lib.pw_fwd(x, w, bias, y, N * H * W, Cin, Cout, 0, dtype_code(odt))
        elif _mfma_slabs_ok(x.dtype, odt, Cin, Cin_w, Cout, KH, KW, stride, padh, padw):
            sums = None
            if stats_box is not None:       # fused train-mode BN statistics of the consumer (MPViT stem[1])
                sums = ZERO.get((2 * Cout,), torch.float64, x.device) if ZERO.active else torch.zeros(2 * Cout, device=x.device, dtype=torch.float64)
                stats_box[1] = sums
            _conv_slabs_fwd(x, w, bias, y, N, H, W, Cin, Cout, KH, KW, padh, padw, False, sums, stats_box[0] if stats_box is not None else 0)
        elif mfma:
            cached = _pack_lookup(w, KH, KW)
            if cached is not None:          # made by this step's single pack launch (begin_step)
                wp, ctx.wp_t = cached
            elif ctx.needs_input_grad[0]:   # the input-gradient pack of the same weights comes out of the same launch (used by backward())
                wp2 = torch.empty(2 * KH * KW * 1024, device=x.device, dtype=torch.bfloat16)
                lib.conv32_pack_weights_both(w, wp2, KH, KW)
                wp, ctx.wp_t = wp2[:KH * KW * 1024], wp2[KH * KW * 1024:]
            else:
                wp = torch.empty(KH * KW * 1024, device=x.device, dtype=torch.bfloat16)
                lib.conv32_pack_weights(w, wp, KH, KW, 0)
            if stats_box is not None:       # fused train-mode BN statistics of the consumer (stats_box = [pre_act_code, None])
                sums = ZERO.get((64,), torch.float64, x.device) if ZERO.active else torch.zeros(64, device=x.device, dtype=torch.float64)
                lib.conv32_fwd_bnstats(x, wp, bias, y, N, H, W, KH, KW, padh, padw, sums, stats_box[0])
                stats_box[1] = sums
            else:
                lib.conv32_fwd(x, wp, bias, y, N, H, W, KH, KW, padh, padw)
        elif _mfma32f_ok(x.dtype, odt, Cin, Cin_w, Cout, KH, KW, stride, padh, padw):
            wp = torch.empty(KH * KW * 1024, device=x.device, dtype=torch.float32)
            lib.conv32f_pack_weights(w, wp, KH, KW, 0)
            lib.conv32f_fwd(x, wp, bias, None, y, N, H, W, KH, KW, padh, padw)
        elif _mfma_slabs_f32_ok(x.dtype, odt, Cin, Cin_w, Cout, KH, KW, stride, padh, padw):
            _conv_slabs_fwd_f32(x, w, bias, y, N, H, W, Cin, Cout, KH, KW, padh, padw, False)
        elif F32_PW[0] and _pwf_ok(x.dtype, odt, Cin, Cin_w, Cout, KH, KW, stride, padh, padw):
            lib.pwf_fwd(x, w, bias, y, N * H * W, Cin, Cout, 0)
        else:
            lib.conv2d_fwd(x, w, bias, y, N, H, W, Cin, Cin_w, Cout, KH, KW, stride, padh, padw, dtype_code(x.dtype),
                           dtype_code(odt))
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padh, padw, bias is not None)
        # gradient destinations: w may be a view of the nn.Linear weight (conv2d wrapper) -> look through ._base
        wsrc = w if hasattr(w, '_grad_slot') or w._base is None else w._base
        ctx.params = (wsrc, bias)
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, w = ctx.saved_tensors
        stride, padh, padw, has_bias = ctx.cfg
        wsrc, bsrc = ctx.params
        if dy is None:          # only the alias was used downstream
            return dskip, None, None, None, None, None, None, None, None
        dy = _c(dy)
        if dskip is not None:
            dskip = _as(dskip, x.dtype)
        N, H, W, Cin = x.shape
        Cout, Cin_w, KH, KW = w.shape
        dx = dw = db = None
        if (FUSED_PW_BWD and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and _pw_bwd_ok(x, dy, Cin, Cin_w, Cout, KH, KW, stride, padh, padw)
                and (dskip is None or dskip.shape == x.shape)):
            # 1x1 convolution: input gradient, weight gradient and bias gradient in ONE pass over dy (tcct_pw_bwd); a second
            # consumer's gradient of x (fork) rides on the dx epilogue
            dx = torch.empty_like(x)
            dw = _grad_out(wsrc, tuple(w.shape))
            db = _grad_out(bsrc) if has_bias else None
            lib.pw_bwd(x, dy, w, dskip, dx, dw, db, N * H * W, Cin, Cout)
            return dx, _ret(dw, wsrc), _ret(db, bsrc), None, None, None, None, None, None
        if (FUSED_CONV_BWD and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and KH == 3 and KW == 3 and dy.dtype == torch.bfloat16
                and _mfma32_ok(x.dtype, dy.dtype, Cin, Cin_w, Cout, KH, KW, stride, padh, padw) and H * W * 64 < 2 ** 31):
            # dense 3x3 32->32: input gradient (+ a second consumer's gradient) and weight / bias gradient from ONE staging of dy
            wp = getattr(ctx, 'wp_t', None)
            if wp is None:
                wp = torch.empty(9 * 1024, device=x.device, dtype=torch.bfloat16)
                lib.conv32_pack_weights(w, wp, 3, 3, 1)
            dx = torch.empty_like(x)
            dw = _grad_out(wsrc, tuple(w.shape))
            db = _grad_out(bsrc) if has_bias else None
            lib.conv32_bwd3x3(x, dy, wp, dskip, dx, dw, db, N, H, W)
            return dx, _ret(dw, wsrc), _ret(db, bsrc), None, None, None, None, None, None
        if ctx.needs_input_grad[0]:
            if stride != 1 or Cin != Cin_w:
                raise TcctError('conv2d dgrad: only stride-1 convs with unpadded channels need an input gradient')
            dx = torch.empty_like(x)
            if (_pw_ok(dy.dtype, Cout, Cout, KH, KW, stride, padh, padw) and x.dtype == torch.bfloat16
                    and not _mfma32_ok(dy.dtype, x.dtype, Cout, Cout, Cin, KH, KW, stride, padh, padw)):
                if dskip is not None and Cin % 32 == 0:
                    lib.pw_dgrad_residual(dy, w, dskip, dx, None, N * H * W, Cout, Cin)
                    dskip = None
                else:
                    lib.pw_fwd(dy, w, None, dx, N * H * W, Cout, Cin, 1, dtype_code(x.dtype))
            elif _mfma_slabs_ok(dy.dtype, x.dtype, Cout, Cout, Cin, KH, KW, stride, padh, padw):
                _conv_slabs_fwd(dy, w, None, dx, N, H, W, Cout, Cin, KH, KW, KH - 1 - padh, KW - 1 - padw, True)
            elif _mfma32_ok(dy.dtype, x.dtype, Cout, Cout, Cin, KH, KW, stride, padh, padw):
                wp = getattr(ctx, 'wp_t', None)
                if wp is None:
                    wp = torch.empty(KH * KW * 1024, device=x.device, dtype=torch.bfloat16)
                    lib.conv32_pack_weights(w, wp, KH, KW, 1)
                if dskip is not None:
                    lib.conv32_fwd_add(dy, wp, None, dskip, dx, N, H, W, KH, KW, KH - 1 - padh, KW - 1 - padw)
                    dskip = None
                else:
                    lib.conv32_fwd(dy, wp, None, dx, N, H, W, KH, KW, KH - 1 - padh, KW - 1 - padw)
            elif _mfma_slabs_f32_ok(dy.dtype, x.dtype, Cout, Cout, Cin, KH, KW, stride, padh, padw):
                _conv_slabs_fwd_f32(dy, w, None, dx, N, H, W, Cout, Cin, KH, KW, KH - 1 - padh, KW - 1 - padw, True)
            elif F32_PW[1] and _pwf_ok(dy.dtype, x.dtype, Cout, Cout, Cin, KH, KW, stride, padh, padw):
                lib.pwf_fwd(dy, w, None, dx, N * H * W, Cout, Cin, 1)
            elif _mfma32f_ok(dy.dtype, x.dtype, Cout, Cout, Cin, KH, KW, stride, padh, padw):
                wp = torch.empty(KH * KW * 1024, device=x.device, dtype=torch.float32)
                lib.conv32f_pack_weights(w, wp, KH, KW, 1)
                lib.conv32f_fwd(dy, wp, None, dskip, dx, N, H, W, KH, KW, KH - 1 - padh, KW - 1 - padw)
                dskip = None
            else:
                lib.conv2d_dgrad(dy, w, dx, N, H, W, Cin, Cout, KH, KW, padh, padw, dtype_code(dy.dtype), dtype_code(x.dtype))
            if dskip is not None:       # kernels without the fused epilogue: one explicit pass
                tot = torch.empty_like(dx)
                lib.add(dx, dskip, tot, tot.numel(), dtype_code(tot.dtype))
                dx = tot
        elif dskip is not None:
            dx = dskip
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            with _wgrad_stream(_slot_written(wsrc, bsrc if has_bias else None), x, dy):
                dw = _grad_out(wsrc, tuple(w.shape))
                db = _grad_out(bsrc) if has_bias else None
                if _mfma32_ok(x.dtype, dy.dtype, Cin, Cin_w, Cout, KH, KW, stride, padh, padw):
                    lib.conv32_wgrad(x, dy, dw, db, N, H, W, KH, KW, padh, padw)
                elif _mfma32f_ok(x.dtype, dy.dtype, Cin, Cin_w, Cout, KH, KW, stride, padh, padw):
                    lib.conv32f_wgrad(x, dy, dw, db, N, H, W, KH, KW, padh, padw)
                elif _mfma_slabs_f32_ok(x.dtype, dy.dtype, Cin, Cin_w, Cout, KH, KW, stride, padh, padw):
                    if not ZERO.active:
                        dw.zero_()
                        if db is not None:
                            db.zero_()
                    for oh in range(Cout // 32):
                        for ih in range(Cin // 32):
                            lib.conv32f_wgrad_strided(x, dy, dw, db if ih == 0 else None, N, H, W, KH, KW, padh, padw, Cin, 32 * ih, Cout, 32 * oh,
                                                      Cin, 32 * oh, 32 * ih)
                elif F32_PW[2] and _pwf_ok(x.dtype, dy.dtype, Cin, Cin_w, Cout, KH, KW, stride, padh, padw) and Cout <= 160:
                    lib.pwf_wgrad(x, dy, dw, db, N * H * W, Cin, Cout)
                elif _mfma_slabs_ok(x.dtype, dy.dtype, Cin, Cin_w, Cout, KH, KW, stride, padh, padw):
                    if not ZERO.active:
                        dw.zero_()
                        if db is not None:
                            db.zero_()
                    for oh in range(Cout // 32):
                        for ih in range(Cin // 32):
                            lib.conv32_wgrad_strided(x, dy, dw, db if ih == 0 else None, N, H, W, KH, KW, padh, padw, Cin, 32 * ih, Cout,
                                                     32 * oh, Cin, 32 * oh, 32 * ih)
                elif (_pw_ok(x.dtype, Cin, Cin_w, KH, KW, stride, padh, padw) and dy.dtype == torch.bfloat16 and Cout % 32 == 0
                      and Cout <= 160):
                    lib.pw_wgrad(x, dy, dw, db, N * H * W, Cin, Cout)
                elif _pw_ok(x.dtype, Cin, Cin_w, KH, KW, stride, padh, padw) and dy.dtype == torch.bfloat16 and Cout % 32 == 0:
                    # wider outputs (the qkv Linear of the factorised attention, Cout = 3 dim): slabs of <= 160 columns on the same kernel
                    dy2, dw2 = dy.view(-1, Cout), dw.view(Cout, Cin)
                    for c0 in range(0, Cout, 160):
                        n = min(160, Cout - c0)
                        lib.pw_wgrad_strided(x, dy2[0, c0:], Cout, dw2[c0:], db[c0:] if db is not None else None, N * H * W, Cin, n)
                elif KH == 1 and KW == 1 and stride == 1 and Cout <= 8 and Cin == Cin_w and Cin <= 256 and (
                        x.dtype == torch.bfloat16 or dy.dtype == torch.float32):
                    lib.pw_wgrad_smalln(x, dy, dw, db, N * H * W, Cin, Cout, dtype_code(x.dtype), dtype_code(dy.dtype))
                else:
                    lib.conv2d_wgrad(x, dy, dw, db, N, H, W, Cin, Cin_w, Cout, KH, KW, stride, padh, padw, dtype_code(x.dtype),
                                     dtype_code(dy.dtype))
        return dx, _ret(dw, wsrc), _ret(db, bsrc), None, None, None, None, None, None


def conv2d_fork(x, w, bias=None, stride=1, pad=0, stats_pre=None):
    """(conv2d(x), x'): x' aliases x and is to be read by the other consumers of x (see _Conv2d.forward `fork`)"""
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    if not (torch.is_grad_enabled() and x.requires_grad and x.dim() == 4):
        return conv2d(x, w, bias, stride, pad, stats_pre=stats_pre), x
    box = [ACT[stats_pre], None] if stats_pre is not None else None
    y, alias = _Conv2d.apply(x, w, bias, stride, ph, pw, None, box, True)
    if box is not None and box[1] is not None:
        y._bn_sums = (box[1], box[0])
    return y, alias


def conv2d(x, w, bias=None, stride=1, pad=0, out_dtype=None, stats_pre=None):
    """x [N,H,W,Cin] (or tokens [B,N,C]); w OIHW fp32 (or [Cout,Cin] for nn.Linear).
    stats_pre: None, or the pre-activation name ('none', 'lrelu', ...) of a train-mode BatchNorm that consumes the output: kernels
    that can, accumulate the BN statistics in their epilogue and tag the result (`_bn_sums`) so `batchnorm` skips its own pass."""
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    tok = x.dim() == 3
    if tok:
        x = x.unsqueeze(2)
    if w.dim() == 2:
        w = w.view(w.shape[0], w.shape[1], 1, 1)
    box = [ACT[stats_pre], None] if stats_pre is not None else None
    y = _Conv2d.apply(x, w, bias, stride, ph, pw, out_dtype, box, False)
    if tok:
        y = y.squeeze(2)
    if box is not None and box[1] is not None:
        y._bn_sums = (box[1], box[0])
    return y


# ---- conv3x3 -> conv3x3 with nothing between them as ONE launch each way (round 5, csrc/conv_chain.hip) ---------------------------------------
# CrossCNNBlock.block12 (reference nets/tcct.py:808-810).  Forward: the intermediate is written once (the backward pass needs it) and not read back;
# backward: the input-gradient chain dy -> d mid -> dx the same way (d mid is the dy operand of the first convolution's weight gradient).  Results
# bit-identical to the two-launch path (TCCT_CONV_CHAIN=0), so no rounding point moves.
CONV_CHAIN = os.environ.get('TCCT_CONV_CHAIN', '1') != '0'
# Levels 0-1 only: on the small maps two launches of the tiled kernel are faster than one row-stream chain (8 x 200 x 276: 0.037 vs 0.040 ms,
# 8 x 100 x 138: 0.017 vs 0.029; `tools/chain_bench.py`) -- a wave walking down a strip is a long serial chain, and the small maps have few strips
CHAIN_MIN_PIXELS = 1 << 20


def conv3x3_chain_ok(x, w1, b1, w2, b2, stride1, pad1, stride2, pad2):
    return (CONV_CHAIN and x.dim() == 4 and x.dtype == torch.bfloat16 and x.is_cuda and x.shape[-1] == 32 and torch.is_grad_enabled()
            and tuple(w1.shape) == (32, 32, 3, 3) and tuple(w2.shape) == (32, 32, 3, 3) and stride1 == 1 and stride2 == 1
            and tuple(pad1) == (1, 1) and tuple(pad2) == (1, 1) and x.shape[1] * x.shape[2] * 64 < 2 ** 31
            and x.shape[0] * x.shape[1] * x.shape[2] >= CHAIN_MIN_PIXELS)


def _packs33(ctx_w, need_t):
    """(forward pack, input-gradient pack or None) of a 3x3 32 -> 32 weight: this step's pack-all launch, or packed here"""
    cached = _pack_lookup(ctx_w, 3, 3)
    if cached is not None:
        return cached
    if need_t:
        wp2 = torch.empty(2 * 9 * 1024, device=ctx_w.device, dtype=torch.bfloat16)
        lib.conv32_pack_weights_both(ctx_w, wp2, 3, 3)
        return wp2[:9 * 1024], wp2[9 * 1024:]
    wp = torch.empty(9 * 1024, device=ctx_w.device, dtype=torch.bfloat16)
    lib.conv32_pack_weights(ctx_w, wp, 3, 3, 0)
    return wp, None


class _ConvChain33(_FastFunction):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, stats_box, fork):
        ctx.set_materialize_grads(False)
        _chk(x, w1, b1, w2, b2)
        N, H, W, _ = x.shape
        mid, y = torch.empty_like(x), torch.empty_like(x)
        need_t = ctx.needs_input_grad[0]
        wp1, ctx.wp_t1 = _packs33(w1, need_t)
        wp2, ctx.wp_t2 = _packs33(w2, True)             # d mid is always needed: it is the dy operand of the first convolution's weight gradient
        sums = None
        if stats_box is not None:
            sums = ZERO.get((64,), torch.float64, x.device) if ZERO.active else torch.zeros(64, device=x.device, dtype=torch.float64)
            stats_box[1] = sums
        lib.conv32_chain33(x, wp1, b1, mid, wp2, b2, y, None, N, H, W, sums)
        ctx.save_for_backward(x, mid, w1, w2)
        ctx.params = (w1, b1, w2, b2)
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, mid, w1, w2 = ctx.saved_tensors
        p1, pb1, p2, pb2 = ctx.params
        if dy is None:          # only the alias was used downstream
            return dskip, None, None, None, None, None, None
        dy = _c(dy)
        N, H, W, _ = x.shape
        # what is asked for (a frozen block12 must neither cost two full-resolution weight-gradient passes nor have its slots in the flat gradient
        # buffer written -- AdamW would step a frozen parameter): as _Conv2d.backward, by ctx.needs_input_grad
        nx, nw1, nb1, nw2, nb2 = ctx.needs_input_grad[:5]
        g1, g2 = nw1 or (pb1 is not None and nb1), nw2 or (pb2 is not None and nb2)
        need_mid = nx or g1             # d mid feeds the input-gradient chain and the first convolution's weight gradient
        dx = dmid = None
        if need_mid:
            wt2 = ctx.wp_t2
            if wt2 is None:
                wt2 = torch.empty(9 * 1024, device=x.device, dtype=torch.bfloat16)
                lib.conv32_pack_weights(w2, wt2, 3, 3, 1)
            dmid = torch.empty_like(x) if g1 else None          # nobody reads d mid when the first convolution is frozen: the chain does not store it
            if nx:
                wt1 = ctx.wp_t1
                if wt1 is None:
                    wt1 = torch.empty(9 * 1024, device=x.device, dtype=torch.bfloat16)
                    lib.conv32_pack_weights(w1, wt1, 3, 3, 1)
                dx = torch.empty_like(x)
                lib.conv32_chain33(dy, wt2, None, dmid, wt1, None, dx, _as(dskip, x.dtype) if dskip is not None else None, N, H, W, None)
            else:
                lib.conv32_fwd(dy, wt2, None, dmid, N, H, W, 3, 3, 1, 1)
                dx = dskip
        else:
            dx = dskip
        dw1 = db1 = dw2 = db2 = None
        if g1 or g2:
            with _wgrad_stream(_slot_written(p1 if g1 else None, pb1 if g1 else None, p2 if g2 else None, pb2 if g2 else None),
                               *[t for t in (x, mid, dy, dmid) if t is not None]):
                if g2:
                    dw2, db2 = _grad_out(p2, tuple(w2.shape)), (_grad_out(pb2) if pb2 is not None else None)
                    lib.conv32_wgrad(mid, dy, dw2, db2, N, H, W, 3, 3, 1, 1)
                if g1:
                    dw1, db1 = _grad_out(p1, tuple(w1.shape)), (_grad_out(pb1) if pb1 is not None else None)
                    lib.conv32_wgrad(x, dmid, dw1, db1, N, H, W, 3, 3, 1, 1)
        return dx, _ret(dw1, p1), _ret(db1, pb1), _ret(dw2, p2), _ret(db2, pb2), None, None


def conv3x3_chain_infer_ok(x, w1, w2, stride1, pad1, stride2, pad2):
    return (CONV_CHAIN and INFER_FUSE and not torch.is_grad_enabled() and x.dim() == 4 and x.dtype == torch.bfloat16 and x.is_cuda and x.shape[-1] == 32
            and tuple(w1.shape) == (32, 32, 3, 3) and tuple(w2.shape) == (32, 32, 3, 3) and stride1 == 1 and stride2 == 1
            and tuple(pad1) == (1, 1) and tuple(pad2) == (1, 1) and x.shape[1] * x.shape[2] * 64 < 2 ** 31
            and x.shape[0] * x.shape[1] * x.shape[2] >= CHAIN_MIN_PIXELS)


def conv3x3_chain_infer(x, w1, b1, w2, b2):
    """inference (no_grad): conv3x3(conv3x3(x)) with the intermediate never written (check conv3x3_chain_infer_ok first)"""
    _chk(x, w1, b1, w2, b2)
    N, H, W, _ = x.shape
    y = torch.empty_like(x)
    wp = torch.empty(2 * 9 * 1024, device=x.device, dtype=torch.bfloat16)
    lib.conv32_pack_weights(w1, wp[:9 * 1024], 3, 3, 0)
    lib.conv32_pack_weights(w2, wp[9 * 1024:], 3, 3, 0)
    lib.conv32_chain33(x, wp[:9 * 1024], b1, None, wp[9 * 1024:], b2, y, None, N, H, W, None)
    return y


def conv3x3_chain(x, w1, b1, w2, b2, stats_pre=None, fork=False):
    """conv3x3(conv3x3(x; w1, b1); w2, b2), both 32 -> 32 'same' (check conv3x3_chain_ok first).  stats_pre: None or 'lrelu' (the train-mode BatchNorm that
    consumes the output: its statistics come out of the same launch, `_bn_sums`).  fork: also return an alias of x for the OTHER consumers of x, whose
    gradient is then added inside the input-gradient chain's epilogue."""
    if stats_pre not in (None, 'lrelu'):
        raise TcctError('conv3x3_chain: fused statistics only behind LeakyReLU')
    box = [ACT[stats_pre], None] if stats_pre is not None else None
    if fork and x.requires_grad:
        y, alias = _ConvChain33.apply(x, w1, b1, w2, b2, box, True)
    else:
        y, alias = _ConvChain33.apply(x, w1, b1, w2, b2, box, False), x
    if box is not None and box[1] is not None:
        y._bn_sums = (box[1], box[0])
    return (y, alias) if fork else y


# The two encoders (CNN / ViT) only share the input and are issued on separate HIP streams:
# the launch-latency-bound kernels of the coarse levels then overlap other work instead of running one after another on an
# otherwise idle chip (-2.5 ms/step; forking block12/block34 and InvRes/token-mixer
# as well gained nothing more).  Autograd replays every node on the stream of its forward, so
# the backward pass overlaps the same way.  TCCT_STREAMS=0 issues everything on one stream.
PARALLEL_BRANCHES = os.environ.get('TCCT_STREAMS', '1') != '0'
# round 6: inside an MHCA stage whose maps have at most this many pixels (B*H*W) the transformer half runs on its own stream beside the InvRes half
# (MHCA_stage.forward); 0 disables.  tools/attrib_trace.sh: ViT L3-L4 are 2.4 ms of back-to-back 10-40 us launches in one stream -- but the fork pays at EVERY
# stage (same-box A/B, ms per step: stages 2-3 only 21.31 -> 21.05; + stage 1 20.94 -> 20.78; all four 20.97 -> 20.67): the depthwise kernels of one half are
# VALU-bound, the GEMMs of the other bandwidth-bound, and they fill each other's gaps.  Default: every stage.
STAGE_FORK_MAX_PIXELS = 1 << 30


FUSION_FORK = True       # FTC.forward: tran_vit / tran_cnn fusion of levels 1-2 on its own stream beside levels 3-4 + head (training, bench-like shapes)


def fusion_fork_ok(c5):
    """c5: the coarsest CNN level [B,h,w,32].  Small crops are launch-bound on the HOST: another fork / join there costs more than it gives"""
    return (FUSION_FORK and STAGE_FORK_MAX_PIXELS > 0 and PARALLEL_BRANCHES and torch.is_grad_enabled() and c5.is_cuda
            and c5.shape[0] * c5.shape[1] * c5.shape[2] >= FUSION_FORK_MIN_PIXELS)


FUSION_FORK_MIN_PIXELS = 4096       # coarsest level of 8 x 256 x 256 crops: 2048


def graphs_exclude_stage_fork(what):
    """hipGraph capture and the nested stage fork exclude each other in one process: a graph captured AFTER the fork's stream has been used crashes inside
    hipGraphLaunch at its first replay on ROCm 7.2 (tests/test_model_gpu.py::test_graphed_train_step_matches_eager; the same late-capture segfault as
    profiles/HISTORY.md 5b, now with a trigger that reproduces).  tcct_amd.graph calls this before a capture: the fork is switched off for the rest of the
    process, and a process that has already used it is refused loudly instead of crashing later."""
    global STAGE_FORK_MAX_PIXELS
    if any(k[0] == 'vit_enc' for k in _SIDE_STREAMS):
        raise TcctError(f'{what}: this process has already run training steps with the nested stage fork (tcct_amd.ops.STAGE_FORK_MAX_PIXELS > 0); a hipGraph '
                        'captured now crashes in hipGraphLaunch (DESIGN 6).  Set tcct_amd.ops.STAGE_FORK_MAX_PIXELS = 0 before the first step (KiteSeg does with '
                        '--graph=true / TCCT_GRAPH=1) or capture in a fresh process.')
    STAGE_FORK_MAX_PIXELS = 0
_SIDE_STREAMS = {}


def _tensors(o):
    if torch.is_tensor(o):
        yield o
    elif isinstance(o, (list, tuple)):
        for e in o:
            yield from _tensors(e)


def run_parallel(tag, fa, fb):
    """(fa(), fb()) with fb issued on the side stream `tag` and joined before returning"""
    if not (PARALLEL_BRANCHES and torch.cuda.is_available()):
        return fa(), fb()
    cur = torch.cuda.current_stream()
    side = _SIDE_STREAMS.get((tag, cur.device.index))
    if side is None:
        side = _SIDE_STREAMS[(tag, cur.device.index)] = fresh_stream(cur.device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        rb = fb()
    ra = fa()
    cur.wait_stream(side)
    if ZERO.active:
        # inside a pooled training step: keep what was allocated on the side stream and is consumed on the current one alive until end_step() instead of
        # record_stream()ing it -- every later piece of side-stream work starts behind a `side.wait_stream(cur)`, i.e. behind the readers, so a block freed after
        # the step's end cannot be overwritten early; record_stream() parks a freed block until the host-side event query sees the GPU catch up, and with the
        # host a third of a step ahead the reserved pool grew by 0.4 GB per step (tests/test_fullsize_gpu.py::test_allocator_pool_stays_bounded_over_steps_fullsize)
        _WGRAD_KEEP.extend(_tensors(rb))
    else:
        for t in _tensors(rb):
            t.record_stream(cur)        # allocated on the side stream, consumed on the current one
    return ra, rb


def side_stream(tag):
    """the library's side stream `tag` of the current device (created on first use)"""
    cur = torch.cuda.current_stream()
    side = _SIDE_STREAMS.get((tag, cur.device.index))
    if side is None:
        side = _SIDE_STREAMS[(tag, cur.device.index)] = fresh_stream(cur.device)
    return side


def keep_until_end_of_step(*tensors_and_stream):
    """tensors allocated on a side stream and consumed on `stream` (the last argument): kept alive until end_step() inside a pooled step, record_stream()ed otherwise
    (see run_parallel)"""
    *ts, stream = tensors_and_stream
    if ZERO.active:
        _WGRAD_KEEP.extend(ts)
    else:
        for t in ts:
            t.record_stream(stream)


def run_interleaved(tag, ga, gb, outs_b):
    """Exhaust the generators ga (current stream) and gb (side stream `tag`) ALTERNATELY, gb first, and join.  The two encoders are
    issued level by level in turn rather than one after the other: the runtime lets the host run only a bounded number of launches
    ahead of the GPU, so with `ViT then CNN` the second stream received its first kernel when the first was nearly done, and -- autograd
    replays nodes in reverse creation order -- the backward pass ran the CNN encoder completely before the first ViT kernel.  Alternating
    creation order keeps both streams fed in both passes.  outs_b: the list gb fills (its tensors are handed to the current stream)."""
    if not (PARALLEL_BRANCHES and torch.cuda.is_available()):
        for _ in gb:
            pass
        for _ in ga:
            pass
        return
    cur = torch.cuda.current_stream()
    side = _SIDE_STREAMS.get((tag, cur.device.index))
    if side is None:
        side = _SIDE_STREAMS[(tag, cur.device.index)] = fresh_stream(cur.device)
    side.wait_stream(cur)
    done_a = done_b = False
    end = object()
    while not (done_a and done_b):
        if not done_b:
            with torch.cuda.stream(side):
                done_b = next(gb, end) is end
        if not done_a:
            done_a = next(ga, end) is end
    cur.wait_stream(side)
    for t in _tensors(outs_b):
        t.record_stream(cur)        # allocated on the side stream, consumed on the current one


INFER_FUSE = True       # eval-mode forward under no_grad folds BatchNorm / activations into the convolution epilogues (False: op by op)


def conv_bn_act(x, w, bias=None, stride=1, pad=0, bn=None, pre_act=None, post_act=None):
    """Inference only: post_act(BN_eval(pre_act(conv(x) + bias))) with the eval-mode BatchNorm and the activations
    folded into the epilogue of the MFMA kernels -- no separate normalisation / activation pass over the output.
    bn: None or (gamma, beta, running_mean, running_var, eps).  bf16 1x1 / 32->32 convolutions take the fused kernels; every
    other shape (and the fp32 parity mode) runs convolution, `batchnorm(training=False)` and `act` one after the other."""
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
        raise TcctError('conv_bn_act is inference-only (call it under torch.no_grad())')
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    tok = x.dim() == 3
    x4 = x.unsqueeze(2) if tok else x
    w4 = w.view(w.shape[0], w.shape[1], 1, 1) if w.dim() == 2 else w
    _chk(x4, w4, bias)
    N, H, W, Cin = x4.shape
    Cout, Cin_w, KH, KW = w4.shape
    pre, post = ACT[pre_act], ACT[post_act]
    mfma = _mfma32_ok(x4.dtype, x4.dtype, Cin, Cin_w, Cout, KH, KW, stride, ph, pw)
    pwk = _pw_ok(x4.dtype, Cin, Cin_w, KH, KW, stride, ph, pw) and not mfma and (bn is not None or pre != 0 or post != 0)
    if (not (mfma or pwk) and bn is not None and Cin == 32 and Cin_w == 32 and _mfma_slabs_ok(x4.dtype, x4.dtype, Cin, Cin_w, Cout, KH, KW, stride, ph, pw)):
        # round 6: a wider convolution with ONE input slab (MPViT stem[1], 32 -> 64): every 32-channel output slab with its part of the eval-mode BatchNorm + activations
        # in the epilogue (no separate normalisation pass over the 64-channel tensor)
        y = torch.empty((N, H, W, Cout), device=x.device, dtype=x.dtype)
        for oh in range(Cout // 32):
            sl = slice(32 * oh, 32 * oh + 32)
            ab = torch.empty(64, device=x.device, dtype=torch.float32)
            lib.bn_eval_ab(32, bn[0][sl], bn[1][sl], float(bn[4]), bn[2][sl], bn[3][sl], torch.empty(64, device=x.device, dtype=torch.float32), ab)
            wp = torch.empty(KH * KW * 1024, device=x.device, dtype=torch.bfloat16)
            lib.conv32_pack_weights_sub(w4, wp, KH, KW, 0, Cin, 32 * oh, 0)
            lib.conv32_fwd_strided_affine(x4, wp, bias[sl] if bias is not None else None, y, N, H, W, KH, KW, ph, pw, Cin, 0, Cout, 32 * oh, ab, pre, post)
        return y.squeeze(2) if tok else y
    if not (mfma or pwk):
        y = conv2d(x, w, bias, stride, pad)
        if bn is not None:
            return batchnorm(y, bn[0], bn[1], bn[2], bn[3], None, bn[4], 0.1, pre_act, post_act, training=False)
        if pre != 0:
            y = act(y, pre_act)
        return act(y, post_act) if post != 0 else y
    ab = None
    if bn is not None:
        ab = torch.empty(2 * Cout, device=x.device, dtype=torch.float32)
        lib.bn_eval_ab(Cout, bn[0], bn[1], float(bn[4]), bn[2], bn[3], torch.empty(2 * Cout, device=x.device, dtype=torch.float32), ab)
    y = torch.empty((N, H, W, Cout), device=x.device, dtype=x.dtype)
    if pwk:
        lib.pw_fwd_affine(x4, w4, bias, y, N * H * W, Cin, Cout, ab, pre, post, dtype_code(y.dtype))
    else:
        wp = torch.empty(KH * KW * 1024, device=x.device, dtype=torch.bfloat16)
        lib.conv32_pack_weights(w4, wp, KH, KW, 0)
        lib.conv32_fwd_affine(x4, wp, bias, y, N, H, W, KH, KW, ph, pw, ab, pre, post)
    return y.squeeze(2) if tok else y


def _eval_ab(bn, C, device):
    ab = torch.empty(2 * C, device=device, dtype=torch.float32)
    lib.bn_eval_ab(C, bn[0], bn[1], float(bn[4]), bn[2], bn[3], torch.empty(2 * C, device=device, dtype=torch.float32), ab)
    return ab


def conv_bn_residual_eval_ok(x, w, bias, stride, pad, pre_act, post_act, residual):
    """inference: residual + BN_eval(conv1x1(x)) as ONE GEMM (tcct_pw_fwd_affine_residual): bf16 rows, square 64 / 96 / 128, no activation"""
    return (INFER_FUSE and not torch.is_grad_enabled() and residual is not None and x.dtype == torch.bfloat16 and x.dim() == 4 and x.is_cuda and stride == 1
            and tuple(pad) == (0, 0) and ACT[pre_act] == 0 and ACT[post_act] == 0 and tuple(w.shape[2:]) == (1, 1) and w.shape[0] == w.shape[1] == x.shape[-1]
            and x.shape[-1] in (64, 96, 128) and residual.shape == x.shape and residual.dtype == x.dtype and x.numel() * 2 < 2 ** 31)


def conv_bn_residual_eval(x, w, bias, bn, residual):
    _chk(x, w, bias, residual)
    C = x.shape[-1]
    y = torch.empty_like(x)
    lib.pw_fwd_affine_residual(x, w, bias, _eval_ab(bn, C, x.device), residual, y, x.numel() // C, C, C)
    return y


def invres_tail_eval_ok(y_dw, w, bias, residual):
    """inference: x + BN(conv2(hswish(BN(y_dw)))) as ONE GEMM (tcct_pw_fwd_xaff_affine_residual)"""
    return (INFER_FUSE and not torch.is_grad_enabled() and y_dw.dtype == torch.bfloat16 and y_dw.dim() == 4 and y_dw.is_cuda and y_dw.shape[-1] in (64, 96, 128)
            and tuple(w.shape) == (y_dw.shape[-1], y_dw.shape[-1], 1, 1) and residual.shape == y_dw.shape and residual.dtype == y_dw.dtype and y_dw.numel() * 2 < 2 ** 31)


def invres_tail_eval(y_dw, bn_norm, w, bias, bn2, residual):
    _chk(y_dw, w, bias, residual)
    C = y_dw.shape[-1]
    out = torch.empty_like(y_dw)
    lib.pw_fwd_xaff_affine_residual(y_dw, _eval_ab(bn_norm, C, y_dw.device), w, bias, _eval_ab(bn2, C, y_dw.device), residual, out, y_dw.numel() // C, C, C)
    return out


def conv1x1_cat2_bn_act_eval_ok(x1, x2, w, post_act):
    """inference: post_act(BN_eval(conv1x1(cat[x1, x2]))) without the concatenation (tcct_pw_fwd_cat2_affine): two 64-channel bf16 tensors, 96 outputs"""
    return (INFER_FUSE and not torch.is_grad_enabled() and x1.dtype == torch.bfloat16 and x1.dim() == 4 and x1.is_cuda and x1.shape[-1] == 64 and x2.shape == x1.shape
            and x2.dtype == x1.dtype and tuple(w.shape) == (96, 128, 1, 1) and x1.numel() * 4 < 2 ** 31)


def conv1x1_cat2_bn_act_eval(x1, x2, w, bn, post_act):
    _chk(x1, x2, w)
    y = torch.empty(x1.shape[:-1] + (96,), device=x1.device, dtype=x1.dtype)
    lib.pw_fwd_cat2_affine(x1, x2, w, None, _eval_ab(bn, 96, x1.device), ACT[post_act], y, x1.numel() // 64, 128, 96)
    return y


def bn2_add_act_eval(xa, bnA, xb, bnB, pre_act='lrelu', act_kind='gelu'):
    """Inference only: act(BN_A(pre(xa)) + BN_B(pre(xb))) with running statistics, one pass (CrossCNNBlock junction).
    bnA/bnB: (gamma, beta, running_mean, running_var, eps)."""
    if torch.is_grad_enabled() and (xa.requires_grad or xb.requires_grad):
        raise TcctError('bn2_add_act_eval is inference-only')
    _chk(xa, xb)
    C = xa.shape[-1]
    M = xa.numel() // C
    abs_ = []
    for bn in (bnA, bnB):
        ab = torch.empty(2 * C, device=xa.device, dtype=torch.float32)
        lib.bn_eval_ab(C, bn[0], bn[1], float(bn[4]), bn[2], bn[3], torch.empty(2 * C, device=xa.device, dtype=torch.float32), ab)
        abs_.append(ab)
    y = torch.empty_like(xa)
    lib.bn2_add_act_fwd(xa, xb, y, M, C, abs_[0], abs_[1], ACT[pre_act], ACT[act_kind], dtype_code(xa.dtype))
    return y


class _LinearResidual(_FastFunction):
    """res + scale[b] * (x W^T + bias) on tokens [B,N,C] (Mlp.fc2 + DropPath scale + residual add in the GEMM epilogue)"""

    @staticmethod
    def forward(ctx, x, w, bias, res, scale):
        _chk(x, w, bias, res, scale)
        B, Nt, K = x.shape
        Cout = w.shape[0]
        y = torch.empty((B, Nt, Cout), device=x.device, dtype=x.dtype)
        lib.pw_fwd_residual(x, w, bias, res, scale, Nt, y, None, B * Nt, K, Cout)
        ctx.save_for_backward(x, w)
        ctx.scale, ctx.params = scale, (w, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        wsrc, bsrc = ctx.params
        dy = _c(dy)
        B, Nt, K = x.shape
        Cout, M = w.shape[0], B * Nt
        dz = dy
        if ctx.scale is not None:
            dz = torch.empty_like(dy)
            lib.scale_rows(dy, ctx.scale, dz, B, dy.numel() // B, dtype_code(dy.dtype))
        dx = torch.empty_like(x)
        if FUSED_PW_BWD and _pw_bwd_ok(x, dz, K, K, Cout, 1, 1, 1, 0, 0):
            dw = _grad_out(wsrc, tuple(w.shape))
            db = _grad_out(bsrc)
            lib.pw_bwd(x, dz, w, None, dx, dw, db, M, K, Cout)
            return dx, _ret(dw, wsrc), _ret(db, bsrc), dy, None
        lib.pw_fwd(dz, w, None, dx, M, Cout, K, 1, dtype_code(x.dtype))
        with _wgrad_stream(_slot_written(wsrc, bsrc), x, dz):
            dw = _grad_out(wsrc, tuple(w.shape))
            db = _grad_out(bsrc)
            lib.pw_wgrad(x, dz, dw, db, M, K, Cout)
        return dx, _ret(dw, wsrc), _ret(db, bsrc), dy, None


MLP_GELU_FUSE = True         # False: GELU as its own pass between fc1 and fc2 (round-3 form; A/B timing)


class _GeluLinearResidual(_FastFunction):
    """res + scale[b] * (gelu(x1) W^T + bias) on tokens [B,N,C] with x1 the PRE-activation of Mlp.fc1 (reference nets/tcct.py:29-53,468): GELU is applied
    while the GEMM kernels stage their tiles (tcct_pw_fwd_gelu_residual / tcct_pw_bwd_gelu), so neither h = gelu(x1) nor dh exist in HBM -- bit-identical
    to act(x1, 'gelu') followed by linear_residual (same formula, same bf16 roundings), two tensor passes fewer forward and three fewer backward."""

    @staticmethod
    def forward(ctx, x1, w, bias, res, scale):
        _chk(x1, w, bias, res, scale)
        B, Nt, K = x1.shape
        Cout = w.shape[0]
        y = torch.empty((B, Nt, Cout), device=x1.device, dtype=x1.dtype)
        lib.pw_fwd_gelu_residual(x1, w, bias, res, scale, Nt, y, B * Nt, K, Cout)
        ctx.save_for_backward(x1, w)
        ctx.scale, ctx.params = scale, (w, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, w = ctx.saved_tensors
        wsrc, bsrc = ctx.params
        dy = _c(dy)
        B, Nt, K = x1.shape
        Cout, M = w.shape[0], B * Nt
        dz = dy
        if ctx.scale is not None:
            dz = torch.empty_like(dy)
            lib.scale_rows(dy, ctx.scale, dz, B, dy.numel() // B, dtype_code(dy.dtype))
        dx1 = torch.empty_like(x1)
        dw = _grad_out(wsrc, tuple(w.shape))
        db = _grad_out(bsrc)
        lib.pw_bwd_gelu(x1, dz, w, dx1, dw, db, M, K, Cout)
        return dx1, _ret(dw, wsrc), _ret(db, bsrc), dy, None


class _MlpTail(_FastFunction):
    """t2 = t1 + scale[b] * fc2(gelu(fc1(cur2))) with cur2 = LayerNorm2(t1) as stored by the node in front (ln_metapool_residual_ln): MHCABlock's second half
    (reference nets/tcct.py:466-468) as ONE autograd node, so that its backward can run LayerNorm2's backward in fc1's input-gradient epilogue
    (tcct_pw_bwd_lnb): the gradient of cur2 is never written, the stand-alone LayerNorm backward pass does not run.  The node returns the COMPLETE gradient of
    t1 (residual path + through LayerNorm2) and none for cur2; dgamma2 / dbeta2 come out of the same kernel."""

    @staticmethod
    def forward(ctx, t1, cur2, mr2, g2, b2, w1, bias1, w2, bias2, scale):
        _chk(t1, cur2, mr2, g2, b2, w1, bias1, w2, bias2, scale)
        B, Nt, K = t1.shape
        M = B * Nt
        y1 = torch.empty_like(t1)
        lib.pw_fwd(cur2, w1, bias1, y1, M, K, K, 0, dtype_code(t1.dtype))
        t2 = torch.empty_like(t1)
        lib.pw_fwd_gelu_residual(y1, w2, bias2, t1, scale, Nt, t2, M, K, K)
        ctx.save_for_backward(t1, cur2, mr2, y1, g2, w1, w2)
        ctx.scale = scale
        ctx.params = (g2, b2, w1, bias1, w2, bias2)
        return t2

    @staticmethod
    def backward(ctx, dt2):
        t1, cur2, mr2, y1, g2, w1, w2 = ctx.saved_tensors
        g2p, b2p, w1p, bias1p, w2p, bias2p = ctx.params
        dt2 = _as(dt2, t1.dtype)
        B, Nt, K = t1.shape
        M = B * Nt
        dz = dt2
        if ctx.scale is not None:
            dz = torch.empty_like(dt2)
            lib.scale_rows(dt2, ctx.scale, dz, B, dt2.numel() // B, dtype_code(dt2.dtype))
        w1s = w1p if hasattr(w1p, '_grad_slot') or w1p._base is None else w1p._base
        w2s = w2p if hasattr(w2p, '_grad_slot') or w2p._base is None else w2p._base
        dy1 = torch.empty_like(y1)
        dw2, db2 = _grad_out(w2s, tuple(w2.shape)), _grad_out(bias2p)
        lib.pw_bwd_gelu(y1, dz, w2, dy1, dw2, db2, M, K, K)
        dt1 = torch.empty_like(t1)
        dw1, db1 = _grad_out(w1s, tuple(w1.shape)), _grad_out(bias1p)
        dg2, dbeta2 = _grad_out(g2p), _grad_out(b2p)
        lib.pw_bwd_lnb(cur2, dy1, w1, t1, mr2, g2, dt2, dt1, dw1, db1, dg2, dbeta2, M, K, K)
        return (dt1, None, None, _ret(dg2, g2p), _ret(dbeta2, b2p), _ret(dw1, w1s), _ret(db1, bias1p), _ret(dw2, w2s), _ret(db2, bias2p), None)


MLP_TAIL_FUSE = True      # False: LayerNorm2 keeps its own backward pass (A/B timing)


def mlp_tail_ok(t1, cur2, w1, bias1, w2, bias2):
    return (MLP_TAIL_FUSE and MLP_GELU_FUSE and FUSED_PW_BWD and torch.is_grad_enabled() and t1.dtype == torch.bfloat16 and t1.dim() == 3 and t1.shape[-1] == 64
            and cur2.shape == t1.shape and tuple(w1.shape) == (64, 64) and tuple(w2.shape) == (64, 64) and bias1 is not None and bias2 is not None
            and t1.is_contiguous() and cur2.is_contiguous() and t1.numel() * 2 < 2 ** 31)


def mlp_tail(t1, cur2, mr2, g2, b2, w1, bias1, w2, bias2, scale=None):
    """t1 + scale[b] * fc2(gelu(fc1(cur2))), cur2 = LayerNorm2(t1) with statistics mr2 (from ln_metapool_residual_ln); check mlp_tail_ok first"""
    return _MlpTail.apply(t1, cur2, mr2, g2, b2, w1, bias1, w2, bias2, scale)


def gelu_linear_residual_ok(x1, w, bias, res):
    return (MLP_GELU_FUSE and FUSED_PW_BWD and torch.is_grad_enabled() and x1.dtype == torch.bfloat16 and x1.dim() == 3 and w.dim() == 2
            and x1.shape[-1] in (64, 96) and w.shape[0] == x1.shape[-1] and w.shape[1] == x1.shape[-1] and bias is not None
            and res.shape == x1.shape and x1.numel() * 2 < 2 ** 31)


def gelu_linear_residual(x1, w, bias, res, scale=None):
    """res + scale[b] * Linear(gelu(x1)); check gelu_linear_residual_ok first"""
    return _GeluLinearResidual.apply(x1, w, bias, res, scale)


class _Conv1x1AndSum(_FastFunction):
    """(d, d + res) with d = conv1x1(x): both written by one GEMM epilogue (decoder `post` convolution + the `x_i + y_i` that follows)"""

    @staticmethod
    def forward(ctx, x, w, bias, res):
        ctx.set_materialize_grads(False)      # an unused output must arrive as None in backward(), not as a zero-filled tensor
        _chk(x, w, bias, res)
        N_, H, W_, K = x.shape
        Cout = w.shape[0]
        d = torch.empty((N_, H, W_, Cout), device=x.device, dtype=x.dtype)
        s_ = torch.empty_like(d)
        lib.pw_fwd_residual(x, w, bias, res, None, 1, s_, d, N_ * H * W_, K, Cout)
        ctx.save_for_backward(x, w)
        ctx.params = (w, bias)
        return d, s_

    @staticmethod
    def backward(ctx, gd, gs):
        x, w = ctx.saved_tensors
        wsrc, bsrc = ctx.params
        if gd is None and gs is None:
            return None, None, None, None
        if gd is None or gs is None:
            dz = _c(gd if gs is None else gs)
        else:
            dz = torch.empty_like(gs)
            lib.add(_c(gd), _c(gs), dz, dz.numel(), dtype_code(dz.dtype))
        N_, H, W_, K = x.shape
        Cout, M = w.shape[0], N_ * H * W_
        dx = torch.empty_like(x)
        if FUSED_PW_BWD and _pw_bwd_ok(x, dz, K, K, Cout, 1, 1, 1, 0, 0):
            dw = _grad_out(wsrc, tuple(w.shape))
            db = _grad_out(bsrc)
            lib.pw_bwd(x, dz, w, None, dx, dw, db, M, K, Cout)
            return dx, _ret(dw, wsrc), _ret(db, bsrc), (None if gs is None else _c(gs))
        lib.pw_fwd(dz, w, None, dx, M, Cout, K, 1, dtype_code(x.dtype))
        with _wgrad_stream(_slot_written(wsrc, bsrc), x, dz):
            dw = _grad_out(wsrc, tuple(w.shape))
            db = _grad_out(bsrc)
            lib.pw_wgrad(x, dz, dw, db, M, K, Cout)
        return dx, _ret(dw, wsrc), _ret(db, bsrc), (None if gs is None else _c(gs))


class _UpSkipConv(_FastFunction):
    """decoder block tail (MPUpBlock, reference tcct.py:908-914 + the `x_i + y_i` of FTC.forward :1028-1031) as one node:
         u = resize_x2(y) + skip;  d = conv1x1(u);  s = d + skip        -> (d, s)
    One node instead of bilinear + conv1x1_and_sum so that the skip tensor's two gradients (through u and through s) leave the
    input-gradient GEMM already summed (tcct_pw_dgrad_residual) -- autograd would otherwise add them in a separate pass."""

    @staticmethod
    def forward(ctx, y, skip, w, bias, align, want_plain=True):
        ctx.set_materialize_grads(False)      # an unused output must arrive as None in backward(), not as a zero-filled tensor
        _chk(y, skip, w, bias)
        N_, H, W_, C = y.shape
        _, Ho, Wo, _ = skip.shape
        Cout = w.shape[0]
        u = torch.empty_like(skip)
        lib.bilinear_add_fwd(y, skip, u, N_, H, W_, C, Ho, Wo, int(align), dtype_code(y.dtype))
        s_ = torch.empty((N_, Ho, Wo, Cout), device=y.device, dtype=y.dtype)
        # want_plain=False (the LAST decoder block: FTC.forward reads only x_0 + y_0, tcct.py:1031-1035): the 452 MB `d` of the bench shape is never written
        d = torch.empty_like(s_) if want_plain else None
        lib.pw_fwd_residual(u, w, bias, skip, None, 1, s_, d, N_ * Ho * Wo, C, Cout)
        ctx.save_for_backward(u, w)
        ctx.params = (w, bias)
        ctx.cfg = (N_, H, W_, C, Ho, Wo, int(align))
        if d is None:
            d = s_.new_empty(0)
            ctx.mark_non_differentiable(d)
        return d, s_

    @staticmethod
    def backward(ctx, gd, gs):
        u, w = ctx.saved_tensors
        wsrc, bsrc = ctx.params
        N_, H, W_, C, Ho, Wo, align = ctx.cfg
        if gd is None and gs is None:
            return None, None, None, None, None, None
        Cout, M = w.shape[0], N_ * Ho * Wo
        if (gd is not None and gs is not None and DY2_FUSE and FUSED_PW_BWD and C == 32 and Cout == 32 and u.dtype == torch.bfloat16
                and _pw_bwd_ok(u, _c(gs), C, C, Cout, 1, 1, 1, 0, 0)):
            # the two output gradients are summed while the backward kernel stages its tiles: no add pass
            gd, gs = _c(gd), _c(gs)
            du, dskip = torch.empty_like(u), torch.empty_like(u)
            dw, db = _grad_out(wsrc, tuple(w.shape)), _grad_out(bsrc)
            lib.pw_bwd_residual2_sum(u, gd, gs, w, gs, dskip, du, dw, db, M, C, Cout)
            dy = torch.empty((N_, H, W_, C), device=u.device, dtype=u.dtype)
            lib.bilinear_bwd(du, dy, N_, H, W_, C, Ho, Wo, align, dtype_code(u.dtype))
            return dy, dskip, _ret(dw, wsrc), _ret(db, bsrc), None, None
        if gd is None or gs is None:
            dz = _c(gd if gs is None else gs)
        else:
            dz = torch.empty_like(gs)
            lib.add(_c(gd), _c(gs), dz, dz.numel(), dtype_code(dz.dtype))
        du = torch.empty_like(u)
        fused = FUSED_PW_BWD and _pw_bwd_ok(u, dz, C, C, Cout, 1, 1, 1, 0, 0)
        if fused:       # both input gradients and the weight / bias gradients from ONE pass over dz
            dw = _grad_out(wsrc, tuple(w.shape))
            db = _grad_out(bsrc)
            if gs is not None:
                dskip = torch.empty_like(u)
                lib.pw_bwd_residual2(u, dz, w, _c(gs), dskip, du, dw, db, M, C, Cout)
            else:
                lib.pw_bwd(u, dz, w, None, du, dw, db, M, C, Cout)
                dskip = du
        elif gs is not None:
            dskip = torch.empty_like(u)
            lib.pw_dgrad_residual(dz, w, _c(gs), dskip, du, M, Cout, C)
        else:
            lib.pw_fwd(dz, w, None, du, M, Cout, C, 1, dtype_code(u.dtype))
            dskip = du
        dy = torch.empty((N_, H, W_, C), device=u.device, dtype=u.dtype)
        lib.bilinear_bwd(du, dy, N_, H, W_, C, Ho, Wo, align, dtype_code(u.dtype))
        if not fused:
            with _wgrad_stream(_slot_written(wsrc, bsrc), u, dz):
                dw = _grad_out(wsrc, tuple(w.shape))
                db = _grad_out(bsrc)
                lib.pw_wgrad(u, dz, dw, db, M, C, Cout)
        return dy, dskip, _ret(dw, wsrc), _ret(db, bsrc), None, None


DY2_FUSE = True      # False: the decoder blocks add their two output gradients in a separate pass (A/B timing)


def up_skip_conv(y, skip, w, bias, align_corners=True, want_plain=True):
    """(d, d + skip) with d = conv1x1(resize(y -> skip's size) + skip); the fused node needs bf16 with square 1x1 weights.
    want_plain=False: d is not needed (returned as None)"""
    ok = (y.dtype == torch.bfloat16 and skip.dtype == y.dtype and y.dim() == 4 and y.shape[-1] % 32 == 0 and y.shape[-1] <= 160
          and w.shape[0] == y.shape[-1] and w.shape[1] == y.shape[-1] and tuple(w.shape[2:]) == (1, 1) and bias is not None
          and skip.shape[-1] == y.shape[-1] and tuple(skip.shape[1:3]) != tuple(y.shape[1:3]) and torch.is_grad_enabled())
    if not ok:
        return conv1x1_and_sum(bilinear(y, tuple(skip.shape[1:3]), align_corners, residual=skip), w, bias, skip)
    d, s_ = _UpSkipConv.apply(y, skip, w, bias, bool(align_corners), bool(want_plain))
    return (d if want_plain else None), s_


TAIL_COMPOSE = True      # False: resize + add, post convolution (+ sum), t324 as three kernels (round-3 form; A/B timing)


class _UpSkipConvT32(_FastFunction):
    """g0 = t32(post(up(y) + skip) + skip) for the LAST decoder block (reference nets/tcct.py:908-914, :1031, :1035-1040) as ONE 64 -> 32 GEMM over the
    never-materialised concatenation [up(y) | skip] with composed weights (csrc/decoder_tail.hip): u, d0 and s0 are not written, the backward pass is
    one fused kernel + a 32 x 32 de-composition of the weight gradients."""

    @staticmethod
    def forward(ctx, y, skip, w1, b1, w2, b2, align):
        _chk(y, skip, w1, b1, w2, b2)
        N_, H, W_, C = y.shape
        _, Ho, Wo, _ = skip.shape
        dev = y.device
        v = torch.empty_like(skip)
        lib.bilinear_fwd(y, v, N_, H, W_, C, Ho, Wo, int(align), dtype_code(y.dtype))
        wc = torch.empty((32, 64), device=dev, dtype=torch.float32)
        c = torch.empty(32, device=dev, dtype=torch.float32)
        lib.tail_compose(w1, b1, w2, b2, wc, c)
        g = torch.empty_like(skip)
        lib.pw_fwd_cat2(v, skip, 32, wc, c, g, N_ * Ho * Wo, 64, 32, None, 0)
        ctx.save_for_backward(v, skip, wc)
        ctx.params = (w1, b1, w2, b2)
        ctx.cfg = (N_, H, W_, C, Ho, Wo, int(align))
        return g

    @staticmethod
    def backward(ctx, dg):
        v, skip, wc = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.params
        N_, H, W_, C, Ho, Wo, align = ctx.cfg
        dg = _as(dg, v.dtype)
        dev = v.device
        M = N_ * Ho * Wo
        dv, dskip = torch.empty_like(v), torch.empty_like(skip)
        dwc = ZERO.get((32, 64), torch.float32, dev)
        dc = ZERO.get((32,), torch.float32, dev)
        lib.pw_bwd_cat2_bias(v, skip, dg, wc, dv, dskip, dwc, dc, M, 64, 32)
        dy = torch.empty((N_, H, W_, C), device=dev, dtype=v.dtype)
        lib.bilinear_bwd(dv, dy, N_, H, W_, C, Ho, Wo, align, dtype_code(v.dtype))
        dw1, db1, dw2, db2 = _grad_out(w1, tuple(w1.shape)), _grad_out(b1), _grad_out(w2, tuple(w2.shape)), _grad_out(b2)
        lib.tail_compose_bwd(w1, b1, w2, dwc, dc, dw1, db1, dw2, db2)
        return dy, dskip, _ret(dw1, w1), _ret(db1, b1), _ret(dw2, w2), _ret(db2, b2), None


def up_skip_conv_t32_ok(y, skip, w1, b1, w2, b2):
    return (TAIL_COMPOSE and FUSED_PW_BWD and y.dtype == torch.bfloat16 and skip.dtype == y.dtype and y.dim() == 4       # (the caller requires a train-mode block)
            and y.shape[-1] == 32 and skip.shape[-1] == 32 and tuple(skip.shape[1:3]) == (2 * y.shape[1], 2 * y.shape[2])
            and tuple(w1.shape) == (32, 32, 1, 1) and tuple(w2.shape) == (32, 32, 1, 1) and b1 is not None and b2 is not None
            and w1.is_contiguous() and w2.is_contiguous() and skip.numel() * 2 < 2 ** 31)


def up_skip_conv_t32(y, skip, w1, b1, w2, b2, align_corners=True):
    """t32(post(resize_x2(y) + skip) + skip) with post = (w1, b1), t32 = (w2, b2), both 1x1 32 -> 32; check up_skip_conv_t32_ok first"""
    return _UpSkipConvT32.apply(y, skip, w1, b1, w2, b2, bool(align_corners))


class _UpSkipConvT32Aux(_FastFunction):
    """logits0 = aux0(t32(post(up(y) + skip) + skip)) (reference nets/tcct.py:908-914,1031,1035-1041) as ONE 64 -> n_class GEMM with fp32 output over
    [up(y) | skip] (csrc/decoder_tail.hip, `compose3`): for steps in which nothing else reads g0 -- the feature-polarization loss is off.  g0 and dg0
    never exist; the backward pass uses the small-N kernels of the aux heads on the two halves.  Returns (logits, v): v = up(y) (no gradient) lets
    `FTC.feats` rebuild g0 on demand."""

    @staticmethod
    def forward(ctx, y, skip, w1, b1, w2, b2, w3, b3, align):
        _chk(y, skip, w1, b1, w2, b2, w3, b3)
        N_, H, W_, C = y.shape
        _, Ho, Wo, _ = skip.shape
        dev, nc = y.device, w3.shape[0]
        v = torch.empty_like(skip)
        lib.bilinear_fwd(y, v, N_, H, W_, C, Ho, Wo, int(align), dtype_code(y.dtype))
        wcc = torch.empty((nc, 64), device=dev, dtype=torch.float32)
        wa, wb = torch.empty((nc, 32, 1, 1), device=dev, dtype=torch.float32), torch.empty((nc, 32, 1, 1), device=dev, dtype=torch.float32)
        ccc = torch.empty(nc, device=dev, dtype=torch.float32)
        lib.tail_compose3(w1, b1, w2, b2, w3, b3, nc, wcc, wa, wb, ccc)
        lg = torch.empty((N_, Ho, Wo, nc), device=dev, dtype=torch.float32)
        lib.pw_fwd_cat2_f32(v, skip, 32, wcc, ccc, lg, N_ * Ho * Wo, 64, nc)
        ctx.save_for_backward(v, skip, wa, wb)
        ctx.params = (w1, b1, w2, b2, w3, b3)
        ctx.cfg = (N_, H, W_, C, Ho, Wo, int(align), nc)
        ctx.mark_non_differentiable(v)
        return lg, v

    @staticmethod
    def backward(ctx, dl, _dv):
        v, skip, wa, wb = ctx.saved_tensors
        w1, b1, w2, b2, w3, b3 = ctx.params
        N_, H, W_, C, Ho, Wo, align, nc = ctx.cfg
        dl = _as(dl, torch.float32)
        dev = v.device
        M = N_ * Ho * Wo
        dv, dskip = torch.empty_like(v), torch.empty_like(skip)
        F32_, BF_ = dtype_code(torch.float32), dtype_code(v.dtype)
        lib.conv2d_dgrad(dl, wa, dv, N_, Ho, Wo, 32, nc, 1, 1, 0, 0, F32_, BF_)
        lib.conv2d_dgrad(dl, wb, dskip, N_, Ho, Wo, 32, nc, 1, 1, 0, 0, F32_, BF_)
        dy = torch.empty((N_, H, W_, C), device=dev, dtype=v.dtype)
        lib.bilinear_bwd(dv, dy, N_, H, W_, C, Ho, Wo, align, BF_)
        params = (w1, b1, w2, b2, w3, b3)
        with _wgrad_stream(_slot_written(*params), v, skip, dl, wa, wb):
            dwa, dwb = ZERO.get((nc, 32), torch.float32, dev), ZERO.get((nc, 32), torch.float32, dev)
            dccc = ZERO.get((nc,), torch.float32, dev)
            lib.pw_wgrad_smalln(v, dl, dwa, dccc, M, 32, nc, BF_, F32_)
            lib.pw_wgrad_smalln(skip, dl, dwb, None, M, 32, nc, BF_, F32_)
            outs = [_grad_out(p, tuple(p.shape)) for p in params]
            lib.tail_compose3_bwd(w1, b1, w2, b2, w3, nc, dwa, dwb, dccc, *outs)
        return (dy, dskip) + tuple(_ret(o, p) for o, p in zip(outs, params)) + (None,)


class _UpSkipConvT32AuxLow(_FastFunction):
    """The same logits0 = aux0(t32(post(up(y) + skip) + skip)) with the resize COMMUTED behind the 1x1 convolution (round 4): bilinear interpolation is linear
    and acts per channel, so  W up(y) = up(W y):
        logits0 = up(Wa y) + Wb skip + c,      Wa = W3 W2 W1 [C x 32] applied at the LOW resolution,  Wb = W3 (W2 W1 + W2),  c = W3 (W2 b1 + b2) + b3
    z = Wa y is n_class channels at a quarter of the pixels (fp32); the resized 32-channel tensor v = up(y) (452 MB at the bench shape) is never written or
    read, and neither is its gradient: backward = bilinear^T on n_class channels, two small-N input-gradient kernels, two small-N weight-gradient
    kernels (one of them at the low resolution) and the de-composition of `compose3`."""

    @staticmethod
    def forward(ctx, y, skip, w1, b1, w2, b2, w3, b3, align):
        _chk(y, skip, w1, b1, w2, b2, w3, b3)
        N_, H, W_, C = y.shape
        _, Ho, Wo, _ = skip.shape
        dev, nc = y.device, w3.shape[0]
        F32_ = dtype_code(torch.float32)
        wcc = torch.empty((nc, 64), device=dev, dtype=torch.float32)
        wa8 = torch.zeros((8, 32, 1, 1), device=dev, dtype=torch.float32)        # rows >= n_class stay zero: the low-resolution product has 8 channels (16-byte taps)
        wa, wb = wa8[:nc], torch.empty((nc, 32, 1, 1), device=dev, dtype=torch.float32)
        ccc = torch.empty(nc, device=dev, dtype=torch.float32)
        lib.tail_compose3(w1, b1, w2, b2, w3, b3, nc, wcc, wa, wb, ccc)
        lg = torch.empty((N_, Ho, Wo, nc), device=dev, dtype=torch.float32)
        if HEAD_UPADD and align and Ho == 2 * H and Wo == 2 * W_:
            z8 = torch.empty((N_, H, W_, 8), device=dev, dtype=torch.float32)
            lib.pw_fwd(y, wa8, None, z8, N_ * H * W_, 32, 8, 0, F32_)
            lib.pw_fwd_f32_upadd(skip, wb, ccc, lg, N_, H, W_, 32, nc, z8)             # up(z) added in the epilogue of the full-resolution GEMM
        else:
            z = torch.empty((N_, H, W_, nc), device=dev, dtype=torch.float32)
            lib.pw_fwd(y, wa, None, z, N_ * H * W_, 32, nc, 0, F32_)
            lib.pw_fwd(skip, wb, ccc, lg, N_ * Ho * Wo, 32, nc, 0, F32_)
            lib.bilinear_add_fwd(z, lg, lg, N_, H, W_, nc, Ho, Wo, int(align), F32_)   # in place: every element reads its own addend, then is written
        ctx.save_for_backward(y, skip, wa, wb)
        ctx.params = (w1, b1, w2, b2, w3, b3)
        ctx.cfg = (N_, H, W_, C, Ho, Wo, int(align), nc)
        return lg

    @staticmethod
    def backward(ctx, dl):
        y, skip, wa, wb = ctx.saved_tensors
        N_, H, W_, C, Ho, Wo, align, nc = ctx.cfg
        dl = _as(dl, torch.float32)
        dev = y.device
        F32_, BF_ = dtype_code(torch.float32), dtype_code(y.dtype)
        dz = torch.empty((N_, H, W_, nc), device=dev, dtype=torch.float32)
        lib.bilinear_bwd(dl, dz, N_, H, W_, nc, Ho, Wo, align, F32_)
        dy, dskip = torch.empty_like(y), torch.empty_like(skip)
        lib.conv2d_dgrad(dz, wa, dy, N_, H, W_, 32, nc, 1, 1, 0, 0, F32_, BF_)
        lib.conv2d_dgrad(dl, wb, dskip, N_, Ho, Wo, 32, nc, 1, 1, 0, 0, F32_, BF_)
        params = ctx.params
        with _wgrad_stream(_slot_written(*params), y, skip, dl, dz, wa, wb):
            dwa, dwb = ZERO.get((nc, 32), torch.float32, dev), ZERO.get((nc, 32), torch.float32, dev)
            dccc = ZERO.get((nc,), torch.float32, dev)
            lib.pw_wgrad_smalln(y, dz, dwa, None, N_ * H * W_, 32, nc, BF_, F32_)
            lib.pw_wgrad_smalln(skip, dl, dwb, dccc, N_ * Ho * Wo, 32, nc, BF_, F32_)
            outs = [_grad_out(p, tuple(p.shape)) for p in params]
            lib.tail_compose3_bwd(params[0], params[1], params[2], params[3], params[4], nc, dwa, dwb, dccc, *outs)
        return (dy, dskip) + tuple(_ret(o, p) for o, p in zip(outs, params)) + (None,)


HEAD_UPADD = True       # False: the resized low-resolution product is added by its own pass (A/B timing)
TAIL_AUX_LOW = True   # False: the composed head reads the resized 32-channel tensor (the first round-4 form; A/B timing)


def up_skip_conv_t32_aux_low(y, skip, w1, b1, w2, b2, w3, b3, align_corners=True):
    """aux(t32(post(resize_x2(y) + skip) + skip)) as fp32 NHWC logits, the resize taken of the n_class-channel product; check up_skip_conv_t32_aux_ok first"""
    return _UpSkipConvT32AuxLow.apply(y, skip, w1, b1, w2, b2, w3, b3, bool(align_corners))


def up_skip_conv_t32_from_y(y, skip, w1, b1, w2, b2, align_corners=True):
    """g0 of the composed tail from the un-resized y (no gradient): what `FTC.feats` needs when the step itself never resized y"""
    with torch.no_grad():
        N_, H, W_, C = y.shape
        v = torch.empty_like(skip)
        lib.bilinear_fwd(y, v, N_, H, W_, C, skip.shape[1], skip.shape[2], int(align_corners), dtype_code(y.dtype))
    return up_skip_conv_t32_from_v(v, skip, w1, b1, w2, b2)


TAIL_AUX = True       # False: the composed tail stops at g0, aux0 stays its own kernels (A/B timing)


def up_skip_conv_t32_aux_ok(y, skip, w1, b1, w2, b2, w3, b3):
    return (TAIL_AUX and up_skip_conv_t32_ok(y, skip, w1, b1, w2, b2) and w3.dim() == 4 and tuple(w3.shape[1:]) == (32, 1, 1) and 2 <= w3.shape[0] <= 8
            and b3 is not None and w3.is_contiguous())


def up_skip_conv_t32_aux(y, skip, w1, b1, w2, b2, w3, b3, align_corners=True):
    """(aux(t32(post(resize_x2(y) + skip) + skip)) as fp32 NHWC logits, resize_x2(y)); check up_skip_conv_t32_aux_ok first"""
    return _UpSkipConvT32Aux.apply(y, skip, w1, b1, w2, b2, w3, b3, bool(align_corners))


def up_skip_conv_t32_from_v(v, skip, w1, b1, w2, b2):
    """g0 of the composed tail from an already resized v (no gradient): what `FTC.feats` needs when the step itself skipped g0"""
    with torch.no_grad():
        wc = torch.empty((32, 64), device=v.device, dtype=torch.float32)
        c = torch.empty(32, device=v.device, dtype=torch.float32)
        lib.tail_compose(w1, b1, w2, b2, wc, c)
        g = torch.empty_like(skip)
        lib.pw_fwd_cat2(v, skip, 32, wc, c, g, v.numel() // 32, 64, 32, None, 0)
    return g


class _HeadThroughT32(_FastFunction):
    """logits_i = aux_i(t32x(s_i)) (reference nets/tcct.py:1036-1044, levels 1-3) as ONE 32 -> n_class GEMM with fp32 output and the composed weight
    Wa Wt (csrc/decoder_tail.hip, `head_compose`): for steps in which nothing else reads g_i = t32x(s_i) -- the feature-polarization loss is off.
    g_i and its gradient never exist; the backward pass runs the aux head's small-N kernels with the composed weight and de-composes the gradients."""

    @staticmethod
    def forward(ctx, s, wt, bt, wa, ba):
        _chk(s, wt, bt, wa, ba)
        N_, H, W_, _ = s.shape
        dev, nc = s.device, wa.shape[0]
        wh = torch.empty((nc, 32, 1, 1), device=dev, dtype=torch.float32)
        ch = torch.empty(nc, device=dev, dtype=torch.float32)
        lib.head_compose(wt, bt, wa, ba, nc, wh, ch)
        lg = torch.empty((N_, H, W_, nc), device=dev, dtype=torch.float32)
        lib.pw_fwd(s, wh, ch, lg, N_ * H * W_, 32, nc, 0, dtype_code(torch.float32))
        ctx.save_for_backward(s, wh)
        ctx.params = (wt, bt, wa, ba)
        return lg

    @staticmethod
    def backward(ctx, dl):
        s, wh = ctx.saved_tensors
        wt, bt, wa, ba = ctx.params
        N_, H, W_, _ = s.shape
        nc, dev = wa.shape[0], s.device
        dl = _as(dl, torch.float32)
        F32_, BF_ = dtype_code(torch.float32), dtype_code(s.dtype)
        ds = torch.empty_like(s)
        lib.conv2d_dgrad(dl, wh, ds, N_, H, W_, 32, nc, 1, 1, 0, 0, F32_, BF_)
        with _wgrad_stream(_slot_written(*ctx.params), s, dl, wh):
            dwh, dch = ZERO.get((nc, 32), torch.float32, dev), ZERO.get((nc,), torch.float32, dev)
            lib.pw_wgrad_smalln(s, dl, dwh, dch, N_ * H * W_, 32, nc, BF_, F32_)
            outs = [_grad_out(p, tuple(p.shape)) for p in ctx.params]
            lib.head_compose_bwd(wt, bt, wa, nc, dwh, dch, *outs)
        return (ds,) + tuple(_ret(o, p) for o, p in zip(outs, ctx.params))


HEAD_COMPOSE = True       # False: t32x and aux_i stay two convolutions at levels 1-3 (A/B timing)


def head_through_t32_ok(s, wt, bt, wa, ba):
    return (HEAD_COMPOSE and s.dtype == torch.bfloat16 and s.dim() == 4 and s.shape[-1] == 32 and s.is_contiguous() and tuple(wt.shape) == (32, 32, 1, 1)
            and bt is not None and wa.dim() == 4 and tuple(wa.shape[1:]) == (32, 1, 1) and 2 <= wa.shape[0] <= 8 and ba is not None
            and wt.is_contiguous() and wa.is_contiguous() and s.numel() // 32 < 2 ** 31)


def head_through_t32(s, wt, bt, wa, ba):
    """aux(t32(s)) as fp32 NHWC logits from one GEMM with the composed weight; check head_through_t32_ok first"""
    return _HeadThroughT32.apply(s, wt, bt, wa, ba)


def conv1x1_and_sum(x, w, bias, res):
    """(conv1x1(x), conv1x1(x) + res); bf16 NHWC with channel counts multiples of 32 takes the double-store epilogue"""
    ok = (x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[-1] % 32 == 0 and w.shape[0] % 32 == 0 and w.shape[0] <= 160
          and tuple(w.shape[2:]) == (1, 1) and w.shape[1] == x.shape[-1] and bias is not None and res.shape == x.shape[:-1] + (w.shape[0],)
          and res.dtype == x.dtype)
    if not ok:
        d = conv2d(x, w, bias)
        return d, add(res, d)
    return _Conv1x1AndSum.apply(x, w, bias, res)


def linear_residual(x, w, bias, res, scale=None):
    """res + scale[b] * Linear(x) for token tensors [B,N,C]; bf16 with C multiples of 32 takes the fused epilogue"""
    ok = (x.dtype == torch.bfloat16 and x.dim() == 3 and w.dim() == 2 and x.shape[-1] % 32 == 0 and w.shape[0] % 32 == 0
          and w.shape[0] <= 160 and bias is not None and res.shape == x.shape[:-1] + (w.shape[0],))
    if not ok:
        return residual(res, conv2d(x, w, bias), scale)
    return _LinearResidual.apply(x, w, bias, res, scale)


class _PwCat2(_FastFunction):
    """1x1 convolution over the channel concatenation [a | b] without materialising it (MHCA_stage.aggregate)"""

    @staticmethod
    def forward(ctx, a, b, w, stats_box):
        _chk(a, b, w)
        N_, H, W_, Ca = a.shape
        Cb = b.shape[-1]
        Cout = w.shape[0]
        M = N_ * H * W_
        y = torch.empty((N_, H, W_, Cout), device=a.device, dtype=a.dtype)
        sums = None
        if stats_box is not None and Cout % 32 == 0 and Cout <= 128:
            sums = ZERO.get((2 * Cout,), torch.float64, a.device) if ZERO.active else torch.zeros(2 * Cout, device=a.device, dtype=torch.float64)
            stats_box[1] = sums
        lib.pw_fwd_cat2(a, b, Ca, w, None, y, M, Ca + Cb, Cout, sums, stats_box[0] if sums is not None else 0)
        ctx.save_for_backward(a, b, w)
        wsrc = w if hasattr(w, '_grad_slot') or w._base is None else w._base
        ctx.wsrc = wsrc
        return y

    @staticmethod
    def backward(ctx, dy):
        a, b, w = ctx.saved_tensors
        dy = _c(dy)
        N_, H, W_, Ca = a.shape
        Cb, Cout, M = b.shape[-1], w.shape[0], N_ * H * W_
        da, db_ = torch.empty_like(a), torch.empty_like(b)
        if (FUSED_PW_BWD and Ca == 64 and Cb == 64 and Cout % 32 == 0 and Cout <= 128 and dy.dtype == torch.bfloat16
                and a.numel() * 2 < 2 ** 31 and dy.numel() * 2 < 2 ** 31):
            dw = _grad_out(ctx.wsrc, tuple(w.shape))
            lib.pw_bwd_cat2(a, b, dy, w, da, db_, dw, M, Ca + Cb, Cout)
            return da, db_, _ret(dw, ctx.wsrc), None
        lib.pw_dgrad_split2(dy, w, da, db_, Ca, M, Cout, Ca + Cb)
        with _wgrad_stream(_slot_written(ctx.wsrc), a, b, dy):
            dw = _grad_out(ctx.wsrc, tuple(w.shape))
            lib.pw_wgrad_cat2(a, b, Ca, dy, dw, None, M, Ca + Cb, Cout)
        return da, db_, _ret(dw, ctx.wsrc), None


def conv1x1_cat2(a, b, w, stats_pre=None):
    """conv1x1(cat([a, b], channel)) with weight w [Cout, Ca+Cb, 1, 1] (no bias): forward, both input gradients and the weight gradient
    read / write the two halves in place -- no concat / split passes.  bf16 with Ca, Cb multiples of 32 and Cout <= 160; any other
    case takes concat2 + conv2d."""
    Ca, Cb, Cout = a.shape[-1], b.shape[-1], w.shape[0]
    ok = (a.dtype == torch.bfloat16 and b.dtype == a.dtype and a.shape[:-1] == b.shape[:-1] and Ca % 32 == 0 and Cb % 32 == 0
          and Cout % 32 == 0 and Cout <= 160 and tuple(w.shape[1:]) == (Ca + Cb, 1, 1) and a.dim() == 4)
    if not ok:
        return conv2d(concat2(a, b), w, None, stats_pre=stats_pre)
    box = [ACT[stats_pre], None] if stats_pre is not None else None
    y = _PwCat2.apply(a, b, w, box)
    if box is not None and box[1] is not None:
        y._bn_sums = (box[1], box[0])
    return y


def im2col3x3_c3(x4, stride=1):
    """x4 NHWC [N,H,W,4] (3 image channels + zero pad) -> 3x3 patch pixels [N,Ho,Wo,32]; no gradient (the image needs none)"""
    _chk(x4)
    N, H, W, C = x4.shape
    if C != 4:
        raise TcctError('im2col3x3_c3 expects the 4-channel NHWC image')
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty((N, Ho, Wo, 32), device=x4.device, dtype=x4.dtype)
    lib.im2col3x3_c3(x4.detach(), out, N, H, W, stride, dtype_code(x4.dtype))
    return out


C3_DIRECT = True       # False: im2col + pointwise GEMM for the 3-channel first layers (A/B timing)


class _ConvC3(_FastFunction):
    """cnn.0 / stem.0 (reference nets/tcct.py:873, :674-681): 3 -> 32 channels, 3x3, pad 1, stride 1 / 2, straight from the 4-channel image
    (tcct_c3_fwd / tcct_c3_wgrad): the patch rows are gathered inside the MFMA kernels, the 32-channel im2col tensor (452 MB at the bench
    shape, written once and read twice) does not exist.  The image gets no gradient."""

    @staticmethod
    def forward(ctx, x4, w, bias, stride, stats_box):
        _chk(x4, w, bias)
        B, H, W, _ = x4.shape
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = torch.empty((B, Ho, Wo, 32), device=x4.device, dtype=torch.bfloat16)
        sums = None
        if stats_box is not None:
            sums = ZERO.get((64,), torch.float64, x4.device) if ZERO.active else torch.zeros(64, device=x4.device, dtype=torch.float64)
            stats_box[1] = sums
        lib.c3_fwd(x4, w, bias, y, B, H, W, stride, sums, stats_box[0] if stats_box is not None else 0, None, 0, 0)
        ctx.save_for_backward(x4)
        ctx.cfg = (stride, w, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x4, = ctx.saved_tensors
        stride, w, bias = ctx.cfg
        dy = _c(dy)
        B, H, W, _ = x4.shape
        with _wgrad_stream(_slot_written(w, bias), x4, dy):
            dw = _grad_out(w, tuple(w.shape))
            db = _grad_out(bias) if bias is not None else None
            lib.c3_wgrad(x4, dy, dw, db, B, H, W, stride)
        return None, _ret(dw, w), _ret(db, bias), None, None


def _c3_direct_ok(x4, w):
    B, H, W, C = x4.shape
    return (C3_DIRECT and x4.dtype == torch.bfloat16 and C == 4 and tuple(w.shape) == (32, 3, 3, 3) and w.dtype == torch.float32
            and w.is_contiguous() and B * H * W * 64 < 2 ** 31)         # 32-bit byte offsets into the image and into the 32-channel output


def conv3x3_c3(x4, w, bias, stride=1, stats_pre=None, infer_bn=None, post_act=None):
    """3-channel 3x3 conv (pad 1).  bf16, 32 output channels: the direct kernels (_ConvC3).  Otherwise (fp32 parity mode, other widths)
    im2col + 32->32 pointwise GEMM: w [32,3,3,3] is re-laid out to [32, 27->32] by view ops
    (differentiable plumbing on 864 elements), so forward and weight gradient both run on the MFMA pointwise kernels.
    infer_bn (inference only): eval-mode BatchNorm tuple folded, with post_act, into the GEMM epilogue (see conv_bn_act)."""
    if _c3_direct_ok(x4, w):
        _chk(x4)
        if infer_bn is not None:
            if torch.is_grad_enabled() and w.requires_grad:
                raise TcctError('conv3x3_c3(infer_bn=...) is inference-only (call it under torch.no_grad())')
            B, H, W, _ = x4.shape
            ab = torch.empty(64, device=x4.device, dtype=torch.float32)
            lib.bn_eval_ab(32, infer_bn[0], infer_bn[1], float(infer_bn[4]), infer_bn[2], infer_bn[3], torch.empty(64, device=x4.device, dtype=torch.float32), ab)
            y = torch.empty((B, (H - 1) // stride + 1, (W - 1) // stride + 1, 32), device=x4.device, dtype=torch.bfloat16)
            if C3_BN_FUSE and ACT[post_act] in (0, ACT['hswish']):      # round 6: the recompute kernel's normalising pass with the eval coefficients (0.10 vs 0.28 ms at level 0)
                lib.c3_bn_fwd_eval(x4, w, bias, y, B, H, W, stride, ab, ACT[post_act])
            else:
                lib.c3_fwd(x4, w, bias, y, B, H, W, stride, None, 0, ab, 0, ACT[post_act])
            return y
        box = [ACT[stats_pre], None] if stats_pre is not None else None
        y = _ConvC3.apply(x4, w, bias, stride, box)
        if box is not None and box[1] is not None:
            y._bn_sums = (box[1], box[0])
        return y
    w2 = torch.nn.functional.pad(w.permute(0, 2, 3, 1).reshape(w.shape[0], 27), (0, 5)).contiguous()
    if infer_bn is not None:
        return conv_bn_act(im2col3x3_c3(x4, stride), w2.view(w.shape[0], 32, 1, 1), bias, bn=infer_bn, post_act=post_act)
    return conv2d(im2col3x3_c3(x4, stride), w2.view(w.shape[0], 32, 1, 1), bias, stats_pre=stats_pre)


C3_BN_FUSE = True       # False: convolution (+ statistics) and BatchNorm as separate nodes, the round-3 form (A/B timing)


class _ConvC3BN(_FastFunction):
    """z = post(BN_train(conv3x3(x4) + bias)) for the two 3-channel first layers as ONE node whose convolution output is never stored
    (csrc/c3_bn.hip; reference nets/tcct.py:873 `cnn.0 -> cnn.1`, :55-97 + :674-681 `stem[0]`): the input is the 4-channel image, 1/8 of the
    bytes of the 32-channel output, so forward (statistics pass, normalising pass) and backward (reduction, weight gradient) recompute y on the
    matrix pipes instead of writing it once and reading it three times."""

    @staticmethod
    def forward(ctx, x4, w, bias, gamma, beta, rm, rv, nbt, eps, momentum, stride, post):
        _chk(x4, w, bias, gamma, beta)
        B, H, W, _ = x4.shape
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        z = torch.empty((B, Ho, Wo, 32), device=x4.device, dtype=torch.bfloat16)
        sums = ZERO.get((64,), torch.float64, x4.device) if ZERO.active else torch.zeros(64, device=x4.device, dtype=torch.float64)
        mean_rstd = torch.empty(64, device=x4.device, dtype=torch.float32)
        ab = torch.empty(64, device=x4.device, dtype=torch.float32)
        lib.c3_bn_fwd_train(x4, w, bias, z, B, H, W, stride, sums, gamma, beta, eps, momentum, rm, rv, nbt, mean_rstd, ab, post)
        ctx.save_for_backward(x4, mean_rstd, ab)
        ctx.cfg = (stride, post, w, bias, gamma, beta)
        return z

    @staticmethod
    def backward(ctx, dz):
        x4, mean_rstd, ab = ctx.saved_tensors
        stride, post, w, bias, gamma, beta = ctx.cfg
        dz = _as(dz, torch.bfloat16)
        B, H, W, _ = x4.shape
        M = dz.numel() // 32
        dev = x4.device
        raw = ZERO.get((64,), torch.float64, dev) if ZERO.active else torch.zeros(64, device=dev, dtype=torch.float64)
        coef = torch.empty(160, device=dev, dtype=torch.float32)
        # nothing downstream waits for this node (the image has no gradient): all of it runs beside the other encoder's tail on the weight-gradient stream
        with _wgrad_stream(_slot_written(w, bias, gamma, beta), x4, dz, coef, mean_rstd, ab):
            dg = _grad_out(gamma) if ZERO.active and getattr(gamma, '_grad_slot', None) is not None else torch.empty(32, device=dev, dtype=torch.float32)
            db_ = _grad_out(beta) if ZERO.active and getattr(beta, '_grad_slot', None) is not None else torch.empty(32, device=dev, dtype=torch.float32)
            dw = _grad_out(w, tuple(w.shape))
            dbias = _grad_out(bias) if bias is not None else None
            if C3_ONEPASS:      # reduction and weight gradient from ONE pass over dz (the BatchNorm backward is linear in per-pixel quantities)
                if C3_BWD_FORM != _C3_FORM_SET[0]:
                    lib.c3_bn_bwd_prefetch(C3_BWD_FORM)
                    _C3_FORM_SET[0] = C3_BWD_FORM
                work = ZERO.get((4160,), torch.float32, dev) if ZERO.active else torch.zeros(4160, device=dev, dtype=torch.float32)
                s96 = ZERO.get((96,), torch.float64, dev) if ZERO.active else torch.zeros(96, device=dev, dtype=torch.float64)
                lib.c3_bn_bwd_onepass(x4, w, bias, dz, B, H, W, stride, mean_rstd, ab, work, s96, dw, dbias, dg, db_, post)
            else:
                lib.c3_bn_bwd_reduce(x4, w, bias, dz, B, H, W, stride, ab, raw, post)
                lib.bn_bwd_coef(raw, 1, M, 32, mean_rstd, ab, coef, dg, db_)
                lib.c3_bn_bwd_wgrad(x4, w, bias, dz, B, H, W, stride, coef, dw, dbias, post)
        return None, _ret(dw, w), _ret(dbias, bias), _ret(dg, gamma), _ret(db_, beta), None, None, None, None, None, None, None


C3_BWD_FORM = 4        # kernel form of the one-pass backward (tcct_c3_bn_bwd_prefetch): 4 = wave-private 32-pixel tiles (round 6), 1 = block tiles of 128 pixels (A/B timing)
_C3_FORM_SET = [4]
C3_ONEPASS = True      # False: BatchNorm reduction and weight gradient of the first layers as two passes over dz (A/B timing)


def conv3x3_c3_bn_ok(x4, w, bn_training, post_act):
    # (train-mode BatchNorm, with or without autograd: a no_grad train-mode forward must round where the training step does)
    return (C3_BN_FUSE and bn_training and _c3_direct_ok(x4, w) and ACT[post_act] in (0, ACT['hswish']))


def conv3x3_c3_bn(x4, w, bias, bn, stride=1, post_act=None):
    """post_act(BN_train(conv3x3_c3(x4))); bn = (gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum); check
    conv3x3_c3_bn_ok first"""
    gamma, beta, rm, rv, nbt, eps, mom = bn
    return _ConvC3BN.apply(x4, w, bias, gamma, beta, rm, rv, nbt, float(eps), float(mom), int(stride), ACT[post_act])


class _DwConv(_FastFunction):
    @staticmethod
    def forward(ctx, x, w, bias, stride, add_input, fork=False, stats_box=None, xab=None, xlink=None):
        """fork: also return an alias of x for its other consumers; their gradient is added inside the input-gradient kernel.
        stats_box ([None]): the kernel also accumulates the statistics of the train-mode BatchNorm that consumes the output.
        xab (round 4): x is y_prev, the input of a train-mode BatchNorm + Hardswish in front whose normalisation pass was not run (xab = its {a[C], b[C]});
        the kernels apply z = hswish(a y_prev + b) as rows enter their register window.  The gradient returned for x is the gradient of that z."""
        ctx.set_materialize_grads(False)      # an unused output must arrive as None in backward(), not as a zero-filled tensor
        _chk(x, w, bias)
        N, H, W, C = x.shape
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = torch.empty((N, Ho, Wo, C), device=x.device, dtype=x.dtype)
        ctx.xab = xab
        ctx.xlink = xlink       # the BnLink of the BatchNorm pending on x: the input-gradient kernel delivers its backward sums through it
        if xab is not None:
            if add_input or fork or C % 4 or C > 256:
                raise TcctError('dwconv3x3(deferred=...): plain convolution with C % 4 == 0, C <= 256 only')
            sums = None
            if stats_box is not None:
                sums = ZERO.get((2 * C,), torch.float64, x.device) if ZERO.active else torch.zeros(2 * C, device=x.device, dtype=torch.float64)
                stats_box[0] = sums
            lib.dwconv3x3_fwd_xaff(x, xab, w, bias, y, N, H, W, C, stride, sums, dtype_code(x.dtype))
        elif stats_box is not None and C % 4 == 0 and C <= 256:
            sums = ZERO.get((2 * C,), torch.float64, x.device) if ZERO.active else torch.zeros(2 * C, device=x.device, dtype=torch.float64)
            lib.dwconv3x3_fwd_bnstats(x, w, bias, y, N, H, W, C, stride, int(add_input), sums, dtype_code(x.dtype))
            stats_box[0] = sums
        else:
            lib.dwconv3x3_fwd(x, w, bias, y, N, H, W, C, stride, int(add_input), dtype_code(x.dtype))
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, add_input, bias is not None)
        ctx.bias_param = bias
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, w = ctx.saved_tensors
        stride, add_input, has_bias = ctx.cfg
        if dy is None:
            return dskip, None, None, None, None, None, None, None, None
        dy = _as(dy, x.dtype)
        N, H, W, C = x.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if dskip is not None:
                lib.dwconv3x3_dgrad_add(dy, w, _as(dskip, x.dtype), dx, N, H, W, C, stride, int(add_input), dtype_code(x.dtype))
            elif ctx.xab is not None and ctx.xlink is not None and stride == 1 and BN_FUSE_RED and BN_RED_DW:
                # x is y_prev of the pending BatchNorm: its two backward sums come out of this launch (raw form), its own reduction pass is skipped
                raw = ZERO.get((2 * C,), torch.float64, x.device) if ZERO.active else torch.zeros(2 * C, device=x.device, dtype=torch.float64)
                lib.dwconv3x3_dgrad_bnred(dy, w, x, ctx.xab, dx, raw, N, H, W, C, dtype_code(x.dtype))
                ctx.xlink.sums = raw
            else:
                lib.dwconv3x3_dgrad(dy, w, dx, N, H, W, C, stride, int(add_input), dtype_code(x.dtype))
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            # on the weight-gradient side stream like the dense convolutions' (round 4, last day: the ten depthwise weight gradients were 0.85 ms of the
            # ViT branch's input-gradient chain; `DW_WGRAD_SIDE = False` / bench.py --set DW_WGRAD_SIDE=0 keeps them inline)
            keep = (x, dy) if ctx.xab is None else (x, dy, ctx.xab)
            with _wgrad_stream(DW_WGRAD_SIDE and _slot_written(w, ctx.bias_param if has_bias else None), *keep):
                dw = _grad_out(w)
                db = _grad_out(ctx.bias_param) if has_bias else None
                if ctx.xab is not None:
                    lib.dwconv3x3_wgrad_xaff(x, ctx.xab, dy, dw, db, N, H, W, C, stride, dtype_code(x.dtype))
                else:
                    lib.dwconv3x3_wgrad(x, dy, dw, db, N, H, W, C, stride, dtype_code(x.dtype))
        return dx, _ret(dw, w), _ret(db, ctx.bias_param), None, None, None, None, None, None


def dwconv3x3(x, w, bias=None, stride=1, add_input=False, bn_stats=False, deferred=None):
    """bn_stats: a train-mode BatchNorm (no activation in front) consumes the output: its statistics come out of the same launch (`_bn_sums`).
    deferred: the BnLink of a BatchNorm + Hardswish whose normalisation is pending on x (batchnorm_deferred / pw_conv_bn(defer_apply=True))"""
    xab = deferred.ab if deferred is not None else None
    if bn_stats:
        box = [None]
        y = _DwConv.apply(x, w, bias, stride, add_input, False, box, xab, deferred)
        if box[0] is not None:
            y._bn_sums = (box[0], ACT['none'])
        return y
    return _DwConv.apply(x, w, bias, stride, add_input, False, None, xab, deferred)


def dwconv3x3_fork(x, w, bias=None, stride=1, add_input=False):
    """(dwconv3x3(x), x'): x' aliases x for the other consumers of x (see _DwConv.forward `fork`)"""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return dwconv3x3(x, w, bias, stride, add_input), x
    return _DwConv.apply(x, w, bias, stride, add_input, True)


# ------------------------------------------------------------------------------------------------- norms
class _BatchNorm(_FastFunction):
    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, nbt, eps, momentum, pre, post, training, res=None, link=None):
        _chk(x, gamma, beta)
        C = x.shape[-1]
        M = x.numel() // C
        dc = dtype_code(x.dtype)
        mean_rstd = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        ab = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        if training:
            fused = getattr(x, '_bn_sums', None)
            if fused is not None and fused[1] == pre and fused[0].numel() == 2 * C:
                sums = fused[0]             # statistics were accumulated by the producing conv's epilogue
            else:
                sums = ZERO.get((2 * C,), torch.float64, x.device)
                lib.bn_stats(x, M, C, pre, sums, dc)
        else:
            lib.bn_eval_ab(C, gamma, beta, eps, rm, rv, mean_rstd, ab)
        y = torch.empty_like(x)
        if res is not None:
            _chk(res)
            if res.shape != x.shape or res.dtype != x.dtype:
                raise TcctError('batchnorm: residual must have the shape and dtype of the input')
        if training:            # statistics finalisation (+ running stats) folded into the apply launch
            lib.bn_apply_train(x, res, y, M, C, sums, gamma, beta, eps, momentum, rm, rv, nbt, mean_rstd, ab, pre, post, dc)
        elif res is not None:
            lib.bn_apply_add(x, res, y, M, C, ab, pre, post, dc)
        else:
            lib.bn_apply(x, y, M, C, ab, pre, post, dc)
        if training:
            ctx.save_for_backward(x, gamma, mean_rstd, ab)
            ctx.cfg = (pre, post)
            ctx.beta_param = beta
            ctx.has_res = res is not None
            ctx.link = link
            if link is not None:
                link.y, link.ab = x, ab
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean_rstd, ab = ctx.saved_tensors
        pre, post = ctx.cfg
        dy = _as(dy, x.dtype)
        C = x.shape[-1]
        M = x.numel() // C
        dc = dtype_code(x.dtype)
        link = getattr(ctx, 'link', None)
        if link is not None and link.sums is not None:      # the consumer's input-gradient kernel accumulated the two batch sums (raw form)
            sums = torch.empty(2 * C, device=x.device, dtype=torch.float64)
            lib.bn_sums_from_raw(link.sums, mean_rstd, C, sums)
            link.sums = None
        else:
            sums = ZERO.get((2 * C,), torch.float64, x.device)
            lib.bn_bwd_reduce(x, dy, M, C, mean_rstd, ab, pre, post, sums, dc)
        dx = torch.empty_like(x)
        dg = _grad_out(gamma) if ZERO.active and getattr(gamma, '_grad_slot', None) is not None else torch.empty(C, device=x.device, dtype=torch.float32)
        db = _grad_out(ctx.beta_param) if ZERO.active and getattr(ctx.beta_param, '_grad_slot', None) is not None else torch.empty(C, device=x.device, dtype=torch.float32)
        lib.bn_bwd_apply(x, dy, dx, M, C, mean_rstd, ab, gamma, sums, pre, post, dg, db, dc)
        return dx, _ret(dg, gamma), _ret(db, ctx.beta_param), None, None, None, None, None, None, None, None, (dy if ctx.has_res else None), None


# ---- BatchNorm backward folded into the neighbouring pointwise-convolution kernels (round 3) ------------------------------------------
# A train-mode BatchNorm costs 2 (apply) + 2 (backward reduce) + 3 (backward apply) tensor passes.  Where a 1x1 convolution PRODUCES the
# BatchNorm's input (Conv2d_BN, DWConv2d_BN.pwconv, FTC.tran_*), the fused pointwise backward kernel rebuilds the convolution's output
# gradient from (dz, y) while staging its tiles (tcct_pw_bwd_bn): the backward apply pass disappears (3 passes -> 1 extra read).  Where a
# 1x1 convolution CONSUMES a BatchNorm's output as the last contributor to its gradient, the same kernel accumulates that BatchNorm's two
# backward sums in its dx epilogue (`BnLink`): the backward reduce pass disappears too (2 passes -> 1 extra read).  `BN_FUSE = False` (bench.py --set BN_FUSE=0) restores
# the separate kernels (A/B timing, bisecting).
BN_FUSE = True
BN_FUSE_RED = True       # False: keep the separate reduction kernels (A/B timing of the epilogue form)


class BnLink:
    """What a consumer needs to run the backward REDUCTION of the BatchNorm that produced its input: y (the BatchNorm input), ab = {a[C], b[C]},
    the post-activation kind; `sums` is filled (raw form {sum dz', sum dz' y}, fp64 [2C]) by the consumer's backward, which autograd runs
    before the BatchNorm's own backward node -- that node then skips its reduction kernel."""
    __slots__ = ('y', 'ab', 'post', 'sums')

    def __init__(self, y, ab, post):
        self.y, self.ab, self.post, self.sums = y, ab, post, None


def _bn_link_of(x, final):
    """the BnLink of tensor x when x is the unmodified output of a BatchNorm node and the caller vouches (`final`) that its convolution's
    input gradient (+ the forked alias' gradient) is the COMPLETE gradient of x"""
    if not (final and BN_FUSE and BN_FUSE_RED):
        return None
    return getattr(x, '_bn_link', None)


class _PwConvBN(_FastFunction):
    """z = post(BN_train(conv1x1(x))) [+ res] as ONE autograd node (Conv2d_BN / DWConv2d_BN.pwconv / FTC.tran_*, reference nets/tcct.py:55-97,
    124-126,966-974): forward = the GEMM with the statistics in its epilogue + the normalisation pass; backward = [reduction, unless a consumer
    delivered the sums through `link`] + ONE kernel for BatchNorm backward apply, dx, dW, dbias (tcct_pw_bwd_bn), optionally with the reduction
    of the BatchNorm in FRONT of the convolution in its epilogue (`prev`).  x2: second half of a concatenated input (MHCA_stage.aggregate)."""

    @staticmethod
    def forward(ctx, x, x2, w, bias, gamma, beta, rm, rv, nbt, eps, momentum, post, res, fork, link, prev, xdef=None, defer_apply=False):
        """xdef (round 4): x is NOT the convolution's input but y_prev, the input of a train-mode BatchNorm + Hardswish in front whose normalisation
        was deferred (batchnorm_deferred): xdef = its {a[K], b[K]}; the kernels apply hswish(a y_prev + b) while they stage their tiles"""
        ctx.set_materialize_grads(False)
        _chk(x, x2, w, bias, gamma, beta, res)
        K1 = x.shape[-1]
        K = K1 + (x2.shape[-1] if x2 is not None else 0)
        N = w.shape[0]
        M = x.numel() // K1
        y = torch.empty(x.shape[:-1] + (N,), device=x.device, dtype=x.dtype)
        sums = ZERO.get((2 * N,), torch.float64, x.device) if ZERO.active else torch.zeros(2 * N, device=x.device, dtype=torch.float64)
        if xdef is not None:
            lib.pw_fwd_bnstats_xaff(x, xdef, w, bias, y, M, K, N, sums)
        elif x2 is not None:
            lib.pw_fwd_cat2(x, x2, K1, w, bias, y, M, K, N, sums, 0)
        else:
            lib.pw_fwd_bnstats(x, w, bias, y, M, K, N, sums, 0)
        mean_rstd = torch.empty(2 * N, device=x.device, dtype=torch.float32)
        ab = torch.empty(2 * N, device=x.device, dtype=torch.float32)
        if defer_apply:     # round 4: the normalisation pass is left to the one consumer (a depthwise convolution built with `deferred=link`): z is never written
            lib.bn_finalize(sums, M, N, gamma, beta, eps, momentum, rm, rv, nbt, mean_rstd, ab)
            z = y.view_as(y)
        else:
            z = torch.empty_like(y)
            lib.bn_apply_train(y, res, z, M, N, sums, gamma, beta, eps, momentum, rm, rv, nbt, mean_rstd, ab, 0, post, dtype_code(y.dtype))
        ctx.save_for_backward(x, x2, w, y, mean_rstd, ab)
        wsrc = w if hasattr(w, '_grad_slot') or w._base is None else w._base
        ctx.cfg = (post, res is not None, wsrc, bias, gamma, beta, link, prev, M, K, N)
        ctx.xdef = xdef
        link.y, link.ab = y, ab
        return (z, x.view_as(x)) if fork else z

    @staticmethod
    def backward(ctx, dz, dalias=None):
        x, x2, w, y, mean_rstd, ab = ctx.saved_tensors
        post, has_res, wsrc, bsrc, gamma, beta, link, prev, M, K, N = ctx.cfg
        if dz is None:
            return (dalias,) + (None,) * 17
        dz = _as(dz, y.dtype)
        if link.sums is not None:               # a consumer's dx epilogue already holds the two batch sums (raw form)
            sums, raw = link.sums, 1
            link.sums = None
        else:
            sums, raw = ZERO.get((2 * N,), torch.float64, y.device), 0
            lib.bn_bwd_reduce(y, dz, M, N, mean_rstd, ab, 0, post, sums, dtype_code(y.dtype))
        dg = _grad_out(gamma) if ZERO.active and getattr(gamma, '_grad_slot', None) is not None else torch.empty(N, device=y.device, dtype=torch.float32)
        db_ = _grad_out(beta) if ZERO.active and getattr(beta, '_grad_slot', None) is not None else torch.empty(N, device=y.device, dtype=torch.float32)
        dx = torch.empty_like(x)
        dx2 = torch.empty_like(x2) if x2 is not None else None
        dw = _grad_out(wsrc, tuple(w.shape))
        dbias = _grad_out(bsrc) if bsrc is not None else None
        ypv = abp = sp = None
        redp = -1
        if prev is not None:
            redp = prev.post
            ypv, abp = prev.y, prev.ab
            sp = prev.sums = ZERO.get((2 * x.shape[-1],), torch.float64, y.device)
        dskip = _as(dalias, x.dtype) if dalias is not None else None
        if ctx.xdef is not None:        # x is y_prev: the operand is rebuilt on load (dx is the gradient of the DEFERRED BatchNorm's output)
            lib.pw_bwd_bn_sums_xaff(x, ctx.xdef, dz, y, sums, raw, mean_rstd, ab, dg, db_, w, dskip, dx, dw, dbias, M, K, N, redp, sp)
        else:
            lib.pw_bwd_bn_sums(x, x2, dz, y, sums, raw, mean_rstd, ab, dg, db_, post, w, dskip, dx, dx2, dw, dbias, M, K, N, ypv, abp, redp, sp)
        return (dx, dx2, _ret(dw, wsrc), _ret(dbias, bsrc), _ret(dg, gamma), _ret(db_, beta), None, None, None, None, None, None,
                (dz if has_res else None), None, None, None, None, None)


class _Affine2Add(_FastFunction):
    """z = BN1(y1) + BN2(y2) with both normalisations PENDING on their inputs (pw_conv_bn(defer_apply=True) with no activation): one pass; the gradient of z is
    the gradient of both BatchNorm outputs"""

    @staticmethod
    def forward(ctx, y1, ab1, y2, ab2, link1=None, link2=None):
        _chk(y1, ab1, y2, ab2)
        z = torch.empty_like(y1)
        C = y1.shape[-1]
        lib.affine2_add(y1, ab1, y2, ab2, z, y1.numel() // C, C, dtype_code(y1.dtype))
        ctx.links = (link1, link2)
        if link1 is not None and link2 is not None:
            ctx.save_for_backward(y1, y2)
        return z

    @staticmethod
    def backward(ctx, dz):
        l1, l2 = ctx.links
        if l1 is not None and l2 is not None and BN_FUSE_RED and TRAN_RED2:
            # both BatchNorms receive dz: their backward sums (raw form, through the links) from ONE pass over dz, y1, y2
            y1, y2 = ctx.saved_tensors
            dz = _as(dz, y1.dtype)
            C = y1.shape[-1]
            dev = y1.device
            r1 = ZERO.get((2 * C,), torch.float64, dev) if ZERO.active else torch.zeros(2 * C, device=dev, dtype=torch.float64)
            r2 = ZERO.get((2 * C,), torch.float64, dev) if ZERO.active else torch.zeros(2 * C, device=dev, dtype=torch.float64)
            lib.bn_bwd_reduce2_raw(y1, y2, dz, y1.numel() // C, C, r1, r2, dtype_code(y1.dtype))
            l1.sums, l2.sums = r1, r2
        return dz, None, dz, None, None, None


TRAN_RED2 = True      # False: the two fused BatchNorms keep separate backward reduction passes (A/B timing)
TRAN_FUSE = True      # False: the two BatchNorms of the encoder fusion keep their own normalisation passes (A/B timing)


def affine2_add(y1, link1, y2, link2):
    """BN1(y1) + BN2(y2) for two tensors that carry a pending train-mode BatchNorm (links from pw_conv_bn(..., defer_apply=True))"""
    if y1.shape != y2.shape or y1.dtype != y2.dtype or y1.shape[-1] % 8:
        raise TcctError('affine2_add: two NHWC tensors of one shape and dtype, channels a multiple of 8')
    return _Affine2Add.apply(y1, link1.ab, y2, link2.ab, link1, link2)


BN_RED_DW = True      # False: the BatchNorms in front of the depthwise convolutions keep their backward reduction pass (A/B)
BN_DEFER_DW = True  # False: the BatchNorms in front of the depthwise convolutions keep their normalisation pass (A/B timing)
BN_DEFER = True        # False: InvRes.norm keeps its own normalisation pass (round-3 form; A/B timing)


class _BatchNormDeferred(_FastFunction):
    """A train-mode BatchNorm (+ Hardswish) whose normalisation pass is NOT run (round 4): the node finalises the batch statistics (mean / rstd / a / b,
    running statistics) and returns an ALIAS OF ITS INPUT; the one consumer -- a 1x1 convolution built with `xdef` (pw_conv_bn(..., deferred=link)) --
    applies z = hswish(a y + b) while it stages its tiles.  The gradient that arrives here is the gradient of that virtual z, so the backward is the
    ordinary BatchNorm backward (reduction, unless the consumer's epilogue delivered the sums through `link`, then the apply pass).
    The alias must go to that consumer and nowhere else."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, nbt, eps, momentum, post, link):
        _chk(x, gamma, beta)
        C = x.shape[-1]
        M = x.numel() // C
        fused = getattr(x, '_bn_sums', None)
        if fused is not None and fused[1] == 0 and fused[0].numel() == 2 * C:
            sums = fused[0]
        else:
            sums = ZERO.get((2 * C,), torch.float64, x.device) if ZERO.active else torch.zeros(2 * C, device=x.device, dtype=torch.float64)
            lib.bn_stats(x, M, C, 0, sums, dtype_code(x.dtype))
        mean_rstd = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        ab = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        lib.bn_finalize(sums, M, C, gamma, beta, eps, momentum, rm, rv, nbt, mean_rstd, ab)
        ctx.save_for_backward(x, gamma, mean_rstd, ab)
        ctx.cfg = (0, post)
        ctx.beta_param = beta
        ctx.has_res = False
        ctx.link = link
        link.y, link.ab = x, ab
        return x.view_as(x)

    backward = None     # assigned below: _BatchNorm.backward with its return tuple cut to this node's inputs


def _bn_deferred_backward(ctx, dy):
    out = _BatchNorm.backward(ctx, dy)
    return out[0], out[1], out[2], None, None, None, None, None, None, None


_BatchNormDeferred.backward = staticmethod(_bn_deferred_backward)


def batchnorm_deferred_ok(x, bn_training, post_act):
    return (BN_DEFER and BN_FUSE and bn_training and torch.is_grad_enabled() and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[-1] in (64, 96)
            and ACT[post_act] == ACT['hswish'] and x.numel() * 2 < 2 ** 31)


def batchnorm_deferred(x, gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum, post_act):
    """-> (alias of x carrying the pending normalisation, link): hand both to pw_conv_bn(alias, ..., deferred=link) and to nothing else"""
    link = BnLink(None, None, ACT[post_act])
    y = _BatchNormDeferred.apply(x, gamma, beta, running_mean, running_var, num_batches_tracked, float(eps), float(momentum), ACT[post_act], link)
    return y, link


def pw_conv_bn_ok(x, w, bias, bn_training, pre_act, post_act, x2=None, prev=None):
    """shapes / modes the fused node takes: train mode with gradients, bf16 rows, 1x1 weights, K and N in the kernel's table"""
    if not (BN_FUSE and FUSED_PW_BWD and bn_training and torch.is_grad_enabled() and ACT[pre_act] == 0):
        return False
    if x.dtype != torch.bfloat16 or x.dim() != 4 or not x.is_cuda or (w.dim() == 4 and tuple(w.shape[2:]) != (1, 1)):
        return False
    K = x.shape[-1] + (x2.shape[-1] if x2 is not None else 0)
    N = w.shape[0]
    if x2 is not None and not (x.shape[-1] == 64 and x2.shape == x.shape and x2.dtype == x.dtype):
        return False
    M = x.numel() // x.shape[-1]
    if w.shape[1] != K or M * max(K, N) * 2 >= 2 ** 31:
        return False
    return bool(lib.pw_bwd_bn_supported(K, N, ACT[post_act], -1 if prev is None else prev.post, 1 if x2 is not None else 0))


def pw_conv_bn(x, w, bias, bn, post_act=None, residual=None, fork=False, x2=None, x_final=False, deferred=None, defer_apply=False):
    """post_act(BN_train(conv1x1(x [| x2]))) [+ residual]; bn = (gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum).
    fork: also return an alias of x for its other consumers (their gradient is added in this node's dx epilogue).  x_final: this
    convolution's input gradient (+ the alias') is the complete gradient of x -- when x came out of a BatchNorm node, that BatchNorm's
    backward reduction rides on this node's kernel.  Check pw_conv_bn_ok first."""
    gamma, beta, rm, rv, nbt, eps, mom = bn
    link = BnLink(None, None, ACT[post_act])
    w4 = w.view(w.shape[0], w.shape[1], 1, 1) if w.dim() == 2 else w
    if deferred is not None:
        # x is the alias batchnorm_deferred returned: y_prev with z = hswish(a y_prev + b) pending; K = N = 64 also carries that BatchNorm's
        # backward reduction in its dx epilogue (the kernel table of tcct_pw_bwd_bn_sums_xaff), 96 leaves it to the BatchNorm's own node
        if post_act is not None or fork or x2 is not None or x.shape[-1] != w.shape[0] or x.shape[-1] not in (64, 96):
            raise TcctError('pw_conv_bn(deferred=...): square 64 / 96 convolution without activation, fork or concatenation only')
        prev = deferred if (x.shape[-1] == 64 and BN_FUSE_RED) else None
        out = _PwConvBN.apply(x, None, w4, bias, gamma, beta, rm, rv, nbt, float(eps), float(mom), 0, residual, False, link, prev, deferred.ab)
        out._bn_link = link
        return out
    prev = _bn_link_of(x, x_final)
    if prev is not None and not pw_conv_bn_ok(x, w, bias, True, None, post_act, x2, prev):
        prev = None
    if defer_apply:
        # the caller hands the result (y with the normalisation + Hardswish PENDING) and `link` to dwconv3x3(..., deferred=link) and to nothing else
        if residual is not None or x2 is not None or ACT[post_act] not in (ACT['hswish'], ACT['none']):
            raise TcctError('pw_conv_bn(defer_apply=True): BatchNorm (+ Hardswish) without residual / concatenation only')
        out = _PwConvBN.apply(x, None, w4, bias, gamma, beta, rm, rv, nbt, float(eps), float(mom), ACT[post_act], None, fork, link, prev, None, True)
        return out, link
    out = _PwConvBN.apply(x, x2, w4, bias, gamma, beta, rm, rv, nbt, float(eps), float(mom), ACT[post_act], residual, fork, link, prev)
    z = out[0] if fork else out
    z._bn_link = link               # (with a residual folded in z is BN output + res, but the gradient of the BatchNorm output still is dz)
    return out


BN_POOL_FUSE = True      # False: BatchNorm pass, then the pooling pass (A/B timing)


class _BnPoolFork(_FastFunction):
    """(maxpool2(z), z) with z = post(BN_train(pre(x))) from ONE pass over x (tcct_bn_pool_fwd_train); in the backward pass the gradient of
    z -- the skip consumers' gradient plus the pooling scatter -- is never written: both BatchNorm backward kernels rebuild it from its
    two sources (tcct_bn_pool_bwd).  Encoder levels 0-3 (reference nets/tcct.py:820-823, :876-884)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, nbt, eps, momentum, pre, post):
        ctx.set_materialize_grads(False)
        _chk(x, gamma, beta)
        N, H, W, C = x.shape
        dc = dtype_code(x.dtype)
        fused = getattr(x, '_bn_sums', None)
        if fused is not None and fused[1] == pre and fused[0].numel() == 2 * C:
            sums = fused[0]
        else:
            sums = ZERO.get((2 * C,), torch.float64, x.device)
            lib.bn_stats(x, N * H * W, C, pre, sums, dc)
        mean_rstd = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        ab = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        z = torch.empty_like(x)
        pooled = torch.empty((N, H // 2, W // 2, C), device=x.device, dtype=x.dtype)
        amax = torch.empty((N, H // 2, W // 2, C // 4), device=x.device, dtype=torch.uint8)        # 2-bit window positions of the maxima
        lib.bn_pool_fwd_train(x, z, pooled, amax, N, H, W, C, sums, gamma, beta, eps, momentum, rm, rv, nbt, mean_rstd, ab, pre, post, dc)
        ctx.save_for_backward(x, mean_rstd, ab, amax)
        ctx.cfg = (pre, post, gamma, beta)
        return pooled, z

    @staticmethod
    def backward(ctx, dpool, dskip):
        x, mean_rstd, ab, amax = ctx.saved_tensors
        pre, post, gamma, beta = ctx.cfg
        N, H, W, C = x.shape
        dc = dtype_code(x.dtype)
        if dpool is None and dskip is None:
            return (None,) * 10
        if dpool is None:           # only the full-size output was used: a plain BatchNorm backward
            dpool = torch.zeros((N, H // 2, W // 2, C), device=x.device, dtype=x.dtype)      # an INPUT of the kernel: must really be zero
        dpool = _as(dpool, x.dtype)
        if dskip is not None:
            dskip = _as(dskip, x.dtype)
        sums = ZERO.get((2 * C,), torch.float64, x.device)
        dx = torch.empty_like(x)
        dg = _grad_out(gamma) if ZERO.active and getattr(gamma, '_grad_slot', None) is not None else torch.empty(C, device=x.device, dtype=torch.float32)
        db = _grad_out(beta) if ZERO.active and getattr(beta, '_grad_slot', None) is not None else torch.empty(C, device=x.device, dtype=torch.float32)
        lib.bn_pool_bwd(x, dpool, dskip, amax, dx, N, H, W, C, mean_rstd, ab, pre, post, sums, dg, db, dc)
        return dx, _ret(dg, gamma), _ret(db, beta), None, None, None, None, None, None, None


def bn_pool_ok(x, training):
    """shapes the fused BatchNorm + MaxPool2d(2) kernels take (train mode, gradients wanted): even extents, C/4 dividing 256"""
    N, H, W, C = x.shape
    return (BN_POOL_FUSE and training and torch.is_grad_enabled() and x.requires_grad and C % 4 == 0 and C <= 256 and 256 % (C // 4) == 0
            and H % 2 == 0 and W % 2 == 0 and H >= 2 and W >= 2)


def batchnorm_maxpool2_fork(x, gamma, beta, running_mean, running_var, num_batches_tracked=None, eps=1e-5, momentum=0.1, pre_act=None,
                            post_act=None):
    """(maxpool2(z), z), z = post_act(BN_train(pre_act(x))): see _BnPoolFork; check bn_pool_ok(x, training) first"""
    return _BnPoolFork.apply(x, gamma, beta, running_mean, running_var, num_batches_tracked, float(eps), float(momentum), ACT[pre_act], ACT[post_act])


def batchnorm(x, gamma, beta, running_mean, running_var, num_batches_tracked=None, eps=1e-5, momentum=0.1,
              pre_act=None, post_act=None, training=True, residual=None):
    """y = post_act(BN(pre_act(x))) [+ residual] over the last (channel) dim, torch train-mode semantics incl. running stats."""
    if not training and torch.is_grad_enabled() and x.requires_grad:
        raise TcctError('eval-mode batchnorm is inference-only here')
    link = None
    if (BN_FUSE and training and torch.is_grad_enabled() and x.dtype == torch.bfloat16 and ACT[pre_act] == 0
            and ACT[post_act] in (0, 2) and x.shape[-1] in (64, 96, 128)):
        link = BnLink(None, None, ACT[post_act])       # a pointwise convolution that consumes the output may run this node's backward reduction
    z = _BatchNorm.apply(x, gamma, beta, running_mean, running_var, num_batches_tracked, float(eps), float(momentum),
                         ACT[pre_act], ACT[post_act], bool(training), residual, link)
    if link is not None:
        z._bn_link = link
    return z


class _Bn2AddAct(_FastFunction):
    """y = act(BN_A(pre(xa)) + BN_B(pre(xb))), both BatchNorms in train mode (CrossCNNBlock junction)"""

    @staticmethod
    def forward(ctx, xa, gA, bA, xb, gB, bB, bufs, eps, momentum, pre, act_kind):
        _chk(xa, xb, gA, bA, gB, bB)
        C = xa.shape[-1]
        M = xa.numel() // C
        dc = dtype_code(xa.dtype)
        st, ss = [], []
        for x, g, b, (rm, rv, nbt) in ((xa, gA, bA, bufs[0]), (xb, gB, bB, bufs[1])):
            fused = getattr(x, '_bn_sums', None)
            if fused is not None and fused[1] == pre and fused[0].numel() == 2 * C:
                sums = fused[0]
            else:
                sums = ZERO.get((2 * C,), torch.float64, x.device)
                lib.bn_stats(x, M, C, pre, sums, dc)
            mr = torch.empty(2 * C, device=x.device, dtype=torch.float32)
            ab = torch.empty(2 * C, device=x.device, dtype=torch.float32)
            st += [mr, ab]
            ss.append(sums)
        y = torch.empty_like(xa)
        (rmA, rvA, nbtA), (rmB, rvB, nbtB) = bufs       # both statistics finalisations ride on the junction launch
        lib.bn2_add_act_train(xa, xb, y, M, C, ss[0], gA, bA, rmA, rvA, nbtA, st[0], st[1], ss[1], gB, bB, rmB, rvB, nbtB, st[2], st[3],
                              eps, momentum, pre, act_kind, dc)
        ctx.save_for_backward(xa, xb, *st)
        ctx.cfg = (pre, act_kind)
        ctx.params = (gA, bA, gB, bB)
        return y

    @staticmethod
    def backward(ctx, dy):
        xa, xb, mrA, abA, mrB, abB = ctx.saved_tensors
        pre, act_kind = ctx.cfg
        gA, bA, gB, bB = ctx.params
        dy = _as(dy, xa.dtype)
        C = xa.shape[-1]
        M = xa.numel() // C
        dc = dtype_code(xa.dtype)
        sums = ZERO.get((4 * C,), torch.float64, xa.device)
        lib.bn2_add_act_bwd_reduce(xa, xb, dy, M, C, mrA, abA, mrB, abB, pre, act_kind, sums, dc)
        dxa, dxb = torch.empty_like(xa), torch.empty_like(xb)
        outs = []
        for p in (gA, bA, gB, bB):
            use_slot = ZERO.active and getattr(p, '_grad_slot', None) is not None
            outs.append(_grad_out(p) if use_slot else torch.empty(C, device=xa.device, dtype=torch.float32))
        lib.bn2_add_act_bwd_apply(xa, xb, dy, dxa, dxb, M, C, mrA, abA, mrB, abB, sums, pre, act_kind, outs[0], outs[1], outs[2],
                                  outs[3], dc)
        return (dxa, _ret(outs[0], gA), _ret(outs[1], bA), dxb, _ret(outs[2], gB), _ret(outs[3], bB), None, None, None, None, None)


def bn2_add_act(xa, bnA, xb, bnB, pre_act='lrelu', act_kind='gelu'):
    """bnA/bnB: (gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum) of the two train-mode BatchNorms"""
    gA, bA, rmA, rvA, nbtA, eps, mom = bnA
    gB, bB, rmB, rvB, nbtB, _, _ = bnB
    return _Bn2AddAct.apply(xa, gA, bA, xb, gB, bB, ((rmA, rvA, nbtA), (rmB, rvB, nbtB)), float(eps), float(mom), ACT[pre_act],
                            ACT[act_kind])


class _LayerNorm(_FastFunction):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, fork=False):
        """fork: also return an alias of x for the residual path around the normalisation; its gradient is then added inside the
        LayerNorm backward kernel (tcct_layernorm_bwd_add) instead of by an autograd accumulation pass"""
        ctx.set_materialize_grads(False)      # an unused output must arrive as None in backward(), not as a zero-filled tensor
        _chk(x, gamma, beta)
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        mr = torch.empty(2 * M, device=x.device, dtype=torch.float32)
        lib.layernorm_fwd(x, y, M, C, gamma, beta, eps, mr, dtype_code(x.dtype))
        ctx.save_for_backward(x, gamma, mr)
        ctx.beta_param = beta
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, gamma, mr = ctx.saved_tensors
        if dy is None:
            return dskip, None, None, None, None
        dy = _as(dy, x.dtype)
        C = x.shape[-1]
        M = x.numel() // C
        dx = torch.empty_like(x)
        dg = _grad_out(gamma)
        db = _grad_out(ctx.beta_param)
        if dskip is None:
            lib.layernorm_bwd(x, dy, dx, M, C, gamma, mr, dg, db, dtype_code(x.dtype))
        else:
            lib.layernorm_bwd_add(x, dy, _as(dskip, x.dtype), dx, M, C, gamma, mr, dg, db, dtype_code(x.dtype))
        return dx, _ret(dg, gamma), _ret(db, ctx.beta_param), None, None


def layernorm(x, gamma, beta, eps=1e-6):
    return _LayerNorm.apply(x, gamma, beta, float(eps), False)


def layernorm_fork(x, gamma, beta, eps=1e-6):
    """(LayerNorm(x), x'): x' aliases x and is to be read by the residual path (see _LayerNorm.forward `fork`)"""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return layernorm(x, gamma, beta, eps), x
    return _LayerNorm.apply(x, gamma, beta, float(eps), True)


# ------------------------------------------------------------------------------------------- elementwise
class _Act(_FastFunction):
    @staticmethod
    def forward(ctx, x, kind):
        _chk(x)
        y = torch.empty_like(x)
        lib.act_fwd(x, y, x.numel(), kind, dtype_code(x.dtype))
        ctx.save_for_backward(x)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _as(dy, x.dtype)
        dx = torch.empty_like(x)
        lib.act_bwd(x, dy, dx, x.numel(), ctx.kind, dtype_code(x.dtype))
        return dx, None


def act(x, kind):
    return _Act.apply(x, ACT[kind])


class _AddAct(_FastFunction):
    @staticmethod
    def forward(ctx, a, b, kind):
        _chk(a, b)
        y = torch.empty_like(a)
        if kind == 0:
            lib.add(a, b, y, a.numel(), dtype_code(a.dtype))
        else:
            lib.add_act_fwd(a, b, y, a.numel(), kind, dtype_code(a.dtype))
            ctx.save_for_backward(a, b)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.kind == 0:
            return dy, dy, None
        a, b = ctx.saved_tensors
        dy = _as(dy, a.dtype)
        dx = torch.empty_like(a)
        lib.add_act_bwd(a, b, dy, dx, a.numel(), ctx.kind, dtype_code(a.dtype))
        return dx, dx, None


def add_act(a, b, kind=None):
    """act(a + b)"""
    return _AddAct.apply(a, b, ACT[kind])


def add(a, b):
    return _AddAct.apply(a, b, 0)


class _Residual(_FastFunction):
    @staticmethod
    def forward(ctx, x, z, scale):
        _chk(x, z, scale)
        B = x.shape[0]
        y = torch.empty_like(x)
        lib.residual_fwd(x, z, scale, y, B, x.numel() // B, dtype_code(x.dtype))
        ctx.save_for_backward(scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        (scale,) = ctx.saved_tensors
        dy = _c(dy)
        B = dy.shape[0]
        dz = torch.empty_like(dy)
        lib.scale_rows(dy, scale, dz, B, dy.numel() // B, dtype_code(dy.dtype))
        return dy, dz, None


def residual(x, z, scale=None):
    """x + scale[b] * z  (scale: fp32 [B] DropPath mask/keep_prob, or None for a plain add)"""
    if scale is None:
        return add(x, z)
    return _Residual.apply(x, z, scale)


class _Concat2(_FastFunction):
    @staticmethod
    def forward(ctx, a, b):
        _chk(a, b)
        Ca, Cb = a.shape[-1], b.shape[-1]
        M = a.numel() // Ca
        y = torch.empty(a.shape[:-1] + (Ca + Cb,), device=a.device, dtype=a.dtype)
        lib.concat2(a, b, y, M, Ca, Cb, dtype_code(a.dtype))
        ctx.cfg = (Ca, Cb)
        return y

    @staticmethod
    def backward(ctx, dy):
        Ca, Cb = ctx.cfg
        dy = _c(dy)
        M = dy.numel() // (Ca + Cb)
        da = torch.empty(dy.shape[:-1] + (Ca,), device=dy.device, dtype=dy.dtype)
        db = torch.empty(dy.shape[:-1] + (Cb,), device=dy.device, dtype=dy.dtype)
        lib.split2(dy, da, db, M, Ca, Cb, dtype_code(dy.dtype))
        return da, db


def concat2(a, b):
    return _Concat2.apply(a, b)


class _Add3Scale(_FastFunction):
    @staticmethod
    def forward(ctx, a, b, c, alpha):
        _chk(a, b, c)
        y = torch.empty_like(a)
        lib.add3_scale(a, b, c, y, a.numel(), alpha, dtype_code(a.dtype))
        ctx.alpha = alpha
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        d = torch.empty_like(dy)
        lib.scale(dy, d, dy.numel(), ctx.alpha, dtype_code(dy.dtype))
        return d, d, d, None


class _GateFusion(_FastFunction):
    @staticmethod
    def forward(ctx, x1, x2, field):
        _chk(x1, x2, field)
        N, H, W, C = x1.shape
        hs, ws = field.shape[1], field.shape[2]
        if field.dtype != torch.float32 or tuple(field.shape) != (N, hs, ws, C) or x2.shape != x1.shape:
            raise TcctError('gate_fusion: field must be fp32 NHWC [N,hs,ws,C] and x2 like x1')
        y = torch.empty_like(x1)
        lib.gate_fusion_fwd(x1, x2, field, y, N, H, W, C, hs, ws, dtype_code(x1.dtype))
        ctx.save_for_backward(field)
        return y

    @staticmethod
    def backward(ctx, dy):
        (field,) = ctx.saved_tensors
        dy = _c(dy)
        N, H, W, C = dy.shape
        d1, d2 = torch.empty_like(dy), torch.empty_like(dy)
        lib.gate_fusion_bwd(dy, field, d1, d2, N, H, W, C, field.shape[1], field.shape[2], dtype_code(dy.dtype))
        return d1, d2, None


def gate_fusion(x1, x2, field):
    """GateFusion training branch: x1*alpha + x2*(1-alpha), alpha = clamp(bicubic(field -> H x W), 0, 1); field fp32 NHWC [N,hs,ws,C]"""
    return _GateFusion.apply(x1, x2, field)


def add3_scale(a, b, c, alpha):
    return _Add3Scale.apply(a, b, c, float(alpha))


# -------------------------------------------------------------------------------------- pooling / resize
class _MetaPool(_FastFunction):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        B, N, C = x.shape
        y = torch.empty_like(x)
        lib.metapool_fwd(x, y, B, N, C, dtype_code(x.dtype))
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        B, N, C = dy.shape
        dx = torch.empty_like(dy)
        lib.metapool_bwd(dy, dx, B, N, C, dtype_code(dy.dtype))
        return dx


class _MetaPoolResidual(_FastFunction):
    """t + scale[b] * metapool(cur): mixer branch, DropPath scale and residual add in one pass (and one pass backward)"""

    @staticmethod
    def forward(ctx, cur, t, scale):
        _chk(cur, t, scale)
        B, N, C = cur.shape
        y = torch.empty_like(cur)
        lib.metapool_residual_fwd(cur, t, scale, y, B, N, C, dtype_code(cur.dtype))
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        B, N, C = dy.shape
        dcur = torch.empty_like(dy)
        lib.metapool_scaled_bwd(dy, ctx.scale, dcur, B, N, C, dtype_code(dy.dtype))
        return dcur, dy, None


def metapool_residual(cur, t, scale=None):
    """t + scale[b] * MetaPool(cur) on tokens [B,N,C] (scale: fp32 [B] or None)"""
    return _MetaPoolResidual.apply(cur, t, scale)


class _LnMetaPoolResidual(_FastFunction):
    """t + scale[b] * (pool(LN(t)) - LN(t)): MHCABlock's first half (reference nets/tcct.py:457-465, MetaPool :405-415) as ONE pass each way
    (csrc/ln_pool.hip): the normalised tensor and its gradient are never written"""

    @staticmethod
    def forward(ctx, t, gamma, beta, eps, scale):
        _chk(t, gamma, beta, scale)
        B, N, C = t.shape
        y = torch.empty_like(t)
        lib.ln_metapool_residual_fwd(t, y, B, N, C, gamma, beta, eps, scale, dtype_code(t.dtype))
        ctx.save_for_backward(t, gamma)
        ctx.beta_param, ctx.scale, ctx.eps = beta, scale, eps
        return y

    @staticmethod
    def backward(ctx, dy):
        t, gamma = ctx.saved_tensors
        dy = _as(dy, t.dtype)
        B, N, C = t.shape
        dt = torch.empty_like(t)
        dg, db = _grad_out(gamma), _grad_out(ctx.beta_param)
        lib.ln_metapool_residual_bwd(t, dy, dt, B, N, C, gamma, ctx.eps, ctx.scale, dg, db, dtype_code(t.dtype))
        return dt, _ret(dg, gamma), _ret(db, ctx.beta_param), None, None


class _LnMetaPoolResidualLn(_FastFunction):
    """(t1, LN2(t1)) with t1 = t + scale[b] * (pool(LN1(t)) - LN1(t)): MHCABlock up to the input of its Mlp (reference nets/tcct.py:457-466) from ONE forward
    pass -- the second LayerNorm is taken of the row while it is in registers.  Backward: LayerNorm-2 backward (+ the residual path's gradient of t1), then
    the one-pass backward of the first half."""

    @staticmethod
    def forward(ctx, t, g1, b1, eps1, scale, g2, b2, eps2):
        ctx.set_materialize_grads(False)
        _chk(t, g1, b1, scale, g2, b2)
        B, N, C = t.shape
        y, y2 = torch.empty_like(t), torch.empty_like(t)
        mr2 = torch.empty(2 * B * N, device=t.device, dtype=torch.float32)
        lib.ln_metapool_residual_ln_fwd(t, y, y2, B, N, C, g1, b1, eps1, scale, g2, b2, eps2, mr2, dtype_code(t.dtype))
        ctx.save_for_backward(t, y, g1, g2, mr2)
        ctx.beta = (b1, b2)
        ctx.cfg = (eps1, scale)
        ctx.mark_non_differentiable(mr2)
        return y, y2, mr2

    @staticmethod
    def backward(ctx, dres, dcur, _dmr=None):
        t, y, g1, g2, mr2 = ctx.saved_tensors
        b1, b2 = ctx.beta
        eps1, scale = ctx.cfg
        B, N, C = t.shape
        dc = dtype_code(t.dtype)
        dg2 = db2 = None
        if dcur is not None:
            dcur = _as(dcur, t.dtype)
            dt1 = torch.empty_like(t)
            dg2, db2 = _grad_out(g2), _grad_out(b2)
            if dres is None:
                lib.layernorm_bwd(y, dcur, dt1, B * N, C, g2, mr2, dg2, db2, dc)
            else:
                lib.layernorm_bwd_add(y, dcur, _as(dres, t.dtype), dt1, B * N, C, g2, mr2, dg2, db2, dc)
        else:
            dt1 = _as(dres, t.dtype)
        dt = torch.empty_like(t)
        dg1, db1 = _grad_out(g1), _grad_out(b1)
        lib.ln_metapool_residual_bwd(t, dt1, dt, B, N, C, g1, eps1, scale, dg1, db1, dc)
        return dt, _ret(dg1, g1), _ret(db1, b1), None, None, _ret(dg2, g2), _ret(db2, b2), None


LN_POOL_FUSE = True       # False: LayerNorm and the token mixer stay separate kernels (A/B timing)


def ln_metapool_residual_ok(t, gamma, beta):
    return (LN_POOL_FUSE and t.dim() == 3 and t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16) and t.shape[-1] % 8 == 0
            and 16 <= t.shape[-1] <= 192 and t.shape[0] <= 65535 and t.shape[1] < 2 ** 30 and gamma is not None and beta is not None)


def ln_metapool_residual(t, gamma, beta, eps=1e-6, scale=None):
    """t + scale[b] * MetaPool(LayerNorm(t)) on tokens [B,N,C] in one pass (scale: fp32 [B] or None); check ln_metapool_residual_ok first"""
    return _LnMetaPoolResidual.apply(t, gamma, beta, float(eps), scale)


LN_POOL_LN2 = True       # False: the second LayerNorm stays its own forward pass (A/B timing)


def ln_metapool_residual_ln(t, g1, b1, eps1, scale, g2, b2, eps2):
    """(t1, LayerNorm2(t1), its statistics [B*N*2]), t1 = t + scale[b] * MetaPool(LayerNorm1(t)): one forward pass; check ln_metapool_residual_ok first"""
    return _LnMetaPoolResidualLn.apply(t, g1, b1, float(eps1), scale, g2, b2, float(eps2))


def metapool(x):
    """x tokens [B,N,C]"""
    return _MetaPool.apply(x)


class _FactorAtt(_FastFunction):
    """FactorAtt_ConvRelPosEnc.forward between the qkv and proj Linear layers (reference nets/tcct.py:316-331 + ConvRelPosEnc.forward
    :265-287): out [B,N,C] = scale * q (softmax_N(k)^T v) + q * crpe(v), qkv [B,N,3C] (dormant in stc_tt)."""

    @staticmethod
    def forward(ctx, qkv, H, W, heads, scale, *wb):
        _chk(qkv, *wb)
        B, N, C3 = qkv.shape
        C = C3 // 3
        if C * 3 != C3 or H * W != N or C % heads or (C // heads) % 4:
            raise TcctError(f'factor_att: qkv {tuple(qkv.shape)} does not fit size ({H},{W}) / heads {heads} (Ch must be a multiple of 4)')
        if sum(w.shape[0] for w in wb[0::2]) != C:
            raise TcctError('factor_att: the crpe window splits do not add up to the channel count')
        Ch, dt, dev = C // heads, dtype_code(qkv.dtype), qkv.device
        ws = torch.empty(lib.fatt_kstats_workspace_bytes(B, N, C), device=dev, dtype=torch.uint8)
        stats = torch.empty((B, C, 2), device=dev, dtype=torch.float32)
        lib.fatt_kstats(qkv, ws, stats, B, N, C, heads, dt)
        M = ZERO.get((B, heads, Ch, Ch), torch.float32, dev)
        lib.fatt_ktv(qkv, stats, M, B, N, C, heads, dt)
        cv = torch.empty((B, N, C), device=dev, dtype=qkv.dtype)
        off = 0
        for w, b in zip(wb[0::2], wb[1::2]):
            Cg, K = w.shape[0], w.shape[2]
            lib.dwk_strided_fwd(qkv[0, 0, 2 * C + off:], 3 * C, w, b, cv[0, 0, off:], C, B, H, W, Cg, K, 0, 0, dt)
            off += Cg
        out = torch.empty((B, N, C), device=dev, dtype=qkv.dtype)
        lib.fatt_apply_fwd(qkv, M, cv, out, float(scale), B, N, C, heads, dt)
        ctx.save_for_backward(qkv, stats, M, cv, *wb[0::2])
        ctx.cfg = (H, W, heads, float(scale))
        ctx.params = wb
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, stats, M, cv, *ws = ctx.saved_tensors
        H, W, heads, scale = ctx.cfg
        wb = ctx.params
        B, N, C3 = qkv.shape
        C = C3 // 3
        dt, dev = dtype_code(qkv.dtype), qkv.device
        dout = _as(dout, qkv.dtype)
        dM = ZERO.get(tuple(M.shape), torch.float32, dev)
        lib.fatt_dktv(qkv, dout, dM, scale, B, N, C, heads, dt)
        dqkv = torch.empty_like(qkv)
        dcv = torch.empty_like(cv)
        lib.fatt_apply_bwd(qkv, stats, M, dM, cv, dout, dqkv, dcv, scale, B, N, C, heads, dt)
        grads, off = [], 0
        for w, b in zip(wb[0::2], wb[1::2]):
            Cg, K = w.shape[0], w.shape[2]
            lib.dwk_strided_fwd(dcv[0, 0, off:], C, w, None, dqkv[0, 0, 2 * C + off:], 3 * C, B, H, W, Cg, K, 1, 1, dt)
            dw = _grad_out(w)
            db = _grad_out(b) if b is not None else None
            lib.dwk_strided_wgrad(qkv[0, 0, 2 * C + off:], 3 * C, dcv[0, 0, off:], C, dw, db, B, H, W, Cg, K, dt)
            grads += [_ret(dw, w), _ret(db, b)]
            off += Cg
        return (dqkv, None, None, None, None) + tuple(grads)


def factor_att(qkv, size, heads, scale, crpe_convs):
    """qkv [B,N,3C] tokens, size = (H, W), crpe_convs = the nn.Conv2d list of ConvRelPosEnc (windows 3/5/7 over head splits)"""
    wb = []
    for m in crpe_convs:
        wb += [m.weight, m.bias]
    return _FactorAtt.apply(qkv, int(size[0]), int(size[1]), int(heads), float(scale), *wb)


class _MaxPool2(_FastFunction):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        N, H, W, C = x.shape
        y = torch.empty((N, H // 2, W // 2, C), device=x.device, dtype=x.dtype)
        lib.maxpool2_fwd(x, y, N, H, W, C, dtype_code(x.dtype))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _as(dy, x.dtype)
        N, H, W, C = x.shape
        dx = torch.empty_like(x)
        lib.maxpool2_bwd(x, dy, dx, N, H, W, C, dtype_code(x.dtype))
        return dx


class _MaxPool2Fork(_FastFunction):
    """(maxpool2(x), x): the second output aliases x and is what the OTHER consumers of x read, so that in the backward pass their
    gradient arrives here and is added inside the pooling scatter kernel (autograd would otherwise run an accumulation pass)"""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)      # an unused output must arrive as None in backward(), not as a zero-filled tensor
        _chk(x)
        N, H, W, C = x.shape
        y = torch.empty((N, H // 2, W // 2, C), device=x.device, dtype=x.dtype)
        lib.maxpool2_fwd(x, y, N, H, W, C, dtype_code(x.dtype))
        ctx.save_for_backward(x)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        (x,) = ctx.saved_tensors
        N, H, W, C = x.shape
        if dy is None:
            return dskip
        dy = _as(dy, x.dtype)
        dx = torch.empty_like(x)
        if dskip is None:
            lib.maxpool2_bwd(x, dy, dx, N, H, W, C, dtype_code(x.dtype))
        else:
            lib.maxpool2_bwd_add(x, dy, _as(dskip, x.dtype), dx, N, H, W, C, dtype_code(x.dtype))
        return dx


def maxpool2(x):
    return _MaxPool2.apply(x)


def maxpool2_fork(x):
    """(maxpool2(x), x'): use x' (an alias of x) for every other consumer of x; see _MaxPool2Fork"""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return _MaxPool2.apply(x), x
    return _MaxPool2Fork.apply(x)


class _Bilinear(_FastFunction):
    @staticmethod
    def forward(ctx, x, Ho, Wo, align, res=None):
        _chk(x)
        N, H, W, C = x.shape
        y = torch.empty((N, Ho, Wo, C), device=x.device, dtype=x.dtype)
        if res is not None:
            _chk(res)
            if tuple(res.shape) != (N, Ho, Wo, C) or res.dtype != x.dtype:
                raise TcctError('bilinear: residual must have the output shape and the input dtype')
            lib.bilinear_add_fwd(x, res, y, N, H, W, C, Ho, Wo, int(align), dtype_code(x.dtype))
        else:
            lib.bilinear_fwd(x, y, N, H, W, C, Ho, Wo, int(align), dtype_code(x.dtype))
        ctx.cfg = (N, H, W, C, Ho, Wo, int(align))
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C, Ho, Wo, align = ctx.cfg
        dy = _c(dy)
        dx = torch.empty((N, H, W, C), device=dy.device, dtype=dy.dtype)
        if dy.dtype == torch.float32 and C % 4 != 0 and Ho >= 4 * H and Wo >= 4 * W and (2 * Wo) // W + 3 <= 40 and Wo * C * 4 <= 65536:
            # narrow fp32 maps enlarged >= 4x (the coarse aux logits): separable transpose, one coalesced pass over dy (x8: 0.28 -> 0.08 ms;
            # at x2 the 2-D table gather is still faster)
            ws = torch.empty(N * Ho * W * C, device=dy.device, dtype=torch.float32)
            lib.bilinear_bwd_separable(dy, dx, ws, N, H, W, C, Ho, Wo, align)
        else:
            lib.bilinear_bwd(dy, dx, N, H, W, C, Ho, Wo, align, dtype_code(dy.dtype))
        return dx, None, None, None, (dy if ctx.has_res else None)


def bilinear(x, size, align_corners, residual=None):
    """F.interpolate(mode='bilinear') on NHWC; residual (output-shaped) is added in the same pass (decoder skip connections)"""
    if tuple(x.shape[1:3]) == tuple(size):
        return x if residual is None else add(x, residual)      # identity resize (torch returns the same values)
    return _Bilinear.apply(x, int(size[0]), int(size[1]), bool(align_corners), residual)


class _L2Norm(_FastFunction):
    @staticmethod
    def forward(ctx, x, eps):
        _chk(x)
        C = x.shape[-1]
        y = torch.empty_like(x)
        lib.l2norm_fwd(x, y, x.numel() // C, C, eps, dtype_code(x.dtype))
        ctx.save_for_backward(x)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _as(dy, x.dtype)
        C = x.shape[-1]
        dx = torch.empty_like(x)
        lib.l2norm_bwd(x, dy, dx, x.numel() // C, C, ctx.eps, dtype_code(x.dtype))
        return dx, None


def l2norm(x, eps=1e-12):
    return _L2Norm.apply(x, float(eps))


# ---- the feature-polarization gradient without its tensor (round 4) ---------------------------------------------------------------------------
# regular_udh's d loss / d feats is g * dpro[label[p]][bin[p]] -- a table lookup by two bytes per pixel.  When feats comes out of the fused norm_add
# node and the FPL is its only differentiable consumer (how FTC / RegNet use it), _Fpl.backward returns a zero-stride placeholder of the right shape
# and registers (labels, binmap, table, upstream gradient) under the placeholder's storage; _NormAdd.backward picks the recipe up and runs the
# lookup forms of its three kernels (tcct_l2norm_bwd_fplgrad, tcct_bilinear_bwd_fplgrad): the 452 MB gradient (bench shape) is neither written nor
# read three times.  A placeholder that autograd had to ADD to another gradient (a second differentiable consumer of feats, the FPL evaluated twice
# on the same feats) arrives as a dense tensor without the FPL part: every recipe is therefore ALSO filed under its producer (`pending`), and a
# producer whose incoming gradient is not the single placeholder materialises the pending recipes (tcct_fpl_backward) and adds them.  A feats tensor
# with hooks / retain_grad (somebody wants to SEE the gradient) never takes the lazy path.  `FPL_LAZY_GRAD = False` (bench.py --set FPL_LAZY_GRAD=0) restores the dense tensor everywhere.
FPL_LAZY_GRAD = True
_FPL_LAZY = {'producers': set(), 'grads': {}, 'pending': {}}


def fpl_lazy_grad_reset():
    """per step (begin_step): forget last step's registrations (storage addresses are recycled by the allocator)"""
    _FPL_LAZY['producers'].clear()
    _FPL_LAZY['grads'].clear()
    _FPL_LAZY['pending'].clear()


class _NormAdd(_FastFunction):
    """norm_add([g0, g1, g2]) (reference nets/tcct.py:937-942): mean of the three L2-normalised maps at g0's size.  Forward: one pass
    (tcct_normadd_fwd); backward: the existing resize / normalise gradients with the 1/3 folded into the last kernel of each chain."""

    @staticmethod
    def forward(ctx, g0, g1, g2, eps, fork=False):
        """fork: also return aliases of g0 / g1 / g2 for their other consumers (the aux heads); the gradients those deliver are added inside this
        node's last backward kernels (tcct_l2norm_bwd_scaled_add) instead of by three autograd accumulation passes"""
        ctx.set_materialize_grads(False)
        _chk(g0, g1, g2)
        N, H, W, C = g0.shape
        (_, h1, w1, _), (_, h2, w2, _) = g1.shape, g2.shape
        out = torch.empty_like(g0)
        inv1 = torch.empty(N * h1 * w1, device=g0.device, dtype=torch.float32)
        inv2 = torch.empty(N * h2 * w2, device=g0.device, dtype=torch.float32)
        lib.normadd_fwd(g0, g1, g2, inv1, inv2, out, N, H, W, C, h1, w1, h2, w2, eps, dtype_code(g0.dtype))
        ctx.save_for_backward(g0, g1, g2)
        ctx.eps = eps
        ctx.out_ptr = out.data_ptr()
        if fork and C == 32:
            _FPL_LAZY['producers'].add(out.data_ptr())
        return (out, g0.view_as(g0), g1.view_as(g1), g2.view_as(g2)) if fork else out

    @staticmethod
    def backward(ctx, dy, *dalias):
        g0, g1, g2 = ctx.saved_tensors
        dalias = tuple(dalias) + (None,) * (3 - len(dalias))
        pending = _FPL_LAZY['pending'].pop(ctx.out_ptr, [])        # recipes of every _Fpl node that consumed THIS node's output
        if dy is None and not pending:          # only the aliases were used downstream
            return dalias[0], dalias[1], dalias[2], None, None
        N, H, W, C = g0.shape
        dc = dtype_code(g0.dtype)
        lazy = None
        if dy is not None and len(pending) == 1 and dy.dim() == 4 and dy.stride(0) == 0 and dy.data_ptr() == pending[0][0].data_ptr():
            lazy = _FPL_LAZY['grads'].pop(dy.data_ptr(), None)     # the single placeholder, untouched by autograd
        elif pending:
            # autograd ADDED the placeholder(s) to another gradient (or to each other): the dense sum lacks the FPL part -- materialise it
            dy = None if dy is None else _c(_as(dy, g0.dtype))
            for marker, labels, binmap, dpro, gup, ncls in pending:
                _FPL_LAZY['grads'].pop(marker.data_ptr(), None)
                dfeat = torch.empty_like(g0)
                lib.fpl_backward(labels, binmap, dpro, gup, 1.0, int(labels.numel()), dfeat, dc)
                dy = dfeat if dy is None else add(dy, dfeat)
        if lazy is not None:    # the gradient is the feature-polarization loss's: looked up, never materialised
            _, labels, binmap, dpro, gup, ncls = lazy
            outs = []
            for g, da in zip((g0, g1, g2), dalias):
                h, w = g.shape[1], g.shape[2]
                d = torch.empty_like(g)
                da = _c(_as(da, g.dtype)) if da is not None else None
                if (h, w) == (H, W):
                    lib.l2norm_bwd_fplgrad(g, labels, binmap, dpro, gup, 1.0, ncls, da, d, g.numel() // C, ctx.eps, 1.0 / 3.0, dc)
                else:
                    dn = torch.empty_like(g)
                    lib.bilinear_bwd_fplgrad(labels, binmap, dpro, gup, 1.0, ncls, dn, N, h, w, H, W, 0, dc)
                    if da is not None:
                        lib.l2norm_bwd_scaled_add(g, dn, da, d, g.numel() // C, C, ctx.eps, 1.0 / 3.0, dc)
                    else:
                        lib.l2norm_bwd_scaled(g, dn, d, g.numel() // C, C, ctx.eps, 1.0 / 3.0, dc)
                outs.append(d)
            return outs[0], outs[1], outs[2], None, None
        dy = _c(_as(dy, g0.dtype))
        outs = []
        for g, da in zip((g0, g1, g2), dalias):
            h, w = g.shape[1], g.shape[2]
            if (h, w) == (H, W):
                dn = dy
            else:
                dn = torch.empty_like(g)
                lib.bilinear_bwd(dy, dn, N, h, w, C, H, W, 0, dc)
            d = torch.empty_like(g)
            if da is not None:
                lib.l2norm_bwd_scaled_add(g, dn, _c(_as(da, g.dtype)), d, g.numel() // C, C, ctx.eps, 1.0 / 3.0, dc)
            else:
                lib.l2norm_bwd_scaled(g, dn, d, g.numel() // C, C, ctx.eps, 1.0 / 3.0, dc)
            outs.append(d)
        return outs[0], outs[1], outs[2], None, None


def norm_add3(g0, g1, g2, eps=1e-12):
    """fused norm_add for three maps with the same channel count (C % 4 == 0, C/4 a power of two <= 64); g1 / g2 coarser than g0"""
    C = g0.shape[-1]
    lp = C // 4
    ok = (g0.dim() == 4 and g1.shape[-1] == C and g2.shape[-1] == C and C % 4 == 0 and 1 <= lp <= 64 and lp & (lp - 1) == 0
          and g0.dtype == g1.dtype == g2.dtype and tuple(g1.shape[1:3]) != tuple(g0.shape[1:3]) and tuple(g2.shape[1:3]) != tuple(g0.shape[1:3]))
    if not ok:
        size = tuple(g0.shape[1:3])
        return add3_scale(l2norm(g0, eps), bilinear(l2norm(g1, eps), size, False), bilinear(l2norm(g2, eps), size, False), 1.0 / 3.0)
    return _NormAdd.apply(g0, g1, g2, float(eps))


def norm_add3_fork(g0, g1, g2, eps=1e-12):
    """(norm_add3(g0, g1, g2), g0', g1', g2'): the aliases are to be read by the other consumers of the three maps (the aux heads); returns None
    when the fused form does not apply (the caller then uses norm_add3 and lets autograd accumulate)"""
    C = g0.shape[-1]
    lp = C // 4
    ok = (g0.dim() == 4 and g1.shape[-1] == C and g2.shape[-1] == C and C % 4 == 0 and 1 <= lp <= 64 and lp & (lp - 1) == 0
          and g0.dtype == g1.dtype == g2.dtype and tuple(g1.shape[1:3]) != tuple(g0.shape[1:3]) and tuple(g2.shape[1:3]) != tuple(g0.shape[1:3])
          and torch.is_grad_enabled())
    if not ok:
        return None
    return _NormAdd.apply(g0, g1, g2, float(eps), True)


# ------------------------------------------------------------------------------------------------ losses
class _SoftmaxDice(_FastFunction):
    @staticmethod
    def forward(ctx, logits, labels):
        _chk(logits, labels)
        C = logits.shape[-1]
        M = logits.numel() // C
        sums = torch.empty(3 * C, device=logits.device, dtype=torch.float64)
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        lib.softmax_dice_fwd(logits, labels, M, C, sums, loss, dtype_code(logits.dtype))
        ctx.save_for_backward(logits, labels, sums)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, labels, sums = ctx.saved_tensors
        C = logits.shape[-1]
        M = logits.numel() // C
        g = _as(g, torch.float32)
        d = torch.empty_like(logits)
        lib.softmax_dice_bwd(logits, labels, M, C, sums, g, 1.0, d, dtype_code(logits.dtype))
        return d, None


def softmax_dice(logits, labels):
    """MultiLoss(DiceLoss): logits NHWC [N,H,W,C], labels uint8 [N,H,W] -> scalar (sum over classes of 1-dice)."""
    return _SoftmaxDice.apply(logits, labels)


class _UpDice(_FastFunction):
    """MultiLoss(DiceLoss)(F.interpolate(low, size, 'bilinear'), labels) without materialising the resized logits (deep-supervision heads)"""

    @staticmethod
    def forward(ctx, low, labels, H, W):
        _chk(low, labels)
        B, h, w, C = low.shape
        sums = torch.empty(3 * C, device=low.device, dtype=torch.float64)
        loss = torch.empty((), device=low.device, dtype=torch.float32)
        lib.updice_fwd(low, labels, B, h, w, H, W, C, sums, loss)
        ctx.save_for_backward(low, labels, sums)
        ctx.size = (H, W)
        return loss

    @staticmethod
    def backward(ctx, g):
        low, labels, sums = ctx.saved_tensors
        B, h, w, C = low.shape
        H, W = ctx.size
        g = _as(g, torch.float32)
        ws = torch.empty((B, H, w, C), device=low.device, dtype=torch.float32)
        d = torch.empty_like(low)
        lib.updice_bwd(low, labels, B, h, w, H, W, C, sums, g, 1.0, ws, d)
        return d, None, None, None


class _DeepSupervisionDice(_FastFunction):
    """KiteSeg.grad_calc with MultiLoss(DiceLoss) (reference kite/loopback.py:62-73): sum_{i=3,2,1} coff * Dice(resize(low_i)) + Dice(logits0) as ONE node: one
    memset, four sums kernels, one finalisation; the scalar multiplications and additions of the loop (a dozen 5-us launches on the single-stream stretch of the
    step, each way) are arithmetic inside the finalisation kernel / the grad_scale argument of the backward kernels."""

    @staticmethod
    def forward(ctx, logits0, labels, coff, H, W, *lows):
        _chk(logits0, labels, *lows)
        B, _, _, C = logits0.shape
        sums = torch.empty(4 * 3 * C, device=logits0.device, dtype=torch.float64)
        loss = torch.empty((), device=logits0.device, dtype=torch.float32)
        a = []
        for i in range(3):
            a += [lows[i], lows[i].shape[1], lows[i].shape[2]] if i < len(lows) else [None, 0, 0]
        lib.dice_ds_fwd(logits0, dtype_code(logits0.dtype), labels, B, H, W, C, *a, coff, sums, loss)
        ctx.save_for_backward(logits0, labels, sums, *lows)
        ctx.cfg = (coff, H, W)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits0, labels, sums, *lows = ctx.saved_tensors
        coff, H, W = ctx.cfg
        B, _, _, C = logits0.shape
        g = _as(g, torch.float32)
        d0 = torch.empty_like(logits0)
        lib.softmax_dice_bwd(logits0, labels, logits0.numel() // C, C, sums[:3 * C], g, 1.0, d0, dtype_code(logits0.dtype))
        dl = []
        for i, low in enumerate(lows):
            _, h, w, _ = low.shape
            ws = torch.empty((B, H, w, C), device=low.device, dtype=torch.float32)
            d = torch.empty_like(low)
            lib.updice_bwd(low, labels, B, h, w, H, W, C, sums[(i + 1) * 3 * C:(i + 2) * 3 * C], g, coff, ws, d)
            dl.append(d)
        return (d0, None, None, None, None) + tuple(dl)


DS_DICE_FUSE = True        # False: one criterion node per head + torch scalar arithmetic (A/B timing)


def deep_supervision_dice_ok(outs, coff):
    """outs = [logits0 (NCHW view of NHWC memory or NHWC), LowResLogits x 1..3]"""
    return (DS_DICE_FUSE and isinstance(outs, (list, tuple)) and 2 <= len(outs) <= 4 and all(isinstance(o, LowResLogits) and o.fusable() for o in outs[1:])
            and torch.is_tensor(outs[0]) and outs[0].dim() == 4 and len({o.size for o in outs[1:]}) == 1)


def deep_supervision_dice(logits0_nhwc, labels, lows, coff):
    """sum_{i = n..1} coff * Dice(resize(lows[i-1])) + Dice(logits0): lows = [LowResLogits of outs[1], outs[2], ...] (the reference's loop runs from the last)"""
    H, W = lows[0].size
    return _DeepSupervisionDice.apply(logits0_nhwc, labels, float(coff), H, W, *[l_.low for l_ in lows])


# The HIP runtime multiplexes its streams onto FOUR hardware queues (GPU_MAX_HW_QUEUES; measured on the bench step: 3 queues 21.2 ms, 4 queues 20.26, 5 or more 24.6):
# streams beyond the fourth share a queue with another one and serialise against it.  The step therefore uses exactly four streams -- current, 'vit', 'vit_enc',
# 'wgrad' -- and the later forks REUSE the encoder streams, which are idle when the fusion / the losses run (forward: joined; backward: not started yet).
FUSE_STREAM_TAG = 'vit_enc'
REG_FORK = True          # RegNet.regular_reg: the `true` chain on the (idle) 'vit' stream beside the `pred` chain
LOSS_FORK = True         # KiteSeg.calc_loss: the boundary-regression loss on the (idle) 'vit_enc' stream beside Dice + feature polarization


def loss_fork_ok(logits):
    return (LOSS_FORK and STAGE_FORK_MAX_PIXELS > 0 and PARALLEL_BRANCHES and torch.is_grad_enabled() and logits.is_cuda and logits.requires_grad
            and logits.shape[0] * logits.shape[2] * logits.shape[3] >= (1 << 20))


class LowResLogits:
    """A deep-supervision head before its resize: `low` fp32 NHWC [B,h,w,C] + the target size.  FTC.forward returns these instead of
    the resized tensors when `defer_aux_resize` is set (KiteSeg.calc_loss does, in training); the Dice criterion consumes them with the
    fused resize + softmax + Dice kernels, anything else calls `.dense()` for the reference's [B,C,H,W] tensor."""

    def __init__(self, low, size):
        self.low, self.size = low, (int(size[0]), int(size[1]))

    def fusable(self):
        B, h, w, C = self.low.shape
        H, W = self.size
        return (self.low.dtype == torch.float32 and 2 <= C <= 8 and H % h == 0 and W % w == 0 and H // h == W // w and H // h in (2, 4, 8, 16))

    def dense(self):
        return bilinear(self.low, self.size, False).permute(0, 3, 1, 2)


def softmax_dice_upsampled(lr, labels):
    """Dice criterion of a LowResLogits head"""
    if not lr.fusable():
        return softmax_dice(bilinear(lr.low, lr.size, False), labels)
    return _UpDice.apply(lr.low, labels, lr.size[0], lr.size[1])


class _Slice(_FastFunction):
    @staticmethod
    def forward(ctx, x, start, n):
        _chk(x)
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty(x.shape[:-1] + (n,), device=x.device, dtype=torch.float32)
        lib.slice_channels_fwd(x, y, M, C, start, n, dtype_code(x.dtype))
        ctx.cfg = (C, start, n, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        C, start, n, dt = ctx.cfg
        dy = _as(dy, torch.float32)
        M = dy.numel() // n
        dx = torch.empty(dy.shape[:-1] + (C,), device=dy.device, dtype=dt)
        lib.slice_channels_bwd(dy, dx, M, C, start, n, dtype_code(dt))
        return dx, None, None


def slice_channels_f32(x, start, n):
    return _Slice.apply(x, start, n)


class _GumbelColSoftmax(_FastFunction):
    @staticmethod
    def forward(ctx, x, eps):
        _chk(x, eps)
        N, H, W, CH = x.shape
        out = torch.empty((N, H, W, 1), device=x.device, dtype=torch.float32)
        stats = torch.empty((N, W, CH, 3), device=x.device, dtype=torch.float32)
        lib.gumbel_colsoftmax_fwd(x, eps, out, stats, N, H, W, CH)
        ctx.save_for_backward(x, eps, stats)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, eps, stats = ctx.saved_tensors
        N, H, W, CH = x.shape
        dout = _as(dout, torch.float32)
        dx = torch.empty_like(x)
        lib.gumbel_colsoftmax_bwd(x, eps, stats, dout, dx, N, H, W, CH)
        return dx, None


def gumbel_colsoftmax_sum(x, eps):
    """x, eps fp32 [N,H,W,4] -> [N,H,W,1]: sum_c sampling_softmax over H (reg.py:118-128)"""
    return _GumbelColSoftmax.apply(x, eps)


class _ColSoftmax(_FastFunction):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        N, H, W = x.shape[:3]
        y = torch.empty_like(x)
        lib.colsoftmax_fwd(x, y, N, H, W)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        N, H, W = y.shape[:3]
        dy = _as(dy, torch.float32)
        dx = torch.empty_like(y)
        lib.colsoftmax_bwd(y, dy, dx, N, H, W)
        return dx


def colsoftmax(x):
    """softmax over H of fp32 [N,H,W,1]"""
    return _ColSoftmax.apply(x)


class _ColWSum(_FastFunction):
    @staticmethod
    def forward(ctx, x, wts):
        _chk(x, wts)
        N, H, W = x.shape[:3]
        out = torch.empty((N, W), device=x.device, dtype=torch.float32)
        lib.colwsum_fwd(x, wts, out, N, H, W)
        ctx.save_for_backward(wts)
        ctx.shape = x.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        (wts,) = ctx.saved_tensors
        N, H, W = ctx.shape[:3]
        dout = _as(dout, torch.float32)
        dx = torch.empty(ctx.shape, device=dout.device, dtype=torch.float32)
        lib.colwsum_bwd(dout, wts, dx, N, H, W)
        return dx, None


def colwsum(x, wts):
    """edge[n,w] = sum_h x[n,h,w]*wts[h]"""
    return _ColWSum.apply(x, wts)


class _Mse(_FastFunction):
    @staticmethod
    def forward(ctx, a, b):
        _chk(a, b)
        acc = torch.empty((), device=a.device, dtype=torch.float64)
        out = torch.empty((), device=a.device, dtype=torch.float32)
        lib.mse_fwd(a, b, a.numel(), acc, out)
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _as(g, torch.float32)
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        lib.mse_bwd(a, b, a.numel(), g, 1.0, da, db)
        return da, db


def mse(a, b):
    return _Mse.apply(a, b)


def label_planes(labels, start, n, want_onehot=True, want_edge=True):
    """labels uint8 [N,H,W] -> (onehot fp32 [N,H,W,n] of classes start.., edge fp32 [N,H,W,1])"""
    _chk(labels)
    N, H, W = labels.shape
    oh = torch.empty((N, H, W, n), device=labels.device, dtype=torch.float32) if want_onehot else None
    ed = torch.empty((N, H, W, 1), device=labels.device, dtype=torch.float32) if want_edge else None
    lib.label_planes(labels, oh, ed, N, H, W, start, n)
    return oh, ed


class _Fpl(_FastFunction):
    @staticmethod
    def forward(ctx, feat, logits, labels, buf_grad, allow_lazy=True):
        ctx.set_materialize_grads(False)      # an unused output must arrive as None in backward(), not as a zero-filled tensor
        _chk(feat, logits, labels, buf_grad)
        C = logits.shape[-1]
        M = logits.numel() // C
        dev = feat.device
        if feat.shape[-1] != 32:
            raise TcctError('fpl: feature width must be 32')
        prob = torch.empty(M, device=dev, dtype=torch.float32)
        lib.softmax_pick(logits, labels, M, C, prob, None, dtype_code(logits.dtype))
        counts = torch.empty(16, device=dev, dtype=torch.int32)        # FPL_MAXC
        pro_sum = torch.empty((C, 32, 32), device=dev, dtype=torch.float32)
        pro = torch.empty((C, 32, 32), device=dev, dtype=torch.float32)
        dpro = torch.empty((C, 32, 32), device=dev, dtype=torch.float32)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        binmap = torch.empty(M, device=dev, dtype=torch.uint8)
        # radix multi-select of the 32 bin boundaries per class + bin sums in pixel order (fpl_select.hip): no sort, no gather
        ws = torch.empty(int(lib.fpl_select_workspace_bytes()), device=dev, dtype=torch.uint8)
        lib.fpl_select(feat, labels, prob, M, C, ws, counts, binmap, pro_sum, dtype_code(feat.dtype))
        lib.fpl_loss(pro_sum, counts, buf_grad, C, pro, loss, dpro)
        ctx.save_for_backward(labels, binmap, dpro)
        ctx.cfg = (feat.shape, feat.dtype, M)
        # feat is the output of a norm_add node that can look its gradient up from (labels, bins, table) itself (FPL_LAZY_GRAD below)
        ctx.lazy = (FPL_LAZY_GRAD and allow_lazy and feat.data_ptr() in _FPL_LAZY['producers'] and feat.shape[-1] == 32 and feat.is_contiguous())
        ctx.producer = feat.data_ptr()
        ctx.mark_non_differentiable(pro)
        return loss, pro

    @staticmethod
    def backward(ctx, g, _gpro):
        labels, binmap, dpro = ctx.saved_tensors
        shape, dt, M = ctx.cfg
        if g is None:
            return None, None, None, None, None
        g = _as(g, torch.float32)
        if ctx.lazy:
            # d loss / d feat is a function of two bytes per pixel: hand norm_add's backward the recipe instead of the 452 MB tensor.  The returned
            # gradient is a zero-stride expansion of one zero element (right shape and dtype, no memory); the consumer recognises its storage.
            marker = torch.zeros(1, device=g.device, dtype=dt)
            recipe = (marker, labels, binmap, dpro, g, int(dpro.shape[0]))
            _FPL_LAZY['grads'][marker.data_ptr()] = recipe
            _FPL_LAZY['pending'].setdefault(ctx.producer, []).append(recipe)
            return marker.expand(shape), None, None, None, None
        dfeat = torch.empty(shape, device=g.device, dtype=dt)
        lib.fpl_backward(labels, binmap, dpro, g, 1.0, M, dfeat, dtype_code(dt))
        return dfeat, None, None, None, None


def grad_is_watched(t):
    """somebody hooks `t` or retains its gradient, i.e. wants to SEE d loss / d t.

    LIMITATION (only what exists when fpl() runs is visible here): `torch.autograd.grad(loss, feats)` and a hook registered on `feats` AFTER the loss was
    formed are not: inside a pooled training step they receive the zero-stride all-zero PLACEHOLDER of the lazy route (the real gradient travels as a
    (labels, bins, table) recipe into norm_add's backward kernels and still reaches every weight).  A caller who wants the dense d loss / d feats calls
    `feats.retain_grad()` / registers the hook BEFORE `RegNet.regular_udh`, or `ops.fpl(..., allow_lazy=False)`, or sets `ops.FPL_LAZY_GRAD = False`."""
    return bool(getattr(t, 'retains_grad', False)) or bool(getattr(t, '_backward_hooks', None))


def fpl(feat, logits, labels, buf_grad, allow_lazy=True):
    """regular_udh: feat NHWC [N,H,W,32], logits NHWC [N,H,W,C] (no grad flows to them), labels uint8 [N,H,W],
    buf_grad fp32 [C,32] -> (loss, prototypes [C,32,32]).  allow_lazy=False (or a watched `feat`): the gradient wrt feat is the dense tensor."""
    return _Fpl.apply(feat, logits.detach(), labels, buf_grad, bool(allow_lazy) and not grad_is_watched(feat))
