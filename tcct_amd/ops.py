"""torch.autograd.Function wrappers over the libtcct_hip.so C-ABI.

PyTorch here is plumbing only (device memory via the caching allocator, the current HIP stream, the autograd tape);
every arithmetic op of the hot path is a HIP kernel behind `tcct_amd._lib.lib`.  No CPU fallback exists: a CPU tensor
or a missing .so raises.  Activations are NHWC-contiguous (`[N,H,W,C]`, tokens `[B,N,C]` are the same memory).
"""
import torch

from ._lib import lib, dtype_code, TcctError, F32, BF16  # noqa: F401

ACT = {None: 0, 'none': 0, 'lrelu': 1, 'hswish': 2, 'gelu': 3, 'sigmoid': 4, 'abs': 5}


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
class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, padh, padw, out_dtype):
        _chk(x, w, bias)
        N, H, W, Cin = x.shape
        Cout, Cin_w, KH, KW = w.shape
        Ho = (H + 2 * padh - KH) // stride + 1
        Wo = (W + 2 * padw - KW) // stride + 1
        odt = out_dtype or x.dtype
        y = torch.empty((N, Ho, Wo, Cout), device=x.device, dtype=odt)
        lib.conv2d_fwd(x, w, bias, y, N, H, W, Cin, Cin_w, Cout, KH, KW, stride, padh, padw, dtype_code(x.dtype),
                       dtype_code(odt))
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padh, padw, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padh, padw, has_bias = ctx.cfg
        dy = _c(dy)
        N, H, W, Cin = x.shape
        Cout, Cin_w, KH, KW = w.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if stride != 1 or Cin != Cin_w:
                raise TcctError('conv2d dgrad: only stride-1 convs with unpadded channels need an input gradient')
            dx = torch.empty_like(x)
            lib.conv2d_dgrad(dy, w, dx, N, H, W, Cin, Cout, KH, KW, padh, padw, dtype_code(dy.dtype), dtype_code(x.dtype))
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(w)
            db = torch.empty(Cout, device=w.device, dtype=torch.float32) if has_bias else None
            lib.conv2d_wgrad(x, dy, dw, db, N, H, W, Cin, Cin_w, Cout, KH, KW, stride, padh, padw, dtype_code(x.dtype),
                             dtype_code(dy.dtype))
        return dx, dw, db, None, None, None, None


def conv2d(x, w, bias=None, stride=1, pad=0, out_dtype=None):
    """x [N,H,W,Cin] (or tokens [B,N,C]); w OIHW fp32 (or [Cout,Cin] for nn.Linear)."""
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    tok = x.dim() == 3
    if tok:
        x = x.unsqueeze(2)
    if w.dim() == 2:
        w = w.view(w.shape[0], w.shape[1], 1, 1)
    y = _Conv2d.apply(x, w, bias, stride, ph, pw, out_dtype)
    return y.squeeze(2) if tok else y


class _DwConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, add_input):
        _chk(x, w, bias)
        N, H, W, C = x.shape
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = torch.empty((N, Ho, Wo, C), device=x.device, dtype=x.dtype)
        lib.dwconv3x3_fwd(x, w, bias, y, N, H, W, C, stride, int(add_input), dtype_code(x.dtype))
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, add_input, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, add_input, has_bias = ctx.cfg
        dy = _as(dy, x.dtype)
        N, H, W, C = x.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            lib.dwconv3x3_dgrad(dy, w, dx, N, H, W, C, stride, int(add_input), dtype_code(x.dtype))
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(w)
            db = torch.empty(C, device=w.device, dtype=torch.float32) if has_bias else None
            lib.dwconv3x3_wgrad(x, dy, dw, db, N, H, W, C, stride, dtype_code(x.dtype))
        return dx, dw, db, None, None


def dwconv3x3(x, w, bias=None, stride=1, add_input=False):
    return _DwConv.apply(x, w, bias, stride, add_input)


# ------------------------------------------------------------------------------------------------- norms
class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rm, rv, nbt, eps, momentum, pre, post, training):
        _chk(x, gamma, beta)
        C = x.shape[-1]
        M = x.numel() // C
        dc = dtype_code(x.dtype)
        mean_rstd = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        ab = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        if training:
            sums = torch.empty(2 * C, device=x.device, dtype=torch.float64)
            lib.bn_stats(x, M, C, pre, sums, dc)
            lib.bn_finalize(sums, M, C, gamma, beta, eps, momentum, rm, rv, nbt, mean_rstd, ab)
        else:
            lib.bn_eval_ab(C, gamma, beta, eps, rm, rv, mean_rstd, ab)
        y = torch.empty_like(x)
        lib.bn_apply(x, y, M, C, ab, pre, post, dc)
        if training:
            ctx.save_for_backward(x, gamma, mean_rstd, ab)
            ctx.cfg = (pre, post)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean_rstd, ab = ctx.saved_tensors
        pre, post = ctx.cfg
        dy = _as(dy, x.dtype)
        C = x.shape[-1]
        M = x.numel() // C
        dc = dtype_code(x.dtype)
        sums = torch.empty(2 * C, device=x.device, dtype=torch.float64)
        lib.bn_bwd_reduce(x, dy, M, C, mean_rstd, ab, pre, post, sums, dc)
        dx = torch.empty_like(x)
        dg = torch.empty(C, device=x.device, dtype=torch.float32)
        db = torch.empty(C, device=x.device, dtype=torch.float32)
        lib.bn_bwd_apply(x, dy, dx, M, C, mean_rstd, ab, gamma, sums, pre, post, dg, db, dc)
        return dx, dg, db, None, None, None, None, None, None, None, None


def batchnorm(x, gamma, beta, running_mean, running_var, num_batches_tracked=None, eps=1e-5, momentum=0.1,
              pre_act=None, post_act=None, training=True):
    """y = post_act(BN(pre_act(x))) over the last (channel) dim, torch train-mode semantics incl. running stats."""
    if not training and torch.is_grad_enabled() and x.requires_grad:
        raise TcctError('eval-mode batchnorm is inference-only here')
    return _BatchNorm.apply(x, gamma, beta, running_mean, running_var, num_batches_tracked, float(eps), float(momentum),
                            ACT[pre_act], ACT[post_act], bool(training))


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _chk(x, gamma, beta)
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        mr = torch.empty(2 * M, device=x.device, dtype=torch.float32)
        lib.layernorm_fwd(x, y, M, C, gamma, beta, eps, mr, dtype_code(x.dtype))
        ctx.save_for_backward(x, gamma, mr)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mr = ctx.saved_tensors
        dy = _as(dy, x.dtype)
        C = x.shape[-1]
        M = x.numel() // C
        dx = torch.empty_like(x)
        dg = torch.empty(C, device=x.device, dtype=torch.float32)
        db = torch.empty(C, device=x.device, dtype=torch.float32)
        lib.layernorm_bwd(x, dy, dx, M, C, gamma, mr, dg, db, dtype_code(x.dtype))
        return dx, dg, db, None


def layernorm(x, gamma, beta, eps=1e-6):
    return _LayerNorm.apply(x, gamma, beta, float(eps))


# ------------------------------------------------------------------------------------------- elementwise
class _Act(torch.autograd.Function):
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


class _AddAct(torch.autograd.Function):
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


class _Residual(torch.autograd.Function):
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


class _Concat2(torch.autograd.Function):
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
