"""`stc_tt` ("TCCT"): CrossResNet(tiny) + MPViT(tiny, pooling token mixer) + FTC fusion/decoder, MI355X-native.

Drop-in for the reference's model factory (reference nets/tcct.py:1090-1096): same factory names, same attribute
names (`base_cnn`, `base_vit`, `feats`), same state_dict keys/shapes (checked against tests/golden/state_dict_keys.json)
and the same forward contract `model(x[B,3,H,W]) -> [y0,y1,y2,y4]` of `[B,n_class,H,W]` logits — but every op is a
HIP kernel on NHWC tensors (tcct_amd.ops); `nn.Conv2d/BatchNorm2d/LayerNorm/Linear` objects are used only as
parameter holders (their forward is never called).  Returned logits / feats are NCHW-shaped *views* of NHWC memory.

Reference call sites are cited per method (paths relative to /root/reference/task1).
"""
import math

import torch
from torch import nn

from .. import ops
from .._lib import TcctError

KSIZES = (13, 11, 9, 7, 5)


def _conv(m, x, out_dtype=None, stats_pre=None):
    return ops.conv2d(x, m.weight, m.bias, stride=m.stride[0], pad=tuple(m.padding), out_dtype=out_dtype, stats_pre=stats_pre)


def _dw(m, x, add_input=False, bn_stats=False, deferred=None):
    return ops.dwconv3x3(x, m.weight, m.bias, stride=m.stride[0], add_input=add_input, bn_stats=bn_stats, deferred=deferred)


def _bn(m, x, pre=None, post=None, residual=None):
    return ops.batchnorm(x, m.weight, m.bias, m.running_mean, m.running_var, m.num_batches_tracked, eps=m.eps,
                         momentum=m.momentum, pre_act=pre, post_act=post, training=m.training, residual=residual)


def _bn_args(m):
    return (m.weight, m.bias, m.running_mean, m.running_var, m.num_batches_tracked, m.eps, m.momentum)


def _conv_bn(mc, mb, x, pre=None, post=None, residual=None, x_final=False):
    """post(BN(pre(conv(x)))).  Training: convolution with the BatchNorm statistics fused into its epilogue, then the BN pass -- as ONE autograd
    node for the 1x1 convolutions (ops.pw_conv_bn: the BatchNorm backward rides on the convolution's backward kernel; x_final: see there);
    eval under no_grad: one kernel, BatchNorm and activations folded into the convolution epilogue (ops.conv_bn_act)."""
    if ops.pw_conv_bn_ok(x, mc.weight, mc.bias, mb.training, pre, post) and mc.stride[0] == 1:
        return ops.pw_conv_bn(x, mc.weight, mc.bias, _bn_args(mb), post, residual, x_final=x_final)
    if mb.training or torch.is_grad_enabled() or not ops.INFER_FUSE:
        return _bn(mb, _conv(mc, x, stats_pre=(pre or 'none') if mb.training else None), pre=pre, post=post, residual=residual)
    bn = (mb.weight, mb.bias, mb.running_mean, mb.running_var, mb.eps)
    if ops.conv_bn_residual_eval_ok(x, mc.weight, mc.bias, mc.stride[0], mc.padding, pre, post, residual):
        return ops.conv_bn_residual_eval(x, mc.weight, mc.bias, bn, residual)      # the residual add in the GEMM epilogue (round 6)
    y = ops.conv_bn_act(x, mc.weight, mc.bias, mc.stride[0], tuple(mc.padding), bn, pre, post)
    return y if residual is None else ops.add(y, residual)


def _nchw_view(y):
    return y.permute(0, 3, 1, 2)


# --------------------------------------------------------------------------------------------- CNN branch
class CrossCNNBlock(nn.Module):
    """reference nets/tcct.py:803-828: gelu(block12(x)+block34(x)) -> block5; conv->conv->LeakyReLU->BN ordering."""

    def __init__(self, in_c, out_c, ksize):
        super().__init__()
        ksize = self._ksize(ksize)
        self.block12 = nn.Sequential(nn.Conv2d(in_c, out_c, 3, padding=1), nn.Conv2d(out_c, out_c, 3, padding=1),
                                     nn.LeakyReLU(), nn.BatchNorm2d(out_c))
        self.block34 = nn.Sequential(nn.Conv2d(in_c, out_c, (1, ksize), padding=(0, ksize // 2)),
                                     nn.Conv2d(out_c, out_c, (ksize, 1), padding=(ksize // 2, 0)),
                                     nn.Conv2d(out_c, out_c, 3, padding=1), nn.LeakyReLU(), nn.BatchNorm2d(out_c))
        self.block5 = nn.Sequential(nn.Conv2d(out_c, out_c, 3, padding=1), nn.LeakyReLU(), nn.BatchNorm2d(out_c))

    @staticmethod
    def _ksize(ksize):
        return ksize

    def forward(self, x, pool=False):
        """pool=True: return (maxpool2(out), out') -- the level's output also feeds `self.pool` (CrossResNet.forward, reference tcct.py:876-884)
        and in training the last BatchNorm and the pooling share one pass (ops.batchnorm_maxpool2_fork); out' is the alias the other
        consumers read"""
        tr = self.training
        m0 = self.block12[0]       # x feeds both branches: block34 reads the alias, its gradient is added in block12[0]'s dgrad epilogue
        m1c = self.block12[1]
        if ops.conv3x3_chain_ok(x, m0.weight, m0.bias, m1c.weight, m1c.bias, m0.stride[0], m0.padding, m1c.stride[0], m1c.padding):
            # conv3x3 -> conv3x3, nothing in between (reference tcct.py:808-810): one launch each way, the intermediate is not read back
            a, x2 = ops.conv3x3_chain(x, m0.weight, m0.bias, m1c.weight, m1c.bias, stats_pre='lrelu' if tr else None, fork=True)
        elif not tr and ops.conv3x3_chain_infer_ok(x, m0.weight, m1c.weight, m0.stride[0], m0.padding, m1c.stride[0], m1c.padding):
            a, x2 = ops.conv3x3_chain_infer(x, m0.weight, m0.bias, m1c.weight, m1c.bias), x      # inference: the intermediate is never written
        else:
            a0, x2 = ops.conv2d_fork(x, m0.weight, m0.bias, m0.stride[0], tuple(m0.padding))
            a = _conv(m1c, a0, stats_pre='lrelu' if tr else None)
        b = _conv(self.block34[2], _conv(self.block34[1], _conv(self.block34[0], x2)), stats_pre='lrelu' if tr else None)
        if tr:      # fused junction: gelu(BN(lrelu(a)) + BN(lrelu(b))) in one pass over a, b (and one fused backward)
            m1, m2 = self.block12[3], self.block34[4]
            c = ops.bn2_add_act(a, (m1.weight, m1.bias, m1.running_mean, m1.running_var, m1.num_batches_tracked, m1.eps, m1.momentum),
                                b, (m2.weight, m2.bias, m2.running_mean, m2.running_var, m2.num_batches_tracked, m2.eps, m2.momentum))
        else:
            if torch.is_grad_enabled() or not ops.INFER_FUSE:
                c = ops.add_act(_bn(self.block12[3], a, pre='lrelu'), _bn(self.block34[4], b, pre='lrelu'), 'gelu')
            else:
                m1, m2 = self.block12[3], self.block34[4]
                c = ops.bn2_add_act_eval(a, (m1.weight, m1.bias, m1.running_mean, m1.running_var, m1.eps),
                                         b, (m2.weight, m2.bias, m2.running_mean, m2.running_var, m2.eps))
        if not pool:
            return _conv_bn(self.block5[0], self.block5[2], c, pre='lrelu')
        mc, mb = self.block5[0], self.block5[2]
        if mb.training:
            y = _conv(mc, c, stats_pre='lrelu')
            if ops.bn_pool_ok(y, True):
                return ops.batchnorm_maxpool2_fork(y, mb.weight, mb.bias, mb.running_mean, mb.running_var, mb.num_batches_tracked, eps=mb.eps,
                                                   momentum=mb.momentum, pre_act='lrelu')
            return ops.maxpool2_fork(_bn(mb, y, pre='lrelu'))
        return ops.maxpool2_fork(_conv_bn(mc, mb, c, pre='lrelu'))


class PlainCNNBlock(CrossCNNBlock):
    """reference nets/tcct.py:830-855: the same block with every cross convolution shrunk to 1x3 / 3x1 (pnnu)"""

    @staticmethod
    def _ksize(ksize):
        return 3


class CrossResNet(nn.Module):
    """reference nets/tcct.py:857-885: widths 32x5 (flag_tiny, stc_tt) or 32-64-96-128-256 (stc_tb / gtc_tb), ksizes 13/11/9/7/5."""
    __name__ = 'crnet'

    def __init__(self, in_ch=3, out_ch=6, flag_tiny=False, Block=CrossCNNBlock):
        super().__init__()
        layers = (32, 32, 32, 32, 32) if flag_tiny else (32, 64, 96, 128, 256)
        self.layer_dims = layers
        self.pool = nn.MaxPool2d(kernel_size=2)
        self.path_estan = nn.ModuleList([Block(layers[0], layers[0], KSIZES[0])])
        for i in range(len(layers) - 1):
            self.path_estan.append(Block(layers[i], layers[i + 1], KSIZES[i + 1]))
        self.cnn = nn.Sequential(nn.Conv2d(3, layers[0], 3, 1, 1), nn.BatchNorm2d(layers[0]))

    def forward(self, x, levels=None):
        """x: NHWC [B,H,W,4] (3 image channels + zero pad); levels: only the first `levels` resolutions (vitu needs level 0 only)."""
        xs = []
        for _ in self.iter_levels(x, xs, levels):
            pass
        return xs

    def iter_levels(self, x, xs, levels=None):
        """generator form of forward(): appends one resolution level to `xs` per step (ops.run_interleaved alternates it with the ViT stages)"""
        m = self.cnn[1]
        if ops.conv3x3_c3_bn_ok(x, self.cnn[0].weight, m.training, None):       # conv + train-mode BatchNorm, the 452 MB conv output never stored
            x = ops.conv3x3_c3_bn(x, self.cnn[0].weight, self.cnn[0].bias, _bn_args(m), 1)
        elif self.training or torch.is_grad_enabled() or not ops.INFER_FUSE:
            x = _bn(m, ops.conv3x3_c3(x, self.cnn[0].weight, self.cnn[0].bias, 1, stats_pre='none' if self.training else None))
        else:
            x = ops.conv3x3_c3(x, self.cnn[0].weight, self.cnn[0].bias, 1, infer_bn=(m.weight, m.bias, m.running_mean, m.running_var, m.eps))
        n = len(self.path_estan) if levels is None else int(levels)
        for i, enc in enumerate(self.path_estan[:n]):
            if i + 1 < n:               # the reference also pools after the last level; that result is unused
                x, skip = enc(x, pool=True)         # skip aliases the level's output: its gradient is added inside the pooling backward
                xs.append(skip)
                if i == 0:              # data-parallel runs: the backward pass crossing this edge has finished levels 1-4 (tcct_amd/dist.py)
                    x = ops.grad_mark(x, 'deep')
            else:
                x = enc(x)
                xs.append(x)
            yield i


# --------------------------------------------------------------------------------------------- ViT branch
class Conv2d_BN(nn.Module):
    """reference nets/tcct.py:55-97"""

    def __init__(self, in_ch, out_ch, kernel_size=1, stride=1, pad=0, act=False):
        super().__init__()
        self.conv = nn.Conv2d(in_ch, out_ch, kernel_size, stride, pad, bias=False)
        self.bn = nn.BatchNorm2d(out_ch)
        fan_out = kernel_size * kernel_size * out_ch
        self.conv.weight.data.normal_(0.0, math.sqrt(2.0 / fan_out))
        self.act = act

    def forward(self, x, residual=None, x_final=False):
        sp = 'none' if self.training else None
        if self.conv.in_channels == 3:          # stem[0]: 3-channel input -> im2col + pointwise MFMA
            if ops.conv3x3_c3_bn_ok(x, self.conv.weight, self.bn.training, 'hswish' if self.act else None):
                return ops.conv3x3_c3_bn(x, self.conv.weight, None, _bn_args(self.bn), self.conv.stride[0], 'hswish' if self.act else None)
            if self.training or torch.is_grad_enabled() or not ops.INFER_FUSE:
                y = ops.conv3x3_c3(x, self.conv.weight, None, self.conv.stride[0], stats_pre=sp)
                return _bn(self.bn, y, post='hswish' if self.act else None)
            m = self.bn
            return ops.conv3x3_c3(x, self.conv.weight, None, self.conv.stride[0], post_act='hswish' if self.act else None,
                                  infer_bn=(m.weight, m.bias, m.running_mean, m.running_var, m.eps))
        return _conv_bn(self.conv, self.bn, x, post='hswish' if self.act else None, residual=residual, x_final=x_final)

    def forward_deferred(self, x):
        """(y, link) with the BatchNorm + Hardswish of this layer PENDING on y for the one consumer, a depthwise convolution (round 4: stem[1] -> the first
        patch embedding), or None when that form does not apply"""
        if not (ops.BN_DEFER_DW and self.act and self.bn.training and self.conv.in_channels != 3):
            return None
        y = _conv(self.conv, x, stats_pre='none')
        m = self.bn
        if not ops.batchnorm_deferred_ok(y, True, 'hswish'):
            return _bn(m, y, post='hswish'), None
        return ops.batchnorm_deferred(y, m.weight, m.bias, m.running_mean, m.running_var, m.num_batches_tracked, m.eps, m.momentum, 'hswish')


class DWConv2d_BN(nn.Module):
    """reference nets/tcct.py:99-147: dw3x3 -> pw1x1 -> BN -> Hardswish"""

    def __init__(self, in_ch, out_ch, kernel_size=3, stride=1):
        super().__init__()
        self.dwconv = nn.Conv2d(in_ch, out_ch, kernel_size, stride, (kernel_size - 1) // 2, groups=out_ch, bias=False)
        self.pwconv = nn.Conv2d(out_ch, out_ch, 1, 1, 0, bias=False)
        self.bn = nn.BatchNorm2d(out_ch)
        for m in (self.dwconv, self.pwconv):
            n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            m.weight.data.normal_(0, math.sqrt(2.0 / n))

    def forward(self, x, fork=False, deferred=None):
        """fork: also return an alias of x for its other consumer (the stage output also feeds FTC's tran_vit convolution): that
        gradient is added inside the depthwise input-gradient kernel.  deferred: x carries a pending BatchNorm + Hardswish (ops.batchnorm_deferred)
        that the depthwise kernels apply on load"""
        if fork:
            m = self.dwconv
            d, alias = ops.dwconv3x3_fork(x, m.weight, m.bias, stride=m.stride[0])
            return _conv_bn(self.pwconv, self.bn, d, post='hswish'), alias
        return _conv_bn(self.pwconv, self.bn, _dw(self.dwconv, x, deferred=deferred), post='hswish')


class DWCPatchEmbed(nn.Module):
    """reference nets/tcct.py:149-171"""

    def __init__(self, embed_dim, stride):
        super().__init__()
        self.patch_conv = DWConv2d_BN(embed_dim, embed_dim, 3, stride)

    def forward(self, x, fork=False, deferred=None):
        return self.patch_conv(x, fork, deferred)


class Patch_Embed_stage(nn.Module):
    """reference nets/tcct.py:173-195 (num_path = 1)"""

    def __init__(self, embed_dim, isPool=False):
        super().__init__()
        self.patch_embeds = nn.ModuleList([DWCPatchEmbed(embed_dim, 2 if isPool else 1)])

    def forward(self, x, fork=False, deferred=None):
        return self.patch_embeds[0](x, fork, deferred)


class ConvPosEnc(nn.Module):
    """reference nets/tcct.py:197-217"""

    def __init__(self, dim, k=3):
        super().__init__()
        self.proj = nn.Conv2d(dim, dim, k, 1, k // 2, groups=dim)


class ConvRelPosEnc(nn.Module):
    """reference nets/tcct.py:219-287.  With the default pooling mixer these are parameters only (never executed by stc_tt; kept for
    checkpoint parity); with `att='factor'` FactorAtt_ConvRelPosEnc runs them through `ops.factor_att` (depthwise 3/5/7 windows over
    the head splits of v, times q)."""

    def __init__(self, Ch, h, window):
        super().__init__()
        if isinstance(window, int):
            window = {window: h}
        elif not isinstance(window, dict):
            raise ValueError()          # reference tcct.py:245
        self.conv_list = nn.ModuleList()
        self.head_splits = []
        for k, split in window.items():
            self.conv_list.append(nn.Conv2d(split * Ch, split * Ch, (k, k), padding=(k // 2, k // 2), groups=split * Ch))
            self.head_splits.append(split)
        self.channel_splits = [x * Ch for x in self.head_splits]


class FactorAtt_ConvRelPosEnc(nn.Module):
    """reference nets/tcct.py:289-341: the factorised-attention token mixer the reference keeps commented out in
    MHCABlock (tcct.py:436-449).  qkv / proj are the pointwise GEMMs; softmax over the tokens, the per-head K^T V and Q (K^T V)
    contractions and the convolutional relative position term are the `tcct_fatt_*` / `tcct_dwk_*` kernels.  The crpe head splits
    (2+3+3) need num_heads = 8; attn_drop / proj_drop are 0 in the reference (DROP_RATE, tcct.py:26)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, shared_crpe=None):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.crpe = shared_crpe

    def mix(self, x, size):
        """everything up to (not including) the output projection: tokens [B,N,C] -> [B,N,C]"""
        if sum(self.crpe.channel_splits) != x.shape[-1]:
            raise ValueError(f'crpe head splits {self.crpe.head_splits} do not cover {self.num_heads} heads')
        qkv = ops.conv2d(x, self.qkv.weight, self.qkv.bias)
        return ops.factor_att(qkv, size, self.num_heads, self.scale, self.crpe.conv_list)

    def forward(self, x, size):
        return ops.conv2d(self.mix(x, size), self.proj.weight, self.proj.bias)


class Mlp(nn.Module):
    """reference nets/tcct.py:29-53"""

    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class MHCABlock(nn.Module):
    """reference nets/tcct.py:417-469 with `att = MetaPool()` (tcct.py:449): cpe -> x+dp(pool(LN1 x)) -> x+dp(mlp(LN2 x))"""

    def __init__(self, dim, mlp_ratio, drop_path, shared_cpe, shared_crpe, att='pool', num_heads=8):
        super().__init__()
        self.cpe = shared_cpe
        self.crpe = shared_crpe
        if att == 'factor':     # the commented-out alternative of the reference (tcct.py:443-448), qkv_bias=True (MHCABlock default :424)
            self.att = FactorAtt_ConvRelPosEnc(dim, num_heads=num_heads, qkv_bias=True, shared_crpe=shared_crpe)
        elif att != 'pool':
            raise ValueError(f"att must be 'pool' (MetaPool, tcct.py:449) or 'factor', got {att!r}")
        self.mlp = Mlp(dim, dim * mlp_ratio)
        self.drop_prob = float(drop_path)
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)

    def forward(self, x, scales, fork=False):
        """x: NHWC image [B,H,W,C] == tokens [B,N,C]; scales: None or (s1,s2) fp32 [B] DropPath mask/keep.
        fork: also return an alias of the input for its other consumers (gradient added inside the ConvPosEnc input-gradient kernel)"""
        B, H, W, C = x.shape
        m = self.cpe.proj
        if fork:
            x, x_alias = ops.dwconv3x3_fork(x, m.weight, m.bias, stride=m.stride[0], add_input=True)
        else:
            x = _dw(m, x, add_input=True)
        t = x.view(B, H * W, C)
        s1, s2 = scales if scales is not None else (None, None)
        # (the residual paths read aliases of t: their gradients are added inside the LayerNorm backward kernels)
        att = getattr(self, 'att', None)
        cur2 = None
        if att is None and ops.ln_metapool_residual_ok(t, self.norm1.weight, self.norm1.bias):
            # t + dp(pool(LN1 t)): LayerNorm, mixer, DropPath scale and residual in ONE pass each way (the normalised tensor is never written) ...
            if ops.LN_POOL_LN2:
                # ... and LN2 of the row while it is in registers
                t, cur2, mr2 = ops.ln_metapool_residual_ln(t, self.norm1.weight, self.norm1.bias, self.norm1.eps, s1, self.norm2.weight, self.norm2.bias,
                                                           self.norm2.eps)
                m_ = self.mlp
                if ops.mlp_tail_ok(t, cur2, m_.fc1.weight, m_.fc1.bias, m_.fc2.weight, m_.fc2.bias):
                    # fc1 -> GELU -> fc2 -> + t as one node: LayerNorm2's backward rides on fc1's input-gradient kernel
                    t = ops.mlp_tail(t, cur2, mr2, self.norm2.weight, self.norm2.bias, m_.fc1.weight, m_.fc1.bias, m_.fc2.weight, m_.fc2.bias, s2)
                    return (t.view(B, H, W, C), x_alias) if fork else t.view(B, H, W, C)
            else:
                t = ops.ln_metapool_residual(t, self.norm1.weight, self.norm1.bias, self.norm1.eps, s1)
        else:
            cur, t = ops.layernorm_fork(t, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            if att is not None:     # t + dp(proj(factor_att(LN1 t))): the projection GEMM carries the DropPath scale and the residual
                t = ops.linear_residual(att.mix(cur, (H, W)), att.proj.weight, att.proj.bias, t, s1)
            else:
                t = ops.metapool_residual(cur, t, s1)       # t + dp(pool(LN1 t)): mixer, DropPath scale and residual in one pass
        if cur2 is None:
            cur, t = ops.layernorm_fork(t, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        else:
            cur = cur2
        if self.training or torch.is_grad_enabled() or not ops.INFER_FUSE:
            y1 = ops.conv2d(cur, self.mlp.fc1.weight, self.mlp.fc1.bias)
            if ops.gelu_linear_residual_ok(y1, self.mlp.fc2.weight, self.mlp.fc2.bias, t):
                # GELU applied while fc2's kernels stage their tiles: h and dh never exist in HBM (stages 0 and 1)
                t = ops.gelu_linear_residual(y1, self.mlp.fc2.weight, self.mlp.fc2.bias, t, s2)
                return (t.view(B, H, W, C), x_alias) if fork else t.view(B, H, W, C)
            h = ops.act(y1, 'gelu')
        else:               # inference: GELU in the GEMM epilogue
            h = ops.conv_bn_act(cur, self.mlp.fc1.weight, self.mlp.fc1.bias, post_act='gelu')
        t = ops.linear_residual(h, self.mlp.fc2.weight, self.mlp.fc2.bias, t, s2)      # t + dp(fc2(h)) in the GEMM epilogue
        return (t.view(B, H, W, C), x_alias) if fork else t.view(B, H, W, C)


class MHCAEncoder(nn.Module):
    """reference nets/tcct.py:471-516 (num_layers = 1)"""

    def __init__(self, dim, num_heads, mlp_ratio, drop_path, att='pool'):
        super().__init__()
        self.cpe = ConvPosEnc(dim, k=3)
        self.crpe = ConvRelPosEnc(Ch=dim // num_heads, h=num_heads, window={3: 2, 5: 3, 7: 3})
        self.MHCA_layers = nn.ModuleList([MHCABlock(dim, mlp_ratio, drop_path, self.cpe, self.crpe, att=att, num_heads=num_heads)])

    def forward(self, x, scales, fork=False):
        return self.MHCA_layers[0](x, scales, fork)


class ResBlock(nn.Module):
    """reference nets/tcct.py:518-572 (InvRes)"""

    def __init__(self, dim):
        super().__init__()
        self.conv1 = Conv2d_BN(dim, dim, act=True)
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=False, groups=dim)
        self.norm = nn.BatchNorm2d(dim)
        self.conv2 = Conv2d_BN(dim, dim)
        for m in (self.conv1.conv, self.dwconv, self.conv2.conv):
            fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
            m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))

    def forward(self, x):
        return self.tail(self.conv1(x), x)

    def tail(self, f, x, flink=None):
        """flink: f is conv1's output with its BatchNorm + Hardswish pending (MHCA_stage.forward)"""
        y = _dw(self.dwconv, f, bn_stats=self.norm.training, deferred=flink)                        # statistics out of the convolution's launch
        c2, m = self.conv2, self.norm
        if (ops.batchnorm_deferred_ok(y, m.training, 'hswish') and c2.bn.training and not c2.act
                and ops.pw_conv_bn_ok(y, c2.conv.weight, None, True, None, None)):
            # norm's normalisation + Hardswish is applied by conv2's kernels while they stage their tiles: the tensor between them never exists
            ya, link = ops.batchnorm_deferred(y, m.weight, m.bias, m.running_mean, m.running_var, m.num_batches_tracked, m.eps, m.momentum, 'hswish')
            return ops.pw_conv_bn(ya, c2.conv.weight, None, _bn_args(c2.bn), None, residual=x, deferred=link)
        if not m.training and not c2.bn.training and not c2.act and ops.invres_tail_eval_ok(y, c2.conv.weight, c2.conv.bias, x):
            # inference (round 6): `norm` + Hardswish on conv2's load path, conv2.bn and the residual in its epilogue: the two passes between dwconv and the stage's sum disappear
            mb = c2.bn
            return ops.invres_tail_eval(y, (m.weight, m.bias, m.running_mean, m.running_var, m.eps), c2.conv.weight, c2.conv.bias,
                                        (mb.weight, mb.bias, mb.running_mean, mb.running_var, mb.eps), x)
        f = _bn(self.norm, y, post='hswish')
        # x + BN(conv2(f)): the add rides on the normalisation pass; conv2 is the only consumer of f, so the backward reduction of `norm`
        # rides on conv2's input-gradient epilogue (x_final)
        return self.conv2(f, residual=x, x_final=True)


class MHCA_stage(nn.Module):
    """reference nets/tcct.py:574-616 (num_path = 1): cat[InvRes(x), Encoder(x)] -> 1x1 -> BN -> Hardswish"""

    def __init__(self, embed_dim, out_embed_dim, num_heads, mlp_ratio, drop_path, att='pool'):
        super().__init__()
        self.mhca_blks = nn.ModuleList([MHCAEncoder(embed_dim, num_heads, mlp_ratio, drop_path, att=att)])
        self.InvRes = ResBlock(embed_dim)
        self.aggregate = Conv2d_BN(embed_dim * 2, out_embed_dim, act=True)

    def forward(self, x, scales):
        if torch.is_grad_enabled() and x.requires_grad:
            # x has three consumers (InvRes.conv1, ConvPosEnc, the InvRes residual).  They read a chain of aliases so that each
            # input-gradient kernel adds the gradient of the consumers behind it: no separate accumulation passes over x
            c1 = self.InvRes.conv1
            flink = None
            if ops.pw_conv_bn_ok(x, c1.conv.weight, None, c1.bn.training, None, 'hswish'):
                # conv1's input gradient + the alias' gradient is the complete gradient of x (= the patch embedding's BatchNorm output)
                if ops.BN_DEFER_DW and x.shape[-1] % 4 == 0 and x.shape[-1] <= 256:
                    # ... and conv1's normalisation + Hardswish is applied by InvRes.dwconv's kernels as rows enter their window: f is never written
                    (f, x1), flink = ops.pw_conv_bn(x, c1.conv.weight, None, _bn_args(c1.bn), 'hswish', fork=True, x_final=True, defer_apply=True)
                else:
                    f, x1 = ops.pw_conv_bn(x, c1.conv.weight, None, _bn_args(c1.bn), 'hswish', fork=True, x_final=True)
            else:
                f, x1 = ops.conv2d_fork(x, c1.conv.weight, None, 1, 0, stats_pre='none' if c1.bn.training else None)
                f = _bn(c1.bn, f, post='hswish')
            if ops.PARALLEL_BRANCHES and x.shape[0] * x.shape[1] * x.shape[2] <= ops.STAGE_FORK_MAX_PIXELS and torch.cuda.is_available():
                # small maps (stages 2-3 at the bench shape): every launch of the stage is latency-bound and the two halves -- InvRes: dw -> norm -> conv2 (+ x);
                # encoder: cpe -> LayerNorm / mixer -> Mlp -- only share x: the encoder goes to a stream of its own (autograd replays its backward there).  The
                # aliases keep their order: x2 is a view of x1, no kernel of the encoder stands between conv1 and the InvRes tail.
                x2h = []

                def enc():
                    e_, x2_ = self.mhca_blks[0](x1, scales, fork=True)
                    x2h.append(x2_)             # the residual alias: a view made by the encoder's first node, no kernel behind it
                    return e_
                # run_parallel issues the encoder first (its stream), then the InvRes tail here, then joins
                r, e = ops.run_parallel('vit_enc', lambda: self.InvRes.tail(f, x2h[0], flink), enc)
            else:
                e, x2 = self.mhca_blks[0](x1, scales, fork=True)
                r = self.InvRes.tail(f, x2, flink)
        elif (not torch.is_grad_enabled() and ops.PARALLEL_BRANCHES and x.is_cuda and x.shape[0] * x.shape[1] * x.shape[2] <= ops.STAGE_FORK_MAX_PIXELS
              and x.shape[0] * x.shape[1] * x.shape[2] >= ops.FUSION_FORK_MIN_PIXELS and not torch.cuda.is_current_stream_capturing()):
            # inference: the same two halves on two streams (no autograd, nothing to alias)
            r, e = ops.run_parallel('vit_enc', lambda: self.InvRes(x), lambda: self.mhca_blks[0](x, scales))
        else:
            r = self.InvRes(x)
            e = self.mhca_blks[0](x, scales)
        ag = self.aggregate
        if ag.bn.training or torch.is_grad_enabled() or not ops.INFER_FUSE:
            # cat([r, e]) -> 1x1 -> BN -> Hardswish with the concatenation folded into the GEMM operands (no concat / split passes)
            if ops.pw_conv_bn_ok(r, ag.conv.weight, None, ag.bn.training, None, 'hswish', x2=e):
                # r = x + BN(conv2(f)) is consumed here only: conv2.bn's backward reduction rides on this kernel's dx epilogue
                return ops.pw_conv_bn(r, ag.conv.weight, None, _bn_args(ag.bn), 'hswish', x2=e, x_final=True)
            y = ops.conv1x1_cat2(r, e, ag.conv.weight, stats_pre='none' if ag.bn.training else None)
            return _bn(ag.bn, y, post='hswish')
        if ops.conv1x1_cat2_bn_act_eval_ok(r, e, ag.conv.weight, 'hswish'):       # stage 0: no concatenation pass, BatchNorm + Hardswish in the GEMM epilogue (round 6)
            m = ag.bn
            return ops.conv1x1_cat2_bn_act_eval(r, e, ag.conv.weight, (m.weight, m.bias, m.running_mean, m.running_var, m.eps), 'hswish')
        return ag(ops.concat2(r, e))


class Cls_head(nn.Module):
    """reference nets/tcct.py:618-633 — parameters only (MPViT.forward is not on the path)."""

    def __init__(self, embed_dim, num_classes):
        super().__init__()
        self.cls = nn.Linear(embed_dim, num_classes)


class MPViT(nn.Module):
    """reference nets/tcct.py:649-753, configuration of mpvit_tiny (tcct.py:766-776)."""
    __name__ = 'mpvit'

    def __init__(self, embed_dims=(64, 96, 128, 160), mlp_ratios=(1, 1, 1, 1), num_heads=(4, 4, 4, 4),
                 drop_path_rate=0.1, num_classes=1000, att='pool'):
        super().__init__()
        if att == 'factor':     # the crpe windows {3:2, 5:3, 7:3} split 8 heads (tcct.py:484-488); mpvit_tiny's 4 only fit the pooling mixer
            num_heads = (8, 8, 8, 8)
        self.att = att
        self.num_stages = 4
        self.embed_dims = list(embed_dims)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, 4)]
        self.drop_probs = dpr
        self.stem = nn.Sequential(Conv2d_BN(3, embed_dims[0] // 2, 3, 2, 1, act=True),
                                  Conv2d_BN(embed_dims[0] // 2, embed_dims[0], 3, 1, 1, act=True))
        self.patch_embed_stages = nn.ModuleList([Patch_Embed_stage(embed_dims[i], isPool=i > 0) for i in range(4)])
        self.mhca_stages = nn.ModuleList([
            MHCA_stage(embed_dims[i], embed_dims[i + 1] if i + 1 < 4 else embed_dims[i], num_heads[i], mlp_ratios[i],
                       dpr[i], att=att) for i in range(4)])
        self.cls_head = Cls_head(embed_dims[-1], num_classes)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        self.forced_dp_masks = None     # list of 6 [B] 0/1 tensors (draw order of the reference) for parity tests

    def _dp_scales(self, B, device):
        """DropPath (timm, reference tcct.py:452,465,468): per-sample Bernoulli(keep)/keep, train mode only."""
        out = []
        forced = list(self.forced_dp_masks) if self.forced_dp_masks is not None else None
        if forced is None and self.training and any(p > 0.0 for p in self.drop_probs):
            # all masks of the step from ONE draw (the draws used to be a dozen 2-us launches in front of the first kernel of the step); the RNG stream is not
            # pinned to the reference's (timm's DropPath version is not, either): parity tests force the masks
            act = [s for s in range(4) if self.drop_probs[s] > 0.0]
            key = (device, B, tuple(self.drop_probs))
            if getattr(self, '_dp_keep', (None,))[0] != key:
                keep = torch.tensor([1.0 - self.drop_probs[s] for s in act for _ in range(2)], device=device, dtype=torch.float32).view(-1, 1)
                self._dp_keep = (key, keep)
            keep = self._dp_keep[1]
            m = (torch.rand((2 * len(act), B), device=device) < keep).to(torch.float32) / keep
            it = iter(range(len(act)))
            return [((m[2 * i], m[2 * i + 1]) if self.drop_probs[s] > 0.0 else None) for s in range(4) for i in ([next(it)] if self.drop_probs[s] > 0.0 else [None])]
        for s in range(4):
            p = self.drop_probs[s]
            if p == 0.0 or not self.training:
                out.append(None)
                continue
            pair = []
            for _ in range(2):
                if forced is not None:
                    m = forced.pop(0).to(device=device, dtype=torch.float32)
                else:
                    m = torch.empty(B, device=device, dtype=torch.float32).bernoulli_(1 - p)
                pair.append((m / (1 - p)).contiguous())
            out.append(tuple(pair))
        return out

    def forward_features(self, x):
        """x NHWC [B,H,W,4] -> [x2,x3,x4,x5] NHWC (reference tcct.py:733-745)."""
        xs = []
        for _ in self.iter_stages(x, xs):
            pass
        return xs

    def iter_stages(self, x, xs):
        """generator form of forward_features(): appends one stage output to `xs` per step"""
        scales = self._dp_scales(x.shape[0], x.device)
        x = self.stem[0](x)
        d0 = self.stem[1].forward_deferred(x)       # stem[1]'s normalisation rides on the first patch embedding's depthwise kernels
        x, link0 = d0 if d0 is not None else (self.stem[1](x), None)
        for i in range(4):
            if i == 1:                  # data-parallel runs: crossing this edge backwards means stages 1-3 are done (tcct_amd/dist.py)
                x = ops.grad_mark(x, 'deep')
            if i > 0 and torch.is_grad_enabled() and x.requires_grad:
                p, xs[-1] = self.patch_embed_stages[i](x, fork=True)    # the returned level is the alias (read by FTC.tran_vit)
            else:
                p = self.patch_embed_stages[i](x, deferred=link0 if i == 0 else None)
            x = self.mhca_stages[i](p, scales[i])
            xs.append(x)
            yield i


def mpvit_tiny(**kw):
    return MPViT(**kw)


# ------------------------------------------------------------------------------------- fusion + decoder
class MPUpBlock(nn.Module):
    """reference nets/tcct.py:887-914: conv3x3-BN-LeakyReLU -> x2 bilinear (align_corners=True) -> +skip -> 1x1"""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.prep = nn.Sequential(nn.Conv2d(in_ch, out_ch, 3, 1, 1), nn.BatchNorm2d(out_ch), nn.LeakyReLU(inplace=True))
        self.post = nn.Sequential(nn.Conv2d(out_ch, out_ch, 1, 1, 0))

    def forward(self, x1, x2, with_sum=False, want_plain=True):
        """with_sum: also return x2 + output (the `x_i + y_i` of FTC.forward, tcct.py:1028-1031) from the same GEMM epilogue;
        want_plain=False: the block's own output is not needed by the caller (returned as None, never written)"""
        y = _conv_bn(self.prep[0], self.prep[1], x1, post='lrelu')
        if with_sum:
            return ops.up_skip_conv(y, x2, self.post[0].weight, self.post[0].bias, True, want_plain=want_plain)
        u = ops.bilinear(y, (x1.shape[1] * 2, x1.shape[2] * 2), True, residual=x2)
        return _conv(self.post[0], u)

    def forward_through(self, x1, x2, t32, aux=None):
        """t32(x2 + forward(x1, x2)) for the LAST decoder block, whose own output nobody else reads (FTC.forward, reference tcct.py:1031-1040):
        resize, `post`, the skip add and `t32` as ONE GEMM with composed weights (ops.up_skip_conv_t32), or None when that form does not apply.
        aux: the level-0 head when the caller needs only ITS output of g0 -> returns the tuple (fp32 logits NHWC, resized y) instead"""
        p, t = self.post[0], t32
        probe = torch.empty((x1.shape[0], x1.shape[1], x1.shape[2], self.prep[0].out_channels), dtype=x1.dtype, device='meta')    # shape / dtype only: no device memory
        # (train mode -- or inference with the fused epilogues: the composition needs no statistics, round 6)
        if not ((self.prep[1].training or (ops.INFER_FUSE and not torch.is_grad_enabled())) and ops.up_skip_conv_t32_ok(probe, x2, p.weight, p.bias, t.weight, t.bias)):
            return None
        y = _conv_bn(self.prep[0], self.prep[1], x1, post='lrelu')
        if aux is not None and ops.up_skip_conv_t32_aux_ok(probe, x2, p.weight, p.bias, t.weight, t.bias, aux.weight, aux.bias):
            # nothing but the aux head reads g0 in this step: (fp32 logits, what `feats` would rebuild g0 from) -- g0 is never written
            if ops.TAIL_AUX_LOW:    # ... and the resize is taken of the n_class-channel product at the low resolution: up(y) is never written either
                return ops.up_skip_conv_t32_aux_low(y, x2, p.weight, p.bias, t.weight, t.bias, aux.weight, aux.bias, True), ('y', y.detach())
            lg, v = ops.up_skip_conv_t32_aux(y, x2, p.weight, p.bias, t.weight, t.bias, aux.weight, aux.bias, True)
            return lg, ('v', v)
        return ops.up_skip_conv_t32(y, x2, p.weight, p.bias, t.weight, t.bias, True)


class FTC(nn.Module):
    """reference nets/tcct.py:944-1046 with SimpleFusion (flag_gate=False)."""
    __name__ = 'gtc'

    def __init__(self, base_cnn, base_vit, out_channels=5, filters=32, compute_dtype=torch.float32, legacy_heads=False,
                 flag_gate=False, flag_cnn=True, flag_vit=True):
        """legacy_heads: the older layout of the reference's shipped GOALS/HCMS/HEG checkpoints (task1/onnx/tcct_goals.py:949-1036):
        no t32x convolutions, the aux heads read the decoder outputs directly"""
        super().__init__()
        self.legacy_heads = bool(legacy_heads)
        # sibling variants of the reference (nets/tcct.py:1090-1136): gtc_* fuse with GateFusion (:916-932), cnnu / vitu drop one
        # encoder from the fusion (:1016-1019).  All modules (and state_dict keys) exist in every variant, as in the reference.
        self.flag_gate, self.flag_cnn, self.flag_vit = bool(flag_gate), bool(flag_cnn), bool(flag_vit)
        self.defer_aux_resize = False       # see forward(): aux heads returned as ops.LowResLogits (set by KiteSeg.calc_loss)
        self.forced_gate_fields = None      # parity tests: list of 4 NHWC alpha fields in draw order (reference: torch.rand on the CPU)
        if not (self.flag_cnn or self.flag_vit):
            raise TcctError('FTC needs at least one of flag_cnn / flag_vit')
        self.base_vit = base_vit
        self.base_cnn = base_cnn
        ed, ld = base_vit.embed_dims, base_cnn.layer_dims
        for j, cin in enumerate((ed[1], ed[2], ed[3], ed[3])):
            setattr(self, f'tran_vit{j}', nn.Sequential(nn.Conv2d(cin, ld[j + 1], 1, 1, 0), nn.BatchNorm2d(ld[j + 1])))
        for j in range(4):
            setattr(self, f'tran_cnn{j}', nn.Sequential(nn.Conv2d(ld[j + 1], ld[j + 1], 1, 1, 0), nn.BatchNorm2d(ld[j + 1])))
        self.head = nn.Sequential(nn.Conv2d(ld[-1], ld[-1], 3, 1, 1), nn.BatchNorm2d(ld[-1]), nn.LeakyReLU())
        self.fuse = nn.Conv2d(ld[4], filters, kernel_size=1)          # never executed (reference tcct.py:982)
        self.dec1 = MPUpBlock(ld[-1], ld[-2])
        self.dec2 = MPUpBlock(ld[-2], ld[-3])
        self.dec3 = MPUpBlock(ld[-3], ld[-4])
        self.dec4 = MPUpBlock(ld[-4], filters)
        if not self.legacy_heads:
            self.t321 = nn.Conv2d(ld[-2], filters, 1)
            self.t322 = nn.Conv2d(ld[-3], filters, 1)
            self.t323 = nn.Conv2d(ld[-4], filters, 1)
            self.t324 = nn.Conv2d(filters, filters, 1)
        self.aux0 = nn.Conv2d(filters, out_channels, 1)
        self.aux1 = nn.Conv2d(filters, out_channels, 1)
        self.aux2 = nn.Conv2d(filters, out_channels, 1)
        self.aux4 = nn.Conv2d(filters, out_channels, 1)
        self.compute_dtype = compute_dtype
        self._feats_src = None
        self._feats = None
        self.eager_feats = False        # True: norm_add is evaluated inside forward() (set by the training loop when the udh loss is on)
        # True: the owner PROMISES that `feats` will not be differentiated in this configuration (KiteSeg sets it when --udh is off): the aux heads
        # are then composed through the t32x convolutions, g0..g2 are never written and `feats` rebuilds them on demand WITHOUT gradient.
        # Default False: `feats` carries gradient as the reference's does (RegNet(stc_tt()) + regular_udh used directly, without KiteSeg).
        self.compose_heads = False
        # True (set by KiteSeg.predict around its forward, inference only): the caller reads head 0 alone (reference kite/loop_seg.py:27-29 `pred = pred[0]`) -- the
        # three deep-supervision heads and their resizes are not evaluated, and the level-0 tail is composed through aux0 as in a --udh=false training step
        self.main_head_only = False

    @property
    def feats(self):
        """[norm_add([y0, y1, y2])] (reference tcct.py:1035).  Inside a pooled training step the feature-polarization loss sends its gradient towards `feats` as a
        recipe, not as a tensor (ops.grad_is_watched documents who still sees the dense d loss / d feats: retain_grad() / hooks set BEFORE the loss is formed)."""
        if self._feats is None and self._feats_src is not None:
            if self.legacy_heads:
                raise TcctError('the legacy-head layout (onnx/tcct_goals.py) is supported for inference and Dice/boundary training; its '
                                'six-tensor `feats` (tcct_goals.py:1021) is not built')
            g0, g1, g2, size = self._feats_src
            # a step with `compose_heads` composed a head through its t32x convolution and never wrote g_i: rebuild it (no gradient -- the owner's promise)
            g0, g1, g2 = [g() if callable(g) else g for g in (g0, g1, g2)]
            self._feats = [_nchw_view(ops.norm_add3(g0, g1, g2))]
            self._feats_src = None
        return self._feats

    @feats.setter
    def feats(self, v):
        self._feats = v
        self._feats_src = None

    def set_compute_dtype(self, dt):
        if dt not in (torch.float32, torch.bfloat16):
            raise TcctError('compute dtype must be float32 or bfloat16')
        self.compute_dtype = dt
        return self

    def _to_nhwc4(self, x):
        if x.dim() != 4 or x.shape[1] not in (1, 3):
            raise TcctError(f'expected image batch [B,3,H,W] (or [B,1,H,W]), got {tuple(x.shape)}')
        if not x.is_cuda:
            raise TcctError('tcct_amd model needs a CUDA(HIP) input; there is no CPU fallback')
        B, Cs, H, W = x.shape
        if H % 16 or W % 16:
            raise TcctError(f'H and W must be multiples of 16 (got {H}x{W}); pad first (see tcct_amd.data.synth)')
        x = x.contiguous().float()
        out = torch.empty((B, H, W, 4), device=x.device, dtype=self.compute_dtype)
        ops.lib.image_to_nhwc4(x, out, B, Cs, H, W, W, ops.dtype_code(self.compute_dtype))
        return out

    def forward(self, x):
        size = (x.shape[2], x.shape[3])
        x = self._to_nhwc4(x)
        mho = self.main_head_only and not torch.is_grad_enabled() and not self.training     # inference, head 0 only (KiteSeg.predict)
        if self.flag_vit and self.flag_cnn:
            cs, vs = [], []
            ops.run_interleaved('vit', self.base_cnn.iter_levels(x, cs), self.base_vit.iter_stages(x, vs), vs)
            # data-parallel runs: once the backward pass has crossed these nine edges, every fusion / decoder / head gradient is final
            (c1, c2, c3, c4, c5), (v2, v3, v4, v5) = [ops.grad_mark(t, 'dec') for t in cs], [ops.grad_mark(t, 'dec') for t in vs]
            def fuse_level(j, v, c):
                tv, tc = getattr(self, f'tran_vit{j}'), getattr(self, f'tran_cnn{j}')
                if self.flag_gate and self.training:
                    # tcct.py:922-929: alpha = clamp(bicubic(rand(B,C,max(3,H/32),max(3,W/32))), 0, 1); the draw is an input here
                    a1, a2 = _conv_bn(tv[0], tv[1], v), _conv_bn(tc[0], tc[1], c)
                    B_, Hh, Ww, Cc = a1.shape
                    fields = self.forced_gate_fields
                    field = fields.pop(0) if fields else torch.rand((B_, max(3, Hh // 32), max(3, Ww // 32), Cc), device=a1.device)
                    return ops.gate_fusion(a1, a2, field.to(device=a1.device, dtype=torch.float32).contiguous())
                if self.flag_gate:
                    sm = _conv_bn(tc[0], tc[1], c, residual=_conv_bn(tv[0], tv[1], v))      # x1*0.5 + x2*0.5 == (x1+x2)*0.5 exactly
                    half = torch.empty_like(sm)
                    ops.lib.scale(sm, half, sm.numel(), 0.5, ops.dtype_code(sm.dtype))
                    return half
                if (ops.TRAN_FUSE and c.shape[-1] % 8 == 0 and ops.pw_conv_bn_ok(v, tv[0].weight, tv[0].bias, tv[1].training, None, None)
                        and ops.pw_conv_bn_ok(c, tc[0].weight, tc[0].bias, tc[1].training, None, None)):
                    # both BatchNorms applied by ONE pass that also adds them: BN(tran_vit(v)) is never written and read back
                    yv, lv = ops.pw_conv_bn(v, tv[0].weight, tv[0].bias, _bn_args(tv[1]), None, defer_apply=True)
                    yc, lc = ops.pw_conv_bn(c, tc[0].weight, tc[0].bias, _bn_args(tc[1]), None, defer_apply=True)
                    return ops.affine2_add(yv, lv, yc, lc)
                return _conv_bn(tc[0], tc[1], c, residual=_conv_bn(tv[0], tv[1], v))
            pairs = ((v2, c2), (v3, c3), (v4, c4), (v5, c5))
            y8 = None
            if ops.fusion_fork_ok(c5) and not self.flag_gate and self.training:
                # round 6: the fusion of the two LARGE levels (bandwidth-bound) on a stream of its own beside the small ones + `head` (latency-bound launches).  The decoder
                # needs f[2] / f[1] only at dec2 / dec3; in the backward pass (autograd replays a node on the stream of its forward) the large levels' fusion backward
                # then runs beside the first encoder levels' backward instead of in front of it
                def high():
                    f3_, f4_ = fuse_level(2, *pairs[2]), fuse_level(3, *pairs[3])
                    return f3_, f4_, _conv_bn(self.head[0], self.head[1], f4_, post='lrelu')
                (f3, f4, y8), (f1, f2) = ops.run_parallel(ops.FUSE_STREAM_TAG, high, lambda: (fuse_level(0, *pairs[0]), fuse_level(1, *pairs[1])))
                f = [c1, f1, f2, f3, f4]
            else:
                f = [c1] + [fuse_level(j, v, c) for j, (v, c) in enumerate(pairs)]
        elif self.flag_cnn:
            # the reference also runs the ViT encoder and discards it (only its BatchNorm running statistics move); skipped here
            f = list(self.base_cnn(x))
        else:
            # only level 0 of the CNN encoder is consumed (c1 is the decoder's last skip); the deeper CNN levels are skipped
            cs, vs = [], []
            ops.run_interleaved('vit', self.base_cnn.iter_levels(x, cs, 1), self.base_vit.iter_stages(x, vs), vs)
            (c1,), (v2, v3, v4, v5) = cs, vs
            f = [c1] + [_conv_bn(getattr(self, f'tran_vit{j}')[0], getattr(self, f'tran_vit{j}')[1], v) for j, v in enumerate((v2, v3, v4, v5))]
        if not (self.flag_vit and self.flag_cnn) or y8 is None:
            y8 = _conv_bn(self.head[0], self.head[1], f[4], post='lrelu')
        y0_direct = None            # level-0 logits when the decoder tail was composed through aux0 (below)
        if self.legacy_heads:       # tcct_goals.py:1027-1033: heads on the decoder outputs
            d3 = self.dec1(y8, f[3])
            d2 = self.dec2(d3, f[2])
            d1 = self.dec3(d2, f[1])
            d0 = self.dec4(d1, f[0])
            g0, g1, g2, g3 = d0, d1, d2, d3
            lg_direct = [None, None, None]
        else:                       # x_i + y_i comes out of the decoder's last GEMM epilogue together with y_i
            d3, s3 = self.dec1(y8, f[3], with_sum=True)
            d2, s2 = self.dec2(d3, f[2], with_sum=True)
            d1, s1 = self.dec3(d2, f[1], with_sum=True)
            # level 0: post, `x_0 + y_0` and t324 as one GEMM (u, d0, s0 never written) -- and through aux0 as well when the feature-polarization
            # loss is off (nothing else reads g0 then; `feats` rebuilds it on demand)
            compose = (self.compose_heads and not self.eager_feats) or mho
            g0 = self.dec4.forward_through(d1, f[0], self.t324, aux=self.aux0 if compose else None)
            if isinstance(g0, tuple):
                y0_direct, (kind, src) = g0
                skip0, pw, tw = f[0], self.dec4.post[0], self.t324
                rebuild = ops.up_skip_conv_t32_from_v if kind == 'v' else ops.up_skip_conv_t32_from_y
                g0 = lambda: rebuild(src, skip0, pw.weight, pw.bias, tw.weight, tw.bias)      # noqa: E731
            if g0 is None:
                d0, s0 = self.dec4(d1, f[0], with_sum=True, want_plain=False)      # only x_0 + y_0 is read below: d0 is never written
                g0 = _conv(self.t324, s0)
            # levels 1-3: aux_i(t32x(s_i)) as one GEMM with the composed weight when nothing else reads g_i (feature-polarization loss off)
            mids, lg_direct = [], []
            for t, aux, s_ in ((self.t323, self.aux1, s1), (self.t322, self.aux2, s2), (self.t321, self.aux4, s3)):
                if mho:                 # nobody reads g1..g3 or their heads in this forward (`feats` rebuilds what it needs on demand)
                    lg_direct.append(None)
                    mids.append(lambda t=t, s_=s_: self._rebuild(t, s_))
                elif self.training and compose and ops.head_through_t32_ok(s_, t.weight, t.bias, aux.weight, aux.bias):
                    lg_direct.append(ops.head_through_t32(s_, t.weight, t.bias, aux.weight, aux.bias))
                    mids.append(lambda t=t, s_=s_: self._rebuild(t, s_))
                else:
                    lg_direct.append(None)
                    mids.append(_conv(t, s_))
            g1, g2, g3 = mids
        # norm_add([y0,y1,y2]) (reference tcct.py:937-942,1035) -> `self.feats`: evaluated lazily on first access (only the
        # feature-polarization loss reads it; with --udh=false the six level-0 passes are simply never launched)
        self._feats_src = (g0, g1, g2, size)
        self._feats = None
        if self.eager_feats and not self.legacy_heads and not mho:
            # the loss WILL read `feats` (KiteSeg sets the flag with --udh): evaluate norm_add here and let the aux heads read aliases of g0..g2
            # returned by its node, so that the heads' gradients are added inside norm_add's backward kernels (three accumulation passes fewer)
            forked = ops.norm_add3_fork(g0, g1, g2)
            if forked is not None:
                feats, g0, g1, g2 = forked
                self._feats, self._feats_src = [_nchw_view(feats)], None
        # aux heads: logits are produced and resized in fp32 in every mode (loss-side precision)
        f32 = torch.float32
        y0 = y0_direct if y0_direct is not None else _conv(self.aux0, g0() if callable(g0) else g0, out_dtype=f32)
        if mho:
            return [_nchw_view(y0)]
        low = [lg if lg is not None else _conv(m, g, out_dtype=f32) for m, g, lg in zip((self.aux1, self.aux2, self.aux4), (g1, g2, g3), lg_direct)]
        if self.defer_aux_resize and torch.is_grad_enabled():
            # training loop (KiteSeg.calc_loss): the three aux heads stay at their own resolution; the Dice criterion resizes on the fly
            return [_nchw_view(y0)] + [ops.LowResLogits(l_, size) for l_ in low]
        return [_nchw_view(y0)] + [_nchw_view(ops.bilinear(l_, size, False)) for l_ in low]

    @staticmethod
    def _rebuild(t, s_):
        with torch.no_grad():
            return _conv(t, s_)


def stc_tt(n_class=8, **args):
    """reference nets/tcct.py:1090-1095"""
    model = FTC(base_vit=mpvit_tiny(att=args.get('att', 'pool')), base_cnn=CrossResNet(flag_tiny=True), out_channels=n_class,
                compute_dtype=args.get('compute_dtype', torch.float32), legacy_heads=args.get('legacy_heads', False))
    model.__name__ = 'stctt'
    return model


tcct = stc_tt


def _variant(name, n_class, args, flag_tiny=True, Block=CrossCNNBlock, **flags):
    model = FTC(base_vit=mpvit_tiny(), base_cnn=CrossResNet(flag_tiny=flag_tiny, Block=Block), out_channels=n_class,
                compute_dtype=args.get('compute_dtype', torch.float32), **flags)
    model.__name__ = name
    return model


def stc_tb(n_class=8, **args):
    """reference nets/tcct.py:1097-1102: stc_tt with the wide CNN encoder (32-64-96-128-256)"""
    return _variant('stctb', n_class, args, flag_tiny=False)


def gtc_tb(n_class=8, **args):
    """reference nets/tcct.py:1056-1061: GateFusion + wide CNN encoder"""
    return _variant('gtctb', n_class, args, flag_tiny=False, flag_gate=True)


def pnnu(n_class=8, **args):
    """reference nets/tcct.py:1117-1122: cnnu with PlainCNNBlock (3-tap cross convolutions)"""
    return _variant('pnnu', n_class, args, Block=PlainCNNBlock, flag_vit=False, flag_cnn=True)


def gtc_tt(n_class=8, **args):
    """reference nets/tcct.py:1048-1053 (GateFusion; eval only here)"""
    return _variant('gtctt', n_class, args, flag_gate=True)


def cnnu(n_class=8, **args):
    """reference nets/tcct.py:1120-1126: CNN encoder + decoder, no ViT fusion"""
    return _variant('cnnu', n_class, args, flag_vit=False, flag_cnn=True)


def vitu(n_class=8, **args):
    """reference nets/tcct.py:1128-1134: ViT encoder features (+ the level-0 CNN skip) + decoder"""
    return _variant('vitu', n_class, args, flag_vit=True, flag_cnn=False)
