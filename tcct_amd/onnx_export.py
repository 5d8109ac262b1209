"""ONNX export of the eval-mode network — counterpart of the reference's `task1/onnx/onnx_save.py:4-15` (`to_onnx`: `torch.onnx.export(net,
dummy[1,3,160,160], path, input_names=['input'], output_names=['output'], dynamic_axes batch/height/width, opset_version=11)`), consumed by
`task1/onnx/onnx_infer.py:13-30` (`session.run(None, {'input': img})[0]`).

The network here is HIP kernels, not ATen ops, so there is nothing for `torch.onnx` to trace (and this image has no `onnx` package):
the graph is EMITTED from the state_dict instead — the eval forward of `RegNet(stc_tt(n))` (reference nets/tcct.py:999-1046 and the modules it
calls) written as standard opset-11 operators (Conv, BatchNormalization, LeakyRelu, HardSigmoid*Mul = Hardswish as torch exports it at opset 11,
Erf-GELU, MaxPool, AveragePool(count_include_pad=0) for MetaPool on the unbatched token plane, decomposed LayerNorm, MatMul, Resize with
`align_corners` / `pytorch_half_pixel`, Reshape / Transpose / Shape / Slice / Concat for the dynamic H x W plumbing), serialised with a
30-line protobuf writer (ONNX is plain proto3; field numbers from onnx.proto3).  Inputs: `input` [batch,3,height,width] fp32 (height, width
multiples of 16); outputs: `output` = the main head's logits [batch,n,height,width] (what onnx_infer.py reads) followed by the three
deep-supervision heads `output_aux1/2/4`, like the reference's export of the 4-element list.  Host-side only, no GPU needed.

Checked without onnx / onnxruntime: `tests/onnx_mini_runtime.py` parses the file back and evaluates it with torch CPU ops; the result equals the
oracle's eval forward (CPU test) and the REAL reference's logits on its trained Duke checkpoint (`ckpt_duke.npz`, 1e-4)."""
import struct

import numpy as np
import torch

from ._lib import TcctError

KSIZES = (13, 11, 9, 7, 5)
VIT_DIMS = (64, 96, 128, 160)


# ------------------------------------------------------------------------------------------------------------------- protobuf writer
def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7f
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def _f_varint(field, v):
    return _key(field, 0) + _varint(int(v))


def _f_bytes(field, b):
    if isinstance(b, str):
        b = b.encode()
    return _key(field, 2) + _varint(len(b)) + b


def _f_float(field, v):
    return _key(field, 5) + struct.pack('<f', float(v))


def _tensor(name, arr):
    """TensorProto: dims=1, data_type=2 (FLOAT=1, INT64=7), name=8, raw_data=9"""
    arr = np.ascontiguousarray(arr)
    if arr.dtype == np.float32:
        dt = 1
    elif arr.dtype == np.int64:
        dt = 7
    else:
        raise TcctError(f'onnx export: unsupported initializer dtype {arr.dtype}')
    out = b''.join(_f_varint(1, d) for d in arr.shape) + _f_varint(2, dt) + _f_bytes(8, name) + _f_bytes(9, arr.tobytes())
    return out


def _attr(name, v):
    """AttributeProto: name=1, f=2, i=3, s=4, floats=7, ints=8, type=20 (FLOAT=1, INT=2, STRING=3, FLOATS=6, INTS=7)"""
    out = _f_bytes(1, name)
    if isinstance(v, float):
        return out + _f_float(2, v) + _f_varint(20, 1)
    if isinstance(v, int):
        return out + _f_varint(3, v) + _f_varint(20, 2)
    if isinstance(v, str):
        return out + _f_bytes(4, v) + _f_varint(20, 3)
    if isinstance(v, (list, tuple)) and all(isinstance(e, int) for e in v):
        return out + b''.join(_f_varint(8, e) for e in v) + _f_varint(20, 7)
    if isinstance(v, (list, tuple)):
        return out + b''.join(_f_float(7, e) for e in v) + _f_varint(20, 6)
    raise TcctError(f'onnx export: attribute {name}={v!r}')


def _value_info(name, dims):
    """ValueInfoProto{name=1, type=2: TypeProto{tensor_type=1: {elem_type=1 (FLOAT), shape=2: {dim=1: {dim_value=1 | dim_param=2}}}}}"""
    shape = b''
    for d in dims:
        dim = _f_bytes(2, d) if isinstance(d, str) else _f_varint(1, d)
        shape += _f_bytes(1, dim)
    tt = _f_varint(1, 1) + _f_bytes(2, shape)
    return _f_bytes(1, name) + _f_bytes(2, _f_bytes(1, tt))


class _Graph:
    def __init__(self):
        self.nodes, self.inits, self.n = [], [], 0
        self._consts = {}

    def name(self, hint):
        self.n += 1
        return f'{hint}_{self.n}'

    def init(self, hint, arr):
        nm = self.name(hint)
        self.inits.append(_tensor(nm, arr))
        return nm

    def const(self, arr, hint='c'):
        arr = np.asarray(arr)
        key = (arr.dtype.str, arr.shape, arr.tobytes())
        if key not in self._consts:
            self._consts[key] = self.init(hint, arr)
        return self._consts[key]

    def op(self, op_type, inputs, hint=None, n_out=1, out_names=None, **attrs):
        """NodeProto: input=1, output=2, name=3, op_type=4, attribute=5"""
        outs = out_names or [self.name(hint or op_type.lower()) for _ in range(n_out)]
        body = b''.join(_f_bytes(1, i) for i in inputs) + b''.join(_f_bytes(2, o) for o in outs)
        body += _f_bytes(3, self.name('n')) + _f_bytes(4, op_type) + b''.join(_f_bytes(5, _attr(k, v)) for k, v in attrs.items())
        self.nodes.append(body)
        return outs[0] if n_out == 1 else outs

    def serialize(self, inputs, outputs, producer='tcct_amd'):
        """GraphProto: node=1, name=2, initializer=5, input=11, output=12; ModelProto: ir_version=1, producer_name=2, graph=7, opset_import=8"""
        g = b''.join(_f_bytes(1, n) for n in self.nodes) + _f_bytes(2, 'stc_tt_eval') + b''.join(_f_bytes(5, t) for t in self.inits)
        g += b''.join(_f_bytes(11, _value_info(n, d)) for n, d in inputs) + b''.join(_f_bytes(12, _value_info(n, d)) for n, d in outputs)
        opset = _f_bytes(1, '') + _f_varint(2, 11)
        return _f_varint(1, 6) + _f_bytes(2, producer) + _f_bytes(3, '3') + _f_bytes(7, g) + _f_bytes(8, opset)


# --------------------------------------------------------------------------------------------------------------------- graph emission
class _Emit:
    """the eval forward of RegNet(stc_tt(n)) as ONNX nodes; `sd`: reference-keyed state_dict (fp32 CPU tensors)"""

    def __init__(self, sd):
        self.sd = {k: v.detach().float().cpu().numpy() if torch.is_tensor(v) and v.is_floating_point() else v for k, v in sd.items()}
        self.g = _Graph()

    def w(self, key):
        if key not in self.sd:
            raise TcctError(f'onnx export: state_dict has no {key!r} (only the current stc_tt layout with the pooling mixer is exported)')
        return self.g.init(key.replace('.', '_'), np.asarray(self.sd[key], dtype=np.float32))

    def conv(self, p, x, stride=1, pad=(0, 0), groups=1):
        wt = self.sd[p + '.weight']
        ins = [x, self.w(p + '.weight')] + ([self.w(p + '.bias')] if (p + '.bias') in self.sd else [])
        return self.g.op('Conv', ins, 'conv', kernel_shape=[int(wt.shape[2]), int(wt.shape[3])], strides=[stride, stride],
                         pads=[pad[0], pad[1], pad[0], pad[1]], group=groups, dilations=[1, 1])

    def bn(self, p, x, eps=1e-5):
        return self.g.op('BatchNormalization', [x, self.w(p + '.weight'), self.w(p + '.bias'), self.w(p + '.running_mean'), self.w(p + '.running_var')],
                         'bn', epsilon=float(eps), momentum=0.9)

    def lrelu(self, x):
        return self.g.op('LeakyRelu', [x], 'lrelu', alpha=0.01)

    def hswish(self, x):        # torch's opset-11 export of nn.Hardswish: x * HardSigmoid(x; 1/6, 0.5)
        return self.g.op('Mul', [x, self.g.op('HardSigmoid', [x], 'hsig', alpha=1.0 / 6.0, beta=0.5)], 'hswish')

    def gelu(self, x):          # exact (erf) GELU: 0.5 x (1 + erf(x / sqrt 2))
        e = self.g.op('Erf', [self.g.op('Mul', [x, self.g.const(np.float32(0.7071067811865476))])], 'erf')
        return self.g.op('Mul', [self.g.op('Mul', [x, self.g.const(np.float32(0.5))]), self.g.op('Add', [e, self.g.const(np.float32(1.0))])], 'gelu')

    def cba(self, pc, pb, x, pre=None, post=None, res=None, **kw):
        y = self.conv(pc, x, **kw)
        if pre == 'lrelu':
            y = self.lrelu(y)
        y = self.bn(pb, y)
        if post == 'lrelu':
            y = self.lrelu(y)
        elif post == 'hswish':
            y = self.hswish(y)
        return y if res is None else self.g.op('Add', [y, res], 'add')

    def layernorm(self, p, t, eps=1e-6):
        mu = self.g.op('ReduceMean', [t], 'mean', axes=[-1], keepdims=1)
        d = self.g.op('Sub', [t, mu], 'cen')
        var = self.g.op('ReduceMean', [self.g.op('Mul', [d, d])], 'var', axes=[-1], keepdims=1)
        y = self.g.op('Div', [d, self.g.op('Sqrt', [self.g.op('Add', [var, self.g.const(np.float32(eps))])])], 'ln')
        return self.g.op('Add', [self.g.op('Mul', [y, self.w(p + '.weight')]), self.w(p + '.bias')], 'ln')

    def linear(self, p, t):
        wt = np.ascontiguousarray(np.asarray(self.sd[p + '.weight'], dtype=np.float32).T)
        return self.g.op('Add', [self.g.op('MatMul', [t, self.g.init(p.replace('.', '_') + '_wT', wt)]), self.w(p + '.bias')], 'fc')

    def cross_block(self, p, x, k):
        a = self.conv(p + '.block12.1', self.conv(p + '.block12.0', x, pad=(1, 1)), pad=(1, 1))
        a = self.bn(p + '.block12.3', self.lrelu(a))
        b = self.conv(p + '.block34.0', x, pad=(0, k // 2))
        b = self.conv(p + '.block34.1', b, pad=(k // 2, 0))
        b = self.conv(p + '.block34.2', b, pad=(1, 1))
        b = self.bn(p + '.block34.4', self.lrelu(b))
        c = self.gelu(self.g.op('Add', [a, b], 'add'))
        return self.cba(p + '.block5.0', p + '.block5.2', c, pre='lrelu', pad=(1, 1))

    def vit_stage(self, p_pe, p_st, x, s):
        C = VIT_DIMS[s]
        pe = p_pe + '.patch_embeds.0.patch_conv'
        pch = self.cba(pe + '.pwconv', pe + '.bn', self.conv(pe + '.dwconv', x, stride=2 if s > 0 else 1, pad=(1, 1), groups=C), post='hswish')
        r = self.cba(p_st + '.InvRes.conv1.conv', p_st + '.InvRes.conv1.bn', pch, post='hswish')
        r = self.cba(p_st + '.InvRes.dwconv', p_st + '.InvRes.norm', r, post='hswish', pad=(1, 1), groups=C)
        r = self.cba(p_st + '.InvRes.conv2.conv', p_st + '.InvRes.conv2.bn', r, res=pch)
        blk = p_st + '.mhca_blks.0.MHCA_layers.0'
        if (blk + '.att.qkv.weight') in self.sd:
            raise TcctError("onnx export: the factorised-attention variant (att='factor') is not exported")
        img = self.g.op('Add', [pch, self.conv(p_st + '.mhca_blks.0.cpe.proj', pch, pad=(1, 1), groups=C)], 'cpe')
        shape = self.g.op('Shape', [img], 'shape')
        t = self.g.op('Transpose', [self.g.op('Reshape', [img, self.g.const(np.array([0, C, -1], dtype=np.int64))])], 'tok', perm=[0, 2, 1])
        cur = self.layernorm(blk + '.norm1', t)
        # MetaPool on the 3-D token tensor: torch's AvgPool2d treats [B,N,C] as ONE unbatched image with B channels (reference tcct.py:405-415)
        pool = self.g.op('AveragePool', [self.g.op('Unsqueeze', [cur], 'u', axes=[0])], 'pool', kernel_shape=[3, 3], strides=[1, 1], pads=[1, 1, 1, 1],
                         count_include_pad=0)
        a = self.g.op('Sub', [self.g.op('Squeeze', [pool], 'sq', axes=[0]), cur], 'mix')
        t = self.g.op('Add', [t, a], 'res1')
        h = self.gelu(self.linear(blk + '.mlp.fc1', self.layernorm(blk + '.norm2', t)))
        t = self.g.op('Add', [t, self.linear(blk + '.mlp.fc2', h)], 'res2')
        e = self.g.op('Reshape', [self.g.op('Transpose', [t], 'img', perm=[0, 2, 1]), shape], 'img')
        return self.cba(p_st + '.aggregate.conv', p_st + '.aggregate.bn', self.g.op('Concat', [r, e], 'cat', axis=1), post='hswish')

    def resize_x2_align(self, x):
        roi = self.g.const(np.zeros(0, dtype=np.float32), 'roi')
        return self.g.op('Resize', [x, roi, self.g.const(np.array([1, 1, 2, 2], dtype=np.float32), 'scales')], 'up', mode='linear',
                         coordinate_transformation_mode='align_corners')

    def resize_to_input(self, x, in_shape):
        roi = self.g.const(np.zeros(0, dtype=np.float32), 'roi')
        empty = self.g.const(np.zeros(0, dtype=np.float32), 'noscales')
        ax0 = self.g.const(np.array([0], dtype=np.int64))
        s0, s2, s4 = (self.g.const(np.array([v], dtype=np.int64)) for v in (0, 2, 4))
        nc = self.g.op('Slice', [self.g.op('Shape', [x], 'shape'), s0, s2, ax0], 'nc')
        hw = self.g.op('Slice', [in_shape, s2, s4, ax0], 'hw')
        sizes = self.g.op('Concat', [nc, hw], 'sizes', axis=0)
        return self.g.op('Resize', [x, roi, empty, sizes], 'resize', mode='linear', coordinate_transformation_mode='pytorch_half_pixel')

    def up_block(self, p, x1, x2):
        y = self.cba(p + '.prep.0', p + '.prep.1', x1, post='lrelu', pad=(1, 1))
        return self.conv(p + '.post.0', self.g.op('Add', [self.resize_x2_align(y), x2], 'skip'))

    def emit(self, p='base'):
        if (p + '.t324.weight') not in self.sd:
            raise TcctError('onnx export: the legacy head layout (onnx/tcct_goals.py) is not exported; convert with legacy_heads=False weights')
        x = 'input'
        in_shape = self.g.op('Shape', [x], 'inshape')
        # CNN branch (CrossResNet.forward, tcct.py:877-885)
        c, cur = [], self.cba(p + '.base_cnn.cnn.0', p + '.base_cnn.cnn.1', x, pad=(1, 1))
        for i, k in enumerate(KSIZES):
            cur = self.cross_block(f'{p}.base_cnn.path_estan.{i}', cur, k)
            c.append(cur)
            if i < 4:
                cur = self.g.op('MaxPool', [cur], 'pool', kernel_shape=[2, 2], strides=[2, 2])
        # ViT branch (MPViT.forward_features, tcct.py:733-745)
        v, cur = [], self.cba(p + '.base_vit.stem.0.conv', p + '.base_vit.stem.0.bn', x, post='hswish', stride=2, pad=(1, 1))
        cur = self.cba(p + '.base_vit.stem.1.conv', p + '.base_vit.stem.1.bn', cur, post='hswish', pad=(1, 1))
        for s in range(4):
            cur = self.vit_stage(f'{p}.base_vit.patch_embed_stages.{s}', f'{p}.base_vit.mhca_stages.{s}', cur, s)
            v.append(cur)
        f = [c[0]]
        for j in range(4):
            tv = self.cba(f'{p}.tran_vit{j}.0', f'{p}.tran_vit{j}.1', v[j])
            f.append(self.cba(f'{p}.tran_cnn{j}.0', f'{p}.tran_cnn{j}.1', c[j + 1], res=tv))
        y8 = self.cba(p + '.head.0', p + '.head.1', f[4], post='lrelu', pad=(1, 1))
        d3 = self.up_block(p + '.dec1', y8, f[3])
        d2 = self.up_block(p + '.dec2', d3, f[2])
        d1 = self.up_block(p + '.dec3', d2, f[1])
        d0 = self.up_block(p + '.dec4', d1, f[0])
        g = [self.conv(f'{p}.t32{4 - i}', self.g.op('Add', [f[i], d], 'sum')) for i, d in enumerate((d0, d1, d2, d3))]
        outs = [self.conv(p + '.aux0', g[0])]
        for name, gi in (('aux1', g[1]), ('aux2', g[2]), ('aux4', g[3])):
            outs.append(self.resize_to_input(self.conv(f'{p}.{name}', gi), in_shape))
        return outs


def export_onnx(model_or_state_dict, path, in_channels=3):
    """Write the eval-mode network as an ONNX file (opset 11, dynamic batch / height / width) -- the counterpart of the reference's
    `to_onnx(net, in_channels=3, path_onnx=...)` (task1/onnx/onnx_save.py:4-15).  `model_or_state_dict`: a `RegNet(stc_tt(n))` (any compute
    dtype; parameters are exported as fp32) or a reference-keyed state_dict / checkpoint dict.  Returns the list of output names."""
    if in_channels != 3:
        raise TcctError('onnx export: the network takes 3-channel input (reference nets/tcct.py:873)')
    sd = model_or_state_dict.state_dict() if hasattr(model_or_state_dict, 'state_dict') else model_or_state_dict
    em = _Emit(sd)
    outs = em.emit('base')
    n_class = int(np.asarray(em.sd['base.aux0.weight']).shape[0])
    names = ['output', 'output_aux1', 'output_aux2', 'output_aux4']
    g = em.g
    for nm, o in zip(names, outs):          # the public output names
        g.op('Identity', [o], out_names=[nm])
    dims = ['batch', n_class, 'height', 'width']
    blob = g.serialize([('input', ['batch', 3, 'height', 'width'])], [(nm, dims) for nm in names])
    with open(path, 'wb') as fh:
        fh.write(blob)
    return names


if __name__ == '__main__':      # the reference's script form (onnx_save.py:19-37): `python -m tcct_amd.onnx_export tcct_duke.pt [tcct_duke.onnx]`
    import sys

    from .checkpoint import read_checkpoint
    if len(sys.argv) < 2:
        raise SystemExit('usage: python -m tcct_amd.onnx_export CHECKPOINT(.pt|.npz) [OUT.onnx]')
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else src.rsplit('.', 1)[0] + '.onnx'
    print('saving onnx to:', dst)
    print('outputs:', export_onnx(read_checkpoint(src), dst))
