"""Test infrastructure: a minimal ONNX reader + evaluator (this image has neither `onnx` nor `onnxruntime`).

`load(path)` decodes the protobuf wire format with the field numbers of onnx.proto3 (independent of the writer in tcct_amd/onnx_export.py:
it is a generic schema-driven decoder) into plain dicts; `run(model, {'input': array})` evaluates the graph with torch CPU ops following the
ONNX operator specifications at opset 11 (the subset the exporter emits) and returns the list of outputs like `InferenceSession.run(None, ...)`
in the reference's task1/onnx/onnx_infer.py:13-30."""
import struct

import numpy as np
import torch
import torch.nn.functional as F

# message schemas: field -> (name, kind, repeated); kind: 'int' | 'float' | 'str' | 'bytes' | schema-name
SCHEMAS = {
    'Model': {1: ('ir_version', 'int', 0), 2: ('producer_name', 'str', 0), 3: ('producer_version', 'str', 0), 7: ('graph', 'Graph', 0),
              8: ('opset_import', 'Opset', 1)},
    'Opset': {1: ('domain', 'str', 0), 2: ('version', 'int', 0)},
    'Graph': {1: ('node', 'Node', 1), 2: ('name', 'str', 0), 5: ('initializer', 'Tensor', 1), 11: ('input', 'ValueInfo', 1),
              12: ('output', 'ValueInfo', 1)},
    'Node': {1: ('input', 'str', 1), 2: ('output', 'str', 1), 3: ('name', 'str', 0), 4: ('op_type', 'str', 0), 5: ('attribute', 'Attr', 1)},
    'Attr': {1: ('name', 'str', 0), 2: ('f', 'float', 0), 3: ('i', 'int', 0), 4: ('s', 'str', 0), 7: ('floats', 'float', 1),
             8: ('ints', 'int', 1), 20: ('type', 'int', 0)},
    'Tensor': {1: ('dims', 'int', 1), 2: ('data_type', 'int', 0), 8: ('name', 'str', 0), 9: ('raw_data', 'bytes', 0)},
    'ValueInfo': {1: ('name', 'str', 0), 2: ('type', 'Type', 0)},
    'Type': {1: ('tensor_type', 'TensorType', 0)},
    'TensorType': {1: ('elem_type', 'int', 0), 2: ('shape', 'Shape', 0)},
    'Shape': {1: ('dim', 'Dim', 1)},
    'Dim': {1: ('dim_value', 'int', 0), 2: ('dim_param', 'str', 0)},
}


def _varint(buf, pos):
    v, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7f) << shift
        shift += 7
        if not b & 0x80:
            return v, pos


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def decode(buf, schema):
    sch, out, pos = SCHEMAS[schema], {}, 0
    for _, (name, _, rep) in sch.items():
        if rep:
            out[name] = []
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            val, pos = _varint(buf, pos)
            val = _signed(val)
        elif wire == 2:
            n, pos = _varint(buf, pos)
            val = bytes(buf[pos:pos + n])
            pos += n
        elif wire == 5:
            val = struct.unpack('<f', buf[pos:pos + 4])[0]
            pos += 4
        elif wire == 1:
            val = struct.unpack('<d', buf[pos:pos + 8])[0]
            pos += 8
        else:
            raise ValueError(f'wire type {wire}')
        if field not in sch:
            continue
        name, kind, rep = sch[field]
        if kind == 'str':
            val = val.decode()
        elif kind in SCHEMAS:
            val = decode(val, kind)
        elif kind in ('int', 'float') and wire == 2:        # packed repeated scalars
            vals, p = [], 0
            while p < len(val):
                if kind == 'int':
                    v, p = _varint(val, p)
                    vals.append(_signed(v))
                else:
                    vals.append(struct.unpack('<f', val[p:p + 4])[0])
                    p += 4
            out[name].extend(vals)
            continue
        if rep:
            out[name].append(val)
        else:
            out[name] = val
    return out


def load(path):
    with open(path, 'rb') as fh:
        return decode(fh.read(), 'Model')


def _tensor(t):
    dt = {1: np.float32, 7: np.int64}[t['data_type']]
    return torch.from_numpy(np.frombuffer(t['raw_data'], dtype=dt).reshape(t['dims']).copy())


def _attrs(node):
    a = {}
    for at in node['attribute']:
        ty = at['type']
        a[at['name']] = {1: at.get('f'), 2: at.get('i'), 3: at.get('s'), 6: at['floats'], 7: at['ints']}[ty]
    return a


def _resize(x, a, scales, sizes):
    assert a['mode'] == 'linear'
    ctm = a['coordinate_transformation_mode']
    if sizes is not None:
        size = [int(v) for v in sizes[2:]]
    else:
        assert float(scales[0]) == 1 and float(scales[1]) == 1
        size = [int(np.floor(x.shape[2] * float(scales[2]))), int(np.floor(x.shape[3] * float(scales[3])))]
    if ctm == 'align_corners':
        return F.interpolate(x, size=size, mode='bilinear', align_corners=True)
    assert ctm in ('pytorch_half_pixel', 'half_pixel')
    return F.interpolate(x, size=size, mode='bilinear', align_corners=False)


def run(model, feeds, dtype=torch.float32):
    """evaluate; `dtype`: float arithmetic of the evaluation (float64 gives a rounding-free reference of the graph)"""
    g = model['graph']
    assert model['opset_import'][0]['version'] == 11
    env = {}
    for t in g['initializer']:
        v = _tensor(t)
        env[t['name']] = v.to(dtype) if v.is_floating_point() else v
    for k, v in feeds.items():
        env[k] = torch.as_tensor(np.asarray(v)).to(dtype)
    for nd in g['node']:
        a, op = _attrs(nd), nd['op_type']
        i = [env[n] if n else None for n in nd['input']]
        if op == 'Conv':
            p = a['pads']
            assert p[0] == p[2] and p[1] == p[3] and a['dilations'] == [1, 1] and list(i[1].shape[2:]) == a['kernel_shape']
            y = F.conv2d(i[0], i[1], i[2] if len(i) > 2 else None, stride=a['strides'], padding=(p[0], p[1]), groups=a['group'])
        elif op == 'BatchNormalization':
            y = F.batch_norm(i[0], i[3], i[4], i[1], i[2], False, 0.0, a['epsilon'])
        elif op == 'LeakyRelu':
            y = F.leaky_relu(i[0], a['alpha'])
        elif op == 'HardSigmoid':
            y = torch.clamp(a['alpha'] * i[0] + a['beta'], 0, 1)
        elif op == 'Erf':
            y = torch.erf(i[0])
        elif op in ('Add', 'Sub', 'Mul', 'Div'):
            y = {'Add': torch.add, 'Sub': torch.sub, 'Mul': torch.mul, 'Div': torch.div}[op](i[0], i[1])
        elif op == 'Sqrt':
            y = torch.sqrt(i[0])
        elif op == 'MaxPool':
            assert a['kernel_shape'] == [2, 2] and a['strides'] == [2, 2]
            y = F.max_pool2d(i[0], 2)
        elif op == 'AveragePool':
            p = a['pads']
            assert p == [1, 1, 1, 1] and a['kernel_shape'] == [3, 3] and a['strides'] == [1, 1]
            y = F.avg_pool2d(i[0], 3, 1, 1, count_include_pad=bool(a.get('count_include_pad', 0)))
        elif op == 'ReduceMean':
            y = i[0].mean(dim=a['axes'], keepdim=bool(a.get('keepdims', 1)))
        elif op == 'MatMul':
            y = torch.matmul(i[0], i[1])
        elif op == 'Shape':
            y = torch.tensor(list(i[0].shape), dtype=torch.int64)
        elif op == 'Reshape':
            shp = [int(i[0].shape[k]) if int(v) == 0 else int(v) for k, v in enumerate(i[1])]
            y = i[0].reshape(shp)
        elif op == 'Transpose':
            y = i[0].permute(a['perm'])
        elif op == 'Unsqueeze':
            y = i[0]
            for ax in sorted(a['axes']):
                y = y.unsqueeze(ax)
        elif op == 'Squeeze':
            y = i[0]
            for ax in sorted(a['axes'], reverse=True):
                assert y.shape[ax] == 1
                y = y.squeeze(ax)
        elif op == 'Concat':
            y = torch.cat(i, dim=a['axis'])
        elif op == 'Slice':
            assert len(i[1]) == 1 and int(i[3][0]) == 0
            y = i[0][int(i[1][0]):int(i[2][0])]
        elif op == 'Resize':
            scales = i[2] if i[2] is not None and i[2].numel() else None
            sizes = i[3] if len(i) > 3 and i[3] is not None else None
            y = _resize(i[0], a, scales, sizes)
        elif op == 'Identity':
            y = i[0]
        else:
            raise NotImplementedError(op)
        env[nd['output'][0]] = y
    return [env[o['name']].float().numpy() for o in g['output']]
