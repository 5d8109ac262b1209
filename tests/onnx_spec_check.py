"""Test infrastructure: a STRUCTURAL validator for the ONNX files tcct_amd/onnx_export.py writes, written from the published ONNX specification
(onnx.proto3 of IR version 6 / onnx 1.6, and docs/Operators.md at opset 11) -- NOT from the exporter.  This image has neither `onnx` nor
`onnxruntime`; it does have Google's `protobuf` runtime, which is used here as the third-party reader of the wire format:

  1. the message schemas of onnx.proto3 (field numbers, labels, scalar types, enum values) are declared below as a FileDescriptorProto and turned
     into message classes by google.protobuf; `ModelProto.ParseFromString` then decodes the file.  A field the writer emitted under a wrong number or
     with a wrong wire type shows up as an UNKNOWN field of its message, which `check_model` rejects;
  2. `check_model` applies the IR rules (`onnx/docs/IR.md`: ir_version / opset_import, SSA form, topological order, initializers, typed graph
     inputs / outputs with dim_param / dim_value) and, per node, the operator schema of opset 11 (input / output arity, attribute names, attribute
     types, required attributes, enumerated string values, the Resize "scales XOR sizes" rule, Slice / Reshape / Resize auxiliary tensor types);
  3. a channel-count propagation over the NCHW part of the graph (Conv / BatchNormalization / Concat / pooling / elementwise) checks that weight
     shapes, groups and normalisation vectors are consistent with what flows into them.

What this is not: `onnx.checker` or an `onnxruntime` session -- no file written by the exporter has been opened by either."""
import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto
_T = dict(int32=_F.TYPE_INT32, int64=_F.TYPE_INT64, uint64=_F.TYPE_UINT64, float=_F.TYPE_FLOAT, double=_F.TYPE_DOUBLE, string=_F.TYPE_STRING,
          bytes=_F.TYPE_BYTES)

# onnx.proto3 (IR version 6).  name: [(field number, field name, type, repeated)]
_MESSAGES = {
    'AttributeProto': [(1, 'name', 'string', 0), (21, 'ref_attr_name', 'string', 0), (13, 'doc_string', 'string', 0), (20, 'type', 'int32', 0),
                       (2, 'f', 'float', 0), (3, 'i', 'int64', 0), (4, 's', 'bytes', 0), (5, 't', 'TensorProto', 0), (6, 'g', 'GraphProto', 0),
                       (22, 'sparse_tensor', 'bytes', 0), (7, 'floats', 'float', 1), (8, 'ints', 'int64', 1), (9, 'strings', 'bytes', 1),
                       (10, 'tensors', 'TensorProto', 1), (11, 'graphs', 'GraphProto', 1), (23, 'sparse_tensors', 'bytes', 1)],
    'ValueInfoProto': [(1, 'name', 'string', 0), (2, 'type', 'TypeProto', 0), (3, 'doc_string', 'string', 0)],
    'NodeProto': [(1, 'input', 'string', 1), (2, 'output', 'string', 1), (3, 'name', 'string', 0), (4, 'op_type', 'string', 0), (7, 'domain', 'string', 0),
                  (5, 'attribute', 'AttributeProto', 1), (6, 'doc_string', 'string', 0)],
    'ModelProto': [(1, 'ir_version', 'int64', 0), (8, 'opset_import', 'OperatorSetIdProto', 1), (2, 'producer_name', 'string', 0),
                   (3, 'producer_version', 'string', 0), (4, 'domain', 'string', 0), (5, 'model_version', 'int64', 0), (6, 'doc_string', 'string', 0),
                   (7, 'graph', 'GraphProto', 0), (14, 'metadata_props', 'StringStringEntryProto', 1)],
    'StringStringEntryProto': [(1, 'key', 'string', 0), (2, 'value', 'string', 0)],
    'GraphProto': [(1, 'node', 'NodeProto', 1), (2, 'name', 'string', 0), (5, 'initializer', 'TensorProto', 1), (15, 'sparse_initializer', 'bytes', 1),
                   (10, 'doc_string', 'string', 0), (11, 'input', 'ValueInfoProto', 1), (12, 'output', 'ValueInfoProto', 1),
                   (13, 'value_info', 'ValueInfoProto', 1), (14, 'quantization_annotation', 'bytes', 1)],
    'TensorProto': [(1, 'dims', 'int64', 1), (2, 'data_type', 'int32', 0), (3, 'segment', 'bytes', 0), (4, 'float_data', 'float', 1),
                    (5, 'int32_data', 'int32', 1), (6, 'string_data', 'bytes', 1), (7, 'int64_data', 'int64', 1), (8, 'name', 'string', 0),
                    (12, 'doc_string', 'string', 0), (9, 'raw_data', 'bytes', 0), (13, 'external_data', 'StringStringEntryProto', 1),
                    (14, 'data_location', 'int32', 0), (10, 'double_data', 'double', 1), (11, 'uint64_data', 'uint64', 1)],
    'TensorShapeProto': [(1, 'dim', 'Dimension', 1)],
    'Dimension': [(1, 'dim_value', 'int64', 0), (2, 'dim_param', 'string', 0), (3, 'denotation', 'string', 0)],
    'TypeProto': [(1, 'tensor_type', 'TypeTensor', 0), (6, 'denotation', 'string', 0)],
    'TypeTensor': [(1, 'elem_type', 'int32', 0), (2, 'shape', 'TensorShapeProto', 0)],
    'OperatorSetIdProto': [(1, 'domain', 'string', 0), (2, 'version', 'int64', 0)],
}
# AttributeProto.AttributeType and TensorProto.DataType enum values (onnx.proto3)
ATTR = dict(FLOAT=1, INT=2, STRING=3, TENSOR=4, GRAPH=5, FLOATS=6, INTS=7, STRINGS=8, TENSORS=9, GRAPHS=10)
ATTR_FIELD = {1: 'f', 2: 'i', 3: 's', 4: 't', 5: 'g', 6: 'floats', 7: 'ints', 8: 'strings', 9: 'tensors', 10: 'graphs'}
DTYPE = {1: ('FLOAT', np.float32), 6: ('INT32', np.int32), 7: ('INT64', np.int64), 11: ('DOUBLE', np.float64)}
IR_VERSION_FOR_OPSET = {9: 4, 10: 5, 11: 6}     # onnx/docs/Versioning.md: onnx 1.4 / 1.5 / 1.6


def _classes():
    fd = descriptor_pb2.FileDescriptorProto(name='onnx_spec_check.proto', package='onnx_spec_check', syntax='proto2')
    for mname, fields in _MESSAGES.items():
        m = fd.message_type.add(name=mname)
        for num, fname, ftype, rep in fields:
            f = m.field.add(name=fname, number=num, label=_F.LABEL_REPEATED if rep else _F.LABEL_OPTIONAL)
            if ftype in _T:
                f.type = _T[ftype]
                if rep and ftype not in ('string', 'bytes'):
                    f.options.packed = True         # proto3 packs repeated scalars by default; the parser accepts both encodings
            else:
                f.type, f.type_name = _F.TYPE_MESSAGE, '.onnx_spec_check.' + ftype
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName('onnx_spec_check.' + n)) for n in _MESSAGES}


_CLS = _classes()


def parse(path_or_bytes):
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, 'rb').read()
    m = _CLS['ModelProto']()
    m.ParseFromString(bytes(data))
    return m


def _no_unknown_fields(msg, where, errs):
    from google.protobuf import unknown_fields
    uf = unknown_fields.UnknownFieldSet(msg)
    if len(uf):
        errs.append(f'{where}: fields not in onnx.proto3 (or with the wrong wire type): {[(u.field_number, u.wire_type) for u in uf]}')
    for fd, val in msg.ListFields():
        if fd.type == fd.TYPE_MESSAGE:
            for i, v in enumerate(val if fd.is_repeated else [val]):
                _no_unknown_fields(v, f'{where}.{fd.name}[{i}]', errs)


# ---- operator schemas at opset 11 (docs/Operators.md).  op: (since_version, min inputs, max inputs, min outputs, max outputs,
#      {attribute: (type, required)}, {string attribute: allowed values})
INF = 1 << 30
_POOL = {'auto_pad': ('STRING', 0), 'ceil_mode': ('INT', 0), 'kernel_shape': ('INTS', 1), 'pads': ('INTS', 0), 'strides': ('INTS', 0)}
_AUTO_PAD = {'auto_pad': {'NOTSET', 'SAME_UPPER', 'SAME_LOWER', 'VALID'}}
OPSET11 = {
    'Conv': (11, 2, 3, 1, 1, {'auto_pad': ('STRING', 0), 'dilations': ('INTS', 0), 'group': ('INT', 0), 'kernel_shape': ('INTS', 0), 'pads': ('INTS', 0),
                              'strides': ('INTS', 0)}, _AUTO_PAD),
    'BatchNormalization': (9, 5, 5, 1, 5, {'epsilon': ('FLOAT', 0), 'momentum': ('FLOAT', 0)}, {}),
    'LeakyRelu': (6, 1, 1, 1, 1, {'alpha': ('FLOAT', 0)}, {}),
    'HardSigmoid': (6, 1, 1, 1, 1, {'alpha': ('FLOAT', 0), 'beta': ('FLOAT', 0)}, {}),
    'Add': (7, 2, 2, 1, 1, {}, {}), 'Sub': (7, 2, 2, 1, 1, {}, {}), 'Mul': (7, 2, 2, 1, 1, {}, {}), 'Div': (7, 2, 2, 1, 1, {}, {}),
    'Erf': (9, 1, 1, 1, 1, {}, {}), 'Sqrt': (6, 1, 1, 1, 1, {}, {}), 'Identity': (1, 1, 1, 1, 1, {}, {}), 'Shape': (1, 1, 1, 1, 1, {}, {}),
    'MaxPool': (11, 1, 1, 1, 2, dict(_POOL, dilations=('INTS', 0), storage_order=('INT', 0)), _AUTO_PAD),
    'AveragePool': (11, 1, 1, 1, 1, dict(_POOL, count_include_pad=('INT', 0)), _AUTO_PAD),
    'Reshape': (5, 2, 2, 1, 1, {}, {}),
    'Transpose': (1, 1, 1, 1, 1, {'perm': ('INTS', 0)}, {}),
    'ReduceMean': (11, 1, 1, 1, 1, {'axes': ('INTS', 0), 'keepdims': ('INT', 0)}, {}),
    'Unsqueeze': (11, 1, 1, 1, 1, {'axes': ('INTS', 1)}, {}),
    'Squeeze': (11, 1, 1, 1, 1, {'axes': ('INTS', 0)}, {}),
    'MatMul': (9, 2, 2, 1, 1, {}, {}),
    'Concat': (11, 1, INF, 1, 1, {'axis': ('INT', 1)}, {}),
    'Resize': (11, 3, 4, 1, 1, {'coordinate_transformation_mode': ('STRING', 0), 'cubic_coeff_a': ('FLOAT', 0), 'exclude_outside': ('INT', 0),
                                'extrapolation_value': ('FLOAT', 0), 'mode': ('STRING', 0), 'nearest_mode': ('STRING', 0)},
               {'coordinate_transformation_mode': {'half_pixel', 'pytorch_half_pixel', 'align_corners', 'asymmetric', 'tf_half_pixel_for_nn',
                                                   'tf_crop_and_resize'},
                'mode': {'nearest', 'linear', 'cubic'}, 'nearest_mode': {'round_prefer_floor', 'round_prefer_ceil', 'floor', 'ceil'}}),
    'Slice': (11, 3, 5, 1, 1, {}, {}),
    'Gather': (11, 2, 2, 1, 1, {'axis': ('INT', 0)}, {}),
    'Cast': (9, 1, 1, 1, 1, {'to': ('INT', 1)}, {}),
    'Constant': (11, 0, 0, 1, 1, {'value': ('TENSOR', 0), 'sparse_value': ('TENSOR', 0)}, {}),
    'Relu': (6, 1, 1, 1, 1, {}, {}), 'Sigmoid': (6, 1, 1, 1, 1, {}, {}), 'Softmax': (11, 1, 1, 1, 1, {'axis': ('INT', 0)}, {}),
    'ArgMax': (11, 1, 1, 1, 1, {'axis': ('INT', 0), 'keepdims': ('INT', 0)}, {}),
}
_KEEPS_CHANNELS = {'LeakyRelu', 'HardSigmoid', 'Erf', 'Sqrt', 'Identity', 'MaxPool', 'AveragePool', 'Resize', 'Relu', 'Sigmoid', 'BatchNormalization'}


def tensor_array(t):
    name, np_t = DTYPE[t.data_type]
    n = int(np.prod(list(t.dims))) if len(t.dims) else 1
    if t.raw_data:
        a = np.frombuffer(t.raw_data, dtype=np.dtype(np_t).newbyteorder('<'))
    elif name == 'FLOAT':
        a = np.array(t.float_data, np.float32)
    elif name == 'INT64':
        a = np.array(t.int64_data, np.int64)
    else:
        a = np.array(t.int32_data, np_t)
    if a.size != n:
        raise ValueError(f'initializer {t.name!r}: {a.size} elements stored for dims {list(t.dims)}')
    return a.reshape(list(t.dims))


def check_model(path_or_bytes, opset=11, expect_inputs=None, expect_outputs=None):
    """raises AssertionError listing every violation; returns (parsed ModelProto, {tensor name: channel count} of the NCHW part)"""
    m = parse(path_or_bytes)
    errs = []
    _no_unknown_fields(m, 'model', errs)
    # ---- IR.md: model level
    if m.ir_version != IR_VERSION_FOR_OPSET[opset]:
        errs.append(f'ir_version {m.ir_version}: opset {opset} was released with IR version {IR_VERSION_FOR_OPSET[opset]}')
    default = [o for o in m.opset_import if o.domain in ('', 'ai.onnx')]
    if len(default) != 1 or default[0].version != opset:
        errs.append(f'opset_import must name the default domain exactly once with version {opset}: {[(o.domain, o.version) for o in m.opset_import]}')
    if not m.HasField('graph'):
        raise AssertionError('no graph')
    g = m.graph
    if not g.name:
        errs.append('GraphProto.name is required')
    # ---- initializers
    init = {}
    for t in g.initializer:
        if not t.name or t.name in init:
            errs.append(f'initializer name missing or repeated: {t.name!r}')
        if t.data_type not in DTYPE:
            errs.append(f'initializer {t.name!r}: data_type {t.data_type} unexpected')
            continue
        if t.data_location != 0:
            errs.append(f'initializer {t.name!r}: external data')
        try:
            init[t.name] = tensor_array(t)
        except ValueError as e:
            errs.append(str(e))
    # ---- typed graph inputs / outputs
    def vinfo(v, what):
        if not v.name:
            errs.append(f'{what}: unnamed')
        tt = v.type.tensor_type
        if not v.type.HasField('tensor_type') or tt.elem_type not in DTYPE:
            errs.append(f'{what} {v.name!r}: needs a tensor type with an element type')
        dims = []
        for d in tt.shape.dim:
            has_v, has_p = d.HasField('dim_value'), d.HasField('dim_param')
            if has_v == has_p:
                errs.append(f'{what} {v.name!r}: every dimension is EITHER dim_value or dim_param')
            dims.append(d.dim_param if has_p else d.dim_value)
        return dims
    in_dims = {v.name: vinfo(v, 'graph input') for v in g.input}
    out_dims = {v.name: vinfo(v, 'graph output') for v in g.output}
    if expect_inputs is not None and in_dims != expect_inputs:
        errs.append(f'graph inputs {in_dims} != {expect_inputs}')
    if expect_outputs is not None and out_dims != expect_outputs:
        errs.append(f'graph outputs {out_dims} != {expect_outputs}')
    # ---- nodes: SSA, topological order, operator schemas
    known = set(in_dims) | set(init)
    if set(in_dims) & set(init) and m.ir_version < 4:
        pass
    produced, names = {}, set()
    chan = {n: d[1] for n, d in in_dims.items() if len(d) == 4 and isinstance(d[1], int)}      # NCHW channel counts
    for idx, n in enumerate(g.node):
        where = f'node[{idx}] {n.op_type} {n.name!r}'
        if n.domain not in ('', 'ai.onnx'):
            errs.append(f'{where}: domain {n.domain!r} is not imported')
        if n.name:
            if n.name in names:
                errs.append(f'{where}: node names must be unique')
            names.add(n.name)
        sch = OPSET11.get(n.op_type)
        if sch is None:
            errs.append(f'{where}: operator not in the opset-{opset} table of this validator')
            continue
        since, imin, imax, omin, omax, attrs, enums = sch
        if since > opset:
            errs.append(f'{where}: introduced in opset {since}')
        if not imin <= len(n.input) <= imax:
            errs.append(f'{where}: {len(n.input)} inputs, schema allows {imin}..{imax}')
        if not omin <= len(n.output) <= omax:
            errs.append(f'{where}: {len(n.output)} outputs, schema allows {omin}..{omax}')
        for i in n.input:
            if i and i not in known:
                errs.append(f'{where}: input {i!r} is not a graph input, an initializer or the output of an EARLIER node (topological order)')
        for o in n.output:
            if not o or o in known:
                errs.append(f'{where}: output {o!r} empty or assigned twice (SSA)')
            known.add(o)
            produced[o] = n
        seen = set()
        for a in n.attribute:
            if a.name in seen:
                errs.append(f'{where}: attribute {a.name!r} given twice')
            seen.add(a.name)
            if a.name not in attrs:
                errs.append(f'{where}: attribute {a.name!r} is not in the schema ({sorted(attrs)})')
                continue
            want = ATTR[attrs[a.name][0]]
            if a.type != want:
                errs.append(f'{where}: attribute {a.name!r} has type {a.type}, schema says {attrs[a.name][0]} = {want}')
            populated = {f.name for f, _ in a.ListFields()} - {'name', 'type', 'doc_string'}
            if populated - {ATTR_FIELD[want]}:
                errs.append(f'{where}: attribute {a.name!r} of type {attrs[a.name][0]} populates {sorted(populated)}')
            if a.name in enums and a.s.decode() not in enums[a.name]:
                errs.append(f'{where}: {a.name} = {a.s.decode()!r} not one of {sorted(enums[a.name])}')
        for an, (_, req) in attrs.items():
            if req and an not in seen:
                errs.append(f'{where}: required attribute {an!r} missing')
        A = {a.name: a for a in n.attribute}
        ins = list(n.input)

        def const(i, dtype, what):
            if i >= len(ins) or not ins[i]:
                return None
            if ins[i] in init:
                if init[ins[i]].dtype != dtype:
                    errs.append(f'{where}: {what} must be {np.dtype(dtype).name}, initializer {ins[i]!r} is {init[ins[i]].dtype}')
                return init[ins[i]]
            return 'dynamic'
        # ---- per-operator rules of Operators.md beyond arity / attributes
        if n.op_type == 'Resize':
            roi, scales, sizes = const(1, np.float32, 'roi'), const(2, np.float32, 'scales'), const(3, np.int64, 'sizes')
            if len(ins) >= 2 and not ins[1]:
                errs.append(f'{where}: roi is a non-optional input at opset 11 (pass an empty tensor)')
            has_scales = scales is not None and (isinstance(scales, str) or scales.size > 0)
            has_sizes = sizes is not None and (isinstance(sizes, str) or sizes.size > 0)
            if has_scales == has_sizes:
                errs.append(f'{where}: exactly ONE of scales / sizes must be given (the other empty)')
            for v, nm in ((scales, 'scales'), (sizes, 'sizes')):
                if isinstance(v, np.ndarray) and v.size and v.shape != (4,):
                    errs.append(f'{where}: {nm} must have one entry per input dimension (4), has shape {v.shape}')
            if isinstance(roi, np.ndarray) and roi.size and ('coordinate_transformation_mode' not in A or A['coordinate_transformation_mode'].s != b'tf_crop_and_resize'):
                errs.append(f'{where}: roi only takes effect with tf_crop_and_resize')
        if n.op_type == 'Slice':
            for i, nm in ((1, 'starts'), (2, 'ends'), (3, 'axes'), (4, 'steps')):
                const(i, np.int64, nm)
        if n.op_type == 'Reshape':
            const(1, np.int64, 'shape')
        if n.op_type in ('MaxPool', 'AveragePool', 'Conv'):
            ks = list(A['kernel_shape'].ints) if 'kernel_shape' in A else None
            for an in ('strides', 'dilations'):
                if an in A and ks is not None and len(A[an].ints) != len(ks):
                    errs.append(f'{where}: {an} needs one value per spatial axis')
            if 'pads' in A and ks is not None and len(A['pads'].ints) != 2 * len(ks):
                errs.append(f'{where}: pads needs 2 values per spatial axis (begins then ends)')
            if 'pads' in A and 'auto_pad' in A and A['auto_pad'].s != b'NOTSET':
                errs.append(f'{where}: pads and auto_pad are mutually exclusive')
        # ---- channel propagation over the NCHW part of the graph
        c_in = chan.get(ins[0]) if ins else None
        c_out = None
        if n.op_type == 'Conv':
            w = init.get(ins[1]) if len(ins) > 1 else None
            if w is None or w.ndim != 4:
                errs.append(f'{where}: weight must be a rank-4 initializer here')
            else:
                grp = A['group'].i if 'group' in A else 1
                if 'kernel_shape' in A and list(A['kernel_shape'].ints) != list(w.shape[2:]):
                    errs.append(f'{where}: kernel_shape {list(A["kernel_shape"].ints)} != weight {w.shape}')
                if w.shape[0] % grp:
                    errs.append(f'{where}: {w.shape[0]} output channels not divisible by group {grp}')
                if c_in is not None and c_in != w.shape[1] * grp:
                    errs.append(f'{where}: input has {c_in} channels, weight {w.shape} x group {grp} expects {w.shape[1] * grp}')
                if len(ins) > 2 and ins[2]:
                    b = init.get(ins[2])
                    if b is None or b.shape != (w.shape[0],):
                        errs.append(f'{where}: bias must be a [{w.shape[0]}] initializer')
                c_out = w.shape[0]
        elif n.op_type == 'BatchNormalization':
            vecs = [init.get(i) for i in ins[1:5]]
            if any(v is None or v.ndim != 1 for v in vecs) or len({v.shape for v in vecs if v is not None}) != 1:
                errs.append(f'{where}: scale / B / mean / var must be 1-D initializers of one length')
            elif c_in is not None and vecs[0].shape[0] != c_in:
                errs.append(f'{where}: {vecs[0].shape[0]} statistics for {c_in} channels')
            if len(vecs) == 4 and vecs[3] is not None and (vecs[3] < 0).any():
                errs.append(f'{where}: negative variance')
            c_out = c_in
        elif n.op_type == 'Concat':
            if A.get('axis') is not None and A['axis'].i == 1 and all(i in chan for i in ins):
                c_out = sum(chan[i] for i in ins)
        elif n.op_type in ('Add', 'Sub', 'Mul', 'Div'):
            cs = [chan[i] for i in ins if i in chan]
            if len(cs) == 2 and cs[0] != cs[1] and 1 not in cs:
                errs.append(f'{where}: operands with {cs} channels do not broadcast')
            c_out = max(cs) if cs and all(i in chan or i in init for i in ins) else None
        elif n.op_type in _KEEPS_CHANNELS:
            c_out = c_in
        if c_out is not None:
            for o in n.output[:1]:
                chan[o] = int(c_out)
    for name, dims in out_dims.items():
        if name not in produced:
            errs.append(f'graph output {name!r} is produced by no node')
        elif len(dims) == 4 and isinstance(dims[1], int) and name in chan and chan[name] != dims[1]:
            errs.append(f'graph output {name!r}: declared {dims[1]} channels, the graph produces {chan[name]}')
    used = {i for n in g.node for i in n.input}
    dead = [t for t in init if t not in used]
    if dead:
        errs.append(f'initializers no node reads: {dead[:5]}')
    assert not errs, '\n'.join(errs[:40]) + (f'\n... {len(errs) - 40} more' if len(errs) > 40 else '')
    return m, chan
