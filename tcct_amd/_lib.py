"""ctypes binding of libtcct_hip.so.  The argtypes are parsed from include/tcct_hip.h so the Python side can never
drift from the C-ABI.  There is NO fallback: if the library is missing or a call fails, this raises."""
import ctypes
import threading
import os
import re

import torch  # noqa: F401  (must be imported first: the .so then binds to torch's already-mapped HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libtcct_hip.so')
if os.environ.get('TCCT_LIB_PATH'):         # kernel A/B measurements only (tools/ab_kernels.sh): another BUILD of the same library, never a fallback
    LIB_PATH = os.path.abspath(os.environ['TCCT_LIB_PATH'])
HEADER = os.path.join(_HERE, '..', 'include', 'tcct_hip.h')

F32, BF16 = 0, 1
ACT_NONE, ACT_LRELU, ACT_HSWISH, ACT_GELU, ACT_SIGMOID, ACT_ABS = range(6)

_CT = {'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float, 'double': ctypes.c_double,
       'uint64_t': ctypes.c_uint64}


def parse_header(path=HEADER):
    """-> {name: (restype, [(ctype, argname), ...])} for every `tcct_*` prototype in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', ' ', src)
    protos = {}
    for m in re.finditer(r'(const\s+char\s*\*|int64_t|int)\s+(tcct_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        sig = []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                if '*' in a or a.startswith('tcct_stream_t'):
                    sig.append((ctypes.c_void_p, a.split()[-1].lstrip('*')))
                else:
                    ty, nm = a.rsplit(' ', 1)
                    sig.append((_CT[ty.replace('const ', '').strip()], nm))
        protos[name] = (ctypes.c_char_p if 'char' in ret else (ctypes.c_int64 if 'int64' in ret else ctypes.c_int), sig)
    return protos


class TcctError(RuntimeError):
    pass


_TLS = threading.local()
# measurement tooling (tools/attrib_trace.py): a callable (symbol, [(ctype, argname)], args-with-stream) run in front of every C-ABI call that takes a stream
_TRACE = [None]


class launch_on:
    """with launch_on(stream): every lib.* call of this thread that does not name a stream goes to `stream` (a torch.cuda.Stream).
    Cheaper than `with torch.cuda.stream(s)` (which re-binds torch's current stream twice) for blocks that only launch kernels of
    this library into pre-allocated outputs -- torch allocations inside the block would still belong to the current stream."""

    def __init__(self, stream):
        self.handle = stream.cuda_stream

    def __enter__(self):
        self.prev = getattr(_TLS, 'stream', None)
        _TLS.stream = self.handle
        return self

    def __exit__(self, *exc):
        _TLS.stream = self.prev
        return False


class _Lib:
    def __init__(self):
        self._dll = None
        self.protos = parse_header()

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise TcctError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                                '(tcct_amd has no CPU fallback)')
            dll = ctypes.CDLL(LIB_PATH)
            for name, (res, sig) in self.protos.items():
                fn = getattr(dll, name)     # AttributeError if the .so lacks a declared symbol
                fn.restype = res
                fn.argtypes = [t for t, _ in sig]
            self._dll = dll
        return self._dll

    def last_error(self):
        return self.load().tcct_last_error().decode()

    def __getattr__(self, name):
        """lib.conv2d_fwd(...) -> calls tcct_conv2d_fwd; tensors are passed as data pointers; raises on rc != 0."""
        full = 'tcct_' + name
        if full not in self.protos:
            raise AttributeError(name)
        dll = self.load()
        fn = getattr(dll, full)
        sig = self.protos[full][1]
        res_is_value = self.protos[full][0] is not ctypes.c_int

        # hot path (about 1800 launches per training step): everything that does not depend on the arguments is resolved here
        n = len(sig)
        has_stream = bool(sig) and sig[-1][1] == 'stream'
        is_ptr = tuple(ct is ctypes.c_void_p for ct, _ in sig)
        raw_stream, cur_dev, Tensor = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice, torch.Tensor

        tls, trace = _TLS, _TRACE

        def call(*args):
            if has_stream and len(args) == n - 1:
                # the caller's current HIP stream (torch.cuda.current_stream() costs 8 us), or the stream a `launch_on` block names
                args = args + (getattr(tls, 'stream', None) or raw_stream(cur_dev()),)
            elif len(args) != n:
                raise TypeError(f'{full}: expected {n} args ({[nm for _, nm in sig]}), got {len(args)}')
            if has_stream and trace[0] is not None:
                trace[0](full, sig, args)
            rc = fn(*[(a.data_ptr() if isinstance(a, Tensor) else a) if p and a is not None else a for a, p in zip(args, is_ptr)])
            if res_is_value:
                return rc
            if rc != 0:
                raise TcctError(f'{full} failed (rc={rc}): {dll.tcct_last_error().decode()}')
        call.__name__ = full
        self.__dict__[name] = call
        return call


lib = _Lib()


def dtype_code(t):
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise TcctError(f'unsupported activation dtype {t}')
