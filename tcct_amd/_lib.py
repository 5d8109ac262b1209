"""ctypes binding of libtcct_hip.so.  The argtypes are parsed from include/tcct_hip.h so the Python side can never
drift from the C-ABI.  There is NO fallback: if the library is missing or a call fails, this raises."""
import ctypes
import os
import re

import torch  # noqa: F401  (must be imported first: the .so then binds to torch's already-mapped HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libtcct_hip.so')
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

        def call(*args):
            if len(args) == len(sig) - 1 and sig and sig[-1][1] == 'stream':
                args = args + (torch.cuda.current_stream().cuda_stream,)
            if len(args) != len(sig):
                raise TypeError(f'{full}: expected {len(sig)} args ({[n for _, n in sig]}), got {len(args)}')
            conv = []
            for a, (ct, nm) in zip(args, sig):
                if ct is ctypes.c_void_p:
                    if a is None:
                        conv.append(None)
                    elif isinstance(a, torch.Tensor):
                        conv.append(a.data_ptr())
                    else:
                        conv.append(int(a))
                else:
                    conv.append(a)
            rc = fn(*conv)
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
