"""ctypes binding of libsimclr_hip.so (the C ABI declared in include/simclr_hip.h).

The signatures are generated from the header itself, so the header is the single
source of truth.  There is NO fallback: if the library is missing or a symbol is
absent, importing/calling fails loudly (the product path never routes around the
HIP kernels).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# SIMCLR_HIP_LIB: alternative build of the SAME library (tools/diag_conv.py uses libsimclr_hip_diag.so)
LIB_PATH = os.environ.get('SIMCLR_HIP_LIB') or os.path.join(_HERE, 'libsimclr_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'simclr_hip.h')

# bumped whenever an entry point's buffer-size contract or argument list changes (csrc/runtime.hip)
ABI_VERSION = 8

DT_F32 = 0
DT_BF16 = 1

_SCALARS = {
    'int': ctypes.c_int, 'float': ctypes.c_float, 'double': ctypes.c_double,
    'long long': ctypes.c_longlong, 'size_t': ctypes.c_size_t, 'unsigned': ctypes.c_uint,
    'simclr_stream_t': ctypes.c_void_p,
}


def _ctype(decl):
    decl = decl.strip()
    if '*' in decl:
        return ctypes.c_void_p
    decl = re.sub(r'\bconst\b', '', decl).strip()
    # drop the parameter name
    for key in sorted(_SCALARS, key=len, reverse=True):
        if decl == key or decl.startswith(key + ' '):
            return _SCALARS[key]
    raise ValueError('unknown C type in header: %r' % decl)


def parse_header(path=HEADER_PATH):
    """Return {name: (restype, [argtypes])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'^\s*#.*$', '', src, flags=re.M)
    out = {}
    for m in re.finditer(r'([A-Za-z_][\w\s\*]*?)\b(simclr_\w+)\s*\(([^;{}]*?)\)\s*;', src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith('typedef'):
            continue
        if '*' in ret:
            restype = ctypes.c_char_p if 'char' in ret else ctypes.c_void_p
        else:
            restype = _ctype(ret)
        argtypes = [] if args in ('', 'void') else [_ctype(a) for a in args.split(',')]
        out[name] = (restype, argtypes)
    return out


class SimclrHipError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise SimclrHipError(
                'libsimclr_hip.so not found at %s -- build it with '
                '`python -c "import __graft_entry__ as g; g.build()"` or simclr_amd/csrc/build.sh. '
                'There is no CPU fallback.' % LIB_PATH)
        # ONE HIP runtime per process: PyTorch ships its own libamdhip64 and owns the device context, streams and allocations this library
        # is handed.  Loaded first, its runtime is the one our DT_NEEDED entry resolves to; loaded second (library first, torch later), the
        # process ends up with two runtimes and every launch of ours fails with "no ROCm-capable device is detected" (seen when build()
        # and smoke() ran in one process, round 6).
        try:
            import torch  # noqa: F401
        except ImportError:  # pragma: no cover
            pass
        self._dll = ctypes.CDLL(LIB_PATH)
        self.signatures = parse_header()
        for name, (restype, argtypes) in self.signatures.items():
            fn = getattr(self._dll, name)   # AttributeError if the symbol is missing
            fn.restype = restype
            fn.argtypes = argtypes
        got = self._dll.simclr_abi_version()
        if got != ABI_VERSION:
            raise SimclrHipError('%s has ABI version %d, this package expects %d -- rebuild it (simclr_amd/csrc/build.sh)'
                                 % (LIB_PATH, got, ABI_VERSION))
        self._int_fns = {n for n, (r, _) in self.signatures.items() if r is ctypes.c_int}
        self._no_check = {'simclr_abi_version', 'simclr_lars_chunk_elems', 'simclr_conv2d_stats_slots',
                          'simclr_stem_stats_slots', 'simclr_bn_bwd_reduce_slots', 'simclr_bn_bwd_pool_slots',
                          'simclr_prep_chunk_elems', 'simclr_get_f32_matmul', 'simclr_conv2d_last_split_parts', 'simclr_conv2d_last_presplit',
                          'simclr_stem_wgrad_ps_supported'}

    def last_error(self):
        return self._dll.simclr_last_error().decode()

    def __getattr__(self, name):
        full = name if name.startswith('simclr_') else 'simclr_' + name
        fn = getattr(self._dll, full)
        if full in self._int_fns and full not in self._no_check:
            trace = os.environ.get('SIMCLR_TRACE_SYNC') == '1'    # debugging aid: name + device sync around every call

            def checked(*args, _fn=fn, _full=full):
                if trace:
                    import sys
                    import torch
                    sys.stderr.write('[simclr] %s %r\n' % (_full, [a if isinstance(a, (int, float)) else '*' for a in args]))
                    sys.stderr.flush()
                    rc = _fn(*args)
                    torch.cuda.synchronize()
                else:
                    rc = _fn(*args)
                if rc != 0:
                    raise SimclrHipError('%s failed (rc=%d): %s' % (_full, rc, self.last_error()))
                return rc
            setattr(self, name, checked)
            return checked
        setattr(self, name, fn)
        return fn


_lib = None


def lib():
    """The loaded library (loads on first use)."""
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
