"""ctypes binding of libfwdtaco_hip.so.

The argument types are derived from include/fwdtaco_hip.h itself, so the Python side can never drift
from the C ABI.  There is NO fallback: if the shared library is missing or a symbol is absent the
import of any op fails loudly.
"""
import ctypes
import os
import re
from typing import Dict, List, Tuple

# torch must be imported BEFORE libfwdtaco_hip.so is dlopen'ed: PyTorch-ROCm bundles its own libamdhip64 /
# libhsa-runtime64 (same SONAME as /opt/rocm's).  Loading ours first pulls in a second HIP runtime that owns
# no device ("no ROCm-capable device is detected"); loading torch first makes both share one runtime, one
# device context and the same streams.
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
# FT_LIB: another build of the same ABI (same-box A/B of kernel variants, lab/ only)
LIB_PATH = os.environ.get('FT_LIB') or os.path.join(HERE, 'libfwdtaco_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(HERE), 'include', 'fwdtaco_hip.h')

_SCALARS = {
    'int': ctypes.c_int, 'long': ctypes.c_long, 'size_t': ctypes.c_size_t, 'float': ctypes.c_float,
    'double': ctypes.c_double, 'int64_t': ctypes.c_int64, 'uint64_t': ctypes.c_uint64,
}
_RET = {'int': ctypes.c_int, 'size_t': ctypes.c_size_t, 'const char*': ctypes.c_char_p, 'void': None}


def parse_header(path: str = HEADER_PATH) -> Dict[str, Tuple[str, List[Tuple[str, str]]]]:
    """Returns {symbol: (return type, [(arg type, arg name), ...])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = re.sub(r'//[^\n]*', '', src)
    src = re.sub(r'^\s*#.*$', '', src, flags=re.M)
    out = {}
    for m in re.finditer(r'(const char\*|size_t|int|void)\s+(ft_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), ' '.join(m.group(3).split())
        alist = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                mm = re.match(r'^(.*?)(\w+)$', a)
                alist.append((mm.group(1).strip(), mm.group(2)))
        out[name] = (ret, alist)
    return out


def _ctype(t: str):
    t = t.replace(' *', '*').strip()
    if t.endswith('*'):
        return ctypes.c_void_p
    return _SCALARS[t]


_lib = None
PROTOS = None


def lib() -> ctypes.CDLL:
    global _lib, PROTOS
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: build it with `python -m forwardtacotron_amd.build` '
            '(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the hot path.')
    L = ctypes.CDLL(LIB_PATH)
    PROTOS = parse_header()
    for name, (ret, args) in PROTOS.items():
        if not hasattr(L, name):
            raise RuntimeError(f'libfwdtaco_hip.so does not export {name} (declared in fwdtaco_hip.h)')
        f = getattr(L, name)
        f.restype = _RET[ret]
        f.argtypes = [_ctype(t) for t, _ in args]
    if L.ft_abi_version() != int(re.search(r'#define FWDTACO_ABI_VERSION (\d+)', open(HEADER_PATH).read()).group(1)):
        raise RuntimeError('libfwdtaco_hip.so ABI version does not match include/fwdtaco_hip.h; rebuild')
    _lib = L
    return L


class FtError(RuntimeError):
    pass


def check(rc: int, what: str = '') -> None:
    if rc != 0:
        raise FtError(f'{what}: {lib().ft_last_error().decode()}')


def call(name: str, *args):
    """Call an int-returning entry point and raise on error."""
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise FtError(f'{name}: {lib().ft_last_error().decode()}')


_ws_cache = {}


def query(name: str, *args):
    """Call a value-returning entry point.  `*_workspace` size queries are pure functions of their integer arguments
    and sit in front of most launches, so their answers are memoised."""
    if name.endswith('_workspace'):
        key = (name, args)
        v = _ws_cache.get(key)
        if v is None:
            v = getattr(lib(), name)(*args)
            _ws_cache[key] = v
        return v
    return getattr(lib(), name)(*args)
