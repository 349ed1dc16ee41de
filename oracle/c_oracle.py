"""ctypes wrapper of the plain-C oracle (oracle/c/isdf_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import numpy as np

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'c')
_lib = None


def _load():
    global _lib
    if _lib is None:
        has_fma = False
        try:
            with open('/proc/cpuinfo') as f:
                has_fma = ' fma ' in f.read()
        except OSError:
            pass
        name = 'liboracle_fma.so' if has_fma else 'liboracle_portable.so'
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            raise RuntimeError('%s missing: run `make -C oracle/c` (or __graft_entry__.build())' % path)
        _lib = ctypes.CDLL(path)
        _lib.oracle_select_ip.restype = ctypes.c_long
        _lib.oracle_select_ip.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_long, ctypes.c_int,
                                          ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
        _lib.oracle_select_ip_cplx.restype = ctypes.c_long
        _lib.oracle_select_ip_cplx.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_long,
                                               ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_long]
    return _lib


def select_ip_cplx(X, k, tol=-1.0, tie_rtol=1e-10):
    """Complex mode: X (2*nh, m) = [Re u; Im u]."""
    lib = _load()
    X = np.ascontiguousarray(X, dtype=np.float64)
    nao, m = X.shape
    k = int(min(k, m))
    piv = np.zeros(max(k, 1), dtype=np.int64)
    L = np.zeros((max(k, 1), m))
    rank = lib.oracle_select_ip_cplx(X.ctypes.data, nao, nao // 2, m, m, k, float(tol), float(tie_rtol),
                                     piv.ctypes.data, L.ctypes.data, m)
    return piv[:rank], L[:rank]


def select_ip(aoT, k, tol=-1.0, tie_rtol=1e-10):
    """Bit-exact counterpart of isdf_select_ip for one block.  Returns (piv[rank], L[rank, m])."""
    lib = _load()
    aoT = np.ascontiguousarray(aoT, dtype=np.float64)
    nao, m = aoT.shape
    k = int(min(k, m))
    piv = np.zeros(max(k, 1), dtype=np.int64)
    L = np.zeros((max(k, 1), m))
    rank = lib.oracle_select_ip(aoT.ctypes.data, nao, m, m, k, float(tol), float(tie_rtol), piv.ctypes.data,
                                L.ctypes.data, m)
    return piv[:rank], L[:rank]
