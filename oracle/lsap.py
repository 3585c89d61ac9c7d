"""ctypes wrapper over oracle/lsap.c (TEST INFRA).  `build()` compiles it with gcc into oracle/_build/."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblsap_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "lsap.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _SO, src, "-lm"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.cdetr_oracle_lsap.restype = ctypes.c_int
        _lib.cdetr_oracle_lsap.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p]
    return _lib


def linear_sum_assignment(cost):
    """Same contract as scipy.optimize.linear_sum_assignment (minimisation): (row_ind, col_ind) int64."""
    c = np.ascontiguousarray(np.asarray(cost, dtype=np.float64))
    if c.ndim != 2:
        raise ValueError("expected a matrix")
    nr, nc = c.shape
    n = min(nr, nc)
    a = np.zeros(n, dtype=np.int64)
    b = np.zeros(n, dtype=np.int64)
    rc = _load().cdetr_oracle_lsap(nr, nc, c.ctypes.data, a.ctypes.data, b.ctypes.data)
    if rc == -2:
        raise ValueError("matrix contains invalid numeric entries")
    if rc == -1:
        raise ValueError("cost matrix is infeasible")
    return a, b
