"""ctypes loader for the C oracle (oracle/rowreduce_ref.c, oracle/lu_twin.c).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

KINDS = ("S", "N", "E", "U")  # ORC_SWAP, ORC_NORM, ORC_BELOW, ORC_ABOVE


class _Step(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32)]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("rowreduce_ref.c", "lu_twin.c", "Makefile")]
    stale = force or not os.path.exists(_SO) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs
    )
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        L.orc_row_reduce_f64.restype = C.c_int
        L.orc_row_reduce_f64.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_int, ip,
                                         C.POINTER(_Step), C.c_int, C.POINTER(C.c_int)]
        L.orc_inconsistent_row_f64.restype = C.c_int
        L.orc_inconsistent_row_f64.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_left_is_identity_f64.restype = C.c_int
        L.orc_left_is_identity_f64.argtypes = [dp, C.c_int, C.c_int]
        L.orc_getrf_f64.restype = C.c_int
        L.orc_getrf_f64.argtypes = [dp, C.c_int, C.c_int, ip]
        L.orc_getrs_f64.restype = None
        L.orc_getrs_f64.argtypes = [dp, C.c_int, C.c_int, ip, dp, C.c_int, C.c_int]
        L.orc_slogdet_f64.restype = None
        L.orc_slogdet_f64.argtypes = [dp, C.c_int, C.c_int, ip, dp, dp]
        _lib = L
    return _lib


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def row_reduce(A: np.ndarray, bar_col: Optional[int] = None, want_steps: bool = True):
    """fp64 restatement of Matrix.row_reduce -> (reduced, pivots, steps)."""
    R = np.array(A, dtype=np.float64, order="C", copy=True)
    m, n = R.shape
    piv = np.zeros(2 * max(1, min(m, n)), dtype=np.int32)
    cap = 3 * (m + n) + 8 if want_steps else 0
    steps = (_Step * max(cap, 1))()
    ns = C.c_int(0)
    np_ = lib().orc_row_reduce_f64(_dp(R), m, n, n, int(bar_col or 0), _ip(piv),
                                   steps if want_steps else None, cap, C.byref(ns))
    pivots: List[Tuple[int, int]] = [(int(piv[2 * i]), int(piv[2 * i + 1])) for i in range(np_)]
    out_steps = []
    if want_steps:
        assert ns.value <= cap, "step buffer too small"
        out_steps = [(KINDS[steps[i].kind], i, int(steps[i].a), int(steps[i].b))
                     for i in range(ns.value)]
    return R, pivots, out_steps


def inconsistent_row(R: np.ndarray, nvars: int, bar_col: int) -> int:
    R = np.ascontiguousarray(R, dtype=np.float64)
    return lib().orc_inconsistent_row_f64(_dp(R), R.shape[0], R.shape[1], nvars, bar_col)


def left_is_identity(R: np.ndarray, n: int) -> bool:
    R = np.ascontiguousarray(R, dtype=np.float64)
    return bool(lib().orc_left_is_identity_f64(_dp(R), n, R.shape[1]))


def getrf(A: np.ndarray):
    """Partial-pivot LU twin -> (LU, ipiv, info)."""
    LU = np.array(A, dtype=np.float64, order="C", copy=True)
    n = LU.shape[0]
    assert LU.shape == (n, n)
    ipiv = np.zeros(n, dtype=np.int32)
    info = lib().orc_getrf_f64(_dp(LU), n, n, _ip(ipiv))
    return LU, ipiv, info


def getrs(LU: np.ndarray, ipiv: np.ndarray, B: np.ndarray) -> np.ndarray:
    X = np.array(B, dtype=np.float64, order="C", copy=True)
    if X.ndim == 1:
        X2 = X.reshape(-1, 1).copy()
        getrs_inplace(LU, ipiv, X2)
        return X2[:, 0].copy()
    getrs_inplace(LU, ipiv, X)
    return X


def getrs_inplace(LU, ipiv, X):
    LU = np.ascontiguousarray(LU, dtype=np.float64)
    ipiv = np.ascontiguousarray(ipiv, dtype=np.int32)
    n = LU.shape[0]
    lib().orc_getrs_f64(_dp(LU), n, n, _ip(ipiv), _dp(X), X.shape[1], X.shape[1])


def slogdet(LU: np.ndarray, ipiv: np.ndarray):
    LU = np.ascontiguousarray(LU, dtype=np.float64)
    ipiv = np.ascontiguousarray(ipiv, dtype=np.int32)
    s, l = C.c_double(0), C.c_double(0)
    lib().orc_slogdet_f64(_dp(LU), LU.shape[0], LU.shape[0], _ip(ipiv), C.byref(s), C.byref(l))
    return s.value, l.value
