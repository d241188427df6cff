"""numpy-level entry points over the C ABI (contiguous fp64 / fp32 buffers).

This is the marshalling-free boundary: `Matrix` (matrix.py) converts its
list-of-lists into one contiguous array and calls these.  Every function runs
on the GPU through liblsx.so; none has a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Tuple

import numpy as np

from . import _native as N

EPS64 = float(np.finfo(np.float64).eps)
EPS32 = float(np.finfo(np.float32).eps)


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a: np.ndarray, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _h(handle: Optional[N.Handle]) -> N.Handle:
    return handle if handle is not None else N.default_handle()


def lu_factor(a, handle: Optional[N.Handle] = None, dtype=np.float64) -> Tuple[np.ndarray, np.ndarray, int]:
    """P A = L U (partial pivoting).  Returns (LU, ipiv, info); ipiv is 0-based LAPACK style."""
    h = _h(handle)
    LU = np.array(a, dtype=dtype, order="C", copy=True)
    n = LU.shape[0]
    if LU.ndim != 2 or LU.shape[1] != n:
        raise ValueError("lu_factor needs a square matrix")
    ipiv = np.zeros(max(n, 1), dtype=np.int32)
    info = C.c_int(0)
    if dtype == np.float64:
        N.check(h.lib.lsx_getrf_f64(h.ptr, n, _ptr(LU, C.c_double), n, _ptr(ipiv, C.c_int32), C.byref(info)), "lsx_getrf_f64")
    elif dtype == np.float32:
        N.check(h.lib.lsx_getrf_f32(h.ptr, n, _ptr(LU, C.c_float), n, _ptr(ipiv, C.c_int32), C.byref(info)), "lsx_getrf_f32")
    else:
        raise TypeError("dtype must be float64 or float32")
    return LU, ipiv[:n], info.value


def lu_solve(LU: np.ndarray, ipiv: np.ndarray, b, handle: Optional[N.Handle] = None) -> np.ndarray:
    h = _h(handle)
    dt = LU.dtype
    LU = np.ascontiguousarray(LU)
    n = LU.shape[0]
    X = np.array(b, dtype=dt, order="C", copy=True)
    vec = X.ndim == 1
    if vec:
        X = X.reshape(n, 1).copy()
    if X.shape[0] != n:
        raise ValueError("right-hand side has the wrong number of rows")
    ipiv = np.ascontiguousarray(ipiv, dtype=np.int32)
    if dt == np.float64:
        N.check(h.lib.lsx_getrs_f64(h.ptr, n, X.shape[1], _ptr(LU, C.c_double), n, _ptr(ipiv, C.c_int32),
                                    _ptr(X, C.c_double), X.shape[1]), "lsx_getrs_f64")
    else:
        N.check(h.lib.lsx_getrs_f32(h.ptr, n, X.shape[1], _ptr(LU, C.c_float), n, _ptr(ipiv, C.c_int32),
                                    _ptr(X, C.c_float), X.shape[1]), "lsx_getrs_f32")
    return X[:, 0].copy() if vec else X


def solve(a, b, handle: Optional[N.Handle] = None, dtype=np.float64):
    """Solve A X = B.  Returns (X, info, pivot_ratio); X is None when info != 0."""
    h = _h(handle)
    A = np.ascontiguousarray(a, dtype=dtype)
    n = A.shape[0]
    if A.ndim != 2 or A.shape[1] != n:
        raise ValueError("solve needs a square matrix")
    X = np.array(b, dtype=dtype, order="C", copy=True)
    vec = X.ndim == 1
    if vec:
        X = X.reshape(n, 1).copy()
    if X.shape[0] != n:
        raise ValueError("right-hand side has the wrong number of rows")
    info, ratio = C.c_int(0), C.c_double(1.0)
    if dtype == np.float64:
        N.check(h.lib.lsx_gesv_f64(h.ptr, n, X.shape[1], _ptr(A, C.c_double), n, _ptr(X, C.c_double), X.shape[1],
                                   C.byref(info), C.byref(ratio)), "lsx_gesv_f64")
    else:
        N.check(h.lib.lsx_gesv_f32(h.ptr, n, X.shape[1], _ptr(A, C.c_float), n, _ptr(X, C.c_float), X.shape[1],
                                   C.byref(info), C.byref(ratio)), "lsx_gesv_f32")
    if info.value != 0:
        return None, info.value, ratio.value
    return (X[:, 0].copy() if vec else X), 0, ratio.value


def _ct(dtype):
    if dtype == np.float64:
        return C.c_double, "f64"
    if dtype == np.float32:
        return C.c_float, "f32"
    raise TypeError("dtype must be float64 or float32")


def solve_refined(a, b, sweeps: int = 3, handle: Optional[N.Handle] = None):
    """Mixed-precision solve of A X = B (lsx_gesv_f32_refined): fp32 factors on the fp32 MFMA tile, residuals in
    fp64, `sweeps` corrections.  a and b are taken in fp32 (BASELINE config 5 computes in fp32); returns
    (X as float64, info, pivot_ratio, last_correction) -- X is None when info != 0.  The forward error against
    the fp64 solution of the same system is far below the 1e-4 the configuration asks for, which a plain fp32
    solve of a large random system does not reach (cond * eps32)."""
    h = _h(handle)
    A = np.ascontiguousarray(a, dtype=np.float32)
    n = A.shape[0]
    if A.ndim != 2 or A.shape[1] != n:
        raise ValueError("solve needs a square matrix")
    B = np.array(b, dtype=np.float32, order="C", copy=True)
    vec = B.ndim == 1
    if vec:
        B = B.reshape(n, 1).copy()
    if B.shape[0] != n:
        raise ValueError("right-hand side has the wrong number of rows")
    nrhs = B.shape[1]
    X = np.zeros((n, nrhs), dtype=np.float64)
    info, ratio, corr = C.c_int(0), C.c_double(1.0), C.c_double(0.0)
    N.check(h.lib.lsx_gesv_f32_refined(h.ptr, n, nrhs, _ptr(A, C.c_float), n, _ptr(B, C.c_float), nrhs, None, nrhs,
                                       _ptr(X, C.c_double), nrhs, int(sweeps), C.byref(info), C.byref(ratio),
                                       C.byref(corr)), "lsx_gesv_f32_refined")
    if info.value != 0:
        return None, info.value, ratio.value, corr.value
    return (X[:, 0].copy() if vec else X), 0, ratio.value, corr.value


def inv(a, handle: Optional[N.Handle] = None, dtype=np.float64):
    """Returns (inverse or None, info, pivot_ratio)."""
    h = _h(handle)
    ct, sfx = _ct(dtype)
    A = np.ascontiguousarray(a, dtype=dtype)
    n = A.shape[0]
    if A.ndim != 2 or A.shape[1] != n:
        raise ValueError("inv needs a square matrix")
    out = np.empty_like(A)
    info, ratio = C.c_int(0), C.c_double(1.0)
    N.check(getattr(h.lib, f"lsx_getri_{sfx}")(h.ptr, n, _ptr(A, ct), n, _ptr(out, ct), n, C.byref(info),
                                                C.byref(ratio)), f"lsx_getri_{sfx}")
    if info.value != 0:
        return None, info.value, ratio.value
    return out, 0, ratio.value


def det_parts(a, handle: Optional[N.Handle] = None, dtype=np.float64) -> Tuple[float, float, int]:
    """det(A) = sign * mant * 2**exp2 with mant in [0.5, 1) (sign 0 for a singular matrix)."""
    h = _h(handle)
    ct, sfx = _ct(dtype)
    A = np.ascontiguousarray(a, dtype=dtype)
    n = A.shape[0]
    if A.ndim != 2 or A.shape[1] != n:
        raise ValueError("det needs a square matrix")
    s, m, e = C.c_double(0), C.c_double(0), C.c_int64(0)
    N.check(getattr(h.lib, f"lsx_det_{sfx}")(h.ptr, n, _ptr(A, ct), n, C.byref(s), C.byref(m), C.byref(e)), f"lsx_det_{sfx}")
    return s.value, m.value, int(e.value)


def slogdet(a, handle: Optional[N.Handle] = None, dtype=np.float64) -> Tuple[float, float]:
    s, m, e = det_parts(a, handle, dtype)
    if s == 0.0:
        return 0.0, -math.inf
    return s, math.log(m) + e * math.log(2.0)


def det(a, handle: Optional[N.Handle] = None) -> float:
    s, m, e = det_parts(a, handle)
    if s == 0.0:
        return 0.0
    try:
        return s * math.ldexp(m, e)
    except OverflowError:
        return s * math.inf


def matmul(a, b, handle: Optional[N.Handle] = None) -> np.ndarray:
    """C = A @ B on the MFMA tile of the trailing update (residual checks, Matrix.__mul__)."""
    h = _h(handle)
    A, B = _f64(a), _f64(b)
    if A.ndim != 2 or B.ndim != 2 or A.shape[1] != B.shape[0]:
        raise ValueError("Matrix dimensions must match")
    m, k = A.shape
    n = B.shape[1]
    Cm = np.empty((m, n), dtype=np.float64)
    if m and n:
        N.check(h.lib.lsx_matmul_f64(h.ptr, m, n, k, _ptr(A, C.c_double), max(k, 1), _ptr(B, C.c_double), max(n, 1),
                                     _ptr(Cm, C.c_double), n), "lsx_matmul_f64")
    return Cm


def rref(a, bar_col: Optional[int] = None, tol: float = -1.0, handle: Optional[N.Handle] = None,
         pivot_rule: int = N.PIVOT_FIRST, dtype=np.float64):
    """Reduced row echelon form over columns [0, bar_col).  Returns (R, pivots, rank).
    pivot_rule: N.PIVOT_FIRST = the reference's first-non-zero rule, N.PIVOT_MAX = largest |a|."""
    h = _h(handle)
    ct, sfx = _ct(dtype)
    A = np.ascontiguousarray(a, dtype=dtype)
    if A.ndim != 2 or A.shape[0] < 1 or A.shape[1] < 1:
        raise ValueError("rref needs a non-empty 2-D matrix")
    m, n = A.shape
    R = np.empty_like(A)
    piv = np.zeros(2 * min(m, n), dtype=np.int32)
    rank = C.c_int(0)
    N.check(getattr(h.lib, f"lsx_rref_{sfx}")(h.ptr, m, n, int(bar_col or 0), _ptr(A, ct), n, _ptr(R, ct), n,
                                               _ptr(piv, C.c_int32), C.byref(rank), float(tol), int(pivot_rule)),
            f"lsx_rref_{sfx}")
    r = rank.value
    return R, [(int(piv[2 * i]), int(piv[2 * i + 1])) for i in range(r)], r


TRACE_KINDS = ("S", "N", "E", "E")   # step record kind -> label letter (swap, normalise, eliminate below / above)


def rref_trace(a, bar_col: Optional[int] = None, max_snapshots: int = 0, int_mask=None,
               handle: Optional[N.Handle] = None):
    """Row reduction in the reference's own operation order with its step log (lsx_rref_trace_f64;
    linalg.py:547-629).  Returns (R, pivots, steps, int_mask, snaps, snap_int_masks): steps is a list of
    (kind, a, b) with kind 0 = swap rows a,b / 1 = normalise row a / 2 = eliminate below the pivot of
    column a / 3 = eliminate above the pivot of column a (1-based, as in the reference's descriptions).
    int_mask (bool m x n, default all False) says which input entries are Python ints; the returned mask
    says which entries the reference would still hold as ints.  snaps / snap_int_masks hold the matrix
    and its mask after each of the first max_snapshots steps (or None)."""
    h = _h(handle)
    A = _f64(a)
    if A.ndim != 2 or A.shape[0] < 1 or A.shape[1] < 1:
        raise ValueError("rref_trace needs a non-empty 2-D matrix")
    m, n = A.shape
    bar = int(bar_col or 0) or n - 1
    max_steps = 4 * min(m, max(bar, 1)) + 4
    R = np.empty_like(A)
    mask = np.zeros((m, n), dtype=np.uint8)
    if int_mask is not None:
        mask[:, :] = np.asarray(int_mask, dtype=bool)
    piv = np.zeros(2 * min(m, n) + 2, dtype=np.int32)
    steps = np.zeros(4 * max_steps, dtype=np.int32)
    nsnap = max(0, min(int(max_snapshots), max_steps))
    snaps = np.empty((nsnap, m, n), dtype=np.float64) if nsnap else None
    snap_m = np.zeros((nsnap, m, n), dtype=np.uint8) if nsnap else None
    npiv, nsteps = C.c_int(0), C.c_int(0)
    N.check(h.lib.lsx_rref_trace_f64(h.ptr, m, n, int(bar_col or 0), _ptr(A, C.c_double), n, _ptr(R, C.c_double), n,
                                     mask.ctypes.data, _ptr(piv, C.c_int32), C.byref(npiv),
                                     _ptr(steps, C.c_int32), max_steps, C.byref(nsteps),
                                     _ptr(snaps, C.c_double) if nsnap else None,
                                     snap_m.ctypes.data if nsnap else None, nsnap), "lsx_rref_trace_f64")
    pivots = [(int(piv[2 * i]), int(piv[2 * i + 1])) for i in range(npiv.value)]
    recs = [(int(steps[4 * i]), int(steps[4 * i + 1]), int(steps[4 * i + 2])) for i in range(nsteps.value)]
    k = min(nsnap, nsteps.value)
    return R, pivots, recs, mask.astype(bool), (snaps[:k] if nsnap else None), (snap_m[:k].astype(bool) if nsnap else None)
