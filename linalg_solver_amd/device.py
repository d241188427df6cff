"""Device-resident entry points: torch tensors in HBM -> liblsx *_dev functions.

torch is plumbing here (allocation, streams, torch.distributed); every
computation is a hand-written HIP kernel reached through the C ABI with the
tensor's ``data_ptr()``.  Tensors must be row-major ("contiguous" or with a
row stride >= ncols and unit column stride), dtype float64 / float32.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _native as N


def _rowmajor(t: torch.Tensor, what: str):
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        raise ValueError(f"{what}: need a row-major 2-D tensor (unit column stride)")
    if not t.is_cuda:
        raise ValueError(f"{what}: tensor must live on the GPU")


class DeviceSolver:
    """One liblsx handle bound to a torch device and (by default) torch's current stream."""

    def __init__(self, device: Optional[int] = None, use_torch_stream: bool = True):
        if not torch.cuda.is_available():
            raise N.LsxError("no GPU visible to torch; linalg_solver_amd has no CPU fallback")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.h = N.Handle(self.device)
        self.lib = self.h.lib
        if use_torch_stream:
            self.h.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    # -- helpers
    def _suffix(self, t: torch.Tensor) -> str:
        if t.dtype == torch.float64:
            return "f64"
        if t.dtype == torch.float32:
            return "f32"
        raise TypeError("dtype must be float64 or float32")

    def fill_(self, A: torch.Tensor, kind: int, seed: int, row_off: int = 0, col_off: int = 0):
        _rowmajor(A, "fill_")
        fn = getattr(self.lib, f"lsx_fill_{self._suffix(A)}_dev")
        N.check(fn(self.h.ptr, kind, seed, A.shape[0], A.shape[1], A.data_ptr(), A.stride(0), row_off, col_off))
        return A

    def getrf_(self, A: torch.Tensor, ipiv: Optional[torch.Tensor] = None, info: Optional[torch.Tensor] = None):
        """In-place P A = L U.  Returns (ipiv int32[n], info int32[1]) device tensors; no host sync."""
        _rowmajor(A, "getrf_")
        n = A.shape[0]
        if A.shape[1] != n:
            raise ValueError("getrf_ needs a square matrix")
        if ipiv is None:
            ipiv = torch.empty(max(n, 1), dtype=torch.int32, device=A.device)
        if info is None:
            info = torch.zeros(1, dtype=torch.int32, device=A.device)
        fn = getattr(self.lib, f"lsx_getrf_{self._suffix(A)}_dev")
        N.check(fn(self.h.ptr, n, A.data_ptr(), A.stride(0), ipiv.data_ptr(), info.data_ptr()), "getrf_dev")
        return ipiv, info

    def getrs_(self, LU: torch.Tensor, ipiv: torch.Tensor, B: torch.Tensor):
        """B <- A^-1 B in place (B: n x nrhs)."""
        _rowmajor(LU, "getrs_")
        _rowmajor(B, "getrs_")
        fn = getattr(self.lib, f"lsx_getrs_{self._suffix(LU)}_dev")
        N.check(fn(self.h.ptr, LU.shape[0], B.shape[1], LU.data_ptr(), LU.stride(0), ipiv.data_ptr(),
                   B.data_ptr(), B.stride(0)), "getrs_dev")
        return B

    def getri(self, LU: torch.Tensor, ipiv: torch.Tensor, out: Optional[torch.Tensor] = None):
        _rowmajor(LU, "getri")
        n = LU.shape[0]
        if out is None:
            out = torch.empty(n, n, dtype=LU.dtype, device=LU.device)
        fn = getattr(self.lib, f"lsx_getri_{self._suffix(LU)}_dev")
        N.check(fn(self.h.ptr, n, LU.data_ptr(), LU.stride(0), ipiv.data_ptr(), out.data_ptr(), out.stride(0)), "getri_dev")
        return out

    def det_parts(self, LU: torch.Tensor, ipiv: torch.Tensor) -> torch.Tensor:
        """Device tensor [sign, mant, exp2]."""
        out = torch.empty(3, dtype=torch.float64, device=LU.device)
        fn = getattr(self.lib, f"lsx_det_{self._suffix(LU)}_dev")
        N.check(fn(self.h.ptr, LU.shape[0], LU.data_ptr(), LU.stride(0), ipiv.data_ptr(), out.data_ptr()), "det_dev")
        return out

    def gesv_refined(self, A: torch.Tensor, B: torch.Tensor, sweeps: int = 3):
        """Mixed-precision solve on tensors in HBM: A (n x n fp32, kept), B (n x nrhs fp32, kept).  Returns
        (X fp64, X fp32, LU fp32, ipiv, info, stats) -- stats = [max|d|, max|x|] of the last sweep, then of the
        unrefined solve (device tensor, 4 doubles).  lsx_gesv_f32_refined_dev; asynchronous."""
        _rowmajor(A, "gesv_refined")
        _rowmajor(B, "gesv_refined")
        if A.dtype != torch.float32 or B.dtype != torch.float32:
            raise TypeError("gesv_refined takes fp32 operands")
        n, nrhs = A.shape[0], B.shape[1]
        LU = torch.empty(n, n, dtype=torch.float32, device=A.device)
        X64 = torch.empty(n, nrhs, dtype=torch.float64, device=A.device)
        X32 = torch.empty(n, nrhs, dtype=torch.float32, device=A.device)
        ipiv = torch.empty(max(n, 1), dtype=torch.int32, device=A.device)
        info = torch.zeros(1, dtype=torch.int32, device=A.device)
        stats = torch.zeros(4, dtype=torch.float64, device=A.device)
        N.check(self.lib.lsx_gesv_f32_refined_dev(self.h.ptr, n, nrhs, A.data_ptr(), A.stride(0), LU.data_ptr(), n,
                                                  ipiv.data_ptr(), info.data_ptr(), B.data_ptr(), B.stride(0),
                                                  X64.data_ptr(), nrhs, X32.data_ptr(), nrhs, int(sweeps),
                                                  stats.data_ptr()), "gesv_refined_dev")
        return X64, X32, LU, ipiv, info, stats

    def gemm_sub_(self, C: torch.Tensor, A: torch.Tensor, B: torch.Tensor):
        """C -= A @ B on the MFMA kernel."""
        for t, w in ((C, "C"), (A, "A"), (B, "B")):
            _rowmajor(t, "gemm_sub_ " + w)
        m, k = A.shape
        n = B.shape[1]
        assert B.shape[0] == k and C.shape == (m, n)
        fn = getattr(self.lib, f"lsx_gemm_sub_{self._suffix(C)}_dev")
        N.check(fn(self.h.ptr, m, n, k, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(),
                   C.stride(0)), "gemm_sub_dev")
        return C

    def gemm_add_(self, C: torch.Tensor, A: torch.Tensor, B: torch.Tensor):
        """C += A @ B (fp64) on the same kernel."""
        for t, w in ((C, "C"), (A, "A"), (B, "B")):
            _rowmajor(t, "gemm_add_ " + w)
        m, k = A.shape
        n = B.shape[1]
        assert B.shape[0] == k and C.shape == (m, n) and C.dtype == torch.float64
        N.check(self.lib.lsx_gemm_add_f64_dev(self.h.ptr, m, n, k, A.data_ptr(), A.stride(0), B.data_ptr(),
                                              B.stride(0), C.data_ptr(), C.stride(0)), "gemm_add_dev")
        return C

    def panel_(self, P: torch.Tensor, row0: int, ipiv: torch.Tensor, info: torch.Tensor):
        _rowmajor(P, "panel_")
        N.check(self.lib.lsx_panel_f64_dev(self.h.ptr, P.shape[0], P.shape[1], P.data_ptr(), P.stride(0), row0,
                                           ipiv.data_ptr(), info.data_ptr()), "panel_dev")

    def laswp_(self, A: torch.Tensor, row0: int, jb: int, ipiv: torch.Tensor):
        _rowmajor(A, "laswp_")
        N.check(self.lib.lsx_laswp_f64_dev(self.h.ptr, A.shape[1], A.data_ptr(), A.stride(0), row0, jb,
                                           ipiv.data_ptr()), "laswp_dev")

    def panel_moves_(self, moves: torch.Tensor) -> bool:
        """Copy the gather list of the last panel_ call into `moves` (int32[512]); False if none."""
        import ctypes as C
        ok = C.c_int(0)
        N.check(self.lib.lsx_panel_moves_dev(self.h.ptr, moves.data_ptr(), C.byref(ok)), "panel_moves_dev")
        return bool(ok.value)

    def laswp_moves_(self, A: torch.Tensor, row0: int, moves: torch.Tensor):
        _rowmajor(A, "laswp_moves_")
        N.check(self.lib.lsx_laswp_moves_f64_dev(self.h.ptr, A.shape[1], A.data_ptr(), A.stride(0), row0,
                                                 moves.data_ptr()), "laswp_moves_dev")

    def trsm_lu_(self, L: torch.Tensor, B: torch.Tensor):
        _rowmajor(L, "trsm_lu_")
        _rowmajor(B, "trsm_lu_")
        N.check(self.lib.lsx_trsm_lu_f64_dev(self.h.ptr, L.shape[0], B.shape[1], L.data_ptr(), L.stride(0),
                                             B.data_ptr(), B.stride(0)), "trsm_lu_dev")

    def rref_(self, R: torch.Tensor, bar_col: int = 0, tol: float = -1.0, pivot_rule: int = N.PIVOT_FIRST):
        _rowmajor(R, "rref_")
        m, n = R.shape
        piv = torch.zeros(2 * min(m, n), dtype=torch.int32, device=R.device)
        rank = torch.zeros(1, dtype=torch.int32, device=R.device)
        N.check(self.lib.lsx_rref_f64_dev(self.h.ptr, m, n, bar_col, R.data_ptr(), R.stride(0), piv.data_ptr(),
                                          rank.data_ptr(), tol, pivot_rule), "rref_dev")
        return piv, rank
