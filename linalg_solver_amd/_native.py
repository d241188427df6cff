"""ctypes binding of liblsx.so (include/lsx.h).

The library is the only compute path of this package: if it cannot be loaded,
or no gfx950 device is visible, every entry point raises -- there is no CPU
fallback (a silent one would make parity claims meaningless).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LSX_LIB_OVERRIDE") or os.path.join(_HERE, "liblsx.so")   # override: timing experiments (tools/)

PROF_BUCKETS = {"panel": 0, "laswp": 1, "trsm": 2, "gemm": 3, "other": 4, "gemm_skinny": 5}
FILL_INT5, FILL_U11 = 0, 1
PIVOT_FIRST, PIVOT_MAX = 0, 1


class LsxError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()

_dp, _fp, _ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32)
_vp, _i, _u64 = C.c_void_p, C.c_int, C.c_uint64

# name -> argtypes; every function returns int status except the two noted below.
# Device-pointer entry points take raw addresses (c_void_p) so torch data_ptr() fits.
_SIGS = {
    "lsx_create": [C.POINTER(_vp), _i],
    "lsx_destroy": [_vp],
    "lsx_set_stream": [_vp, _vp],
    "lsx_use_own_stream": [_vp],
    "lsx_synchronize": [_vp],
    "lsx_set_option": [_vp, C.c_char_p, _i],
    "lsx_check_status": [_vp],
    "lsx_get_option": [_vp, C.c_char_p, C.POINTER(_i)],
    "lsx_getrf_f64": [_vp, _i, _dp, _i, _ip, C.POINTER(_i)],
    "lsx_getrs_f64": [_vp, _i, _i, _dp, _i, _ip, _dp, _i],
    "lsx_gesv_f64": [_vp, _i, _i, _dp, _i, _dp, _i, C.POINTER(_i), _dp],
    "lsx_getri_f64": [_vp, _i, _dp, _i, _dp, _i, C.POINTER(_i), _dp],
    "lsx_det_f64": [_vp, _i, _dp, _i, _dp, _dp, C.POINTER(C.c_int64)],
    "lsx_rref_f64": [_vp, _i, _i, _i, _dp, _i, _dp, _i, _ip, C.POINTER(_i), C.c_double, _i],
    "lsx_rref_trace_f64": [_vp, _i, _i, _i, _dp, _i, _dp, _i, C.c_void_p, _ip, C.POINTER(_i), _ip, _i,
                           C.POINTER(_i), _dp, C.c_void_p, _i],
    "lsx_getri_f32": [_vp, _i, _fp, _i, _fp, _i, C.POINTER(_i), _dp],
    "lsx_det_f32": [_vp, _i, _fp, _i, _dp, _dp, C.POINTER(C.c_int64)],
    "lsx_rref_f32": [_vp, _i, _i, _i, _fp, _i, _fp, _i, _ip, C.POINTER(_i), C.c_double, _i],
    "lsx_gesv_f32_refined": [_vp, _i, _i, _fp, _i, _fp, _i, _fp, _i, _dp, _i, _i, C.POINTER(_i), _dp, _dp],
    "lsx_getri_f32_dev": [_vp, _i, _vp, _i, _vp, _vp, _i],
    "lsx_det_f32_dev": [_vp, _i, _vp, _i, _vp, _vp],
    "lsx_gesv_f32_refined_dev": [_vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _vp],
    "lsx_getrf_f32": [_vp, _i, _fp, _i, _ip, C.POINTER(_i)],
    "lsx_getrs_f32": [_vp, _i, _i, _fp, _i, _ip, _fp, _i],
    "lsx_gesv_f32": [_vp, _i, _i, _fp, _i, _fp, _i, C.POINTER(_i), _dp],
    "lsx_getrf_f64_dev": [_vp, _i, _vp, _i, _vp, _vp],
    "lsx_getrs_f64_dev": [_vp, _i, _i, _vp, _i, _vp, _vp, _i],
    "lsx_getri_f64_dev": [_vp, _i, _vp, _i, _vp, _vp, _i],
    "lsx_det_f64_dev": [_vp, _i, _vp, _i, _vp, _vp],
    "lsx_getrf_f32_dev": [_vp, _i, _vp, _i, _vp, _vp],
    "lsx_getrs_f32_dev": [_vp, _i, _i, _vp, _i, _vp, _vp, _i],
    "lsx_rref_f64_dev": [_vp, _i, _i, _i, _vp, _i, _vp, _vp, C.c_double, _i],
    "lsx_panel_f64_dev": [_vp, _i, _i, _vp, _i, _i, _vp, _vp],
    "lsx_laswp_f64_dev": [_vp, _i, _vp, _i, _i, _i, _vp],
    "lsx_trsm_lu_f64_dev": [_vp, _i, _i, _vp, _i, _vp, _i],
    "lsx_panel_moves_dev": [_vp, _vp, C.POINTER(_i)],
    "lsx_laswp_moves_f64_dev": [_vp, _i, _vp, _i, _i, _vp],
    "lsx_gemm_sub_f64_dev": [_vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i],
    "lsx_gemm_sub_f32_dev": [_vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i],
    "lsx_gemm_add_f64_dev": [_vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i],
    "lsx_matmul_f64": [_vp, _i, _i, _i, _dp, _i, _dp, _i, _dp, _i],
    "lsx_fill_f64_dev": [_vp, _i, _u64, _i, _i, _vp, _i, _i, _i],
    "lsx_fill_f32_dev": [_vp, _i, _u64, _i, _i, _vp, _i, _i, _i],
    "lsx_diag_mfma_peak": [_vp, _i, _i, _i, _dp, _dp],
    "lsx_diag_read_scratch": [_vp, C.c_size_t, _vp, C.c_size_t],
    "lsx_diag_xchg_probe": [_vp, _i, _i, _i, _i, _i, _dp, _ip, _ip],
    "lsx_diag_chain_head_f32": [_vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp],
    "lsx_diag_chain_head_f64": [_vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp],
    "lsx_getrf_mg_f64": [C.POINTER(_vp), _i, _i, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_vp), C.POINTER(_vp)],
    "lsx_diag_occupy": [_vp, _i, _i, _i],
    "lsx_diag_cu_mask_probe": [_vp, C.POINTER(C.c_uint32), _i, _i, C.POINTER(C.c_uint32)],
    "lsx_prof_enable": [_vp, _i],
    "lsx_prof_reset": [_vp],
    "lsx_prof_read": [_vp, _i, _dp, C.POINTER(C.c_longlong), _dp, _dp],
}
EXPORTS = sorted(list(_SIGS) + ["lsx_device_count", "lsx_last_error"])


def _hip_runtime_path() -> str:
    """The ONE HIP runtime this process should use.  torch wheels bundle their own
    libamdhip64 (same SONAME as /opt/rocm's); two copies in one process break device
    discovery for whichever initialises second, so prefer torch's when torch is installed."""
    import importlib.util

    cands = []
    try:
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.submodule_search_locations:
            cands.append(os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so"))
    except Exception:
        pass
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cands += [os.path.join(rocm, "lib", "libamdhip64.so.7"), os.path.join(rocm, "lib", "libamdhip64.so")]
    for c in cands:
        if os.path.exists(c):
            return c
    raise LsxError("no libamdhip64 found (looked in torch/lib and $ROCM_PATH/lib)")


def load():
    """dlopen the HIP runtime, then liblsx.so (no device is touched yet)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise LsxError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                    "g.build()'` (hipcc --offload-arch=gfx950). This package has no CPU fallback.")
            C.CDLL(_hip_runtime_path(), mode=C.RTLD_GLOBAL)
            L = C.CDLL(LIB_PATH)
            for name, args in _SIGS.items():
                fn = getattr(L, name)
                fn.restype = C.c_int
                fn.argtypes = args
            L.lsx_device_count.restype = C.c_int
            L.lsx_device_count.argtypes = []
            L.lsx_last_error.restype = C.c_char_p
            L.lsx_last_error.argtypes = []
            _lib = L
    return _lib


def check(status: int, what: str = ""):
    if status != 0:
        msg = load().lsx_last_error().decode("utf-8", "replace")
        raise LsxError(f"{what or 'liblsx'} failed (status {status}): {msg}")


class Handle:
    """One GPU, one stream, growable device workspaces (lsx_create / lsx_destroy)."""

    def __init__(self, device: int = 0):
        self.lib = load()
        self._h = _vp()
        check(self.lib.lsx_create(C.byref(self._h), int(device)), "lsx_create")
        self.device = device

    @property
    def ptr(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.lsx_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key: str, value: int):
        check(self.lib.lsx_set_option(self._h, key.encode(), int(value)), f"set_option({key})")

    def get_option(self, key: str) -> int:
        v = _i(0)
        check(self.lib.lsx_get_option(self._h, key.encode(), C.byref(v)), f"get_option({key})")
        return v.value

    def set_stream(self, stream_ptr: int):
        """Enqueue on this hipStream_t; 0 / None is HIP's default stream (torch's default too)."""
        check(self.lib.lsx_set_stream(self._h, _vp(stream_ptr or None)), "lsx_set_stream")

    def use_own_stream(self):
        check(self.lib.lsx_use_own_stream(self._h), "lsx_use_own_stream")

    def check_status(self):
        """Synchronise and raise LsxError if a cooperative kernel timed out since the last check (lsx_check_status)."""
        check(self.lib.lsx_check_status(self._h), "lsx_check_status")

    def synchronize(self):
        check(self.lib.lsx_synchronize(self._h), "lsx_synchronize")

    def mfma_peak(self, is_f32: bool = False, iters: int = 20000, blocks_per_cu: int = 1):
        """(sustained TFLOP/s, median in-kernel shader clock in MHz [fp64 only, else 0])."""
        t, c = C.c_double(0), C.c_double(0)
        check(self.lib.lsx_diag_mfma_peak(self._h, 1 if is_f32 else 0, iters, blocks_per_cu, C.byref(t),
                                          C.byref(c)))
        return t.value, c.value

    def xchg_probe(self, mode: int, G: int, stride: int, write_through: bool, epochs: int = 2000):
        """(us per exchange round, XCC id of every participant, failures) -- see lsx_diag_xchg_probe."""
        us, nf = C.c_double(0), C.c_int32(0)
        ids = (C.c_int32 * G)()
        check(self.lib.lsx_diag_xchg_probe(self._h, mode, G, stride, 1 if write_through else 0, epochs,
                                           C.byref(us), ids, C.byref(nf)), "diag_xchg_probe")
        return us.value, list(ids), nf.value

    def occupy(self, xcc: int, wgs: int, ms: int):
        """Hold `wgs` CUs of XCD `xcc` for `ms` milliseconds (asynchronous filler; residency tests)."""
        check(self.lib.lsx_diag_occupy(self._h, xcc, wgs, ms), "diag_occupy")

    def cu_mask_probe(self, mask_bits, nblocks: int = 2048):
        """Where the workgroups of a CU-masked stream land: list of (xcc, hw_id) per block; mask_bits = iterable
        of enabled bit positions (None: the handle's own stream)."""
        out = (C.c_uint32 * (2 * nblocks))()
        if mask_bits is None:
            check(self.lib.lsx_diag_cu_mask_probe(self._h, None, 0, nblocks, out), "cu_mask_probe")
        else:
            words = [0] * 8
            for b in mask_bits:
                words[b // 32] |= 1 << (b % 32)
            arr = (C.c_uint32 * 8)(*words)
            check(self.lib.lsx_diag_cu_mask_probe(self._h, arr, 8, nblocks, out), "cu_mask_probe")
        return [(out[2 * i], out[2 * i + 1]) for i in range(nblocks)]

    def read_scratch(self, offset: int, nbytes: int) -> bytes:
        buf = C.create_string_buffer(nbytes)
        check(self.lib.lsx_diag_read_scratch(self._h, offset, buf, nbytes), "diag_read_scratch")
        return buf.raw

    # measurement
    def prof_enable(self, on=True, buckets=None):
        """on=True: every bucket; buckets=("gemm",): only those; on=False: off."""
        if buckets:
            arg = 0
            for b in buckets:
                arg |= 2 << PROF_BUCKETS[b]
        else:
            arg = 1 if on else 0
        check(self.lib.lsx_prof_enable(self._h, arg))

    def prof_reset(self):
        check(self.lib.lsx_prof_reset(self._h))

    def prof_read(self):
        out = {}
        for name, b in PROF_BUCKETS.items():
            ms, fl, by = C.c_double(0), C.c_double(0), C.c_double(0)
            n = C.c_longlong(0)
            check(self.lib.lsx_prof_read(self._h, b, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
            out[name] = {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}
        return out


_default = {}


def default_handle(device: int = 0) -> Handle:
    h = _default.get(device)
    if h is None:
        h = _default[device] = Handle(device)
    return h
