"""Parity of the HIP path (through the C ABI) against the CPU oracle and the
golden vectors generated from the reference.  Runs on the MI355X box: -m gpu.

Tolerances (BASELINE.json north_star): pivot positions and rank exact; fp64
values within 1e-9 relative; fp32 within 1e-4.
"""
from fractions import Fraction

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from helpers import dec, dec_mat, is_numeric_case, load_big, load_small_cases, relerr  # noqa: E402
from oracle import capi, rowreduce  # noqa: E402

TOL64 = 1e-9
TOL32 = 1e-4


@pytest.fixture(scope="module")
def la():
    import linalg_solver_amd as la

    la.default_handle()  # raises loudly when liblsx.so / the GPU is missing
    return la


@pytest.fixture(scope="module")
def dev():
    import torch

    from linalg_solver_amd.device import DeviceSolver

    assert torch.cuda.is_available()
    return DeviceSolver()


DEFAULT_PANEL = 4   # common.h: panel_mode (XCD-scope exchange; taller panels than one XCD holds take mode 3)


def test_native_library_is_the_compute_path(la):
    import ctypes

    from linalg_solver_amd import _native

    assert isinstance(_native.load(), ctypes.CDLL)
    assert _native.load().lsx_device_count() >= 1
    assert la.default_handle().get_option("num_cu") == 256


# --------------------------------------------------------------------- generators
def test_device_fill_matches_host_generator(dev):
    import torch

    from linalg_solver_amd import gen

    for kind in (gen.INT5, gen.U11):
        A = torch.empty(37, 53, dtype=torch.float64, device="cuda")
        dev.fill_(A, kind, 11, row_off=3, col_off=5)
        assert np.array_equal(A.cpu().numpy(), gen.fill(kind, 11, 37, 53, row_off=3, col_off=5))
    A32 = torch.empty(8, 9, dtype=torch.float32, device="cuda")
    dev.fill_(A32, gen.U11, 5)
    assert np.array_equal(A32.cpu().numpy(), gen.fill(gen.U11, 5, 8, 9, dtype=np.float32))


# --------------------------------------------------------------------- MFMA update kernel
@pytest.mark.parametrize("m,n,k", [(16, 16, 4), (128, 128, 128), (300, 200, 64), (1000, 130, 128),
                                   (257, 513, 100), (64, 1, 64), (500, 7, 128), (129, 16, 3),
                                   (1984, 128, 128), (64, 256, 16), (4032, 384, 128)])   # 64-row tiles
def test_gemm_sub_fp64_against_numpy(dev, m, n, k):
    import torch

    rng = np.random.default_rng(m * 7 + n * 3 + k)
    A, B, C = rng.uniform(-1, 1, (m, k)), rng.uniform(-1, 1, (k, n)), rng.uniform(-1, 1, (m, n))
    want = C - A @ B
    dC = torch.from_numpy(C).cuda()
    dev.gemm_sub_(dC, torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda())
    got = dC.cpu().numpy()
    assert np.max(np.abs(got - want)) < 1e-13 * k


def test_gemm_sub_layout_with_identity_and_asymmetric_operand(dev):
    """A = -I (so C - A*B = C + B) with an asymmetric B catches transposed / permuted fragments."""
    import torch

    n = 128
    B = np.arange(n * n, dtype=np.float64).reshape(n, n)  # B[i][j] = i*n + j, exact in fp64
    dC = torch.zeros(n, n, dtype=torch.float64, device="cuda")
    dev.gemm_sub_(dC, torch.from_numpy(-np.eye(n)).cuda(), torch.from_numpy(B).cuda())
    assert np.array_equal(dC.cpu().numpy(), B)


def test_gemm_sub_strided_views(dev):
    """Operands as sub-blocks of a larger row-major matrix (how the LU driver calls it)."""
    import torch

    rng = np.random.default_rng(5)
    big = rng.uniform(-1, 1, (400, 400))
    d = torch.from_numpy(big.copy()).cuda()
    k0, jb = 64, 96
    dev.gemm_sub_(d[k0 + jb:, k0 + jb:], d[k0 + jb:, k0:k0 + jb], d[k0:k0 + jb, k0 + jb:])
    want = big.copy()
    want[k0 + jb:, k0 + jb:] -= big[k0 + jb:, k0:k0 + jb] @ big[k0:k0 + jb, k0 + jb:]
    assert np.max(np.abs(d.cpu().numpy() - want)) < 1e-12


def test_gemm_sub_fp32(dev):
    import torch

    rng = np.random.default_rng(9)
    A, B, C = (rng.uniform(-1, 1, s).astype(np.float32) for s in ((300, 128), (128, 260), (300, 260)))
    dC = torch.from_numpy(C.copy()).cuda()
    dev.gemm_sub_(dC, torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda())
    want = C.astype(np.float64) - A.astype(np.float64) @ B.astype(np.float64)
    assert np.max(np.abs(dC.cpu().numpy() - want)) < 1e-4


# --------------------------------------------------------------------- LU against the CPU twin
def _plu_residual(A, LU, ipiv):
    n = A.shape[0]
    L = np.tril(LU, -1) + np.eye(n)
    U = np.triu(LU)
    PA = A.copy()
    for k in range(n):
        p = ipiv[k]
        if p != k:
            PA[[k, p]] = PA[[p, k]]
    return np.max(np.abs(PA - L @ U)) / max(np.max(np.abs(A)), 1e-300)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 16, 17, 63, 64, 65, 100, 128, 129, 200, 256, 300, 511, 640, 1000])
def test_getrf_matches_cpu_twin(la, n):
    """panel mode 0: two launches per column (the simple fallback path)."""
    from linalg_solver_amd import dense, gen

    A, b = gen.system(gen.U11, 100 + n, n)
    h = la.default_handle()
    h.set_option("panel", 0)
    try:
        LU, ipiv, info = dense.lu_factor(A)
    finally:
        h.set_option("panel", DEFAULT_PANEL)
    oLU, oipiv, oinfo = capi.getrf(A)
    assert info == oinfo == 0
    assert np.array_equal(ipiv, oipiv), "pivot sequence differs from the partial-pivot twin"
    assert np.max(np.abs(np.tril(LU, -1))) <= 1.0
    assert _plu_residual(A, LU, ipiv) < 50 * n * 2.3e-16
    assert relerr(LU, oLU) < TOL64
    x = dense.lu_solve(LU, ipiv, b)
    assert relerr(x, capi.getrs(oLU, oipiv, b)) < TOL64


@pytest.mark.parametrize("n", [1, 3, 64, 127, 128, 129, 200, 256, 300, 640, 1000, 2048, 3000])
def test_getrf_cooperative_panel_matches_cpu_twin(la, n):
    """The one-launch cooperative panels against the CPU twin, and against each other bit for bit: pipelined
    (mode 3, device-scope exchange) in every workgroup shape, XCD-scope (mode 4, the default) alone and under its
    look-ahead schedule, update depth 256 -- and, in the diagnostic build (make DIAG=1), the two superseded
    kernels (modes 1 and 2), which perform the same fused multiply-adds in the same order."""
    from linalg_solver_amd import dense, gen

    h = la.default_handle()
    A, b = gen.system(gen.U11, 500 + n, n)
    oLU, oipiv, oinfo = capi.getrf(A)

    def run(**opts):
        for k, v in opts.items():
            h.set_option(k, v)
        return dense.lu_factor(A)

    try:
        base = run(panel=3, panel_nt=0, panel_rt=4, lookahead=0, kblock=1, lookahead_min=128)
        for nt, rt, look in ((256, 4, 0), (512, 4, 0), (512, 8, 0), (0, 4, 1)):
            LUp, ipivp, infop = run(panel=3, panel_nt=nt, panel_rt=rt, lookahead=look)
            assert infop == 0 and np.array_equal(ipivp, base[1]) and np.array_equal(LUp, base[0]), \
                f"pipelined panel nt={nt} rt={rt} lookahead={look} differs"
        h.set_option("panel_nt", 0)
        h.set_option("panel_rt", 4)
        for look in (0, 1):
            LUx, ipivx, infox = run(panel=4, lookahead=look)
            assert infox == 0 and np.array_equal(ipivx, base[1]) and np.array_equal(LUx, base[0]), \
                f"XCD-scope panel lookahead={look} differs"
        if h.get_option("diag_panels"):
            for mode, rt, nt in ((1, 4, 256), (1, 8, 512), (1, 2, 1024), (2, 4, 0)):
                LUd, ipivd, infod = run(panel=mode, panel_rt=rt, panel_nt=nt, lookahead=0)
                assert infod == 0 and np.array_equal(ipivd, base[1]) and np.array_equal(LUd, base[0]), \
                    f"superseded panel mode {mode} rt={rt} nt={nt} differs"
        # the update depth (K = 128 vs 256) changes summation order only
        deep = run(panel=3, panel_nt=0, panel_rt=4, lookahead=0, kblock=2)
    finally:
        for k, v in (("panel", DEFAULT_PANEL), ("panel_nt", 0), ("panel_rt", 4), ("lookahead", 1), ("kblock", 1),
                     ("lookahead_min", 0)):
            h.set_option(k, v)
    assert np.array_equal(deep[1], base[1]) and relerr(deep[0], base[0]) < 1e-12
    LU, ipiv, info = base
    assert info == oinfo == 0
    assert np.array_equal(ipiv, oipiv), "pivot sequence differs from the partial-pivot twin"
    assert np.max(np.abs(np.tril(LU, -1))) <= 1.0
    assert _plu_residual(A, LU, ipiv) < 50 * n * 2.3e-16
    assert relerr(LU, oLU) < TOL64


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
@pytest.mark.parametrize("n", [16, 200, 513])
def test_getrf_cooperative_panel_integer_and_singular(la, n, mode):
    from linalg_solver_amd import dense, gen

    h = la.default_handle()
    if mode in (1, 2) and not h.get_option("diag_panels"):
        pytest.skip("superseded panel kernels are only in the diagnostic build (make DIAG=1)")
    h.set_option("panel", mode)
    try:
        A, _ = gen.system(gen.INT5, 60 + n, n)
        LU, ipiv, info = dense.lu_factor(A)
        S = A.copy()
        S[:, 5] = 0.0  # a zero column stays exactly zero: both sides must report column 6
        _, _, sinfo = dense.lu_factor(S)
        LU32, ipiv32, info32 = dense.lu_factor(A.astype(np.float32), dtype=np.float32)
    finally:
        h.set_option("panel", DEFAULT_PANEL)
    assert info == 0 and _plu_residual(A, LU, ipiv) < 50 * n * 2.3e-16
    assert np.max(np.abs(np.tril(LU, -1))) <= 1.0
    assert sinfo == capi.getrf(S)[2] == 6
    assert info32 == 0 and _plu_residual(A, LU32.astype(np.float64), ipiv32) < 1e-4


@pytest.mark.parametrize("n", [8, 64, 200, 384])
def test_getrf_integer_matrices(la, n):
    """The reference's own distribution (random_matrix.py:104): many exact ties in |a|."""
    from linalg_solver_amd import dense, gen

    A, b = gen.system(gen.INT5, 40 + n, n)
    LU, ipiv, info = dense.lu_factor(A)
    oLU, oipiv, oinfo = capi.getrf(A)
    assert info == oinfo
    if info == 0:
        assert _plu_residual(A, LU, ipiv) < 50 * n * 2.3e-16
        assert np.max(np.abs(np.tril(LU, -1))) <= 1.0
        x = dense.lu_solve(LU, ipiv, b)
        assert relerr(x, capi.getrs(oLU, oipiv, b)) < 1e-8


def test_getrf_exactly_singular_reports_info(la):
    from linalg_solver_amd import dense

    A = np.array([[1.0, 2.0, 3.0], [2.0, 4.0, 6.0], [1.0, 0.0, 1.0]])
    _, _, info = dense.lu_factor(A)
    _, _, oinfo = capi.getrf(A)
    assert info == oinfo and info > 0
    Z = np.zeros((5, 5))
    assert dense.lu_factor(Z)[2] == 1


@pytest.mark.parametrize("nrhs", [1, 3, 17, 130])
def test_multi_rhs_solve(la, nrhs):
    from linalg_solver_amd import dense, gen

    n = 300
    A = gen.fill(gen.U11, 77, n, n)
    B = gen.fill(gen.U11, 78, n, nrhs)
    X, info, ratio = dense.solve(A, B)
    assert info == 0 and ratio > 1e-12
    oLU, oipiv, _ = capi.getrf(A)
    assert relerr(X, capi.getrs(oLU, oipiv, B)) < TOL64
    assert np.max(np.abs(A @ X - B)) < 1e-10


def test_inverse_and_determinant(la):
    from linalg_solver_amd import dense, gen

    for n in (1, 2, 7, 64, 129, 400):
        A = gen.fill(gen.U11, 300 + n, n, n)
        Ai, info, _ = dense.inv(A)
        assert info == 0
        assert np.max(np.abs(A @ Ai - np.eye(n))) < 1e-9
        s, l = dense.slogdet(A)
        s2, l2 = np.linalg.slogdet(A)
        assert s == s2 and abs(l - l2) <= TOL64 * max(1.0, abs(l2))
        oLU, oipiv, _ = capi.getrf(A)
        os_, ol = capi.slogdet(oLU, oipiv)
        assert s == os_ and abs(l - ol) <= TOL64 * max(1.0, abs(ol))


def test_determinant_does_not_overflow(la):
    from linalg_solver_amd import dense, gen

    n = 600
    A = gen.fill(gen.INT5, 2, n, n) * 1e3
    s, m, e = dense.det_parts(A)
    s2, l2 = np.linalg.slogdet(A)
    assert s == s2 and 0.5 <= m < 1 and abs((np.log(m) + e * np.log(2)) - l2) < 1e-9 * abs(l2)
    assert np.isinf(dense.det(A))


def test_fp32_lu_within_1e4_of_fp64(la):
    from linalg_solver_amd import dense, gen

    n = 1024
    A, b = gen.system(gen.U11, 5, n)
    x64, info, _ = dense.solve(A, b)
    x32, info32, _ = dense.solve(A.astype(np.float32), b.astype(np.float32), dtype=np.float32)
    assert info == 0 and info32 == 0
    # backward-error form: the fp32 factorisation of a random matrix is only conditionally
    # accurate forward; north_star's 1e-4 is checked on the scaled residual and on the factors
    r = np.max(np.abs(A @ x32.astype(np.float64) - b)) / (np.max(np.abs(A)) * np.max(np.abs(x32)) * n)
    assert r < TOL32
    LU32, ipiv32, _ = dense.lu_factor(A.astype(np.float32), dtype=np.float32)
    LU64, ipiv64, _ = dense.lu_factor(A)
    assert _plu_residual(A.astype(np.float32).astype(np.float64), LU32.astype(np.float64), ipiv32) < TOL32


# --------------------------------------------------------------------- golden vectors from the reference
CASES = load_small_cases()


def _exact(items):
    return [[Fraction(v) for v in row] for row in items]


def _as_float_rows(M):
    return np.array([[float(v) for v in row] for row in M], dtype=np.float64)


RR = [c for c in CASES if c["op"] == "row_reduce"]


@pytest.mark.parametrize("case", RR, ids=[c["name"] for c in RR])
def test_row_reduce_golden(la, case):
    items = dec_mat(case["items"])
    if not is_numeric_case(items):
        # exact-entry twin of a float case: feed its float image, expect the exact answer
        fitems = [[float(v) for v in row] for row in items]
    else:
        fitems = items
    red, pivots, mats, steps = la.Matrix(fitems).row_reduce(case["bar_col"])
    assert mats == [] and steps == []
    # ground truth in exact arithmetic (same algorithm, Fractions): pivots and rank must match it
    ered, epiv, _ = rowreduce.row_reduce(_exact(fitems), case["bar_col"])
    assert pivots == epiv
    want = _as_float_rows(ered)
    scale = max(1.0, float(np.max(np.abs(want))))
    assert np.max(np.abs(np.array(red) - want)) <= TOL64 * scale
    # and, wherever the reference's float run did not suffer the rank artefact, its own numbers
    if is_numeric_case(items) and [list(p) for p in epiv] == case["pivots"]:
        ref = _as_float_rows(dec_mat(case["reduced"]))
        assert np.max(np.abs(np.array(red) - ref)) <= TOL64 * max(1.0, float(np.max(np.abs(ref))))


PRE = [c for c in CASES if c["op"] == "find_preimage_of"]


@pytest.mark.parametrize("case", PRE, ids=[c["name"] for c in PRE])
def test_find_preimage_golden(la, case):
    items = [[float(v) for v in row] for row in dec_mat(case["items"])]
    vec = [float(dec(v)) for v in case["vec"]]
    got = la.Matrix(items).find_preimage_of(vec, log_steps=True)
    exact = rowreduce.find_preimage_of(_exact(items), [Fraction(v) for v in vec])
    if exact == rowreduce.NO_SOLUTION:
        assert isinstance(got, la.Matrix.NoSolution)
        return
    part, gens, _ = exact
    assert isinstance(got, la.Matrix.AffineSubspace)
    want = np.array([float(v) for v in part])
    assert np.max(np.abs(np.array(got.vec, dtype=float) - want)) <= TOL64 * max(1.0, np.max(np.abs(want)))
    if gens is None:
        assert got.generators is None
    else:
        G = _as_float_rows(gens)
        assert got.generators is not None and got.dim() == G.shape[1]
        assert np.max(np.abs(_as_float_rows(got.generators.items) - G)) <= TOL64 * max(1.0, np.max(np.abs(G)))
    # same carrier shape as the reference's own float run whenever that run had no rank artefact
    want_ref = case["result"]
    if want_ref["kind"] == "AffineSubspace" and (want_ref["generators"] is None) == (gens is None):
        ref = np.array([float(dec(v)) for v in want_ref["particular"]])
        assert np.max(np.abs(np.array(got.vec, dtype=float) - ref)) <= 1e-8 * max(1.0, np.max(np.abs(ref)))


INV = [c for c in CASES if c["op"] == "inverse"]


@pytest.mark.parametrize("case", INV, ids=[c["name"] for c in INV])
def test_inverse_golden(la, case):
    items = [[float(v) for v in row] for row in dec_mat(case["items"])]
    got = la.Matrix(items).inverse(log_steps=True)
    exact = rowreduce.inverse(_exact(items))
    if exact == rowreduce.NO_SOLUTION:
        assert isinstance(got, la.Matrix.NoSolution)
        return
    want = _as_float_rows(exact)
    assert isinstance(got, la.Matrix)
    assert np.max(np.abs(_as_float_rows(got.items) - want)) <= TOL64 * max(1.0, np.max(np.abs(want)))
    if case["result"]["kind"] == "Matrix":
        ref = _as_float_rows(dec_mat(case["result"]["items"]))
        assert np.max(np.abs(_as_float_rows(got.items) - ref)) <= 1e-8 * max(1.0, np.max(np.abs(ref)))


def test_cfg1_64x64_reference_stream(la, golden_dir):
    """BASELINE config #1: 64 x 64, random.seed(2026), row_reduce + find_preimage_of."""
    z = np.load(golden_dir + "/cfg1_n64.npz")
    A, b = z["A"], z["b"]
    M = la.Matrix(A.tolist())
    red, pivots, _, _ = la.Matrix(np.hstack([A, b[:, None]]).tolist()).row_reduce()
    assert pivots == [(k, k) for k in range(64)] == [tuple(p) for p in z["pivots"].tolist()]
    assert relerr(np.array(red), z["reduced"]) < TOL64
    sol = M.find_preimage_of(b.tolist(), log_steps=True)
    assert sol.generators is None and relerr(sol.vec, z["x"]) < TOL64
    sol2 = M.find_preimage_of(b.tolist())
    assert sol2.generators.rows == 64 and sol2.generators.cols == 0 and sol2.dim() == 0
    inv = M.inverse(log_steps=True)
    assert relerr(np.array(inv.items), z["inverse"]) < TOL64
    assert M.rank() == 64
    s, l = M.slogdet()
    s2, l2 = np.linalg.slogdet(A)
    assert s == s2 and abs(l - l2) < 1e-9 * abs(l2)
    assert abs(M.determinant() / (s2 * np.exp(l2)) - 1) < 1e-9


@pytest.mark.parametrize("name,tol", [("n128_int5", 1e-9), ("n128_u11", 1e-9), ("n256_int5", 1e-9),
                                      ("n256_u11", 1e-9), ("n512_u11", 1e-8)])
def test_reference_solutions_at_larger_n(la, name, tol):
    """The reference is unpivoted, so at n=512 it is the less accurate side;
    1e-8 there is the reference's own distance from LAPACK, not ours."""
    A, b, z = load_big(name)
    sol = la.Matrix(A.tolist()).find_preimage_of(b.tolist(), log_steps=True)
    assert isinstance(sol, la.Matrix.AffineSubspace) and sol.generators is None
    assert relerr(sol.vec, z["x"]) < tol
    oLU, oipiv, _ = capi.getrf(A)
    assert relerr(sol.vec, capi.getrs(oLU, oipiv, b)) < TOL64


# --------------------------------------------------------------------- general RREF
def test_rref_rank_deficient_and_rectangular(la):
    from linalg_solver_amd import dense

    from linalg_solver_amd import _native as N

    rng = np.random.default_rng(3)
    for m, n, r, rule in ((40, 60, 13, N.PIVOT_MAX), (60, 40, 25, N.PIVOT_MAX), (100, 100, 1, N.PIVOT_FIRST),
                          (33, 70, 33, N.PIVOT_MAX), (200, 300, 50, N.PIVOT_MAX), (30, 20, 6, N.PIVOT_FIRST)):
        P = rng.integers(-3, 4, (m, r)).astype(float)
        Q = rng.integers(-3, 4, (r, n)).astype(float)
        A = P @ Q
        R, pivots, rank = dense.rref(A, bar_col=n, pivot_rule=rule)
        true_rank = np.linalg.matrix_rank(A)
        assert rank == true_rank == len(pivots)
        pc = [c for _, c in pivots]
        assert [rw for rw, _ in pivots] == list(range(rank)) and pc == sorted(pc)
        # RREF property: A = A[:, pivot columns] @ R[:rank]
        assert np.max(np.abs(A[:, pc] @ R[:rank] - A)) < 1e-9 * np.max(np.abs(A))
        assert np.all(R[rank:] == 0)
        sub = R[:rank][:, pc]
        assert np.array_equal(sub, np.eye(rank))
        assert la.Matrix(A.tolist()).rank() == true_rank


def test_rref_medium_against_exact(la):
    from linalg_solver_amd import dense

    rng = np.random.default_rng(8)
    A = rng.integers(-5, 6, (24, 31)).astype(float)
    A[7] = A[3] - 2 * A[5]
    A[:, 11] = 0
    A[:, 4] = A[:, 2]
    for bar in (None, 31, 20, 1):
        ered, epiv, _ = rowreduce.row_reduce(_exact(A.tolist()), bar)
        want = _as_float_rows(ered)
        red, pivots, _, _ = la.Matrix(A.tolist()).row_reduce(bar)
        assert pivots == epiv
        assert np.max(np.abs(np.array(red) - want)) < 1e-9 * max(1.0, np.max(np.abs(want)))
        # the max-|a| rule alone: same pivots, same left block (the carried columns may differ)
        R, pivots2, rank = dense.rref(A, bar_col=bar, pivot_rule=1)
        b = bar or A.shape[1] - 1
        assert pivots2 == epiv and np.max(np.abs(R[:, :b] - want[:, :b])) < 1e-9


def test_matrix_product_on_the_mfma_tile(la):
    """Matrix.__mul__ (linalg.py:101-158) and the residual checks it is for."""
    from linalg_solver_amd import dense, gen

    rng = np.random.default_rng(12)
    for m, k, n in ((3, 4, 2), (64, 64, 64), (130, 257, 65), (300, 1, 200), (1, 500, 1)):
        A, B = rng.uniform(-1, 1, (m, k)), rng.uniform(-1, 1, (k, n))
        assert np.max(np.abs(dense.matmul(A, B) - A @ B)) < 1e-12 * max(k, 1)
    M1, M2 = la.Matrix([[1, 2], [3, 4]]), la.Matrix([[0.5, 0.0], [1.0, -1.0]])
    assert (M1 * M2).items == [[2.5, -2.0], [5.5, -4.0]]
    assert (M1 * 2).items == [[2, 4], [6, 8]] and (-M1).items == [[-1, -2], [-3, -4]]
    with pytest.raises(ValueError, match="dimensions must match"):
        la.Matrix([[1.0, 2.0]]) * la.Matrix([[1.0, 2.0]])
    A = gen.fill(gen.U11, 9, 200, 200)
    Ai = la.Matrix(A.tolist()).inverse()
    P = np.array((la.Matrix(A.tolist()) * Ai).items)
    assert np.max(np.abs(P - np.eye(200))) < 1e-10


def test_kernel_and_underdetermined(la):
    A = [[1.0, 2.0, 3.0, 4.0], [2.0, 4.0, 6.0, 8.0], [1.0, 0.0, 1.0, 0.0]]
    ker = la.Matrix(A).kernel()
    assert ker.dim() == 2 and all(v == 0 for v in ker.vec)
    G = np.array(ker.generators.items, dtype=float)
    assert np.max(np.abs(np.array(A) @ G)) < 1e-12
    assert la.Matrix(A).rank() == 2
    assert isinstance(la.Matrix([[1.0, 2.0], [2.0, 4.0]]).inverse(), la.Matrix.NoSolution)
    assert isinstance(la.Matrix([[1.0, 2.0], [2.0, 4.0]]).find_preimage_of([1.0, 3.0]), la.Matrix.NoSolution)


def test_row_reduce_bar_col_past_the_matrix_fails_like_the_reference(la):
    assert la.Matrix([[1.0, 2.0], [3.0, 4.0]]).row_reduce(bar_col=3)[1] == [(0, 0), (1, 1)]
    with pytest.raises(IndexError):
        la.Matrix([[1.0, 2.0], [2.0, 4.0], [0.0, 0.0]]).row_reduce(bar_col=3)


# --------------------------------------------------------------------- BASELINE sizes: invariants
@pytest.mark.parametrize("n,kind", [(4096, "u11"), (4096, "int5"), (8192, "u11"), (16384, "u11")])
def test_full_size_lu_invariants(dev, n, kind):
    """configs 2, 3 and 4 (the 16384 matrix on one GPU): P A = L U, |L| <= 1, solve residual, inverse,
    determinant sign/log."""
    import torch

    from linalg_solver_amd import gen

    A = torch.empty(n, n, dtype=torch.float64, device="cuda")
    dev.fill_(A, gen.INT5 if kind == "int5" else gen.U11, 1)
    LU = A.clone()
    ipiv, info = dev.getrf_(LU)
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    piv = ipiv.cpu().numpy()
    assert np.all(piv >= np.arange(n)) and np.all(piv < n)
    L = torch.tril(LU, -1)
    assert float(L.abs().max()) <= 1.0
    L.diagonal().fill_(1.0)
    U = torch.triu(LU)
    # P A from the interchange list
    perm = np.arange(n)
    for k in range(n):
        p = piv[k]
        if p != k:
            perm[k], perm[p] = perm[p], perm[k]
    PA = A[torch.from_numpy(perm).cuda()]
    res = float((PA - L @ U).abs().max() / A.abs().max())
    assert res < 1e-11, res
    # one right-hand side (config 2) -- backward error
    b = torch.empty(n, 1, dtype=torch.float64, device="cuda")
    dev.fill_(b, gen.U11, 1, col_off=gen.RHS_COL)
    x = b.clone()
    dev.getrs_(LU, ipiv, x)
    berr = float((A @ x - b).abs().max() / (A.abs().max() * x.abs().max() * n))
    assert berr < 1e-14, berr
    # determinant against torch's own slogdet
    parts = dev.det_parts(LU, ipiv).cpu().numpy()
    s2, l2 = torch.linalg.slogdet(A)
    assert parts[0] == float(s2)
    assert abs(np.log(parts[1]) + parts[2] * np.log(2.0) - float(l2)) < 1e-9 * abs(float(l2))
    if n <= 4096 or (kind == "u11" and n <= 8192):
        Ainv = dev.getri(LU, ipiv)
        eye_err = float((A @ Ainv - torch.eye(n, dtype=torch.float64, device="cuda")).abs().max())
        assert eye_err < 1e-7, eye_err


@pytest.mark.parametrize("n", [129, 255, 300, 641, 1000, 1537, 2500, 3001])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_lookahead_driver_on_ragged_sizes_is_bit_identical(dev, n, dtype):
    """The look-ahead driver (default above 7168 / 11264) forced onto small and odd orders: last panel narrower
    than nb, odd leading dimension (no 16-byte paths), fewer tiles than CUs.  Same bits as
    the sequential driver, for every variant."""
    import torch

    from linalg_solver_amd import gen

    tdt = torch.float64 if dtype == "f64" else torch.float32
    A0 = torch.empty(n, n, dtype=tdt, device="cuda")
    dev.fill_(A0, gen.U11, 40 + n)
    outs = []
    try:
        dev.h.set_option("lookahead_min", 128)
        # last entry: hand-over from the shared-CU schedule to the XCD-scope driver forced at 2/5 of the order
        for mode, look, xl in ((3, 0, 0), (3, 1, 0), (4, 0, 0), (4, 1, 0), (4, 1, max(128, n * 2 // 5))):
            dev.h.set_option("panel", mode)
            dev.h.set_option("lookahead", look)
            dev.h.set_option("xrows_limit", xl)
            LU = A0.clone()
            ipiv, info = dev.getrf_(LU)
            torch.cuda.synchronize()
            assert int(info.item()) == 0
            outs.append((LU, ipiv.clone()))
    finally:
        dev.h.set_option("panel", DEFAULT_PANEL)
        dev.h.set_option("lookahead", 1)
        dev.h.set_option("lookahead_min", 0)
        dev.h.set_option("xrows_limit", 0)
    for LU, ipiv in outs[1:]:
        assert torch.equal(ipiv, outs[0][1]) and torch.equal(LU, outs[0][0])


@pytest.mark.parametrize("n,dtype", [(2048, "f64"), (2075, "f64"), (3072, "f64"), (3101, "f64"), (7168, "f64"), (7203, "f64"), (8192, "f64"), (8320, "f64"),
                                     (9000, "f64"), (12288, "f64"), (4096, "f32"), (4131, "f32"), (10240, "f32"), (10307, "f32"),
                                     (16533, "f32")])
def test_default_driver_around_the_lookahead_thresholds(dev, n, dtype):
    """At and just above the orders where the look-ahead driver takes over by default (aligned and odd): same
    bits as the sequential driver, and P A = L U to working precision."""
    import torch

    from linalg_solver_amd import gen

    tdt = torch.float64 if dtype == "f64" else torch.float32
    A0 = torch.empty(n, n, dtype=tdt, device="cuda")
    dev.fill_(A0, gen.U11, 7)
    assert dev.h.get_option("lookahead") == 1 and dev.h.get_option("lookahead_min") == 0
    LU = A0.clone()
    ipiv, info = dev.getrf_(LU)
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    try:
        dev.h.set_option("lookahead", 0)
        LUs = A0.clone()
        ipivs, infos = dev.getrf_(LUs)
        torch.cuda.synchronize()
    finally:
        dev.h.set_option("lookahead", 1)
    assert torch.equal(ipiv, ipivs) and torch.equal(LU, LUs)
    piv = ipiv.cpu().numpy()
    perm = np.arange(n)
    for k in range(n):
        p = piv[k]
        if p != k:
            perm[k], perm[p] = perm[p], perm[k]
    L = torch.tril(LU, -1).double()
    L.diagonal().fill_(1.0)
    D = A0[torch.from_numpy(perm).cuda()].double() - L @ torch.triu(LU).double()
    res = float(torch.linalg.norm(D) / torch.linalg.norm(A0.double()))
    assert res < (1e-13 if dtype == "f64" else TOL32), res


@pytest.mark.parametrize("n,dtype", [(3200, "f64"), (4096, "f64"), (5120, "f32")])
def test_x_schedule_orderings_and_tile_heights_agree(dev, n, dtype):
    """The XCD-scope schedule orders its panel chain behind a COUNT of the previous update's first tile column (waited
    for inside the chain head; option x_events=0, default) or behind an event on the whole update plus the gate
    (x_events=1, also the form ragged orders take); the next panel's column block is updated on 32-row or 64-row
    tiles (option gemm_tiles32).  All four combinations and the sequential driver give the same bits."""
    import torch

    from linalg_solver_amd import gen

    tdt = torch.float64 if dtype == "f64" else torch.float32
    A0 = torch.empty(n, n, dtype=tdt, device="cuda")
    dev.fill_(A0, gen.U11, 90 + n)
    outs = []
    try:
        dev.h.set_option("lookahead", 0)
        LU = A0.clone()
        ipiv, info = dev.getrf_(LU)
        torch.cuda.synchronize()
        assert int(info.item()) == 0
        outs.append((LU, ipiv.clone()))
        dev.h.set_option("lookahead", 1)
        for xev in (0, 1):
            for t32 in (1, 0):
                dev.h.set_option("x_events", xev)
                dev.h.set_option("gemm_tiles32", t32)
                LU = A0.clone()
                ipiv, info = dev.getrf_(LU)
                torch.cuda.synchronize()
                assert int(info.item()) == 0
                from linalg_solver_amd import _native
                _native.check(dev.h.lib.lsx_check_status(dev.h.ptr), "status after the factorisation")
                outs.append((LU, ipiv.clone()))
    finally:
        dev.h.set_option("lookahead", 1)
        dev.h.set_option("x_events", 0)
        dev.h.set_option("gemm_tiles32", 1)
    for LU, ipiv in outs[1:]:
        assert torch.equal(ipiv, outs[0][1]) and torch.equal(LU, outs[0][0])


@pytest.mark.parametrize("n,dtype", [(600, "f64"), (1000, "f64"), (2048, "f64"), (1537, "f32"), (4096, "f64")])
def test_structured_inverse_equals_the_plain_solve_of_the_permuted_identity(dev, n, dtype):
    """getri: U^-1 L^-1 P with the forward substitution restricted to the columns where L^-1 can be non-zero and the
    column permutation applied last (4/3 n^3 flops) against the plain n-right-hand-side solve of P*I (2 n^3): the
    skipped operations multiply exact zeros, so the two inverses agree bit for bit; A * A^-1 = I to working precision."""
    import torch

    from linalg_solver_amd import gen

    tdt = torch.float64 if dtype == "f64" else torch.float32
    A0 = torch.empty(n, n, dtype=tdt, device="cuda")
    dev.fill_(A0, gen.U11, 17 + n)
    LU = A0.clone()
    ipiv, info = dev.getrf_(LU)
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    invs = []
    try:
        for structured in (1, 0):
            dev.h.set_option("getri_structured", structured)
            invs.append(dev.getri(LU, ipiv).clone())
            torch.cuda.synchronize()
    finally:
        dev.h.set_option("getri_structured", 1)
    # signed zeros aside (0.0 == -0.0), the same bits
    assert torch.equal(invs[0], invs[1])
    R = A0.double() @ invs[0].double() - torch.eye(n, dtype=torch.float64, device="cuda")
    assert float(R.abs().max()) < (1e-9 if dtype == "f64" else 5e-2)


def test_lookahead_variants_are_bit_identical_at_8192(dev):
    """Look-ahead (panel k+1 under the update of step k; with panel = 4 on an XCD of its own) only reorders
    launches: the factors must not change by a single bit."""
    import torch

    from linalg_solver_amd import gen

    n = 8192
    A0 = torch.empty(n, n, dtype=torch.float64, device="cuda")
    dev.fill_(A0, gen.U11, 4)
    outs = []
    try:
        for mode, look in ((3, 0), (3, 1), (4, 0), (4, 1)):
            dev.h.set_option("panel", mode)
            dev.h.set_option("lookahead", look)
            LU = A0.clone()
            ipiv, info = dev.getrf_(LU)
            torch.cuda.synchronize()
            assert int(info.item()) == 0
            outs.append((LU, ipiv.clone()))
    finally:
        dev.h.set_option("panel", DEFAULT_PANEL)
        dev.h.set_option("lookahead", 1)
    for LU, ipiv in outs[1:]:
        assert torch.equal(ipiv, outs[0][1]) and torch.equal(LU, outs[0][0])


def test_full_size_fp32(dev):
    """config 5: 8192 x 8192 fp32 LU, tolerance 1e-4."""
    import torch

    from linalg_solver_amd import gen

    n = 8192
    A = torch.empty(n, n, dtype=torch.float32, device="cuda")
    dev.fill_(A, gen.U11, 1)
    LU = A.clone()
    ipiv, info = dev.getrf_(LU)
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    piv = ipiv.cpu().numpy()
    perm = np.arange(n)
    for k in range(n):
        p = piv[k]
        if p != k:
            perm[k], perm[p] = perm[p], perm[k]
    L = torch.tril(LU, -1).double()
    assert float(L.abs().max()) <= 1.0
    L.diagonal().fill_(1.0)
    # norm-wise backward error of the factorisation (forward error of an fp32 solve at this size is
    # cond(A) * eps32 and cannot meet 1e-4 without refinement; the factors themselves can)
    D = A[torch.from_numpy(perm).cuda()].double() - L @ torch.triu(LU).double()
    res = float(torch.linalg.norm(D) / torch.linalg.norm(A.double()))
    assert res < TOL32, res
    assert float(D.abs().max() / A.abs().max()) < 1e-3


@pytest.mark.parametrize("n", [129, 200, 1000, 2500, 4096, 4100])
@pytest.mark.parametrize("nrhs", [1, 2, 3, 8])
def test_cooperative_triangular_solve_matches_the_step_path(la, n, nrhs):
    """Few right-hand sides: the three forms of the solve-latency path -- 128-row steps with helper workgroups
    and the interchanges fused into the preparation launch (trsv=2, default), one cooperative launch per direction
    with 64-row steps (trsv=1), one launch per 128-row step (trsv=0) -- against each other and the CPU twin."""
    from linalg_solver_amd import dense, gen

    A, _ = gen.system(gen.U11, 900 + n, n)
    rng = np.random.default_rng(n + nrhs)
    B = rng.uniform(-1, 1, (n, nrhs))
    h = la.default_handle()
    LU, ipiv, info = dense.lu_factor(A)
    assert info == 0
    try:
        h.set_option("trsv", 0)
        x0 = dense.lu_solve(LU, ipiv, B)
        h.set_option("trsv", 1)
        x1 = dense.lu_solve(LU, ipiv, B)
        h.set_option("trsv", 2)
        x2 = dense.lu_solve(LU, ipiv, B)
        x2b = dense.lu_solve(LU, ipiv, B)
    finally:
        h.set_option("trsv", 2)
    oLU, oipiv, _ = capi.getrf(A)
    xo = capi.getrs(oLU, oipiv, B)
    assert relerr(x1, x0) < 1e-11 and relerr(x1, xo) < TOL64
    assert relerr(x2, x0) < 1e-11 and relerr(x2, xo) < TOL64
    assert np.array_equal(x2, x2b), "the solve is not deterministic"
    assert np.max(np.abs(A @ x1 - B)) < 1e-9 * n and np.max(np.abs(A @ x2 - B)) < 1e-9 * n


@pytest.mark.parametrize("n,nrhs", [(300, 1), (1000, 4), (2048, 1)])
def test_cooperative_triangular_solve_fp32(la, n, nrhs):
    from linalg_solver_amd import dense, gen

    A, _ = gen.system(gen.U11, 950 + n, n)
    rng = np.random.default_rng(n)
    B = rng.uniform(-1, 1, (n, nrhs))
    LU, ipiv, info = dense.lu_factor(A.astype(np.float32), dtype=np.float32)
    assert info == 0
    h = la.default_handle()
    try:
        h.set_option("trsv", 0)
        x0 = dense.lu_solve(LU, ipiv, B.astype(np.float32))
        h.set_option("trsv", 1)
        x1o = dense.lu_solve(LU, ipiv, B.astype(np.float32))
        h.set_option("trsv", 2)
        x1 = dense.lu_solve(LU, ipiv, B.astype(np.float32))
    finally:
        h.set_option("trsv", 2)
    assert x1.dtype == np.float32 and relerr(x1.astype(np.float64), x0.astype(np.float64)) < 1e-4
    assert relerr(x1o.astype(np.float64), x0.astype(np.float64)) < 1e-4
    # norm-wise backward error of the fp32 solve
    resid = np.linalg.norm(A @ x1.astype(np.float64) - B) / (np.linalg.norm(A) * np.linalg.norm(x1) + np.linalg.norm(B))
    assert resid < 1e-4


@pytest.mark.parametrize("n", [1, 2, 7, 64, 129, 300, 1000, 2049, 5000])
@pytest.mark.parametrize("pattern", ["random", "all_last", "identity", "next", "clustered"])
def test_interchange_list_to_permutation(la, n, pattern):
    """ipiv -> perm conversion (index arrays + chase) against the sequential definition, through getrs on
    an identity factor: x = P b.  Includes lists that make the chase long (every step picks the last row)."""
    from linalg_solver_amd import dense

    rng = np.random.default_rng(n)
    k = np.arange(n)
    if pattern == "random":
        ipiv = rng.integers(k, n)
    elif pattern == "all_last":
        ipiv = np.full(n, n - 1)
    elif pattern == "identity":
        ipiv = k.copy()
    elif pattern == "next":
        ipiv = np.minimum(k + 1, n - 1)
    else:
        ipiv = np.minimum(k + rng.integers(0, 3, n), n - 1)
    for nrhs in (1, 9):     # the few-RHS and the many-RHS paths
        b = rng.uniform(-1, 1, (n, nrhs))
        want = b.copy()
        for i in range(n):
            p = int(ipiv[i])
            if p != i:
                want[[i, p]] = want[[p, i]]
        got = dense.lu_solve(np.eye(n), ipiv.astype(np.int32), b)
        assert np.array_equal(got, want), f"nrhs={nrhs}"


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("jb,ld,ncols", [(128, 300, 128), (128, 300, 44), (64, 128, 64), (100, 304, 32)])
def test_chain_head_fused_equals_separate(la, dtype, jb, ld, ncols):
    """The fused head of the look-ahead chain (block inverses + gather list in one launch) gives the SAME BITS as
    the block-inverse launch on its own, in both precisions, and both are the inverse.  Round 1 found the fp32
    instantiation differing: hipcc contracted the substitution's multiply-adds differently per inlining context;
    they are explicit fused operations now (kernels_misc.hip: fma_t)."""
    import torch

    from linalg_solver_amd import _native

    h = la.default_handle()
    tdt = torch.float32 if dtype == "f32" else torch.float64
    fn = getattr(h.lib, f"lsx_diag_chain_head_{dtype}")
    torch.manual_seed(jb + ld)
    Tm = torch.rand(jb + 200, ld, dtype=tdt, device="cuda") - 0.5
    A0 = torch.rand(jb + 200, ld, dtype=tdt, device="cuda")
    mv = torch.full((256, 2), -1, dtype=torch.int32, device="cuda")
    perm = torch.randperm(jb)[:40].tolist()
    for i in range(0, 40, 2):   # 20 row exchanges
        mv[i, 0], mv[i, 1] = perm[i], perm[i + 1]
        mv[i + 1, 0], mv[i + 1, 1] = perm[i + 1], perm[i]
    nblk = (jb + 63) // 64
    outs, As = [], []
    for fused in (0, 1):
        Ti = torch.zeros(nblk * 4096, dtype=tdt, device="cuda")
        A = A0.clone()
        _native.check(fn(h._h, fused, jb, Tm.data_ptr(), ld, Ti.data_ptr(), ncols, A.data_ptr(), ld, 0, mv.data_ptr()),
                      "diag_chain_head")
        h.synchronize()
        torch.cuda.synchronize()
        outs.append(Ti)
        As.append(A)
    assert torch.equal(outs[0], outs[1]), "fused and separate block inverses differ"
    for b in range(nblk):
        w = min(64, jb - 64 * b)
        L = torch.tril(Tm[64 * b:64 * b + w, 64 * b:64 * b + w].double(), -1) + torch.eye(w, dtype=torch.float64, device="cuda")
        got = outs[1][4096 * b:4096 * (b + 1)].view(64, 64)[:w, :w].double()
        assert float((got - torch.linalg.inv(L)).abs().max()) < (1e-4 if dtype == "f32" else 1e-12)
    # the gather list moved the rows of the first ncols columns and nothing else
    exp = A0.clone()
    src = A0.clone()
    for d, s_ in mv.tolist():
        if d >= 0:
            exp[d, :ncols] = src[s_, :ncols]
    assert torch.equal(As[1], exp) and torch.equal(As[0], A0)


@pytest.mark.parametrize("n", [64, 300, 1024])
def test_fp32_solve_refined_matches_the_fp64_oracle_to_1e4(la, n):
    """BASELINE config 5, to the letter: the fp32 path's SOLUTION within 1e-4 of the fp64 result.  Factors in fp32,
    residuals in fp64 (lsx_gesv_f32_refined); the oracle is the CPU twin in fp64 on the same (fp32-valued) system.
    The unrefined fp32 solve is reported beside it: it is the one that cannot promise 1e-4 (cond * eps32)."""
    from linalg_solver_amd import dense, gen

    A, b = gen.system(gen.U11, 900 + n, n)
    A32, b32 = A.astype(np.float32), b.astype(np.float32)
    A64, b64 = A32.astype(np.float64), b32.astype(np.float64)
    oLU, oipiv, oinfo = capi.getrf(A64)
    xo = capi.getrs(oLU, oipiv, b64)
    x, info, ratio, corr = dense.solve_refined(A32, b32, sweeps=3)
    x0, info0, _ = dense.solve(A32, b32, dtype=np.float32)
    assert info == 0 and info0 == 0 and oinfo == 0
    err = relerr(x, xo)
    err0 = relerr(x0.astype(np.float64), xo)
    print(f"n={n}: forward error refined {err:.2e}, unrefined {err0:.2e}, last correction {corr:.2e}")
    assert err < 1e-9, err          # fp64-residual refinement converges to the fp64 solution itself
    assert err < TOL32 and corr < 1e-6
    assert err <= err0


def test_fp32_inverse_determinant_and_rank(la):
    """lsx_getri_f32 / lsx_det_f32 / lsx_rref_f32 (SURVEY 8b lists the callers for both precisions)."""
    from linalg_solver_amd import dense, gen

    n = 200
    A, _ = gen.system(gen.U11, 77, n)
    A32 = A.astype(np.float32)
    Ai, info, _ = dense.inv(A32, dtype=np.float32)
    assert info == 0 and Ai.dtype == np.float32
    assert np.max(np.abs(A32.astype(np.float64) @ Ai.astype(np.float64) - np.eye(n))) < 5e-3
    Ai64, _, _ = dense.inv(A32.astype(np.float64))
    assert relerr(Ai.astype(np.float64), Ai64) < 5e-3
    s32, l32 = dense.slogdet(A32, dtype=np.float32)
    s64, l64 = np.linalg.slogdet(A32.astype(np.float64))
    assert s32 == s64 and abs(l32 - l64) < 1e-3 * max(1.0, abs(l64))
    # rank: a rank-5 product, fp32 entries
    rng = np.random.default_rng(3)
    M = (rng.integers(-3, 4, (12, 5)) @ rng.integers(-3, 4, (5, 9))).astype(np.float32)
    R, piv, r = dense.rref(M, bar_col=9, dtype=np.float32, pivot_rule=la._native.PIVOT_MAX)
    assert r == np.linalg.matrix_rank(M.astype(np.float64)) == 5 and R.dtype == np.float32
    R64, piv64, r64 = dense.rref(M.astype(np.float64), bar_col=9, pivot_rule=la._native.PIVOT_MAX)
    assert piv == piv64 and relerr(R.astype(np.float64), R64) < 1e-4


def test_full_size_fp32_solution_within_1e4_of_the_fp64_solution(dev):
    """config 5 at its own size: 8192 x 8192 fp32 factors, 4 right-hand sides; the refined solution against the
    fp64 GPU solution of the same system (which test_full_size_lu_invariants holds to a 1e-9 backward error).
    The bound: BASELINE asks for 1e-4.  This test first asserted 1e-9 for the refined fp64-accumulated solution,
    which three sweeps miss at this size (5.97e-9 measured in round 2: each sweep gains about two digits from
    cond * eps32 ~ 1e-2); it was set to 1e-6 -- a hundred times inside the required tolerance -- rather than adding
    a fourth sweep to the default."""
    import torch

    from linalg_solver_amd import gen

    n, nrhs = 8192, 4
    A = torch.empty(n, n, dtype=torch.float32, device="cuda")
    B = torch.empty(n, nrhs, dtype=torch.float32, device="cuda")
    dev.fill_(A, gen.U11, 1)
    dev.fill_(B, gen.U11, 2)
    X64, X32, LU32, ipiv32, info, stats = dev.gesv_refined(A, B, sweeps=3)
    A64 = A.double()
    X = B.double().clone()
    ipiv, info64 = dev.getrf_(A64)
    dev.getrs_(A64, ipiv, X)
    torch.cuda.synchronize()
    assert int(info.item()) == 0 and int(info64.item()) == 0
    st = stats.cpu().numpy()
    # the unrefined fp32 solve, from the same factors
    X0 = B.clone()
    dev.getrs_(LU32, ipiv32, X0)
    torch.cuda.synchronize()
    scale = float(X.abs().max())
    err = float((X64 - X).abs().max()) / scale
    err32 = float((X32.double() - X).abs().max()) / scale
    err0 = float((X0.double() - X).abs().max()) / scale
    print(f"8192 fp32: forward error refined {err:.2e} (rounded to fp32 {err32:.2e}), unrefined {err0:.2e}; "
          f"last correction {st[0] / st[1]:.2e}")
    # three sweeps gain about two digits each here (cond * eps32 ~ 1e-2): 9e-3 -> 6e-9 measured
    assert err < 1e-6 and err32 < 1e-6 < TOL32 and err0 > err32
    assert st[0] / st[1] < 1e-5


def test_cooperative_solve_timeout_is_an_error_not_a_result(la):
    """a time-out of the few-RHS cooperative solve (x_k replaced by zero inside the kernel) must come
    back as LSX_ERR_INTERNAL from the host entry points, never as a solution.  Forced here by a spin limit of 0
    (the first unanswered poll gives up); the handle is usable again afterwards."""
    from linalg_solver_amd import _native, dense, gen

    h = la.default_handle()
    n = 3000
    A, b = gen.system(gen.U11, 31, n)
    LU, ipiv, info = dense.lu_factor(A)
    assert info == 0
    try:
        h.set_option("trsv_spin_limit", 0)
        for mode in (2, 1):
            h.set_option("trsv", mode)
            with pytest.raises(_native.LsxError):
                dense.lu_solve(LU, ipiv, b)
    finally:
        h.set_option("trsv_spin_limit", 1 << 20)
        h.set_option("trsv", 2)
    h.check_status()   # the word was cleared with the error
    x = dense.lu_solve(LU, ipiv, b)
    assert np.max(np.abs(A @ x - b)) / (np.max(np.abs(A)) * np.max(np.abs(x)) * n) < 1e-14


def test_getrf_dev_without_an_info_word_still_records_failures(dev):
    """*_dev entry points accept d_info = NULL; the factorisation then keeps an internal word that
    lsx_check_status reads.  (Here: nothing failed, the check passes and the factors are right.)"""
    import torch

    from linalg_solver_amd import _native, gen

    n = 700
    A0 = torch.empty(n, n, dtype=torch.float64, device="cuda")
    dev.fill_(A0, gen.U11, 12)
    LU = A0.clone()
    ipiv = torch.empty(n, dtype=torch.int32, device="cuda")
    _native.check(dev.lib.lsx_getrf_f64_dev(dev.h.ptr, n, LU.data_ptr(), n, ipiv.data_ptr(), None), "getrf_dev")
    dev.h.check_status()
    LU2 = A0.clone()
    ipiv2, info2 = dev.getrf_(LU2)
    torch.cuda.synchronize()
    assert int(info2.item()) == 0 and torch.equal(LU, LU2) and torch.equal(ipiv, ipiv2)


def test_panel_exchange_timeout_falls_back_to_per_column_launches(la):
    """VERDICT r1 item 6 / ADVICE: the cooperative panel polls across workgroups, so all of them must be resident at
    once.  (i) A filler holding 30 of the 32 CUs of the panel's XCD: the dispatcher places workgroups in order, the
    panel simply starts when the CUs are free -- right factors, no fall-back.  (ii) A participant that shows up late
    (fault injection: the last workgroup sleeps ~3 ms, spin limit 500): the others' bounded spins run out, the
    factorisation reports it, and the host entry point redoes it with panel mode 0 (no workgroup waits for another)
    -- right factors, one recorded fall-back, no hang and no LSX_ERR_INTERNAL."""
    import torch

    from linalg_solver_amd import dense, gen

    h = la.default_handle()
    n = 1500
    A, b = gen.system(gen.U11, 5, n)
    ref = dense.lu_factor(A)
    before = h.get_option("panel_fallbacks")
    h.occupy(0, 30, 200)
    LU, ipiv, info = dense.lu_factor(A)
    torch.cuda.synchronize()
    assert info == 0 and np.array_equal(ipiv, ref[1]) and np.array_equal(LU, ref[0])
    assert h.get_option("panel_fallbacks") == before
    try:
        h.set_option("panel_spin_limit", -500)
        LU, ipiv, info = dense.lu_factor(A)
        x, sinfo, _ = dense.solve(A, b)
    finally:
        h.set_option("panel_spin_limit", 1 << 20)
    assert info == 0 and np.array_equal(ipiv, ref[1]) and np.array_equal(LU, ref[0])
    assert sinfo == 0 and np.max(np.abs(A @ x - b)) / (np.max(np.abs(A)) * np.max(np.abs(x)) * n) < 1e-14
    assert h.get_option("panel_fallbacks") == before + 2
    # and without the fault the cooperative path is back
    LU2, ipiv2, info2 = dense.lu_factor(A)
    assert info2 == 0 and np.array_equal(LU2, ref[0]) and h.get_option("panel_fallbacks") == before + 2


@pytest.mark.parametrize("n", [2560, 5000, 9216])
def test_left_hand_interchanges_trailing_their_step_give_the_same_factors(la, dev, n):
    """The XCD look-ahead driver applies a panel's interchanges to the columns LEFT of it behind that step's update
    (option left_per_step, default) instead of in one launch at the end; update-bound steps with a wide left-hand side are
    caught up later in one launch (9216: the driver starts at column 1024 behind the shared-CU phase, ragged 5000: the
    event form of the chain).  Same interchanges in the same order on every column: same bits, same pivots."""
    import torch

    from linalg_solver_amd import gen

    h = dev.h
    A0 = torch.empty(n, n, dtype=torch.float64, device="cuda")
    dev.fill_(A0, gen.U11, 3 + n)
    res = []
    try:
        for v in (0, 1):
            h.set_option("left_per_step", v)
            A = A0.clone()
            ipiv, info = dev.getrf_(A)
            torch.cuda.synchronize()
            res.append((A, ipiv, int(info.item())))
    finally:
        h.set_option("left_per_step", 1)
    assert res[0][2] == res[1][2] == 0
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][0], res[1][0])


@pytest.mark.parametrize("n,dt", [(4096, "f64"), (7200, "f64"), (4096, "f32")])
def test_two_blocks_per_update_in_the_block_sweeps_keep_the_bits(la, dev, n, dt):
    """Inverse and many-right-hand-side solve at orders where the block substitution takes TWO 128-row blocks per
    trailing update (depth 256; going upwards the MFMA tile rotates its k index so that the block applied first is the
    one that comes second in memory): every entry has the bits of one block at a time (option getri_pairs = 0), also at
    a ragged order (edge kernels, a last block of 32 rows), and the inverse is one."""
    import torch

    from linalg_solver_amd import gen

    h = dev.h
    tdt = torch.float64 if dt == "f64" else torch.float32
    A = torch.empty(n, n, dtype=tdt, device="cuda")
    dev.fill_(A, gen.U11, 11)
    A0 = A.clone()
    ipiv, info = dev.getrf_(A)
    B0 = torch.empty(n, 1536, dtype=tdt, device="cuda")
    dev.fill_(B0, gen.U11, 12)
    outs = []
    try:
        for pairs in (0, 1):
            h.set_option("getri_pairs", pairs)
            inv = dev.getri(A, ipiv)
            X = B0.clone()
            dev.getrs_(A, ipiv, X)
            torch.cuda.synchronize()
            outs.append((inv, X))
    finally:
        h.set_option("getri_pairs", 1)
    assert int(info.item()) == 0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    tol = 1e-9 if dt == "f64" else 5e-2
    assert float((A0 @ outs[1][0] - torch.eye(n, dtype=tdt, device="cuda")).abs().max()) < tol
    assert float((A0 @ outs[1][1] - B0).abs().max()) < tol * 10


@pytest.mark.parametrize("wt", [0, 1])
def test_column_distributed_panel_is_a_second_implementation_with_the_same_bits(la, dev, wt):
    """kernels_panel_c.hip (option panel_col = 1): the XCD panel cut the other way -- a workgroup owns four COLUMNS of all
    rows, left-looking inside the panel, multipliers handed on through an all-ones-initialised buffer in the L2 -- shares no
    exchange code with kernels_panel_x.hip and must reproduce its factors, pivots, info and gather list bit for bit:
    ragged heights and widths, the reference's tie-rich integer distribution, a zero column (the slow path of the pivot
    search), one to sixteen rows per lane, plain and write-through stores (wt), and whole factorisations under the
    look-ahead driver (which hands panels above 4096 rows to the row-distributed kernel).  It is slower and off by
    default: this test is what it is kept for."""
    import torch

    from linalg_solver_amd import gen

    h = dev.h
    shapes = [(4096, 128), (4000, 100), (2500, 128), (2049, 5), (1025, 4), (1000, 3), (384, 128), (257, 100), (129, 128),
              (100, 17), (64, 1), (5, 5), (1, 1)]
    try:
        h.set_option("panel_col_wt", wt)
        for m, jb in shapes:
            for kind in (gen.U11, gen.INT5):
                P0 = torch.empty(m, jb, dtype=torch.float64, device="cuda")
                dev.fill_(P0, kind, 3)
                if kind == gen.INT5 and m >= 100:
                    P0[:, min(3, jb - 1)] = 0
                outs = []
                for pc in (0, 1):
                    h.set_option("panel_col", pc)
                    before = h.get_option("panel_col_launches")
                    P = P0.clone()
                    ipiv = torch.zeros(jb, dtype=torch.int32, device="cuda")
                    info = torch.zeros(1, dtype=torch.int32, device="cuda")
                    dev.panel_(P, 0, ipiv, info)
                    mv = torch.zeros(512, dtype=torch.int32, device="cuda")
                    assert dev.panel_moves_(mv)
                    torch.cuda.synchronize()
                    assert h.get_option("panel_col_launches") - before == pc, "the wrong kernel took the panel"
                    outs.append((P, ipiv, int(info.item()), mv))
                assert outs[0][2] == outs[1][2], (m, jb, kind)
                assert torch.equal(outs[0][1], outs[1][1]), (m, jb, kind)
                assert torch.equal(outs[0][0], outs[1][0]), (m, jb, kind)
                assert torch.equal(outs[0][3], outs[1][3]), (m, jb, kind)
        for n, dt in ((1000, torch.float64), (2304, torch.float64), (5000, torch.float64), (3000, torch.float32)):
            A0 = torch.empty(n, n, dtype=dt, device="cuda")
            dev.fill_(A0, gen.U11, 77 + n)
            res = []
            for pc in (0, 1):
                h.set_option("panel_col", pc)
                before = h.get_option("panel_col_launches")
                A = A0.clone()
                ipiv, info = dev.getrf_(A)
                torch.cuda.synchronize()
                took = h.get_option("panel_col_launches") - before
                assert (took > 0) == (pc == 1)
                res.append((A, ipiv, int(info.item())))
            assert res[0][2] == res[1][2] == 0 and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][0], res[1][0]), n
    finally:
        h.set_option("panel_col", 0)
        h.set_option("panel_col_wt", 0)


def test_chain_wait_timeout_reaches_info_and_the_host_entry_point_recovers(la, dev):
    """the look-ahead chain waits INSIDE a kernel for the previous update's first tile column (bounded
    spin).  A time-out there used to set only the status word and go on with stale columns: the device entry point
    must leave info < 0, lsx_check_status must report it once (and clear it), and the host-buffer entry point must
    hand back right factors through its per-column fallback -- not LSX_OK with garbage.  Fault injection: a wait
    limit of 0 polls."""
    import torch

    from linalg_solver_amd import _native, dense, gen

    h = la.default_handle()
    n = 2560   # look-ahead driver with the counted wait (n >= 2048, no ragged edge)
    A, b = gen.system(gen.U11, 21, n)
    ref = dense.lu_factor(A)
    before = h.get_option("panel_fallbacks")
    try:
        h.set_option("chain_wait_limit", 0)
        dev.h.set_option("chain_wait_limit", 0)
        LU, ipiv, info = dense.lu_factor(A)            # host buffers: falls back, right result
        dA = torch.from_numpy(A).cuda()
        dp, dinfo = dev.getrf_(dA)                     # device pointers: info < 0, nothing hidden
        torch.cuda.synchronize()
    finally:
        h.set_option("chain_wait_limit", 1 << 21)
        dev.h.set_option("chain_wait_limit", 1 << 21)
    assert info == 0 and np.array_equal(ipiv, ref[1]) and np.array_equal(LU, ref[0])
    assert h.get_option("panel_fallbacks") == before + 1
    h.check_status()                                   # the host entry point left no stale status word behind
    assert int(dinfo.item()) < 0
    with pytest.raises(_native.LsxError):
        dev.h.check_status()
    dev.h.check_status()                               # reported once, then clean
    dA = torch.from_numpy(A).cuda()
    dp, dinfo = dev.getrf_(dA)
    torch.cuda.synchronize()
    assert int(dinfo.item()) == 0 and np.array_equal(dA.cpu().numpy(), ref[0])


def test_from_dlpack_of_a_device_tensor_is_a_view_not_a_copy(la):
    """VERDICT r1 housekeeping: Matrix.from_dlpack of a tensor in HBM keeps the tensor; solve_array / inverse_array
    run through the *_dev entry points and return device tensors; the host copy appears only when asked for."""
    import torch

    from linalg_solver_amd import gen

    n = 500
    A_np, b_np = gen.system(gen.U11, 9, n)
    A = torch.from_numpy(A_np).cuda()
    M = la.Matrix.from_dlpack(A)
    assert M._dev is not None and M._dev.data_ptr() == A.data_ptr() and M._src is None and M._items is None
    assert (M.rows, M.cols) == (n, n)
    x = M.solve_array(torch.from_numpy(b_np).cuda())
    assert isinstance(x, torch.Tensor) and x.is_cuda and M._src is None
    assert relerr(x.cpu().numpy(), np.linalg.solve(A_np, b_np)) < TOL64
    Ai = M.inverse_array()
    assert isinstance(Ai, torch.Tensor) and Ai.is_cuda
    assert float((A @ Ai - torch.eye(n, dtype=torch.float64, device="cuda")).abs().max()) < 1e-9
    LU, ipiv, info = M.lu_device()
    assert int(info.item()) == 0 and torch.equal(A, torch.from_numpy(A_np).cuda())   # the view is not modified
    t = torch.from_dlpack(M)
    assert t.data_ptr() == A.data_ptr()
    S = A.clone()
    S[:, 3] = 0.0
    assert isinstance(la.Matrix.from_dlpack(S).solve_array(b_np), la.Matrix.NoSolution)
    assert M.items[2][3] == A_np[2, 3] and M._dev is None   # the lists, once handed out, are the only truth


@pytest.mark.parametrize("m,n,rank,bar", [(300, 300, 300, 300), (512, 700, 200, 600), (700, 512, 333, 512),
                                          (1000, 1000, 517, 999), (640, 900, 640, 900), (1030, 1030, 129, 1030)])
def test_blocked_rref_matches_the_unblocked_kernel(la, m, n, rank, bar):
    """SURVEY 8f item 1 at scale: the blocked rank-revealing row reduction (column skip inside 128-column blocks,
    MFMA trailing updates, blocked back substitution) against the per-column kernel with the same rule:
    identical pivot positions and rank, values to working precision, and the defining identity A = A[:, pc] R."""
    from linalg_solver_amd import _native, dense

    h = la.default_handle()
    rng = np.random.default_rng(m * 7 + n)
    # integer factors: the product is exact, so the rank is exactly `rank` (a float product of Gaussians sits at
    # the tolerance's edge: its "zero" columns carry the rounding noise of the product itself)
    A = (rng.integers(-3, 4, (m, rank)) @ rng.integers(-3, 4, (rank, n))).astype(np.float64)
    # plant zero and dependent columns so that skips happen inside blocks
    if n > 40:
        A[:, 5] = 0.0
        A[:, 17] = 2.0 * A[:, 3] - A[:, 11]
    try:
        h.set_option("rref_blocked", 0)
        R0, piv0, r0 = dense.rref(A, bar_col=bar, pivot_rule=_native.PIVOT_MAX)
        h.set_option("rref_blocked", 1)
        R1, piv1, r1 = dense.rref(A, bar_col=bar, pivot_rule=_native.PIVOT_MAX)
    finally:
        h.set_option("rref_blocked", 1)
    true_rank = np.linalg.matrix_rank(A[:, :bar])
    assert r0 == r1 == true_rank
    assert piv0 == piv1
    pc = [c for _, c in piv1]
    assert np.allclose(R1[:r1][:, pc], np.eye(r1), atol=0) and not R1[r1:, :bar].any()
    scale = max(1.0, np.max(np.abs(R0)))
    assert np.max(np.abs(R1 - R0)) / scale < 1e-8
    # row space: A[:, :bar] = A[:, pc] @ R[:r, :bar]
    assert np.max(np.abs(A[:, pc] @ R1[:r1, :bar] - A[:, :bar])) / np.max(np.abs(A)) < 1e-9


@pytest.mark.parametrize("m,n,rank,bar,swaps", [(300, 340, 40, 320, False), (300, 340, 40, 320, True), (520, 300, 130, 280, True),
                                                 (260, 600, 70, 600, False)])
def test_first_rule_blocked_reduction_matches_the_exact_reference_rule(la, m, n, rank, bar, swaps):
    """the reference's first-non-zero pivot rule (linalg.py:548-552) WITHOUT the per-column pass --
    pivot columns from the rank-revealing reduction, the reference's row choice from a blocked LU of those columns
    under the first-non-zero rule, carried-along columns and rows below the rank from one solve + one MFMA update.
    Checked against (i) the oracle's restatement of the reference run in exact rational arithmetic -- pivots equal,
    every entry within 1e-9, including the columns right of bar_col and the rows below the rank, which are the part
    that depends on the rule -- and (ii) the per-column kernels.  `swaps`: zero entries planted on the diagonal path so
    that the rule has to exchange rows (and a max-rule reduction would pick different rows everywhere anyway)."""
    from fractions import Fraction

    from linalg_solver_amd import _native, dense

    h = la.default_handle()
    rng = np.random.default_rng(m * 3 + n + rank)
    Bf = rng.integers(-2, 3, (m, rank))
    Cf = rng.integers(-2, 3, (rank, n))
    if swaps:
        Bf[0, :] = 0           # row 0 of the product is zero: the very first pivot needs an interchange
        Bf[5, :] = Bf[3, :]    # a duplicate row: becomes a zero row mid-way
        Bf[7, :] = 0
    Ai = Bf @ Cf
    A = Ai.astype(np.float64)
    try:
        h.set_option("rref_first_fast", 1)
        R1, piv1, r1 = dense.rref(A, bar_col=bar, pivot_rule=_native.PIVOT_FIRST)
        used = h.get_option("rref_first_used")
        h.set_option("rref_first_fast", 0)
        R0, piv0, r0 = dense.rref(A, bar_col=bar, pivot_rule=_native.PIVOT_FIRST)
    finally:
        h.set_option("rref_first_fast", 1)
    assert used == 1, "the blocked first-rule form did not run"
    exact, epiv, _ = rowreduce.row_reduce([[Fraction(int(v)) for v in row] for row in Ai], bar)
    E = np.array([[float(v) for v in row] for row in exact])
    assert piv1 == epiv == piv0 and r1 == len(epiv)
    scale = max(1.0, float(np.max(np.abs(E))))
    assert np.max(np.abs(R1 - E)) / scale < 1e-9, "blocked first-rule reduction differs from the exact reference run"
    assert np.max(np.abs(R0 - E)) / scale < 1e-9
    pc = [c for _, c in piv1]
    assert np.array_equal(R1[:r1][:, pc], np.eye(r1)) and not R1[r1:, :bar].any()


def test_first_rule_reduction_at_8192_rank_4096_through_the_matrix_surface(la):
    """The same at the size of bench.py's row-reduction line, through Matrix.row_reduce_array with bar_col < n: rank and
    pivots as planted, the defining identities, and the rule itself -- with the first 4096 rows independent the
    reference never exchanges rows, so the rows below the rank are (row i of A) - A[i, pc] R[:r], in the ORIGINAL row
    order, and the top rows are spanned by the first 4096 rows of A."""
    import torch

    from linalg_solver_amd import gen
    from linalg_solver_amd.device import DeviceSolver

    dev = DeviceSolver()
    n, rk, bar = 8192, 4096, 6000
    B = torch.empty(n, rk, dtype=torch.float64, device="cuda")
    C = torch.empty(rk, n - rk, dtype=torch.float64, device="cuda")
    dev.fill_(B, gen.U11, 3)
    dev.fill_(C, gen.U11, 4)
    A = torch.cat([B, B @ C / 64.0], dim=1).contiguous().cpu().numpy()
    import time
    t0 = time.perf_counter()
    R, piv = la.Matrix.from_numpy(A).row_reduce_array(bar_col=bar)
    dt = time.perf_counter() - t0
    assert la.default_handle().get_option("rref_first_used") == 1
    print(f"Matrix.row_reduce_array 8192^2 rank 4096 bar_col {bar}: {dt:.3f} s (host arrays in and out)")
    assert piv == [(k, k) for k in range(rk)]
    assert np.array_equal(R[:rk, :rk], np.eye(rk)) and not R[rk:, :bar].any()
    Cn = C.cpu().numpy() / 64.0
    assert np.max(np.abs(R[:rk, rk:] - Cn)) < 1e-7          # carried-along columns of the pivot rows included
    # rows below the rank, columns right of the bar: A2 - A2[:, pc] R_top with NO interchanges (first-rule semantics)
    want = A[rk:, bar:] - A[rk:, :rk] @ R[:rk, bar:]
    assert np.max(np.abs(R[rk:, bar:] - want)) < 1e-9 * max(1.0, np.max(np.abs(want)))
    assert dt < 3.0


def test_blocked_rref_8192_rank_4096(dev):
    """8192 x 8192 of rank 4096 on the device -- pivots against the planted structure (the first
    4096 columns are independent, every later column is a combination of them) and A = A[:, pc] R."""
    import torch

    from linalg_solver_amd import _native, gen

    n, rk = 8192, 4096
    B = torch.empty(n, rk, dtype=torch.float64, device="cuda")
    C = torch.empty(rk, n - rk, dtype=torch.float64, device="cuda")
    dev.fill_(B, gen.U11, 3)
    dev.fill_(C, gen.U11, 4)
    A = torch.cat([B, B @ C / 64.0], dim=1).contiguous()
    R = A.clone()
    piv, rank = dev.rref_(R, bar_col=n, pivot_rule=_native.PIVOT_MAX)
    torch.cuda.synchronize()
    r = int(rank.item())
    assert r == rk
    p = piv[:2 * r].view(r, 2).cpu().numpy()
    assert np.array_equal(p[:, 0], np.arange(rk)) and np.array_equal(p[:, 1], np.arange(rk))
    assert float(R[rk:, :].abs().max()) == 0.0
    assert torch.equal(R[:rk, :rk], torch.eye(rk, dtype=torch.float64, device="cuda"))
    # the reduced right block is the planted combination: R[:rk, rk:] = C / 64
    assert float((R[:rk, rk:] - C / 64.0).abs().max()) < 1e-7
    assert float((B @ R[:rk, :] - A).abs().max() / A.abs().max()) < 1e-9
