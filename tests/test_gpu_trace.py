"""Traced row reduction on the GPU (lsx_rref_trace_f64): the reference's own operation order with its
step log.  Everything here is compared BIT FOR BIT with vectors captured from the reference."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from helpers import dec, dec_mat, is_numeric_case, load_big, load_small_cases, same_matrix, same_scalar  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def la():
    import linalg_solver_amd as la_
    return la_


def _latex_cases():
    with open(os.path.join(HERE, "golden", "latex_cases.json")) as f:
        return json.load(f)["cases"]


def _bits(M):
    return [[float(v).hex() for v in row] for row in M]


@pytest.mark.parametrize("case", _latex_cases(), ids=[c["name"] for c in _latex_cases()])
def test_traced_row_reduce_reproduces_the_reference_log(la, case):
    items = [[float.fromhex(v) for v in row] for row in case["items"]]
    red, pivots, mats, steps = la.Matrix(items).row_reduce(case["bar_col"], trace=True)
    assert _bits(red) == case["reduced"]
    assert [list(p) for p in pivots] == case["pivots"]
    assert [list(s) for s in steps] == case["steps"]
    assert mats == case["matrices"]
    red2, piv2, mats2, steps2 = la.Matrix(items).row_reduce(case["bar_col"], trace="steps")
    assert _bits(red2) == case["reduced"] and piv2 == pivots and steps2 == steps and mats2 == []


RR = [c for c in load_small_cases() if c["op"] == "row_reduce" and is_numeric_case(dec_mat(c["items"]))]


@pytest.mark.parametrize("case", RR, ids=[c["name"] for c in RR])
def test_traced_row_reduce_small_golden(la, case):
    """All numeric small cases (int and float entries): every entry identical in TYPE and bits (untouched
    ints stay ints) -- including the reference's float rank artefacts -- pivots
    and (label, description) steps identical."""
    items = dec_mat(case["items"])
    red, pivots, _, steps = la.Matrix(items).row_reduce(case["bar_col"], trace="steps")
    want = dec_mat(case["reduced"])
    assert same_matrix(red, want), (red, want)
    assert [list(p) for p in pivots] == case["pivots"]
    assert [list(s) for s in steps] == case["steps"]


def _same_result(la, got, want) -> bool:
    """Typed, bit-level comparison of a caller result with the reference's."""
    if want["kind"] == "NoSolution":
        return isinstance(got, la.Matrix.NoSolution)
    if want["kind"] == "AffineSubspace":
        if not isinstance(got, la.Matrix.AffineSubspace):
            return False
        part = [dec(v) for v in want["particular"]]
        if len(part) != len(got.vec) or not all(same_scalar(a, b) for a, b in zip(got.vec, part)):
            return False
        if want["generators"] is None:
            return got.generators is None
        return got.generators is not None and same_matrix(got.generators.items, dec_mat(want["generators"]))
    return isinstance(got, la.Matrix) and same_matrix(got.items, dec_mat(want["items"]))


OTHER = [c for c in load_small_cases() if c["op"] in ("find_preimage_of", "inverse")
         and is_numeric_case(dec_mat(c["items"]))]


@pytest.mark.parametrize("case", OTHER, ids=[c["name"] for c in OTHER])
def test_traced_solve_and_inverse_small_golden(la, case):
    items = dec_mat(case["items"])
    M = la.Matrix(items)
    if case["op"] == "find_preimage_of":
        got = M.find_preimage_of([dec(v) for v in case["vec"]], log_steps=True, trace="steps")
    else:
        got = M.inverse(log_steps=True, trace="steps")
    assert _same_result(la, got, case["result"]), (got, case["result"])
    assert isinstance(M.last_trace, tuple) and len(M.last_trace) == 2


def test_cfg1_64x64_traced_is_bit_identical(la):
    z = np.load(os.path.join(HERE, "golden", "cfg1_n64.npz"))
    A, b = z["A"], z["b"]
    red, pivots, mats, steps = la.Matrix(np.hstack([A, b[:, None]]).tolist()).row_reduce(trace=True)
    assert np.array_equal(np.array(red), z["reduced"])
    assert [list(p) for p in pivots] == z["pivots"].tolist()
    assert [s[0] for s in steps] == z["labels"].tolist()
    assert len(mats) == len(steps) + 1 and mats[0].startswith(r"\left(\begin{array}{" + "c" * 64 + "|c}")
    sol = la.Matrix(A.tolist()).find_preimage_of(b.tolist(), trace="steps")
    assert np.array_equal(np.array(sol.vec), z["x"]) and sol.generators is None
    inv = la.Matrix(A.tolist()).inverse(trace="steps")
    assert np.array_equal(np.array(inv.items), z["inverse"])


@pytest.mark.parametrize("name", ["n128_int5", "n128_u11", "n256_u11", "n512_u11"])
def test_traced_larger_n_is_bit_identical(la, name):
    A, b, z = load_big(name)
    red, pivots, _, steps = la.Matrix(np.hstack([A, b[:, None]]).tolist()).row_reduce(trace="steps")
    assert [s[0] for s in steps] == z["labels"].tolist()
    assert np.array_equal(np.array(red)[:, -1], z["x"])
    n = A.shape[0]
    assert pivots == [(k, k) for k in range(n)]


def test_trace_log_overflow_and_arguments(la):
    from linalg_solver_amd import dense
    R, piv, recs, touched, snaps, snap_t = dense.rref_trace(np.array([[0.0, 0.0], [0.0, 0.0]]), bar_col=2)
    assert piv == [] and recs == [] and snaps is None and snap_t is None and not touched.any()
    assert np.array_equal(R, np.zeros((2, 2)))
    with pytest.raises(ValueError):
        dense.rref_trace(np.zeros((0, 3)))


def _builder_cases():
    with open(os.path.join(HERE, "golden", "latex_cases.json")) as f:
        return json.load(f)["builder"]


@pytest.mark.parametrize("case", _builder_cases(), ids=[f"seed{c['seed']}" for c in _builder_cases()])
def test_random_matrix_builders_follow_the_reference_stream(la, case):
    """random_matrix.py:109-130 with the rank tests and the product on the GPU: same seed, same matrices."""
    import random
    random.seed(case["seed"])
    reg = la.gen_regular_matrix(6)
    rk = la.gen_matrix_with_rank(5, 7, 3)
    assert same_matrix(reg.items, case["regular6"])
    assert same_matrix(rk.items, case["rank3_5x7"])
    assert reg.rank() == 6 and rk.rank() == 3
    with pytest.raises(NotImplementedError):
        la.RandomMatrixBuilder.new().with_eigenvalues([1.0, 2.0])


def test_array_backed_matrix_and_dlpack(la):
    """from_numpy / from_dlpack keep the array (no list-of-lists until `.items` is read); the array-valued
    twins agree with the list-valued API."""
    import torch
    rng = np.random.default_rng(5)
    A = rng.uniform(-1, 1, (300, 300))
    b = rng.uniform(-1, 1, 300)
    M = la.Matrix.from_numpy(A)
    assert M._items is None and M.rows == 300 and M.cols == 300
    x = M.solve_array(b)
    assert M._items is None, "the numeric path must not materialise the Python lists"
    assert np.max(np.abs(A @ x - b)) < 1e-9
    inv = M.inverse_array()
    assert np.max(np.abs(A @ inv - np.eye(300))) < 1e-8
    R, piv = la.Matrix.from_numpy(np.hstack([A, b[:, None]])).row_reduce_array()
    assert piv == [(k, k) for k in range(300)] and np.max(np.abs(R[:, -1] - x)) < 1e-9
    sol = M.find_preimage_of(b.tolist())
    assert np.max(np.abs(np.array(sol.vec) - x)) < 1e-12
    assert M.items[0][0] == A[0, 0] and M._src is None          # lists on demand; then they are the truth
    M.items[0][0] = 2.0
    assert M.to_numpy()[0, 0] == 2.0
    T = torch.tensor(A, device="cuda")
    Md = la.Matrix.from_dlpack(T)
    assert np.array_equal(Md.to_numpy(), A)
    Mi = la.Matrix.from_numpy(np.array([[1, 2], [3, 4]]))
    assert Mi.items == [[1, 2], [3, 4]] and all(isinstance(v, int) for v in Mi.items[0])
    assert np.array_equal(np.from_dlpack(Mi), np.array([[1.0, 2.0], [3.0, 4.0]]))
    singular = la.Matrix.from_numpy(np.ones((4, 4)))
    assert isinstance(singular.inverse_array(), la.Matrix.NoSolution)
    assert isinstance(singular.solve_array(np.ones(4)), la.Matrix.NoSolution)
    with pytest.raises(TypeError):
        la.Matrix.from_numpy(np.array([["a", "b"]]))
