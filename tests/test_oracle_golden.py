"""The CPU oracle against outputs of the reference itself (tests/golden)."""
import numpy as np
import pytest

from oracle import capi, rowreduce
from helpers import (dec, dec_mat, is_numeric_case, load_big, load_small_cases, same_matrix,
                     same_scalar)

CASES = load_small_cases()
RR = [c for c in CASES if c["op"] == "row_reduce"]
PRE = [c for c in CASES if c["op"] == "find_preimage_of"]
INV = [c for c in CASES if c["op"] == "inverse"]


def test_fixture_inventory():
    assert len(RR) > 100 and len(PRE) > 90 and len(INV) > 30
    assert any(c["steps"] and c["steps"][0][0].startswith("S") for c in RR), "no swap case"


@pytest.mark.parametrize("case", RR, ids=[c["name"] for c in RR])
def test_python_restatement_row_reduce_bit_exact(case):
    items = dec_mat(case["items"])
    red, pivots, steps = rowreduce.row_reduce(items, case["bar_col"])
    assert same_matrix(red, dec_mat(case["reduced"]))
    assert [list(p) for p in pivots] == case["pivots"]
    assert [[rowreduce.step_label(s), rowreduce.step_text(s)] for s in steps] == case["steps"]
    assert case["n_intermediate"] == len(steps) + 1  # linalg.py:544 initial matrix + one per step


@pytest.mark.parametrize("case", [c for c in RR if is_numeric_case(dec_mat(c["items"]))],
                         ids=[c["name"] for c in RR if is_numeric_case(dec_mat(c["items"]))])
def test_c_restatement_row_reduce_bit_exact(case):
    items = dec_mat(case["items"])
    A = np.array(items, dtype=np.float64)
    red, pivots, steps = capi.row_reduce(A, case["bar_col"])
    want = np.array([[float(v) for v in row] for row in dec_mat(case["reduced"])])
    # values bit-exact (the C port has no int/float type distinction; -0.0 vs 0 differ only by type there)
    assert red.shape == want.shape
    assert np.array_equal(red, want)
    neg = np.signbit(red) != np.signbit(want)
    ints = np.array([[isinstance(v, int) for v in row] for row in dec_mat(case["reduced"])])
    assert not np.any(neg & ~ints)
    assert [list(p) for p in pivots] == case["pivots"]
    assert [rowreduce.step_label(s) for s in steps] == [s[0] for s in case["steps"]]
    assert [rowreduce.step_text(s) for s in steps] == [s[1] for s in case["steps"]]


def _check_result(got, want):
    if want["kind"] == "NoSolution":
        assert got == rowreduce.NO_SOLUTION
        return
    assert got != rowreduce.NO_SOLUTION
    if want["kind"] == "AffineSubspace":
        part, gens, _ = got
        wp = [dec(v) for v in want["particular"]]
        assert len(part) == len(wp) and all(same_scalar(a, b) for a, b in zip(part, wp))
        if want["generators"] is None:
            assert gens is None
        else:
            assert same_matrix(gens, dec_mat(want["generators"]))
    else:
        assert same_matrix(got, dec_mat(want["items"]))


@pytest.mark.parametrize("case", PRE, ids=[c["name"] for c in PRE])
def test_python_restatement_find_preimage(case):
    got = rowreduce.find_preimage_of(dec_mat(case["items"]), [dec(v) for v in case["vec"]])
    _check_result(got, case["result"])


@pytest.mark.parametrize("case", INV, ids=[c["name"] for c in INV])
def test_python_restatement_inverse(case):
    _check_result(rowreduce.inverse(dec_mat(case["items"])), case["result"])


def test_cfg1_64_bit_exact(golden_dir):
    z = np.load(golden_dir + "/cfg1_n64.npz")
    aug = np.hstack([z["A"], z["b"][:, None]])
    red, pivots, steps = capi.row_reduce(aug)
    assert np.array_equal(red, z["reduced"])
    assert pivots == [tuple(p) for p in z["pivots"].tolist()] == [(k, k) for k in range(64)]
    assert [rowreduce.step_label(s) for s in steps] == z["labels"].tolist()
    assert np.array_equal(red[:, 64], z["x"])
    # inverse through [A|I] with bar_col = n (linalg.py:704-711)
    augI = np.hstack([z["A"], np.eye(64)])
    redI, _, _ = capi.row_reduce(augI, 64)
    assert capi.left_is_identity(redI, 64)
    assert np.array_equal(redI[:, 64:], z["inverse"])
    # python restatement agrees too
    red_py, piv_py, _ = rowreduce.row_reduce(aug.tolist())
    assert np.array_equal(np.array(red_py), z["reduced"]) and piv_py == pivots


@pytest.mark.parametrize("name", ["n128_int5", "n128_u11", "n256_int5", "n256_u11", "n512_u11"])
def test_c_restatement_large_bit_exact(name):
    A, b, z = load_big(name)
    n = A.shape[0]
    red, pivots, steps = capi.row_reduce(np.hstack([A, b[:, None]]))
    assert np.array_equal(red[:, n], z["x"])
    assert pivots == [tuple(p) for p in z["pivots"].tolist()]
    assert [rowreduce.step_label(s) for s in steps] == z["labels"].tolist()
    if "reduced" in z:
        assert np.array_equal(red, z["reduced"])


def test_lu_twin_against_reference_solution():
    """The partial-pivot twin reaches the reference's solution (RREF is unique)."""
    for name, tol in (("n128_u11", 1e-10), ("n256_int5", 1e-9), ("n512_u11", 1e-8)):
        A, b, z = load_big(name)
        LU, ipiv, info = capi.getrf(A)
        assert info == 0
        x = capi.getrs(LU, ipiv, b)
        # the reference is unpivoted, so it is the less accurate side here
        assert np.max(np.abs(x - z["x"])) / np.max(np.abs(z["x"])) < tol
        assert np.max(np.abs(np.tril(LU, -1))) <= 1.0
        s, l = capi.slogdet(LU, ipiv)
        s2, l2 = np.linalg.slogdet(A)
        assert s == s2 and abs(l - l2) < 1e-9 * abs(l2)
