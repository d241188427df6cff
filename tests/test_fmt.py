"""The LaTeX formatter (linalg_solver_amd/fmt.py) against sympy.latex -- what the reference's
fmt.cformat calls for every entry (fmt.py:26) -- and against strings captured from the reference."""
import importlib.util
import json
import math
import os
import random
import struct

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("lsx_fmt", os.path.join(HERE, "..", "linalg_solver_amd", "fmt.py"))
fmt = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(fmt)  # plain Python module: loadable without the HIP library


def _cases():
    with open(os.path.join(HERE, "golden", "latex_cases.json")) as f:
        return json.load(f)["cases"]


def test_latex_float_matches_sympy():
    sympy = pytest.importorskip("sympy")
    rng = random.Random(11)
    vals = [0.5, 1e-17, -2.0, 1 / 3, 100.0, 1e15, 1e16, 1e14, 0.0001, 1e-5, 1.5e-5, -0.0, 0.0, 0.1 + 0.2, 2.675, 1e22,
            5e-324, 1234567.891, 999999999999999.9, 99999999999999.99, 9.999999999999999e-5, 1.7976931348623157e308,
            2.0 ** -22, 3 * 2.0 ** -22, 448385270955718.5, -4553657109316065.0, 999999999999999.5, float("inf"),
            float("-inf"), float("nan")]
    vals += [rng.uniform(-1, 1) for _ in range(1500)]
    vals += [rng.uniform(-5, 5) * 10 ** rng.randint(-20, 20) for _ in range(800)]
    vals += [rng.randint(-5, 5) / rng.randint(1, 9) for _ in range(400)]
    vals += [m * 2.0 ** k for k in range(-40, 40) for m in (1, 3, 5, 25, 125)]   # exact ties live here
    for _ in range(800):
        v = struct.unpack("<d", struct.pack("<Q", rng.getrandbits(64)))[0]
        if v == v and abs(v) != math.inf:
            vals.append(v)
    bad = [(v, fmt.latex_float(v), sympy.latex(v)) for v in vals if fmt.latex_float(v) != sympy.latex(v)]
    assert not bad, bad[:5]


def test_cformat_entry_types():
    assert fmt.cformat(3) == "3" and fmt.cformat(-3.0) == "-3.0" and fmt.cformat("x") == "x"
    assert fmt.cformat(True) == r"\text{True}"

    class WithCformat:
        def cformat(self, arg_of=None):
            return "custom"
    assert fmt.cformat(WithCformat()) == "custom"


@pytest.mark.parametrize("case", _cases(), ids=[c["name"] for c in _cases()])
def test_matrices_match_the_reference_strings(case):
    """First intermediate matrix = the input (linalg.py:544), last one = the reduced matrix."""
    items = [[float.fromhex(v) for v in row] for row in case["items"]]
    n = len(items[0])
    bar = case["bar_col"] or n - 1
    assert fmt.make_latex_augmented_matrix(items, bar_col=bar) == case["matrices"][0]
    if case["steps"]:
        red = [[float.fromhex(v) for v in row] for row in case["reduced"]]
        assert fmt.make_latex_augmented_matrix(red, bar_col=bar) == case["matrices"][-1]
    assert len(case["matrices"]) == len(case["steps"]) + 1


def test_vector_and_plain_matrix_layout():
    assert fmt.make_latex_vector([1.0, 2.5]) == "\\begin{pmatrix}1.0\\\\[0.1em]\n2.5\\end{pmatrix}"
    assert fmt.make_latex_matrix([[1.0, 2.0], [3.0, 4.0]]) == \
        "\\begin{pmatrix}1.0 & 2.0\\\\[0.1em]\n3.0 & 4.0\\end{pmatrix}"
    assert fmt.make_latex_augmented_matrix([[2.0], [4.0]]) == fmt.make_latex_matrix([[2.0], [4.0]])
