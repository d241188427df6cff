#!/usr/bin/env python3
"""Generate golden vectors by running the reference itself (build container only).

Imports /root/reference/linalg_solver read-only and records inputs + outputs of
Matrix.row_reduce / find_preimage_of / inverse as DATA fixtures:

  tests/golden/small_cases.json   typed scalars (int / float.hex / Fraction)
  tests/golden/cfg1_n64.npz       BASELINE config #1 (64x64, random.seed(2026))
  tests/golden/n128.npz n256.npz n512.npz   numeric-only larger cases

The reference's native helper (a Rust crate) cannot be built here, so an empty
stand-in module named ``linalg_helper`` is registered before the import; the
row-reduction path never touches it.  For N > 16 the
per-step LaTeX renderer is replaced by a no-op -- it produces strings only and
leaves every numeric result identical (checked below at N = 16).

Nothing of the reference is copied: fixtures hold inputs and expected outputs.
Run:  python tests/golden/gen_golden.py
"""
from __future__ import annotations

import json
import os
import random
import sys
import time
import types
from fractions import Fraction

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    stub = types.ModuleType("linalg_helper")
    stub.Permutation = type("Permutation", (), {})
    stub.RowColPermutation = type("RowColPermutation", (), {})
    sys.modules["linalg_helper"] = stub
    sys.path.insert(0, REF)
    import linalg_solver  # noqa: F401
    import linalg_solver.linalg as L
    from linalg_solver.log import global_logger

    global_logger._auto_print = False
    return L


L = load_reference()
Matrix = L.Matrix
_real_latex = L.make_latex_augmented_matrix


def quiet_latex(on: bool):
    L.make_latex_augmented_matrix = (lambda *a, **k: "") if on else _real_latex


# ---------------------------------------------------------------- encoding
def enc(x):
    if isinstance(x, bool):
        raise TypeError("bool entry")
    if isinstance(x, int):
        return ["i", x]
    if isinstance(x, float):
        return ["f", x.hex()]
    if isinstance(x, Fraction):
        return ["q", x.numerator, x.denominator]
    raise TypeError(f"cannot encode {type(x)}")


def enc_mat(M):
    return [[enc(v) for v in row] for row in M]


def enc_result(res):
    if isinstance(res, Matrix.NoSolution):
        return {"kind": "NoSolution"}
    if isinstance(res, Matrix.AffineSubspace):
        g = res.generators
        return {
            "kind": "AffineSubspace",
            "particular": [enc(v) for v in res.vec],
            "generators": None if g is None else enc_mat(g.items),
        }
    if isinstance(res, Matrix):
        return {"kind": "Matrix", "items": enc_mat(res.items)}
    raise TypeError(type(res))


# ---------------------------------------------------------------- case runners
def case_row_reduce(name, items, bar_col=None):
    red, pivots, mats, steps = Matrix(items).row_reduce(bar_col=bar_col)
    return {
        "name": name,
        "op": "row_reduce",
        "items": enc_mat(items),
        "bar_col": bar_col,
        "reduced": enc_mat(red),
        "pivots": [list(p) for p in pivots],
        "steps": [[lab, txt] for lab, txt in steps],
        "n_intermediate": len(mats),
    }


def case_preimage(name, items, vec):
    res = Matrix([list(r) for r in items]).find_preimage_of(list(vec), log_steps=True)
    return {"name": name, "op": "find_preimage_of", "items": enc_mat(items),
            "vec": [enc(v) for v in vec], "result": enc_result(res)}


def case_inverse(name, items):
    res = Matrix([list(r) for r in items]).inverse(log_steps=True)
    return {"name": name, "op": "inverse", "items": enc_mat(items), "result": enc_result(res)}


def rand_int_matrix(rng, m, n, as_float):
    conv = float if as_float else int
    return [[conv(rng.randint(-5, 5)) for _ in range(n)] for _ in range(m)]


def small_cases():
    out = []
    quiet_latex(False)
    # random square systems, the reference's own distribution (random_matrix.py:104)
    for n in (1, 2, 3, 4, 8, 16):
        for seed in (2026, 1, 2, 3, 4, 5):
            rng = random.Random(seed * 1000 + n)
            for as_float in (True, False):
                tag = "f" if as_float else "i"
                A = rand_int_matrix(rng, n, n, as_float)
                b = [float(rng.randint(-5, 5)) if as_float else rng.randint(-5, 5) for _ in range(n)]
                aug = [r + [b[i]] for i, r in enumerate(A)]
                if n >= 2 or True:
                    out.append(case_row_reduce(f"int5_{tag}_n{n}_s{seed}_aug", aug))
                out.append(case_preimage(f"int5_{tag}_n{n}_s{seed}_solve", A, b))
                if seed in (2026, 1):
                    out.append(case_inverse(f"int5_{tag}_n{n}_s{seed}_inv", A))
    # uniform (-1,1) floats
    for n in (2, 3, 4, 8, 16):
        for seed in (11, 12):
            rng = random.Random(seed * 77 + n)
            A = [[rng.uniform(-1, 1) for _ in range(n)] for _ in range(n)]
            b = [rng.uniform(-1, 1) for _ in range(n)]
            out.append(case_row_reduce(f"u11_n{n}_s{seed}_aug", [r + [b[i]] for i, r in enumerate(A)]))
            out.append(case_preimage(f"u11_n{n}_s{seed}_solve", A, b))
            out.append(case_inverse(f"u11_n{n}_s{seed}_inv", A))
    # structural edge cases
    out.append(case_row_reduce("leading_zero_pivot", [[0.0, 2.0, 1.0], [1.0, 1.0, 3.0], [2.0, 0.0, 5.0]], 2))
    out.append(case_row_reduce("zero_column", [[0, 0, 1], [0, 2, 1], [0, 4, 2]]))
    out.append(case_row_reduce("zero_column_f", [[0.0, 0.0, 1.0], [0.0, 2.0, 1.0], [0.0, 4.0, 2.0]]))
    out.append(case_row_reduce("wide_inconsistent", [[1, 2, 3, 4], [2, 4, 6, 9]]))
    out.append(case_preimage("wide_inconsistent_solve", [[1, 2, 3], [2, 4, 6]], [4, 9]))
    out.append(case_preimage("wide_inconsistent_solve_f", [[1.0, 2.0, 3.0], [2.0, 4.0, 6.0]], [4.0, 9.0]))
    out.append(case_row_reduce("tall_4x2", [[1.0, 2.0], [3.0, 4.0], [5.0, 6.0], [7.0, 9.0]]))
    out.append(case_row_reduce("tall_4x3", [[2.0, 1.0, 3.0], [4.0, 2.0, 6.0], [1.0, 0.0, 1.0], [3.0, 1.0, 5.0]]))
    out.append(case_preimage("tall_consistent", [[1.0, 2.0], [3.0, 4.0], [4.0, 6.0]], [3.0, 7.0, 10.0]))
    out.append(case_preimage("tall_inconsistent", [[1.0, 2.0], [3.0, 4.0], [4.0, 6.0]], [3.0, 7.0, 11.0]))
    out.append(case_preimage("underdetermined", [[1, 1, 0], [0, 0, 1]], [3, 4]))
    out.append(case_preimage("underdetermined_f", [[1.0, 1.0, 0.0], [0.0, 0.0, 1.0]], [3.0, 4.0]))
    out.append(case_preimage("kernel_3x4", [[1.0, 2.0, 3.0, 4.0], [2.0, 4.0, 6.0, 8.0], [1.0, 0.0, 1.0, 0.0]], [0.0, 0.0, 0.0]))
    for bc in (None, 0, 1, 2):
        out.append(case_row_reduce(f"barcol_{bc}", [[1.0, 2.0], [3.0, 4.0]], bc))
    for bc in (None, 0, 2, 3, 4):
        out.append(case_row_reduce(f"barcol4_{bc}", [[2.0, 1.0, 1.0, 5.0], [4.0, -6.0, 0.0, -2.0], [-2.0, 7.0, 2.0, 9.0]], bc))
    out.append(case_row_reduce("identity3", [[1.0, 0.0, 0.0, 2.0], [0.0, 1.0, 0.0, 3.0], [0.0, 0.0, 1.0, 4.0]]))
    out.append(case_row_reduce("identity3_int", [[1, 0, 0, 2], [0, 1, 0, 3], [0, 0, 1, 4]]))
    out.append(case_inverse("singular_inv", [[1.0, 2.0], [2.0, 4.0]]))
    out.append(case_inverse("singular_inv_int", [[1, 2, 3], [4, 5, 6], [7, 8, 9]]))
    out.append(case_inverse("inv_1x1", [[4.0]]))
    out.append(case_inverse("inv_needs_swap", [[0.0, 1.0], [1.0, 0.0]]))
    out.append(case_row_reduce("one_by_two", [[3.0, 6.0]]))
    out.append(case_row_reduce("all_zero", [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0]]))
    out.append(case_preimage("all_zero_consistent", [[0.0, 0.0], [0.0, 0.0]], [0.0, 0.0]))
    out.append(case_preimage("all_zero_inconsistent", [[0.0, 0.0], [0.0, 0.0]], [0.0, 1.0]))
    # rank-deficient: float entries vs exact entries (documents the float-rank artefact)
    for n, r, seed in ((8, 5, 7), (6, 3, 8), (12, 4, 9)):
        rng = random.Random(seed)
        P = [[rng.randint(-5, 5) for _ in range(r)] for _ in range(n)]
        Q = [[rng.randint(-5, 5) for _ in range(n)] for _ in range(r)]
        prod = [[sum(P[i][k] * Q[k][j] for k in range(r)) for j in range(n)] for i in range(n)]
        b = [sum(prod[i][j] * ((j % 3) - 1) for j in range(n)) for i in range(n)]  # consistent rhs
        for tag, conv in (("float", float), ("frac", Fraction)):
            A = [[conv(v) for v in row] for row in prod]
            bb = [conv(v) for v in b]
            out.append(case_row_reduce(f"rankdef_{tag}_n{n}_r{r}", [row + [bb[i]] for i, row in enumerate(A)]))
            out.append(case_preimage(f"rankdef_{tag}_n{n}_r{r}_solve", A, bb))
    # exact rational full-rank systems
    for n, seed in ((3, 21), (5, 22), (8, 23)):
        rng = random.Random(seed)
        A = [[Fraction(rng.randint(-5, 5), rng.randint(1, 4)) for _ in range(n)] for _ in range(n)]
        b = [Fraction(rng.randint(-5, 5)) for _ in range(n)]
        out.append(case_preimage(f"frac_n{n}_s{seed}_solve", A, b))
        out.append(case_inverse(f"frac_n{n}_s{seed}_inv", A))
    return out


def check_quiet_latex_is_harmless():
    rng = random.Random(99)
    A = [[float(rng.randint(-5, 5)) for _ in range(17)] for _ in range(16)]
    quiet_latex(False)
    r1 = Matrix(A).row_reduce()
    quiet_latex(True)
    r2 = Matrix(A).row_reduce()
    assert r1[0] == r2[0] and r1[1] == r2[1] and r1[3] == r2[3]
    assert [[v.hex() for v in row] for row in r1[0]] == [[v.hex() for v in row] for row in r2[0]]


def big_inputs(n, seed, stream):
    """The survey's input stream: random.seed(seed); A row-major, then b.
    tests/helpers.py holds the same eight lines so inputs need not be stored."""
    random.seed(seed)
    if stream == "int5":
        A = [[float(random.randint(-5, 5)) for _ in range(n)] for _ in range(n)]
        b = [float(random.randint(-5, 5)) for _ in range(n)]
    else:
        A = [[random.uniform(-1, 1) for _ in range(n)] for _ in range(n)]
        b = [random.uniform(-1, 1) for _ in range(n)]
    return A, b


def big_case(n, seed, stream):
    A, b = big_inputs(n, seed, stream)
    quiet_latex(True)
    aug = [r + [b[i]] for i, r in enumerate(A)]
    t0 = time.perf_counter()
    red, pivots, _, steps = Matrix(aug).row_reduce()
    t_rr = time.perf_counter() - t0
    sol = Matrix([list(r) for r in A]).find_preimage_of(list(b), log_steps=True)
    assert isinstance(sol, Matrix.AffineSubspace) and sol.generators is None
    return (np.array(A), np.array(b), np.array(red, dtype=np.float64), pivots, steps,
            np.array(sol.vec, dtype=np.float64), t_rr)


def main():
    check_quiet_latex_is_harmless()
    cases = small_cases()
    with open(os.path.join(HERE, "small_cases.json"), "w") as f:
        json.dump({"generator": "tests/golden/gen_golden.py", "reference": "koskja/linalg-solver",
                   "cases": cases}, f, separators=(",", ":"))
    print(f"small_cases.json: {len(cases)} cases")

    meta = {}
    # BASELINE config #1: 64x64, random.seed(2026), int[-5,5] as floats
    A, b, red, pivots, steps, x, t = big_case(64, 2026, "int5")
    quiet_latex(True)
    inv = Matrix(A.tolist()).inverse(log_steps=True)
    assert isinstance(inv, Matrix)
    np.savez_compressed(os.path.join(HERE, "cfg1_n64.npz"), A=A, b=b, reduced=red,
                        pivots=np.array(pivots, dtype=np.int32), x=x,
                        inverse=np.array(inv.items, dtype=np.float64),
                        labels=np.array([s[0] for s in steps]))
    meta["cfg1_n64"] = {"row_reduce_numeric_only_s": t, "steps": len(steps)}
    for n, stream in ((128, "int5"), (128, "u11"), (256, "int5"), (256, "u11"), (512, "u11")):
        A, b, red, pivots, steps, x, t = big_case(n, n, stream)
        name = f"n{n}_{stream}"
        # inputs are NOT stored: tests rebuild them from random.seed(n) (see big_inputs)
        keep = {"x": x, "pivots": np.array(pivots, dtype=np.int32),
                "labels": np.array([s[0] for s in steps]),
                "a_checksum": np.array([A.sum(), np.abs(A).sum(), b.sum()])}
        if n <= 128:
            keep["reduced"] = red
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **keep)
        meta[name] = {"row_reduce_numeric_only_s": t, "steps": len(steps),
                      "swaps": sum(1 for s in steps if s[0].startswith("S"))}
        print(name, meta[name])
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump({"note": "reference timings, 1 core of the build container, LaTeX renderer disabled",
                   "cases": meta}, f, indent=1)


if __name__ == "__main__":
    main()
