#!/usr/bin/env python3
"""LaTeX golden vectors: the reference's own `intermediate_matrices` strings (build container only).

Runs the reference's Matrix.row_reduce with its real LaTeX renderer (fmt.make_latex_augmented_matrix
through sympy.latex) on a few small all-float inputs and records inputs + the returned strings and
step list as data:  tests/golden/latex_cases.json.  Same import arrangement as gen_golden.py.
Run:  python tests/golden/gen_latex_golden.py
"""
from __future__ import annotations

import json
import os
import random
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402  (loads the reference read-only)

Matrix = G.Matrix


def case(name, items, bar_col=None):
    red, pivots, mats, steps = Matrix([list(r) for r in items]).row_reduce(bar_col=bar_col)
    return {"name": name, "items": [[float(v).hex() for v in r] for r in items], "bar_col": bar_col,
            "reduced": [[float(v).hex() for v in r] for r in red], "pivots": [list(p) for p in pivots],
            "steps": [[a, b] for a, b in steps], "matrices": mats}


def main():
    G.quiet_latex(False)
    rng = random.Random(7)
    cases = []
    for n in (2, 3, 4, 6):
        cases.append(case(f"u11_n{n}_aug", [[rng.uniform(-1, 1) for _ in range(n + 1)] for _ in range(n)]))
        cases.append(case(f"int5f_n{n}_aug", [[float(rng.randint(-5, 5)) for _ in range(n + 1)] for _ in range(n)]))
    cases.append(case("leading_zero", [[0.0, 2.0, 1.0], [3.0, 1.0, 2.0], [0.0, 4.0, 2.0]]))
    cases.append(case("zero_column", [[0.0, 0.0, 1.0], [0.0, 2.0, 1.0], [0.0, 4.0, 2.0]]))
    cases.append(case("wide_inconsistent", [[1.0, 2.0, 3.0, 4.0], [2.0, 4.0, 6.0, 9.0]]))
    cases.append(case("tall", [[1.0, 2.0, 0.5], [3.0, 1.0, 0.25], [2.0, 2.0, 1.5], [1.0, 1.0, 1.0]]))
    cases.append(case("inverse_shape_bar2", [[2.0, 1.0, 1.0, 0.0], [1.0, 3.0, 0.0, 1.0]], bar_col=2))
    cases.append(case("tiny_and_huge", [[1e-7, 3e15, 1.0], [2.5e-6, 1e16, 7.0]]))
    cases.append(case("single_column", [[2.0], [4.0]]))
    # the reference's rank-constrained builders under fixed seeds (random_matrix.py:109-130, 222-247)
    import linalg_solver.random_matrix as RM
    builder = []
    for seed in (1, 2, 3, 4, 5):
        random.seed(seed)
        reg = RM.gen_regular_matrix(6)
        rk = RM.gen_matrix_with_rank(5, 7, 3)
        builder.append({"seed": seed, "regular6": reg.items, "rank3_5x7": rk.items})
    out = {"generator": "tests/golden/gen_latex_golden.py", "reference": G.REF, "cases": cases, "builder": builder}
    with open(os.path.join(HERE, "latex_cases.json"), "w") as f:
        json.dump(out, f, indent=0, ensure_ascii=False)
    print(f"wrote {len(cases)} cases, {sum(len(c['matrices']) for c in cases)} LaTeX matrices")


if __name__ == "__main__":
    main()
