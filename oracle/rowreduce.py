"""Generic-scalar restatement of the reference's row reduction path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Works for any entry type
that supports ``/ - * == !=`` (int, float, fractions.Fraction, ...), exactly
like the reference, because it performs the same scalar operations in the same
order.  Each function names the reference lines it follows
(paths relative to /root/reference).

Steps are reported as tuples ``(kind, number, a, b)``:
  ("S", step, r1, r2)   row swap of 1-based rows r1, r2      linalg.py:552-563
  ("N", step, r, 0)     normalisation of 1-based row r       linalg.py:576-583
  ("E", step, c, 0)     elimination below pivot, 1-based col linalg.py:597-606
  ("U", step, c, 0)     elimination above pivot, 1-based col linalg.py:622-629
``step_label`` / ``step_text`` turn them back into the reference's
``(label, description)`` pairs (the reference labels both E and U as ``E<k>``).
"""
from __future__ import annotations

from copy import deepcopy
from typing import Any, List, Optional, Sequence, Tuple

Step = Tuple[str, int, int, int]


def effective_bar_col(bar_col: Optional[int], ncols: int) -> int:
    """linalg.py:543 -- ``bar_col or n - 1``: both None and 0 mean n-1."""
    return bar_col if bar_col else ncols - 1


def row_reduce(items: Sequence[Sequence[Any]], bar_col: Optional[int] = None):
    """Gauss-Jordan to RREF over columns [0, bar_col); linalg.py:534-630.

    Returns ``(A, pivots, steps)``; the reference additionally returns LaTeX
    renderings of every intermediate matrix, which carry no numeric content.
    """
    A = deepcopy([list(r) for r in items])  # :539 -- input is never mutated
    m, n = len(A), len(A[0])
    bar = effective_bar_col(bar_col, n)
    pivots: List[Tuple[int, int]] = []
    steps: List[Step] = []
    counter = 0
    pi = pj = 0
    while pi < m and pj < bar:  # :547
        if A[pi][pj] == 0:  # :548 exact zero test
            hit = None
            for i in range(pi + 1, m):  # :550 first non-zero below, not the largest
                if A[i][pj] != 0:
                    hit = i
                    break
            if hit is None:  # :565-567 zero column -> next column, same row
                pj += 1
                continue
            A[pi], A[hit] = A[hit], A[pi]  # :552
            steps.append(("S", counter, pi + 1, hit + 1))
            counter += 1
        piv = A[pi][pj]
        changed = False
        if piv != 1:  # :571 skip when already 1
            row = A[pi]
            for j in range(pj, n):  # :572 only from the pivot column rightwards
                old = row[j]
                row[j] = row[j] / piv
                changed = changed or row[j] != old
        if changed:  # :576
            steps.append(("N", counter, pi + 1, 0))
            counter += 1
        touched = False
        changed = False
        prow = A[pi]
        for k in range(pi + 1, m):  # :587
            f = A[k][pj]
            if f == 0:
                continue
            touched = True
            rk = A[k]
            for j in range(pj, n):  # :593
                old = rk[j]
                rk[j] = rk[j] - f * prow[j]
                changed = changed or rk[j] != old
        if touched and changed:  # :597
            steps.append(("E", counter, pj + 1, 0))
            counter += 1
        pivots.append((pi, pj))  # :607
        pi += 1
        pj += 1
    for r, c in reversed(pivots):  # :611 back elimination, last pivot first
        changed = False
        prow = A[r]
        for k in range(r):
            f = A[k][c]
            if f == 0:
                continue
            rk = A[k]
            for j in range(c, n):  # :618
                old = rk[j]
                rk[j] = rk[j] - f * prow[j]
                changed = changed or rk[j] != old
        if changed:  # :622
            steps.append(("U", counter, c + 1, 0))
            counter += 1
    return A, pivots, steps


def step_label(s: Step) -> str:
    kind, num = s[0], s[1]
    return ("E" if kind == "U" else kind) + str(num)


def step_text(s: Step) -> str:
    """Czech descriptions, character-identical to linalg.py:557,581,602,627."""
    kind, _, a, b = s
    if kind == "S":
        return r"Výměna řádků $R_{%d}$ a $R_{%d}$" % (a, b)
    if kind == "N":
        return r"Normalizace pivotního řádku %s" % a
    if kind == "E":
        return r"Eliminace prvků pod pivotem ve sloupci %s" % a
    return r"Eliminace nad pivotem ve sloupci %s" % a


def is_inconsistent(reduced, nvars: int, bar_col: int) -> bool:
    """linalg.py:913-934: a row with all-zero coefficients and non-zero rhs."""
    for row in reduced:
        if all(row[j] == 0 for j in range(nvars)) and row[bar_col] != 0:
            return True
    return False


def affine_subspace(reduced, pivots, nvars: int, bar_col: int):
    """linalg.py:937-999 -> (particular, generators-as-columns or None)."""
    m = len(reduced)
    col_of_row = [-1] * m
    for r, c in pivots:
        col_of_row[r] = c
    pivot_cols = {c for _, c in pivots}
    free = [j for j in range(nvars) if j not in pivot_cols]
    particular: List[Any] = [0] * nvars  # :960 free variables are the int 0
    for i in range(m):
        if col_of_row[i] != -1:
            particular[col_of_row[i]] = reduced[i][bar_col]
    gens = []
    for fj in free:
        g: List[Any] = [0] * nvars
        g[fj] = 1
        for i in range(m):
            if col_of_row[i] != -1:
                g[col_of_row[i]] = -reduced[i][fj]  # :981, may be -0.0
        gens.append(g)
    if not gens:
        return particular, None  # :998
    return particular, [list(col) for col in zip(*gens)]  # :985 generators are columns


NO_SOLUTION = "NoSolution"


def find_preimage_of(items, vec):
    """Logging branch of Matrix.find_preimage_of, linalg.py:632-680.

    Returns NO_SOLUTION or ``(particular, generator_columns_or_None, pivots)``.
    """
    if len(items) != len(vec):
        raise ValueError("Matrix dimensions must match")  # :642-643
    aug = [list(r) + [vec[i]] for i, r in enumerate(items)]  # :649-651
    bar = len(aug[0]) - 1
    red, pivots, _ = row_reduce(aug, bar_col=bar)
    nvars = len(red[0]) - 1
    if is_inconsistent(red, nvars, bar):
        return NO_SOLUTION
    part, gens = affine_subspace(red, pivots, nvars, bar)
    return part, gens, pivots


def inverse(items):
    """Logging branch of Matrix.inverse, linalg.py:682-743."""
    n = len(items)
    if any(len(r) != n for r in items):
        raise ValueError("Matrix must be square to invert.")  # :692-693
    aug = [list(r) + [1 if i == j else 0 for j in range(n)] for i, r in enumerate(items)]  # :704-706
    red, _, _ = row_reduce(aug, bar_col=n)  # :707-711 (bar_col = cols-1+1)
    for i in range(n):  # :725-737 left block within 1e-12 of I
        for j in range(n):
            want = 1 if i == j else 0
            if abs(red[i][j] - want) > 1e-12:
                return NO_SOLUTION
    return [row[n:] for row in red]  # :739
