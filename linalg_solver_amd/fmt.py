"""LaTeX formatting of numeric matrices: the output of the reference's `linalg_solver/fmt.py`
(`cformat` :16-29, `make_latex_matrix` :61-65, `make_latex_vector` :68-71,
`make_latex_augmented_matrix` :75-86) for int / float entries, without going through `sympy.latex`
for every entry -- that call is >99 % of the reference's row_reduce time.

`latex_float(x)` reproduces `sympy.latex(float)`: 15 significant digits, trailing zeros stripped (one
digit always kept after the point), positional notation for decimal exponents -4 .. 14 and
`m \\cdot 10^{e}` outside, `-0.0 -> 0.0`.  tests/test_fmt.py checks it against sympy on random and
edge-case doubles and against LaTeX strings captured from the reference.
"""
from __future__ import annotations

import math
from typing import Any, List, Sequence

_DPS = 15          # sympy Float(float) precision: 53 bits -> 15 decimal digits
_MIN_FIXED = -5    # mpmath.libmp.to_str defaults for dps = 15: positional iff min_fixed < exponent < max_fixed
_MAX_FIXED = 15


def latex_float(x: float) -> str:
    if x != x:
        return r"\text{NaN}"
    if x in (math.inf, -math.inf):
        return r"\infty" if x > 0 else r"-\infty"
    if x == 0.0:
        return "0.0"
    # 15 significant digits of the exact binary value, rounded to nearest with exact ties away from
    # zero (what mpmath's to_str gives sympy).  C's %e conversion is exact but breaks ties to even, so a
    # candidate tie (16th digit 5) is checked against a longer expansion and rounded up by hand.
    ax = abs(x)
    mant, exp = ("%.14e" % ax).split("e")
    keep = mant.replace(".", "")            # 15 digits
    exponent = int(exp)
    m16 = ("%.15e" % ax).split("e")[0]
    if m16[-1] == "5":
        long_digits = ("%.60e" % ax).split("e")[0].replace(".", "")
        if long_digits[15] == "5" and not long_digits[16:].strip("0"):   # exact tie
            up = str(int(long_digits[:15]) + 1)
            if len(up) > _DPS:
                up, exponent = up[:_DPS], int(("%.60e" % ax).split("e")[1]) + 1
            else:
                exponent = int(("%.60e" % ax).split("e")[1])
            keep = up
    sign = "-" if x < 0 else ""
    if _MIN_FIXED < exponent < _MAX_FIXED:
        if exponent >= 0:
            ip, fp = keep[: exponent + 1], keep[exponent + 1:]
        else:
            ip, fp = "0", "0" * (-exponent - 1) + keep
        fp = fp.rstrip("0") or "0"
        return f"{sign}{ip}.{fp}"
    m = keep[0] + "." + (keep[1:].rstrip("0") or "0")
    return f"{sign}{m} \\cdot 10^{{{exponent}}}"


def cformat(val: Any, arg_of: Any = None) -> str:
    """fmt.py:16-29 for the entry types the numeric path produces."""
    if hasattr(val, "cformat") and callable(val.cformat):
        return val.cformat(arg_of)
    if isinstance(val, str):
        return val
    if isinstance(val, bool):
        return r"\text{True}" if val else r"\text{False}"
    if isinstance(val, int):
        return str(val)
    if isinstance(val, float):
        return latex_float(val)
    try:                                    # numpy scalars
        import numpy as np
        if isinstance(val, np.integer):
            return str(int(val))
        if isinstance(val, np.floating):
            return latex_float(float(val))
    except ImportError:  # pragma: no cover
        pass
    return str(val)


def _rows(items: Sequence[Sequence[Any]]) -> List[str]:
    # floats repeat a lot in a matrix under reduction (0.0, 1.0, whole columns): format each value once.
    # Keyed by the float alone is safe: only floats enter, and -0.0 / 0.0 share the string "0.0".
    seen: dict = {}
    out = []
    for row in items:
        cells = []
        for item in row:
            if type(item) is float:
                s = seen.get(item)
                if s is None:
                    s = latex_float(item)
                    seen[item] = s
            else:
                s = cformat(item)
            cells.append(s)
        out.append(r" & ".join(cells))
    return out


def make_latex_matrix(items: Sequence[Sequence[Any]]) -> str:
    """fmt.py:61-65."""
    return r"\begin{pmatrix}" + (r"\\[0.1em]" + "\n").join(_rows(items)) + r"\end{pmatrix}"


def make_latex_vector(items: Sequence[Any]) -> str:
    """fmt.py:68-71."""
    return r"\begin{pmatrix}" + (r"\\[0.1em]" + "\n").join([cformat(item) for item in items]) + r"\end{pmatrix}"


def make_latex_augmented_matrix(items: Sequence[Sequence[Any]], bar_col: int = None) -> str:
    """fmt.py:75-86: one vertical bar before column bar_col (default: the last column)."""
    if len(items[0]) <= 1:
        return make_latex_matrix(items)
    if bar_col is None:
        bar_col = len(items[0]) - 1
    n_cols = len(items[0])
    col_format = "".join([("|c" if j == bar_col else "c") for j in range(n_cols)])
    start = r"\left(\begin{array}{" + col_format + "}\n"
    end = "\n" + r"\end{array}\right)"
    return start + (r" \\[0.1em]" + "\n").join(_rows(items)) + end
