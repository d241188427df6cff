"""Deterministic counter-based input generators.

Element (i, j) depends only on (seed, i, j), so the host (numpy, here) and the
device (lsx_fill_*_dev) produce bit-identical matrices without shipping files.
"""
from __future__ import annotations

import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
INT5, U11 = 0, 1


def splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def fill(kind: int, seed: int, m: int, n: int, row_off: int = 0, col_off: int = 0, dtype=np.float64) -> np.ndarray:
    i = (np.arange(m, dtype=np.uint64) + np.uint64(row_off))[:, None]
    j = (np.arange(n, dtype=np.uint64) + np.uint64(col_off))[None, :]
    with np.errstate(over="ignore"):
        h = splitmix64(np.uint64(seed) * GOLDEN + (i << np.uint64(32)) + j)
    if kind == INT5:
        v = (h % np.uint64(11)).astype(np.float64) - 5.0
    else:
        v = (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0) * 2.0 - 1.0
    return v.astype(dtype)


RHS_COL = 0x7FFFFFFF  # column index used for the right-hand side b_i


def system(kind: int, seed: int, n: int, dtype=np.float64):
    return fill(kind, seed, n, n, dtype=dtype), fill(kind, seed, n, 1, col_off=RHS_COL, dtype=dtype)[:, 0].copy()
