"""Shared test helpers: golden-fixture decoding and reproducible inputs."""
from __future__ import annotations

import json
import os
import random
from fractions import Fraction

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dec(x):
    t = x[0]
    if t == "i":
        return int(x[1])
    if t == "f":
        return float.fromhex(x[1])
    if t == "q":
        return Fraction(x[1], x[2])
    raise ValueError(t)


def dec_mat(M):
    return [[dec(v) for v in row] for row in M]


def same_scalar(a, b) -> bool:
    """Bit-level identity including type, sign of zero and NaN payload class."""
    if type(a) is not type(b):
        return False
    if isinstance(a, float):
        return a.hex() == b.hex()
    return a == b


def same_matrix(A, B) -> bool:
    return (len(A) == len(B) and all(len(r) == len(s) for r, s in zip(A, B))
            and all(same_scalar(x, y) for r, s in zip(A, B) for x, y in zip(r, s)))


def load_small_cases():
    with open(os.path.join(GOLDEN, "small_cases.json")) as f:
        return json.load(f)["cases"]


def is_numeric_case(items) -> bool:
    return all(isinstance(v, (int, float)) for row in items for v in row)


def big_inputs(n, seed, stream):
    """Same stream as tests/golden/gen_golden.py:big_inputs (inputs are not stored)."""
    random.seed(seed)
    if stream == "int5":
        A = [[float(random.randint(-5, 5)) for _ in range(n)] for _ in range(n)]
        b = [float(random.randint(-5, 5)) for _ in range(n)]
    else:
        A = [[random.uniform(-1, 1) for _ in range(n)] for _ in range(n)]
        b = [random.uniform(-1, 1) for _ in range(n)]
    return np.array(A), np.array(b)


def load_big(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    n = int(name.split("_")[0][1:])
    stream = name.split("_")[1]
    A, b = big_inputs(n, n, stream)
    chk = z["a_checksum"]
    assert A.sum() == chk[0] and np.abs(A).sum() == chk[1] and b.sum() == chk[2], "input stream drifted"
    return A, b, z


def relerr(x, ref):
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(x - ref)) / max(np.max(np.abs(ref)), 1e-300))
