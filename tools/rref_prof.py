import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import linalg_solver_amd as la
from linalg_solver_amd import dense
n, r = 8192, 4096
rng = np.random.default_rng(5)
L = rng.integers(-3, 4, size=(n, r)).astype(np.float64)
R = rng.integers(-3, 4, size=(r, n)).astype(np.float64)
A = L @ R
for rep in range(2):
    t = time.time()
    red, piv, rank = dense.rref(A, n, pivot_rule=la._native.PIVOT_MAX)[:3]
    print("rref", time.time() - t, "rank", rank, flush=True)
