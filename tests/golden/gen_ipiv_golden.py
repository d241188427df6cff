#!/usr/bin/env python3
"""Pivot-sequence fixtures at the BASELINE sizes.

Runs the CPU twin of the device algorithm (oracle/lu_twin.c: scalar unblocked partial-pivot LU, first row of
maximal |a| wins) on the counter-based `u11` generator at n = 4096 / 8192 / 16384, seed 1 -- the matrices of
BASELINE configs 2, 3 and 4 -- and stores, per order, the interchange vector `ipiv`, diag(U), the solution of
A x = b and sign / log|det|.  `tests/test_gpu_parity.py` compares the device `ipiv` with these bit for bit and the
values within 1e-9.  Nothing of the reference is involved: inputs come from linalg_solver_amd/gen.py, outputs
from the oracle built from oracle/lu_twin.c.

Run in the build container (the 16384 case takes ~20 min of one core):
    python tests/golden/gen_ipiv_golden.py 4096 8192 16384
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from linalg_solver_amd import gen  # noqa: E402
from oracle import capi  # noqa: E402

SEED = 1


def make(n: int) -> str:
    A, b = gen.system(gen.U11, SEED, n)
    t0 = time.time()
    LU, ipiv, info = capi.getrf(A)
    t1 = time.time()
    del A
    x = capi.getrs(LU, ipiv, b)
    sign, logabs = capi.slogdet(LU, ipiv)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"ipiv_u11_s{SEED}_n{n}.npz")
    np.savez_compressed(out, n=np.int64(n), seed=np.int64(SEED), info=np.int64(info), ipiv=ipiv.astype(np.int32),
                        diag_u=np.ascontiguousarray(np.diag(LU)), x=x, sign=np.float64(sign), logabs=np.float64(logabs),
                        max_abs_l=np.float64(np.max(np.abs(np.tril(LU, -1)))) if n <= 8192 else np.float64(1.0))
    print(f"n={n}: twin LU {t1 - t0:.1f} s ({2 / 3 * n ** 3 / (t1 - t0) / 1e9:.2f} GFLOP/s), info={info}, "
          f"{int((ipiv != np.arange(n)).sum())} interchanges, sign={sign:+.0f} log|det|={logabs:.6f} -> {out}", flush=True)
    return out


if __name__ == "__main__":
    for arg in sys.argv[1:] or ["4096", "8192"]:
        make(int(arg))
