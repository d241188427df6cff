#!/usr/bin/env python3
"""Development: one factorisation of 4096^2 and a few single-right-hand-side solves, for
`rocprofv3 --kernel-trace --stats -- python3 tools/solve_trace.py` (which kernels make up the solve latency)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from linalg_solver_amd import gen
from linalg_solver_amd.device import DeviceSolver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nrhs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = DeviceSolver()
A = torch.empty(n, n, dtype=torch.float64, device="cuda")
dev.fill_(A, gen.U11, 1)
ipiv, info = dev.getrf_(A)
b = torch.empty(n, nrhs, dtype=torch.float64, device="cuda")
ts = []
for r in range(8):
    dev.fill_(b, gen.U11, 1, col_off=gen.RHS_COL)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev.getrs_(A, ipiv, b)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print(f"n={n} nrhs={nrhs}: solve latency min {min(ts[1:]) * 1e3:.3f} ms  all {[round(t * 1e3, 3) for t in ts]}")
