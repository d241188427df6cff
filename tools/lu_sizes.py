#!/usr/bin/env python3
"""Development: LU time at the BASELINE orders with the default driver, and bit identity against the sequential driver
with the per-column panel at the smaller ones (the slow, simple path)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from linalg_solver_amd import gen
from linalg_solver_amd.device import DeviceSolver
dev = DeviceSolver()
for n, dt in ((8192, torch.float64), (4096, torch.float64), (2048, torch.float64), (12288, torch.float64), (16384, torch.float64), (8192, torch.float32)):
    A0 = torch.empty(n, n, dtype=dt, device="cuda")
    dev.fill_(A0, gen.U11, 1)
    A = A0.clone()
    ts = []
    for r in range(6):
        A.copy_(A0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ipiv, info = dev.getrf_(A)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    same = ""
    if n <= 4096:
        dev.h.set_option("lookahead", 0); dev.h.set_option("panel", 3)
        B = A0.clone()
        ip2, inf2 = dev.getrf_(B)
        torch.cuda.synchronize()
        dev.h.set_option("lookahead", 1); dev.h.set_option("panel", 4)
        same = "  bits == sequential/device-scope panel" if torch.equal(A, B) and torch.equal(ipiv, ip2) else "  MISMATCH vs sequential"
    print(f"n={n} {str(dt).split('.')[-1]}: {min(ts[1:]) * 1e3:.3f} ms  {2 / 3 * n ** 3 / min(ts[1:]) / 1e12:.2f} TF  info={int(info.item())}{same}", flush=True)
    del A0, A
