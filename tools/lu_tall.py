#!/usr/bin/env python3
"""Development: orders above one XCD's panel height -- the phase in front of the XCD-scope driver with the chain behind
an event on the whole previous update (x_events=1) or behind the counted first tile column (0, default)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from linalg_solver_amd import gen
from linalg_solver_amd.device import DeviceSolver
dev = DeviceSolver()
for n in (16384, 12288, 10240):
    A0 = torch.empty(n, n, dtype=torch.float64, device="cuda")
    dev.fill_(A0, gen.U11, 1)
    ref = None
    for xe in (1, 0, 1, 0):
        dev.h.set_option("x_events", xe)
        A = A0.clone()
        ts = []
        for r in range(4):
            A.copy_(A0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ipiv, info = dev.getrf_(A)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        if ref is None:
            ref = (A.clone(), ipiv.clone()); same = "ref"
        else:
            same = "same bits" if torch.equal(A, ref[0]) and torch.equal(ipiv, ref[1]) else "MISMATCH"
        print(f"n={n} x_events={xe}: {min(ts[1:]) * 1e3:.2f} ms  {2 / 3 * n ** 3 / min(ts[1:]) / 1e12:.1f} TF  info={int(info.item())} {same}", flush=True)
    dev.h.set_option("x_events", 0)
    del A0, A, ref
