#!/usr/bin/env python3
"""Development: orders above one XCD's panel height -- slice height of the device-scope panel in the first phase."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from linalg_solver_amd import gen
from linalg_solver_amd.device import DeviceSolver
dev = DeviceSolver()
for n in (16384, 12288):
    A0 = torch.empty(n, n, dtype=torch.float64, device="cuda")
    dev.fill_(A0, gen.U11, 1)
    ref = None
    for nt, rt in ((0, 4), (512, 8), (256, 8), (512, 4), (0, 4)):
        try:
            dev.h.set_option("panel_nt", nt); dev.h.set_option("panel_rt", rt)
        except Exception as e:
            print(f"n={n} nt={nt} rt={rt}: refused {e}"); continue
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
        print(f"n={n} panel_nt={nt} panel_rt={rt}: {min(ts[1:]) * 1e3:.2f} ms  {2 / 3 * n ** 3 / min(ts[1:]) / 1e12:.1f} TF  info={int(info.item())} {same}", flush=True)
    dev.h.set_option("panel_nt", 0); dev.h.set_option("panel_rt", 4)
    del A0, A, ref
