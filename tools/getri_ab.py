"""Inverse from the factors with one / two 128-row blocks per trailing update (option getri_pairs): same bits, times."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from linalg_solver_amd import gen
from linalg_solver_amd.device import DeviceSolver

dev = DeviceSolver()
h = dev.h
for n, dt in ((8192, torch.float64), (4096, torch.float64), (8000, torch.float64), (7168, torch.float64), (5000, torch.float64), (1000, torch.float64),
              (520, torch.float64), (8192, torch.float32), (3000, torch.float32)):
    A = torch.empty(n, n, dtype=dt, device="cuda")
    dev.fill_(A, gen.U11, 5)
    A0 = A.clone()
    ipiv, info = dev.getrf_(A)
    ref = None
    for pairs in (0, 1, 0, 1):
        h.set_option("getri_pairs", pairs)
        ts = []
        for r in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            inv = dev.getri(A, ipiv)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        if ref is None:
            ref = inv.clone()
            tag = "ref"
        else:
            tag = "same bits" if torch.equal(inv, ref) else f"DIFFERENT (max rel {float(((inv - ref).abs().max() / ref.abs().max())):.2e})"
        err = float((A0 @ inv - torch.eye(n, dtype=dt, device="cuda")).abs().max())
        print(f"n={n} {str(dt)[6:]} getri_pairs={pairs}: {min(ts[1:]) * 1e3:.3f} ms  {tag}  |A*inv - I| {err:.2e}", flush=True)
h.set_option("getri_pairs", 1)
