import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from linalg_solver_amd import gen
from linalg_solver_amd.device import DeviceSolver
dev = DeviceSolver()
OPT = sys.argv[1] if len(sys.argv) > 1 else "chain_fused"
VALS = [int(v) for v in sys.argv[2:4]] if len(sys.argv) > 3 else [0, 1]
for n in (16384, 12288, 10240, 8192):
    A0 = torch.empty(n, n, dtype=torch.float64, device="cuda")
    dev.fill_(A0, gen.U11, 1)
    ref = None
    for fused in (VALS[0], VALS[1], VALS[0], VALS[1]):
        dev.h.set_option(OPT, fused)
        A = A0.clone()
        ts = []
        for r in range(6):
            A.copy_(A0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ipiv, info = dev.getrf_(A)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        if ref is None:
            ref = (A.clone(), ipiv.clone())
            same = "ref"
        else:
            same = "same bits" if torch.equal(A, ref[0]) and torch.equal(ipiv, ref[1]) else "MISMATCH"
        print(f"n={n} {OPT}={fused}: {min(ts[1:]) * 1e3:.3f} ms  info={int(info.item())} {same}", flush=True)
