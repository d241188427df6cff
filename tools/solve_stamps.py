#!/usr/bin/env python3
"""Development: LSX_S2_DBG=1 python tools/solve_stamps.py [n] -- the owner chain of the few-RHS solve, stamp by stamp."""
import os, sys
os.environ["LSX_S2_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from linalg_solver_amd import gen
from linalg_solver_amd.device import DeviceSolver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = DeviceSolver()
A = torch.empty(n, n, dtype=torch.float64, device="cuda")
dev.fill_(A, gen.U11, 1)
ipiv, info = dev.getrf_(A)
b = torch.empty(n, 1, dtype=torch.float64, device="cuda")
for r in range(3):
    dev.fill_(b, gen.U11, 1, col_off=gen.RHS_COL)
    dev.getrs_(A, ipiv, b)
torch.cuda.synchronize()
NB = (n + 127) // 128
raw = np.frombuffer(dev.h.read_scratch(0, NB * 128), dtype=np.uint64).reshape(2, NB, 8).astype(np.int64)
for d, nm in ((0, "lower"), (1, "upper")):
    t = raw[d] - raw[d][0, 0]
    us = t / 100.0
    print(f"{nm}: per owner [start, early partials done, x valid, after A, after B, stored]; period = stored - previous stored")
    for o in list(range(0, min(NB, 6))) + list(range(max(6, NB - 6), NB)):
        per = us[o, 5] - us[o - 1, 5] if o else 0.0
        print(f"  ord {o:3d}: " + " ".join(f"{v:8.2f}" for v in us[o, :6]) + f"   period {per:5.2f}  x valid after prev stored {us[o, 2] - (us[o - 1, 5] if o else 0):5.2f}")
    print(f"  mean period {np.diff(us[:, 5]).mean():.3f} us; valid->A {np.mean(us[1:, 3] - us[1:, 2]):.3f}; A->B {np.mean(us[1:, 4] - us[1:, 3]):.3f}; B->stored {np.mean(us[1:, 5] - us[1:, 4]):.3f}; prev stored->valid {np.mean(us[1:, 2] - us[:-1, 5]):.3f}")
