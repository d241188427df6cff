#!/usr/bin/env python3
"""Panel kernel time only (development): python tools/panel_time.py [mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from linalg_solver_amd import gen
from linalg_solver_amd.device import DeviceSolver

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = DeviceSolver()
dev.h.set_option("panel", mode)
for m in (8192, 4096, 1024, 256):
    P0 = torch.empty(m, 128, dtype=torch.float64, device="cuda")
    dev.fill_(P0, gen.U11, 3)
    ipiv = torch.zeros(128, dtype=torch.int32, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    P = P0.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts, tc = [], []
    for rep in range(9):
        P.copy_(P0)
        e0.record(); dev.panel_(P, 0, ipiv, info); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[1]
    print(f"{os.environ.get('LSX_LIB_OVERRIDE', 'product')}: panel m={m}: {t * 1e3:.1f} us ({t * 1e3 / 128:.2f} us/col) info={int(info.item())}", flush=True)
