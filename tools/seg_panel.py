#!/usr/bin/env python3
"""Development: shader clocks per column of ONE segment of the panel's owner step (build: tools/seg_panel.sh;
run with LSX_LIB_OVERRIDE=.../liblsx_seg<k>.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from linalg_solver_amd import gen
from linalg_solver_amd.device import DeviceSolver

names = ["0 step start -> shot A landed", "1 absorb + header arg-max + readlanes", "2 near load issue, 1/pivot, multipliers, LDS",
         "3 wait: shot B + near granules", "4 update of the block", "5 choose + arg-max + announce (header, near)",
         "6 s_info, shot A", "7 barrier (+ shot B)"]
k = int(os.environ.get("LSX_LIB_OVERRIDE", "seg-1").split("seg")[-1].split(".")[0])
dev = DeviceSolver()
dev.h.set_option("panel", 4)
dev.h.set_option("panel_debug", 1)
for m in (8192, 256):
    P = torch.empty(m, 128, dtype=torch.float64, device="cuda")
    ipiv = torch.zeros(128, dtype=torch.int32, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    for rep in range(2):
        dev.fill_(P, gen.U11, 3)
        dev.panel_(P, 0, ipiv, info)
    torch.cuda.synchronize()
    G = (m + 255) // 256
    need = 256 + 2 * G * 128 + 4 * G * 128 * 16
    off = (need + 255) & ~255
    raw = np.frombuffer(dev.h.read_scratch(off, G * 128), dtype=np.uint64).reshape(G, 16)
    cyc = raw[:, 12].astype(np.float64) / 112.0   # 112 owner steps with a successor in the same block per panel (7 of 8 columns)
    print(f"seg {names[k]:52s} m={m:5d}: {cyc.mean():7.0f} cycles per column (max {cyc.max():7.0f})", flush=True)
