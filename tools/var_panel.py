#!/usr/bin/env python3
"""Development: time the XCD-scope panel of the library named by LSX_LIB_OVERRIDE (tools/var_panel.sh) and check it
against the first protocol bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from linalg_solver_amd import gen
from linalg_solver_amd.device import DeviceSolver

tag = os.path.basename(os.environ.get("LSX_LIB_OVERRIDE", "liblsx.so"))
dev = DeviceSolver()
dev.h.set_option("panel", 4)
f32 = "--f32" in sys.argv
dt = torch.float32 if f32 else torch.float64
line = []
for m in (8192, 6144, 4096, 2048, 1024, 256):
    P0 = torch.empty(m, 128, dtype=dt, device="cuda")
    dev.fill_(P0, gen.U11, 3)
    outs = []
    for proto in (0, 1):
        dev.h.set_option("panel_proto", proto)
        P = P0.clone()
        ipiv = torch.zeros(128, dtype=torch.int32, device="cuda")
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        dev.panel_(P, 0, ipiv, info)
        torch.cuda.synchronize()
        outs.append((P, ipiv.clone(), int(info.item())))
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]
    ts = {}
    for proto in (0, 1):
        dev.h.set_option("panel_proto", proto)
        P = P0.clone()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        best = 1e9
        for rep in range(12):
            P.copy_(P0)
            torch.cuda.synchronize()
            ev[0].record()
            dev.panel_(P, 0, ipiv, info)
            ev[1].record()
            torch.cuda.synchronize()
            if rep >= 2:
                best = min(best, ev[0].elapsed_time(ev[1]))
        ts[proto] = best * 1e3
    line.append(f"m={m}: {ts[0]:.0f}/{ts[1]:.0f} us ({ts[1] / 128:.2f} us/col){'' if same else ' MISMATCH'}")
print(f"{tag:22s} old/new  " + "  ".join(line), flush=True)
