#!/usr/bin/env python3
"""Fused chain head against the separate block-inverse launch, fp32 and fp64 (development check)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from linalg_solver_amd import _native as N

h = N.default_handle()
lib = h.lib
for dt, fn in ((torch.float32, lib.lsx_diag_chain_head_f32), (torch.float64, lib.lsx_diag_chain_head_f64)):
    for jb, ld, ncols, with_moves in ((128, 300, 128, False), (128, 300, 44, True), (128, 256, 128, True), (64, 128, 64, False), (100, 304, 32, True)):
        torch.manual_seed(jb + ld)
        Tm = (torch.rand(jb + 200, ld, dtype=dt, device="cuda") - 0.5)
        A = torch.rand(jb + 200, ld, dtype=dt, device="cuda")
        A0 = A.clone()
        mv = torch.full((256, 2), -1, dtype=torch.int32, device="cuda")
        if with_moves:
            perm = torch.randperm(jb)[:40]
            for i in range(0, 40, 2):
                mv[i, 0], mv[i, 1] = int(perm[i]), int(perm[i + 1])
                mv[i + 1, 0], mv[i + 1, 1] = int(perm[i + 1]), int(perm[i])
        nblk = (jb + 63) // 64
        out = []
        for fused in (0, 1, 1, 1):
            Ti = torch.zeros(nblk * 4096, dtype=dt, device="cuda")
            A.copy_(A0)
            N.check(fn(h._h, fused, jb, Tm.data_ptr(), ld, Ti.data_ptr(), ncols, A.data_ptr(), ld, 0, mv.data_ptr()), "chain_head")
            h.synchronize()
            torch.cuda.synchronize()
            out.append(Ti.clone())
        # reference: inverse of the unit-lower 64-blocks in float64
        ok = [bool(torch.equal(out[0], o)) for o in out[1:]]
        worst = 0.0
        for b in range(nblk):
            w = min(64, jb - 64 * b)
            L = torch.tril(Tm[64 * b:64 * b + w, 64 * b:64 * b + w].double(), -1) + torch.eye(w, dtype=torch.float64, device="cuda")
            ref = torch.linalg.inv(L)
            got0 = out[0][4096 * b:4096 * (b + 1)].view(64, 64)[:w, :w].double()
            got1 = out[1][4096 * b:4096 * (b + 1)].view(64, 64)[:w, :w].double()
            worst = max(worst, float((got0 - ref).abs().max()), float((got1 - ref).abs().max()))
        print(f"{str(dt)[6:]} jb={jb} ld={ld} ncols={ncols} moves={with_moves}: fused == separate {ok}  max err vs fp64 inverse {worst:.2e}", flush=True)
