"""Column-distributed XCD panel (panel_col = 1) against the row-distributed one (panel_col = 0): same factors, pivots and info,
bit for bit, on ragged shapes and tie-rich data; then per-column times of both.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from linalg_solver_amd import gen
from linalg_solver_amd.device import DeviceSolver

dev = DeviceSolver()
h = dev.h
f32 = "--f32" in sys.argv
dt = torch.float32 if f32 else torch.float64
h.set_option("panel", 4)
bad = 0
shapes = [(8192, 128), (8000, 128), (6000, 100), (4096, 128), (4097, 128), (3000, 17), (2500, 128), (2048, 128), (2049, 5), (1024, 128), (1025, 4),
          (1000, 3), (384, 128), (257, 100), (129, 128), (128, 128), (100, 100), (100, 17), (64, 1), (5, 5), (1, 1)]
if "--quick" in sys.argv:
    shapes = [(1024, 128), (2048, 128), (4096, 128), (8192, 128), (257, 100)]
for wt in (0, 1):
    h.set_option("panel_col_wt", wt)
    for m, jb in shapes:
        if jb > m:
            continue
        for kind in (gen.U11, gen.INT5):
            P0 = torch.empty(m, jb, dtype=dt, device="cuda")
            dev.fill_(P0, kind, 3)
            if kind == gen.INT5 and m >= 100:
                P0[:, min(3, jb - 1)] = 0          # a zero column: info, no pivot
            outs = []
            for pc in (0, 1):
                h.set_option("panel_col", pc)
                before = h.get_option("panel_col_launches")
                P = P0.clone()
                ipiv = torch.zeros(jb, dtype=torch.int32, device="cuda")
                info = torch.zeros(1, dtype=torch.int32, device="cuda")
                dev.panel_(P, 0, ipiv, info)
                torch.cuda.synchronize()
                mv = torch.zeros(512, dtype=torch.int32, device="cuda")
                okm = dev.panel_moves_(mv)
                torch.cuda.synchronize()
                outs.append((P, ipiv.clone(), int(info.item()), mv.clone() if okm else None, h.get_option("panel_col_launches") - before))
            same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]
            if outs[0][3] is not None and outs[1][3] is not None:
                same = same and torch.equal(outs[0][3], outs[1][3])
            ran = outs[1][4] == (1 if m <= 4096 else 0) and outs[0][4] == 0
            if not (same and ran):
                bad += 1
                d = (outs[0][0] != outs[1][0]).nonzero()
                print(f"   first differing entries: {d[:4].tolist()}  ipiv diff at {(outs[0][1] != outs[1][1]).nonzero()[:4].flatten().tolist()}")
            print(f"panelc check wt={wt} m={m} jb={jb} kind={kind}: {'same' if same else 'MISMATCH'} info {outs[0][2]}/{outs[1][2]} ran={ran}", flush=True)
h.set_option("panel_col_wt", 0)
print("MISMATCHES:", bad, flush=True)
if bad == 0 or "--time" in sys.argv:
    for pc in (0, 1):
        h.set_option("panel_col", pc)
        for m in (8192, 6144, 4096, 2048, 1024, 256):
            P0 = torch.empty(m, 128, dtype=dt, device="cuda")
            dev.fill_(P0, gen.U11, 3)
            ipiv = torch.zeros(128, dtype=torch.int32, device="cuda")
            info = torch.zeros(1, dtype=torch.int32, device="cuda")
            P = P0.clone()
            ts = []
            for r in range(9):
                P.copy_(P0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                dev.panel_(P, 0, ipiv, info)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            t = min(ts[2:])
            print(f"panel_col={pc} m={m}: {t * 1e6:.1f} us launch-to-sync ({t * 1e6 / 128:.2f} us/col incl. ~launch)", flush=True)
if os.environ.get("LSX_PC_DBG"):
    import numpy as np
    h.set_option("panel_col", 1)
    for m in (4096, 2048, 1024, 256):
        P = torch.empty(m, 128, dtype=dt, device="cuda")
        ipiv = torch.zeros(128, dtype=torch.int32, device="cuda")
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        for rep in range(3):
            dev.fill_(P, gen.U11, 3)
            dev.panel_(P, 0, ipiv, info)
        torch.cuda.synchronize()
        mpad = (m + 255) & ~255
        need = ((256 + 32 * (2 * 128 + 4 * 128 * 16) + 255) & ~255) + 128 * 8 + 128 * mpad * (4 if f32 else 8)
        off = (need + 255) & ~255
        raw = np.frombuffer(h.read_scratch(off, 32 * 128), dtype=np.uint64).reshape(32, 16).astype(np.int64)
        t0 = raw[:, 0].min()
        us = (raw[:, :6] - t0) / 100.0
        print(f"stamps m={m}: per workgroup [start, loaded, left done, own done, right done, exit] us; waits in the left loop")
        for g in range(0, 32, 5):
            hand = us[g, 2] - us[g - 1, 3] if g else 0.0
            print(f"  g={g:2d} " + " ".join(f"{v:8.2f}" for v in us[g]) + f"   own {us[g, 3] - us[g, 2]:6.2f}  hand-over {hand:6.2f}  waits {raw[g, 6]}  own segs (clk/col): " + " ".join(f"{raw[g, 8 + k] / 4:6.0f}" for k in range(4)))
