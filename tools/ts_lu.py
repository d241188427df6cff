"""Development: timeline of the look-ahead LU from device-clock stamps (liblsx_ts.so, tools/ts_lu.sh).
    LSX_LIB_OVERRIDE=linalg_solver_amd/liblsx_ts.so python tools/ts_lu.py [n] [first_step] [nsteps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from linalg_solver_amd import gen
from linalg_solver_amd.device import DeviceSolver

NAMES = {1: "panel_x", 2: "chain_head", 3: "trsm_block2", 4: "gemm_sub", 6: "gemm_queue", 7: "gate", 8: "laswp_moves", 9: "laswp_left_all"}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 2
xev = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = DeviceSolver()
dev.h.set_option("x_events", xev)
buf = torch.zeros(1 + 2 * (1 << 20), dtype=torch.int64, device="cuda")
for name in ("panelx", "misc", "gemm"):
    fn = getattr(dev.lib, "lsx_ts_set_" + name)
    fn.argtypes = [C.c_void_p]
    fn.restype = None
    fn(C.c_void_p(0))
A0 = torch.empty(n, n, dtype=torch.float64, device="cuda")
dev.fill_(A0, gen.U11, 1)
A = A0.clone()
for _ in range(3):
    A.copy_(A0)
    dev.getrf_(A)
torch.cuda.synchronize()
for name in ("panelx", "misc", "gemm"):
    getattr(dev.lib, "lsx_ts_set_" + name)(C.c_void_p(buf.data_ptr()))
A.copy_(A0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ipiv, info = dev.getrf_(A)
e1.record()
torch.cuda.synchronize()
print(f"n={n}: LU {e0.elapsed_time(e1):.2f} ms with stamps, info={int(info.item())}")
b = buf.cpu().numpy()
cnt = int(b[0] & 0xffffffff)
tags, ts = b[1:1 + 2 * cnt:2], b[2:2 + 2 * cnt:2]
order = np.argsort(ts, kind="stable")
tags, ts = tags[order], ts[order]
# kernel instances: a start stamp opens one; its end = the last end stamp of that id before the id's next start
inst = []
open_ = {}
for tag, t in zip(tags, ts):
    kid = int(tag) & 0xff
    if int(tag) & 0x100:
        if kid in open_:
            open_[kid][2] = t
    else:
        if kid in open_:
            inst.append(open_[kid])
        open_[kid] = [kid, t, t]
inst += list(open_.values())
inst.sort(key=lambda x: x[1])
pan = [i for i, x in enumerate(inst) if x[0] == 1]
print(f"{cnt} stamps, {len(inst)} kernel instances, {len(pan)} panels")
if len(pan) > s0 + ns:
    i0, i1 = pan[s0], pan[s0 + ns]
    t0 = inst[i0][1]
    for kid, a, z in inst[i0:i1 + 1]:
        print(f"  {(a - t0) / 100.0:9.1f} {(z - t0) / 100.0:9.1f} {(z - a) / 100.0:8.1f}  {NAMES.get(kid, kid)}")
# averages over all steps
gaps = [(inst[pan[i + 1]][1] - inst[pan[i]][2]) / 100.0 for i in range(len(pan) - 1)]
durs = [(inst[p][2] - inst[p][1]) / 100.0 for p in pan]
print(f"panel duration: mean {np.mean(durs):.1f} us (first {durs[0]:.1f}, last {durs[-1]:.1f}); panel-to-panel gap: mean {np.mean(gaps):.1f} us, median {np.median(gaps):.1f}, first 15 steps mean {np.mean(gaps[:15]):.1f}, rest {np.mean(gaps[15:]):.1f}")
print("period per step (us): " + " ".join(f"{(inst[pan[i + 1]][1] - inst[pan[i]][1]) / 100.0:.0f}" for i in range(len(pan) - 1)))
print(f"sum of panels {np.sum(durs) / 1e3:.2f} ms + sum of gaps {np.sum(gaps) / 1e3:.2f} ms; first panel start to last panel end {(inst[pan[-1]][2] - inst[pan[0]][1]) / 1e5:.2f} ms")
