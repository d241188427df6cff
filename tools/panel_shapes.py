"""Per-column time of the one-launch panel kernels over thread shapes and panel heights (GPU)."""
import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
from kbench import timeit
from linalg_solver_amd import gen
from linalg_solver_amd.device import DeviceSolver
dev = DeviceSolver()
for mode, nt, rt in ((1,0,4),(3,512,4),(3,256,4),(3,256,8),(3,512,8)):
    dev.h.set_option("panel", mode); dev.h.set_option("panel_rt", rt); dev.h.set_option("panel_nt", nt)
    out=[]
    for m in (16384, 8192, 6144, 4096, 2048, 1024, 256):
        P0 = torch.empty(m, 128, dtype=torch.float64, device="cuda"); dev.fill_(P0, gen.U11, 3)
        ipiv = torch.zeros(128, dtype=torch.int32, device="cuda"); info = torch.zeros(1, dtype=torch.int32, device="cuda")
        P = P0.clone()
        def run():
            P.copy_(P0); dev.panel_(P, 0, ipiv, info)
        tmin,_ = timeit(run, reps=5, warm=1); tc,_ = timeit(lambda: P.copy_(P0), reps=5, warm=1)
        out.append(f"{m}:{(tmin-tc)*1e3/128:.2f}")
    print(f"mode={mode} nt={nt} rt={rt}  us/col  " + "  ".join(out), flush=True)
