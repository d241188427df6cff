"""Three factorisations of one order (default 8192) and nothing else: the workload for a rocprofv3 kernel trace.
usage: lu_once.py [n]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from linalg_solver_amd import gen
from linalg_solver_amd.device import DeviceSolver
dev = DeviceSolver()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
A0 = torch.empty(n, n, dtype=torch.float64, device="cuda")
dev.fill_(A0, gen.U11, 1)
A = A0.clone()
for r in range(3):
    A.copy_(A0); dev.getrf_(A)
torch.cuda.synchronize()
