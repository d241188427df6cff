import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from linalg_solver_amd import gen
from linalg_solver_amd.device import DeviceSolver
dev = DeviceSolver()
n = 8192
A0 = torch.empty(n, n, dtype=torch.float64, device="cuda"); dev.fill_(A0, gen.U11, 1)
A = A0.clone(); ipiv = torch.zeros(n, dtype=torch.int32, device="cuda"); info = torch.zeros(1, dtype=torch.int32, device="cuda")
dev.h.set_option("lookahead", 2)
for _ in range(2):
    A.copy_(A0); dev.getrf_(A, ipiv, info); torch.cuda.synchronize()
