"""CPU stand-in for the local kernels of linalg_solver_amd.dist.ShardedLU (TESTS ONLY).

Same method names and argument meaning as linalg_solver_amd.device.DeviceSolver, on CPU
torch tensors, implemented with numpy/scipy.  It lets the gloo world_size-2 tests exercise
the distribution logic (ownership, offsets, the panel broadcast) without a GPU; the product
never imports this file."""
import numpy as np
import scipy.linalg as sl

from linalg_solver_amd import gen


class CpuOps:
    def fill_(self, A, kind, seed, row_off=0, col_off=0):
        A.numpy()[...] = gen.fill(kind, seed, A.shape[0], A.shape[1], row_off, col_off)
        return A

    def panel_(self, P, row0, ipiv, info):
        a = P.numpy()
        m, jb = a.shape
        piv = ipiv.numpy()
        for j in range(min(m, jb)):
            p = j + int(np.argmax(np.abs(a[j:, j])))  # first maximum: lowest row wins ties
            piv[j] = row0 + p
            if p != j:
                a[[j, p]] = a[[p, j]]
            if a[j, j] == 0.0:
                if info[0] == 0:
                    info[0] = row0 + j + 1
                continue
            a[j + 1:, j] *= 1.0 / a[j, j]
            a[j + 1:, j + 1:] -= np.outer(a[j + 1:, j], a[j, j + 1:])

    def laswp_(self, A, row0, jb, ipiv):
        a = A.numpy()
        piv = ipiv.numpy()
        for k in range(jb):
            p = int(piv[k])
            if p != row0 + k:
                a[[row0 + k, p]] = a[[p, row0 + k]]

    def trsm_lu_(self, L, B):
        b = B.numpy()
        b[...] = sl.solve_triangular(L.numpy(), b, lower=True, unit_diagonal=True, check_finite=False)

    def gemm_sub_(self, C, A, B):
        C.numpy()[...] -= A.numpy() @ B.numpy()
        return C
