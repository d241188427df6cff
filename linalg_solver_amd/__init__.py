"""MI355X-native dense LU / row reduction behind the linalg-solver `Matrix` surface.

    from linalg_solver_amd import Matrix
    Matrix([[2.0, 1.0], [1.0, 3.0]]).find_preimage_of([3.0, 5.0], log_steps=True)

Layers: matrix.py (reference-compatible surface) -> dense.py (numpy buffers)
-> _native.py (ctypes) -> liblsx.so (hand-written HIP for gfx950, csrc/).
fmt.py renders the reference's LaTeX; random_matrix.py mirrors its rank-constrained builders.
"""
from . import dense, fmt, gen
from ._native import Handle, LsxError, default_handle
from .matrix import Matrix
from .random_matrix import RandomMatrixBuilder, gen_matrix_with_rank, gen_regular_matrix, raw_gen_rand_matrix

__all__ = ["Matrix", "Handle", "LsxError", "default_handle", "dense", "gen", "fmt", "RandomMatrixBuilder",
           "gen_matrix_with_rank", "gen_regular_matrix", "raw_gen_rand_matrix"]
