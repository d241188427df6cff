"""Random test-matrix builders: the rank-constrained part of the reference's
`linalg_solver/random_matrix.py` (`RandomMatrixBuilder` :7-136, helpers :222-247) on top of the GPU
`Matrix.rank()` (rank-revealing RREF) and `Matrix.__mul__` (MFMA tile).

The draws are made in the reference's order (row-major `dist()` calls, default `random.randint(-5, 5)`,
retry until the rank test passes), so the same `random.seed` gives the same matrices
(tests/golden/latex_cases.json, key "builder").  The eigenvalue / Jordan-form builders belong to the
reference's symbolic eigen stack and are out of scope: they raise.
"""
from __future__ import annotations

import random
from typing import Any, Callable, Optional

from .matrix import Matrix


def _default_dist() -> int:
    return random.randint(-5, 5)   # random_matrix.py:104


class RandomMatrixBuilder:
    rank: Optional[int] = None
    num_rows: Optional[int] = None
    num_cols: Optional[int] = None
    dist: Optional[Callable[[], Any]] = None
    eigenvalues = None
    jordan_blocks = None

    @classmethod
    def new(cls, **kwargs) -> "RandomMatrixBuilder":
        b = cls()
        for k, v in kwargs.items():
            setattr(b, k, v)
        return b

    def with_size(self, num_rows: int, num_cols: int) -> "RandomMatrixBuilder":
        self.num_rows, self.num_cols = num_rows, num_cols
        return self

    def with_rank(self, rank: int) -> "RandomMatrixBuilder":
        self.rank = rank
        return self

    def with_dist(self, dist: Optional[Callable[[], Any]]) -> "RandomMatrixBuilder":
        self.dist = dist
        return self

    def with_eigenvalues(self, eigenvalues) -> "RandomMatrixBuilder":
        raise NotImplementedError("eigenvalue-constrained builders are part of the reference's symbolic eigen stack: out of scope")

    def with_jordan_blocks(self, blocks) -> "RandomMatrixBuilder":
        raise NotImplementedError("Jordan-form builders are part of the reference's symbolic eigen stack: out of scope")

    def is_square(self) -> bool:
        return self.num_rows == self.num_cols

    def assert_requirements(self) -> None:
        if self.rank is not None:
            assert self.rank <= min(self.num_rows, self.num_cols), "Rank cannot exceed min(num_rows, num_cols)."

    def build_sized(self, num_rows: int, num_cols: Optional[int] = None) -> Matrix:
        self.num_rows = num_rows
        self.num_cols = num_cols if num_cols is not None else num_rows
        return self.build()

    def build(self) -> Matrix:
        self.assert_requirements()
        if self.rank is not None:
            if self.rank == min(self.num_rows, self.num_cols) and self.is_square():
                return self.build_full_rank()
            return self.build_rank()
        return self.build_random()

    def _draw(self, rows: int, cols: int) -> Matrix:
        dist = self.dist or _default_dist
        return Matrix([[dist() for _ in range(cols)] for _ in range(rows)])

    def build_random(self) -> Matrix:
        return self._draw(self.num_rows, self.num_cols)

    def build_full_rank(self) -> Matrix:
        """random_matrix.py:109-115 with the rank test on the GPU."""
        n = self.num_rows
        while True:
            val = self._draw(n, n)
            if val.rank() == n:
                return val

    def build_rank(self) -> Matrix:
        """random_matrix.py:117-130: (rows x rank) * (rank x cols), both factors of full rank."""
        rows, cols, rank = self.num_rows, self.num_cols, self.rank
        while True:
            A = self._draw(rows, rank)
            if A.rank() == rank:
                break
        while True:
            B = self._draw(rank, cols)
            if B.rank() == rank:
                break
        P = A * B
        if all(isinstance(v, int) for M in (A, B) for row in M.items for v in row):
            # the reference multiplies Python ints exactly; the MFMA product of small ints is exact in fp64
            P = Matrix([[int(round(v)) for v in row] for row in P.items])
        return P


def raw_gen_rand_matrix(rows: int, cols: int, dist: Optional[Callable[[], Any]] = None) -> Matrix:
    return RandomMatrixBuilder.new().with_size(rows, cols).with_dist(dist).build_random()


def gen_regular_matrix(N: int, dist: Optional[Callable[[], Any]] = None) -> Matrix:
    return RandomMatrixBuilder.new().with_size(N, N).with_dist(dist).build_full_rank()


def gen_matrix_with_rank(rows: int, cols: int, rank: Optional[int] = None,
                         dist: Optional[Callable[[], Any]] = None) -> Matrix:
    return (RandomMatrixBuilder.new().with_size(rows, cols).with_rank(rank or min(rows, cols))
            .with_dist(dist).build_rank())
