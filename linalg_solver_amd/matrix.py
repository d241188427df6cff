"""`Matrix`: host-side mirror of the reference's row-reduction surface.

Same names, argument order, defaults, return types and error behaviour as
koskja/linalg-solver's ``linalg_solver.linalg.Matrix`` for the path this
package replaces (citations are to /root/reference/linalg_solver/linalg.py):

    Matrix(items)                      :14-32   validation, public ``items``
    row_reduce(bar_col=None)           :534-630 -> (items, pivots, mats, steps)
    find_preimage_of(vec, log_*...)    :632-680 -> AffineSubspace | NoSolution
    inverse(log_*...)                  :682-743 -> Matrix | NoSolution
    determinant(...)                   :183-207 -> scalar
    rank()                             :745-747 -> int
    kernel()                           :749-756
    AffineSubspace / NoSolution        :491-532
    zero / identity / new_vector / transpose   :410-415, 483-489
    __mul__ / scalar_mul / __neg__             :91-158 (matrix products on the MFMA tile)

All arithmetic runs on the GPU through liblsx.so (dense.py -> _native.py).
There is no CPU fallback: entries must be Python/numpy ints or floats, anything
else (Fraction, sympy objects, Polynomial) raises TypeError -- exact and
symbolic entries remain the reference's job.

Deliberate differences (documented in DESIGN.md):
  * no per-step LaTeX: ``row_reduce`` returns empty ``intermediate_matrices`` /
    ``intermediate_steps`` (these cost the reference >99 % of its run time);
    the ``log_*`` flags are accepted and ignored.
  * pivoting is by largest magnitude with a tolerance (32*eps*max(m,n)*max|working matrix|), so
    ``rank`` / pivot positions are those of exact arithmetic rather than the
    rounding artefacts an exact ``== 0`` test produces on floats
    (SURVEY.md appendix A.9).
  * reduced entries are always ``float`` (the reference leaves untouched entries
    as ``int``); ``determinant`` of an integer matrix is a ``float``.
"""
from __future__ import annotations

import math
from typing import Any, List, Optional, Tuple

import numpy as np

from . import dense

_NUMERIC = (int, float, np.integer, np.floating)


def _as_array(items: List[List[Any]]) -> np.ndarray:
    for row in items:
        for v in row:
            if not isinstance(v, _NUMERIC):
                raise TypeError(
                    f"linalg_solver_amd.Matrix computes on the GPU in fp64 and accepts only int/float "
                    f"entries; got {type(v).__name__}. Exact or symbolic entries are not supported.")
    return np.array(items, dtype=np.float64, order="C").reshape(len(items), len(items[0]) if items else 0)


class Matrix:
    """Mirror of the reference's `Matrix` carrier (linalg.py:11-58).  `items` is the public, mutable
    list-of-lists of the reference.  A matrix built with `from_numpy` / `from_dlpack` keeps the array
    and materialises `items` only when somebody asks for it: at N >= 4096 the O(N^2) Python objects cost
    seconds, the factorisation milliseconds."""

    def __init__(self, items: List[List[Any]], *, backend: str = "auto"):
        # SURVEY 8b's opt-in selector.  "auto" and "hip" are the same thing here -- this package IS the HIP path and
        # ships no other; "cpu" is refused rather than served by a Python fall-back (the reference is the CPU path).
        if backend not in ("auto", "hip"):
            if backend == "cpu":
                raise NotImplementedError("linalg_solver_amd has no CPU backend by design: use the reference "
                                          "linalg_solver.Matrix for exact/symbolic or CPU-only work")
            raise ValueError(f"unknown backend {backend!r}: expected 'auto' or 'hip'")
        self.backend = "hip"
        # linalg.py:14-32 -- same checks, same messages
        if not items:
            raise ValueError("Matrix cannot be empty")
        if not all(isinstance(row, list) for row in items):
            raise ValueError("Matrix items must be a list of lists")
        if not items[0]:
            if any(row for row in items):
                raise ValueError("Matrix rows cannot be empty if columns exist")
            row_len = 0
        else:
            row_len = len(items[0])
            if not all(len(row) == row_len for row in items):
                raise ValueError("All matrix rows must have the same length")
        self._cols = row_len
        self._items = items
        self._src = None   # array backing (from_numpy); dropped as soon as `items` is handed out
        self._dev = None   # device backing (from_dlpack of a tensor in HBM): a view, never copied to the host unasked

    def _host(self):
        """The array backing on the host, fetched from the device view on first need."""
        if self._src is None and getattr(self, "_dev", None) is not None:
            self._src = self._dev.detach().cpu().numpy()
        return self._src

    @property
    def items(self) -> List[List[Any]]:
        if self._items is None:
            self._items = self._host().tolist()
            self._src = None   # the lists are mutable: from here on they are the only truth
            self._dev = None
        return self._items

    @items.setter
    def items(self, value: List[List[Any]]):
        self._items = value
        self._src = None
        self._dev = None

    def _array(self) -> np.ndarray:
        """fp64 image of the entries for the device (validated)."""
        if self._items is None:
            return np.ascontiguousarray(self._host(), dtype=np.float64)
        return _as_array(self._items)

    # ---- container surface ------------------------------------------------
    def __str__(self) -> str:
        return "\n".join(" ".join(str(v) for v in row) for row in self.items)

    def __repr__(self) -> str:
        return f"Matrix({self.items!r})"

    @property
    def rows(self) -> int:
        if self._items is None:
            return (self._dev if getattr(self, "_dev", None) is not None else self._src).shape[0]
        return len(self._items)

    @property
    def cols(self) -> int:
        if self._items is None:
            return (self._dev if getattr(self, "_dev", None) is not None else self._src).shape[1]
        return len(self._items[0]) if self._items else self._cols

    def get_row(self, i: int) -> List[Any]:
        return self.items[i]

    def get_col(self, j: int) -> List[Any]:
        return [row[j] for row in self.items]

    def set_item(self, i: int, j: int, value: Any):
        self.items[i][j] = value
        return self

    @classmethod
    def zero(cls, rows: int, cols: int) -> "Matrix":
        return cls([[0] * cols for _ in range(rows)])

    @classmethod
    def identity(cls, size: int) -> "Matrix":
        return cls([[1 if i == j else 0 for j in range(size)] for i in range(size)])

    @classmethod
    def new_vector(cls, items: List[Any]) -> "Matrix":
        return cls([[v] for v in items])

    @classmethod
    def from_numpy(cls, a) -> "Matrix":
        """2-D integer or floating array -> Matrix without creating the list-of-lists (made on first use
        of `.items`; integer dtypes give int entries, as the reference's examples use)."""
        a = np.array(a, copy=True, order="C")
        if a.ndim != 2 or a.shape[0] == 0:
            raise ValueError("from_numpy needs a non-empty 2-D array")
        if a.dtype.kind not in "iuf":
            raise TypeError(f"from_numpy needs an integer or floating array, got dtype {a.dtype}")
        m = cls.__new__(cls)
        m.backend = "hip"
        m._cols = a.shape[1]
        m._items = None
        m._src = a
        m._dev = None
        return m

    @classmethod
    def from_dlpack(cls, x) -> "Matrix":
        """Any DLPack producer -> Matrix.  A 2-D floating tensor that lives in HBM (torch on the ROCm device) is kept
        as a VIEW: solve_array / inverse_array / lu_device then run on it through the device-pointer entry points
        (`*_dev`) and hand back device tensors, nothing crosses PCIe; the host copy and the list-of-lists are made
        only if `.items` / `to_numpy()` are asked for.  Host producers (numpy, CPU tensors) are copied as before."""
        if hasattr(x, "detach") and hasattr(x, "is_cuda"):
            if x.is_cuda:
                if x.dim() != 2 or x.shape[0] == 0:
                    raise ValueError("from_dlpack needs a non-empty 2-D tensor")
                if not x.dtype.is_floating_point:
                    raise TypeError(f"from_dlpack of a device tensor needs a floating dtype, got {x.dtype}")
                m = cls.__new__(cls)
                m.backend = "hip"
                m._cols = x.shape[1]
                m._items = None
                m._src = None
                m._dev = x.detach()
                return m
            return cls.from_numpy(x.detach().numpy())
        return cls.from_numpy(np.from_dlpack(x))

    def to_numpy(self) -> np.ndarray:
        """fp64 copy of the entries."""
        return np.array(self._array(), copy=True)

    def __dlpack__(self, stream=None):
        if getattr(self, "_dev", None) is not None:
            return self._dev.__dlpack__(stream=stream) if stream is not None else self._dev.__dlpack__()
        return self.to_numpy().__dlpack__()

    def __dlpack_device__(self):
        if getattr(self, "_dev", None) is not None:
            return self._dev.__dlpack_device__()
        return self.to_numpy().__dlpack_device__()

    # ---- device-resident twins: operands and results stay in HBM ------------------------------------
    def _dev64(self):
        import torch

        t = self._dev
        return t if (t.dtype == torch.float64 and t.is_contiguous()) else t.to(torch.float64).contiguous()

    def lu_device(self):
        """(LU, ipiv, info) as device tensors: P A = L U of a matrix built from a device tensor (the view is not
        modified).  lsx_getrf_f64_dev."""
        from .device import DeviceSolver

        if getattr(self, "_dev", None) is None:
            raise ValueError("lu_device needs a Matrix built by from_dlpack from a tensor in HBM")
        dev = DeviceSolver(self._dev.device.index)
        LU = self._dev64().clone()
        ipiv, info = dev.getrf_(LU)
        return LU, ipiv, info

    def _solve_device(self, rhs):
        import torch

        from .device import DeviceSolver

        A = self._dev64()
        n = A.shape[0]
        if A.shape[1] != n:
            raise ValueError("solve_array needs a square matrix")
        dev = DeviceSolver(self._dev.device.index)
        LU = A.clone()
        ipiv, info = dev.getrf_(LU)
        if rhs is None:
            X = dev.getri(LU, ipiv)
        else:
            B = rhs if hasattr(rhs, "is_cuda") else torch.as_tensor(np.asarray(rhs, dtype=np.float64))
            B = B.to(device=A.device, dtype=torch.float64)
            vec = B.dim() == 1
            if B.shape[0] != n:
                raise ValueError("Matrix dimensions must match")
            X = B.reshape(n, -1).contiguous().clone()
            dev.getrs_(LU, ipiv, X)
            if vec:
                X = X[:, 0]
        # singular to working precision: the same test as the host path (smallest |pivot| against the largest entry)
        ratio = float(LU.diagonal().abs().min() / A.abs().max().clamp_min(1e-300))
        code = int(info.item())
        if code < 0:
            # a cooperative kernel timed out on the device (lsx.h: lsx_check_status): that is a failure of the run,
            # not a property of the matrix -- never report it as "singular"
            dev.h.check_status()
            raise RuntimeError(f"device factorisation failed (info = {code})")
        if code != 0 or not (ratio > dense.EPS64 * n):
            return Matrix.NoSolution()
        return X

    # ---- array-valued twins of the list-valued API (no O(N^2) Python objects) ----------------------
    def row_reduce_array(self, bar_col: int = None):
        """row_reduce on arrays: (reduced ndarray, pivots)."""
        A = self._array()
        n = A.shape[1]
        if n == 0:
            raise IndexError("list index out of range")
        bar = bar_col or n - 1
        R, pivots = self._reduce(A, min(bar, n))
        if bar > n and len(pivots) < A.shape[0]:
            raise IndexError("list index out of range")
        return R, pivots

    def solve_array(self, rhs) -> "np.ndarray | Matrix.NoSolution":
        """Unique solution(s) of self * X = rhs for a square matrix (rhs: vector or matrix) as an ndarray;
        NoSolution() when the matrix is singular to working precision (use find_preimage_of for the
        general affine answer)."""
        if getattr(self, "_dev", None) is not None and self._items is None:
            return self._solve_device(rhs)   # operands in HBM: the result is a device tensor too
        A = self._array()
        if A.shape[0] != A.shape[1]:
            raise ValueError("solve_array needs a square matrix")
        B = np.asarray(rhs, dtype=np.float64)
        vec = B.ndim == 1
        if B.shape[0] != A.shape[0]:
            raise ValueError("Matrix dimensions must match")
        X, info, ratio = dense.solve(A, B.reshape(A.shape[0], -1))
        if info != 0 or not (ratio > dense.EPS64 * A.shape[0]):
            return Matrix.NoSolution()
        return X[:, 0] if vec else X

    def inverse_array(self) -> "np.ndarray | Matrix.NoSolution":
        if getattr(self, "_dev", None) is not None and self._items is None:
            if self._dev.shape[0] != self._dev.shape[1]:
                raise ValueError("Matrix must be square to invert.")
            return self._solve_device(None)
        A = self._array()
        if A.shape[0] != A.shape[1]:
            raise ValueError("Matrix must be square to invert.")
        X, info, ratio = dense.inv(A)
        if info != 0 or not (ratio > dense.EPS64 * A.shape[0]):
            return Matrix.NoSolution()
        return X

    def transpose(self) -> "Matrix":
        return Matrix([[self.items[j][i] for j in range(self.rows)] for i in range(self.cols)])

    # ---- products (linalg.py:91-158): matrix x matrix on the GPU, scalars elementwise
    def scalar_mul(self, scalar: Any) -> "Matrix":
        return Matrix([[item * scalar for item in row] for row in self.items])  # linalg.py:91-92

    def __neg__(self) -> "Matrix":
        return self.scalar_mul(-1)

    def __mul__(self, other) -> "Matrix":
        if not isinstance(other, Matrix):
            return self.scalar_mul(other)  # linalg.py:101-103
        if self.cols != other.rows:
            raise ValueError("Matrix dimensions must match")  # linalg.py:104-105
        return Matrix(dense.matmul(self._array(), other._array()).tolist())

    def cformat(self, _arg_of: str = "") -> str:
        body = r"\\".join(" & ".join(_fmt(v) for v in row) for row in self.items)
        return r"\begin{pmatrix}" + body + r"\end{pmatrix}"

    # ---- result carriers (linalg.py:491-532) --------------------------------
    class AffineSubspace:
        def __init__(self, vec: List[Any], mat: Optional["Matrix"]):
            self.vec = vec
            self.generators = mat

        def get_one(self) -> List[Any]:
            return self.vec

        def dim(self) -> int:
            return self.generators.cols  # raises on None exactly like the reference (:500)

        def basis(self) -> List[List[Any]]:
            return self.generators.transpose().items

        def cformat(self, arg_of: str = "") -> str:
            g = self.generators
            point = Matrix.new_vector(self.vec).cformat()
            if g is None or g.rows == 0 or g.cols == 0:
                return " %s " % point
            gens = ", ".join(Matrix.new_vector(g.get_col(i)).cformat() for i in range(g.cols))
            span = r" \LO \left\{ %s \right\} " % gens
            return " %s %s  " % ("" if all(v == 0 for v in self.vec) else point + " + ", span)

        def __repr__(self) -> str:
            return f"AffineSubspace(vec={self.vec!r}, generators={self.generators!r})"

    class NoSolution:
        def __repr__(self) -> str:
            return "NoSolution()"

        def cformat(self, arg_of: str = "") -> str:
            return r"\text{Žádné řešení}"

    # ---- the replaced path --------------------------------------------------
    def _reduce(self, A: np.ndarray, bar: int) -> Tuple[np.ndarray, List[Tuple[int, int]]]:
        """RREF of A over columns [0, bar) on the GPU -> (reduced, pivots)."""
        m, n = A.shape
        if bar <= 0:
            return A.copy(), []
        if m == bar:
            # [A | B] with square A: blocked LU + two block solves instead of a 2n-wide
            # Gauss-Jordan (what linalg.py:649-656 / 704-711 spell as row_reduce(bar_col=n))
            X, info, ratio = dense.solve(A[:, :m], A[:, m:])
            if info == 0 and ratio > dense.EPS64 * m:
                R = np.zeros_like(A)
                R[:, :m] = np.eye(m)
                R[:, m:] = X
                return R, [(k, k) for k in range(m)]
        # Large inputs: ONE call.  The blocked first-rule reduction (csrc/api.hip: rref_first_fast) takes rank and pivot
        # columns from the rank-revealing pass and the reference's row choice (first non-zero row, linalg.py:548-552)
        # from a blocked LU of those columns under that rule, so its result IS the reference's reduction.
        h = dense._h(None)
        R1, pivots1, rank1 = dense.rref(A, bar_col=bar, pivot_rule=dense.N.PIVOT_FIRST)
        if h.get_option("rref_first_used"):
            return R1, pivots1
        # Small inputs (per-column kernels): rank and pivot positions from the well-conditioned rule (largest |a|) ...
        R, pivots, rank = dense.rref(A, bar_col=bar, pivot_rule=dense.N.PIVOT_MAX)
        if rank < m:
            # ... but with rank < m the carried-along columns depend on WHICH rows became pivot rows, so the values
            # come from the pass under the reference's rule when the two agree on the pivots
            if pivots1 == pivots:
                R = R1
            else:
                # the first-non-zero rule, run without magnitude pivoting, settled on other pivot columns than the
                # rank-revealing pass (noise in a dependent column passed its tolerance): the rank-revealing pivots and
                # values are returned, and the caller is TOLD that the carried-along columns are then not the
                # reference's (this used to pass silently)
                import warnings

                warnings.warn("row_reduce: the reference's first-non-zero pivot rule and the rank-revealing pass disagree on "
                              f"the pivot columns ({len(pivots1)} vs {len(pivots)} pivots); returning the rank-revealing "
                              "reduction -- columns right of bar_col and rows below the rank follow its row choice, not the "
                              "reference's", RuntimeWarning, stacklevel=3)
        return R, pivots

    # step descriptions of the reference's log (linalg.py:557-560, 580, 602-604, 626), verbatim: they are
    # part of what row_reduce returns
    _STEP_TEXT = (r"Výměna řádků $R_{%d}$ a $R_{%d}$", r"Normalizace pivotního řádku %s",
                  r"Eliminace prvků pod pivotem ve sloupci %s", r"Eliminace nad pivotem ve sloupci %s")
    TRACE_SNAPSHOT_BYTES = 256 << 20   # cap on the device memory spent on intermediate matrices

    def row_reduce(self, bar_col: int = None, trace=False):
        """linalg.py:534-630.  Returns (reduced_items, pivots, intermediate_matrices, intermediate_steps).

        trace=False (default): the fast paths (blocked LU / rank-revealing RREF); the two logs are empty.
        trace="steps": the reduction runs in the reference's own operation order on the device
        (lsx_rref_trace_f64): reduced_items, pivots and the (label, description) list are identical to
        the reference's, bit for bit; intermediate_matrices stays empty.
        trace=True / "full": additionally the LaTeX of the matrix after every step (first = the initial
        matrix, linalg.py:544), formatted like the reference's fmt.make_latex_augmented_matrix.  Meant
        for small inputs, like the reference's own log; beyond TRACE_SNAPSHOT_BYTES of snapshots the
        list is truncated to the first steps.
        Entry types follow the reference's object arithmetic: ints stay ints until a division or a float
        operand touches them (exact up to 2^53)."""
        A = self._array()
        n = A.shape[1]
        if n == 0:
            raise IndexError("list index out of range")  # reference fails at len(A[0]) / A[0][0]
        bar = bar_col or n - 1  # :543 -- 0 and None both mean n-1
        if trace:
            return self._row_reduce_traced(A, bar, full=(trace != "steps"))
        R, pivots = self._reduce(A, min(bar, n))
        if bar > n and len(pivots) < A.shape[0]:
            # the reference walks pivot_j past the last column here (:548) and fails the same way
            raise IndexError("list index out of range")
        return R.tolist(), pivots, [], []

    def _row_reduce_traced(self, A: np.ndarray, bar: int, full: bool):
        from . import fmt
        m, n = A.shape
        if bar > n:
            raise IndexError("list index out of range")  # the reference indexes A[pi][pj] with pj >= n (:548)
        max_steps = 4 * min(m, bar) + 4
        nsnap = min(max_steps, max(1, self.TRACE_SNAPSHOT_BYTES // (9 * m * n))) if full else 0
        # the reference computes on Python objects: int - int*int stays an int (appendix A.2); the device
        # carries that flag along with every entry
        int_in = [[isinstance(v, (int, np.integer)) and not isinstance(v, bool) for v in row] for row in self.items]
        R, pivots, recs, imask, snaps, snap_m = dense.rref_trace(A, bar, max_snapshots=nsnap, int_mask=int_in)

        def typed(values, mask):
            return [[int(values[i][j]) if mask[i][j] else float(values[i][j]) for j in range(n)] for i in range(m)]

        steps, mats = [], []
        if full:
            mats.append(fmt.make_latex_augmented_matrix(self.items, bar_col=bar))
        for k, (kind, a, b) in enumerate(recs):
            text = self._STEP_TEXT[kind] % ((a, b) if kind == 0 else (a,))
            steps.append((f"{dense.TRACE_KINDS[kind]}{k}", text))
            if full and k < len(snaps):
                mats.append(fmt.make_latex_augmented_matrix(typed(snaps[k], snap_m[k]), bar_col=bar))
        return typed(R, imask), pivots, mats, steps

    def find_preimage_of(self, vec: List[Any], log_matrices: bool = False, log_steps: bool = False,
                         log_result: bool = False, trace=False) -> "Matrix.AffineSubspace | Matrix.NoSolution":
        """All solutions of self * x = vec (linalg.py:632-680).

        The log_* flags select the reference's carrier shapes (:998 vs :888) on the fast path.  With
        trace="steps" / True the reduction runs in the reference's own operation order instead (exact
        zero tests, its float rank artefacts included) and `self.last_trace` = (matrices, steps) holds
        what the reference would have written to its LaTeX log."""
        if self.rows != len(vec):
            raise ValueError("Matrix dimensions must match")  # :642-643
        logged = log_matrices or log_steps or log_result
        A = self._array()
        b = _as_array([[v] for v in vec])
        n = A.shape[1]
        if n == 0:
            raise IndexError("list index out of range")
        aug = np.hstack([A, b])
        if trace:
            typed_aug = [list(row) + [v] for row, v in zip(self.items, vec)]   # entry types matter there
            return self._find_preimage_traced(typed_aug, n, full=(trace != "steps"))
        R, pivots = self._reduce(aug, n)
        # :913-934 with a tolerance on the right-hand side: rows below the rank have exact
        # zero coefficients; their rhs is rounding noise unless the system is inconsistent
        amax = float(np.max(np.abs(aug))) if aug.size else 0.0
        rank = len(pivots)
        xmax = float(np.max(np.abs(R[:rank, n]))) if rank else 0.0
        rhs_tol = 8.0 * dense.EPS64 * max(A.shape) * amax * (1.0 + xmax)
        if rank < A.shape[0] and np.any(np.abs(R[rank:, n]) > rhs_tol):
            return Matrix.NoSolution()
        # :937-999
        col_of_row = {r: c for r, c in pivots}
        pivot_cols = {c for _, c in pivots}
        free = [j for j in range(n) if j not in pivot_cols]
        particular: List[Any] = [0] * n
        for r, c in col_of_row.items():
            particular[c] = float(R[r, n])
        gens = []
        for fj in free:
            g: List[Any] = [0] * n
            g[fj] = 1
            for r, c in col_of_row.items():
                g[c] = -float(R[r, fj])
            gens.append(g)
        if gens:
            gen_mat = Matrix([list(col) for col in zip(*gens)])  # generators are columns (:985)
        else:
            gen_mat = None if logged else Matrix.zero(n, 0)  # :998 vs :888
        return Matrix.AffineSubspace(particular, gen_mat)

    def _find_preimage_traced(self, aug: List[List[Any]], n: int, full: bool):
        """The reference's logging path (linalg.py:655-680): reduction in its own operation order, its exact
        inconsistency test (:913-934) and extraction (:937-999).  The reference writes the step log to
        its global LaTeX logger; here it is kept on the object as `last_trace` = (matrices, steps)."""
        items, pivots, mats, steps = Matrix(aug).row_reduce(bar_col=n, trace=("full" if full else "steps"))
        self.last_trace = (mats, steps)
        for row in items:  # :918-934, exact comparisons as in the reference
            if all(v == 0 for v in row[:n]) and row[n] != 0:
                return Matrix.NoSolution()
        col_of_row = {r: c for r, c in pivots}
        pivot_cols = {c for _, c in pivots}
        particular: List[Any] = [0] * n
        for r, c in col_of_row.items():
            particular[c] = items[r][n]
        gens = []
        for fj in [j for j in range(n) if j not in pivot_cols]:
            g: List[Any] = [0] * n
            g[fj] = 1
            for r, c in col_of_row.items():
                g[c] = -items[r][fj]
            gens.append(g)
        gen_mat = Matrix([list(col) for col in zip(*gens)]) if gens else None  # :985, :998
        return Matrix.AffineSubspace(particular, gen_mat)

    def inverse(self, log_matrices: bool = False, log_steps: bool = False, log_result: bool = False, trace=False):
        """linalg.py:682-743: Matrix, or NoSolution() when singular.  trace: as in find_preimage_of."""
        if self.rows != self.cols:
            raise ValueError("Matrix must be square to invert.")  # :692-693
        A = self._array()
        n = A.shape[0]
        if trace:
            # the reference's logging path (:703-743): [A|I] reduced in its own operation order, left block
            # compared with I to 1e-12, right block returned; the log is kept as `last_trace`
            ident = Matrix.identity(n).items
            items, _piv, mats, steps = Matrix([list(self.items[i]) + ident[i] for i in range(n)]).row_reduce(
                bar_col=n, trace=("steps" if trace == "steps" else "full"))
            self.last_trace = (mats, steps)
            for i in range(n):
                for j in range(n):
                    if abs(items[i][j] - (1.0 if i == j else 0.0)) > 1e-12:
                        return Matrix.NoSolution()  # :737
            return Matrix([row[n:] for row in items])
        X, info, ratio = dense.inv(A)
        if info != 0 or not (ratio > dense.EPS64 * n):
            return Matrix.NoSolution()  # :701 / :737
        return Matrix(X.tolist())

    def determinant(self, log_permutation_details: bool = False, use_optimal: bool = True) -> Any:
        """linalg.py:183-207.  sign * prod(diag U) from the pivoted LU."""
        if self.rows != self.cols:
            raise ValueError("Determinant requires a square matrix")
        if self.rows == 1:
            return self.items[0][0]  # :200-201
        return dense.det(self._array())

    def slogdet(self) -> Tuple[float, float]:
        """(sign, log|det|): the overflow-free form of determinant() (new API)."""
        if self.rows != self.cols:
            raise ValueError("Determinant requires a square matrix")
        return dense.slogdet(self._array())

    def rank(self) -> int:
        """linalg.py:745-747."""
        A = self._array()
        if A.shape[1] == 0:
            return 0
        # partial pivoting for the rank decision: the first-non-zero rule is the reference's order of
        # operations, but its unpivoted growth can lift rounding noise over the tolerance (a 5 x 7 integer
        # matrix of rank 3 came out as 4)
        from ._native import PIVOT_MAX
        return dense.rref(A, bar_col=A.shape[1], pivot_rule=PIVOT_MAX)[2]

    def kernel(self) -> "Matrix.AffineSubspace":
        """linalg.py:749-756."""
        return self.find_preimage_of([0] * self.rows)


def _fmt(v) -> str:
    if isinstance(v, (int, np.integer)):
        return str(int(v))
    f = float(v)
    if f == math.floor(f) and abs(f) < 1e15:
        return str(int(f))
    return repr(f)
