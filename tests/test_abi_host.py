"""CPU-side checks: the C ABI exports what include/lsx.h declares; host logic of the
Matrix mirror that needs no device (validation, type gate, error behaviour)."""
import os
import re
from fractions import Fraction

import pytest

import linalg_solver_amd as la
from linalg_solver_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lsx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lsx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _native.load()
    declared = _declared_symbols()
    assert len(declared) >= 34
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in lsx.h but not exported by liblsx.so"
    assert set(declared) == set(_native.EXPORTS), set(declared) ^ set(_native.EXPORTS)


def test_last_error_is_a_string():
    assert isinstance(_native.load().lsx_last_error(), bytes)


def test_matrix_validation_matches_reference_messages():
    # linalg.py:14-32
    with pytest.raises(ValueError, match="Matrix cannot be empty"):
        la.Matrix([])
    with pytest.raises(ValueError, match="list of lists"):
        la.Matrix([(1, 2)])
    with pytest.raises(ValueError, match="same length"):
        la.Matrix([[1, 2], [3]])
    with pytest.raises(ValueError, match="rows cannot be empty"):
        la.Matrix([[], [1]])
    z = la.Matrix.zero(3, 0)
    assert z.rows == 3 and z.cols == 0
    assert la.Matrix.identity(2).items == [[1, 0], [0, 1]]
    assert la.Matrix.new_vector([1, 2]).items == [[1], [2]]
    assert la.Matrix([[1, 2], [3, 4]]).transpose().items == [[1, 3], [2, 4]]


def test_shape_errors_raise_before_any_device_work():
    m = la.Matrix([[1.0, 2.0], [3.0, 4.0]])
    with pytest.raises(ValueError, match="dimensions must match"):  # linalg.py:642-643
        m.find_preimage_of([1.0])
    with pytest.raises(ValueError, match="square"):  # linalg.py:692-693
        la.Matrix([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]]).inverse()


def test_non_numeric_entries_are_rejected_not_routed_to_a_cpu_path():
    m = la.Matrix([[Fraction(1, 2), 1], [1, 1]])
    for call in (lambda: m.row_reduce(), lambda: m.inverse(), lambda: m.find_preimage_of([1, 2]),
                 lambda: m.rank(), lambda: m.determinant()):
        with pytest.raises(TypeError, match="int/float"):
            call()


def test_backend_selector_has_no_cpu_route():
    # SURVEY 8b's opt-in selector: "auto" / "hip" are this package; "cpu" is refused, not served in Python
    assert la.Matrix([[1, 2], [3, 4]]).backend == "hip"
    assert la.Matrix([[1, 2], [3, 4]], backend="hip").backend == "hip"
    with pytest.raises(NotImplementedError, match="no CPU backend"):
        la.Matrix([[1, 2], [3, 4]], backend="cpu")
    with pytest.raises(ValueError, match="unknown backend"):
        la.Matrix([[1, 2], [3, 4]], backend="cuda")


def test_result_carriers():
    s = la.Matrix.AffineSubspace([1.0, 0], la.Matrix([[-1.0], [1]]))
    assert s.get_one() == [1.0, 0] and s.dim() == 1 and s.basis() == [[-1.0, 1]]
    assert repr(la.Matrix.NoSolution()) == "NoSolution()"
    with pytest.raises(AttributeError):
        la.Matrix.AffineSubspace([1.0], None).dim()  # same failure as the reference (:500)


def test_generator_is_deterministic_and_in_range():
    from linalg_solver_amd import gen
    a = gen.fill(gen.INT5, 3, 5, 7)
    assert a.shape == (5, 7) and set(a.ravel()) <= set(float(v) for v in range(-5, 6))
    b = gen.fill(gen.INT5, 3, 2, 3, row_off=1, col_off=2)
    assert (b == a[1:3, 2:5]).all()
    u = gen.fill(gen.U11, 9, 64, 64)
    assert (u >= -1).all() and (u < 1).all() and abs(u.mean()) < 0.1


def test_bench_lapack_baseline_runs_in_a_child_process_and_prints_one_json_line():
    """bench.py times LAPACK's dgetrf beside the oracle -- in a CHILD process (`bench.py --lapack-child`), because a crash in
    a BLAS thread pool (seen on rank 0 under torch.distributed.run, which exports OMP_NUM_THREADS=1) must not take the
    benchmark line with it; and only at N = 1.  The child needs no GPU and no torch."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="1")   # what a rank under the launcher sees; the parent strips it for the child
    env.pop("OMP_NUM_THREADS")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--lapack-child"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-500:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert ("value" in d and d["value"] > 0 and d["n"] == 4096) or "error" in d
    src = open(os.path.join(root, "bench.py")).read()
    assert 'if not args.no_cpu and world == 1' in src   # the CPU baseline is rank 0's job at N = 1 only
