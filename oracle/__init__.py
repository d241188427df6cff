"""CPU oracle for the row-reduction / solve path of koskja/linalg-solver.

TEST INFRASTRUCTURE ONLY.  Nothing in ``linalg_solver_amd`` (the product) may
import, link or call anything in this directory.  The only allowed users are
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- and there only as the checker / the timed CPU baseline, never
as the thing shipped.

Parity status: PINNED.  The reference's own test-suite holds no fixture for
this path, so the restatement is pinned against outputs
of the reference itself, produced in the build container by
``tests/golden/gen_golden.py`` (which imports ``/root/reference`` read-only)
and committed as data under ``tests/golden/``.

Contents
  rowreduce.py      generic (any scalar type) restatement of Matrix.row_reduce,
                    _check_inconsistency, _extract_affine_subspace and the
                    find_preimage_of / inverse callers
                    (reference linalg_solver/linalg.py:534-743, 913-999)
  rowreduce_ref.c   the same algorithm for IEEE fp64 in plain C, compiled with
                    -ffp-contract=off so every a/f and a-f*b rounds exactly as
                    CPython's float ops do
  lu_twin.c         partial-pivot LU / solve / inverse / slogdet in plain C: the
                    CPU twin of the GPU algorithm, used to check big sizes
  capi.py           ctypes loader for the two C files (built by oracle/Makefile)
"""
