"""The pivot sequence at the BASELINE sizes, pinned against the CPU twin.

tests/golden/ipiv_u11_s1_n{4096,8192,16384}.npz hold, for the matrices of BASELINE configs 2, 3 and 4 (counter-based
`u11` generator, seed 1), the interchange vector, diag(U), the solution of A x = b and sign / log|det| computed by
oracle/lu_twin.c in the build container (tests/golden/gen_ipiv_golden.py).  The device factorisation must choose the
same rows, bit for bit; the values agree to the north-star's 1e-9.
"""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SIZES = [4096, 8192, 16384]


def _load(n):
    return np.load(os.path.join(GOLDEN, f"ipiv_u11_s1_n{n}.npz"))


@pytest.mark.parametrize("n", SIZES)
def test_fixture_is_a_valid_interchange_vector(n):
    """CPU: the committed fixtures are well-formed (and the generator that made their inputs has not drifted:
    the first pivot is the arg-max of the regenerated column 0)."""
    from linalg_solver_amd import gen

    z = _load(n)
    ipiv = z["ipiv"]
    assert int(z["n"]) == n and int(z["info"]) == 0 and ipiv.dtype == np.int32 and ipiv.shape == (n,)
    k = np.arange(n)
    assert np.all(ipiv >= k) and np.all(ipiv < n)
    col0 = gen.fill(gen.U11, 1, n, 1)[:, 0]
    assert int(ipiv[0]) == int(np.argmax(np.abs(col0)))
    assert z["diag_u"].shape == (n,) and np.all(z["diag_u"] != 0) and float(z["diag_u"][0]) == float(col0[ipiv[0]])
    assert float(z["sign"]) in (-1.0, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("n", SIZES)
def test_device_pivot_sequence_equals_the_twin_fixture(n):
    import torch

    from linalg_solver_amd import gen
    from linalg_solver_amd.device import DeviceSolver

    z = _load(n)
    dev = DeviceSolver()
    A = torch.empty(n, n, dtype=torch.float64, device="cuda")
    dev.fill_(A, gen.U11, 1)
    b = torch.empty(n, 1, dtype=torch.float64, device="cuda")
    dev.fill_(b, gen.U11, 1, col_off=gen.RHS_COL)
    ipiv, info = dev.getrf_(A)
    x = b.clone()
    dev.getrs_(A, ipiv, x)
    parts = dev.det_parts(A, ipiv).cpu().numpy()
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    got = ipiv.cpu().numpy()
    assert np.array_equal(got, z["ipiv"]), f"first differing column {int(np.nonzero(got != z['ipiv'])[0][0])}"
    du = torch.diagonal(A).cpu().numpy()
    assert np.max(np.abs(du - z["diag_u"]) / np.abs(z["diag_u"])) < 1e-9       # every diagonal entry, relatively
    xs = x[:, 0].cpu().numpy()
    assert np.max(np.abs(xs - z["x"])) / np.max(np.abs(z["x"])) < 1e-9
    sign, mant, ex = float(parts[0]), float(parts[1]), float(parts[2])
    assert sign == float(z["sign"])
    logabs = np.log(abs(mant)) + ex * np.log(2.0)
    assert abs(logabs - float(z["logabs"])) < 1e-9 * abs(float(z["logabs"]))
