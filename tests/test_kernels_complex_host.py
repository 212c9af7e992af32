"""The plain-C complex restatement of the device layer (oracle/hipk_cpu_complex.c) against numpy, WITHOUT a GPU: the bodies of
tests/test_kernels_complex_gpu.py — which compare the HIP kernels with this restatement AND the restatement with numpy — run
here with the checker on both sides, so that the second comparison (the one that pins the checker) is part of the
`-m "not gpu"` suite too.  Small and medium shapes of each test; the full list runs on the device."""
import numpy as np
import pytest

from primme_amd import _ffi as F
import kernel_harness
import test_kernels_complex_gpu as G

ZDT = [F.HIPK_C64, F.HIPK_C32]


@pytest.fixture()
def checker_on_both_sides(monkeypatch):
    monkeypatch.setattr(G, "Dev", kernel_harness.Host)


@pytest.mark.parametrize("dt", ZDT)
@pytest.mark.parametrize("m,k,L,nx", [(1, 1, 0, 1), (1000, 3, 2, 5), (70001, 12, 5, 4), (4099, 140, 3, 2)])
def test_checker_complex_panel_dots_against_numpy(built, checker_on_both_sides, dt, m, k, L, nx):
    G.test_complex_panel_dots(True, dt, m, k, L, nx)


@pytest.mark.parametrize("dt", ZDT)
@pytest.mark.parametrize("m,k,L,nx", [(1000, 3, 2, 1), (70001, 12, 5, 2), (3001, 300, 0, 3)])
def test_checker_complex_panel_project_against_numpy(built, checker_on_both_sides, dt, m, k, L, nx):
    G.test_complex_panel_project(True, dt, m, k, L, nx)


@pytest.mark.parametrize("dt", ZDT)
def test_checker_complex_column_utilities_against_numpy(built, checker_on_both_sides, dt):
    G.test_complex_column_utilities(True, dt)


@pytest.mark.parametrize("dt", ZDT)
@pytest.mark.parametrize("ncols", [1, 4])
def test_checker_complex_csr_matvec_against_numpy(built, checker_on_both_sides, dt, ncols):
    G.test_complex_csr_matvec(True, dt, ncols)
