"""Fixture F6: the kernel oracle oracle/hipk_cpu.c pinned DIRECTLY to outputs of the
reference's own panel routines (tests/reference_kernel_cases.py), and the same for the HIP kernels on the GPU."""
import pytest

import reference_kernel_cases as RK
from kernel_harness import Dev, Host


@pytest.mark.parametrize("case", ["update_projection", "update_vwxr", "bortho_gen", "bortho_block"])
def test_oracle_kernels_against_reference_routines(built, case):
    side = Host()
    getattr(RK, "check_" + case)(side)
    side.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["update_projection", "update_vwxr", "bortho_gen", "bortho_block"])
def test_hip_kernels_against_reference_routines(built, case):
    side = Dev()
    getattr(RK, "check_" + case)(side)
    side.close()


# round 5: the same routines of the reference at m = 4 099 and 100 003, k = 15 and 41, blocks of 1 and 8, 0 and 9 locked vectors
@pytest.mark.parametrize("idx", range(len(RK.GOLD["wide"])))
def test_oracle_kernels_against_reference_routines_large_shapes(built, idx):
    side = Host()
    RK.check_wide(side, idx)
    side.close()


@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(len(RK.GOLD["wide"])))
def test_hip_kernels_against_reference_routines_large_shapes(built, idx):
    side = Dev()
    RK.check_wide(side, idx)
    side.close()
