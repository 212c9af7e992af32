"""The drop-in boundary exercised from plain C (no Python, no HIP headers in the caller): the compiled
programs under examples/ link libprimme_amd.so, solve the reference's example problems on the GPU and
return 0 when the results match the analytic values."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("prog", ["ex_eigs_dhip", "ex_svds_dhip", "ex_svds_zhip", "ex_eigs_zhip", "ex_eigs_dseq_host", "ex_svds_dseq_host", "test_hip_wrapper"])
def test_c_example(built, prog):
    exe = os.path.join(ROOT, "examples", prog)
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "returned 0" in out.stdout
