"""Hermitian problems on the MI355X: hip_zprimme / hip_cprimme through the C ABI, against the
oracle-backed host run on the same inputs, the reference driver's complex cases, and BASELINE
configs[3]'s matrix family through size-independent properties."""
import ctypes as C
import numpy as np
import pytest

from primme_amd import _ffi as F
from primme_amd import problems
from checkers import Operator, eigsh
import reference_driver_cases as RD
from test_complex_host import hermitian_band

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_pair_rotate_kernel(built, dtype):
    import torch
    lib = F.load_product()
    ctx = C.c_void_p()
    assert lib.hipk_ctx_create(C.byref(ctx), None) == 0
    npairs, nx, ld = 100003, 3, 200010
    x = np.random.default_rng(0).standard_normal((nx, ld)).astype(dtype)
    X = torch.from_numpy(x).cuda()
    Y = torch.zeros_like(X)
    dt = F.HIPK_F64 if dtype == np.float64 else F.HIPK_F32
    assert lib.hipk_pair_rotate(ctx, dt, npairs, C.c_void_p(X.data_ptr()), ld, C.c_void_p(Y.data_ptr()), ld, nx) == 0
    lib.hipk_sync(ctx)
    y = Y.cpu().numpy()
    z = x[:, :2 * npairs].reshape(nx, npairs, 2)
    want = np.stack([-z[..., 1], z[..., 0]], axis=-1).reshape(nx, 2 * npairs)
    assert np.array_equal(y[:, :2 * npairs], want) and not y[:, 2 * npairs:].any()
    lib.hipk_ctx_destroy(ctx)


@pytest.mark.parametrize("dtype,eps,tol", [(np.complex128, 1e-10, 1e-10), (np.complex64, 1e-4, 2e-4)])
@pytest.mark.parametrize("method,b", [("GD_plusK", 1), ("GD_plusK", 4), ("JDQMR", 1)])
def test_hip_hermitian_against_oracle(built, dtype, eps, tol, method, b):
    n = 3000
    A, csr = hermitian_band(n, seed=1)
    w = np.linalg.eigvalsh(A)
    kw = dict(numEvals=4, eps=eps, method=method, maxBlockSize=b, iseed=(1, 2, 3, 5))
    r = eigsh(Operator(n, csr=csr), backend="hip", dtype=dtype, **kw)
    h = eigsh(Operator(n, csr=csr), backend="hostcheck", dtype=dtype, **kw)
    assert r.ret == 0 and h.ret == 0 and r.initSize == 4
    aN = r.params["aNorm"]
    assert np.max(np.abs(r.evals - w[:4])) <= tol * aN
    assert np.max(np.abs(r.evals.astype(np.float64) - h.evals)) <= tol * aN
    X = r.evecs.astype(np.complex128)
    assert np.max(np.abs(X.conj().T @ X - np.eye(4))) <= (1e-9 if dtype == np.complex128 else 1e-4)
    res = np.linalg.norm(A @ X - X * r.evals.astype(np.float64), axis=0)
    assert np.all(res <= 1.5 * eps * aN + 10 * np.finfo(r.evals.dtype).eps * aN)


@pytest.mark.parametrize("name", ["test_101", "test_105", "test_106"])
def test_hip_reference_driver_complex_case(built, name):
    rp, ci, va, n = RD.mhd()
    case = RD.CASES_Z[name]
    r = eigsh(Operator(n, csr=(rp, ci, va)), backend="hip", dtype=np.complex128, **case["kw"])
    X = RD.read_sol_z(case["sol"], n)
    bad = RD.check_solution(lambda v: problems.csr_matvec_numpy(rp, ci, va, v.reshape(-1, 1)).ravel(),
                            r.evals, r.evecs, r.resNorms, r.params["aNorm"], case["kw"]["eps"], X)
    assert r.ret == 0 and r.initSize == case["kw"]["numEvals"] and not bad, bad


def test_hip_constraints_and_guesses(built):
    n = 2000
    A, csr = hermitian_band(n, seed=4)
    w, U = np.linalg.eigh(A)
    Q = U[:, :2] * np.exp(1j * np.array([0.3, 1.1]))
    r = eigsh(Operator(n, csr=csr), backend="hip", dtype=np.complex128, numEvals=3, eps=1e-10, constraints=Q, iseed=(1, 1, 1, 1))
    assert r.ret == 0 and r.initSize == 3
    assert np.max(np.abs(r.evals - w[2:5])) <= 1e-9 * r.params["aNorm"]
    assert np.max(np.abs(Q.conj().T @ r.evecs)) <= 1e-8


def test_hip_config4_family_properties(built):
    """BASELINE configs[3]'s band matrix at n = 400 000 (the per-GPU slab of the 8-GPU run is
    500 000): 6 largest, block size 4, GD+k; true residuals, orthonormality, and the same
    eigenvalues as a 20 000-row leading block gives for this Toeplitz-like band to 1e-6."""
    n = 400_000
    rp, ci, va = problems.hermitian_banded_csr(n)
    r = eigsh(Operator(n, csr=(rp, ci, va)), backend="hip", dtype=np.complex128, numEvals=6, target="largest", eps=1e-8,
              maxBlockSize=4, maxBasisSize=20, minRestartSize=8, method="GD_plusK", iseed=(2, 3, 5, 7))
    assert r.ret == 0 and r.initSize == 6
    aN = r.params["aNorm"]
    AX = problems.csr_matvec_numpy(rp, ci, va, r.evecs)
    res = np.linalg.norm(AX - r.evecs * r.evals, axis=0)
    assert np.all(res <= 1.5e-8 * aN)
    assert np.allclose(res, r.resNorms, rtol=0.2, atol=1e-12 * aN)
    assert np.max(np.abs(r.evecs.conj().T @ r.evecs - np.eye(6))) <= 1e-8
    assert np.all(np.diff(r.evals) <= 1e-12) and r.evals[0] < 2 + 2 * (1 / 2 + 1 / 3 + 1 / 4) + 1.0


import complex_fixture_cases as ZF


@pytest.mark.parametrize("name", sorted(ZF.FIX))
def test_hip_native_complex_reproduces_zprimme_fixture(built, name):
    """hip_zprimme / hip_cprimme on complex panels (csrc/hipk_complex.hip under the complex host solver) against the
    committed outputs of the real reference's zprimme / cprimme: eigenvalues, residual norms, true residuals and — in
    double precision, extremal targets, exactly — the outer-iteration, matvec, restart and preconditioner counts."""
    ZF.check(name, "hip")


def test_hip_native_complex_halves_the_real_form(built):
    """The same Hermitian problem through the native path and through the real-equivalent form (every eigenvalue
    doubled): same pairs, about half the operator applications."""
    n = 3000
    A, csr = hermitian_band(n, seed=1)
    kw = dict(numEvals=4, eps=1e-10, method="GD_plusK", maxBlockSize=4, iseed=(1, 2, 3, 5))
    nat = eigsh(Operator(n, csr=csr), backend="hip", dtype=np.complex128, **kw)
    rea = eigsh(Operator(n, csr=csr), backend="hip", dtype=np.complex128, complex_form="real", **kw)
    assert nat.ret == 0 and rea.ret == 0
    assert np.max(np.abs(nat.evals - rea.evals)) <= 1e-9 * nat.params["aNorm"]
    assert rea.stats["numMatvecs"] >= 1.6 * nat.stats["numMatvecs"]


def test_hip_native_complex_dynamic_method(built):
    """method = DYNAMIC through hip_zprimme on complex panels: the pairs of the dense solve, true residuals."""
    n = 3000
    A, csr = hermitian_band(n, seed=7)
    w = np.linalg.eigvalsh(A)
    r = eigsh(Operator(n, csr=csr), backend="hip", dtype=np.complex128, numEvals=6, eps=1e-10, method="DYNAMIC", iseed=(1, 2, 3, 5))
    assert r.ret == 0 and r.initSize == 6 and r.params["dynamicMethodSwitch"] in (-1, -2, -3)
    aN = r.params["aNorm"]
    assert np.max(np.abs(r.evals - w[:6])) <= 1e-9 * aN
    AX = A @ r.evecs
    assert np.all(np.linalg.norm(AX - r.evecs * r.evals, axis=0) <= 1.5e-10 * aN)


@pytest.mark.parametrize("K,mr", [(260, 60), (520, 100)])
def test_hip_complex_wide_basis(built, K, mr):
    """maxBasisSize beyond 255 on complex panels on the device (round 6): the restart through the sliced complex update, against
    numpy's dense spectrum and — at 260 — the CPU checker's history (tied to live zprimme in tests/test_complex_host.py)."""
    n = 3000
    A, csr = hermitian_band(n, seed=1)
    kw = dict(numEvals=6, eps=1e-10, iseed=(1, 2, 3, 5), maxBasisSize=K, minRestartSize=mr, maxBlockSize=1, dtype=np.complex128)
    a = eigsh(Operator(n, csr=csr), backend="hip", **kw)
    assert a.ret == 0 and a.params["maxBasisSize"] == K and a.stats["numRestarts"] >= 1
    aN = a.params["aNorm"]
    assert np.max(np.abs(a.evals - np.linalg.eigvalsh(A)[:6])) <= 1e-9 * aN
    X = a.evecs
    assert np.max(np.abs(X.conj().T @ X - np.eye(6))) <= 1e-9
    assert np.all(np.linalg.norm(A @ X - X * a.evals, axis=0) <= 1.5e-10 * aN)
    if K == 260:
        b = eigsh(Operator(n, csr=csr), backend="hostcheck", **kw)
        assert a.stats["numRestarts"] == b.stats["numRestarts"]
        assert abs(a.stats["numOuterIterations"] - b.stats["numOuterIterations"]) <= 0.03 * b.stats["numOuterIterations"]


@pytest.mark.parametrize("kw", [dict(numEvals=4, eps=1e-9, target="largest", method="GD_plusK"), dict(numEvals=4, eps=1e-9, target="largest", maxBlockSize=2),
                                dict(numEvals=4, eps=1e-9, target="largest", method="JDQMR", locking=1),
                                dict(numEvals=4, eps=1e-9, target="largest", method="JDQMR_ETol", locking=1, precond="jacobi"),
                                dict(numEvals=3, eps=1e-9, target="smallest", method="JDQMR", locking=1, maxBlockSize=2)])
def test_hip_generalized_hermitian(built, kw):
    """Generalised Hermitian problems on the device (round 6; the live-reference leg is tests/test_complex_host.py): scipy's dense
    truth, B-orthonormal vectors, true residuals, the CPU checker's counts to 5 %."""
    from test_complex_host import _generalized_hermitian
    kw = dict(kw, iseed=(2, 3, 5, 7))
    a, A, B, truth = _generalized_hermitian("hip", kw)
    b, _, _, _ = _generalized_hermitian("hostcheck", kw)
    assert a.ret == b.ret == 0
    aN = a.params["aNorm"]
    assert np.max(np.abs(np.sort(a.evals) - np.sort(truth))) <= 1e-10 * aN
    X = a.evecs
    assert np.max(np.abs(X.conj().T @ B @ X - np.eye(X.shape[1]))) <= 1e-9
    assert np.max(np.abs(np.linalg.norm(A @ X - (B @ X) * a.evals, axis=0) - a.resNorms)) <= 1e-9 * aN
    tol = 0.15 if "JDQMR" in kw.get("method", "") else 0.05
    assert abs(a.stats["numOuterIterations"] - b.stats["numOuterIterations"]) <= max(2, tol * b.stats["numOuterIterations"])

