"""Singular value path on the MI355X (hip_dprimme_svds / hip_sprimme_svds through the C ABI):
against the oracle on the same inputs, against the reference driver's regression cases, and at
BASELINE configs[4] scale through size-independent properties."""
import numpy as np
import pytest

from primme_amd import problems
from checkers import svds, transpose_csr
import reference_driver_cases as RD
from test_svds_host import _rect, _rect_complex

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,n,k,target", [(300, 200, 5, "largest"), (200, 300, 4, "largest"), (300, 200, 3, "smallest"),
                                          (5000, 3000, 6, "largest")])
def test_hip_svds_against_oracle(built, m, n, k, target):
    A, csr = _rect(m, n)
    s = np.linalg.svd(A, compute_uv=False)
    want = s[:k] if target == "largest" else s[::-1][:k]
    r = svds(m, n, csr, numSvals=k, target=target, eps=1e-10, methodStage1="GD_plusK", backend="hip")
    h = svds(m, n, csr, numSvals=k, target=target, eps=1e-10, methodStage1="GD_plusK", backend="hostcheck")
    assert r.ret == 0 and h.ret == 0 and r.initSize == k
    assert np.max(np.abs(r.svals - want)) <= 1e-10 * s[0]
    assert np.max(np.abs(r.svals - h.svals)) <= 1e-10 * s[0]
    assert np.all(r.resNorms <= 1e-10 * r.params["aNorm"] * (1 + 1e-6))
    assert np.linalg.norm(A @ r.V - r.U * r.svals) <= 1e-8 * s[0]
    assert np.linalg.norm(r.U.T @ r.U - np.eye(k)) <= 1e-8 and np.linalg.norm(r.V.T @ r.V - np.eye(k)) <= 1e-8
    assert abs(r.stats["numOuterIterations"] - h.stats["numOuterIterations"]) <= max(2, 0.03 * h.stats["numOuterIterations"])


@pytest.mark.parametrize("method", ["hybrid", "augmented"])
def test_hip_svds_hybrid_and_augmented(built, method):
    A, csr = _rect(300, 200)
    s = np.linalg.svd(A, compute_uv=False)
    r = svds(300, 200, csr, numSvals=4, target="largest", eps=1e-9, method=method, methodStage1="GD_plusK", backend="hip")
    h = svds(300, 200, csr, numSvals=4, target="largest", eps=1e-9, method=method, methodStage1="GD_plusK", backend="hostcheck")
    assert r.ret == 0 and h.ret == 0 and r.initSize == 4
    assert np.max(np.abs(r.svals - s[:4])) <= 1e-9 * s[0] and np.max(np.abs(r.svals - h.svals)) <= 1e-9 * s[0]
    assert np.linalg.norm(A @ r.V - r.U * r.svals) <= 1e-7 * s[0]
    assert abs(r.stats["numOuterIterations"] - h.stats["numOuterIterations"]) <= max(3, 0.1 * h.stats["numOuterIterations"])


def test_hip_svds_float(built):
    A, csr = _rect(400, 250)
    s = np.linalg.svd(A, compute_uv=False)
    r = svds(400, 250, csr, numSvals=3, eps=1e-4, methodStage1="GD_plusK", backend="hip", dtype=np.float32)
    assert r.ret == 0 and np.max(np.abs(r.svals - s[:3])) <= 1e-4 * s[0]


@pytest.mark.parametrize("name", sorted(RD.SVDS_CASES))
def test_hip_svds_reference_driver_case(built, name):
    case = RD.SVDS_CASES[name]
    rp, ci, va, m, n = RD.svds_matrix(case.get("matrix", "rect.mtx"))
    rpT, ciT, vaT = transpose_csr(m, n, rp, ci, va)
    r = svds(m, n, (rp, ci, va), backend="hip", **{"methodStage1": "GD_plusK", **case["kw"]})
    assert r.ret == 0 and r.initSize == case["kw"]["numSvals"]
    XU, _ = RD.read_sol_svds(case["sol"], m, n)
    bad = RD.check_solution_svds(lambda v: problems.csr_matvec_numpy(rp, ci, va, v.reshape(-1, 1)).ravel(),
                                 lambda u: problems.csr_matvec_numpy(rpT, ciT, vaT, u.reshape(-1, 1)).ravel(),
                                 r.svals, r.U, r.V, r.resNorms, r.params["aNorm"], case["kw"]["eps"], XU)
    assert not bad, bad


def test_hip_svds_interior_target_with_blocks(built):
    """closest_abs singular values with a block: explicit_I + refined extraction in the eigensolver stage."""
    A, csr = _rect(60, 40)
    s = np.linalg.svd(A, compute_uv=False)
    r = svds(60, 40, csr, numSvals=2, target="closest_abs", targetShifts=[7.0], maxBlockSize=2, eps=1e-9, backend="hip")
    assert r.ret == 0 and r.initSize == 2
    want = s[np.argsort(np.abs(s - 7.0))][:2]
    assert np.max(np.abs(np.sort(r.svals) - np.sort(want))) <= 1e-8 * s[0]


def test_hip_svds_config5_shape(built):
    """BASELINE configs[4] on one GPU at 1/8 of the rows: A (1 000 000 x 250 000), row i has 5
    nonzeros at columns (i*p_q + q) mod n with values 1 + ((i+q) mod 13)/13;
    10 largest singular triplets through A'A.  Checked through properties: orthonormal U and V,
    A v = sigma u, A' u = sigma v to the reported residual, sigma_1 against a power-iteration bound."""
    m, n, k = 1_000_000, 250_000, 10
    rp, ci, va = problems.svds_synthetic_csr(m, n)
    r = svds(m, n, (rp, ci, va), numSvals=k, eps=1e-8, methodStage1="GD_plusK", backend="hip")
    assert r.ret == 0 and r.initSize == k
    rpT, ciT, vaT = transpose_csr(m, n, rp, ci, va)
    AV = problems.csr_matvec_numpy(rp, ci, va, r.V)
    AtU = problems.csr_matvec_numpy(rpT, ciT, vaT, r.U)
    res = np.sqrt(np.sum((AV - r.U * r.svals) ** 2, axis=0) + np.sum((AtU - r.V * r.svals) ** 2, axis=0))
    tol = 1e-8 * r.params["aNorm"]
    assert np.all(res <= 10 * tol) and np.all(r.resNorms <= tol * (1 + 1e-6))
    assert np.linalg.norm(r.U.T @ r.U - np.eye(k)) <= 1e-8 and np.linalg.norm(r.V.T @ r.V - np.eye(k)) <= 1e-8
    assert np.all(np.diff(r.svals) <= 1e-12 * r.svals[0])
    # sigma_1 >= |A x| / |x| for any x: a few power iterations from the ones vector
    x = np.ones((n, 1))
    for _ in range(5):
        x = problems.csr_matvec_numpy(rpT, ciT, vaT, problems.csr_matvec_numpy(rp, ci, va, x))
        x /= np.linalg.norm(x)
    lower = np.linalg.norm(problems.csr_matvec_numpy(rp, ci, va, x))
    assert r.svals[0] >= lower * (1 - 1e-12)


@pytest.mark.parametrize("form", ["native", "real_equivalent"])
@pytest.mark.parametrize("m,n,k,target,method,dtype,eps", [(1200, 800, 4, "largest", "normalequations", np.complex128, 1e-10),
                                                           (800, 1200, 3, "largest", "hybrid", np.complex128, 1e-10),
                                                           (600, 400, 3, "smallest", "hybrid", np.complex128, 1e-9),
                                                           (900, 700, 3, "largest", "normalequations", np.complex64, 1e-4)])
def test_hip_complex_svds(built, m, n, k, target, method, dtype, eps, form, monkeypatch):
    """hip_zprimme_svds / hip_cprimme_svds: complex singular triplets on the device — the native complex front end (default since
    round 5: csrc/svds_main.c on complex panels) and the real-equivalent form (PRIMME_AMD_COMPLEX_REAL_FORM=1,
    csrc/svds_complex.c) — against numpy's dense SVD and the checker run of the same call"""
    if form == "real_equivalent":
        monkeypatch.setenv("PRIMME_AMD_COMPLEX_REAL_FORM", "1")
    Z, csr = _rect_complex(m, n)
    s = np.linalg.svd(Z, compute_uv=False)
    want = s[:k] if target == "largest" else s[::-1][:k]
    r = svds(m, n, csr, numSvals=k, target=target, eps=eps, method=method, backend="hip", dtype=dtype)
    assert r.ret == 0 and r.initSize == k
    tol = 10 * eps * s[0]
    assert np.max(np.abs(np.sort(r.svals) - np.sort(want))) <= tol
    assert np.linalg.norm(Z @ r.V - r.U * r.svals) <= 100 * tol and np.linalg.norm(Z.conj().T @ r.U - r.V * r.svals) <= 100 * tol
    assert np.linalg.norm(r.V.conj().T @ r.V - np.eye(k)) <= 1e3 * eps and np.linalg.norm(r.U.conj().T @ r.U - np.eye(k)) <= 1e4 * eps
    if dtype == np.complex128:
        h = svds(m, n, csr, numSvals=k, target=target, eps=eps, method=method, backend="hostcheck", dtype=dtype)
        assert h.ret == 0 and np.max(np.abs(np.sort(r.svals) - np.sort(h.svals))) <= tol
