"""Hermitian problems (hip_zprimme / hip_cprimme, csrc/eigs_complex.c) on the CPU checker back end:
the product's host code over oracle/hipk_cpu.c, compared with dense truth, with the live reference
zprimme / cprimme (oracle/_ref) and with the reference driver's complex regression cases
(tests/tests/test_10N on mhd1280b.mtx, stored vectors sol_10N_doublecomplex)."""
import os
import numpy as np
import pytest

from primme_amd import _ffi as F

import checkers
from primme_amd import problems
from checkers import Operator, eigsh
import reference_driver_cases as RD


def hermitian_band(n, seed=0, band=4):
    rng = np.random.default_rng(seed)
    A = np.zeros((n, n), dtype=np.complex128)
    for k in range(1, band):
        v = (rng.standard_normal(n - k) + 1j * rng.standard_normal(n - k)) * 0.3
        A += np.diag(v, k) + np.diag(v.conj(), -k)
    A += np.diag(np.arange(1, n + 1) * 0.5 + rng.standard_normal(n) * 0.1)
    rows, cols = np.nonzero(A)
    rp = np.zeros(n + 1, dtype=np.int32)
    np.add.at(rp, rows + 1, 1)
    return A, (np.cumsum(rp).astype(np.int32), cols.astype(np.int32), A[rows, cols])


def test_real_equivalent_expansion(built):
    """primme_amd_csr_complex_to_real: M [re,im interleaved] = A z, M symmetric, spectrum doubled."""
    import ctypes as C
    lib = checkers.load_hostcheck()
    n = 40
    A, (rp, ci, va) = hermitian_band(n, seed=3)
    o = [C.c_void_p(), C.c_void_p(), C.c_void_p()]
    assert lib.primme_amd_csr_complex_to_real(n, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                                              va.ctypes.data_as(C.c_void_p), *[C.byref(x) for x in o]) == 0
    rp2 = np.ctypeslib.as_array(C.cast(o[0], C.POINTER(C.c_int32)), shape=(2 * n + 1,)).copy()
    nnz2 = int(rp2[-1])
    assert nnz2 == 4 * len(va)
    ci2 = np.ctypeslib.as_array(C.cast(o[1], C.POINTER(C.c_int32)), shape=(nnz2,)).copy()
    va2 = np.ctypeslib.as_array(C.cast(o[2], C.POINTER(C.c_double)), shape=(nnz2,)).copy()
    for h in o: lib.primme_amd_host_free(h)
    M = np.zeros((2 * n, 2 * n))
    M[np.repeat(np.arange(2 * n), np.diff(rp2)), ci2] = va2
    assert np.array_equal(M, M.T)
    z = np.random.default_rng(0).standard_normal(n) + 1j * np.random.default_rng(1).standard_normal(n)
    assert np.allclose((M @ z.view(np.float64)).view(np.complex128), A @ z, rtol=0, atol=1e-12)
    assert np.allclose(np.linalg.eigvalsh(M)[::2], np.linalg.eigvalsh(A), atol=1e-11)
    for i in range(2 * n):
        assert np.all(np.diff(ci2[rp2[i]:rp2[i + 1]]) > 0)     # sorted rows, as the kernels expect


@pytest.mark.parametrize("dtype,eps,tol", [(np.complex128, 1e-10, 1e-10), (np.complex64, 1e-4, 2e-4)])
@pytest.mark.parametrize("target", ["smallest", "largest"])
def test_hermitian_against_dense_truth(built, dtype, eps, tol, target):
    n = 300
    A, csr = hermitian_band(n, seed=1)
    w = np.linalg.eigvalsh(A)
    r = eigsh(Operator(n, csr=csr), backend="hostcheck", dtype=dtype, numEvals=4, eps=eps, target=target, iseed=(1, 2, 3, 5))
    assert r.ret == 0 and r.initSize == 4
    want = w[:4] if target == "smallest" else w[::-1][:4]
    aN = r.params["aNorm"]
    assert np.max(np.abs(r.evals - want)) <= tol * aN
    X = r.evecs.astype(np.complex128)
    assert np.max(np.abs(X.conj().T @ X - np.eye(4))) <= (1e-9 if dtype == np.complex128 else 1e-4)
    res = np.linalg.norm(A @ X - X * r.evals.astype(np.float64), axis=0)
    assert np.all(res <= 1.5 * eps * aN + 10 * np.finfo(r.evals.dtype).eps * aN)
    assert np.allclose(res, r.resNorms, rtol=0.5, atol=20 * np.finfo(r.evals.dtype).eps * aN)


def test_degenerate_hermitian_spectrum(built):
    """Multiplicity-2 eigenvalues of A are multiplicity 4 of the real form: the final complex
    Gram-Schmidt sweep must still hand back numEvals independent, orthonormal eigenvectors."""
    n = 120
    A1, _ = hermitian_band(n // 2, seed=5)
    A = np.kron(np.eye(2), A1)                                    # every eigenvalue twice
    rows, cols = np.nonzero(A)
    rp = np.zeros(n + 1, dtype=np.int32); np.add.at(rp, rows + 1, 1)
    csr = (np.cumsum(rp).astype(np.int32), cols.astype(np.int32), A[rows, cols])
    w = np.linalg.eigvalsh(A)
    r = eigsh(Operator(n, csr=csr), backend="hostcheck", dtype=np.complex128, numEvals=6, eps=1e-10, maxBlockSize=2, iseed=(7, 7, 7, 7))
    assert r.ret == 0 and r.initSize == 6
    assert np.max(np.abs(np.sort(r.evals) - w[:6])) <= 1e-9 * r.params["aNorm"]
    X = r.evecs
    assert np.max(np.abs(X.conj().T @ X - np.eye(6))) <= 1e-8
    assert np.max(np.linalg.norm(A @ X - X * r.evals, axis=0)) <= 1e-8 * r.params["aNorm"]


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
@pytest.mark.parametrize("method,extra", [("GD_plusK", {}), ("JDQMR", {}), ("GD_Olsen_plusK", dict(precond="jacobi")),
                                          ("DEFAULT_MIN_TIME", dict(target="closest_abs", targetShifts=[40.3]))])
def test_hermitian_against_live_reference(built, method, extra):
    """Same eigenvalues and residual norms as zprimme (the iteration counts differ by design:
    the real-equivalent form carries every eigenvalue twice)."""
    n = 240
    A, csr = hermitian_band(n, seed=2)
    kw = dict(numEvals=3, eps=1e-10, method=method, iseed=(3, 1, 4, 1))
    kw.update(extra)
    ref = eigsh(Operator(n, csr=csr), backend="reference", dtype=np.complex128, **kw)
    got = eigsh(Operator(n, csr=csr), backend="hostcheck", dtype=np.complex128, **kw)
    assert ref.ret == 0 and got.ret == 0 and got.initSize == 3
    aN = ref.params["aNorm"]
    assert np.max(np.abs(np.sort(got.evals) - np.sort(ref.evals))) <= 1e-10 * aN
    assert np.all(got.resNorms <= 1e-10 * aN * 1.01)
    # the eigenvectors span the same lines
    order_g, order_r = np.argsort(got.evals), np.argsort(ref.evals)
    for a, b in zip(order_g, order_r):
        assert abs(abs(np.vdot(got.evecs[:, a], ref.evecs[:, b])) - 1.0) <= 1e-7


def test_constraints_and_initial_guesses(built):
    """numOrthoConst complex constraint vectors become [Q | iQ] of the real problem."""
    n = 200
    A, csr = hermitian_band(n, seed=4)
    w, U = np.linalg.eigh(A)
    Q = U[:, :2] * np.exp(1j * np.array([0.3, 1.1]))             # arbitrary complex phases
    v0 = U[:, 2:5] * np.exp(1j * np.array([0.0, 2.0, -1.0]))     # the wanted vectors themselves
    r = eigsh(Operator(n, csr=csr), backend="hostcheck", dtype=np.complex128, numEvals=3, eps=1e-10, constraints=Q, v0=v0)
    assert r.ret == 0 and r.initSize == 3
    assert np.max(np.abs(r.evals - w[2:5])) <= 1e-9 * r.params["aNorm"]
    assert np.max(np.abs(Q.conj().T @ r.evecs)) <= 1e-8
    # the guesses [X0 | iX0] are the whole wanted invariant subspace of the real form: nothing to iterate
    assert r.stats["numMatvecs"] <= 12     # 2 x 3 to verify the guesses, 3 for the final residual norms
    r2 = eigsh(Operator(n, csr=csr), backend="hostcheck", dtype=np.complex128, numEvals=3, eps=1e-10, constraints=Q, iseed=(1, 1, 1, 1))
    assert r2.ret == 0 and np.max(np.abs(r2.evals - w[2:5])) <= 1e-9 * r2.params["aNorm"]
    assert np.max(np.abs(Q.conj().T @ r2.evecs)) <= 1e-8


def _run_z_case(name, backend):
    rp, ci, va, n = RD.mhd()
    case = RD.CASES_Z[name]
    r = eigsh(Operator(n, csr=(rp, ci, va)), backend=backend, dtype=np.complex128, **case["kw"])
    X = RD.read_sol_z(case["sol"], n)
    bad = RD.check_solution(lambda v: problems.csr_matvec_numpy(rp, ci, va, v.reshape(-1, 1)).ravel(),
                            r.evals, r.evecs, r.resNorms, r.params["aNorm"], case["kw"]["eps"], X)
    return r, bad


@pytest.mark.parametrize("name", sorted(RD.CASES_Z))
def test_reference_driver_complex_case(built, name):
    """tests/tests/test_10N through hip_zprimme's host code; accepted by the reference driver's
    check_solution against the reference's stored eigenvectors."""
    r, bad = _run_z_case(name, "hostcheck")
    assert r.ret == 0 and r.initSize == RD.CASES_Z[name]["kw"]["numEvals"]
    assert not bad, bad


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
@pytest.mark.parametrize("name", ["test_101", "test_106"])
def test_reference_driver_complex_case_pins_the_checker(built, name):
    r, bad = _run_z_case(name, "reference")
    assert r.ret == 0 and not bad, bad
    h, _ = _run_z_case(name, "hostcheck")
    assert np.max(np.abs(np.sort(h.evals) - np.sort(r.evals))) <= 1e-10 * r.params["aNorm"]


def _generalized_hermitian(backend, kw, n=600):
    import scipy.linalg as sl
    import scipy.sparse as sp
    rp, ci, va = problems.hermitian_banded_csr(n)
    brp, bci, bva = problems.hermitian_mass_matrix_csr(n)
    r = eigsh(Operator(n, csr=(rp, ci, va)), backend=backend, mass=Operator(n, csr=(brp, bci, bva)), dtype=np.complex128, **kw)
    A = sp.csr_matrix((va, ci, rp), shape=(n, n)).toarray()
    B = sp.csr_matrix((bva, bci, brp), shape=(n, n)).toarray()
    w = sl.eigh(A, B, eigvals_only=True)
    k = kw["numEvals"]
    truth = w[::-1][:k] if kw.get("target") == "largest" else w[:k]
    return r, A, B, truth


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
@pytest.mark.parametrize("kw", [dict(numEvals=4, eps=1e-9, target="largest", method="GD_plusK"), dict(numEvals=4, eps=1e-9, target="largest", maxBlockSize=2),
                                dict(numEvals=3, eps=1e-9, target="smallest", locking=0),
                                # the JDQMR inner solver with B on complex panels (complex projector coefficients, real QMR recurrences)
                                dict(numEvals=4, eps=1e-9, target="largest", method="JDQMR", locking=1),
                                dict(numEvals=4, eps=1e-9, target="largest", method="JDQMR_ETol", locking=1, precond="jacobi"),
                                dict(numEvals=3, eps=1e-9, target="smallest", method="JDQMR", locking=1, maxBlockSize=2),
                                dict(numEvals=3, eps=1e-9, target="smallest", method="JDQMR", locking=0)])
def test_generalized_hermitian_against_live_reference(built, kw):
    """Generalised HERMITIAN problems A x = lambda B x (round 6): zprimme with massMatrixMatvec against the native complex path —
    the same eigenvalues (and scipy's dense truth), B-orthonormal vectors, true residuals, the reference's outer-iteration
    and restart counts."""
    kw = dict(kw, iseed=(2, 3, 5, 7))
    a, A, B, truth = _generalized_hermitian("reference", kw)
    b, _, _, _ = _generalized_hermitian("hostcheck", kw)
    assert a.ret == b.ret == 0 and b.initSize == kw["numEvals"]
    aN = b.params["aNorm"]
    assert np.max(np.abs(np.sort(b.evals) - np.sort(truth))) <= 1e-10 * aN and np.max(np.abs(np.sort(a.evals) - np.sort(b.evals))) <= 1e-10 * aN
    X = b.evecs
    assert np.max(np.abs(X.conj().T @ B @ X - np.eye(X.shape[1]))) <= 1e-9
    res = np.linalg.norm(A @ X - (B @ X) * b.evals, axis=0)
    assert np.max(np.abs(res - b.resNorms)) <= 1e-9 * aN
    assert (a.stats["numOuterIterations"], a.stats["numRestarts"]) == (b.stats["numOuterIterations"], b.stats["numRestarts"])
    assert a.stats["numPreconds"] == b.stats["numPreconds"]


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
def test_complex_wide_basis_against_live_reference(built):
    """maxBasisSize beyond 255 on complex panels (round 6: up to 1 023 like the real ones; the complex Ritz / restart update
    stages its coefficients in slices of 64 basis columns and its outputs in groups of 16): basis 260, restart 60 on a Hermitian
    band matrix — zprimme's outer-iteration, restart and matvec counts exactly."""
    n = 2000
    A, csr = hermitian_band(n, seed=1)
    kw = dict(numEvals=6, eps=1e-10, iseed=(1, 2, 3, 5), maxBasisSize=260, minRestartSize=60, maxBlockSize=1, dtype=np.complex128)
    a = eigsh(Operator(n, csr=csr), backend="reference", **kw)
    b = eigsh(Operator(n, csr=csr), backend="hostcheck", **kw)
    assert a.ret == b.ret == 0 and b.params["maxBasisSize"] == 260 and b.stats["numRestarts"] >= 2
    assert (a.stats["numOuterIterations"], a.stats["numRestarts"], a.stats["numMatvecs"]) == (b.stats["numOuterIterations"], b.stats["numRestarts"], b.stats["numMatvecs"])
    assert np.max(np.abs(a.evals - b.evals)) <= 1e-10 * b.params["aNorm"]
    assert np.max(np.abs(b.evals - np.linalg.eigvalsh(A)[:6])) <= 1e-9 * b.params["aNorm"]


def test_complex_unsupported_and_argument_errors(built):
    import ctypes as C
    lib = checkers.load_hostcheck()
    p = F.PrimmeParams()
    lib.primme_initialize(C.byref(p))
    p.n, p.numEvals = 10, 2
    ev, rn, vec = np.zeros(2), np.zeros(2), np.zeros(40)
    # no matvec
    assert lib.hip_zprimme(ev.ctypes.data_as(C.c_void_p), vec.ctypes.data_as(C.c_void_p), rn.ctypes.data_as(C.c_void_p), C.byref(p)) == -6
    assert lib.hip_zprimme(None, vec.ctypes.data_as(C.c_void_p), rn.ctypes.data_as(C.c_void_p), C.byref(p)) == -30
    # defaults query, as dprimme(NULL, NULL, NULL, primme)
    assert lib.hip_cprimme(None, None, None, C.byref(p)) == 0 and p.maxBasisSize > 0 and p.nLocal == 10


import complex_fixture_cases as ZF


@pytest.mark.parametrize("name", sorted(ZF.FIX))
def test_native_complex_reproduces_zprimme_fixture(built, name):
    """The native complex path (complex Hermitian projected problem, csrc/eigs_*_z.c) on the CPU checker against the
    committed outputs of the real reference's zprimme / cprimme: eigenvalues, residual norms and — in double
    precision exactly — the outer-iteration, matvec, restart and preconditioner counts."""
    ZF.check(name, "hostcheck")


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
@pytest.mark.parametrize("kw", [dict(numEvals=4, method="GD_plusK", maxBlockSize=1), dict(numEvals=4, method="GD_plusK", maxBlockSize=4),
                                dict(numEvals=6, method="GD_Olsen_plusK", target="largest"), dict(numEvals=5, method="LOBPCG_OrthoBasis", maxBlockSize=5, eps=1e-9),
                                dict(numEvals=4, method="GD_plusK", maxBlockSize=2, locking=0)])
def test_native_complex_follows_live_reference_and_halves_the_real_form(built, kw):
    """Live comparison on a random Hermitian band matrix: the native path takes exactly zprimme's iterations; the
    real-equivalent form (every eigenvalue doubled) needs about twice the operator applications for the same pairs."""
    n = 600
    A, csr = hermitian_band(n, seed=1)
    kw = dict(dict(eps=1e-10, iseed=(1, 2, 3, 5)), **kw)
    nat = eigsh(Operator(n, csr=csr), backend="hostcheck", dtype=np.complex128, **kw)
    ref = eigsh(Operator(n, csr=csr), backend="reference", dtype=np.complex128, **kw)
    rea = eigsh(Operator(n, csr=csr), backend="hostcheck", dtype=np.complex128, complex_form="real", **kw)
    assert nat.ret == 0 and ref.ret == 0 and rea.ret == 0
    for k in ("numOuterIterations", "numMatvecs", "numRestarts"):
        assert nat.stats[k] == ref.stats[k], k
    assert np.max(np.abs(nat.evals - ref.evals)) <= 1e-10 * nat.params["aNorm"]
    assert np.max(np.abs(nat.evals - rea.evals)) <= 1e-9 * nat.params["aNorm"]
    assert rea.stats["numMatvecs"] >= 1.6 * nat.stats["numMatvecs"]


@pytest.mark.skipif(not os.path.exists(checkers.REFERENCE_LIB), reason="oracle/_ref not built")
@pytest.mark.parametrize("kw", [dict(numEvals=6, eps=1e-10), dict(numEvals=3, eps=1e-9, target="largest", locking=0),
                                dict(numEvals=8, eps=1e-9, precond="jacobi", maxBlockSize=2)])
def test_native_complex_dynamic_method(built, kw):
    """method = DYNAMIC on complex panels (round 3: eigs_dynamic.c holds timings and ratios only and serves both
    instantiations).  The switch between GD+k and JDQMR follows measured times, so the counts are not reproducible; the
    pairs are zprimme's, true residuals within the tolerance, and the method it ends on is one of the two."""
    n = 600
    rp, ci, va = problems.hermitian_graded_csr(n)
    v0 = problems.complex_start_vector(n)
    got = eigsh(Operator(n, csr=(rp, ci, va)), backend="hostcheck", dtype=np.complex128, v0=v0, method="DYNAMIC", **kw)
    ref = eigsh(Operator(n, csr=(rp, ci, va)), backend="reference", dtype=np.complex128, v0=v0, method="DYNAMIC", **kw)
    assert got.ret == 0 and ref.ret == 0 and got.initSize == ref.initSize == kw["numEvals"]
    aN = got.params["aNorm"]
    assert np.max(np.abs(np.sort(got.evals) - np.sort(ref.evals))) <= 1e-9 * aN
    AX = problems.csr_matvec_numpy(rp, ci, va, got.evecs)
    assert np.all(np.linalg.norm(AX - got.evecs * got.evals, axis=0) <= 1.5 * kw["eps"] * aN)
    assert got.params["dynamicMethodSwitch"] in (-1, -2, -3)
    assert got.stats["numMatvecs"] <= 4 * ref.stats["numMatvecs"]      # (the switch follows measured times: the two runs may end on different methods)


@pytest.mark.parametrize("name,method", [("jdqmr", "JDQMR"), ("jdqmr_etol", "JDQMR_ETol"), ("gd_olsen", "GD_Olsen_plusK")])
def test_non_hermitian_preconditioner_counts_equal_the_reference(name, method):
    """A NON-Hermitian complex diagonal preconditioner K = diag(A)(1 + 0.1 i w_j): x'K^-1 x of the skew projector is a
    complex number (reference src/eigs/correction.c:969-977, inner_solve.c:737-741).  The product's host solver over
    the plain-C kernels against the real reference's zprimme (tests/golden/reference_zprecond.json, made by
    tests/golden/make_zprecond_golden.py): identical outer-iteration / matvec / restart / preconditioner counts."""
    import ctypes as C
    import json
    import os
    from primme_amd import _ffi as F
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_zprecond.json")))
    N, fx = g["n"], g["cases"][name]
    rp, ci, va, d, a = problems.hermitian_tridiag_graded(N)
    rot = problems.zjacobi_rotation(N, g["gamma"])

    def pc(x, ldx, y, ldy, bs, pp, ierr):
        nb, lx, ly = bs[0], ldx[0], ldy[0]
        X = np.ctypeslib.as_array(C.cast(x, C.POINTER(C.c_double)), shape=(nb, lx * 2)).view(np.complex128)
        Y = np.ctypeslib.as_array(C.cast(y, C.POINTER(C.c_double)), shape=(nb, ly * 2)).view(np.complex128)
        for c in range(nb):
            Y[c, :N] = X[c, :N] / (d * rot)
        ierr[0] = 0
    r = eigsh(Operator(N, csr=(rp, ci, va)), backend="hostcheck", dtype=np.complex128, numEvals=4, eps=1e-10, aNorm=2001.0,
              maxMatvecs=20000, v0=problems.rational_complex_start_vector(N), method=method, user_precond=F.BLOCK_OP(pc))
    assert r.ret == fx["ret"] == 0 and r.initSize == 4
    assert np.max(np.abs(np.asarray(r.evals) - np.array(fx["evals"]))) <= 1e-10 * 2001.0
    for k in ("numOuterIterations", "numMatvecs", "numRestarts", "numPreconds"):
        assert r.stats[k] == fx["stats"][k], (k, r.stats[k], fx["stats"][k])
