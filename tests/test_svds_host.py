"""Singular value front end on the CPU: parameter block ABI against the live
reference, the product host logic (hostcheck backend) against the reference's dprimme_svds on the
same inputs, and the reference driver's own svds regression cases with its acceptance test."""
import ctypes as C
import os

import numpy as np
import pytest

from primme_amd import _ffi as F

import checkers
from primme_amd import problems
from checkers import svds, transpose_csr
import reference_driver_cases as RD

HAVE_REF = os.path.exists(checkers.REFERENCE_LIB)


def _rect(m, n, seed=0):
    rng = np.random.default_rng(seed)
    d = min(m, n)
    rows = np.concatenate([np.repeat(np.arange(m), 3), np.arange(d)])
    cols = np.concatenate([rng.integers(0, n, size=3 * m), np.arange(d)])
    vals = np.concatenate([rng.standard_normal(3 * m), 5 + np.arange(d) * 10.0 / d])
    A = np.zeros((m, n))
    np.add.at(A, (rows, cols), vals)
    r, c = np.nonzero(A)
    rp = np.zeros(m + 1, dtype=np.int64)
    np.add.at(rp, r + 1, 1)
    return A, (np.cumsum(rp).astype(np.int32), c.astype(np.int32), A[r, c])


def test_svds_params_abi():
    assert C.sizeof(F.PrimmeSvdsParams) == 1720 and C.sizeof(F.PrimmeSvdsStats) == 128
    assert F.PrimmeSvdsParams.primmeStage2.offset == 640 and F.PrimmeSvdsParams.stats.offset == 1528


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built")
def test_svds_defaults_byte_identical_to_reference(built):
    lib, ref = checkers.load_hostcheck(), checkers.load_reference()
    a, b = F.PrimmeSvdsParams(), F.PrimmeSvdsParams()
    for (m, n, k, meth, tgt, stage1) in [(1000, 500, 5, 2, 0, 0), (500, 1000, 3, 2, 1, F.METHODS["GD_plusK"]),
                                         (800, 800, 10, 2, 0, F.METHODS["JDQMR"]), (900, 700, 4, 1, 0, 0),
                                         (900, 700, 4, 3, 1, 0), (300, 700, 2, 0, 2, 0)]:
        for x, l in ((a, lib), (b, ref)):
            l.primme_svds_initialize(C.byref(x))
            x.outputFile = None
            x.m, x.n, x.numSvals, x.target = m, n, k, tgt
            l.primme_svds_set_method(meth, stage1, 0, C.byref(x))
        assert bytes(a) == bytes(b), (m, n, k, meth, tgt)


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built")
@pytest.mark.parametrize("m,n,k,target", [(300, 200, 5, "largest"), (200, 300, 4, "largest"), (300, 200, 3, "smallest")])
def test_svds_hostcheck_follows_reference(built, m, n, k, target):
    A, csr = _rect(m, n)
    s = np.linalg.svd(A, compute_uv=False)
    want = s[:k] if target == "largest" else s[::-1][:k]
    out = {}
    for be in ("hostcheck", "reference"):
        r = svds(m, n, csr, numSvals=k, target=target, eps=1e-10, methodStage1="GD_plusK", backend=be)
        assert r.ret == 0 and r.initSize == k
        assert np.max(np.abs(r.svals - want)) <= 1e-10 * s[0]
        assert np.all(r.resNorms <= 1e-10 * r.params["aNorm"] * (1 + 1e-6))
        assert np.linalg.norm(A @ r.V - r.U * r.svals) <= 1e-8 * s[0]
        assert np.linalg.norm(r.U.T @ r.U - np.eye(k)) <= 1e-8 and np.linalg.norm(r.V.T @ r.V - np.eye(k)) <= 1e-8
        out[be] = r
    h, r = out["hostcheck"], out["reference"]
    for key in ("numOuterIterations", "numMatvecs", "numRestarts"):
        assert h.stats[key] == r.stats[key], key
    assert h.params["aNorm"] == pytest.approx(r.params["aNorm"], rel=1e-12)


@pytest.mark.parametrize("name", sorted(RD.SVDS_CASES))
@pytest.mark.parametrize("backend", ["hostcheck"] + (["reference"] if HAVE_REF else []))
def test_svds_reference_driver_case(built, name, backend):
    """tests/tests/test_20N on rect.mtx, accepted by the driver's check_solution_svds against the
    reference's stored singular vectors (the `reference` leg pins the checker itself)."""
    case = RD.SVDS_CASES[name]
    rp, ci, va, m, n = RD.svds_matrix(case.get("matrix", "rect.mtx"))
    rpT, ciT, vaT = transpose_csr(m, n, rp, ci, va)
    r = svds(m, n, (rp, ci, va), backend=backend, **{"methodStage1": "GD_plusK", **case["kw"]})
    assert r.ret == 0 and r.initSize == case["kw"]["numSvals"]
    XU, _ = RD.read_sol_svds(case["sol"], m, n)
    bad = RD.check_solution_svds(lambda v: problems.csr_matvec_numpy(rp, ci, va, v.reshape(-1, 1)).ravel(),
                                 lambda u: problems.csr_matvec_numpy(rpT, ciT, vaT, u.reshape(-1, 1)).ravel(),
                                 r.svals, r.U, r.V, r.resNorms, r.params["aNorm"], case["kw"]["eps"], XU)
    assert not bad, bad
    A = np.zeros((m, n))
    A[np.repeat(np.arange(m), np.diff(rp)), ci] = va
    s = np.linalg.svd(A, compute_uv=False)
    k = len(r.svals)
    want = s[:k] if case["kw"]["target"] == "largest" else s[::-1][:k]
    assert np.max(np.abs(r.svals - want)) <= max(case["kw"]["eps"], 1e-10) * s[0]


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built")
@pytest.mark.parametrize("m,n,k,method,eps", [(300, 200, 5, "hybrid", 1e-10), (200, 300, 4, "hybrid", 1e-12),
                                              (300, 200, 4, "augmented", 1e-9), (200, 300, 3, "augmented", 1e-9)])
def test_svds_two_stage_and_augmented_follow_reference(built, m, n, k, method, eps):
    """Hybrid (normal equations, then [0 A'; A 0]) and the augmented operator alone for the largest
    triplets; the hybrid runs reproduce the reference's counts exactly."""
    A, csr = _rect(m, n)
    s = np.linalg.svd(A, compute_uv=False)
    out = {}
    for be in ("hostcheck", "reference"):
        r = svds(m, n, csr, numSvals=k, target="largest", eps=eps, method=method, methodStage1="GD_plusK", backend=be)
        assert r.ret == 0 and r.initSize == k
        assert np.max(np.abs(r.svals - s[:k])) <= max(eps, 1e-10) * s[0]
        assert np.all(r.resNorms <= eps * r.params["aNorm"] * 2.0)
        assert np.linalg.norm(A @ r.V - r.U * r.svals) <= 10 * eps * s[0] * np.sqrt(k)
        assert np.linalg.norm(r.U.T @ r.U - np.eye(k)) <= 1e-7 and np.linalg.norm(r.V.T @ r.V - np.eye(k)) <= 1e-7
        out[be] = r
    h, r = out["hostcheck"], out["reference"]
    if method == "hybrid":
        for key in ("numOuterIterations", "numMatvecs", "numRestarts"):
            assert h.stats[key] == r.stats[key], key
    else:
        assert abs(h.stats["numOuterIterations"] - r.stats["numOuterIterations"]) <= 0.1 * r.stats["numOuterIterations"]


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref not built")
@pytest.mark.parametrize("m,n,k,method,target", [(300, 200, 3, "hybrid", "smallest"), (200, 300, 3, "default", "smallest"),
                                                 (300, 200, 3, "normalequations", "closest_abs"), (300, 200, 3, "hybrid", "closest_abs")])
def test_svds_refined_stages_follow_reference(built, m, n, k, method, target):
    """Smallest and interior triplets: the stages that use the refined extraction (normal equations for
    closest_abs; the augmented second stage for smallest / closest_abs)."""
    A, csr = _rect(m, n)
    s = np.linalg.svd(A, compute_uv=False)
    kw = dict(targetShifts=[8.0]) if target == "closest_abs" else {}
    want = s[::-1][:k] if target == "smallest" else s[np.argsort(np.abs(s - 8.0))][:k]
    out = {}
    for be in ("hostcheck", "reference"):
        r = svds(m, n, csr, numSvals=k, target=target, eps=1e-10, method=method, methodStage1="GD_plusK", backend=be,
                 maxMatvecs=60000, **kw)
        assert r.ret == 0 and r.initSize == k
        assert np.max(np.abs(np.sort(r.svals) - np.sort(want))) <= 1e-9 * s[0]
        out[be] = r
    h, r = out["hostcheck"], out["reference"]
    assert abs(h.stats["numOuterIterations"] - r.stats["numOuterIterations"]) <= 0.05 * r.stats["numOuterIterations"] + 2


def test_svds_interior_blocks_and_input_checks(built):
    A, csr = _rect(60, 40)
    s = np.linalg.svd(A, compute_uv=False)
    # interior targets with blocks: explicit_I + refined extraction in the eigensolver
    r = svds(60, 40, csr, numSvals=2, target="closest_abs", targetShifts=[7.0], maxBlockSize=2, eps=1e-9, backend="hostcheck")
    assert r.ret == 0 and r.initSize == 2
    want = s[np.argsort(np.abs(s - 7.0))][:2]
    assert np.max(np.abs(np.sort(r.svals) - np.sort(want))) <= 1e-8 * s[0]
    assert svds(60, 40, csr, numSvals=70, backend="hostcheck").ret == -10


def _rect_complex(m, n, seed=0):
    A, _ = _rect(m, n, seed)
    rng = np.random.default_rng(seed + 100)
    Z = A.astype(np.complex128)
    r, c = np.nonzero(A)
    Z[r, c] *= np.exp(1j * rng.uniform(0, 2 * np.pi, size=len(r)))          # same pattern, complex phases
    rp = np.zeros(m + 1, dtype=np.int64)
    np.add.at(rp, r + 1, 1)
    return Z, (np.cumsum(rp).astype(np.int32), c.astype(np.int32), Z[r, c])


def test_complex_svds_single_precision(built):
    Z, csr = _rect_complex(150, 90)
    s = np.linalg.svd(Z, compute_uv=False)
    r = svds(150, 90, csr, numSvals=3, eps=1e-4, method="normalequations", backend="hostcheck", dtype=np.complex64)
    assert r.ret == 0 and r.initSize == 3 and r.V.dtype == np.complex64
    assert np.max(np.abs(np.sort(r.svals)[::-1] - s[:3])) <= 1e-3 * s[0]
    assert np.linalg.norm(Z @ r.V - r.U * r.svals) <= 1e-2 * s[0]


@pytest.mark.parametrize("form", ["native", "real_equivalent"])
@pytest.mark.parametrize("m,n,k,target,method", [(120, 80, 4, "largest", "normalequations"), (80, 120, 3, "largest", "hybrid"),
                                                 (120, 80, 3, "smallest", "hybrid"), (300, 200, 5, "largest", "augmented")])
def test_complex_svds_native_and_through_the_real_equivalent_form(built, m, n, k, target, method, form, monkeypatch):
    """hip_zprimme_svds: complex singular triplets; against numpy's dense SVD, and the values / residual level against the
    reference's zprimme_svds on the same inputs.  native (default since round 5): csrc/svds_main.c on complex panels over the
    native complex eigensolver, numSvals triplets — its operator-application count is the reference's to within what the
    default method's wall-clock decisions move it (the real front end shows the same spread); real_equivalent
    (PRIMME_AMD_COMPLEX_REAL_FORM=1, csrc/svds_complex.c): every singular value twice, 1.5-2.7 times the applications."""
    if form == "real_equivalent":
        monkeypatch.setenv("PRIMME_AMD_COMPLEX_REAL_FORM", "1")
    Z, csr = _rect_complex(m, n)
    s = np.linalg.svd(Z, compute_uv=False)
    want = s[:k] if target == "largest" else s[::-1][:k]
    backends = ["hostcheck"] + (["reference"] if HAVE_REF else [])
    mv = {}
    for be in backends:
        r = svds(m, n, csr, numSvals=k, target=target, eps=1e-10, method=method, backend=be, dtype=np.complex128)
        mv[be] = r.stats["numMatvecs"]
        assert r.ret == 0 and r.initSize == k, (be, r.ret, r.initSize)
        assert np.max(np.abs(np.sort(r.svals) - np.sort(want))) <= 1e-9 * s[0], be
        assert np.all(r.resNorms <= 1e-10 * r.params["aNorm"] * 3), be
        assert r.U.dtype == np.complex128 and r.V.dtype == np.complex128
        assert np.linalg.norm(Z @ r.V - r.U * r.svals) <= 1e-8 * s[0], be
        assert np.linalg.norm(Z.conj().T @ r.U - r.V * r.svals) <= 1e-8 * s[0], be
        assert np.linalg.norm(r.V.conj().T @ r.V - np.eye(k)) <= 1e-8 and np.linalg.norm(r.U.conj().T @ r.U - np.eye(k)) <= 1e-7, be
    if form == "native" and "reference" in mv:
        assert 0.7 * mv["reference"] <= mv["hostcheck"] <= 1.4 * mv["reference"], mv


@pytest.mark.parametrize("m,n,k,method", [(300, 200, 5, "normalequations"), (200, 300, 4, "hybrid")])
def test_host_pointer_svds_entry_points(built, m, n, k, method):
    """dprimme_svds() with the reference's HOST-pointer contract (csrc/svds_hostapi.c; BASELINE configs[4] is worded
    with this entry point): host svecs, host callbacks that receive the CALLER's primme_svds_params.  The same
    program, callbacks included, runs against the live reference's dprimme_svds: same counts, same singular values."""
    A, (rp, ci, va) = _rect(m, n, seed=3)
    s = np.linalg.svd(A, compute_uv=False)
    libs = [("hostcheck", checkers.load_hostcheck())] + ([("reference", checkers.load_reference())] if HAVE_REF else [])
    out = {}
    for name, lib in libs:
        ps = F.PrimmeSvdsParams()
        lib.primme_svds_initialize(C.byref(ps))
        seen = {"mv": 0, "user_struct": True}

        def mv(x, ldx, y, ldy, bs, tr, pp, ierr, ps=ps, seen=seen):
            seen["mv"] += bs[0]
            seen["user_struct"] &= (C.addressof(pp[0]) == C.addressof(ps))
            rin, rout = (m, n) if tr[0] else (n, m)
            X = np.ctypeslib.as_array(C.cast(x, C.POINTER(C.c_double)), shape=(bs[0], ldx[0]))
            Y = np.ctypeslib.as_array(C.cast(y, C.POINTER(C.c_double)), shape=(bs[0], ldy[0]))
            Y[:, :rout] = (A.T @ X[:, :rin].T).T if tr[0] else (A @ X[:, :rin].T).T
            ierr[0] = 0
        cmv = F.SVDS_BLOCK_OP(mv)
        ps.m, ps.n, ps.numSvals, ps.eps, ps.printLevel, ps.outputFile = m, n, k, 1e-10, 0, None
        ps.target = F.SVDS_TARGETS["largest"]
        ps.matrixMatvec = C.cast(cmv, C.c_void_p)
        lib.primme_svds_set_method(F.SVDS_METHODS[method], F.METHODS["GD_plusK"], F.METHODS["GD_plusK"], C.byref(ps))
        svecs = np.zeros((m + n) * k); svals, rn = np.zeros(k), np.zeros(k)
        f = lib.dprimme_svds
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(F.PrimmeSvdsParams)]; f.restype = C.c_int
        ret = f(svals.ctypes.data, svecs.ctypes.data, rn.ctypes.data, C.byref(ps))
        assert ret == 0 and ps.initSize == k and seen["user_struct"], (name, ret)
        U = svecs[:m * k].reshape(k, m).T; V = svecs[m * k:].reshape(k, n).T
        assert np.max(np.abs(svals - s[:k])) <= 1e-10 * s[0]
        assert np.linalg.norm(A @ V - U * svals) <= 1e-8 * s[0]
        assert np.linalg.norm(U.T @ U - np.eye(k)) <= 1e-8 and np.linalg.norm(V.T @ V - np.eye(k)) <= 1e-8
        # stats.numMatvecs is the algorithm's count (the reference's accounting); the device path also applies the
        # operator speculatively ahead of the host's decision: a few of those are discarded
        assert ps.stats.numMatvecs <= seen["mv"] <= ps.stats.numMatvecs * 1.05 + 2
        if name == "reference":
            assert seen["mv"] == ps.stats.numMatvecs
        assert C.cast(ps.matrixMatvec, C.c_void_p).value == C.cast(cmv, C.c_void_p).value and not ps.queue   # caller's struct intact
        out[name] = (svals.copy(), ps.stats.numOuterIterations, ps.stats.numMatvecs, ps.stats.numRestarts, seen["mv"])
    if "reference" in out:
        assert out["hostcheck"][1:4] == out["reference"][1:4], out
        assert np.max(np.abs(out["hostcheck"][0] - out["reference"][0])) <= 1e-12 * s[0]
