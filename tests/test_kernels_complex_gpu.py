"""Kernel-level parity of the COMPLEX instantiation of the device layer (csrc/hipk_complex.hip, HIPK_C64 / HIPK_C32):
every entry point hip_zprimme / hip_cprimme's native path launches, through the C ABI on the MI355X, against the plain-C
complex restatement (oracle/hipk_cpu_complex.c) on the same seeded inputs and against numpy.  Conventions: panel
elements, inner products, coefficients and axpy factors are (re, im) pairs; Ritz values, shifts, squared norms and
scale factors are real."""
import ctypes as C

import numpy as np
import pytest

from primme_amd import _ffi as F
from primme_amd import problems
from kernel_harness import Dev, Host, segs_array, NPDT

pytestmark = pytest.mark.gpu
ZDT = [F.HIPK_C64, F.HIPK_C32]


def _z(rng, shape, dt):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(NPDT[dt])


def _tol(dt):
    return 1e-12 if dt == F.HIPK_C64 else 2e-5


@pytest.mark.parametrize("dt", ZDT)
@pytest.mark.parametrize("m,k,L", [(1, 1, 0), (1000, 3, 2), (70001, 12, 5), (250003, 41, 9), (4099, 140, 3)])
@pytest.mark.parametrize("nx", [1, 2, 4, 5, 8, 11])
def test_complex_panel_dots(built, dt, m, k, L, nx):
    """out = [V Q x]^H X: the conjugate-transposed TN panel (Num_gemm_ddh "C","N" with SCALAR = complex)."""
    rng = np.random.default_rng(m + k + nx)
    ld, ldq = m + 3, m + 1
    V, Q, X = _z(rng, (k, ld), dt), _z(rng, (max(L, 1), ldq), dt), _z(rng, (nx, ld), dt)
    tot = k + L + 1
    for ldout in (tot, tot + 2):
        outs = []
        for side in (Dev(), Host()):
            v, q, x = side.arr(V), side.arr(Q), side.arr(X)
            out = side.arr(np.zeros((nx, ldout), dtype=np.complex128))
            segs = segs_array(side, [(v, 0, ld, k), (q, 0, ldq, L), (x, 0, ld, 1)])
            assert side.lib.hipk_panel_dots(side.ctx, dt, m, segs, 3, side.ptr(x), ld, nx, side.ptr(out), ldout) == 0
            outs.append(side.get(out)[:, :tot])
            side.close()
        A = np.concatenate([V[:, :m], Q[:L, :m], X[:1, :m]]).astype(np.complex128)
        ref = (A.conj() @ X[:, :m].astype(np.complex128).T).T
        scale = np.sqrt(m) * 4
        assert np.max(np.abs(outs[0] - outs[1])) <= _tol(dt) * scale * max(1.0, np.abs(ref).max() / scale)
        assert np.max(np.abs(outs[1] - ref)) <= _tol(dt) * scale * max(1.0, np.abs(ref).max() / scale)


@pytest.mark.parametrize("dt", ZDT)
@pytest.mark.parametrize("m,k,L,nx", [(1000, 3, 2, 1), (70001, 12, 5, 2), (250003, 41, 9, 4), (50001, 20, 0, 8), (3001, 300, 0, 3)])
def test_complex_panel_project(built, dt, m, k, L, nx):
    """X -= [V Q] c with complex coefficients, |X|^2; in place and out of place."""
    rng = np.random.default_rng(m + k + nx + 1)
    ld = m + 2
    V, Q, X = _z(rng, (k, ld), dt), _z(rng, (max(L, 1), ld), dt), _z(rng, (nx, ld), dt)
    coef = (rng.standard_normal((nx, k + L + 1)) + 1j * rng.standard_normal((nx, k + L + 1))) * 0.1
    for out_of_place in (False, True):
        res = []
        for side in (Dev(), Host()):
            v, q, x, c = side.arr(V), side.arr(Q), side.arr(X), side.arr(coef)
            xo = side.arr(np.full((nx, ld), np.nan + 0j, NPDT[dt])) if out_of_place else x
            n2 = side.arr(np.zeros(nx))
            segs = segs_array(side, [(v, 0, ld, k), (q, 0, ld, L)])
            if out_of_place:
                rc = side.lib.hipk_panel_project_to(side.ctx, dt, m, segs, 2, side.ptr(c), k + L + 1, side.ptr(x), ld, side.ptr(xo), ld, nx, side.ptr(n2))
            else:
                rc = side.lib.hipk_panel_project(side.ctx, dt, m, segs, 2, side.ptr(c), k + L + 1, side.ptr(x), ld, nx, side.ptr(n2))
            assert rc == 0
            res.append((side.get(xo)[:, :m], side.get(n2), side.get(x)))
            side.close()
        A = np.concatenate([V[:, :m], Q[:L, :m]]).astype(np.complex128)
        want = X[:, :m].astype(np.complex128) - coef[:, :k + L] @ A
        tol = _tol(dt) * 50 * (1 + np.abs(want).max())
        assert np.max(np.abs(res[0][0] - res[1][0])) <= tol and np.max(np.abs(res[1][0] - want)) <= tol
        assert np.allclose(res[0][1], res[1][1], rtol=_tol(dt) * 1e3)
        assert np.allclose(res[1][1], np.sum(np.abs(res[1][0].astype(np.complex128)) ** 2, axis=1), rtol=_tol(dt) * 1e3)
        if out_of_place:
            assert np.array_equal(res[0][2], X)          # the source is untouched


@pytest.mark.parametrize("dt", ZDT)
@pytest.mark.parametrize("m,k,L,nx", [(70003, 9, 4, 8), (70002, 24, 0, 4), (1000, 3, 2, 2), (129, 0, 0, 3), (50001, 40, 20, 5)])
def test_complex_panel_project_mul(built, dt, m, k, L, nx):
    """X <- (X - [Q V] A) M in one pass: the device step of a complex CholQR / SVQB sweep."""
    rng = np.random.default_rng(m + k + nx + 2)
    ld = m + 1
    V, Q, X = _z(rng, (max(k, 1), ld), dt), _z(rng, (max(L, 1), ld), dt), _z(rng, (nx, ld), dt)
    coef = (rng.standard_normal((nx, k + L + 1)) + 1j * rng.standard_normal((nx, k + L + 1))) * 0.1
    M = rng.standard_normal((nx, nx)) + 1j * rng.standard_normal((nx, nx))        # M[c, q] = M(q, c): column-major nx x nx
    res = []
    for side in (Dev(), Host()):
        v, q, x, c, mm = side.arr(V), side.arr(Q), side.arr(X), side.arr(coef), side.arr(M)
        segs = segs_array(side, [(q, 0, ld, L), (v, 0, ld, k)])
        assert side.lib.hipk_panel_project_mul(side.ctx, dt, m, segs, 2, side.ptr(c), k + L + 1, side.ptr(mm), side.ptr(x), ld, nx) == 0
        res.append(side.get(x))
        side.close()
    A = np.concatenate([Q[:L, :m], V[:k, :m]]).astype(np.complex128)
    Y = X[:, :m].astype(np.complex128) - coef[:, :k + L] @ A
    want = M @ Y                                # row c of the result = sum_q M(q, c) Y_q
    tol = _tol(dt) * 100 * (1 + np.abs(want).max())
    assert np.max(np.abs(res[0][:, :m] - res[1][:, :m])) <= tol and np.max(np.abs(res[1][:, :m] - want)) <= tol
    assert np.array_equal(res[0][:, m:], X[:, m:])


@pytest.mark.parametrize("dt", ZDT)
@pytest.mark.parametrize("m,k,shape", [(999, 3, None), (50003, 15, None), (200001, 24, None), (40000, 41, None), (30011, 140, None),
                                       (70001, 20, (10, 4, 0)), (70002, 20, (5, 3, 3)), (1000001, 20, (8, 4, 0)),
                                       (70003, 20, (14, 4, 0)), (70004, 30, (20, 4, 2)), (1000002, 20, (14, 4, 0)),
                                       (70005, 40, (24, 4, 0)), (70006, 12, (3, 2, 3)), (70007, 26, (18, 4, 0)), (70008, 40, (22, 4, 0))])
def test_complex_ritz_update_inplace_restart(built, dt, m, k, shape):
    """The restart shape with complex coefficient vectors: V, W <- V h, W h in place, the next block's X and R, copies
    of locked vectors and residual norms.  shape = (restart size, block, locked) -> (V-products, W-products, residuals):
    (10,4,0) -> (14,10,4) and (8,4,0) run the two-lane kernel <7,7,2>, (3,2,3) -> (8,3,5) the two-lane <4,4,8>,
    (5,3,3) -> (11,5,6) the four-lane <8,4,4>, (14,4,0) -> (18,14,4) the two-lane <9,7,2> (the configs[3] restart, also
    at its full-size kernel grid: 1 M rows against numpy), (18,4,0) -> (22,18,4) the two-lane <12,10,2>, (22,4,0) -> (26,22,4) with k = 40 the four-lane <8,7,1>, (24,4,0) -> (28,24,4) likewise, both with the
    coefficient block staged in two tiles; (20,4,2) -> (26,20,6) and the larger default shapes fit none of them and take
    the chunked path."""
    rng = np.random.default_rng(m + k)
    ld, K = m + 1, k + 6
    V, W = _z(rng, (K, ld), dt), _z(rng, (K, ld), dt)
    E = np.zeros((4, ld), dtype=NPDT[dt])
    h = np.linalg.qr(rng.standard_normal((k, k)) + 1j * rng.standard_normal((k, k)))[0]
    hfull = np.zeros((k, K + 2), dtype=np.complex128)
    hfull[:, :k] = h.T
    theta = rng.standard_normal(k)
    rs = max(1, k // 2)
    nb = min(2, k - rs) if k > rs else 0
    nlock = min(3, k - rs)
    if shape is not None:
        rs, nb, nlock = shape
    jobs = []
    for c in range(rs): jobs.append((F.HIPK_JOB_XV, c, ("V", c), -1))
    for c in range(nb): jobs.append((F.HIPK_JOB_XV, c, ("V", rs + nlock + c), -1))
    for c in range(nlock): jobs.append((F.HIPK_JOB_XV, rs + c, ("E", c), -1))
    for c in range(rs): jobs.append((F.HIPK_JOB_XW, c, ("W", c), -1))
    for c in range(nb): jobs.append((F.HIPK_JOB_RES, c, ("W", rs + nlock + c), c))
    for c in range(nlock): jobs.append((F.HIPK_JOB_RES, rs + c, None, nb + c))
    res = []
    sides = (Dev(), Host()) if m < 1000000 else (Dev(),)          # full size: against numpy on sampled rows only
    for side in sides:
        v, w, e, hh, th = side.arr(V), side.arr(W), side.arr(E), side.arr(hfull), side.arr(theta)
        n2 = side.arr(np.zeros(nb + nlock + 1))
        base = {"V": v, "W": w, "E": e}
        arr = (F.HipkJob * len(jobs))()
        for i, (kind, col, dst, slot) in enumerate(jobs):
            arr[i].kind, arr[i].col, arr[i].slot = kind, col, slot
            arr[i].dst = None if dst is None else side.ptr(base[dst[0]], dst[1] * ld).value
        assert side.lib.hipk_ritz_update(side.ctx, dt, m, side.ptr(v), side.ptr(w), ld, k, side.ptr(hh), K + 2,
                                         side.ptr(th), arr, len(jobs), side.ptr(n2)) == 0
        res.append((side.get(v), side.get(w), side.get(e), side.get(n2)))
        side.close()
    if len(res) == 1:
        rows = np.concatenate([np.arange(0, 300), np.arange(m // 2, m // 2 + 300), np.arange(m - 300, m)])
        Vs, Ws = V[:k][:, rows].astype(np.complex128), W[:k][:, rows].astype(np.complex128)
        tol = (1e-12 if dt == F.HIPK_C64 else 1e-4) * 100
        assert np.max(np.abs(res[0][0][:rs][:, rows] - h.T[:rs] @ Vs)) <= tol and np.max(np.abs(res[0][1][:rs][:, rows] - h.T[:rs] @ Ws)) <= tol
        for c in range(nb):
            r = h.T[c] @ W[:k, :m].astype(np.complex128) - theta[c] * (h.T[c] @ V[:k, :m].astype(np.complex128))
            assert np.max(np.abs(res[0][1][rs + nlock + c, rows] - r[rows])) <= tol
            assert np.isclose(res[0][3][c], np.sum(np.abs(r) ** 2), rtol=tol * 10)
        return
    Vd, Wd = V[:k, :m].astype(np.complex128), W[:k, :m].astype(np.complex128)
    want_V0 = h.T[:rs] @ Vd                      # row c = sum_j h(j, c) V_j
    tol = (1e-12 if dt == F.HIPK_C64 else 1e-4) * 10
    for a, b in zip(res[0][:3], res[1][:3]):
        assert np.max(np.abs(a - b)) <= tol
    assert np.max(np.abs(res[1][0][:rs, :m] - want_V0)) <= tol * 10
    assert np.allclose(res[0][3][:nb + nlock], res[1][3][:nb + nlock], rtol=tol * 100)
    for c in range(nlock):
        r = h.T[rs + c] @ Wd - theta[rs + c] * (h.T[rs + c] @ Vd)
        assert np.isclose(res[1][3][nb + c], np.sum(np.abs(r) ** 2), rtol=tol * 100)


@pytest.mark.parametrize("dt", ZDT)
def test_complex_column_utilities(built, dt):
    """axpy / xpay with complex factors, x^H y, real scale, squared norms, w -= theta x, copy, gather."""
    rng = np.random.default_rng(7)
    m, nx, ld = 123457, 3, 123460
    X, Y = _z(rng, (nx, ld), dt), _z(rng, (nx, ld), dt)
    a = rng.standard_normal(nx) + 1j * rng.standard_normal(nx)
    sc = rng.standard_normal(nx)
    th = rng.standard_normal(nx)
    res = []
    for side in (Dev(), Host()):
        out = {}
        x, y = side.arr(X), side.arr(Y)
        af = np.ascontiguousarray(a).view(np.float64)
        assert side.lib.hipk_axpy_cols(side.ctx, dt, m, af.ctypes.data_as(C.POINTER(C.c_double)), side.ptr(x), ld, side.ptr(y), ld, nx) == 0
        out["axpy"] = side.get(y)
        assert side.lib.hipk_xpay_cols(side.ctx, dt, m, af.ctypes.data_as(C.POINTER(C.c_double)), side.ptr(x), ld, side.ptr(y), ld, nx) == 0
        out["xpay"] = side.get(y)
        d = side.arr(np.zeros(nx, dtype=np.complex128))
        assert side.lib.hipk_pair_dots(side.ctx, dt, m, side.ptr(x), ld, side.ptr(y), ld, nx, side.ptr(d)) == 0
        out["dots"] = side.get(d)
        assert side.lib.hipk_scale_cols(side.ctx, dt, m, side.ptr(y), ld, nx, sc.ctypes.data_as(C.POINTER(C.c_double))) == 0
        out["scale"] = side.get(y)
        n2 = side.arr(np.zeros(nx))
        assert side.lib.hipk_col_norms2(side.ctx, dt, m, side.ptr(y), ld, nx, side.ptr(n2)) == 0
        out["n2"] = side.get(n2)
        r2 = side.arr(np.zeros(nx))
        assert side.lib.hipk_residual_cols(side.ctx, dt, m, side.ptr(x), ld, side.ptr(y), ld, nx, th.ctypes.data_as(C.POINTER(C.c_double)), side.ptr(r2)) == 0
        out["res"], out["r2"] = side.get(y), side.get(r2)
        z = side.arr(np.zeros((nx, ld), NPDT[dt]))
        assert side.lib.hipk_copy_cols(side.ctx, dt, m, side.ptr(y), ld, side.ptr(z), ld, nx) == 0
        out["copy"] = side.get(z)
        perm = (C.c_int * nx)(2, 0, 1)
        assert side.lib.hipk_gather_cols(side.ctx, dt, m, side.ptr(y), ld, perm, nx, side.ptr(z), ld) == 0
        out["gather"] = side.get(z)
        res.append(out)
        side.close()
    tol = _tol(dt) * 100
    for key in ("axpy", "xpay", "scale", "res", "copy", "gather"):
        assert np.max(np.abs(res[0][key][:, :m] - res[1][key][:, :m])) <= tol * (1 + np.abs(res[1][key][:, :m]).max()), key
    for key in ("dots", "n2", "r2"):
        assert np.allclose(res[0][key], res[1][key], rtol=tol * 10, atol=tol * np.sqrt(m)), key
    Xd, Yd = X[:, :m].astype(np.complex128), Y[:, :m].astype(np.complex128)
    y1 = a[:, None] * Xd + Yd
    y2 = a[:, None] * y1 + Xd
    assert np.max(np.abs(res[1]["xpay"][:, :m] - y2)) <= tol * (1 + np.abs(y2).max())
    assert np.allclose(res[1]["dots"], np.sum(Xd.conj() * y2, axis=1), rtol=tol * 10, atol=tol * np.sqrt(m))
    assert np.array_equal(res[0]["gather"][0, :m], res[0]["res"][2, :m])


def _hermitian_csr(n, seed, band=7):
    rng = np.random.default_rng(seed)
    import scipy.sparse as sp
    diags = {0: rng.standard_normal(n) * 4}
    A = sp.diags(diags[0], 0, dtype=np.complex128).tolil()
    for d in range(1, band + 1):
        if d in (2, 5): continue
        v = rng.standard_normal(n - d) + 1j * rng.standard_normal(n - d)
        A = A + sp.diags(v, d) + sp.diags(v.conj(), -d)
    A = sp.csr_matrix(A)
    A.sort_indices()
    return A


@pytest.mark.parametrize("dt", ZDT)
@pytest.mark.parametrize("ncols", [1, 2, 3, 8])
def test_complex_csr_matvec(built, dt, ncols):
    """y = A x with complex values and vectors, plain and with per-column real shifts, and the Jacobi preconditioner on
    the (real) diagonal; banded Hermitian (BASELINE configs[3]'s family), a ragged matrix with empty rows and one row
    longer than a tile."""
    import scipy.sparse as sp
    rng = np.random.default_rng(11 + ncols)
    mats = [("band", _hermitian_csr(30011, 3))]
    n = 5000
    counts = rng.integers(0, 9, size=n); counts[17] = 0; counts[100] = 3000; counts[n - 1] = 0
    rp = np.zeros(n + 1, dtype=np.int64); np.cumsum(counts, out=rp[1:])
    ci = np.concatenate([np.sort(rng.choice(n, size=c, replace=False)) for c in counts]).astype(np.int32)
    va = rng.standard_normal(len(ci)) + 1j * rng.standard_normal(len(ci))
    mats.append(("ragged", sp.csr_matrix((va, ci, rp), shape=(n, n))))
    for name, A in mats:
        n = A.shape[0]
        rp, ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
        va = np.ascontiguousarray(A.data, dtype=NPDT[dt])
        ld = n + 2
        X = _z(rng, (ncols, ld), dt)
        shifts = rng.standard_normal(ncols)
        res = []
        for side in (Dev(), Host()):
            H = C.c_void_p()
            assert side.lib.hipk_csr_create(side.ctx, dt, n, n, 0, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                                            va.ctypes.data_as(C.c_void_p), C.byref(H)) == 0
            x = side.arr(X); y = side.arr(np.full((ncols, ld), np.nan + 0j, NPDT[dt])); ys = side.arr(np.zeros((ncols, ld), NPDT[dt]))
            yj = side.arr(np.zeros((ncols, ld), NPDT[dt]))
            assert side.lib.hipk_csr_matvec(H, None, side.ptr(x), ld, side.ptr(y), ld, ncols) == 0
            assert side.lib.hipk_csr_matvec_shifted(H, None, side.ptr(x), ld, side.ptr(ys), ld, ncols, shifts.ctypes.data_as(C.POINTER(C.c_double))) == 0
            side.lib.hipk_csr_diag.restype = C.c_void_p
            assert side.lib.hipk_jacobi_apply(None if side.name == "oracle" else side.lib.hipk_ctx_stream(side.ctx), dt, n, C.c_void_p(side.lib.hipk_csr_diag(H)),
                                              shifts.ctypes.data_as(C.POINTER(C.c_double)), 1e-8, side.ptr(x), ld, side.ptr(yj), ld, ncols) == 0
            res.append((side.get(y)[:, :n], side.get(ys)[:, :n], side.get(yj)[:, :n]))
            side.lib.hipk_csr_destroy(H)
            side.close()
        Ad = sp.csr_matrix((va.astype(np.complex128), ci, rp), shape=(n, n))
        Xd = X[:, :n].astype(np.complex128)
        want = (Ad @ Xd.T).T
        tol = _tol(dt) * 50 * (1 + np.abs(want).max())
        assert np.max(np.abs(res[0][0] - res[1][0])) <= tol and np.max(np.abs(res[1][0] - want)) <= tol, name
        assert np.max(np.abs(res[0][1] - (want - shifts[:, None] * Xd))) <= tol, name
        dg = Ad.diagonal().real
        den = dg[None, :] - shifts[:, None]
        den = np.where(np.abs(den) > 1e-8, den, np.copysign(1e-8, den))
        assert np.max(np.abs(res[0][2] - res[1][2])) <= tol * (1 + np.abs(res[1][2]).max()), name
        assert np.max(np.abs(res[1][2] - Xd / den)) <= tol * (1 + np.abs(Xd / den).max()), name


@pytest.mark.parametrize("dt", ZDT)
def test_complex_csr_row_slab_with_halo(built, dt):
    """A slab of rows of a Hermitian band matrix with GLOBAL column numbers and hand-filled halo buffers (what a
    row-partitioned complex run asks of the SpMM kernel)."""
    rng = np.random.default_rng(5)
    A = _hermitian_csr(9000, 9)
    n, nslabs, nc = A.shape[0], 3, 3
    X = _z(rng, (nc, n), dt)
    want = (A.astype(np.complex128) @ X.astype(np.complex128).T).T
    for sidx in range(nslabs):
        r0, r1 = sidx * n // nslabs, (sidx + 1) * n // nslabs
        S = A[r0:r1]
        rp, ci = S.indptr.astype(np.int32), S.indices.astype(np.int32)
        va = np.ascontiguousarray(S.data, dtype=NPDT[dt])
        nloc = r1 - r0
        outs = []
        for side in (Dev(), Host()):
            H = C.c_void_p()
            assert side.lib.hipk_csr_create(side.ctx, dt, nloc, n, r0, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                                            va.ctypes.data_as(C.c_void_p), C.byref(H)) == 0
            lo, hi = int(side.lib.hipk_csr_halo_lo(H)), int(side.lib.hipk_csr_halo_hi(H))
            xl = side.arr(np.ascontiguousarray(X[:, r0:r1]))
            xlo = side.arr(np.ascontiguousarray(X[:, r0 - lo:r0]) if lo else np.zeros((nc, 1), NPDT[dt]))
            xhi = side.arr(np.ascontiguousarray(X[:, r1:r1 + hi]) if hi else np.zeros((nc, 1), NPDT[dt]))
            y = side.arr(np.full((nc, nloc), np.nan + 0j, NPDT[dt]))
            assert side.lib.hipk_csr_set_halo_ld(H, side.ptr(xlo), max(lo, 1), side.ptr(xhi), max(hi, 1)) == 0
            assert side.lib.hipk_csr_matvec(H, None, side.ptr(xl), nloc, side.ptr(y), nloc, nc) == 0
            outs.append(side.get(y))
            side.lib.hipk_csr_destroy(H)
            side.close()
        tol = _tol(dt) * 50 * (1 + np.abs(want).max())
        assert np.max(np.abs(outs[0] - want[:, r0:r1])) <= tol and np.max(np.abs(outs[1] - want[:, r0:r1])) <= tol
