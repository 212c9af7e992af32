"""Kernel-level parity against the REFERENCE's own panel routines (fixture F6: tests/golden/reference_kernels.json,
made by tests/golden/make_kernel_golden.py from update_projection_dprimme, Num_update_VWXR_dprimme,
Bortho_gen_dprimme and Bortho_block_dprimme).  `check_all(side)` drives one device-layer implementation —
the plain-C oracle (kernel_harness.Host) or the HIP kernels (kernel_harness.Dev) — through the same steps
with the same inputs and compares with what the reference produced."""
import ctypes as C
import json
import os

import numpy as np

from primme_amd import _ffi as F
from kernel_harness import segs_array

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kernels.json")))
DT = F.HIPK_F64


def mat(e, ld=None):
    """column-major fixture entry -> array of shape (cols, ld): row j = column j, as the panels are laid out"""
    a = np.array(e["data"], dtype=np.float64).reshape(e["cols"], e["rows"])
    if ld is None or ld == e["rows"]:
        return a
    out = np.zeros((e["cols"], ld))
    out[:, :e["rows"]] = a
    return out


def close(a, b, tol, what):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    err = np.max(np.abs(a - b)) if a.size else 0.0
    assert err <= tol * max(1.0, np.max(np.abs(b)) if b.size else 1.0), (what, err)


def check_update_projection(side):
    g = GOLD["update_projection"]
    m, ld, k, b = GOLD["m"], GOLD["ld"], GOLD["k"], GOLD["b"]
    V, W = mat(g["V"], ld), mat(g["W"], ld)
    v, w = side.arr(V), side.arr(W)
    out = side.arr(np.zeros((b, k + b)))
    segs = segs_array(side, [(v, 0, ld, k + b)])
    assert side.lib.hipk_panel_dots(side.ctx, DT, m, segs, 1, side.ptr(w, k * ld), ld, b, side.ptr(out), k + b) == 0
    got, ref = side.get(out), mat(g["H_new_columns"])
    for c in range(b):                       # the reference keeps the upper part: rows 0 .. k + c of column k + c
        close(got[c, :k + c + 1], ref[c, :k + c + 1], 1e-13, ("update_projection", c))
    close(got, W[k:k + b, :m] @ V[:k + b, :m].T, 1e-13, "update_projection vs numpy")


def check_update_vwxr(side):
    g = GOLD["update_VWXR"]
    m, ld, k, b = GOLD["m"], GOLD["ld"], GOLD["k"], GOLD["b"]
    V, W, h, theta = mat(g["V"], ld), mat(g["W"], ld), mat(g["h"]), mat(g["theta"])[0]
    nh = h.shape[0]
    v, w, hh, th = side.arr(V), side.arr(W), side.arr(h), side.arr(theta)
    x0 = side.arr(np.zeros((b, ld))); r = side.arr(np.zeros((b, ld))); x1 = side.arr(np.zeros((nh - b, ld))); wo = side.arr(np.zeros((nh - b, ld)))
    n2 = side.arr(np.zeros(b))
    jobs = (F.HipkJob * (2 * b + 2 * (nh - b)))()
    q = 0
    for c in range(b):
        jobs[q].kind, jobs[q].col, jobs[q].dst, jobs[q].slot = F.HIPK_JOB_XV, c, side.ptr(x0, c * ld).value, -1; q += 1
        jobs[q].kind, jobs[q].col, jobs[q].dst, jobs[q].slot = F.HIPK_JOB_RES, c, side.ptr(r, c * ld).value, c; q += 1
    for c in range(b, nh):
        jobs[q].kind, jobs[q].col, jobs[q].dst, jobs[q].slot = F.HIPK_JOB_XV, c, side.ptr(x1, (c - b) * ld).value, -1; q += 1
        jobs[q].kind, jobs[q].col, jobs[q].dst, jobs[q].slot = F.HIPK_JOB_XW, c, side.ptr(wo, (c - b) * ld).value, -1; q += 1
    assert side.lib.hipk_ritz_update(side.ctx, DT, m, side.ptr(v), side.ptr(w), ld, k, side.ptr(hh), k, side.ptr(th), jobs, q, side.ptr(n2)) == 0
    close(side.get(x0)[:, :m], mat(g["X0"]), 1e-13, "X0"); close(side.get(r)[:, :m], mat(g["R"]), 1e-13, "R")
    close(side.get(x1)[:, :m], mat(g["X1"]), 1e-13, "X1"); close(side.get(wo)[:, :m], mat(g["Wo"]), 1e-13, "Wo")
    close(np.sqrt(side.get(n2)), mat(g["Rnorms"])[0], 1e-13, "Rnorms")
    close(np.linalg.norm(side.get(x0)[:, :m], axis=1), mat(g["xnorms"])[0], 1e-13, "xnorms")


def _dots(side, segs, nseg, x_t, x_off, ld, nx, m, ntot):
    out = side.arr(np.zeros((nx, ntot)))
    assert side.lib.hipk_panel_dots(side.ctx, DT, m, segs, nseg, side.ptr(x_t, x_off), ld, nx, side.ptr(out), ntot) == 0
    return side.get(out)


def check_bortho_gen(side):
    """One new vector against [V | locked]: the dots -> update -> norm chain with Daniel's test
    (ortho.c:229-309), built from the device layer exactly as eigs_ops.c:pa_ortho_cgs chains it."""
    g = GOLD["Bortho_gen"]
    m, ld, k, L = GOLD["m"], GOLD["ld"], GOLD["k"], GOLD["L"]
    Q, V, xin = mat(g["locked"], ld), mat(g["V_orthonormal"], ld), mat(g["new_column_in"], ld)
    q, v, x = side.arr(Q), side.arr(V), side.arr(xin)
    rlocked = np.zeros(L)
    s0 = None
    for npass in range(3):
        first = npass == 0
        segs = segs_array(side, [(v, 0, ld, k), (q, 0, ld, L), (x, 0, ld, 1 if first else 0)])
        ov = _dots(side, segs, 3, x, 0, ld, 1, m, k + L + 1)[0]
        if first:
            s0 = np.sqrt(ov[k + L])
        rlocked += ov[k:k + L] if npass == 0 else 0.0 * ov[k:k + L]
        cf = side.arr(ov[:k + L].copy()); n2 = side.arr(np.zeros(1))
        segs2 = segs_array(side, [(v, 0, ld, k), (q, 0, ld, L)])
        assert side.lib.hipk_panel_project(side.ctx, DT, m, segs2, 2, side.ptr(cf), k + L, side.ptr(x), ld, 1, side.ptr(n2)) == 0
        s1 = np.sqrt(side.get(n2)[0])
        if s1 > np.sqrt(2.0) / 2.0 * s0:
            break
        s0 = s1
    a = (C.c_double * 1)(1.0 / s1)
    assert side.lib.hipk_scale_cols(side.ctx, DT, m, side.ptr(x), ld, 1, a) == 0
    close(side.get(x)[0, :m], mat(g["new_column_out"])[0], 1e-13, "Bortho_gen vector")
    close(rlocked, mat(g["RLocked"])[0], 1e-13, "RLocked")


def check_bortho_block(side):
    """A block of b columns against [locked | V] and itself: projection + Cholesky QR sweeps made of the
    device layer's TN panel (the matrix-core kernel on the GPU) and the fused update * right-multiply
    (ortho.c:497-803, :963-1072).  The orthonormal block with R upper triangular, positive diagonal is unique."""
    g = GOLD["Bortho_block"]
    m, ld, k, L, b = GOLD["m"], GOLD["ld"], GOLD["k"], GOLD["L"], GOLD["b"]
    Q, V, X = mat(g["locked"], ld), mat(g["V_orthonormal"], ld), mat(g["block_in"], ld)
    q, v, x = side.arr(Q), side.arr(V), side.arr(X)
    segs = segs_array(side, [(q, 0, ld, L), (v, 0, ld, k)])
    segsx = segs_array(side, [(q, 0, ld, L), (v, 0, ld, k), (x, 0, ld, b)])
    for sweep in range(3):
        G = _dots(side, segsx, 3, x, 0, ld, b, m, L + k + b)           # rows: right-hand columns
        A = G[:, :L + k].T                                             # [Q V]' X
        Cm = G[:, L + k:].T - A.T @ A                                  # X'X - X'[Q V][Q V]'X
        Rc = np.linalg.cholesky((Cm + Cm.T) / 2).T                     # upper
        cf = side.arr(np.ascontiguousarray(A.T)); mm = side.arr(np.ascontiguousarray(np.linalg.inv(Rc).T))
        assert side.lib.hipk_panel_project_mul(side.ctx, DT, m, segs, 2, side.ptr(cf), L + k, side.ptr(mm), side.ptr(x), ld, b) == 0
    got = side.get(x)[:, :m]
    close(got, mat(g["block_out"]), 5e-13, "Bortho_block block")
    # the tracked Gram columns the reference leaves behind: [locked V X]' X of the OUTPUT block
    Gout = _dots(side, segsx, 3, x, 0, ld, b, m, L + k + b)
    close(Gout, mat(g["gram_new_columns"]), 5e-13, "tracked Gram columns")


def check_all(side):
    check_update_projection(side)
    check_update_vwxr(side)
    check_bortho_gen(side)
    check_bortho_block(side)


# ---- round 5: the same four routines at LARGE shapes (GOLD["wide"]) -----------------------------------------------------
# Many workgroups, ragged tails, second stages over thousands of partial sums, two-tile matrix-core panels, the wide-basis
# Ritz kernels.  Inputs are closed forms restated here from oracle/ref_kernel_harness.c (hash-uniform numbers: bit-exact;
# sine-basis columns: orthonormal analytically, equal to an ulp); the reference's m-sized outputs are stored as three sums per
# column (plain, hash-weighted, squares) and a strided sample of 64 elements.
def _hu(i, j, salt):
    i = np.asarray(i, dtype=np.uint64); j = np.asarray(j, dtype=np.uint64)
    v = (i * np.uint64(2654435761) + j * np.uint64(2246822519) + np.uint64(salt) * np.uint64(3266489917)) & np.uint64(0xFFFFFFFF)
    return (v >> np.uint64(8)).astype(np.float64) / 16777216.0 - 0.5


def _hpanel(m, n, salt):
    return _hu(np.arange(m)[None, :], np.arange(n)[:, None], salt)          # (cols, rows): row j = column j


def _spanel(m, j0, n):
    return np.sqrt(2.0 / (m + 1)) * np.sin(np.pi * (np.arange(m)[None, :] + 1.0) * (np.arange(j0, j0 + n)[:, None] + 1.0) / (m + 1.0))


def _digest_close(got, e, tol, what):
    """got: (cols, m) array; e: the fixture's digest of the reference's output"""
    m, step = e["rows"], e["step"]
    got = np.asarray(got, dtype=np.float64)[:, :m]
    w = _hu(np.arange(m)[None, :], np.arange(got.shape[0])[:, None], 77)
    sums = np.array(e["sums"])
    scale = np.sqrt(np.maximum(sums[:, 2], 1e-300))                         # |column|
    sq = np.sum(got * got, axis=1)
    assert np.all(np.abs(np.sum(got, axis=1) - sums[:, 0]) <= tol * np.sqrt(m) * scale), (what, "sum")
    assert np.all(np.abs(np.sum(got * w, axis=1) - sums[:, 1]) <= tol * np.sqrt(m) * scale), (what, "weighted sum")
    assert np.all(np.abs(sq - sums[:, 2]) <= tol * np.maximum(sums[:, 2], 1e-300) * 10), (what, "squares")
    samp = np.array(e["sample"])
    err = np.max(np.abs(got[:, ::step] - samp))
    assert err <= tol * max(1.0, np.max(np.abs(samp))), (what, "sample", err)


def check_wide(side, idx):
    g = GOLD["wide"][idx]
    m, ld, k, b, L, nh = g["m"], g["ld"], g["k"], g["b"], g["L"], g["nh"]
    tol = 2e-13
    # update_projection: H(0:k+b, k:k+b) = V' W(:, k:k+b)
    V, W = _hpanel(m, k + b, 1), _hpanel(m, k + b, 2)
    v, w = side.arr(V), side.arr(W)
    out = side.arr(np.zeros((b, k + b)))
    segs = segs_array(side, [(v, 0, ld, k + b)])
    assert side.lib.hipk_panel_dots(side.ctx, DT, m, segs, 1, side.ptr(w, k * ld), ld, b, side.ptr(out), k + b) == 0
    got, ref = side.get(out), mat(g["update_projection"]["H_new_columns"])
    for c in range(b):
        close(got[c, :k + c + 1], ref[c, :k + c + 1], tol * np.sqrt(m), ("wide update_projection", idx, c))
    # Num_update_VWXR: X0 = V h(:,0:b), R = W h(:,0:b) - X0 diag(theta), |R|, X1 = V h(:,b:nh), Wo = W h(:,b:nh)
    u = g["update_VWXR"]
    h = _hu(np.arange(k)[None, :], np.arange(nh)[:, None], 3) / np.sqrt(float(k))           # (nh, k): row j = column j
    theta = 0.3 + 0.11 * np.arange(nh)
    hh, th = side.arr(h), side.arr(theta)
    x0 = side.arr(np.zeros((b, ld))); r = side.arr(np.zeros((b, ld))); x1 = side.arr(np.zeros((nh - b, ld))); wo = side.arr(np.zeros((nh - b, ld)))
    n2 = side.arr(np.zeros(b))
    jobs = (F.HipkJob * (2 * b + 2 * (nh - b)))()
    q = 0
    for c in range(b):
        jobs[q].kind, jobs[q].col, jobs[q].dst, jobs[q].slot = F.HIPK_JOB_XV, c, side.ptr(x0, c * ld).value, -1; q += 1
        jobs[q].kind, jobs[q].col, jobs[q].dst, jobs[q].slot = F.HIPK_JOB_RES, c, side.ptr(r, c * ld).value, c; q += 1
    for c in range(b, nh):
        jobs[q].kind, jobs[q].col, jobs[q].dst, jobs[q].slot = F.HIPK_JOB_XV, c, side.ptr(x1, (c - b) * ld).value, -1; q += 1
        jobs[q].kind, jobs[q].col, jobs[q].dst, jobs[q].slot = F.HIPK_JOB_XW, c, side.ptr(wo, (c - b) * ld).value, -1; q += 1
    assert side.lib.hipk_ritz_update(side.ctx, DT, m, side.ptr(v), side.ptr(w), ld, k, side.ptr(hh), k, side.ptr(th), jobs, q, side.ptr(n2)) == 0
    _digest_close(side.get(x0), u["X0"], tol, ("wide X0", idx)); _digest_close(side.get(r), u["R"], tol, ("wide R", idx))
    _digest_close(side.get(x1), u["X1"], tol, ("wide X1", idx)); _digest_close(side.get(wo), u["Wo"], tol, ("wide Wo", idx))
    close(np.sqrt(side.get(n2)), mat(u["Rnorms"])[0], tol * 10, ("wide Rnorms", idx))
    close(np.linalg.norm(side.get(x0)[:, :m], axis=1), mat(u["xnorms"])[0], tol * 10, ("wide xnorms", idx))
    # Bortho_gen: one new column against [V | locked] (sine basis), the dots -> update -> norm chain with Daniel's test
    Vs, Qs = _spanel(m, 0, k), _spanel(m, k, max(L, 1))
    xin = (_hu(np.arange(m), 0, 4) + 0.5 * Vs[0] + (0.25 * Qs[0] if L > 0 else 0.0))[None, :]
    qd, vd, x = side.arr(Qs), side.arr(Vs), side.arr(xin)
    rlocked = np.zeros(L)
    s0 = s1 = None
    for npass in range(3):
        first = npass == 0
        segs = segs_array(side, [(vd, 0, ld, k), (qd, 0, ld, L), (x, 0, ld, 1 if first else 0)])
        ov = _dots(side, segs, 3, x, 0, ld, 1, m, k + L + 1)[0]
        if first:
            s0 = np.sqrt(ov[k + L])
            rlocked += ov[k:k + L]
        cf = side.arr(ov[:k + L].copy()); nn = side.arr(np.zeros(1))
        segs2 = segs_array(side, [(vd, 0, ld, k), (qd, 0, ld, L)])
        assert side.lib.hipk_panel_project(side.ctx, DT, m, segs2, 2, side.ptr(cf), k + L, side.ptr(x), ld, 1, side.ptr(nn)) == 0
        s1 = np.sqrt(side.get(nn)[0])
        if s1 > np.sqrt(2.0) / 2.0 * s0:
            break
        s0 = s1
    a = (C.c_double * 1)(1.0 / s1)
    assert side.lib.hipk_scale_cols(side.ctx, DT, m, side.ptr(x), ld, 1, a) == 0
    _digest_close(side.get(x), g["Bortho_gen"]["new_column_out"], 5 * tol, ("wide Bortho_gen", idx))
    if L > 0:
        close(rlocked, mat(g["Bortho_gen"]["RLocked"])[0], 5 * tol, ("wide RLocked", idx))
    # Bortho_block: b columns against [locked | V] and themselves, CholQR sweeps from the TN panel + the fused update * right-multiply
    X = np.stack([_hu(np.arange(m), c, 5) + 0.5 * Vs[c % k] for c in range(b)])
    xb = side.arr(X)
    segs = segs_array(side, [(qd, 0, ld, L), (vd, 0, ld, k)])
    segsx = segs_array(side, [(qd, 0, ld, L), (vd, 0, ld, k), (xb, 0, ld, b)])
    for sweep in range(3):
        G = _dots(side, segsx, 3, xb, 0, ld, b, m, L + k + b)
        A = G[:, :L + k].T
        Cm = G[:, L + k:].T - A.T @ A
        Rc = np.linalg.cholesky((Cm + Cm.T) / 2).T
        cf = side.arr(np.ascontiguousarray(A.T)); mm = side.arr(np.ascontiguousarray(np.linalg.inv(Rc).T))
        assert side.lib.hipk_panel_project_mul(side.ctx, DT, m, segs, 2, side.ptr(cf), L + k, side.ptr(mm), side.ptr(xb), ld, b) == 0
    _digest_close(side.get(xb), g["Bortho_block"]["block_out"], 10 * tol, ("wide Bortho_block", idx))
    Gout = _dots(side, segsx, 3, xb, 0, ld, b, m, L + k + b)
    close(Gout, mat(g["Bortho_block"]["gram_new_columns"]), 10 * tol, ("wide tracked Gram columns", idx))
