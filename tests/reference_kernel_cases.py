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
