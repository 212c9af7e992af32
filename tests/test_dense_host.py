"""Host-side dense helpers of the projected problem (primme_amd/csrc/eigs_dense.c) against
numpy, and the xLARNV stream against the LAPACK fixture (tests/golden/lapack_dlarnv.json)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from primme_amd import _ffi as F

import checkers


@pytest.fixture(scope="module")
def lib(built):
    l = C.CDLL(checkers.HOSTCHECK_LIB)
    return l


@pytest.mark.parametrize("n", [1, 2, 3, 7, 15, 24, 41, 64])
def test_sym_eig(lib, n):
    rng = np.random.default_rng(n)
    for trial in range(4):
        A = rng.standard_normal((n, n)); A = A + A.T
        if trial == 1:   # clustered / repeated eigenvalues
            Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
            A = Q @ np.diag(np.repeat(rng.standard_normal((n + 2) // 3), 3)[:n]) @ Q.T
        if trial == 2:   # diagonal plus small dense block, the shape after a restart
            A = np.diag(np.sort(rng.standard_normal(n)))
            if n > 3:
                B = rng.standard_normal((2, 2)); A[n - 3:n - 1, n - 3:n - 1] = B + B.T
        if trial == 3:   # widely scaled
            A = A * 1e8
        lda = n + 3
        Af = np.full((lda, n), np.nan, order="F")
        Af[:n, :] = np.triu(A)                 # only the upper triangle is referenced
        Af[:n, :][np.tril_indices(n, -1)] = 777.0
        w = np.zeros(n); Z = np.zeros((n, n), order="F")
        rc = lib.pa_sym_eig(n, Af.ctypes.data_as(C.c_void_p), lda, w.ctypes.data_as(C.c_void_p), Z.ctypes.data_as(C.c_void_p), n)
        assert rc == 0
        wr = np.linalg.eigvalsh(A)
        scale = max(1.0, np.abs(wr).max())
        assert np.max(np.abs(w - wr)) <= 5e-14 * scale * n
        assert np.max(np.abs(Z.T @ Z - np.eye(n))) <= 1e-13 * n
        assert np.max(np.abs(A @ Z - Z * w)) <= 2e-13 * scale * n


@pytest.mark.parametrize("n", [2, 9, 33])
def test_sym_eig_generalized(lib, n):
    rng = np.random.default_rng(100 + n)
    A = rng.standard_normal((n, n)); A = A + A.T
    M = rng.standard_normal((n, n)) * 0.05; G = np.eye(n) + M + M.T
    w = np.zeros(n); Z = np.zeros((n, n), order="F")
    Au = np.asfortranarray(np.triu(A)); Gu = np.asfortranarray(np.triu(G))
    rc = lib.pa_sym_eig_gen(n, Au.ctypes.data_as(C.c_void_p), n, Gu.ctypes.data_as(C.c_void_p), n,
                            w.ctypes.data_as(C.c_void_p), Z.ctypes.data_as(C.c_void_p), n)
    assert rc == 0
    import scipy.linalg as sl
    wr = sl.eigh(A, G, eigvals_only=True)
    assert np.max(np.abs(w - wr)) <= 1e-12 * max(1, np.abs(wr).max())
    assert np.max(np.abs(Z.T @ G @ Z - np.eye(n))) <= 1e-12
    assert np.max(np.abs(A @ Z - G @ Z * w)) <= 1e-11


def test_larnv_stream_matches_lapack(lib):
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "lapack_dlarnv.json")))
    for key, g in gold.items():
        seed = (C.c_int64 * 4)(*[int(t) for t in key.split(",")])
        x = np.zeros(300)
        lib.pa_larnv_uniform11(seed, C.c_int64(300), x.ctypes.data_as(C.c_void_p))
        assert np.array_equal(x, np.array(g["values"]))          # bit-exact
        assert list(seed) == g["seed_after"]
    # the stream is continuous across calls
    seed = (C.c_int64 * 4)(0, 0, 0, 1)
    a = np.zeros(100); b = np.zeros(200)
    lib.pa_larnv_uniform11(seed, C.c_int64(100), a.ctypes.data_as(C.c_void_p))
    lib.pa_larnv_uniform11(seed, C.c_int64(200), b.ctypes.data_as(C.c_void_p))
    assert np.array_equal(np.concatenate([a, b]), np.array(gold["0,0,0,1"]["values"]))
