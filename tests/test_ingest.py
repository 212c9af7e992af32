"""Matrix ingest: the C Matrix-Market reader, the block-diagonal tiler and the
transpose of include/primme_amd_io.h against numpy on the reference's own data files (LUNDA.mtx,
rect.mtx) and on small files of every Matrix-Market variant the reader accepts."""
import ctypes as C
import os

import numpy as np
import pytest

from primme_amd import _ffi as F

import checkers
from primme_amd import problems
import reference_driver_cases as RD


def _mm_read(lib, path):
    m, n, nnz, cplx = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
    rp, ci, va = C.c_void_p(), C.c_void_p(), C.c_void_p()
    rc = lib.primme_amd_mm_read(path.encode(), C.byref(m), C.byref(n), C.byref(nnz), C.byref(rp), C.byref(ci),
                                C.byref(va), C.byref(cplx))
    if rc:
        return rc, None
    rowptr = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_int32)), shape=(m.value + 1,)).copy()
    colind = np.ctypeslib.as_array(C.cast(ci, C.POINTER(C.c_int32)), shape=(max(nnz.value, 1),))[:nnz.value].copy()
    w = 2 if cplx.value else 1
    vals = np.ctypeslib.as_array(C.cast(va, C.POINTER(C.c_double)), shape=(max(nnz.value, 1) * w,))[:nnz.value * w].copy()
    if cplx.value:
        vals = vals[0::2] + 1j * vals[1::2]
    for p in (rp, ci, va):
        lib.primme_amd_host_free(p)
    return 0, (m.value, n.value, rowptr, colind, vals)


def _dense(m, n, rp, ci, va):
    A = np.zeros((m, n), dtype=va.dtype)
    A[np.repeat(np.arange(m), np.diff(rp)), ci] = va
    return A


@pytest.mark.parametrize("name", ["LUNDA.mtx", "rect.mtx"])
def test_mm_read_reference_files(built, name):
    lib = checkers.load_hostcheck()
    path = os.path.join(RD.DATA, name)
    rc, (m, n, rp, ci, va) = _mm_read(lib, path)
    assert rc == 0
    prp, pci, pva, pm, pn = problems.read_matrix_market(path)
    assert (m, n) == (pm, pn) and np.array_equal(rp, prp) and np.array_equal(ci, pci) and np.array_equal(va, pva)
    for i in range(m):                         # sorted rows, like readfullMTX (csr.c:186-214)
        assert np.all(np.diff(ci[rp[i]:rp[i + 1]]) > 0)
    if name == "LUNDA.mtx":
        A = _dense(m, n, rp, ci, va)
        assert np.array_equal(A, A.T) and len(va) == 2449
        w = np.linalg.eigvalsh(A)
        assert w[0] == pytest.approx(80.03510932165608, rel=1e-9) and w[-1] == pytest.approx(2.238540643913540e8, rel=1e-12)


def test_mm_read_variants(built, tmp_path):
    lib = checkers.load_hostcheck()
    cases = {
        "general": ("%%MatrixMarket matrix coordinate real general\n% c\n3 4 4\n1 1 1.5\n3 4 -2\n2 2 3\n1 3 4e0\n",
                    np.array([[1.5, 0, 4, 0], [0, 3, 0, 0], [0, 0, 0, -2]])),
        "symmetric": ("%%MatrixMarket matrix coordinate real symmetric\n3 3 3\n1 1 2\n3 1 5\n2 2 7\n",
                      np.array([[2, 0, 5], [0, 7, 0], [5, 0, 0.0]])),
        "skew": ("%%MatrixMarket matrix coordinate real skew-symmetric\n3 3 2\n2 1 3\n3 2 -4\n",
                 np.array([[0, -3, 0], [3, 0, 4], [0, -4, 0.0]])),
        "pattern": ("%%MatrixMarket matrix coordinate pattern symmetric\n2 2 2\n1 1\n2 1\n", np.array([[1, 1], [1, 0.0]])),
        "integer": ("%%MatrixMarket matrix coordinate integer general\n2 3 2\n1 3 7\n2 1 -1\n", np.array([[0, 0, 7], [-1, 0, 0.0]])),
        "hermitian": ("%%MatrixMarket matrix coordinate complex hermitian\n2 2 2\n1 1 2 0\n2 1 1 3\n",
                      np.array([[2, 1 - 3j], [1 + 3j, 0]])),
    }
    for name, (text, want) in cases.items():
        p = tmp_path / f"{name}.mtx"
        p.write_text(text)
        rc, (m, n, rp, ci, va) = _mm_read(lib, str(p))
        assert rc == 0, name
        assert np.array_equal(_dense(m, n, rp, ci, va), want), name
    (tmp_path / "bad.mtx").write_text("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n")
    assert _mm_read(lib, str(tmp_path / "bad.mtx"))[0] == -3
    (tmp_path / "trunc.mtx").write_text("%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1\n")
    assert _mm_read(lib, str(tmp_path / "trunc.mtx"))[0] == -2
    assert _mm_read(lib, str(tmp_path / "missing.mtx"))[0] == -1


def test_tiler_and_transpose(built):
    lib = checkers.load_hostcheck()
    rp, ci, va, n0 = RD.lunda()
    T, first = 5, 3
    orp, oci, ova = C.c_void_p(), C.c_void_p(), C.c_void_p()
    va64 = np.ascontiguousarray(va, dtype=np.float64)
    assert lib.primme_amd_csr_tile_block_diagonal(n0, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                                                  va64.ctypes.data_as(C.c_void_p), T, first, 1.0, 0.125,
                                                  C.byref(orp), C.byref(oci), C.byref(ova)) == 0
    nnz0 = len(va)
    trp = np.ctypeslib.as_array(C.cast(orp, C.POINTER(C.c_int32)), shape=(n0 * T + 1,)).copy()
    tci = np.ctypeslib.as_array(C.cast(oci, C.POINTER(C.c_int32)), shape=(nnz0 * T,)).copy()
    tva = np.ctypeslib.as_array(C.cast(ova, C.POINTER(C.c_double)), shape=(nnz0 * T,)).copy()
    for p in (orp, oci, ova):
        lib.primme_amd_host_free(p)
    prp, pci, pva = problems.tile_block_diagonal(rp, ci, va, T, lambda t: 1.0 + 0.125 * t, row0_tile=first)
    assert np.array_equal(trp, prp) and np.array_equal(tci, pci) and np.array_equal(tva, pva)

    # transpose of a rectangular matrix
    rrp, rci, rva, m, n = RD.rect()
    a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
    rva = np.ascontiguousarray(rva, dtype=np.float64)
    assert lib.primme_amd_csr_transpose(m, n, rrp.ctypes.data_as(C.c_void_p), rci.ctypes.data_as(C.c_void_p),
                                        rva.ctypes.data_as(C.c_void_p), 8, C.byref(a), C.byref(b), C.byref(c)) == 0
    rpT = np.ctypeslib.as_array(C.cast(a, C.POINTER(C.c_int32)), shape=(n + 1,)).copy()
    ciT = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_int32)), shape=(len(rva),)).copy()
    vaT = np.ctypeslib.as_array(C.cast(c, C.POINTER(C.c_double)), shape=(len(rva),)).copy()
    for p in (a, b, c):
        lib.primme_amd_host_free(p)
    assert np.array_equal(_dense(n, m, rpT, ciT, vaT), _dense(m, n, rrp, rci, rva).T)
    for j in range(n):
        assert np.all(np.diff(ciT[rpT[j]:rpT[j + 1]]) > 0)
