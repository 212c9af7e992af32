"""Matrix-Market file -> CSR and block-diagonal tiling through the library's C ingest (include/primme_amd_io.h:
primme_amd_mm_read, primme_amd_csr_tile_block_diagonal — the reference's tests/COMMON/mmio.c + csr.c:46-265 path),
returned as numpy arrays: ctypes plumbing only.  BASELINE configs[2] is fed through this in bench.py and in the tests."""
import ctypes as C

import numpy as np


def mm_read(lib, path):
    m, n, nnz, cplx = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
    rp, ci, va = C.c_void_p(), C.c_void_p(), C.c_void_p()
    lib.primme_amd_mm_read.restype = C.c_int
    rc = lib.primme_amd_mm_read(path.encode(), C.byref(m), C.byref(n), C.byref(nnz), C.byref(rp), C.byref(ci),
                                C.byref(va), C.byref(cplx))
    if rc:
        raise RuntimeError(f"primme_amd_mm_read({path}) = {rc}")
    rowptr = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_int32)), shape=(m.value + 1,)).copy()
    colind = np.ctypeslib.as_array(C.cast(ci, C.POINTER(C.c_int32)), shape=(max(nnz.value, 1),))[:nnz.value].copy()
    w = 2 if cplx.value else 1
    vals = np.ctypeslib.as_array(C.cast(va, C.POINTER(C.c_double)), shape=(max(nnz.value, 1) * w,))[:nnz.value * w].copy()
    if cplx.value:
        vals = vals[0::2] + 1j * vals[1::2]
    for p in (rp, ci, va):
        lib.primme_amd_host_free(p)
    return rowptr, colind, vals, m.value, n.value


def tile_block_diagonal(lib, rowptr, colind, values, ntiles, scale0=1.0, scale_step=0.0, first_tile=0):
    n0, nnz0 = len(rowptr) - 1, len(values)
    rp, ci, va = C.c_void_p(), C.c_void_p(), C.c_void_p()
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colind = np.ascontiguousarray(colind, dtype=np.int32)
    values = np.ascontiguousarray(values, dtype=np.float64)
    lib.primme_amd_csr_tile_block_diagonal.restype = C.c_int
    lib.primme_amd_csr_tile_block_diagonal.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                                       C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = lib.primme_amd_csr_tile_block_diagonal(n0, rowptr.ctypes.data, colind.ctypes.data, values.ctypes.data, ntiles, first_tile,
                                                scale0, scale_step, C.byref(rp), C.byref(ci), C.byref(va))
    if rc:
        raise RuntimeError(f"primme_amd_csr_tile_block_diagonal = {rc}")
    n, nnz = n0 * ntiles, nnz0 * ntiles
    out = (np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_int32)), shape=(n + 1,)).copy(),
           np.ctypeslib.as_array(C.cast(ci, C.POINTER(C.c_int32)), shape=(nnz,)).copy(),
           np.ctypeslib.as_array(C.cast(va, C.POINTER(C.c_double)), shape=(nnz,)).copy())
    for p in (rp, ci, va):
        lib.primme_amd_host_free(p)
    return out
