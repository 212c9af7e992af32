"""Helpers for the kernel-level parity tests: run one entry point of the C ABI on the
HIP library (device memory through the library's own C ABI) and on the plain-C oracle (host memory)."""
import ctypes as C
import numpy as np

from primme_amd import _ffi as F

import checkers

NPDT = {F.HIPK_F64: np.float64, F.HIPK_F32: np.float32, F.HIPK_C64: np.complex128, F.HIPK_C32: np.complex64}


class DevArray:
    """A device buffer owned by a Dev: what the tests pass around instead of a tensor."""

    def __init__(self, ptr, shape, dtype):
        self.ptr, self.shape, self.dtype = ptr, tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize

    def data_ptr(self):
        return self.ptr

    def element_size(self):
        return self.dtype.itemsize


class Dev:
    """Device side: the product library, and ONLY the product library — buffers come from hipk_malloc, uploads and
    read-backs are hipk_h2d / hipk_d2h through a pinned staging buffer, all on the context's own stream, so that every
    byte a kernel reads or a test compares travelled on the one in-order queue the kernels run on (round 2 went through
    torch on the NULL stream, which that stream is not ordered against; the driver's one red run could not be pinned on
    it, nor reproduced in 32 400 products — profiles/r03_halo_campaign.log — but nothing needs it either)."""
    name = "hip"

    def __init__(self):
        self.lib = F.load_product()
        self.ctx = C.c_void_p()
        assert self.lib.hipk_ctx_create(C.byref(self.ctx), None) == 0
        self.keep = []
        self.stage, self.stage_bytes = C.c_void_p(), 0

    def _staging(self, nbytes):
        if nbytes > self.stage_bytes:
            if self.stage:
                self.lib.hipk_host_free(self.ctx, self.stage)
            self.stage = C.c_void_p()
            self.stage_bytes = max(nbytes, 1 << 20)
            assert self.lib.hipk_host_alloc(self.ctx, self.stage_bytes, C.byref(self.stage)) == 0
        return self.stage

    def arr(self, a):
        a = np.ascontiguousarray(a)
        d = C.c_void_p()
        assert self.lib.hipk_malloc(self.ctx, max(a.nbytes, 8), C.byref(d)) == 0
        t = DevArray(d.value, a.shape, a.dtype)
        if a.nbytes:
            st = self._staging(a.nbytes)
            C.memmove(st, a.ctypes.data, a.nbytes)
            assert self.lib.hipk_h2d(self.ctx, d, st, a.nbytes) == 0
            assert self.lib.hipk_sync(self.ctx) == 0
        self.keep.append(t)
        return t

    def ptr(self, t, offset_elems=0):
        return C.c_void_p(t.data_ptr() + offset_elems * t.element_size())

    def get(self, t):
        out = np.empty(t.shape, t.dtype)
        if t.nbytes:
            st = self._staging(t.nbytes)
            C.memset(st, 0xFF, t.nbytes)           # a short copy shows up as NaN
            assert self.lib.hipk_d2h(self.ctx, st, C.c_void_p(t.ptr), t.nbytes) == 0
            assert self.lib.hipk_sync(self.ctx) == 0
            C.memmove(out.ctypes.data, st, t.nbytes)
        else:
            assert self.lib.hipk_sync(self.ctx) == 0
        return out

    def close(self):
        self.lib.hipk_sync(self.ctx)
        for t in self.keep:
            self.lib.hipk_free(self.ctx, C.c_void_p(t.ptr))
        self.keep = []
        if self.stage:
            self.lib.hipk_host_free(self.ctx, self.stage)
        self.lib.hipk_ctx_destroy(self.ctx)


class Host:
    """Oracle side: numpy arrays, oracle/hipk_cpu.c."""
    name = "oracle"

    def __init__(self):
        self.lib = checkers.load_hostcheck()
        self.ctx = C.c_void_p()
        assert self.lib.hipk_ctx_create(C.byref(self.ctx), None) == 0
        self.keep = []

    def arr(self, a):
        t = np.array(a, copy=True, order="C")
        self.keep.append(t)
        return t

    def ptr(self, t, offset_elems=0):
        return C.c_void_p(t.ctypes.data + offset_elems * t.itemsize)

    def get(self, t):
        return t.copy()

    def close(self):
        self.lib.hipk_ctx_destroy(self.ctx)


def segs_array(side, segs):
    """segs: list of (tensor, col0, ld, ncols)"""
    arr = (F.HipkSeg * max(len(segs), 1))()
    for i, (t, col0, ld, nc) in enumerate(segs):
        arr[i].base = side.ptr(t, col0 * ld).value
        arr[i].ld = ld
        arr[i].ncols = nc
    return arr


if __import__("os").environ.get("PRIMME_AMD_HARNESS_HOST_ONLY"):
    # heap-corruption hunt (round 4): both sides of every kernel test on the plain-C oracle, so that the oracle's code runs
    # under AddressSanitizer on a box without a GPU (the product-only entry points then fail: expected)
    class Dev(Host):          # noqa: F811
        name = "hip"
