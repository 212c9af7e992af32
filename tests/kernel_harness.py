"""Helpers for the kernel-level parity tests: run one entry point of the C ABI on the
HIP library (device memory through torch) and on the plain-C oracle (host memory)."""
import ctypes as C
import numpy as np

from primme_amd import _ffi as F

import checkers

NPDT = {F.HIPK_F64: np.float64, F.HIPK_F32: np.float32}


class Dev:
    """Device side: torch tensors on cuda:0, product library."""
    name = "hip"

    def __init__(self):
        import torch
        self.torch = torch
        self.lib = F.load_product()
        self.ctx = C.c_void_p()
        assert self.lib.hipk_ctx_create(C.byref(self.ctx), None) == 0
        self.keep = []

    def arr(self, a):
        t = self.torch.from_numpy(np.ascontiguousarray(a)).to("cuda")
        self.torch.cuda.synchronize()
        self.keep.append(t)
        return t

    def ptr(self, t, offset_elems=0):
        return C.c_void_p(t.data_ptr() + offset_elems * t.element_size())

    def get(self, t):
        self.lib.hipk_sync(self.ctx)
        self.torch.cuda.synchronize()
        return t.cpu().numpy()

    def close(self):
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
