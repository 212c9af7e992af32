"""Python driver over the singular value C ABI (include/primme_amd_svds.h): plumbing only.

    svds(m, n, (rowptr, colind, values), numSvals=...)

hip_dprimme_svds / hip_sprimme_svds of primme_amd/libprimme_amd.so with the matrix and its
transpose resident in HBM.  `backend` is "hip" or a backend object (see api.HipBackend); the CPU
checkers of the tests are built by oracle/checkers.py, never from here."""
import ctypes as C
import numpy as np

from . import _ffi as F
from .api import _resolve_backend


class SvdsResult:
    def __init__(self, ret, svals, U, V, resNorms, ps):
        self.ret, self.svals, self.U, self.V, self.resNorms = ret, svals, U, V, resNorms
        self.initSize = ps.initSize
        self.stats = {k: getattr(ps.stats, k) for k, _ in F.PrimmeSvdsStats._fields_}
        self.eig_stats = {k: getattr(ps.primme.stats, k) for k, _ in F.PrimmeStats._fields_}
        self.params = dict(aNorm=ps.aNorm, eps=ps.eps, method=ps.method, methodStage2=ps.methodStage2,
                           maxBasisSize=ps.primme.maxBasisSize, maxBlockSize=ps.primme.maxBlockSize,
                           minRestartSize=ps.primme.minRestartSize, locking=ps.primme.locking,
                           eig_n=ps.primme.n, eig_target=ps.primme.target)


def complex_csr_to_real(m, rp, ci, va, rdtype):
    """real-equivalent form of a complex CSR matrix for vectors stored (re0, im0, re1, im1, ...): every entry
    a + ib at (i, j) becomes the block [[a, -b], [b, a]] at rows (2i, 2i+1), columns (2j, 2j+1)"""
    cnt = np.diff(rp).astype(np.int64)
    rp2 = np.zeros(2 * m + 1, dtype=np.int64)
    rp2[1::2] = 2 * cnt
    rp2[2::2] = 2 * cnt
    rp2 = np.cumsum(rp2)
    nnz = len(va)
    ci2 = np.empty(4 * nnz, dtype=np.int32)
    va2 = np.empty(4 * nnz, dtype=rdtype)
    rows = np.repeat(np.arange(m, dtype=np.int64), cnt)
    pos = np.arange(nnz, dtype=np.int64) - np.asarray(rp, dtype=np.int64)[rows]       # position inside the row
    e = rp2[2 * rows] + 2 * pos
    o = rp2[2 * rows + 1] + 2 * pos
    ci2[e], ci2[e + 1], va2[e], va2[e + 1] = 2 * ci, 2 * ci + 1, va.real, -va.imag
    ci2[o], ci2[o + 1], va2[o], va2[o + 1] = 2 * ci, 2 * ci + 1, va.imag, va.real
    return rp2.astype(np.int32), ci2, va2


def transpose_csr(m, n, rp, ci, va):
    order = np.argsort(ci, kind="stable")
    rows = np.repeat(np.arange(m, dtype=np.int64), np.diff(rp))
    rpT = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rpT, ci.astype(np.int64) + 1, 1)
    return np.cumsum(rpT).astype(np.int32), rows[order].astype(np.int32), va[order]


def svds(m, n, csr, numSvals=1, target="largest", method="normalequations", methodStage1="DEFAULT_METHOD",
         eps=1e-8, aNorm=0.0, backend="hip", dtype=np.float64, maxBlockSize=0, maxBasisSize=0, locking=None,
         maxMatvecs=0, v0=None, iseed=None, printLevel=0, return_vectors=True, targetShifts=None, precond=None):
    dtype = np.dtype(dtype)
    cplx = dtype.kind == "c"                       # hip_zprimme_svds / hip_cprimme_svds (csrc/svds_complex.c)
    rdtype = np.dtype(np.float64 if dtype in (np.float64, np.complex128) else np.float32)
    dt = F.HIPK_F64 if rdtype == np.float64 else F.HIPK_F32
    ctype = C.c_double if rdtype == np.float64 else C.c_float
    rp, ci, va = csr
    rp = np.ascontiguousarray(rp, dtype=np.int32)
    ci = np.ascontiguousarray(ci, dtype=np.int32)
    va = np.ascontiguousarray(va, dtype=dtype)
    be = _resolve_backend(backend)
    lib = be.lib
    keep = []
    ps = F.PrimmeSvdsParams()
    lib.primme_svds_initialize(C.byref(ps))
    ps.m, ps.n, ps.numSvals = m, n, numSvals
    ps.target = F.SVDS_TARGETS[target]
    ps.eps, ps.aNorm, ps.printLevel, ps.outputFile = eps, aNorm, printLevel, None
    if maxBlockSize: ps.maxBlockSize = maxBlockSize
    if maxBasisSize: ps.maxBasisSize = maxBasisSize
    if locking is not None: ps.locking = locking
    if maxMatvecs: ps.maxMatvecs = maxMatvecs
    if targetShifts is not None:
        ts = (C.c_double * len(targetShifts))(*targetShifts)
        keep.append(ts)
        ps.targetShifts, ps.numTargetShifts = ts, len(targetShifts)
    if iseed is not None:
        for i in range(4): ps.iseed[i] = iseed[i]
    v0 = None if v0 is None else np.asarray(v0, dtype=dtype).reshape(n, -1)
    ps.initSize = 0 if v0 is None else v0.shape[1]
    ncols = max(numSvals, ps.initSize)
    handles = []

    if not be.native_operator:
        solver = be.setup_svds_operator(ps, keep, m, n, rp, ci, va, ctype, precond, dtype)
    else:
        ctx = C.c_void_p()
        if lib.hipk_ctx_create(C.byref(ctx), None):
            raise RuntimeError("hipk_ctx_create failed: no HIP device (primme_amd has no CPU path)")
        handles.append(("ctx", ctx))
        oph = C.c_void_p()
        if cplx:
            rp2, ci2, va2 = complex_csr_to_real(m, rp, ci, va, rdtype)
            rc = lib.primme_amd_svds_operator_create(C.byref(oph), ctx, dt, 2 * m, 2 * n, rp2.ctypes.data_as(C.c_void_p),
                                                     ci2.ctypes.data_as(C.c_void_p), va2.ctypes.data_as(C.c_void_p))
        else:
            rc = lib.primme_amd_svds_operator_create(C.byref(oph), ctx, dt, m, n, rp.ctypes.data_as(C.c_void_p),
                                                     ci.ctypes.data_as(C.c_void_p), va.ctypes.data_as(C.c_void_p))
        if rc:
            raise RuntimeError(f"svds operator creation failed: {rc}")
        handles.append(("op", oph))
        if cplx:
            lib.primme_amd_svds_operator_set_complex(oph, 1)
            if precond is not None:
                raise ValueError("the library's Jacobi preconditioner is for real matrices")
        ps.matrix = oph
        ps.matrixMatvec = C.cast(lib.primme_amd_svds_matvec, C.c_void_p)
        if precond is not None:
            shift = 0.0 if precond == "jacobi" else float(precond[1])
            if lib.primme_amd_svds_operator_set_jacobi(oph, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                                                       va.ctypes.data_as(C.c_void_p), shift):
                raise RuntimeError("svds Jacobi set-up failed")
            ps.preconditioner = oph
            ps.applyPreconditioner = C.cast(lib.primme_amd_svds_jacobi_precond, C.c_void_p)
        solver = be.svds_solver(dtype.name)

    mset = getattr(F, "PRIMME_" + methodStage1) if isinstance(methodStage1, str) and hasattr(F, "PRIMME_" + methodStage1) \
        else F.METHODS.get(methodStage1, 0) if isinstance(methodStage1, str) else methodStage1
    lib.primme_svds_set_method(F.SVDS_METHODS[method], mset, 0, C.byref(ps))

    svals = np.zeros(numSvals, dtype=rdtype)
    rnorms = np.zeros(numSvals, dtype=rdtype)
    total = (m + n) * ncols
    sv_t = None
    if be.native_operator and be.device:
        import torch
        tdt = {"float64": torch.float64, "float32": torch.float32, "complex128": torch.complex128, "complex64": torch.complex64}[dtype.name]
        sv_t = torch.zeros(total, dtype=tdt, device="cuda")
        if v0 is not None:
            # [U0 (m x initSize) | V0 (n x initSize)]: only V0 is used by A'A, U0 by AA'
            sv_t[m * ps.initSize: m * ps.initSize + n * ps.initSize] = torch.from_numpy(np.ascontiguousarray(v0.T).ravel()).cuda()
        torch.cuda.synchronize()
        svp = C.c_void_p(sv_t.data_ptr())
    else:
        sv = np.zeros(total, dtype=dtype)
        if v0 is not None:
            sv[m * ps.initSize: m * ps.initSize + n * ps.initSize] = np.ascontiguousarray(v0.T).ravel()
        svp = sv.ctypes.data_as(C.c_void_p)
    ret = solver(svals.ctypes.data_as(C.c_void_p), svp, rnorms.ctypes.data_as(C.c_void_p), C.byref(ps))
    k = ps.initSize
    U = V = None
    if return_vectors and k > 0:
        if be.native_operator and be.device:
            import torch
            torch.cuda.synchronize()
            sv = sv_t.cpu().numpy()
        U = sv[:m * k].reshape(k, m).T.copy()
        V = sv[m * k:m * k + n * k].reshape(k, n).T.copy()
    res = SvdsResult(ret, svals[:max(k, 0)].copy(), U, V, rnorms[:max(k, 0)].copy(), ps)
    for kind, h in reversed(handles):
        if kind == "op": lib.primme_amd_svds_operator_destroy(h)
        elif kind == "ctx": lib.hipk_ctx_destroy(h)
    return res


class SvdsSession:
    """The singular value operator (A and A' resident in HBM, built once) for repeated solves of the same problem:
    what bench.py times for BASELINE configs[4].  Device library only; real matrices."""

    def __init__(self, m, n, csr, dtype=np.float64, backend="hip"):
        self.dtype = np.dtype(dtype)
        if self.dtype.kind == "c":
            raise ValueError("SvdsSession: real matrices only (complex data: svds())")
        self.be = _resolve_backend(backend)
        if not (self.be.native_operator and self.be.device):
            raise ValueError("SvdsSession needs the device library")
        self.lib, self.m, self.n = self.be.lib, m, n
        self.dt = F.HIPK_F64 if self.dtype == np.float64 else F.HIPK_F32
        rp, ci, va = csr
        rp = np.ascontiguousarray(rp, dtype=np.int32); ci = np.ascontiguousarray(ci, dtype=np.int32)
        va = np.ascontiguousarray(va, dtype=self.dtype)
        self.ctx = C.c_void_p()
        if self.lib.hipk_ctx_create(C.byref(self.ctx), None):
            raise RuntimeError("hipk_ctx_create failed: no HIP device (primme_amd has no CPU path)")
        self.op = C.c_void_p()
        rc = self.lib.primme_amd_svds_operator_create(C.byref(self.op), self.ctx, self.dt, m, n, rp.ctypes.data_as(C.c_void_p),
                                                      ci.ctypes.data_as(C.c_void_p), va.ctypes.data_as(C.c_void_p))
        if rc:
            self.lib.hipk_ctx_destroy(self.ctx)
            raise RuntimeError(f"svds operator creation failed: {rc}")

    def solve(self, numSvals=1, target="largest", method="normalequations", methodStage1="DEFAULT_METHOD", eps=1e-8, aNorm=0.0,
              maxBlockSize=0, maxBasisSize=0, maxMatvecs=0, iseed=None):
        import torch
        lib, m, n = self.lib, self.m, self.n
        ps = F.PrimmeSvdsParams()
        lib.primme_svds_initialize(C.byref(ps))
        ps.m, ps.n, ps.numSvals = m, n, numSvals
        ps.target = F.SVDS_TARGETS[target]
        ps.eps, ps.aNorm, ps.printLevel, ps.outputFile = eps, aNorm, 0, None
        if maxBlockSize: ps.maxBlockSize = maxBlockSize
        if maxBasisSize: ps.maxBasisSize = maxBasisSize
        if maxMatvecs: ps.maxMatvecs = maxMatvecs
        if iseed is not None:
            for i in range(4): ps.iseed[i] = iseed[i]
        ps.initSize = 0
        ps.matrix = self.op
        ps.matrixMatvec = C.cast(lib.primme_amd_svds_matvec, C.c_void_p)
        mset = getattr(F, "PRIMME_" + methodStage1) if hasattr(F, "PRIMME_" + methodStage1) else F.METHODS.get(methodStage1, 0)
        lib.primme_svds_set_method(F.SVDS_METHODS[method], mset, 0, C.byref(ps))
        rdtype = self.dtype
        svals = np.zeros(numSvals, dtype=rdtype); rnorms = np.zeros(numSvals, dtype=rdtype)
        tdt = torch.float64 if rdtype == np.float64 else torch.float32
        sv_t = torch.zeros((m + n) * numSvals, dtype=tdt, device="cuda")
        torch.cuda.synchronize()
        ret = self.be.svds_solver(self.dtype.name)(svals.ctypes.data_as(C.c_void_p), C.c_void_p(sv_t.data_ptr()),
                                                    rnorms.ctypes.data_as(C.c_void_p), C.byref(ps))
        torch.cuda.synchronize()
        k = ps.initSize
        return SvdsResult(ret, svals[:max(k, 0)].copy(), None, None, rnorms[:max(k, 0)].copy(), ps)

    def close(self):
        if self.op:
            self.lib.primme_amd_svds_operator_destroy(self.op); self.op = None
            self.lib.hipk_ctx_destroy(self.ctx); self.ctx = None
