"""Thin Python driver over the C ABI (plumbing only: memory, handles, parameter block).

`Session` keeps a sparse operator resident in HBM and runs hip_dprimme / hip_sprimme /
hip_zprimme / hip_cprimme solves on it; `eigsh(...)` is the one-shot form.  This package loads the
product library only (primme_amd/libprimme_amd.so).  Test code can drive the very same parameter
plumbing over another library by passing a *backend object* (see `HipBackend` for the interface);
the checker back ends live under oracle/ (oracle/checkers.py) and are never imported from here.
"""
import ctypes as C
import hashlib

import numpy as np

from . import _ffi as F


class Operator:
    """A sparse operator: CSR arrays (global column indices) or a Laplacian stencil,
    rows [row0, row0+nrows) of an n x n matrix."""

    def __init__(self, n, csr=None, stencil=None, row0=0, nrows=None):
        self.n = int(n)
        self.csr = csr            # (rowptr int32, colind int32, values)
        self.stencil = stencil    # (nx, ny, nz)
        self.row0 = int(row0)
        self.nrows = int(nrows if nrows is not None else (len(csr[0]) - 1 if csr is not None else n))

    def apply_numpy(self, x):
        from .problems import csr_matvec_numpy, laplacian_csr
        if self.csr is None:
            dims = tuple(d for d in self.stencil if d and d > 0)
            self.csr = laplacian_csr(dims, self.row0, self.nrows)[:3]
        assert self.nrows == self.n, "numpy apply is single-rank only"
        return csr_matvec_numpy(self.csr[0], self.csr[1], self.csr[2], x)

    def diagonal(self):
        if self.csr is None:
            d = len([x for x in self.stencil if x and x > 1])
            return np.full(self.nrows, 2.0 * max(d, 1))
        rp, ci, va = self.csr
        va = np.real(va)
        rows = np.repeat(np.arange(self.nrows), np.diff(rp))
        dg = np.zeros(self.nrows)
        m = ci == rows + self.row0
        dg[rows[m]] = va[m]
        return dg


class Result:
    def __init__(self, ret, evals, evecs, resNorms, params):
        self.ret, self.evals, self.evecs, self.resNorms = ret, evals, evecs, resNorms
        self.initSize = params.initSize
        s = params.stats
        self.stats = {k: getattr(s, k) for k, _ in F.PrimmeStats._fields_}
        self.params = {k: getattr(params, k) for k in ("maxBasisSize", "minRestartSize", "maxBlockSize",
                                                       "locking", "orth", "aNorm", "eps", "initSize", "dynamicMethodSwitch")}
        self.params["maxPrevRetain"] = params.restartingParams.maxPrevRetain


class HipBackend:
    """The product: libprimme_amd.so, vectors in HBM (torch tensors hold the device memory)."""
    name = "hip"
    device = True            # evecs live in HBM
    native_operator = True   # the library's own operator handle / ready-made callbacks are used

    def __init__(self):
        self.lib = F.load_product()

    def solver(self, dtype_name):
        return getattr(self.lib, {"float64": "hip_dprimme", "float32": "hip_sprimme", "complex128": "hip_zprimme",
                                  "complex64": "hip_cprimme"}[dtype_name])

    def svds_solver(self, dtype_name):
        return getattr(self.lib, {"float64": "hip_dprimme_svds", "float32": "hip_sprimme_svds", "complex128": "hip_zprimme_svds",
                                  "complex64": "hip_cprimme_svds"}[dtype_name])


def _resolve_backend(backend):
    if backend == "hip":
        return HipBackend()
    if isinstance(backend, str):
        raise ValueError(f"backend {backend!r}: primme_amd has one back end, the MI355X library; the CPU checkers "
                         "used by the tests are constructed by oracle/checkers.py")
    return backend


class Session:
    def __init__(self, op, comm=None, dtype=np.float64, backend="hip", complex_form="native", mass=None):
        self.op, self.comm = op, comm
        self.mass = mass         # Operator of the mass matrix B of a generalised problem A x = lambda B x (real, CSR), or None
        self.mass_oph = None
        self.be = _resolve_backend(backend)
        self.backend = self.be.name
        self.dtype = np.dtype(dtype)
        # Hermitian problems: complex vectors (csrc/eigs_complex.c)
        self.cplx = self.dtype.kind == "c"
        self.rdtype = np.dtype(np.float64 if self.dtype in (np.float64, np.complex128) else np.float32)
        self.dt = F.HIPK_F64 if self.rdtype == np.float64 else F.HIPK_F32
        self.handles = []
        self.keep = []
        self._v0_cache = None
        self.lib = self.be.lib
        if not self.be.native_operator:
            return
        lib = self.lib
        ctx = C.c_void_p()
        if lib.hipk_ctx_create(C.byref(ctx), None):
            raise RuntimeError("hipk_ctx_create failed: no HIP device (primme_amd has no CPU path)")
        self.handles.append(("ctx", ctx))
        A = C.c_void_p()
        if op.csr is not None:
            rp, ci, va = op.csr
            rp = np.ascontiguousarray(rp, dtype=np.int32)
            ci = np.ascontiguousarray(ci, dtype=np.int32)
            self.real_form = self.cplx and complex_form != "native"
            if self.cplx and not self.real_form:
                # a complex CSR matrix on the device: hip_zprimme / hip_cprimme run natively on complex panels
                vz = np.ascontiguousarray(va, dtype=self.dtype)
                cdt = F.HIPK_C64 if self.dtype == np.complex128 else F.HIPK_C32
                rc = lib.hipk_csr_create(ctx, cdt, op.nrows, op.n, op.row0, rp.ctypes.data_as(C.c_void_p),
                                         ci.ctypes.data_as(C.c_void_p), vz.ctypes.data_as(C.c_void_p), C.byref(A))
            elif self.cplx:
                # the real-equivalent 2n x 2n expansion (csrc/eigs_complex.c): what extractions / methods without
                # complex objects fall back to
                vz = np.ascontiguousarray(va, dtype=np.complex128)
                rp2, ci2, va2 = C.c_void_p(), C.c_void_p(), C.c_void_p()
                rc = lib.primme_amd_csr_complex_to_real(op.nrows, rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                                                        vz.ctypes.data_as(C.c_void_p), C.byref(rp2), C.byref(ci2), C.byref(va2))
                if rc:
                    raise RuntimeError(f"real-equivalent expansion failed: {rc}")
                nnz2 = 4 * int(rp[-1])
                v2 = np.ctypeslib.as_array(C.cast(va2, C.POINTER(C.c_double)), shape=(max(nnz2, 1),)).astype(self.rdtype)
                rc = lib.hipk_csr_create(ctx, self.dt, 2 * op.nrows, 2 * op.n, 2 * op.row0, rp2, ci2, v2.ctypes.data_as(C.c_void_p), C.byref(A))
                for h in (rp2, ci2, va2): lib.primme_amd_host_free(h)
            else:
                va = np.ascontiguousarray(va, dtype=self.dtype)
                rc = lib.hipk_csr_create(ctx, self.dt, op.nrows, op.n, op.row0, rp.ctypes.data_as(C.c_void_p),
                                         ci.ctypes.data_as(C.c_void_p), va.ctypes.data_as(C.c_void_p), C.byref(A))
        else:
            nx, ny, nz = (list(op.stencil) + [1, 1])[:3]
            rc = lib.hipk_stencil_create(ctx, self.dt, nx, ny or 1, nz or 1, op.row0, op.nrows, C.byref(A))
        if rc:
            raise RuntimeError(f"operator creation failed: {rc}")
        self.handles.append(("csr", A))
        oph = C.c_void_p()
        if lib.primme_amd_operator_create(C.byref(oph), A, comm):
            raise RuntimeError("operator handle creation failed")
        self.handles.append(("op", oph))
        if self.cplx and self.real_form: lib.primme_amd_operator_set_complex(oph, 1)
        self.oph = oph
        if mass is not None:
            if (self.cplx and self.real_form) or mass.csr is None:
                raise ValueError("mass matrix: CSR operators, complex ones on the native complex panels")
            rpb, cib, vab = mass.csr
            rpb = np.ascontiguousarray(rpb, dtype=np.int32); cib = np.ascontiguousarray(cib, dtype=np.int32)
            vab = np.ascontiguousarray(vab, dtype=self.dtype)
            Bh = C.c_void_p()
            bdt = self.dt if not self.cplx else (F.HIPK_C64 if self.dtype == np.complex128 else F.HIPK_C32)
            if lib.hipk_csr_create(ctx, bdt, mass.nrows, mass.n, mass.row0, rpb.ctypes.data_as(C.c_void_p),
                                   cib.ctypes.data_as(C.c_void_p), vab.ctypes.data_as(C.c_void_p), C.byref(Bh)):
                raise RuntimeError("mass matrix creation failed")
            self.handles.append(("csr", Bh))
            ophB = C.c_void_p()
            if lib.primme_amd_operator_create(C.byref(ophB), Bh, comm):
                raise RuntimeError("mass operator handle creation failed")
            self.handles.append(("op", ophB))
            self.mass_oph = ophB

    def close(self):
        for kind, h in reversed(self.handles):
            if kind == "op": self.lib.primme_amd_operator_destroy(h)
            elif kind == "csr": self.lib.hipk_csr_destroy(h)
            elif kind == "ctx": self.lib.hipk_ctx_destroy(h)
        self.handles = []

    def solve(self, numEvals=1, target="smallest", method="GD_plusK", eps=1e-8, aNorm=0.0, v0=None,
              maxBlockSize=0, maxBasisSize=0, minRestartSize=0, maxPrevRetain=None, locking=None,
              maxMatvecs=0, maxOuterIterations=0, targetShifts=None, precond=None, printLevel=0,
              initBasisMode=None, global_sum=None, numProcs=1, procID=0, orth=None, iseed=None,
              profile=False, return_evecs=True, monitor=None, user_matvec=None, projection=None,
              constraints=None, user_precond=None, tweak=None):
        lib, op, dtype = self.lib, self.op, self.dtype
        keep = []
        p = F.PrimmeParams()
        nLocal = op.nrows
        v0 = None if v0 is None else np.asarray(v0, dtype=dtype).reshape(nLocal, -1)
        initSize = 0 if v0 is None else v0.shape[1]
        if initBasisMode is None and v0 is not None:
            initBasisMode = F.primme_init_user

        lib.primme_initialize(C.byref(p))
        p.n = op.n
        p.numEvals = numEvals
        p.target = F.TARGETS[target] if isinstance(target, str) else target
        p.eps, p.aNorm, p.printLevel, p.outputFile = eps, aNorm, printLevel, None
        if maxBlockSize: p.maxBlockSize = maxBlockSize
        if maxBasisSize: p.maxBasisSize = maxBasisSize
        if minRestartSize: p.minRestartSize = minRestartSize
        if maxPrevRetain is not None: p.restartingParams.maxPrevRetain = maxPrevRetain
        if locking is not None: p.locking = locking
        if maxMatvecs: p.maxMatvecs = maxMatvecs
        if maxOuterIterations: p.maxOuterIterations = maxOuterIterations
        if orth is not None: p.orth = orth
        if projection is not None:
            p.projectionParams.projection = {"RR": 1, "harmonic": 2, "refined": 3}[projection]
        if iseed is not None:
            for i in range(4): p.iseed[i] = iseed[i]
        if targetShifts is not None:
            ts = (C.c_double * len(targetShifts))(*targetShifts)
            keep.append(ts)
            p.targetShifts, p.numTargetShifts = ts, len(targetShifts)
        p.initSize = initSize
        # orthogonality constraints: the first numOrthoConst columns of evecs (primme_eigs.h:266-269)
        cons = None if constraints is None else np.asarray(constraints, dtype=dtype).reshape(nLocal, -1)
        nOC = 0 if cons is None else cons.shape[1]
        p.numOrthoConst = nOC
        if initBasisMode is not None: p.initBasisMode = initBasisMode
        p.numProcs, p.procID, p.nLocal = numProcs, procID, nLocal
        m = F.METHODS[method] if isinstance(method, str) else method
        ncols = nOC + max(numEvals, initSize)
        ctype = C.c_double if self.rdtype == np.float64 else C.c_float
        cplx = self.cplx
        evecs_t = None

        def view(ptr, nb, ld):
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(nb, ld * (2 if cplx else 1)))
            return a.view(dtype) if cplx else a

        if not self.be.native_operator:
            # a checker that brings its own operator callbacks (oracle/checkers.py)
            evecs, evecs_ptr = self.be.setup_operator(self, p, keep, precond, view, ncols, nLocal, cons, nOC, v0, initSize)
            solver = self.be.solver(dtype.name)
        else:
            p.matrix = self.oph
            p.matrixMatvec = C.cast(lib.primme_amd_matvec, C.c_void_p)
            if self.mass_oph is not None:    # generalised problem: B through the ready-made callback over a second operator handle
                p.massMatrix = self.mass_oph
                p.massMatrixMatvec = C.cast(lib.primme_amd_mass_matvec, C.c_void_p)
            if user_matvec is not None:      # an application callback instead of the ready-made one
                keep.append(user_matvec)
                p.matrixMatvec = C.cast(user_matvec, C.c_void_p)
            if user_precond is not None:     # an application preconditioner callback (same pointer conventions as the matvec)
                keep.append(user_precond)
                p.applyPreconditioner = C.cast(user_precond, C.c_void_p)
                p.correctionParams.precondition = 1
            elif precond is not None:
                # "jacobi": per-vector shifts of the solver; ("jacobi", s): fixed K = diag(A) - s
                if precond == "jacobi": lib.primme_amd_operator_set_jacobi(self.oph, 0, 0.0)
                else: lib.primme_amd_operator_set_jacobi(self.oph, 1, float(precond[1]))
                p.preconditioner = self.oph
                p.applyPreconditioner = C.cast(lib.primme_amd_jacobi_precond, C.c_void_p)
                p.correctionParams.precondition = 1
            if self.comm is not None:
                p.commInfo = self.comm
                p.globalSumReal = C.cast(lib.primme_amd_global_sum, C.c_void_p)
            if profile:
                p.profile = b"phases"
            if self.be.device:
                import torch
                tdt = {"float64": torch.float64, "float32": torch.float32, "complex128": torch.complex128, "complex64": torch.complex64}[dtype.name]
                # (a zero-sized problem still gets a valid device pointer)
                evecs_buf = torch.zeros(max(ncols * nLocal, 1), dtype=tdt, device="cuda")
                evecs_t = evecs_buf[:ncols * nLocal].view(ncols, nLocal)
                if v0 is not None:
                    # the start vectors are uploaded once per Session and stay in HBM
                    key = (v0.shape, v0.dtype.str, hashlib.sha1(np.ascontiguousarray(v0).tobytes()).hexdigest())
                    if self._v0_cache is None or self._v0_cache[0] != key:
                        self._v0_cache = (key, torch.from_numpy(np.ascontiguousarray(v0.T)).to("cuda"))
                    evecs_t[nOC:nOC + initSize] = self._v0_cache[1]
                if cons is not None:
                    evecs_t[:nOC] = torch.from_numpy(np.ascontiguousarray(cons.T)).to("cuda")
                torch.cuda.synchronize()
                evecs_ptr = C.c_void_p(evecs_buf.data_ptr())
            else:
                evecs = np.zeros((ncols, nLocal), dtype=dtype)
                if cons is not None:
                    evecs[:nOC] = cons.T
                if v0 is not None:
                    evecs[nOC:nOC + initSize] = v0.T
                evecs_ptr = evecs.ctypes.data_as(C.c_void_p)
            solver = self.be.solver(dtype.name)

        if global_sum is not None:
            def gs(send, recv, count, pp, ierr):
                # operands arrive in the declared type: float for the single-precision entry points
                # unless the application set globalSumReal_type (reference primme_c.c:170-183)
                n_ = count[0]
                ct = C.c_float if pp[0].globalSumReal_type == F.primme_op_float else C.c_double
                a = np.ctypeslib.as_array(C.cast(send, C.POINTER(ct)), shape=(n_,)).astype(np.float64)
                np.ctypeslib.as_array(C.cast(recv, C.POINTER(ct)), shape=(n_,))[:] = global_sum(a)
                ierr[0] = 0
            gcb = F.GLOBAL_SUM(gs)
            keep.append(gcb)
            p.globalSumReal = C.cast(gcb, C.c_void_p)
        if monitor is not None:
            mcb = F.MONITOR(monitor)
            keep.append(mcb)
            p.monitorFun = C.cast(mcb, C.c_void_p)

        if lib.primme_set_method(m, C.byref(p)):
            raise ValueError("unknown method")
        if tweak is not None:            # last word on the parameter structure (e.g. the correction equation's projectors)
            tweak(p)
        evals = np.zeros(numEvals, dtype=self.rdtype)
        resNorms = np.zeros(numEvals, dtype=self.rdtype)
        ret = solver(evals.ctypes.data_as(C.c_void_p), evecs_ptr, resNorms.ctypes.data_as(C.c_void_p), C.byref(p))
        if self.be.native_operator and self.be.device:
            import torch
            torch.cuda.synchronize()
            evecs = evecs_t.cpu().numpy() if return_evecs else None
        return Result(ret, evals, None if evecs is None else evecs[nOC:nOC + numEvals].T.copy(), resNorms, p)


def eigsh(op, backend="hip", comm=None, dtype=np.float64, complex_form="native", mass=None, **kw):
    """One-shot: compute a few eigenpairs of the symmetric operator `op` (see Session.solve).

    v0: optional (nLocal x initSize) initial guesses -> initBasisMode defaults to
    primme_init_user so that no random numbers enter (parity runs).
    precond: None | "jacobi" (Davidson: per-vector shifts) | ("jacobi", shift) (fixed shift).
    """
    s = Session(op, comm=comm, dtype=dtype, backend=backend, complex_form=complex_form, mass=mass)
    try:
        return s.solve(**kw)
    finally:
        s.close()
