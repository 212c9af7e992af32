"""oracle/checkers.py — TEST INFRASTRUCTURE ONLY: the CPU checker back ends.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package primme_amd/ cannot load anything under oracle/ (it has no path to it).

    hostcheck   oracle/_build/libprimme_hostcheck.so: the product's host solver sources linked over
                the plain-C kernel restatement oracle/hipk_cpu.c (vectors in host memory)
    reference   oracle/_ref/libprimme_ref.so: the REAL reference, compiled by oracle/Makefile from
                /root/reference (never copied); operators are numpy callbacks

eigsh / Session / svds below take backend="hip" | "hostcheck" | "reference" and drive the same
parameter plumbing (primme_amd/api.py, svds_api.py) over the chosen library, so that a test compares
like with like.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from primme_amd import _ffi as F                     # noqa: E402
from primme_amd import api as _api                   # noqa: E402
from primme_amd import svds_api as _svds_api         # noqa: E402
from primme_amd.api import Operator, Result          # noqa: E402,F401
from primme_amd.problems import csr_matvec_numpy     # noqa: E402
from primme_amd.svds_api import transpose_csr        # noqa: E402,F401

# PRIMME_AMD_HOSTCHECK_LIB: another build of the checker (the AddressSanitizer build, scripts/build_hostasan.sh)
HOSTCHECK_LIB = os.environ.get("PRIMME_AMD_HOSTCHECK_LIB") or os.path.join(_HERE, "_build", "libprimme_hostcheck.so")
REFERENCE_LIB = os.path.join(_HERE, "_ref", "libprimme_ref.so")
PRODUCT_LIB = F.PRODUCT_LIB

_cache = {}


def load_hostcheck():
    """Product host solver linked over oracle/hipk_cpu.c."""
    if "hostcheck" not in _cache:
        lib = C.CDLL(HOSTCHECK_LIB)
        F.declare_solver(lib, "hip_")
        F.declare_kernels(lib)
        _cache["hostcheck"] = lib
    return _cache["hostcheck"]


def load_reference():
    """The real reference built by oracle/Makefile from /root/reference."""
    if "reference" not in _cache:
        # The reference is linked with the image's MKL (libmkl_rt), whose default threading layer brings Intel's OpenMP
        # runtime; a process that has imported torch (primme_amd._ffi does, for the one HIP runtime) already runs GNU's.
        # Two OpenMP runtimes under MKL make the reference return garbage or hang — whether it happened depended on which
        # test module touched MKL first (round 5: a new module moved the order and three reference runs went wrong).
        # One runtime for everybody; must be in the environment before MKL's first call.
        os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
        lib = C.CDLL(REFERENCE_LIB)
        F.declare_solver(lib, "")
        _cache["reference"] = lib
    return _cache["reference"]


load_product = F.load_product


class HostcheckBackend:
    name = "hostcheck"
    device = False
    native_operator = True

    def __init__(self):
        self.lib = load_hostcheck()

    def solver(self, dtype_name):
        return getattr(self.lib, {"float64": "hip_dprimme", "float32": "hip_sprimme", "complex128": "hip_zprimme",
                                  "complex64": "hip_cprimme"}[dtype_name])

    def svds_solver(self, dtype_name):
        return getattr(self.lib, {"float64": "hip_dprimme_svds", "float32": "hip_sprimme_svds", "complex128": "hip_zprimme_svds",
                                  "complex64": "hip_cprimme_svds"}[dtype_name])


class ReferenceBackend:
    """dprimme / sprimme / zprimme / cprimme of the reference with numpy operator callbacks."""
    name = "reference"
    device = False
    native_operator = False

    def __init__(self):
        self.lib = load_reference()

    def solver(self, dtype_name):
        return getattr(self.lib, {"float64": "dprimme", "float32": "sprimme", "complex128": "zprimme",
                                  "complex64": "cprimme"}[dtype_name])

    def setup_operator(self, sess, p, keep, precond, view, ncols, nLocal, cons, nOC, v0, initSize):
        op, cplx, dtype = sess.op, sess.cplx, sess.dtype

        def mv(x, ldx, y, ldy, bs, pp, ierr):
            nb, lx, ly = bs[0], ldx[0], ldy[0]
            X = view(x, nb, lx)
            Y = view(y, nb, ly)
            Y[:, :nLocal] = op.apply_numpy(X[:, :nLocal].T.astype(np.complex128 if cplx else np.float64)).T
            ierr[0] = 0
        cb = F.BLOCK_OP(mv)
        keep.append(cb)
        p.matrixMatvec = C.cast(cb, C.c_void_p)
        if sess.mass is not None:      # generalised problem: the mass matrix as a numpy callback

            def bmv(x, ldx, y, ldy, bs, pp, ierr):
                nb, lx, ly = bs[0], ldx[0], ldy[0]
                X = view(x, nb, lx)
                Y = view(y, nb, ly)
                Y[:, :nLocal] = sess.mass.apply_numpy(X[:, :nLocal].T.astype(np.complex128 if cplx else np.float64)).T
                ierr[0] = 0
            bcb = F.BLOCK_OP(bmv)
            keep.append(bcb)
            p.massMatrixMatvec = C.cast(bcb, C.c_void_p)
        if precond is not None:
            dg = np.real(op.diagonal())        # Hermitian: the diagonal is real (divide by a real number, as the device kernel does)
            zrot = None
            if precond != "jacobi" and precond[0] == "zjacobi":
                # NON-Hermitian diagonal preconditioner K = diag(A) (1 + i gamma w_j): makes x'K^-1 x complex
                # (the reference keeps it as an HSCALAR, src/eigs/correction.c:969-977); examples/ex_eigs_zhip_precond.hip
                from primme_amd.problems import zjacobi_rotation
                zrot = zjacobi_rotation(len(dg), float(precond[1]))
            jfixed = None if precond == "jacobi" else 0.0 if zrot is not None else float(precond[1])   # zjacobi: K = diag(A) (1 + i gamma w)

            def pc(x, ldx, y, ldy, bs, pp, ierr):
                nb, lx, ly = bs[0], ldx[0], ldy[0]
                if nb <= 0 or not x or not y:
                    ierr[0] = 0
                    return
                X = view(x, nb, lx)
                Y = view(y, nb, ly)
                sh = pp[0].ShiftsForPreconditioner
                an = pp[0].aNorm
                mind = 1e-14 * (an if an >= 0 else 1.0)
                for c in range(nb):
                    d = dg - (jfixed if jfixed is not None else (sh[c] if sh else 0.0))
                    small = ~(np.abs(d) > mind)
                    d[small] = np.copysign(mind, d[small])
                    Y[c, :nLocal] = X[c, :nLocal] / (d if zrot is None else d * zrot)
                ierr[0] = 0
            pcb = F.BLOCK_OP(pc)
            keep.append(pcb)
            p.applyPreconditioner = C.cast(pcb, C.c_void_p)
            p.correctionParams.precondition = 1
        evecs = np.zeros((ncols, nLocal), dtype=dtype)  # row-major (ncols x n) == col-major n x ncols
        if cons is not None:
            evecs[:nOC] = cons.T
        if v0 is not None:
            evecs[nOC:nOC + initSize] = v0.T
        return evecs, evecs.ctypes.data_as(C.c_void_p)

    def setup_svds_operator(self, ps, keep, m, n, rp, ci, va, ctype, precond, dtype):
        cplx = np.dtype(dtype).kind == "c"
        rpT, ciT, vaT = transpose_csr(m, n, rp, ci, va)
        if cplx:
            vaT = np.conj(vaT)                    # the callback's transpose flag means A^H
        wide = np.complex128 if cplx else np.float64

        def mv(x, ldx, y, ldy, bs, tr, pp, ierr):
            nb, lx, ly = bs[0], ldx[0], ldy[0]
            if cplx:
                X = np.ctypeslib.as_array(C.cast(x, C.POINTER(ctype)), shape=(nb, 2 * lx)).view(dtype)
                Y = np.ctypeslib.as_array(C.cast(y, C.POINTER(ctype)), shape=(nb, 2 * ly)).view(dtype)
            else:
                X = np.ctypeslib.as_array(C.cast(x, C.POINTER(ctype)), shape=(nb, lx))
                Y = np.ctypeslib.as_array(C.cast(y, C.POINTER(ctype)), shape=(nb, ly))
            if tr[0]:
                Y[:, :n] = csr_matvec_numpy(rpT, ciT, vaT, X[:, :m].T.astype(wide)).T
            else:
                Y[:, :m] = csr_matvec_numpy(rp, ci, va, X[:, :n].T.astype(wide)).T
            ierr[0] = 0
        cb = F.SVDS_BLOCK_OP(mv)
        keep.append(cb)
        ps.matrixMatvec = C.cast(cb, C.c_void_p)
        if precond is not None:
            # the test driver's "jacobi" for singular value problems (tests/COMMON/mat.c:353-426)
            shift = 0.0 if precond == "jacobi" else float(precond[1])
            rows = np.repeat(np.arange(m), np.diff(rp))
            sumr = np.bincount(rows, weights=va.astype(np.float64) ** 2, minlength=m) - shift * shift
            sumc = np.bincount(ci, weights=va.astype(np.float64) ** 2, minlength=n) - shift * shift
            for d in (sumr, sumc):
                small = np.abs(d) < 1e-14
                d[small] = np.copysign(1e-14, d[small])

            def pc(x, ldx, y, ldy, bs, mode, pp, ierr):
                nb, lx, ly = bs[0], ldx[0], ldy[0]
                if nb <= 0 or not x or not y:
                    ierr[0] = 0
                    return
                X = np.ctypeslib.as_array(C.cast(x, C.POINTER(ctype)), shape=(nb, lx))
                Y = np.ctypeslib.as_array(C.cast(y, C.POINTER(ctype)), shape=(nb, ly))
                an = pp[0].aNorm
                md = 1e-14 * (an if an >= 0 else 1.0)

                def div(d):
                    d = d.copy()
                    small = ~(np.abs(d) > md)
                    d[small] = np.copysign(md, d[small])
                    return d
                if mode[0] == 1: Y[:, :n] = X[:, :n] / div(sumc)
                elif mode[0] == 2: Y[:, :m] = X[:, :m] / div(sumr)
                else:
                    Y[:, :n] = X[:, :n] / div(sumc)
                    Y[:, n:n + m] = X[:, n:n + m] / div(sumr)
                ierr[0] = 0
            pcb = F.SVDS_BLOCK_OP(pc)
            keep.append(pcb)
            ps.applyPreconditioner = C.cast(pcb, C.c_void_p)
        if cplx and precond is not None:
            raise ValueError("the test preconditioner is for real matrices")
        return getattr(self.lib, {"float64": "dprimme_svds", "float32": "sprimme_svds", "complex128": "zprimme_svds",
                                  "complex64": "cprimme_svds"}[np.dtype(dtype).name])


def backend_object(backend):
    if backend == "hip" and os.environ.get("PRIMME_AMD_TESTS_HIP_IS_HOSTCHECK"):
        # heap-corruption hunt (round 4, profiles/r04_gpu_suite_exit_crash.md): run the GPU test modules' own Python and the host
        # solver paths they reach on a box without a GPU, under AddressSanitizer, with the plain-C checker standing in for the device
        return HostcheckBackend()
    if backend == "hip":
        return _api.HipBackend()
    if backend == "hostcheck":
        return HostcheckBackend()
    if backend == "reference":
        return ReferenceBackend()
    if isinstance(backend, str):
        raise ValueError(backend)
    return backend


class Session(_api.Session):
    def __init__(self, op, comm=None, dtype=np.float64, backend="hip", complex_form="native", mass=None):
        super().__init__(op, comm=comm, dtype=dtype, backend=backend_object(backend), complex_form=complex_form, mass=mass)


def eigsh(op, backend="hip", comm=None, dtype=np.float64, complex_form="native", mass=None, **kw):
    s = Session(op, comm=comm, dtype=dtype, backend=backend, complex_form=complex_form, mass=mass)
    try:
        return s.solve(**kw)
    finally:
        s.close()


def svds(m, n, csr, backend="hip", **kw):
    return _svds_api.svds(m, n, csr, backend=backend_object(backend), **kw)
