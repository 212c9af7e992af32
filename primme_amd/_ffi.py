"""ctypes view of the C ABI in include/primme_amd.h / primme_amd_kernels.h / primme_amd_comm.h.

The structures mirror reference include/primme_eigs.h:109-253 field for field.  This module
loads ONE library: the product, primme_amd/libprimme_amd.so (GPU only).  The test-only checker
builds are loaded by oracle/checkers.py, which declares the same signatures through
declare_solver / declare_kernels.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)

PRIMME_INT = C.c_int64

# enums (values are ABI, reference include/primme_eigs.h:47-107)
primme_smallest, primme_largest, primme_closest_geq, primme_closest_leq, primme_closest_abs, primme_largest_abs = range(6)
primme_proj_default, primme_proj_RR, primme_proj_harmonic, primme_proj_refined = range(4)
primme_init_default, primme_init_krylov, primme_init_random, primme_init_user = range(4)
primme_orth_default, primme_orth_implicit_I, primme_orth_explicit_I = range(3)
primme_op_default, primme_op_half, primme_op_float, primme_op_double, primme_op_quad, primme_op_int = range(6)
(PRIMME_DEFAULT_METHOD, PRIMME_DYNAMIC, PRIMME_DEFAULT_MIN_TIME, PRIMME_DEFAULT_MIN_MATVECS, PRIMME_Arnoldi,
 PRIMME_GD, PRIMME_GD_plusK, PRIMME_GD_Olsen_plusK, PRIMME_JD_Olsen_plusK, PRIMME_RQI, PRIMME_JDQR,
 PRIMME_JDQMR, PRIMME_JDQMR_ETol, PRIMME_STEEPEST_DESCENT, PRIMME_LOBPCG_OrthoBasis,
 PRIMME_LOBPCG_OrthoBasis_Window) = range(16)
METHODS = {
    "DYNAMIC": PRIMME_DYNAMIC, "DEFAULT_MIN_TIME": PRIMME_DEFAULT_MIN_TIME,
    "DEFAULT_MIN_MATVECS": PRIMME_DEFAULT_MIN_MATVECS, "Arnoldi": PRIMME_Arnoldi, "GD": PRIMME_GD,
    "GD_plusK": PRIMME_GD_plusK, "GD_Olsen_plusK": PRIMME_GD_Olsen_plusK,
    "JD_Olsen_plusK": PRIMME_JD_Olsen_plusK, "RQI": PRIMME_RQI, "JDQR": PRIMME_JDQR,
    "JDQMR": PRIMME_JDQMR, "JDQMR_ETol": PRIMME_JDQMR_ETol,
    "STEEPEST_DESCENT": PRIMME_STEEPEST_DESCENT, "LOBPCG_OrthoBasis": PRIMME_LOBPCG_OrthoBasis,
    "LOBPCG_OrthoBasis_Window": PRIMME_LOBPCG_OrthoBasis_Window,
}
TARGETS = {"smallest": 0, "largest": 1, "closest_geq": 2, "closest_leq": 3, "closest_abs": 4, "largest_abs": 5}

HIPK_F64, HIPK_F32, HIPK_C64, HIPK_C32 = range(4)
HIPK_JOB_XV, HIPK_JOB_XW, HIPK_JOB_RES = range(3)


class PrimmeStats(C.Structure):
    _fields_ = [(n, PRIMME_INT) for n in (
        "numOuterIterations", "numRestarts", "numMatvecs", "numPreconds", "numGlobalSum", "numBroadcast",
        "volumeGlobalSum", "volumeBroadcast")] + [(n, C.c_double) for n in (
        "flopsDense", "numOrthoInnerProds", "elapsedTime", "timeMatvec", "timePrecond", "timeOrtho",
        "timeGlobalSum", "timeBroadcast", "timeDense", "estimateMinEVal", "estimateMaxEVal",
        "estimateLargestSVal", "estimateBNorm", "estimateInvBNorm", "maxConvTol",
        "estimateResidualError")] + [("lockingIssue", PRIMME_INT)]


class JDProjectors(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("LeftQ", "LeftX", "RightQ", "RightX", "SkewQ", "SkewX")]


class ProjectionParams(C.Structure):
    _fields_ = [("projection", C.c_int)]


class CorrectionParams(C.Structure):
    _fields_ = [("precondition", C.c_int), ("robustShifts", C.c_int), ("maxInnerIterations", C.c_int),
                ("projectors", JDProjectors), ("convTest", C.c_int), ("relTolBase", C.c_double)]


class RestartingParams(C.Structure):
    _fields_ = [("maxPrevRetain", C.c_int)]


class PrimmeParams(C.Structure):
    pass


BLOCK_OP = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(PRIMME_INT), C.c_void_p, C.POINTER(PRIMME_INT),
                       C.POINTER(C.c_int), C.POINTER(PrimmeParams), C.POINTER(C.c_int))
GLOBAL_SUM = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(PrimmeParams),
                         C.POINTER(C.c_int))
BROADCAST = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int), C.POINTER(PrimmeParams), C.POINTER(C.c_int))
CONVTEST = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int),
                       C.POINTER(PrimmeParams), C.POINTER(C.c_int))
MONITOR = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                      C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int),
                      C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_char_p,
                      C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(PrimmeParams), C.POINTER(C.c_int))

PrimmeParams._fields_ = [
    ("n", PRIMME_INT),
    ("matrixMatvec", C.c_void_p), ("matrixMatvec_type", C.c_int),
    ("applyPreconditioner", C.c_void_p), ("applyPreconditioner_type", C.c_int),
    ("massMatrixMatvec", C.c_void_p), ("massMatrixMatvec_type", C.c_int),
    ("numProcs", C.c_int), ("procID", C.c_int), ("nLocal", PRIMME_INT), ("commInfo", C.c_void_p),
    ("globalSumReal", C.c_void_p), ("globalSumReal_type", C.c_int),
    ("broadcastReal", C.c_void_p), ("broadcastReal_type", C.c_int),
    ("numEvals", C.c_int), ("target", C.c_int), ("numTargetShifts", C.c_int),
    ("targetShifts", C.POINTER(C.c_double)),
    ("dynamicMethodSwitch", C.c_int), ("locking", C.c_int), ("initSize", C.c_int), ("numOrthoConst", C.c_int),
    ("maxBasisSize", C.c_int), ("minRestartSize", C.c_int), ("maxBlockSize", C.c_int),
    ("maxMatvecs", PRIMME_INT), ("maxOuterIterations", PRIMME_INT), ("iseed", PRIMME_INT * 4),
    ("aNorm", C.c_double), ("BNorm", C.c_double), ("invBNorm", C.c_double), ("eps", C.c_double),
    ("orth", C.c_int), ("internalPrecision", C.c_int),
    ("printLevel", C.c_int), ("outputFile", C.c_void_p),
    ("matrix", C.c_void_p), ("preconditioner", C.c_void_p), ("massMatrix", C.c_void_p),
    ("ShiftsForPreconditioner", C.POINTER(C.c_double)), ("initBasisMode", C.c_int),
    ("ldevecs", PRIMME_INT), ("ldOPs", PRIMME_INT),
    ("projectionParams", ProjectionParams), ("restartingParams", RestartingParams),
    ("correctionParams", CorrectionParams), ("stats", PrimmeStats),
    ("convTestFun", C.c_void_p), ("convTestFun_type", C.c_int), ("convtest", C.c_void_p),
    ("monitorFun", C.c_void_p), ("monitorFun_type", C.c_int), ("monitor", C.c_void_p),
    ("queue", C.c_void_p), ("profile", C.c_char_p),
]


class PrimmeSvdsStats(C.Structure):
    _fields_ = [(n, PRIMME_INT) for n in (
        "numOuterIterations", "numRestarts", "numMatvecs", "numPreconds", "numGlobalSum", "numBroadcast",
        "volumeGlobalSum", "volumeBroadcast")] + [(n, C.c_double) for n in (
        "numOrthoInnerProds", "elapsedTime", "timeMatvec", "timePrecond", "timeOrtho", "timeGlobalSum",
        "timeBroadcast")] + [("lockingIssue", PRIMME_INT)]


class PrimmeSvdsParams(C.Structure):
    """primme_svds_params (reference include/primme_svds.h:84-168; include/primme_amd_svds.h)."""
    _fields_ = [
        ("primme", PrimmeParams), ("primmeStage2", PrimmeParams), ("m", PRIMME_INT), ("n", PRIMME_INT),
        ("matrixMatvec", C.c_void_p), ("matrixMatvec_type", C.c_int),
        ("applyPreconditioner", C.c_void_p), ("applyPreconditioner_type", C.c_int),
        ("numProcs", C.c_int), ("procID", C.c_int), ("mLocal", PRIMME_INT), ("nLocal", PRIMME_INT),
        ("commInfo", C.c_void_p), ("globalSumReal", C.c_void_p), ("globalSumReal_type", C.c_int),
        ("broadcastReal", C.c_void_p), ("broadcastReal_type", C.c_int),
        ("numSvals", C.c_int), ("target", C.c_int), ("numTargetShifts", C.c_int),
        ("targetShifts", C.POINTER(C.c_double)), ("method", C.c_int), ("methodStage2", C.c_int),
        ("matrix", C.c_void_p), ("preconditioner", C.c_void_p), ("locking", C.c_int), ("numOrthoConst", C.c_int),
        ("aNorm", C.c_double), ("eps", C.c_double), ("precondition", C.c_int), ("initSize", C.c_int),
        ("maxBasisSize", C.c_int), ("maxBlockSize", C.c_int), ("maxMatvecs", PRIMME_INT),
        ("iseed", PRIMME_INT * 4), ("printLevel", C.c_int), ("internalPrecision", C.c_int),
        ("outputFile", C.c_void_p), ("stats", PrimmeSvdsStats),
        ("convTestFun", C.c_void_p), ("convTestFun_type", C.c_int), ("convtest", C.c_void_p),
        ("monitorFun", C.c_void_p), ("monitorFun_type", C.c_int), ("monitor", C.c_void_p),
        ("queue", C.c_void_p), ("profile", C.c_char_p),
    ]


SVDS_BLOCK_OP = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(PRIMME_INT), C.c_void_p, C.POINTER(PRIMME_INT),
                            C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(PrimmeSvdsParams), C.POINTER(C.c_int))
SVDS_TARGETS = {"largest": 0, "smallest": 1, "closest_abs": 2}
SVDS_METHODS = {"default": 0, "hybrid": 1, "normalequations": 2, "augmented": 3}


class HipkSeg(C.Structure):
    _fields_ = [("base", C.c_void_p), ("ld", C.c_int64), ("ncols", C.c_int)]


class HipkJob(C.Structure):
    _fields_ = [("kind", C.c_int), ("col", C.c_int), ("dst", C.c_void_p), ("slot", C.c_int)]


class HipkRrIn(C.Structure):
    """include/primme_amd_kernels.h: hipk_rr_in (what the host passes to the one-wave Rayleigh-Ritz kernel by value)"""
    _fields_ = [("k", C.c_int), ("L", C.c_int), ("cand", C.c_int), ("largest", C.c_int), ("grow_row", C.c_int), ("pad", C.c_int),
                ("theta", C.c_double * 16), ("Y", C.c_double * 256), ("G", C.c_double * 160)]


# PRIMME_AMD_LIB: another build of the product library (measurement only: the build-time variants of scripts/build_variant.sh)
PRODUCT_LIB = os.environ.get("PRIMME_AMD_LIB") or os.path.join(_HERE, "libprimme_amd.so")

_vp, _i, _i64, _dp = C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_double)


def declare_solver(lib, prefix):
    lib.primme_initialize.argtypes = [C.POINTER(PrimmeParams)]
    lib.primme_initialize.restype = None
    lib.primme_set_method.argtypes = [C.c_int, C.POINTER(PrimmeParams)]
    lib.primme_set_method.restype = C.c_int
    lib.primme_svds_initialize.argtypes = [C.POINTER(PrimmeSvdsParams)]
    lib.primme_svds_initialize.restype = None
    lib.primme_svds_set_method.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(PrimmeSvdsParams)]
    lib.primme_svds_set_method.restype = C.c_int
    for t in "dszc":
        f = getattr(lib, f"{prefix}{t}primme", None)
        if f is not None:
            f.argtypes = [_vp, _vp, _vp, C.POINTER(PrimmeParams)]
            f.restype = C.c_int
        f = getattr(lib, f"{prefix}{t}primme_svds", None)
        if f is not None:
            f.argtypes = [_vp, _vp, _vp, C.POINTER(PrimmeSvdsParams)]
            f.restype = C.c_int
            f.restype = C.c_int


def declare_kernels(lib):
    P = C.POINTER
    sig = {
        "hipk_ctx_create": [P(_vp), _vp], "hipk_ctx_destroy": [_vp], "hipk_sync": [_vp],
        "hipk_malloc": [_vp, C.c_size_t, P(_vp)], "hipk_free": [_vp, _vp],
        "hipk_h2d": [_vp, _vp, _vp, C.c_size_t], "hipk_d2h": [_vp, _vp, _vp, C.c_size_t],
        "hipk_host_alloc": [_vp, C.c_size_t, P(_vp)], "hipk_host_free": [_vp, _vp],
        "hipk_larnv_uniform11": [_vp, _i, P(C.c_int64), _i64, _vp],
        "hipk_timer_start": [_vp], "hipk_timer_stop": [_vp, P(C.c_float)],
        "hipk_panel_dots": [_vp, _i, _i64, P(HipkSeg), _i, _vp, _i64, _i, _vp, _i],
        "hipk_panel_project": [_vp, _i, _i64, P(HipkSeg), _i, _vp, _i, _vp, _i64, _i, _vp],
        "hipk_ritz_update": [_vp, _i, _i64, _vp, _vp, _i64, _i, _vp, _i, _vp, P(HipkJob), _i, _vp],
        "hipk_ritz_residual_overlaps": [_vp, _i, _i64, _vp, _vp, _i64, _i, _vp, C.c_double, _vp, _vp, _i64, _i, _i, _vp],
        "hipk_ritz_residual_overlaps_dev": [_vp, _i, _i64, _vp, _vp, _i64, _i, _vp, _vp, _vp, _i64, _i, _i, _vp],
        "hipk_rr_arrow": [_vp, P(HipkRrIn), _vp, _i, _vp, _vp],
        "hipk_tail_defer": [_vp, _i], "hipk_tail_pending": [_vp],
        "hipk_tail_finish": [_vp, P(HipkRrIn), _vp, _i, _vp, _vp],
        "hipk_ritz_update_overlaps": [_vp, _i, _i64, _vp, _vp, _i64, _i, _vp, _i, _vp, P(HipkJob), _i, _vp, _i, _vp, _i64, _i, _vp],
        "hipk_pair_dots": [_vp, _i, _i64, _vp, _i64, _vp, _i64, _i, _vp],
        "hipk_sym_eig": [_vp, _i, _vp, _i, _vp, _vp, _i],
        "hipk_xpay_cols": [_vp, _i, _i64, _dp, _vp, _i64, _vp, _i64, _i],
        "hipk_axpy_dot": [_vp, _i, _i64, _i, _dp, _vp, _i64, _vp, _i64, _vp, _i64, _vp],
        "hipk_qmr_update": [_vp, _i, _i64, _i, _dp, _dp, _vp, _i64, _vp, _i64, _vp, _i64, _vp],
        "hipk_scale_cols": [_vp, _i, _i64, _vp, _i64, _i, _dp],
        "hipk_scale_cols_rsqrt_dev": [_vp, _i, _i64, _vp, _i64, _i, _vp],
        "hipk_axpy_cols": [_vp, _i, _i64, _dp, _vp, _i64, _vp, _i64, _i],
        "hipk_copy_cols": [_vp, _i, _i64, _vp, _i64, _vp, _i64, _i],
        "hipk_gather_cols": [_vp, _i, _i64, _vp, _i64, P(_i), _i, _vp, _i64],
        "hipk_col_norms2": [_vp, _i, _i64, _vp, _i64, _i, _vp],
        "hipk_residual_cols": [_vp, _i, _i64, _vp, _i64, _vp, _i64, _i, _dp, _vp],
        "hipk_csr_create": [_vp, _i, _i64, _i64, _i64, _vp, _vp, _vp, P(_vp)],
        "hipk_stencil_create": [_vp, _i, _i, _i, _i, _i64, _i64, P(_vp)],
        "hipk_csr_destroy": [_vp], "hipk_csr_matvec": [_vp, _vp, _vp, _i64, _vp, _i64, _i],
        "hipk_csr_set_halo": [_vp, _vp, _vp],
        "hipk_csr_set_halo_ld": [_vp, _vp, _i64, _vp, _i64],
        "hipk_triple_dots": [_vp, _i, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i, _vp],
        "hipk_project_triple_dots": [_vp, _i, _i64, P(HipkSeg), _i, _vp, _i, _vp, _i64, _i, _vp, _i64, _vp, _i64, _vp],
        "hipk_axpy_proj_dot": [_vp, _i, _i64, _i, _dp, _dp, _vp, _i64, _vp, _i64, _vp, _i64, _vp],
        "hipk_csr_matvec_shifted": [_vp, _vp, _vp, _i64, _vp, _i64, _i, _dp],
        "hipk_axpy_proj_dot_jacobi": [_vp, _i, _i64, _i, _dp, _dp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _dp, C.c_double, _vp],
        "hipk_qmr_update_dir": [_vp, _i, _i64, _i, _dp, _dp, _dp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _dp, C.c_double, _vp],
        "hipk_axpy_proj_dot_jacobi_dev": [_vp, _i, _i64, _i, _vp, _dp, C.c_double, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _dp, C.c_double, _vp],
        "hipk_qmr_update_dir_dev": [_vp, _i, _i64, _i, _vp, _vp, _dp, _dp, _dp, C.c_double, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _dp,
                                    C.c_double, _vp],
        "hipk_qmr_update_jacobi": [_vp, _i, _i64, _i, _dp, _dp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _dp, C.c_double, _vp, _i64, _vp],
        "hipk_csr_matvec_scaled": [_vp, _vp, _vp, _vp, _vp, _vp, _vp],
        "hipk_panel_project_mul": [_vp, _i, _i64, P(HipkSeg), _i, _vp, _i, _vp, _vp, _i64, _i],
        "hipk_panel_project_to": [_vp, _i, _i64, P(HipkSeg), _i, _vp, _i, _vp, _i64, _vp, _i64, _i, _vp],
        "hipk_jacobi_apply": [_vp, _i, _i64, _vp, _dp, C.c_double, _vp, _i64, _vp, _i64, _i],
        "primme_amd_operator_set_jacobi": [_vp, _i, C.c_double],
        "hipk_bandwidth_probe": [_vp, C.c_size_t, _i, _dp],
        "hipk_prof_enable": [_i], "hipk_prof_get": [_i, _dp, P(C.c_long), _dp],
        "primme_amd_operator_create": [P(_vp), _vp, _vp], "primme_amd_operator_destroy": [_vp],
        "primme_amd_svds_operator_create": [P(_vp), _vp, _i, _i64, _i64, _vp, _vp, _vp],
        "primme_amd_svds_operator_destroy": [_vp],
        "primme_amd_svds_operator_set_jacobi": [_vp, _vp, _vp, _vp, C.c_double],
        "hipk_csr_create_rect": [_vp, _i, _i64, _i64, _vp, _vp, _vp, P(_vp)],
        "primme_amd_mm_read": [C.c_char_p, P(_i64), P(_i64), P(_i64), P(_vp), P(_vp), P(_vp), P(_i)],
        "primme_amd_csr_transpose": [_i64, _i64, _vp, _vp, _vp, C.c_size_t, P(_vp), P(_vp), P(_vp)],
        "primme_amd_csr_tile_block_diagonal": [_i64, _vp, _vp, _vp, _i64, _i64, C.c_double, C.c_double, P(_vp), P(_vp), P(_vp)],
        "primme_amd_operator_apply": [_vp, _vp, _vp, _i64, _vp, _i64, _i],
        "primme_amd_csr_complex_to_real": [_i64, _vp, _vp, _vp, P(_vp), P(_vp), P(_vp)],
        "primme_amd_operator_set_complex": [_vp, _i],
        "hipk_pair_rotate": [_vp, _i, _i64, _vp, _i64, _vp, _i64, _i],
    }
    for name, args in sig.items():
        f = getattr(lib, name)
        f.argtypes = args
        f.restype = C.c_int
    lib.primme_amd_host_free.argtypes = [_vp]
    lib.primme_amd_host_free.restype = None
    for name in ("hipk_csr_diag", "hipk_ctx_stream"):
        getattr(lib, name).restype = _vp
        getattr(lib, name).argtypes = [_vp]
    for name in ("hipk_csr_nnz", "hipk_csr_halo_lo", "hipk_csr_halo_hi", "hipk_csr_nrows"):
        getattr(lib, name).restype = C.c_int64
        getattr(lib, name).argtypes = [_vp]
    if hasattr(lib, "primme_amd_comm_create"):
        lib.primme_amd_comm_unique_id.argtypes = [_vp]
        lib.primme_amd_comm_create.argtypes = [P(_vp), _vp, _i, _i]
        lib.primme_amd_comm_destroy.argtypes = [_vp]


_cache = {}


def load_product():
    """The MI355X library.  No fallback: a missing .so is an error."""
    if "product" not in _cache:
        if not os.path.exists(PRODUCT_LIB):
            raise RuntimeError(
                f"{PRODUCT_LIB} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). primme_amd has no CPU implementation.")
        # torch wheels ship their own libamdhip64 / librccl: load torch first so that this
        # process ends up with ONE HIP runtime (same SONAME -> the loader reuses torch's copy)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(PRODUCT_LIB)
        declare_solver(lib, "hip_")
        declare_kernels(lib)
        _cache["product"] = lib
    return _cache["product"]
