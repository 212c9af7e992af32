"""CPU baseline leg of bench.py: times the REAL reference (oracle/_ref/libprimme_ref.so,
PRIMME 3.2 built from /root/reference by oracle/Makefile, BLAS/LAPACK = the image's MKL)
on the host cores, on a bounded sample of the bench workload.

Run as a subprocess (`python oracle/cpu_baseline.py ...`) so that the thread
settings are in the environment before MKL loads.  Prints one JSON object.
This module is measurement/test infrastructure under oracle/: the product package cannot import it.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, nargs="+", default=[125, 126, 127])
    ap.add_argument("--num-evals", type=int, default=10)
    ap.add_argument("--eps", type=float, default=1e-8)
    ap.add_argument("--anorm", type=float, default=12.0)
    ap.add_argument("--max-matvecs", type=int, default=150)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--total-iterations", type=int, default=0,
                    help="outer iterations the full solve needs (from the GPU run) for extrapolation")
    a = ap.parse_args()
    # 16 threads was the fastest of {1,4,16,32,128} on the 2x64-core EPYC 9575F GPU-box host
    # (profiles/r01_cpu_thread_scan.log): MKL's tall-skinny BLAS-2 does not scale beyond that
    threads = a.threads or min(os.cpu_count() or 1, 16)
    for k in ("MKL_NUM_THREADS", "OMP_NUM_THREADS"):
        os.environ[k] = str(threads)
    os.environ.setdefault("MKL_DYNAMIC", "FALSE")

    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import checkers
    from primme_amd import _ffi as F
    from primme_amd import problems

    ref = checkers.load_reference()
    cb = C.CDLL(os.path.join(os.path.dirname(checkers.HOSTCHECK_LIB), "librefcb.so"))
    rp, ci, va, n = problems.laplacian_csr(tuple(a.dims))

    class RefCsr(C.Structure):
        _fields_ = [("n", C.c_int64), ("rowptr", C.c_void_p), ("colind", C.c_void_p), ("values", C.c_void_p)]

    A = RefCsr(n, rp.ctypes.data, ci.ctypes.data, va.ctypes.data)
    p = F.PrimmeParams()
    ref.primme_initialize(C.byref(p))
    p.n = n
    p.numEvals = a.num_evals
    p.eps = a.eps
    p.aNorm = a.anorm
    p.printLevel = 0
    p.matrix = C.addressof(A)
    p.matrixMatvec = C.cast(cb.ref_csr_matvec, C.c_void_p)
    p.initSize = 1
    p.initBasisMode = F.primme_init_user
    p.maxMatvecs = a.max_matvecs
    ref.primme_set_method(F.PRIMME_GD_plusK, C.byref(p))
    evecs = np.zeros((a.num_evals, n))
    evecs[0] = problems.start_vector(n)
    evals = np.zeros(a.num_evals)
    rn = np.zeros(a.num_evals)
    t0 = time.time()
    ret = ref.dprimme(evals.ctypes.data_as(C.c_void_p), evecs.ctypes.data_as(C.c_void_p),
                      rn.ctypes.data_as(C.c_void_p), C.byref(p))
    wall = time.time() - t0
    its = int(p.stats.numOuterIterations)
    sec_per_it = p.stats.elapsedTime / max(its, 1)
    out = {
        "kind": "reference", "cores": threads, "ret": ret, "sample_outer_iterations": its,
        "sample_seconds": p.stats.elapsedTime, "wall_seconds": wall, "seconds_per_outer_iteration": sec_per_it,
        "timeMatvec": p.stats.timeMatvec, "timeOrtho": p.stats.timeOrtho, "timeDense": p.stats.timeDense,
    }
    if a.total_iterations > 0:
        out["value"] = a.num_evals / (sec_per_it * a.total_iterations)
        out["unit"] = "eigenpairs/s"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
