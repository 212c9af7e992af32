"""SpMV/SpMM timing of the CSR kernels on the two matrix families of BASELINE configs 2 and 3.
   python scripts/spmm_perf.py   (GPU)"""
import ctypes as C, os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from primme_amd import problems, _ffi as F
from checkers import Operator, Session
import reference_driver_cases as RD

def run(name, rp, ci, va, n):
    op = Operator(n, csr=(rp, ci, va))
    sess = Session(op, backend="hip")
    lib = sess.lib
    A = [h for k, h in sess.handles if k == "csr"][0]
    nnz = len(va)
    ts = torch.cuda.Stream()
    torch.cuda.set_stream(ts)
    for nc in (1, 2, 4, 8):
        x = torch.randn((nc, n), dtype=torch.float64, device="cuda")
        y = torch.zeros_like(x)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            lib.hipk_csr_matvec(A, C.c_void_p(st), C.c_void_p(x.data_ptr()), n, C.c_void_p(y.data_ptr()), n, nc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            lib.hipk_csr_matvec(A, C.c_void_p(st), C.c_void_p(x.data_ptr()), n, C.c_void_p(y.data_ptr()), n, nc)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        alg = nnz * 12 + (n + 1) * 4 + 2 * n * 8 * nc
        print(json.dumps(dict(matrix=name, n=n, nnz=nnz, ncols=nc, us=round(ms * 1e3, 1), us_per_col=round(ms * 1e3 / nc, 1),
                              alg_GBps=round(alg / ms / 1e6, 1))), flush=True)
    sess.close()

rp, ci, va, n0 = RD.lunda()
T = 34014
trp, tci, tva = problems.tile_block_diagonal(rp, ci, va, T, lambda t: 1.0 + t / T)
run("lunda_tiled", trp, tci, tva, n0 * T)
rp, ci, va, n = problems.laplacian_csr((125, 126, 127))
run("lap3d_2m", rp, ci, va, n)
