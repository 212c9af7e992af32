"""Block JDQMR with the reference's own indexing (PRIMME_AMD_JDQMR_REF_INDEXING=1, csrc/eigs_jd.c) against the LIVE reference
(oracle/_ref) on the CPU checker: JDQMR / JDQMR_ETol x block sizes 2, 4, 8 x Jacobi preconditioner or none x a 3-D and a 2-D
Laplacian; outer iterations / matvecs / restarts of both.  Build container only:
    python scripts/jdqmr_ref_indexing_sweep.py > profiles/r06_jdqmr_reference_indexing.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["PRIMME_AMD_JDQMR_REF_INDEXING"] = "1"
from primme_amd import problems
from checkers import eigsh, Operator
same = total = 0
print("# PRIMME_AMD_JDQMR_REF_INDEXING=1: (outer iterations, matvecs, restarts) of the live reference and of the CPU checker")
for dims in ((20, 21, 22), (30, 31)):
    rp, ci, va, n = problems.laplacian_csr(dims)
    op = Operator(n, csr=(rp, ci, va))
    for method in ("JDQMR", "JDQMR_ETol"):
        for b in (2, 4, 8):
            for pc in (None, "jacobi"):
                kw = dict(numEvals=10 if len(dims) == 3 else 6, method=method, eps=1e-9, aNorm=12.0 if len(dims) == 3 else 8.0, maxBlockSize=b, v0=problems.start_vector(n))
                if pc: kw["precond"] = pc
                a = eigsh(op, backend="reference", **kw); c = eigsh(op, backend="hostcheck", **kw)
                f = lambda r: (r.stats["numOuterIterations"], r.stats["numMatvecs"], r.stats["numRestarts"])
                total += 1; same += f(a) == f(c)
                print(f"{method:11s} b={b} precond={str(pc):6s} {str(dims):13s} reference {f(a)}  here {f(c)}  {'EXACT' if f(a) == f(c) else 'matvecs %+.1f %%' % (100.0 * (f(c)[1] - f(a)[1]) / f(a)[1])}", flush=True)
print(f"# {same} of {total} configurations count-exact")
