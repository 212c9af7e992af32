"""One solve for profiling: BASELINE configs[1] (3-D 7-pt Laplacian 125x126x127, default) or the north-star workload
(`python scripts/one_solve.py csr lap2d_10m [max_outer]`: 2-D 5-pt Laplacian 3162x3163, optionally cut after
max_outer outer iterations — per-launch kernel statistics and PMC traffic do not need the whole 30 000-iteration solve)."""
import numpy as np, sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from primme_amd import problems
from checkers import eigsh, Operator
kind = sys.argv[1] if len(sys.argv) > 1 else "csr"
wl = sys.argv[2] if len(sys.argv) > 2 else "lap3d_2m"
max_outer = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dims, aNorm = ((125, 126, 127), 12.0) if wl == "lap3d_2m" else ((3162, 3163), 8.0)
n = int(np.prod(dims))
if kind == "csr":
    rp, ci, va, n = problems.laplacian_csr(dims); op = Operator(n, csr=(rp, ci, va))
else:
    op = Operator(n, stencil=dims)
v0 = problems.start_vector(n)
reps = int(os.environ.get("REPS", "1"))      # REPS=n: n solves through one session (matrix and panels stay resident), every solver time printed
if reps > 1:
    from checkers import Session
    sess = Session(op, backend="hip")
    ts = []
    for _ in range(reps):
        r = sess.solve(numEvals=10, eps=1e-8, aNorm=aNorm, v0=v0, return_evecs=False, maxOuterIterations=max_outer)
        ts.append(r.stats["elapsedTime"])
    sess.close()
    its = r.stats["numOuterIterations"]
    print(kind, wl, "ret", r.ret, "its", its, "t", " ".join(f"{t:.4f}" for t in ts), "us/iteration (fastest)", round(1e6 * min(ts) / its, 2))
else:
    r = eigsh(op, numEvals=10, eps=1e-8, aNorm=aNorm, v0=v0, backend="hip", return_evecs=False, maxOuterIterations=max_outer)
    print(kind, wl, "ret", r.ret, "its", r.stats["numOuterIterations"], "t", r.stats["elapsedTime"])
