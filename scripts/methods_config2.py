"""Time-to-solution of the preset methods on the BASELINE configs[1] problem (n = 2 000 250, 10 smallest)."""
import numpy as np, sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from primme_amd import problems
from checkers import Operator
from checkers import Session
dims = (125, 126, 127)
rp, ci, va, n = problems.laplacian_csr(dims)
s = Session(Operator(n, csr=(rp, ci, va)))
v0 = problems.start_vector(n)
ex = problems.laplacian_eigenvalues(dims, 10)
for method, kw in [("GD_plusK", {}), ("JDQMR", {}), ("JDQMR_ETol", {}), ("GD_plusK", dict(maxBlockSize=4)), ("JDQMR", dict(maxBlockSize=4)),
                   ("LOBPCG_OrthoBasis", dict(maxBlockSize=10))]:
    for rep in range(2):
        r = s.solve(numEvals=10, eps=1e-8, aNorm=12.0, v0=v0, method=method, return_evecs=False, **kw)
    print(json.dumps(dict(method=method, **kw, ret=r.ret, seconds=round(r.stats["elapsedTime"], 3), outer=r.stats["numOuterIterations"],
                          matvecs=r.stats["numMatvecs"], err=float(np.max(np.abs(np.sort(r.evals) - ex))))), flush=True)
