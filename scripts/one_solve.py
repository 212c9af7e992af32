"""One config-2 solve (3-D 7-pt Laplacian 125x126x127, 10 smallest, GD+k) for profiling."""
import numpy as np, sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from primme_amd import problems
from checkers import eigsh, Operator
kind = sys.argv[1] if len(sys.argv) > 1 else "csr"
dims = (125, 126, 127)
n = int(np.prod(dims))
if kind == "csr":
    rp, ci, va, n = problems.laplacian_csr(dims); op = Operator(n, csr=(rp, ci, va))
else:
    op = Operator(n, stencil=dims)
v0 = problems.start_vector(n)
r = eigsh(op, numEvals=10, eps=1e-8, aNorm=12.0, v0=v0, backend="hip", return_evecs=False)
print(kind, "ret", r.ret, "its", r.stats["numOuterIterations"], "t", r.stats["elapsedTime"])
