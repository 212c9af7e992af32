import numpy as np, sys, time
sys.path.insert(0,'/root/repo')
from primme_amd import eigsh, Operator, problems
import __graft_entry__ as g
g.smoke()
dims=(20,21)
rp,ci,va,n = problems.laplacian_csr(dims)
op = Operator(n, csr=(rp,ci,va)); v0 = problems.start_vector(n)
for be in ("reference","hostcheck","hip"):
    r = eigsh(op, numEvals=10, eps=1e-10, aNorm=8.0, v0=v0, backend=be)
    print(be, r.ret, r.initSize, r.stats["numOuterIterations"], r.stats["numMatvecs"], r.stats["numRestarts"], r.resNorms.max(), r.stats["elapsedTime"])
for dims in [(125,126,127)]:
    n=int(np.prod(dims))
    for kind in ("csr","stencil"):
        if kind=="csr":
            rp,ci,va,n = problems.laplacian_csr(dims); op=Operator(n,csr=(rp,ci,va))
        else:
            op=Operator(n, stencil=dims)
        v0=problems.start_vector(n)
        for rep in range(2):
            t=time.time()
            r = eigsh(op, numEvals=10, eps=1e-8, aNorm=12.0, v0=v0, backend="hip", return_evecs=False)
            print(kind, dims, "ret",r.ret,"conv",r.initSize,"its",r.stats["numOuterIterations"],"mv",r.stats["numMatvecs"],"rst",r.stats["numRestarts"],"t",r.stats["elapsedTime"], "wall",time.time()-t, "ms/it", 1e3*r.stats["elapsedTime"]/r.stats["numOuterIterations"])
        ex=problems.laplacian_eigenvalues(dims,10)
        print(" err", np.abs(r.evals-ex).max(), "rn", r.resNorms.max())
