import numpy as np, sys, time, faulthandler
faulthandler.dump_traceback_later(150, exit=True)
sys.path.insert(0,'/root/repo')
from primme_amd import eigsh, Operator, problems
import __graft_entry__ as g
t=time.time(); g.smoke(); print("smoke", time.time()-t, flush=True)
for dims in [(60,61,62),(125,126,127)]:
    n=int(np.prod(dims))
    for kind in ("stencil","csr"):
        if kind=="csr":
            rp,ci,va,n = problems.laplacian_csr(dims); op=Operator(n,csr=(rp,ci,va))
        else:
            op=Operator(n, stencil=dims)
        v0=problems.start_vector(n)
        for mm in (200, 10**8):
            t=time.time()
            r = eigsh(op, numEvals=10, eps=1e-8, aNorm=12.0, v0=v0, backend="hip", return_evecs=False, maxMatvecs=mm)
            print(kind, dims, "ret",r.ret,"conv",r.initSize,"its",r.stats["numOuterIterations"],"mv",r.stats["numMatvecs"],"rst",r.stats["numRestarts"],"t %.3f"%r.stats["elapsedTime"], "wall %.3f"%(time.time()-t), "us/it %.1f"%(1e6*r.stats["elapsedTime"]/max(1,r.stats["numOuterIterations"])), flush=True)
        ex=problems.laplacian_eigenvalues(dims,10)
        print(" err", np.abs(r.evals-ex).max(), "rn", r.resNorms.max(), flush=True)
