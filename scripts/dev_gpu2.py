import numpy as np, sys, time, faulthandler
faulthandler.dump_traceback_later(60, exit=True)
sys.path.insert(0,'/root/repo')
from primme_amd import eigsh, Operator, problems
dims=(20,21)
rp,ci,va,n = problems.laplacian_csr(dims)
op = Operator(n, csr=(rp,ci,va)); v0 = problems.start_vector(n)
for mm in (5, 30, 200, 100000):
    t=time.time()
    r = eigsh(op, numEvals=10, eps=1e-10, aNorm=8.0, v0=v0, backend="hip", maxMatvecs=mm)
    print("hip", mm, r.ret, r.initSize, r.stats["numOuterIterations"], r.stats["numMatvecs"], r.stats["numRestarts"], r.resNorms.max(), time.time()-t, flush=True)
