import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import torch
from primme_amd import problems, Operator
from primme_amd.api import Session
rp, ci, va, n = problems.laplacian_csr((125,126,127))
s = Session(Operator(n, csr=(rp, ci, va)))
v0 = problems.start_vector(n)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = s.solve(numEvals=10, eps=1e-8, aNorm=12.0, v0=v0, maxOuterIterations=3, return_evecs=False)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("solve with 3 outer iterations: %.2f ms wall, solver elapsedTime %.2f ms, ret %d its %d" % (1e3*(t1-t0), 1e3*r.stats["elapsedTime"], r.ret, r.stats["numOuterIterations"]))
s.close()
