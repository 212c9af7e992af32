"""Config-2 solve with the host QL Rayleigh-Ritz (default) and with the device Jacobi kernel
(PRIMME_AMD_DEVICE_RR=1): what the north star's 'solve_H as a HIP kernel' costs per outer iteration."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys, json, numpy as np
sys.path.insert(0, %r)
from primme_amd import problems
from checkers import Operator
from checkers import Session
dims=(125,126,127); rp,ci,va,n=problems.laplacian_csr(dims)
s=Session(Operator(n,csr=(rp,ci,va))); v0=problems.start_vector(n)
for rep in range(2): r=s.solve(numEvals=10,eps=1e-8,aNorm=12.0,v0=v0,return_evecs=False)
print(json.dumps(dict(ret=r.ret, seconds=r.stats["elapsedTime"], outer=r.stats["numOuterIterations"], us_per_outer=1e6*r.stats["elapsedTime"]/r.stats["numOuterIterations"])))
''' % ROOT
for env in ({}, {"PRIMME_AMD_DEVICE_RR": "1"}):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, "-c", code], env=e, stdout=subprocess.PIPE, text=True).stdout.strip().splitlines()[-1]
    print("device_rr" if env else "host_rr  ", out, flush=True)
