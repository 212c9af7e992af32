"""Run one of the reference's interface cases (tests/reference_driver_cases.py) on a back end:
   python scripts/one_interface_case.py hip LOBPCG_OrthoBasis 100 100 smallest RR"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import reference_driver_cases as RD
from checkers import Operator, eigsh
from primme_amd import _ffi as F
be, method, n, nev, target, proj = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
print(RD.run_testi_case(eigsh, Operator, F.METHODS, be, method, n, nev, target, proj))
