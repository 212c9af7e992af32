"""primme_amd — MI355X-native Davidson/GD+k eigensolver path behind PRIMME's primme_params ABI.

The product is the C/HIP shared library primme_amd/libprimme_amd.so (include/*.h);
this package is ctypes plumbing for tests, smoke() and bench.py.
"""
from .api import eigsh, Operator, Result  # noqa: F401
from . import problems  # noqa: F401
