import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))   # checkers.py: test infrastructure


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native pieces are built once per session (no-op when up to date)."""
    import __graft_entry__ as g
    g.build()
    return True


# Order of the test MODULES (the driver runs `pytest -x`): parity against what the reference holds comes first —
# its own panel routines (fixture F6), its solves and residual norms, its singular-value cases, the full-size
# BASELINE configs — then the C-ABI examples and the rest; kernel-vs-oracle sweeps last.  Within a module the
# order is untouched.
_MODULE_ORDER = ["test_reference_kernels", "test_solver_gpu", "test_svds_gpu", "test_full_size_configs_gpu", "test_complex_gpu",
                 "test_interface_cases_gpu", "test_c_examples_gpu", "test_multigpu_rccl", "test_comm_gpu", "test_kernels_gpu"]


def pytest_collection_modifyitems(config, items):
    rank = {name: i for i, name in enumerate(_MODULE_ORDER)}

    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return rank.get(mod, len(rank))
    items.sort(key=key)          # stable: collection order inside a module is kept
