import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# The GPU suite runs with -x: the headline path (BPR step, eval, metrics, class surface, data-parallel, LOO) goes
# first so that a failure in a side model cannot hide it.
_ORDER = ["test_gpu_parity.py", "test_gpu_sampler.py", "test_gpu_big.py", "test_gpu_ncf.py", "test_gpu_fm.py", "test_gpu_cml.py"]


def pytest_collection_modifyitems(config, items):
    def key(item):
        name = os.path.basename(str(item.fspath))
        return _ORDER.index(name) if name in _ORDER else len(_ORDER)
    items.sort(key=key)  # stable: file order inside a module is kept
