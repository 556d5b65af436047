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
_ORDER = ["test_gpu_parity.py", "test_gpu_sharded.py", "test_gpu_sampler.py", "test_gpu_big.py", "test_gpu_ncf.py", "test_gpu_fm.py", "test_gpu_cml.py", "test_gpu_sbpr.py", "test_gpu_lrml.py", "test_gpu_transcf.py"]


def pytest_collection_modifyitems(config, items):
    def key(item):
        name = os.path.basename(str(item.fspath))
        return _ORDER.index(name) if name in _ORDER else len(_ORDER)
    items.sort(key=key)  # stable: file order inside a module is kept


def pytest_sessionfinish(session, exitstatus):
    """How much of the rounding envelope the GPU comparisons used (tests/_adam_close.py): kept as evidence under gpurun_out/."""
    mod = sys.modules.get("_adam_close")
    out = os.path.join(ROOT, "gpurun_out")
    if mod is None or not mod.MARGINS or not os.path.isdir(out):
        return
    import json
    worst = {}
    for name, what, ratio, dmax in mod.MARGINS:
        k = what
        if ratio > worst.get(k, (0, 0, ""))[0]:
            worst[k] = (ratio, dmax, name)
    with open(os.path.join(out, "adam_margins.json"), "w") as f:
        json.dump({"comparisons": len(mod.MARGINS), "worst_ratio_of_bound": worst}, f)
