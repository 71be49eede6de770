import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: wall-clock throughput floors on a real MI355X; only run when asked for by name "
                                       "(pytest -m perf) -- never part of the -m gpu parity run or of the CPU suite")


def pytest_collection_modifyitems(config, items):
    """`perf` tests read a clock on a GPU: they are deselected unless the marker expression names them, so neither
    `-m gpu` (parity, run with -x by the driver) nor `-m "not gpu"` (CPU) ever contains a wall-clock assertion."""
    if "perf" in (config.getoption("-m") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("perf") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "ni_closed_form.json")) as f:
        return json.load(f)["cases"]
