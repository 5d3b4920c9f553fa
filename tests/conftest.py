import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# libmdt_hip.so's test hooks (mdt_test_occupy, the "pair_capacity" override) act only in a process that asks for them
os.environ.setdefault("MDT_TEST_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the rest of a parametrised sweep whose covering subset runs by default; "
                                       "`-m \"gpu and slow\"` runs it (round 6: the default GPU suite must stay well under the "
                                       "driver's 1,200 s)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` (the driver's command) runs the covering subset; tests marked `slow` run only when the -m expression names them."""
    if "slow" in (config.getoption("-m") or ""):
        return
    drop = [it for it in items if it.get_closest_marker("slow")]
    if drop:
        items[:] = [it for it in items if not it.get_closest_marker("slow")]
        config.hook.pytest_deselected(items=drop)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden
