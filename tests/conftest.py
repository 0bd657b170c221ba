import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: large CPU case, opt in with EXAMG_SLOW=1")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("EXAMG_SLOW", "0") == "1":
        return
    skip = pytest.mark.skip(reason="set EXAMG_SLOW=1 to run")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)
