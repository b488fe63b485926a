import os

# HIP runtime switches the split-graph executor needs (bist_amd/__init__.py), set before anything can initialise the runtime
if os.environ.get("BIST_SPLIT_GRAPH", "1") != "0":
    os.environ.setdefault("DEBUG_HIP_DYNAMIC_QUEUES", "0")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "1")
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
