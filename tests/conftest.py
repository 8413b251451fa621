import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


LAB = os.environ.get("VPU_LIB_DIAG", "0") == "1"     # this process loads the laboratory library (pvpuformer_amd/_lib.py)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "lab: kernel families compiled into the laboratory library only (libvpu_hip_diag.so, "
                                       "VPU_LIB_DIAG=1); test_lab_library_families runs them in a child process")


def pytest_collection_modifyitems(config, items):
    if LAB:
        return
    skip = pytest.mark.skip(reason="laboratory library only (VPU_LIB_DIAG=1): run by test_lab_library_families in a child process")
    for it in items:
        if "lab" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
