import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("COVA_ALLOW_OPTION_CHANGES", "1")     # the kernel tests compare kernel forms inside one process
import cova_amd  # noqa: E402,F401  registers cova_web_object_detection_amd

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _f2x2_support_library():
    """GPU runs: the F(2x2,3x3) Winograd kernels of rounds 1-3 -- a TEST-SUPPORT library since round 6 (tools/csrc/conv_wino_f2x2.hip,
    built by __graft_entry__.build()), not part of the product -- are registered beside the product library's entry points so
    that the F(4x4) kernel tests can cross-check against them."""
    import torch
    if torch.cuda.is_available():
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import f2x2_lib
        f2x2_lib.load()
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
