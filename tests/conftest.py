import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def ensure_library_is_built():
    """The in-tree C-ABI library is (re)built when missing or older than its sources (hipcc cross-compiles without a GPU;
    a no-op otherwise).  The product itself never builds on demand: a missing library makes every entry point raise."""
    from flash_attention_annotated_amd import _lib
    _lib.build()


@pytest.fixture(scope="session")
def golden():
    import torch
    path = os.path.join(ROOT, "tests", "golden", "attention_ref_golden.pt")
    return torch.load(path, weights_only=True)


@pytest.fixture(scope="session")
def built_lib():
    """Build (if stale) and load the C-ABI library.  hipcc cross-compiles without a GPU."""
    from flash_attention_annotated_amd import _lib
    _lib.build()
    return _lib.load()


@pytest.fixture(scope="session")
def golden_grads():
    """Gradients frozen from the reference's oracle + autograd by oracle/make_golden.py (tensors only)."""
    import torch
    return torch.load(os.path.join(ROOT, "tests", "golden", "attention_grad_golden.pt"), weights_only=True)
