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
    """In the build container (no GPU) the in-tree C-ABI library is (re)built when it does not match the sources (hipcc
    cross-compiles).  On a GPU box nothing is ever built: a library that was not built from the sources in the tree fails
    the session loudly (content hash recorded by build(); the product itself never builds on demand either)."""
    import torch
    from flash_attention_annotated_amd import _lib
    if os.environ.get("FA_FWD_LIB"):
        # the developer override (ablation / instrumented builds) would silently swap the library under test: the staleness
        # check below only covers the in-tree path
        pytest.exit("FA_FWD_LIB is set: the test suite only runs against the in-tree libfa_fwd_gfx950.so", returncode=3)
    if torch.cuda.is_available():
        if _lib.is_stale():
            pytest.exit("libfa_fwd_gfx950.so is missing or stale on this GPU box: run `python -c 'import __graft_entry__ as g; "
                        "g.build()'` in the build container before sending the tree", returncode=3)
        return
    _lib.build()


@pytest.fixture(scope="session")
def golden():
    import torch
    path = os.path.join(ROOT, "tests", "golden", "attention_ref_golden.pt")
    return torch.load(path, weights_only=True)


@pytest.fixture(scope="session")
def golden_fa3():
    """Outputs of the reference's FA3 oracle (hopper/test_util.py:226-348) for the `attention_chunk` / `dv` cases of
    oracle/cases.py:FA3_CASES, frozen by oracle/make_golden.py (tensors only)."""
    import torch
    return torch.load(os.path.join(ROOT, "tests", "golden", "attention_fa3_golden.pt"), weights_only=True)


@pytest.fixture(scope="session")
def built_lib():
    """Build (if stale) and load the C-ABI library.  hipcc cross-compiles without a GPU."""
    import torch
    from flash_attention_annotated_amd import _lib
    if not torch.cuda.is_available():
        _lib.build()
    return _lib.load()


@pytest.fixture(scope="session")
def golden_grads():
    """Gradients frozen from the reference's oracle + autograd by oracle/make_golden.py (tensors only)."""
    import torch
    return torch.load(os.path.join(ROOT, "tests", "golden", "attention_grad_golden.pt"), weights_only=True)
