import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "pointnet2_golden.npz")
    return np.load(path)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    # The HIP extension must be the thing under test: fail loudly if absent.
    from adaptpoint_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def golden_ap():
    """G9-G13: the AdaptPoint half (generator, discriminator, the two training steps)."""
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "adaptpoint_golden.npz"))


@pytest.fixture
def cpu_mirrors(monkeypatch):
    """The product's host-side modules on CPU tensors: the C oracle stands in for the extension
    underneath `adaptpoint_amd.ops` (oracle/cpu_block.py) and the reference's composition for the
    attention core, which otherwise refuses CPU tensors.  Modules must be built with fused=False."""
    from oracle import cpu_block as CB
    from adaptpoint_amd import attention as A
    monkeypatch.setattr(A, "attention", A._reference)
    with CB.CpuOps():
        yield


@pytest.fixture(scope="session")
def golden_b8():
    """G17 / G18: the reference classifier at B = 8 in training mode, per-parameter gradients (make_golden.py classifier)."""
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "classifier_b8_golden.npz"))
