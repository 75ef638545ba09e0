import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    """Vectors produced by the reference's own Python (tests/golden/make_golden.py)."""
    path = os.path.join(REPO, 'tests', 'golden', 'reference_vectors.npz')
    z = np.load(path)
    return {k: z[k] for k in z.files}


@pytest.fixture(autouse=True)
def _gpu_guard(request):
    """Every test marked `gpu` needs a device: skipped on machines without an AMD GPU driver
    node, a failure on the GPU box (tests/gpu_helpers.require_gpu)."""
    if request.node.get_closest_marker('gpu') is not None:
        from tests.gpu_helpers import require_gpu
        require_gpu()
