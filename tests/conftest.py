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


@pytest.fixture(autouse=True)
def _switches_follow_monkeypatch(monkeypatch):
    """libstx reads its STX_* switches from a snapshot of the environment (stx_reread_env), not with
    getenv() at every launch: a test that sets or deletes one through `monkeypatch` gets a fresh snapshot
    with it, and another one when the environment is restored at the end of the test."""
    from style_transfer_amd import lib
    setenv, delenv = monkeypatch.setenv, monkeypatch.delenv
    touched = []

    def setenv_and_reread(name, value, *args, **kwargs):
        setenv(name, value, *args, **kwargs)
        if name.startswith('STX_'):
            touched.append(name)
            lib.reread_env()

    def delenv_and_reread(name, *args, **kwargs):
        delenv(name, *args, **kwargs)
        if name.startswith('STX_'):
            touched.append(name)
            lib.reread_env()
    monkeypatch.setenv, monkeypatch.delenv = setenv_and_reread, delenv_and_reread
    yield
    if touched:
        monkeypatch.undo()
        lib.reread_env()
