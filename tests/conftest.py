import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:  # helper modules next to the tests (lk_orders.py)
    sys.path.insert(0, HERE)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _has_gpu():
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        return False


@pytest.fixture(scope="session")
def has_gpu():
    return _has_gpu()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


def scene_batches(W, H, n_batches, rate, seed, **kw):
    from esvio_amd.synth import SceneStream
    s = SceneStream(W, H, rate=rate, seed=seed, **kw)
    return [s.next_batch() for _ in range(n_batches)]
