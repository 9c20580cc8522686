import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_golden.npz'))


@pytest.fixture(scope='session')
def golden_heads():
    """outputs of the reference's own model.LSTM / model.Residual / sampler.scoring (tests/golden/make_golden.py: heads_fixture)"""
    import numpy as np
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_golden_heads.npz'))


@pytest.fixture(scope='session')
def gpu():
    """cuda:0 on a box where libse_amd.so is built and a gfx950 is visible; fails (never skips) otherwise,
    so a silent fallback cannot make GPU tests pass."""
    import torch
    from speech_enhancement_by_s3prl_amd import _lib
    lib = _lib.load()
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    assert lib.se_device_available() == 1, 'libse_amd.so does not see a gfx950 device'
    return torch.device('cuda:0')


def bounded(name, value, bound):
    """assert value < bound, and log the measured value next to its bound (gpurun_out/parity_measured.txt on the GPU box): the bf16 bounds of
    the parity tests are kept at ~2x what is measured, so that a precision regression (e.g. a bf16 residual stream: x2 error) fails."""
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'parity_measured.txt'), 'a') as fh:
            fh.write(f'{name}\t{value:.4e}\t{bound:.1e}\n')
    except OSError:
        pass
    assert value < bound, (name, value, bound)
