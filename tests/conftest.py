import os
import sys
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

warnings.filterwarnings("ignore", message=".*padding='same' with even kernel lengths.*")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device (no CPU fallback exists)")
    return torch.device("cuda:0")


def act_err_ok(act, ref, tol=1e-4):
    """Elementwise bar for per-kernel activations on binary occupancy (the int8 kernels: exact integer sums, the only error
    is the weights' 24-bit quantisation, bounded ABSOLUTELY per kernel by the device-side guard): |err| <= tol where
    |ref| <= 1, tol * |ref| above (fp32 recombination / storage rounding scales with the value).  VERDICT r3: the former bar,
    tol * max(1, max|ref|) over the whole tensor, let a small activation err by the largest one's allowance."""
    import torch
    err = (act.detach().cpu().double() - ref.double()).abs()
    return bool((err <= tol * ref.double().abs().clamp_min(1.0)).all())
