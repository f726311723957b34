import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def _report_path():
    d = ROOT / "gpurun_out"
    d.mkdir(exist_ok=True)
    return d / "parity_report.txt"


@pytest.fixture(scope="session")
def report():
    """Append 'name err scale tol' lines to gpurun_out/parity_report.txt (read back after a gpurun call)."""
    path = _report_path()

    def _rep(name, err, scale=None, tol=None):
        with open(path, "a") as f:
            f.write(f"{name}\terr={err:.3e}\tscale={'' if scale is None else f'{scale:.3e}'}"
                    f"\ttol={'' if tol is None else f'{tol:.1e}'}\n")
    return _rep


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from adaface_amd import _lib
    _lib.load()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


@pytest.fixture
def knobs(gpu):
    """Set planner / kernel-variant knobs through the C ABI (af_knob_set) for one test; load-time values restored after."""
    from adaface_amd import _lib
    yield _lib.set_knob
    _lib.reset_knobs()
