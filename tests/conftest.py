"""pytest configuration: registers the ``gpu`` marker and puts the repo root and the product
package directory on sys.path (the package directory name contains '-', so its modules are imported
as top-level ``mvn`` -- the same import names the reference uses -- and ``lt_hip``)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "learnable-triangulation-pytorch_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The golden fixtures were generated with 8 CPU threads; oneDNN's reduction order (hence the fp32 rounding noise of the
# reference itself: 1e-4 relative on the sharpened joints between 1 and 8 threads) depends on the thread count.
try:
    import torch
    torch.set_num_threads(min(8, os.cpu_count() or 1))
except Exception:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    try:
        import gpu_util
        if gpu_util.REPORT:
            gpu_util.flush_report()
    except Exception:
        pass
