import os
import sys

import pytest

try:  # torch bundles its own HIP runtime: it must be loaded before libbposd_mi355x.so pulls in the
    import torch  # noqa: F401  system one, or torch later reports "No HIP GPUs are available"
except Exception:  # pragma: no cover - torch is plumbing for two tests only
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the parity suites' run time is the CPU oracle's: let its batch decodes use the host's cores (one handle per thread;
    # bench.py's one-core baseline does not go through here)
    import oracle.oracle as _oracle

    _oracle.DEFAULT_THREADS = max(1, min(8, (os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def surface13():
    from bp_osd_amd.codes import surface13 as f

    return f()


@pytest.fixture(scope="session")
def h1922():
    from bp_osd_amd.codes import h1922 as f

    return f()


@pytest.fixture(scope="session")
def hgp400():
    import numpy as np
    from bp_osd_amd.codes import hgp

    seed = np.loadtxt(os.path.join(ROOT, "tests", "golden", "mkmn_16_4_6.txt")).astype(np.uint8)
    return hgp(seed)
