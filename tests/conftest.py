import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle is torch-CPU / numpy.  On the GPU box (256 hardware threads) torch's default intra-op pool makes its small matrix products
    # crawl (profiles/r03_bench_line_first.json: one 576-ray pass 860 s with 256 threads, 0.5 s with 8): bound it once for the whole session.
    try:
        import torch
        torch.set_num_threads(min(8, os.cpu_count() or 8))
    except ImportError:
        pass


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load
