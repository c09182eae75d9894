import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with `-m gpu` on the GPU box)")
    # The GPU box's host has 256 hardware threads; torch's intra-op pool on 256 threads makes the SMALL CPU ops of the oracle (VP block on 50
    # tokens, 64-channel norms) 20-50x SLOWER than on 8 - the round-3 suite spent ~600 of its 750-1070 s there (profiles/r04_gpu_test_durations_before.txt).
    import torch

    torch.set_num_threads(min(16, os.cpu_count() or 1))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
