import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle / mirror legs of the tests: torch's default (one thread per logical CPU, 128-256 on the GPU box's host) is
    # 3.5-20x SLOWER than 16 threads on these small convs (measured in round 5: a 64 px oracle train step 1.50 s at 128
    # threads, 0.069 s at 16); two thirds of the GPU suite's wall time was this oversubscription.
    import torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


def pytest_collection_modifyitems(config, items):
    """GPU tests skip themselves (not fail) when collected on a box without a GPU and no -m filter."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
