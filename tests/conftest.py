import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'st-p3_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)




@pytest.fixture(autouse=True)
def restore_thread_count():
    """Some tests pin torch's intra-op thread count; the oracle's bit-exact comparisons (float32 cumsum) depend on it,
    so it is put back after every test."""
    import torch
    n = torch.get_num_threads()
    yield
    if torch.get_num_threads() != n:
        torch.set_num_threads(n)
