import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'sba-gan_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'statistical: compares two runs of the DEFAULT (atomic-order dependent) mode with '
                                       'fixed noise bounds; ordered after every oracle / golden / boundary test')


def pytest_collection_modifyitems(config, items):
    """Order: every test that compares with the oracle, the reference's golden vectors or the boundary contract
    first; the few `statistical` tests (run-to-run comparisons of the default, atomic-order dependent mode) last, so
    that `pytest -x` reaches every parity test before any noise-bounded one.
    GPU tests are skipped (not failed) when no device is visible, so a bare `pytest tests/` stays green in the
    CPU-only build container."""
    items.sort(key=lambda it: 1 if 'statistical' in it.keywords else 0)     # (stable: file order otherwise kept)
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
