"""pytest configuration: `gpu` marker + shared paths/fixtures."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, 'tests', 'golden')
ASSET_DIR = os.path.join(REPO, 'assets')
REFERENCE_DIR = '/root/reference'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (authoring container only)')


def pytest_collection_modifyitems(config, items):
    have_reference = os.path.isdir(REFERENCE_DIR)
    skip_ref = pytest.mark.skip(reason='/root/reference not present')
    for item in items:
        if 'reference' in item.keywords and not have_reference:
            item.add_marker(skip_ref)


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False)

    return load
