import os
import sys

import pytest

os.environ.setdefault('NM_TESTING', '1')  # the library honours its test-only hooks (NM_ASSUME_CUS, NM_INJECT_*) only under this switch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle():
    """the CPU oracle (test infrastructure); built on demand with gcc"""
    from oracle import oracle as O
    O.build()
    O.lib()
    return O
