import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session', autouse=True)
def oracle_c_port():
    """The oracle's C / OpenMP port (oracle/libeincm_ref.so: test infrastructure, the multi-core checker of the full-size parity tests)
    is built by the TEST session, so that no test depends on the product's build() having done it."""
    import subprocess
    mk = os.path.join(ROOT, 'oracle', 'Makefile')
    if os.path.exists(mk):
        subprocess.run(['make', '-s', '-C', os.path.join(ROOT, 'oracle')], check=True)


@pytest.fixture(scope='session')
def built_lib():
    """The in-tree HIP library (built on demand with hipcc; cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    import importlib
    return importlib.import_module('edge-informed-contrast-maximization_amd._lib').load()
