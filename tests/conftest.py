import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def smplx_data():
    import synthetic
    return synthetic.make_smplx_data(seed=0)


@pytest.fixture(scope="session")
def mean_params():
    import synthetic
    return synthetic.make_mean_params(seed=0)
