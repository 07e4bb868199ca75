import os, sys, subprocess
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def gpu_api():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ngspeciesid_amd import runtime
    return runtime.get_api()
