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


def release_gpu_memory():
    """before a test starts other PROCESSES on the same GPU: give back what this process's contexts (the shared one and its lane contexts) hold - their grow-only scratch and the
    allocator's cache - and torch's cache"""
    try:
        import gc, torch
        from ngspeciesid_amd import runtime
        gc.collect()
        for api in list(runtime._apis.values()):
            api.set_option("release_scratch", 1)
        if torch.cuda.is_available(): torch.cuda.empty_cache()
    except Exception:
        pass
