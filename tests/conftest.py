import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experiments: covers kernels of the experiments build only (make experiments; "
                            "RMD_LIB_PATH=build/variants/librmd_experiments.so); skipped against the product library")


@pytest.fixture(scope="session")
def orc():
    import oracle_lib
    return oracle_lib


@pytest.fixture(scope="session")
def rmd():
    import raymarchdenoisercuda_amd
    return raymarchdenoisercuda_amd


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU: torch.cuda.is_available() is False")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _experiments_only(request):
    """Tests marked `experiments`, and parametrised cases whose `variant` is an experimental a-trous formulation,
    run only against a library built with -DRMD_EXPERIMENTS."""
    marked = request.node.get_closest_marker("experiments") is not None
    variant = getattr(getattr(request.node, "callspec", None), "params", {}).get("variant")
    if not marked and variant in (None, 0, 1, 3):
        return
    import raymarchdenoisercuda_amd
    if not raymarchdenoisercuda_amd.HAS_EXPERIMENTS:
        pytest.skip("experiments build only (rmd_has_experiments() == 0)")
