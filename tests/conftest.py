import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import torch

    def load(name):
        # "double:<name>" -> the DoubleDQN fixture ddqn_<name>.pt (config["learner"] == "double")
        stem = f"ddqn_{name[len('double:'):]}" if name.startswith("double:") else f"dqn_{name}"
        return torch.load(os.path.join(GOLDEN_DIR, f"{stem}.pt"), map_location="cpu",
                          weights_only=False)
    return load


DQN_NAMES = ["tiny", "tiny_dynamic", "cfg1_cartpole_shape", "cfg2_shape_small_batch"]
DOUBLE_NAMES = ["double:tiny_dynamic", "double:cfg2_shape_small_batch"]
GOLDEN_NAMES = DQN_NAMES + DOUBLE_NAMES
