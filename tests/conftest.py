import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests are skipped (not failed) on a host without a HIP device or without the built library"""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    have_lib = os.path.exists(os.path.join(ROOT, "gigl_amd", "libgigl_hip.so"))
    if have_gpu and have_lib:
        return
    why = "no HIP device" if not have_gpu else "gigl_amd/libgigl_hip.so is not built"
    skip = pytest.mark.skip(reason=f"needs a real MI355X ({why})")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN



def seed_trainer() -> int:
    """the ONE seed of every test that trains through the entry points (the trainers seed nothing themselves, like the
    reference's: the initialisation comes from the process RNG).  Falling-loss / metric checks of those tests are smoke
    checks on this shared initialisation, not parity claims; GIGL_TEST_SEED overrides it for all of them at once."""
    import torch
    seed = int(os.environ.get("GIGL_TEST_SEED", "1"))
    torch.manual_seed(seed)
    return seed
