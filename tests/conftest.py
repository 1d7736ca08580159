import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are the parity tests proper (they call the HIP library through the C ABI): skipped, not failed, on a box without
    an MI355X, so a plain `pytest` in the build container runs the CPU suite only."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP GPU on this box (run with -m gpu on the MI355X)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def hiplib():
    """Build (if stale) and load libdd3d_hip.so; hipcc cross-compiles gfx950 without a GPU."""
    import __graft_entry__ as g
    g.build()
    from dd3d_amd import hip
    return hip.lib()


@pytest.fixture(scope="session")
def kitti_dla34():
    """(cfg, cpu model, synthetic state_dict) of DD3D-DLA34 / KITTI."""
    import dd3d_amd.modeling  # noqa: F401
    from dd3d_amd import META_ARCH_REGISTRY, get_cfg
    from dd3d_amd.synthetic import load_calib, make_state_dict
    cfg = get_cfg("dd3d_kitti_dla34")
    model = META_ARCH_REGISTRY.get("DD3D")(cfg)
    sd = make_state_dict(model, calib=load_calib("dla34_kitti"))
    return cfg, model, sd
