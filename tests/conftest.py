import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        from palace_amd import lib

        return lib.load().pa_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a HIP device (or without the built library) skips the gpu-marked tests instead
    of erroring out in each of them; with `-m gpu` on such a box every test shows up as skipped, not as passed."""
    if not any("gpu" in it.keywords for it in items):
        return
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no HIP device / libpalace_amd.so: there is no CPU fallback")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def cylinder_mesh():
    """The reference's cylinder_hex.msh (80 hex27) from the committed fixture."""
    import numpy as np

    from palace_amd.fem.mesh import HexMesh

    d = np.load(os.path.join(ROOT, "tests", "golden", "cylinder_hex_mesh.npz"))
    m = HexMesh(x=d["x"], elem_nodes=d["elem_nodes"].astype(np.int64), attr=d["attr"],
                bdr_faces=d["bdr_faces"], bdr_attr=d["bdr_attr"])
    m.check()
    return m
