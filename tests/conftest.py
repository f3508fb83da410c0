import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
