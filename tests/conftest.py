import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

DATA = os.path.join(ROOT, "tests", "golden", "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")
DATASETS = ["INTEL", "M3500", "MIT", "CSAIL", "FR079", "FRH"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


def _have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def pgo():
    import toy_robust_backend_slam_amd as P
    P.lib()
    return P


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.lib()
    return O


def oracle_graph(O, g):
    """pgo Graph -> oracle Graph (copies)"""
    import numpy as np
    return O.Graph(np.array(g.pose_ids), np.array(g.poses), np.array(g.ia), np.array(g.ib), np.array(g.meas),
                   np.array(g.info), np.array(g.kind))
