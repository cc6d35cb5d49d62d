import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def locref():
    """The CPU oracle (test infrastructure; oracle/locref.cpp)."""
    from oracle import locref as m
    m.lib()
    return m


@pytest.fixture(scope="session")
def synth():
    from loc_lib_amd import synth as m
    return m


@pytest.fixture(scope="session")
def api():
    from loc_lib_amd import api as m
    return m


@pytest.fixture(scope="session")
def gpu_ctx(api):
    """A product context on cuda:0. Fails loudly (no CPU fallback) when the HIP library or the GPU is missing."""
    api.lib()
    ctx = api.Context(0)
    yield ctx
    ctx.close()


# ---- small deterministic worlds shared by CPU and GPU tests (seconds to generate) -------------------------------
@pytest.fixture(scope="session")
def small_world(synth):
    """Dense box-cropped local map (±40 m, ≈20 pts/m², ground + building walls) + a 2 k-pt and a 10 k-pt scan of pose 3."""
    m = synth.make_local_map(200000, 3, half=40.0)
    scan2k = synth.make_scan(3, subsample=2000, crop_half=36.0)
    scan10k = synth.make_scan(3, subsample=10000, crop_half=36.0)
    true_pose, init_pose = synth.make_pose(3)
    return dict(map=m, scan2k=scan2k, scan10k=scan10k, true_pose=true_pose, init_pose=init_pose)


def pose_delta(a, b):
    """(translation distance [m], rotation angle [rad]) between two 7-double poses (quaternion xyzw + t)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    dt = float(np.linalg.norm(a[4:] - b[4:]))
    qa, qb = a[:4] / np.linalg.norm(a[:4]), b[:4] / np.linalg.norm(b[:4])
    d = abs(float(np.dot(qa, qb)))
    return dt, 2.0 * float(np.arccos(min(1.0, d)))
