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
def golden():
    """The reference's shipped golden filter sets (tests/golden/make_fixtures.py)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_fixtures.npz"))


@pytest.fixture(scope="session")
def grids(golden):
    return dict(
        azi=golden["grid/hrirGridAziRad"], zen=golden["grid/hrirGridZenRad"],
        mic_azi=golden["grid/micGridAziRad"], mic_zen=golden["grid/micGridZenRad"],
        mic_radius=float(golden["real_eMagLS_woDC/micRadius"]),
    )


@pytest.fixture(scope="session")
def hrirs(grids):
    from emagls_amd import synth
    return synth.rigid_sphere_hrirs(grids["azi"], grids["zen"])
