import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def golden_soup(clockwise, prefix):
    """Scene2D rebuilt from tests/golden/soup30_cw{0,1}.npz (inputs produced by the reference's own example)."""
    from deodr_amd.differentiable_renderer import Scene2D

    d = np.load(os.path.join(GOLDEN, f"soup30_cw{int(clockwise)}.npz"))
    h, w = int(d["height"]), int(d["width"])
    keys = ["faces", "faces_uv", "ij", "depths", "textured", "uv", "shade", "colors", "shaded", "edgeflags"]
    scene = Scene2D(
        height=h, width=w, nb_colors=3, texture=d["texture_u8"].astype(np.float64) / 255,
        background_image=np.tile(d["background_rgb"][None, None, :], (h, w, 1)), clockwise=bool(d["clockwise"]),
        backface_culling=True, **{k: d[prefix + k] for k in keys},
    )  # fmt: skip
    return scene, d


@pytest.fixture(scope="session")
def oracle_api():
    from oracle import api

    return api


@pytest.fixture(params=["staged", "generic"])
def family(request):
    """Run the test once per kernel family of the HIP library (LDS-staged / un-staged): deodr_hip_force_generic."""
    from deodr_amd import hip_renderer as hr

    hr.force_generic(request.param == "generic")
    yield request.param
    hr.force_generic(False)
