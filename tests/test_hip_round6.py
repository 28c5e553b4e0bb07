"""GPU tests added in round 6 (through the C ABI, like the others):

* `antialiase_error` on the LDS-staged kernels (the mode ran on the un-staged kernels only, 9.6 x the fused step per pixel):
  forward instances of raster_fwd_fast_kernel whose edges blend the error buffer, raster_bwd_fast_kernel for the tiles without silhouette
  edges, raster_bwd_edge_err_kernel (staged sweep over the error buffer, un-blend, 15-moment butterfly) for those with -- against the checker
  (adjoint: the build with defect D2 repaired) and against the un-staged family, on meshes with many-edge tiles, textured and not, both pixel
  types, several views;
* more than four channels on the staged forward when the frame has no silhouette edge (the frame of Scene3D.render_deferred, dr.py:1053-1174:
  15 channels of a triangle soup, sigma = 0, background image): forward against the checker, then the (un-staged) adjoint on the state the
  staged forward left.
"""

import numpy as np
import pytest
import torch

from deodr_amd import scenes

pytestmark = pytest.mark.gpu

F32, F64 = torch.float32, torch.float64


def checker(api, fixed=False):
    return api.ref(fixed=fixed) or api.port(fixed=fixed)


def _aa_case(oracle_api, views, sigma, dt, seed=5):
    """antialiase_error forward + adjoint of `views` (one topology) on the default (staged) family against the checker, view by view."""
    from hip_util import hip_grads, hip_render, rel_err

    views = views if isinstance(views, list) else [views]
    s0 = views[0]
    H, W, C = s0.height, s0.width, s0.nb_colors
    rs = np.random.RandomState(seed)
    obs = rs.rand(len(views), H, W, C)
    err_b = rs.rand(len(views), H, W) + 0.1
    ds, r, out = hip_render(views, sigma, dt, True, obs)
    g = hip_grads(ds, r, err_buffer_b=err_b)
    fixed = checker(oracle_api, fixed=True)
    tol_img, tol_g = (1e-9, 1e-8) if dt == F64 else (1e-5, 1e-4)
    uv_sum, tex_sum = 0.0, 0.0
    for i, s in enumerate(views):
        image, z, err = fixed.render(s, sigma, True, obs[i])
        assert np.abs(out[0][i] - image).max() < tol_img
        assert np.abs(out[2][i] - err).max() < tol_img * max(1.0, err.max())
        fin = np.isfinite(z)
        assert (np.isfinite(out[1][i]) == fin).all()
        g_fix = fixed.grads(s, sigma, image, z, None, True, obs[i], err, err_b[i])
        for k in ("ij_b", "colors_b", "shade_b"):
            assert rel_err(g[k][i], g_fix[k]) < tol_g, (k, i)
        uv_sum, tex_sum = uv_sum + g_fix["uv_b"], tex_sum + g_fix["texture_b"]
    if np.size(s0.texture):
        assert rel_err(g["uv_b"], uv_sum) < tol_g and rel_err(g["texture_b"], tex_sum) < tol_g
    return ds, r, out, g, obs, err_b


@pytest.mark.parametrize("dt", [F32, F64])
@pytest.mark.parametrize("textured", [False, True])
def test_antialiase_error_on_a_mesh_with_many_edge_tiles(oracle_api, dt, textured):
    """A small frame of a fine mesh: tiles at the limb hold several batches of silhouette edges (the staged sweep over the error buffer
    stages them 16 at a time); sigma = 2.5 widens the bands."""
    from deodr_amd import hip_renderer as hr
    from hip_util import rel_err

    s = scenes.sphere_scene(size=96, nu=60, n_rings=48, nb_colors=3, depth_channel=False, textured=textured, texture_size=32, angle=0.1)
    ds, r, out, g, obs, err_b = _aa_case(oracle_api, s, 2.5, dt)
    from deodr_amd.hip_renderer import tile_census

    assert tile_census(r, ds)[1] > 20
    # the same call on the un-staged family: the two families agree far inside the tolerance of either against the checker
    hr.force_generic(True)
    try:
        from hip_util import hip_grads, hip_render

        ds2, r2, out2 = hip_render(s, 2.5, dt, True, obs)
        g2 = hip_grads(ds2, r2, err_buffer_b=err_b)
    finally:
        hr.force_generic(False)
    tol = 1e-10 if dt == F64 else 2e-6
    assert np.abs(out[2][0] - out2[2][0]).max() < tol * max(1.0, np.abs(out2[2][0]).max())
    for k in ("ij_b", "colors_b", "shade_b", "uv_b"):
        assert rel_err(g[k], g2[k]) < (1e-9 if dt == F64 else 1e-5), k


@pytest.mark.parametrize("dt", [F32, F64])
def test_antialiase_error_three_views_and_sigma_zero(oracle_api, dt):
    views = [scenes.sphere_scene(size=128, nu=30, n_rings=24, nb_colors=4, angle=float(a)) for a in (-0.4, 0.0, 0.3)]
    _aa_case(oracle_api, views, 1.0, dt)
    _aa_case(oracle_api, views[:1], 0.0, dt)  # no edge anywhere: every tile through the owner-tile kernel


def test_antialiase_error_soup_every_edge_flagged(oracle_api):
    """The soup of the reference's own antialiase_error goldens (tests/test_triangle_soup_fitting.py:50-67): large triangles, every edge a
    silhouette edge, a background image, tiles of more than one batch of edges; ragged frame size."""
    s = scenes.soup_scene(n_tri=60, width=100, height=83, seed=4)
    _aa_case(oracle_api, s, 1.0, F64)
    _aa_case(oracle_api, s, 1.0, F32)


@pytest.mark.parametrize("dt", [F32, F64])
@pytest.mark.parametrize("channels,background_image,w,h", [(15, True, 160, 120), (6, False, 93, 70)])
def test_many_channel_frame_on_the_staged_forward(oracle_api, dt, channels, background_image, w, h):
    from deodr_amd import hip_renderer as hr
    from hip_util import hip_grads, hip_render, image_report, rel_err

    s = scenes.deferred_scene(channels=channels, nu=40, n_rings=30, angle=0.2, width=w, height=h, background_image=background_image)
    ref = checker(oracle_api)
    image_ref, z_ref = ref.render(s, 0.0)
    ds, r, out = hip_render(s, 0.0, dt)
    err, flipped = image_report(out[0][0], image_ref, out[1][0], z_ref, 1e-5)
    assert flipped == 0 and err < (1e-9 if dt == F64 else 1e-5)
    fin = np.isfinite(z_ref)
    assert np.abs(out[1][0][fin] - z_ref[fin]).max() < (1e-9 if dt == F64 else 1e-5)
    # the adjoint (un-staged kernels: more than four channels) on the state the staged forward left: owner ids, edge counts
    image_b = np.random.RandomState(7).randn(h, w, channels)
    g = hip_grads(ds, r, image_b=image_b[None])
    g_ref = ref.grads(s, 0.0, image_ref, z_ref, image_b)
    for k in ("ij_b", "colors_b"):
        assert rel_err(g[k][0], g_ref[k]) < (1e-8 if dt == F64 else 1e-4), k
    # and the un-staged forward gives the same frame
    hr.force_generic(True)
    try:
        _ds, _r, out2 = hip_render(s, 0.0, dt)
    finally:
        hr.force_generic(False)
    assert np.abs(out2[0][0] - out[0][0]).max() < (1e-12 if dt == F64 else 1e-6)
    assert (np.isfinite(out2[1][0]) == np.isfinite(out[1][0])).all()


@pytest.mark.parametrize("dt", [F32, F64])
def test_many_channel_frame_64_channels_two_views(oracle_api, dt):
    """The channel limit (DEODR_HIP_MAX_COLORS = 64): a tile row of 64 float64 channels is 4 KB -- the frame leaves LDS one tile row per pass --, two views."""
    from hip_util import hip_render, image_report

    views = [scenes.deferred_scene(channels=64, nu=24, n_rings=20, angle=a, width=88, height=64, background_image=True, seed=9) for a in (0.0, 0.4)]
    ref = checker(oracle_api)
    ds, r, out = hip_render(views, 0.0, dt)
    for i, s in enumerate(views):
        image_ref, z_ref = ref.render(s, 0.0)
        err, flipped = image_report(out[0][i], image_ref, out[1][i], z_ref, 1e-5)
        assert flipped == 0 and err < (1e-9 if dt == F64 else 1e-5), (i, err, flipped)


def test_antialiase_error_forward_perspective_correct(oracle_api):
    """perspective_correct has no adjoint in the reference (H.h:810), but its forward exists in every mode: the staged antialiase_error forward too."""
    from hip_util import hip_render

    s = scenes.sphere_scene(size=112, nu=30, n_rings=24, nb_colors=3, depth_channel=False, angle=0.15)
    s.perspective_correct = True
    rs = np.random.RandomState(3)
    obs = rs.rand(s.height, s.width, 3)
    ref = checker(oracle_api)
    image, z, err = ref.render(s, 1.5, True, obs)
    for dt, tol in ((F64, 1e-9), (F32, 1e-5)):
        ds, r, out = hip_render(s, 1.5, dt, True, obs)
        assert np.abs(out[0][0] - image).max() < tol and np.abs(out[2][0] - err).max() < tol * max(1.0, err.max())
        assert (np.isfinite(out[1][0]) == np.isfinite(z)).all()
