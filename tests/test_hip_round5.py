"""GPU tests added in round 5 (through the C ABI, like the others):

* textured && !shaded triangles (H.h:2798, 2813: out of pass 1; H.h:2868-2895: their silhouette edges drawn INTERPOLATED from the vertex
  colours; the same branches of renderScene_B) on a mesh with shared vertices -- the soups of FLAG_CASES cover the flag space
  (tests/test_oracle.py, cases 7 and 8);
* the deterministic mode raises DEODR_HIP_ERR_DET_RANGE when a contribution or a running sum leaves its fixed-point range.
"""

import numpy as np
import pytest
import torch

from deodr_amd import scenes

pytestmark = pytest.mark.gpu

F32, F64 = torch.float32, torch.float64


def checker(api, fixed=False):
    return api.ref(fixed=fixed) or api.port(fixed=fixed)


def mixed_shading_sphere(size=160, angle=0.0, seed=3):
    """A textured sphere whose triangles are textured && !shaded in bands: pass 1 leaves holes there (the back faces are culled, so the
    background shows through), the silhouette edges of those bands blend the vertex colours."""
    s = scenes.sphere_scene(size=size, nu=30, n_rings=24, nb_colors=3, textured=True, texture_size=32, angle=angle, seed=seed)
    s.shaded = s.shaded.copy()
    s.shaded[(np.arange(s.shaded.size) // 5) % 3 == 0] = False
    s.colors = np.random.RandomState(seed).rand(*s.colors.shape)
    # edges of the holes are silhouette edges too: flag every edge of an unshaded triangle and of its shaded neighbours' shared edges
    s.edgeflags = s.edgeflags.copy()
    s.edgeflags[~s.shaded] = True
    return s


@pytest.mark.parametrize("dt", [F32, F64])
@pytest.mark.parametrize("sigma", [0.0, 1.5])
def test_textured_unshaded_triangles_on_a_mesh(oracle_api, family, sigma, dt):
    from test_hip_parity import compare_backward, compare_fit_step

    s = mixed_shading_sphere()
    assert (s.textured & ~s.shaded).sum() > 100 and (s.textured & s.shaded).sum() > 100
    compare_backward(oracle_api, s, sigma, dt)
    compare_fit_step(oracle_api, s, sigma, dt)
    views = [mixed_shading_sphere(angle=a) for a in (-0.3, 0.2)]
    for v in views[1:]:
        v.shaded, v.edgeflags = views[0].shaded, views[0].edgeflags  # one topology
    if sigma > 0:
        compare_fit_step(oracle_api, views, sigma, dt)


def test_textured_unshaded_antialiase_error(oracle_api):
    from hip_util import hip_grads, hip_render, rel_err

    s = mixed_shading_sphere(size=96)
    rs = np.random.RandomState(11)
    obs, err_b = rs.rand(s.height, s.width, 3), rs.rand(s.height, s.width)
    ds, r, out = hip_render(s, 1.0, F64, True, obs)
    fixed = checker(oracle_api, fixed=True)
    image, z, err = fixed.render(s, 1.0, True, obs)
    assert np.abs(out[0][0] - image).max() < 1e-9 and np.abs(out[2][0] - err).max() < 1e-8 * max(1.0, err.max())
    g = hip_grads(ds, r, err_buffer_b=err_b)
    g_fix = fixed.grads(s, 1.0, image, z, None, True, obs, err, err_b)
    for k in ("ij_b", "colors_b", "shade_b"):
        assert rel_err(g[k][0], g_fix[k]) < 1e-8, k
    assert rel_err(g["uv_b"], g_fix["uv_b"]) < 1e-8 and rel_err(g["texture_b"], g_fix["texture_b"]) < 1e-8
    assert np.abs(g_fix["colors_b"]).max() > 0  # (the unshaded triangles' edges did put gradient on the vertex colours)


def test_deterministic_mode_reports_a_sum_beyond_its_range(oracle_api):
    """|sum| < 2^31 is the fixed-point range of the integer accumulation: beyond it the sticky bit is raised (it used to wrap silently)."""
    from hip_util import device_scene
    from deodr_amd import hip_renderer as hr
    from test_oracle import random_scene

    s = random_scene(4100)
    s.backface_culling = True
    hr.set_deterministic(True)
    try:
        for scale, expect in ((1.0, 0), (1e13, hr.ERR_DET_RANGE)):
            ds = device_scene(s, F64)
            r = hr.HipRasterizer.for_scene(ds)
            image_b = torch.as_tensor(np.random.RandomState(3).randn(1, s.height, s.width, s.nb_colors) * scale, device=ds.device)
            r.render(ds, 1.0, check_overflow=True)
            r.render_backward(ds, image_b=image_b)
            _over, _need, errs = r.status(ds)
            assert errs == expect, (scale, errs)
            assert ("fixed-point range" in hr.scene_error_message(errs)) == bool(expect)
    finally:
        hr.set_deterministic(False)


@pytest.mark.parametrize("dt", [F32, F64])
def test_overlapped_reduction_of_a_textured_scene(oracle_api, dt):
    """OverlappedViewsReduction on a textured scene: texture_b and uv_b are part of the packed buffer (the library accumulates the views' taps
    straight into it), next to the vertex / colour sums -- against the sum over the views of per-view calls of the checker, and against the
    plain (unpacked) gradient arrays of the same step; float32 buffers pack in float32, float64 in float64."""
    from hip_util import device_scene, rel_err
    from deodr_amd import fronthalf
    from deodr_amd.distributed import OverlappedViewsReduction
    from deodr_amd.hip_renderer import HipRasterizer
    from deodr_amd.scene3d import DeviceCamera

    n, S = 3, 128
    poses = np.linspace(-0.3, 0.3, n)
    views = [scenes.sphere_scene(size=S, nu=24, n_rings=20, nb_colors=3, textured=True, texture_size=16, angle=float(a)) for a in poses]
    verts, _f = scenes.bumpy_sphere(24, 20)
    cams = [scenes.fit_camera(S, S, 60.0, verts, scenes.rotx(0.37) @ scenes.roty(0.23 + float(a))) for a in poses]
    ds = device_scene(views, dt)
    r = HipRasterizer.for_scene(ds)
    camera = DeviceCamera(np.stack([c.extrinsic for c in cams]), np.stack([c.intrinsic for c in cams]), S, S, None, ds.device)
    posed = torch.as_tensor(np.ascontiguousarray(verts), device=ds.device)[None].expand(n, -1, -1).contiguous()
    V, C = posed.shape[1], ds.nb_colors
    red = OverlappedViewsReduction(ds, camera, posed)
    assert red.textured and red.slots[0].shared.dtype == dt and red.slots[0].grads["texture_b"].data_ptr() == red.slots[0].shared.data_ptr()
    assert red.slots[0].shared.numel() == 16 * 16 * 3 + V * (3 + C) + 2 * ds.uv.shape[0]
    rng = np.random.RandomState(4)
    r.render(ds, 1.0, check_overflow=True)
    kept = []
    for step in range(5):
        obs = torch.as_tensor(rng.rand(n, S, S, C), device=ds.device).to(dt)
        slot = red.begin()
        r.render_fit(ds, obs, 1.0, grads=slot.grads, clear_grads=True, check_overflow=False, done_flag=slot.done_flag)
        red.reduce(slot)
        kept.append((slot, obs))
    red.finish()
    slot, obs = kept[-1]
    # the same step into plain arrays
    image, _z, g = r.render_fit(ds, obs, 1.0, check_overflow=False, clear_grads=True)
    vb, cs = torch.zeros(V, 3, dtype=F64, device=ds.device), torch.zeros(V, C, dtype=F64, device=ds.device)
    fronthalf.views_gradient_sum(posed, camera, g["ij_b"], vb, colors_b=g["colors_b"], colors_sum=cs)
    torch.cuda.synchronize()
    tol = 1e-6 if dt == F32 else 1e-11  # (run-to-run order of the atomics; float32: the packed buffer's own rounding)
    assert rel_err(slot.vertices_b.cpu().numpy(), vb.cpu().numpy()) < tol
    assert rel_err(slot.uv_b.cpu().numpy(), g["uv_b"].cpu().numpy()) < tol
    assert rel_err(slot.texture_b.cpu().numpy(), g["texture_b"].cpu().numpy()) < (1e-5 if dt == F32 else 1e-11)
    # the checker, view by view
    fixed = checker(oracle_api, fixed=True)
    uv_ref = tex_ref = 0
    for i, s in enumerate(views):
        img_ref, z_ref = fixed.render(s, 1.0)
        image_b = 2 * (image[i].cpu().numpy().astype(np.float64) - obs[i].cpu().numpy().astype(np.float64))
        gr = fixed.grads(s, 1.0, img_ref, z_ref, image_b)
        uv_ref, tex_ref = uv_ref + gr["uv_b"], tex_ref + gr["texture_b"]
    tol = 1e-4 if dt == F32 else 1e-8
    assert np.abs(tex_ref).max() > 0 and rel_err(slot.texture_b.cpu().numpy(), tex_ref) < tol and rel_err(slot.uv_b.cpu().numpy(), uv_ref) < tol


def test_views_gradient_sum_refuses_what_it_cannot_read():
    """the kernel reads contiguous float64 of fixed shapes -- a float32 vertex_dtype or a strided view used to be read out of bounds"""
    from hip_util import device_scene
    from deodr_amd import fronthalf
    from deodr_amd.distributed import OverlappedViewsReduction
    from deodr_amd.scene3d import DeviceCamera

    n, S = 2, 64
    poses = np.linspace(-0.2, 0.2, n)
    views = [scenes.sphere_scene(size=S, nu=10, n_rings=8, angle=float(a)) for a in poses]
    verts, _f = scenes.bumpy_sphere(10, 8)
    cams = [scenes.fit_camera(S, S, 60.0, verts, scenes.rotx(0.37) @ scenes.roty(0.23 + float(a))) for a in poses]
    ds32 = device_scene(views, F32, vertex_dtype=F32)
    camera = DeviceCamera(np.stack([c.extrinsic for c in cams]), np.stack([c.intrinsic for c in cams]), S, S, None, ds32.device)
    posed = torch.as_tensor(np.ascontiguousarray(verts), device=ds32.device)[None].expand(n, -1, -1).contiguous()
    V = posed.shape[1]
    with pytest.raises(ValueError, match="float64"):
        OverlappedViewsReduction(ds32, camera, posed)
    g = ds32.zero_grads()
    vb = torch.zeros(V, 3, dtype=F64, device=ds32.device)
    with pytest.raises(ValueError, match="ij_b"):
        fronthalf.views_gradient_sum(posed, camera, g["ij_b"], vb)  # float32 ij_b
    with pytest.raises(ValueError, match="vertices_b"):
        fronthalf.views_gradient_sum(posed, camera, g["ij_b"].double(), torch.zeros(V, 6, dtype=F64, device=ds32.device)[:, ::2])  # strided


@pytest.mark.parametrize("n_views", [1, 9])
def test_textured_tiles_of_many_edges_in_the_fused_forward(oracle_api, n_views):
    """Round 5: the forward raster of a TEXTURED fit step back-propagates its tiles with silhouette edges as well (no edge-tile kernel, no saved
    sweep): ~60 textured edges crowded into one tile of a 512^2 frame (a grid with a head of the list: the tile is listed once per part of 8
    edges for one view, 16 for nine), ~200 edges (more than the staged sweep orders: the un-staged tile code inside the forward raster) --
    against the checker (uv_b, texture_b included) and the two-call path, which still runs raster_bwd_edge_kernel."""
    from test_hip_parity import compare_fit_step
    from test_hip_parity2 import crowded_scene

    def textured_crowd(n_tri, seed):
        s = crowded_scene(n_tri, seed=seed, size=512)
        rs = np.random.RandomState(seed + 50)
        t = rs.rand(n_tri) < 0.7  # (the others stay interpolated: both kinds of edges in the same tile)
        t3 = np.repeat(t, 3)
        s.textured, s.shaded = t, t.copy()
        s.colors = np.where(t3[:, None], 0.0, s.colors)
        s.shade = np.where(t3, rs.rand(3 * n_tri), 0.0)
        s.uv = np.where(t3[:, None], rs.rand(3 * n_tri, 2) * (min(s.texture.shape[:2]) - 1), 0.0)
        return s

    views = [textured_crowd(20, 3 + i) for i in range(n_views)]
    for v in views[1:]:  # (one mesh, one texture: which triangles are textured, and their uv, are shared by the views; positions, colours, shade are per view)
        t3 = np.repeat(views[0].textured, 3)
        v.textured, v.shaded, v.uv, v.texture = views[0].textured, views[0].shaded, views[0].uv, views[0].texture
        v.colors = np.where(t3[:, None], 0.0, np.random.RandomState(9).rand(*v.colors.shape))
        v.shade = np.where(t3, np.abs(v.shade) + 0.25, 0.0)
    assert np.size(views[0].texture) > 0 and views[0].textured.sum() > 8
    compare_fit_step(oracle_api, views, 1.0, F64)
    if n_views == 1:
        compare_fit_step(oracle_api, views, 2.5, F32)
        compare_fit_step(oracle_api, textured_crowd(70, 5), 1.0, F64)


def test_textured_fit_step_launches_no_edge_tile_kernel(oracle_api):
    """the structure itself: with sigma > 0 a textured fit step is set-up, scan + forward raster, finalize -- the edge-tile kernel of the two-call
    path is not launched (per-kernel launch counts of the profiling hook); and the frame and gradients do not depend on the kernel family"""
    import ctypes

    from hip_util import device_scene
    from deodr_amd import hip_renderer as hr

    s = scenes.sphere_scene(size=256, nu=40, n_rings=30, nb_colors=3, textured=True, texture_size=32)
    ds = device_scene(s, F32)
    r = hr.HipRasterizer.for_scene(ds)
    obs = torch.rand((1, 256, 256, 3), dtype=F32, device=ds.device)
    r.render_fit(ds, obs, 1.0, check_overflow=True, clear_grads=True)
    torch.cuda.synchronize()
    ms, launches = (ctypes.c_double * 4)(), (ctypes.c_ulonglong * 4)()
    hr.lib().deodr_hip_profile_enable(1)
    try:
        hr.lib().deodr_hip_profile_read(ms, launches)  # (reset)
        for _ in range(3):
            r.render_fit(ds, obs, 1.0, check_overflow=False, clear_grads=True)
        torch.cuda.synchronize()
        hr.lib().deodr_hip_profile_read(ms, launches)
        assert launches[1] == 3 and launches[3] == 3 and launches[2] == 0, list(launches)
        r.render(ds, 1.0)
        r.render_backward(ds, residual_obs=obs)
        torch.cuda.synchronize()
        hr.lib().deodr_hip_profile_read(ms, launches)
        assert launches[2] >= 1, list(launches)  # (the two-call path: the edge-tile kernel)
    finally:
        hr.lib().deodr_hip_profile_enable(0)


def test_deterministic_steps_on_two_streams_do_not_share_their_shadows(oracle_api):
    """the int64 shadows of the deterministic mode were one buffer per device -- two fit steps in flight on two streams added into,
    converted and cleared the same words.  One per (device, stream) now: two different scenes stepped concurrently on two streams give, bit for bit,
    what each gives alone."""
    from hip_util import device_scene
    from deodr_amd import hip_renderer as hr
    from test_oracle import random_scene

    hr.set_deterministic(True)
    try:
        jobs = []
        for seed in (4200, 4201):
            s = random_scene(seed)
            s.backface_culling = True
            ds = device_scene(s, F64)
            r = hr.HipRasterizer.for_scene(ds)
            obs = torch.as_tensor(np.random.RandomState(seed).rand(1, s.height, s.width, s.nb_colors), device=ds.device)
            r.render_fit(ds, obs, 1.0, check_overflow=True, clear_grads=True)  # (pools sized, shadows of the default stream allocated)
            alone = {k: v.clone() for k, v in r.render_fit(ds, obs, 1.0, check_overflow=False, clear_grads=True)[2].items() if v is not None}
            jobs.append((ds, r, obs, alone, torch.cuda.Stream()))
        torch.cuda.synchronize()
        for ds, r, obs, _alone, st in jobs:  # (first use of each stream: its shadows are allocated, which synchronises)
            with torch.cuda.stream(st):
                r.render_fit(ds, obs, 1.0, check_overflow=False, clear_grads=True)
        torch.cuda.synchronize()
        for _rep in range(20):
            got = []
            for ds, r, obs, _alone, st in jobs:
                with torch.cuda.stream(st):
                    got.append(r.render_fit(ds, obs, 1.0, check_overflow=False, clear_grads=True)[2])
            torch.cuda.synchronize()
            for (_ds, _r, _obs, alone, _st), g in zip(jobs, got):
                for k, v in alone.items():
                    assert torch.equal(g[k], v), k
    finally:
        hr.set_deterministic(False)


def test_deterministic_mode_per_scene(oracle_api):
    """ABI v12, DeodrHipScene::deterministic: the integer accumulation as a property of the scene a call is made on (missing 6: the mode
    was only reachable through a process-wide switch) -- two scenes stepped alternately, one deterministic: its gradients are bit-identical from
    run to run and equal what the process-wide switch gives; the other scene keeps the default path (close to, and in general not bit-equal to, it)."""
    from hip_util import device_scene, rel_err
    from deodr_amd import hip_renderer as hr
    from test_oracle import random_scene

    s = random_scene(4300)
    s.backface_culling = True
    obs_np = np.random.RandomState(5).rand(1, s.height, s.width, s.nb_colors)
    det, plain = device_scene(s, F64), device_scene(s, F64)
    det.deterministic = True
    r_det, r_plain = hr.HipRasterizer.for_scene(det), hr.HipRasterizer.for_scene(plain)
    obs = torch.as_tensor(obs_np, device=det.device)
    runs = []
    for _rep in range(6):
        g_det = r_det.render_fit(det, obs, 1.0, check_overflow=True, clear_grads=True)[2]
        g_plain = r_plain.render_fit(plain, obs, 1.0, check_overflow=True, clear_grads=True)[2]
        torch.cuda.synchronize()
        runs.append({k: v.clone() for k, v in g_det.items() if v is not None})
        assert rel_err(g_plain["ij_b"].cpu().numpy(), g_det["ij_b"].cpu().numpy()) < 1e-8
    for k, v in runs[0].items():
        assert all(torch.equal(v, r[k]) for r in runs[1:]), k
    hr.set_deterministic(True)
    try:
        g_switch = r_plain.render_fit(plain, obs, 1.0, check_overflow=True, clear_grads=True)[2]
        torch.cuda.synchronize()
        for k, v in runs[0].items():
            assert torch.equal(v, g_switch[k]), k
    finally:
        hr.set_deterministic(False)


def test_textured_fit_step_of_eight_views_as_two_kernels(oracle_api):
    """From 8 views per launch on the forward raster of a textured fit step runs as two kernels on two streams (the head walkers with the edge
    adjoint on the library's side stream, the others on the caller's; joined in front of finalize): 8 views of a textured sphere against the checker,
    view by view, and against the two-call path; the same step captured in a HIP graph (capture takes the one-kernel form) gives the eager result."""
    from hip_util import device_scene, rel_err
    from deodr_amd.hip_renderer import HipRasterizer
    from test_hip_parity import compare_fit_step

    views = [scenes.sphere_scene(size=256, nu=40, n_rings=30, nb_colors=3, textured=True, texture_size=32, angle=float(a)) for a in np.linspace(-0.4, 0.4, 8)]
    compare_fit_step(oracle_api, views, 1.0, F32)
    compare_fit_step(oracle_api, views, 2.0, F64)
    ds = device_scene(views, F32)
    r = HipRasterizer.for_scene(ds)
    obs = torch.rand((8, 256, 256, 3), dtype=F32, device=ds.device)
    image, z = torch.empty((8, 256, 256, 3), dtype=F32, device=ds.device), torch.empty((8, 256, 256), dtype=F32, device=ds.device)
    grads = ds.zero_grads()
    for _ in range(3):  # (two kernels; also back to back: the side stream's events are re-recorded every step)
        r.render_fit(ds, obs, 1.0, grads=grads, out=(image, z), check_overflow=True, clear_grads=True)
    torch.cuda.synchronize()
    eager = (image.clone(), {k: v.clone() for k, v in grads.items() if v is not None})
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        r.render_fit(ds, obs, 1.0, grads=grads, out=(image, z), check_overflow=False, clear_grads=True)
    for _ in range(2):
        image.zero_()
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(image, eager[0])
    for k, v in eager[1].items():
        assert rel_err(grads[k].cpu().numpy(), v.cpu().numpy()) < (1e-5 if k == "texture_b" else 1e-9), k
