"""GPU parity tests added in round 2 (all through the C ABI, all against oracle/_ref or the restatement):

* both kernel families (LDS-staged and un-staged "generic") on the same scenes -- ``deodr_hip_force_generic`` -- so that the
  driver's plain ``pytest -m gpu`` run covers both;
* nb_colors = 1 (depth fitting) and nb_colors = 10 (``render_deferred``), forward + adjoint;
* a tile with more than 128 silhouette edges (the un-staged fallback inside the adjoint's edge kernel);
* the 8-view textured batch: ``uv_b`` / ``texture_b`` summed over the views against the sum of per-view oracle gradients;
* BASELINE configs[3] at full size (8 views of the hand mesh at 1024 x 1024), every view against the oracle;
* a 50-iteration triangle-soup fit through ``Scene2D.render_compare_and_backward`` in lock-step with the reference's own
  loss curve (tests/golden/soup30_cw*.npz, produced by the reference's Python + Cython build);
* run-to-run determinism bound of the float64 atomics;
* checkSceneValid's index checks on a device-resident scene (sticky error word, no out-of-bounds access);
* HIP-graph capture of a fit step; two renders in one autograd graph; replaced background / texture objects.
"""

import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_soup
from deodr_amd import scenes
from test_oracle import BACKWARD_CASES, FLAG_CASES, random_scene

pytestmark = pytest.mark.gpu

F32, F64 = torch.float32, torch.float64
TOL = {F32: (1e-5, 1e-4), F64: (1e-9, 1e-8)}


def checker(api, fixed=False):
    return api.ref(fixed=fixed) or api.port(fixed=fixed)


def many_channel_scene(seed, nb_colors, **flags):
    """random_scene with `nb_colors` channels (untextured triangles only: a texture would need that many channels too)."""
    s = random_scene(seed, **flags)
    rs = np.random.RandomState(seed + 1000)
    n_tri = s.faces.shape[0]
    s.textured = np.zeros(n_tri, dtype=bool)
    s.shaded = np.zeros(n_tri, dtype=bool)
    s.colors = rs.rand(s.depths.shape[0], nb_colors)
    s.colors_b = np.zeros_like(s.colors)
    s.nb_colors = nb_colors
    s.texture = np.zeros((0, 0, nb_colors))
    s.texture_b = np.zeros((0, 0, nb_colors))
    s.background_image, s.background_color = None, rs.rand(nb_colors)
    s.backface_culling = True
    return s


def compare_all(api, s, sigma, dt, fit=True):
    """forward, adjoint (explicit image_b) and -- staged kernels permitting -- the one-call fit step, against the checker"""
    from hip_util import hip_grads, hip_render, image_report, rel_err

    ref = checker(api)
    img_ref, z_ref = ref.render(s, sigma)
    ds, r, out = hip_render(s, sigma, dt)
    tol_img, tol = TOL[dt]
    err, flipped = image_report(out[0][0], img_ref, out[1][0], z_ref, tol_img)
    assert flipped == 0 and err < tol_img, (flipped, err)
    image_b = np.random.RandomState(3).randn(*img_ref.shape)
    g = hip_grads(ds, r, image_b=image_b)
    g_ref = ref.grads(s, sigma, img_ref, z_ref, image_b)
    for k in ("ij_b", "colors_b", "shade_b"):
        assert rel_err(g[k][0], g_ref[k]) < tol, k
    assert rel_err(g["uv_b"], g_ref["uv_b"]) < tol
    if fit:
        C = img_ref.shape[2]
        obs = torch.as_tensor(np.random.RandomState(4).rand(1, s.height, s.width, C), device=ds.device, dtype=dt)
        image, z, gf = r.render_fit(ds, obs, sigma)
        torch.cuda.synchronize()
        res_b = 2 * (image[0].cpu().numpy().astype(np.float64) - obs[0].cpu().numpy().astype(np.float64))
        g_ref = ref.grads(s, sigma, img_ref, z_ref, res_b)
        for k in ("ij_b", "colors_b"):
            assert rel_err(gf[k][0].cpu().numpy(), g_ref[k]) < tol, ("fit", k)


@pytest.mark.parametrize("dt", [F32, F64])
@pytest.mark.parametrize("case", BACKWARD_CASES)
@pytest.mark.parametrize("sigma", [0.0, 1.0, 2.5])
def test_both_kernel_families_flag_space(oracle_api, family, case, sigma, dt):
    """The suite of test_hip_parity.py, reduced, on BOTH families (textured + untextured triangles, three channels)."""
    from test_hip_parity import compare_backward, compare_fit_step

    s = random_scene(500 + case, **FLAG_CASES[case])
    s.backface_culling = True
    compare_backward(oracle_api, s, sigma, dt)
    compare_fit_step(oracle_api, s, sigma, dt)


@pytest.mark.parametrize("case", [4, 5, 6, 7, 8])
def test_both_kernel_families_forward_only_flags(oracle_api, family, case):
    """perspective_correct and backface_culling=False are forward-only in the reference (H.h:2922, 810)."""
    from test_hip_parity import compare_forward

    s = random_scene(510 + case, **FLAG_CASES[case])
    for sigma in (0.0, 1.5):
        compare_forward(oracle_api, s, sigma, F64)


@pytest.mark.parametrize("dt", [F32, F64])
@pytest.mark.parametrize("nb_colors", [1, 2, 10])
def test_channel_counts(oracle_api, nb_colors, dt):
    """C = 1 is the depth-image fit (Scene3D.render_depth), C = 10 the stacked channels of Scene3D.render_deferred
    (dr.py:1053-1174); C <= 4 runs on the staged kernels, C = 10 on the un-staged ones in chunks of four channels."""
    for sigma in (0.0, 1.0):
        compare_all(oracle_api, many_channel_scene(600 + nb_colors, nb_colors), sigma, dt)
    compare_all(oracle_api, many_channel_scene(620 + nb_colors, nb_colors, clockwise=True, integer_pixel_centers=False), 2.0, dt)


def test_channel_counts_generic_family(oracle_api, family):
    compare_all(oracle_api, many_channel_scene(640, 1), 1.0, F64)
    compare_all(oracle_api, many_channel_scene(641, 4), 1.0, F64)


def crowded_scene(n_tri, seed=0, size=24, spread=2.0, extent=5.0):
    """n_tri small triangles (all three edges flagged) crowded around the middle of the frame: one 8 x 8 tile receives
    several hundred silhouette edges."""
    s = scenes.soup_scene(n_tri=n_tri, width=size, height=size, seed=seed, min_area=4.0, flat=False)
    rs = np.random.RandomState(seed + 7)
    ij = np.zeros((n_tri, 3, 2))
    front_sign = 1.0 if s.clockwise else -1.0
    for t in range(n_tri):
        while True:
            p = size / 2 + 0.37 + spread * (rs.rand(2) - 0.5) + extent * (rs.rand(3, 2) - 0.5)
            u, v = p[1] - p[0], p[2] - p[0]
            a2 = u[0] * v[1] - u[1] * v[0]
            if abs(a2) > 6:
                break
        ij[t] = p if a2 * front_sign > 0 else p[::-1]
    s.ij = ij.reshape(-1, 2)
    s.depths = np.repeat(rs.rand(n_tri) + 0.5, 3) + 0.01 * rs.rand(3 * n_tri)  # distinct depth sums: a defined blending order
    return s


@pytest.mark.parametrize("n_tri,dt", [(20, F64), (70, F64), (70, F32)])
def test_many_edges_in_one_tile(oracle_api, n_tri, dt):
    """20 triangles: ~60 edges in a tile (four batches of the staged reverse sweep, shared by four wavefronts); 70 triangles:
    ~200 edges, more than the staged kernels order (EMAX = 128) -> ordered search in the forward, the un-staged tile code
    inside the adjoint's edge kernel.  Two-call path and fit step."""
    from hip_util import device_scene, hip_grads, hip_render, rel_err
    from deodr_amd.hip_renderer import HipRasterizer

    s = crowded_scene(n_tri)
    ref = checker(oracle_api)
    img_ref, z_ref = ref.render(s, 1.0)
    ds, r, out = hip_render(s, 1.0, dt)
    tol_img, tol = TOL[dt]
    assert np.abs(out[0][0] - img_ref).max() < 10 * tol_img  # hundreds of blends per pixel
    image_b = np.random.RandomState(1).randn(*img_ref.shape)
    g = hip_grads(ds, r, image_b=image_b)
    g_ref = ref.grads(s, 1.0, img_ref, z_ref, image_b)
    for k in ("ij_b", "colors_b"):
        assert rel_err(g[k][0], g_ref[k]) < 10 * tol, k
    obs = torch.as_tensor(np.random.RandomState(2).rand(1, s.height, s.width, 3), device=ds.device, dtype=dt)
    image, z, gf = r.render_fit(ds, obs, 1.0)
    torch.cuda.synchronize()
    res_b = 2 * (image[0].cpu().numpy().astype(np.float64) - obs[0].cpu().numpy().astype(np.float64))
    g_ref = ref.grads(s, 1.0, img_ref, z_ref, res_b)
    for k in ("ij_b", "colors_b"):
        assert rel_err(gf[k][0].cpu().numpy(), g_ref[k]) < 10 * tol, ("fit", k)


def test_textured_8_view_batch_sums_over_views(oracle_api):
    """BASELINE configs[4] shape at test size: 8 views of one textured mesh in one launch; uv_b and texture_b are summed over
    the views by the kernels and compared with the SUM of the per-view oracle gradients (texture_b: repaired reference, D1)."""
    from hip_util import device_scene, rel_err
    from deodr_amd.hip_renderer import HipRasterizer

    big = dict(size=256, nu=40, n_rings=40, nb_colors=3, textured=True, texture_size=64)
    views = [scenes.sphere_scene(angle=float(a), **big) for a in np.linspace(-0.5, 0.5, 8)]
    ds = device_scene(views, F32)
    r = HipRasterizer.for_scene(ds)
    n, H, W, Cc = 8, 256, 256, 3
    obs = torch.as_tensor(np.random.RandomState(5).rand(n, H, W, Cc).astype(np.float32), device=ds.device)
    image, z, g = r.render_fit(ds, obs, 1.0, check_overflow=True, clear_grads=True)
    image2, z2 = r.render(ds, 1.0)
    g2 = r.render_backward(ds, residual_obs=obs)
    torch.cuda.synchronize()
    ref, fixed = checker(oracle_api), checker(oracle_api, fixed=True)
    uv_sum, tex_sum = 0.0, 0.0
    for i, s in enumerate(views):
        img_ref, z_ref = ref.render(s, 1.0)
        assert np.abs(image[i].cpu().numpy() - img_ref).max() < 1e-5
        image_b = 2 * (image[i].cpu().numpy().astype(np.float64) - obs[i].cpu().numpy().astype(np.float64))
        gr = ref.grads(s, 1.0, img_ref, z_ref, image_b)
        uv_sum = uv_sum + gr["uv_b"]
        tex_sum = tex_sum + fixed.grads(s, 1.0, img_ref, z_ref, image_b)["texture_b"]
        for k in ("ij_b", "shade_b"):
            assert rel_err(g[k][i].cpu().numpy(), gr[k]) < 1e-4, (k, i)
    for got in (g, g2):
        assert rel_err(got["uv_b"].cpu().numpy(), uv_sum) < 1e-4
        assert rel_err(got["texture_b"].cpu().numpy(), tex_sum) < 1e-4


def test_config4_hand_8_views_full_size(oracle_api):
    """BASELINE configs[3]: 8 independent 1024 x 1024 views of the hand mesh in one launch, EVERY view against the oracle."""
    from hip_util import device_scene, rel_err
    from deodr_amd.hip_renderer import HipRasterizer

    path = os.path.join(GOLDEN, "hand_mesh.npz")
    views = [scenes.hand_scene(path, size=1024, angle=float(a), textured=False) for a in np.linspace(-0.5, 0.5, 8)]
    ds = device_scene(views, F32)
    r = HipRasterizer.for_scene(ds)
    obs = torch.as_tensor(np.random.RandomState(6).rand(8, 1024, 1024, 3).astype(np.float32), device=ds.device)
    image, z, g = r.render_fit(ds, obs, 1.0, check_overflow=True, clear_grads=True)
    torch.cuda.synchronize()
    ref = checker(oracle_api)
    for i, s in enumerate(views):
        img_ref, z_ref = ref.render(s, 1.0)
        assert np.abs(image[i].cpu().numpy() - img_ref).max() < 1e-5
        fin = np.isfinite(z_ref)
        assert np.array_equal(np.isfinite(z[i].cpu().numpy()), fin)
        image_b = 2 * (image[i].cpu().numpy().astype(np.float64) - obs[i].cpu().numpy().astype(np.float64))
        gr = ref.grads(s, 1.0, img_ref, z_ref, image_b)
        for k in ("ij_b", "colors_b"):
            assert rel_err(g[k][i].cpu().numpy(), gr[k]) < 1e-4, (k, i)


@pytest.mark.parametrize("clockwise,aa", [(0, 0), (1, 0), (0, 1)])
def test_soup_fit_50_iterations_lockstep(oracle_api, clockwise, aa):
    """the fit loop of deodr/examples/triangle_soup_fitting.py:100-184 driven through
    Scene2D.render_compare_and_backward (NumPy drop-in entry points, float64 buffers), 50 iterations.

    aa = 0: in lock-step with the reference's own trajectory, and free-running against the reference's published loss curve
    (tests/golden `aa0_losses50`, last value = the golden of the reference's tests/test_triangle_soup_fitting.py).
    aa = 1: the reference's antialiase_error adjoint carries defect D2, so its published curve is the curve of a wrong
    gradient; lock-step against the REPAIRED reference instead."""
    from hip_util import rel_err

    gt, d = golden_soup(clockwise, "gt_")
    ref = checker(oracle_api, fixed=bool(aa))
    target = ref.render(gt, 1)[0]
    free, _ = golden_soup(clockwise, "init_")  # driven by its own gradients
    lock, _ = golden_soup(clockwise, "init_")  # follows the checker's trajectory
    follow, _ = golden_soup(clockwise, "init_")  # the checker's own run
    speed_free, speed_ref = np.zeros_like(free.ij), np.zeros_like(follow.ij)
    losses_free, losses_ref = [], []
    for it in range(50):
        # checker: one iteration at follow.ij
        follow.clear_gradients()
        if aa:
            image_r, z_r, err_r = ref.render(follow, 1, True, target)
            loss_r = float(np.sum(err_r))
            ref.renderSceneBCpp(follow, 1, image_r, z_r, None, True, target, err_r.copy(), np.ones_like(err_r))
        else:
            image_r, z_r = ref.render(follow, 1)
            loss_r = float(np.sum((image_r - target) ** 2))
            ref.renderSceneBCpp(follow, 1, image_r.copy(), z_r, 2 * (image_r - target))
        # HIP at the same point
        lock.ij = follow.ij.copy()
        image, z, err_buffer, loss = lock.render_compare_and_backward(obs=target, sigma=1, antialiase_error=bool(aa))
        assert abs(loss - loss_r) <= 1e-9 * loss_r, (it, loss, loss_r)
        assert np.abs(image - image_r).max() < 1e-9, it
        assert rel_err(lock.ij_b, follow.ij_b) < 1e-8, it
        assert rel_err(lock.colors_b, follow.colors_b) < 1e-8, it
        losses_ref.append(loss_r)
        speed_ref = 0.80 * speed_ref - follow.ij_b * 0.01
        follow.ij = follow.ij + speed_ref
        # HIP free-running
        _, _, _, loss_f = free.render_compare_and_backward(obs=target, sigma=1, antialiase_error=bool(aa))
        losses_free.append(loss_f)
        speed_free = 0.80 * speed_free - free.ij_b * 0.01
        free.ij = free.ij + speed_free
    if not aa:
        golden = d["aa0_losses50"]
        assert np.array_equal(np.array(losses_ref), golden)  # the checker reproduces the reference's curve bit for bit
        assert np.abs(np.array(losses_free) - golden).max() <= 1e-6 * golden.max(), "free-running fit leaves the reference's loss curve"
    assert abs(losses_free[-1] - losses_ref[-1]) <= 1e-6 * losses_ref[-1]


def test_run_to_run_determinism_bound(oracle_api):
    """The gradient accumulators are float64 atomics: the ORDER of additions varies from run to run, the result may only differ
    by float64 rounding -- frames are bit-identical, gradients within 1e-12 of their scale."""
    from hip_util import device_scene, rel_err
    from deodr_amd.hip_renderer import HipRasterizer

    s = scenes.sphere_scene(size=512, nu=60, n_rings=60)
    ds = device_scene(s, F32)
    r = HipRasterizer.for_scene(ds)
    obs = torch.as_tensor(np.random.RandomState(8).rand(1, 512, 512, 4).astype(np.float32), device=ds.device)
    runs = []
    for _ in range(3):
        image, z, g = r.render_fit(ds, obs, 1.0, check_overflow=True, clear_grads=True)
        torch.cuda.synchronize()
        runs.append((image.clone(), z.clone(), {k: v.clone() for k, v in g.items() if v is not None}))
    for image, z, g in runs[1:]:
        assert torch.equal(image, runs[0][0]) and torch.equal(z, runs[0][1])
        for k, v in g.items():
            assert rel_err(v.cpu().numpy(), runs[0][2][k].cpu().numpy()) < 1e-12, k


def test_invalid_indices_are_reported_not_dereferenced(oracle_api):
    """checkSceneValid (H.h:2700-2712) on a device-resident scene: faces >= V / faces_uv >= Vuv raise the workspace's sticky
    error word in the set-up kernel; the offending triangle is dropped (the frame equals the frame of the scene without it)
    and nothing out of bounds is read.  DeviceScene's own one-time host check is bypassed to reach the kernel."""
    from hip_util import device_scene
    from deodr_amd import hip_renderer as hr
    from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

    s = random_scene(700)
    s.backface_culling = True
    good = device_scene(s, F64)
    r0 = HipRasterizer.for_scene(good)
    assert r0.status(good) == (False, 0, 0)
    for field, bit, what in (("faces", hr.ERR_FACES, "faces"), ("faces_uv", hr.ERR_FACES_UV, "faces_uv")):
        bad = getattr(s, field).copy()
        bad[5, 1] = 4000000000  # far outside any allocation
        kw = dict(faces=s.faces, faces_uv=s.faces_uv)
        kw[field] = bad
        with pytest.raises(ValueError, match=what):  # the host-side check of the constructor
            DeviceScene(kw["faces"], kw["faces_uv"], s.textured, s.shaded, s.uv, s.ij[None], s.depths[None], s.colors[None], s.shade[None],
                        s.edgeflags[None], s.height, s.width, texture=s.texture, background_image=s.background_image[None],
                        clockwise=s.clockwise, pixel_dtype=F64)  # fmt: skip
        ds = DeviceScene(kw["faces"], kw["faces_uv"], s.textured, s.shaded, s.uv, s.ij[None], s.depths[None], s.colors[None], s.shade[None],
                         s.edgeflags[None], s.height, s.width, texture=s.texture, background_image=s.background_image[None],
                         clockwise=s.clockwise, pixel_dtype=F64, validate=False)  # fmt: skip
        r = HipRasterizer.for_scene(ds)
        with pytest.raises(RuntimeError, match="invalid scene"):
            r.render(ds, 1.0)  # first render checks synchronously
        over, need, errs = r.status(ds)
        assert errs == bit and not over
        # the frame that was produced = the scene without triangle 5
        image, z = r.render(ds, 1.0, check_overflow=False)
        keep = np.ones(s.faces.shape[0], dtype=bool)
        keep[5] = False
        import copy

        s2 = copy.copy(s)
        s2.faces, s2.faces_uv, s2.textured, s2.shaded, s2.edgeflags = s.faces[keep], s.faces_uv[keep], s.textured[keep], s.shaded[keep], s.edgeflags[keep]
        img_ref, _ = checker(oracle_api).render(s2, 1.0)
        torch.cuda.synchronize()
        assert np.abs(image[0].cpu().numpy() - img_ref).max() < 1e-9
        # the deferred (asynchronous) check reports it as well, a call or two later
        with pytest.raises(RuntimeError, match="invalid scene"):
            for _ in range(4):
                r.render(ds, 1.0, check_overflow=False)
                torch.cuda.synchronize()
    # textured + shaded triangle in a scene without texture (H.h:2687-2694)
    ds = DeviceScene(s.faces, s.faces_uv, np.ones_like(s.textured), np.ones_like(s.shaded), s.uv, s.ij[None], s.depths[None], s.colors[None],
                     s.shade[None], s.edgeflags[None], s.height, s.width, texture=None, background_image=s.background_image[None],
                     clockwise=s.clockwise, pixel_dtype=F64, validate=False)  # fmt: skip
    r = HipRasterizer.for_scene(ds)
    with pytest.raises(RuntimeError, match="no texture"):
        r.render(ds, 1.0)
    # a texture without texture_b is rejected by the adjoint (H.h:2694)
    g = good.zero_grads()
    g["texture_b"] = None
    r0.render(good, 1.0)
    with pytest.raises(RuntimeError, match="texture_b"):
        r0.render_backward(good, image_b=torch.zeros(1, s.height, s.width, 3), grads=g)


def test_fit_step_under_hip_graph_capture(oracle_api):
    """The calls neither allocate nor synchronise: a whole fit step (set-up, fused forward, edge tiles, finalize) is captured
    in a HIP graph and replayed; the replays give the eager result, also after the inputs changed in place."""
    from hip_util import device_scene, rel_err
    from deodr_amd.hip_renderer import HipRasterizer

    s = scenes.sphere_scene(size=256, nu=40, n_rings=40)
    ds = device_scene(s, F32)
    r = HipRasterizer.for_scene(ds)
    n, H, W, Cc = 1, 256, 256, 4
    obs = torch.as_tensor(np.random.RandomState(9).rand(n, H, W, Cc).astype(np.float32), device=ds.device)
    image = torch.empty((n, H, W, Cc), dtype=F32, device=ds.device)
    z = torch.empty((n, H, W), dtype=F32, device=ds.device)
    grads = ds.zero_grads()
    r.render_fit(ds, obs, 1.0, grads=grads, out=(image, z), check_overflow=True, clear_grads=True)  # sizes the pool, warms up
    torch.cuda.synchronize()
    eager = (image.clone(), {k: v.clone() for k, v in grads.items() if v is not None})
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        r.render_fit(ds, obs, 1.0, grads=grads, out=(image, z), check_overflow=False, clear_grads=True)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        r.render_fit(ds, obs, 1.0, grads=grads, out=(image, z), check_overflow=False, clear_grads=True)
    for _ in range(3):
        image.zero_()
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(image, eager[0])
    for k, v in eager[1].items():
        assert rel_err(grads[k].cpu().numpy(), v.cpu().numpy()) < 1e-12, k
    # new vertex positions written IN PLACE into the captured buffers: the replay renders them
    s2 = scenes.sphere_scene(size=256, nu=40, n_rings=40, angle=0.3)
    ds.ij.copy_(torch.as_tensor(s2.ij[None]))
    ds.depths.copy_(torch.as_tensor(s2.depths[None]))
    ds.colors.copy_(torch.as_tensor(s2.colors[None]))
    ds.edgeflags.copy_(torch.as_tensor(s2.edgeflags[None].astype(np.uint8)))
    graph.replay()
    torch.cuda.synchronize()
    img_ref, z_ref = checker(oracle_api).render(s2, 1.0)
    assert np.abs(image[0].cpu().numpy() - img_ref).max() < 1e-5
    image_b = 2 * (image[0].cpu().numpy().astype(np.float64) - obs[0].cpu().numpy().astype(np.float64))
    g_ref = checker(oracle_api).grads(s2, 1.0, img_ref, z_ref, image_b)
    assert rel_err(grads["ij_b"][0].cpu().numpy(), g_ref["ij_b"]) < 1e-4
    assert not r.status(ds)[0]


def test_two_renders_in_one_autograd_graph(oracle_api):
    """Forward state is shared by every render of a scene: a second forward before the first backward must not make the first
    backward return the gradients of the wrong render (each forward is stamped; a stale stamp rebuilds the state)."""
    from types import SimpleNamespace

    from hip_util import rel_err
    from deodr_amd.pytorch import TorchDifferentiableRender2D

    s, _ = golden_soup(0, "init_")
    scene = SimpleNamespace(scene_2d=s)
    ref = checker(oracle_api)
    rs = np.random.RandomState(12)
    ij_a = torch.tensor(s.ij, dtype=torch.float64, device="cuda", requires_grad=True)
    ij_b = torch.tensor(s.ij + rs.randn(*s.ij.shape), dtype=torch.float64, device="cuda", requires_grad=True)
    col = torch.tensor(s.colors, dtype=torch.float64, device="cuda", requires_grad=True)
    w_a, w_b = rs.rand(s.height, s.width, 3), rs.rand(s.height, s.width, 3)
    img_a = TorchDifferentiableRender2D(ij_a, col, scene)
    img_b = TorchDifferentiableRender2D(ij_b, col, scene)  # overwrites the shared forward state
    loss = (img_a * torch.as_tensor(w_a, device="cuda")).sum() + (img_b * torch.as_tensor(w_b, device="cuda")).sum()
    loss.backward()
    import copy

    col_ref = 0.0
    for ij, w, got in ((ij_a, w_a, ij_a.grad), (ij_b, w_b, ij_b.grad)):
        s1 = copy.copy(s)
        s1.ij = ij.detach().cpu().numpy()
        image, z = ref.render(s1, 1)
        g = ref.grads(s1, 1, image, z, w)
        assert rel_err(got.cpu().numpy(), g["ij_b"]) < 1e-8
        col_ref = col_ref + g["colors_b"]
    assert rel_err(col.grad.cpu().numpy(), col_ref) < 1e-8


def test_replaced_background_and_texture_are_uploaded(oracle_api):
    """The cached device copy of a Scene2D follows replaced arrays: new `faces` OBJECT with equal content (what Scene3D.render
    produces every frame) keeps the device state; a new background / texture object is rendered, not the stale one."""
    from types import SimpleNamespace

    from deodr_amd.pytorch import TorchDifferentiableRender2D

    s, _ = golden_soup(0, "init_")
    scene = SimpleNamespace(scene_2d=s)
    ref = checker(oracle_api)
    ij = torch.tensor(s.ij, dtype=torch.float64, device="cuda")
    col = torch.tensor(s.colors, dtype=torch.float64, device="cuda")
    img0 = TorchDifferentiableRender2D(ij, col, scene)
    state0 = scene._hip_state["ds"], scene._hip_state["r"]
    assert np.abs(img0.cpu().numpy() - ref.render(s, 1)[0]).max() < 1e-9
    s.faces = s.faces.copy()  # same content, new object
    s.background_image = np.ascontiguousarray(s.background_image[::-1] * 0.5)
    s.texture = np.ascontiguousarray(s.texture[:, ::-1] * 0.9)
    img1 = TorchDifferentiableRender2D(ij, col, scene)
    assert (scene._hip_state["ds"], scene._hip_state["r"]) == state0, "equal topology must not rebuild the device scene"
    assert np.abs(img1.cpu().numpy() - ref.render(s, 1)[0]).max() < 1e-9
    s.faces = s.faces[::-1].copy()  # different topology: rebuilt, workspace kept
    s.faces_uv = s.faces_uv[::-1].copy()
    s.textured, s.shaded, s.edgeflags = s.textured[::-1].copy(), s.shaded[::-1].copy(), s.edgeflags[::-1].copy()
    img2 = TorchDifferentiableRender2D(ij, col, scene)
    assert scene._hip_state["ds"] is not state0[0] and scene._hip_state["r"] is state0[1]
    assert np.abs(img2.cpu().numpy() - ref.render(s, 1)[0]).max() < 1e-9


def test_spill_pool_overflow_later_in_a_fit_is_detected(oracle_api):
    """A scene whose tile lists start to spill AFTER the first (checked) render: the asynchronous poll of the status block
    finds the overflow a call or two later, regrows the workspace and says so instead of returning incomplete frames for ever."""
    from hip_util import device_scene
    from deodr_amd.hip_renderer import HipRasterizer

    import copy

    big = scenes.soup_scene(n_tri=700, width=64, height=64, seed=9, min_area=600.0)  # large triangles: every tile list (64 inline slots) spills
    small = copy.copy(big)  # the same triangles shrunk about their centroids, no silhouette edges: nothing spills
    tri = big.ij.reshape(-1, 3, 2)
    small.ij = (tri.mean(axis=1, keepdims=True) + 0.1 * (tri - tri.mean(axis=1, keepdims=True))).reshape(-1, 2)
    small.edgeflags = np.zeros_like(big.edgeflags)
    ds = device_scene(small, F64)
    r = HipRasterizer.for_scene(ds, pool_pairs=16)
    r.render(ds, 1.0)  # synchronous first check
    assert r.pool_pairs == 16
    ds.set_views(ij=big.ij[None], depths=big.depths[None], colors=big.colors[None], shade=big.shade[None], edgeflags=big.edgeflags[None])
    with pytest.raises(RuntimeError, match="overflowed"):
        for _ in range(6):
            r.render(ds, 1.0)
            torch.cuda.synchronize()
    assert r.pool_pairs > 16
    ref = checker(oracle_api).render(big, 1.0)
    image, z = r.render(ds, 1.0)  # regrown: checked again, complete
    assert np.abs(image[0].cpu().numpy() - ref[0]).max() < 1e-9


def test_backward_after_fit_step_rebuilds_the_owner_buffer(oracle_api):
    """The fused forward does not keep owner ids of the tiles it back-propagated through: a two-call adjoint that follows a
    fit step on the same workspace must rebuild the forward state (and does, without being told)."""
    from hip_util import device_scene, rel_err
    from deodr_amd import hip_renderer as hr
    from deodr_amd.hip_renderer import HipRasterizer
    import ctypes as C

    s = random_scene(800)
    s.backface_culling = True
    ds = device_scene(s, F64)
    r = HipRasterizer.for_scene(ds)
    obs = torch.as_tensor(np.random.RandomState(2).rand(1, s.height, s.width, 3), device=ds.device)
    image, z, g_fit = r.render_fit(ds, obs, 1.0, check_overflow=True)
    g_two = r.render_backward(ds, residual_obs=obs)  # python layer: knows the last forward was fused
    # straight through the C ABI, claiming to have the forward state
    grads = ds.zero_grads()
    sc = ds.c_struct(grads)
    hr._check(hr.lib().deodr_hip_render_scene_b(C.byref(sc), hr._ptr(image), None, None, 1.0, 0, hr._ptr(obs.contiguous()), None, None,
                                                hr._ptr(r.workspace), r.nbytes, 1, hr._stream()))  # fmt: skip
    torch.cuda.synchronize()
    for k in ("ij_b", "colors_b", "uv_b", "shade_b", "texture_b"):
        assert rel_err(g_two[k].cpu().numpy(), g_fit[k].cpu().numpy()) < 1e-9, k
        assert rel_err(grads[k].cpu().numpy(), g_fit[k].cpu().numpy()) < 1e-9, k


@pytest.mark.parametrize("dt", [F32, F64])
@pytest.mark.parametrize("sigma", [0.0, 1.0])
@pytest.mark.parametrize("size,nb_colors,bg_image", [((64, 96), 3, False), ((37, 53), 3, True), ((40, 64), 1, False), ((72, 56), 4, True)])
def test_fit_step_background_of_empty_tiles(oracle_api, size, nb_colors, bg_image, sigma, dt):
    """The background fill of a fit step rides on the adjoint's kernels (edge tiles + finalize; finalize alone when sigma = 0) and
    writes runs of empty tiles in 16-byte pieces: every channel count / pixel type / ragged width, against the frame of the
    forward-only call (fill kernel on the forked stream) and against the checker."""
    from hip_util import device_scene

    H, W = size
    s = scenes.soup_scene(n_tri=6, width=W, height=H, seed=3, flat=False, min_area=30.0)
    rs = np.random.RandomState(5)
    V = s.depths.shape[0]
    s.colors = rs.rand(V, nb_colors)
    s.colors_b = np.zeros_like(s.colors)
    s.nb_colors = nb_colors
    s.texture = np.zeros((0, 0, nb_colors))
    s.texture_b = np.zeros((0, 0, nb_colors))
    s.textured[:] = False
    s.shaded[:] = False
    if bg_image:
        s.background_image, s.background_color = rs.rand(H, W, nb_colors), None
    else:
        s.background_image, s.background_color = None, rs.rand(nb_colors)
    ds = device_scene([s], dt)
    from deodr_amd.hip_renderer import HipRasterizer

    r = HipRasterizer.for_scene(ds)
    obs = torch.as_tensor(rs.rand(1, H, W, nb_colors), device=ds.device, dtype=dt)
    image = torch.full((1, H, W, nb_colors), 777.0, dtype=dt, device=ds.device)
    z = torch.full((1, H, W), 777.0, dtype=dt, device=ds.device)
    r.render_fit(ds, obs, sigma, out=(image, z), check_overflow=True, clear_grads=True)
    image2, z2 = r.render(ds, sigma)
    torch.cuda.synchronize()
    assert torch.equal(image, image2) and torch.equal(z, z2)
    img_ref, z_ref = checker(oracle_api).render(s, sigma)
    assert np.abs(image[0].cpu().numpy() - img_ref).max() < TOL[dt][0]
    assert np.array_equal(np.isinf(z[0].cpu().numpy()), np.isinf(z_ref))


def test_scene_without_triangles_is_rejected_like_the_reference(oracle_api):
    """T = 0 means NULL face arrays: checkSceneValid (H.h:2667-2680) refuses them, and so does the C ABI -- with an error, not a crash"""
    from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

    H, W, C = 48, 80, 3
    ds = DeviceScene(np.zeros((0, 3), dtype=np.int64), np.zeros((0, 3), dtype=np.int64), np.zeros(0), np.zeros(0), np.zeros((1, 2)), np.zeros((1, 1, 2)),
                     np.ones((1, 1)), np.zeros((1, 1, C)), np.zeros((1, 1)), np.zeros((1, 0, 3)), H, W, background_color=np.zeros(C), pixel_dtype=F32)  # fmt: skip
    r = HipRasterizer.for_scene(ds)
    with pytest.raises(RuntimeError, match="NULL"):
        r.render(ds, 1.0)


@pytest.mark.parametrize("size", [(4096, 4096), (24, 16384)])
def test_large_frames(oracle_api, size):
    """4096 x 4096 (262 144 tiles, 8 192 bitmap words) and a 16 384-pixel-wide strip: forward, fit step and adjoint against the checker"""
    from hip_util import device_scene, rel_err
    from deodr_amd.hip_renderer import HipRasterizer

    H, W = size
    s = scenes.soup_scene(n_tri=24, width=W, height=H, seed=9, flat=False)
    ds = device_scene([s], F32)
    r = HipRasterizer.for_scene(ds)
    rs = np.random.RandomState(2)
    obs = torch.as_tensor(rs.rand(1, H, W, 3), device=ds.device, dtype=F32)
    image, z, g = r.render_fit(ds, obs, 1.0, check_overflow=True, clear_grads=True)
    image2, z2 = r.render(ds, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(image, image2) and torch.equal(z, z2)
    ref = checker(oracle_api)
    img_ref, z_ref = ref.render(s, 1.0)
    assert np.abs(image[0].cpu().numpy() - img_ref).max() < 1e-5
    res_b = 2 * (image[0].cpu().numpy().astype(np.float64) - obs[0].cpu().numpy().astype(np.float64))
    g_ref = ref.grads(s, 1.0, img_ref, z_ref, res_b)
    for k in ("ij_b", "colors_b"):
        assert rel_err(g[k][0].cpu().numpy(), g_ref[k]) < 1e-4, k


def test_two_call_path_under_hip_graph_capture(oracle_api):
    """render + render_backward captured in one HIP graph: the forward-only call forks its background fill onto the library's
    side stream and joins it back (event record / wait), which stream capture turns into edges of the graph"""
    from hip_util import device_scene, rel_err
    from deodr_amd.hip_renderer import HipRasterizer

    s = scenes.sphere_scene(size=256, nu=40, n_rings=40)
    ds = device_scene(s, F32)
    r = HipRasterizer.for_scene(ds)
    n, H, W, Cc = 1, 256, 256, 4
    image_b = torch.as_tensor(np.random.RandomState(4).randn(n, H, W, Cc).astype(np.float32), device=ds.device)
    image = torch.empty((n, H, W, Cc), dtype=F32, device=ds.device)
    z = torch.empty((n, H, W), dtype=F32, device=ds.device)
    grads = ds.zero_grads()

    def step():
        for v in grads.values():
            if v is not None:
                v.zero_()
        r.render(ds, 1.0, out=(image, z), check_overflow=False)
        r.render_backward(ds, image_b=image_b, grads=grads)

    r.render(ds, 1.0, out=(image, z), check_overflow=True)
    step()
    torch.cuda.synchronize()
    eager = (image.clone(), z.clone(), {k: v.clone() for k, v in grads.items() if v is not None})
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        step()
    for _ in range(3):
        image.zero_()
        z.zero_()
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(image, eager[0]) and torch.equal(z, eager[1])
    for k, v in eager[2].items():
        assert rel_err(grads[k].cpu().numpy(), v.cpu().numpy()) < 1e-12, k


@pytest.mark.parametrize("seed,n_tri,size", [(7, 60, (32, 40)), (32007, 150, (200, 96)), (27007, 400, (24, 96))])
def test_non_strict_rule_draws_the_clamped_column(oracle_api, family, seed, n_tri, size):
    """strict_edge=False: the reference clamps the left end of a row to x_max (ceil_div, H.h:895), so rows whose span lies
    between x_max and the rightmost vertex -- or beyond the right border of the frame -- draw the pixel of column x_max although
    it is outside the left edge.  The tile binning must keep that column (found by tests/fuzz_parity.py: scenes 27 and 32)."""
    s = scenes.soup_scene(n_tri=n_tri, width=size[1], height=size[0], seed=seed, clockwise=bool(seed & 1), textured_ratio=0.5, flat=False,
                          texture_size=8, min_area=30.0)
    s.strict_edge = False
    compare_all(oracle_api, s, 1.0, F64)
