"""GPU tests added in round 4 (through the C ABI, like the others):

* the deterministic mode (deodr_hip_set_deterministic, SURVEY.md section 7 "provide a deterministic two-stage mode for tests"): every
  gradient -- texture gradient and loss included -- BIT-IDENTICAL from run to run, within the usual tolerance of the checker, on a
  textured soup, an 8-view mesh batch, the antialiase_error variant; a 30-iteration fit run twice, bit for bit;
* finalize_kernel with and without its per-workgroup vertex table (launches on either side of the size threshold) against the checker;
* a tile of many silhouette edges listed once per part of its edges (parts of 8 and of 16), against the checker and the two-call path;
* the step-done flag + deodr_hip_wait_flag (a consumer on another stream), deodr_hip_views_gradient_sum against autograd and the pose adjoint;
* a camera shared by several views (`DeviceCamera` expands every array per view);
* the library's streaming-copy probe moves the bytes it says.
"""

import numpy as np
import pytest
import torch

from deodr_amd import scenes
from test_oracle import random_scene

pytestmark = pytest.mark.gpu

F32, F64 = torch.float32, torch.float64


def checker(api, fixed=False):
    return api.ref(fixed=fixed) or api.port(fixed=fixed)


@pytest.fixture
def deterministic():
    from deodr_amd import hip_renderer as hr

    hr.set_deterministic(True)
    yield
    hr.set_deterministic(False)


def _clone(g):
    return {k: v.clone() for k, v in g.items() if v is not None}


def _all_equal(a, b):
    return all(torch.equal(a[k], b[k]) for k in a)


@pytest.mark.parametrize("dt", [F32, F64])
def test_deterministic_mode_textured_soup_bit_identical_and_right(oracle_api, deterministic, dt):
    """two-call path (render + render_backward) and the fit step, mixed textured / untextured soup with silhouette edges everywhere"""
    from hip_util import device_scene, rel_err
    from deodr_amd.hip_renderer import HipRasterizer

    s = random_scene(4100)  # 24 slanted triangles, half of them textured, every edge flagged
    s.backface_culling = True
    ds = device_scene(s, dt)
    r = HipRasterizer.for_scene(ds)
    rng = np.random.RandomState(3)
    image_b = torch.as_tensor(rng.randn(1, s.height, s.width, s.nb_colors), device=ds.device).to(dt)
    obs = torch.as_tensor(rng.rand(1, s.height, s.width, s.nb_colors), device=ds.device).to(dt)
    runs, fits = [], []
    for _ in range(3):
        image, z = r.render(ds, 1.0, check_overflow=True)
        runs.append(_clone(r.render_backward(ds, image_b=image_b)))
        image_f, z_f, g = r.render_fit(ds, obs, 1.0, check_overflow=True, clear_grads=True)
        fits.append(_clone(g))
        torch.cuda.synchronize()
    assert all(_all_equal(runs[0], x) for x in runs[1:]), "two-call gradients differ from run to run in the deterministic mode"
    assert all(_all_equal(fits[0], x) for x in fits[1:]), "fit-step gradients differ from run to run in the deterministic mode"
    ref = checker(oracle_api, fixed=True)
    img_ref, z_ref = ref.render(s, 1.0)
    g_ref = ref.grads(s, 1.0, img_ref, z_ref, image_b[0].cpu().numpy().astype(np.float64))
    tol = 1e-4 if dt == F32 else 1e-6  # (contributions are rounded to 2^-32: not the 1e-8 of the floating-point path)
    for k in ("ij_b", "colors_b", "uv_b", "shade_b", "texture_b"):
        got = runs[0][k].cpu().numpy()
        got = got[0] if k in ("ij_b", "colors_b", "shade_b") else got
        assert rel_err(got, g_ref[k]) < tol, k


def test_deterministic_mode_eight_views_and_error_buffer(oracle_api, deterministic):
    from hip_util import device_scene
    from deodr_amd.hip_renderer import HipRasterizer

    views = [scenes.sphere_scene(size=256, nu=40, n_rings=40, angle=float(a), textured=True, texture_size=64, nb_colors=3) for a in np.linspace(-0.4, 0.4, 8)]
    ds = device_scene(views, F32)
    r = HipRasterizer.for_scene(ds)
    obs = torch.as_tensor(np.random.RandomState(5).rand(8, 256, 256, 3).astype(np.float32), device=ds.device)
    fits = []
    for _ in range(3):
        _, _, g = r.render_fit(ds, obs, 1.0, check_overflow=True, clear_grads=True)
        fits.append(_clone(g))
    assert all(_all_equal(fits[0], x) for x in fits[1:])
    assert float(fits[0]["texture_b"].abs().max()) > 0 and float(fits[0]["uv_b"].abs().max()) > 0
    # antialiase_error: forward err buffer, adjoint from err_buffer_b
    errs = []
    for _ in range(2):
        image, z, err = r.render(ds, 1.0, True, obs, check_overflow=True)
        errs.append(_clone(r.render_backward(ds, err_buffer_b=torch.ones_like(err))))
    assert _all_equal(errs[0], errs[1])


def test_deterministic_mode_fit_twice_bit_for_bit(oracle_api, deterministic):
    """a 30-iteration soup fit through Scene2D.render_compare_and_backward (the loop of deodr/examples/triangle_soup_fitting.py:100-184):
    losses and final vertices identical, bit for bit, in two runs -- as the single-threaded reference is"""
    from conftest import golden_soup

    def run():
        gt, _ = golden_soup(0, "gt_")
        target = checker(oracle_api).render(gt, 1)[0]
        scene, _ = golden_soup(0, "init_")
        speed = np.zeros_like(scene.ij)
        losses = []
        for _ in range(30):
            scene.clear_gradients()
            _, _, _, loss = scene.render_compare_and_backward(obs=target, sigma=1, antialiase_error=False)
            losses.append(loss)
            speed = 0.80 * speed - scene.ij_b * 0.01
            scene.ij = scene.ij + speed
        return np.array(losses), np.array(scene.ij)

    l1, ij1 = run()
    l2, ij2 = run()
    assert np.array_equal(l1, l2) and np.array_equal(ij1, ij2)
    assert l1[-1] < 0.6 * l1[0]  # (and it does fit: 4186 -> 2116 in 30 iterations)


@pytest.mark.parametrize("n_views", [1, 3])
def test_finalize_with_and_without_vertex_table(oracle_api, n_views):
    """20 000 triangles: one view stays below the size at which finalize_kernel merges vertex adjoints in its LDS table, three views
    are above it -- same gradients (to rounding) as the checker's, view by view"""
    from hip_util import device_scene, rel_err
    from deodr_amd.hip_renderer import HipRasterizer

    views = [scenes.sphere_scene(size=256, angle=float(a)) for a in np.linspace(-0.3, 0.3, n_views)]
    ds = device_scene(views, F64)
    r = HipRasterizer.for_scene(ds)
    obs = torch.as_tensor(np.random.RandomState(6).rand(n_views, 256, 256, 4), device=ds.device)
    image, z, g = r.render_fit(ds, obs, 1.0, check_overflow=True, clear_grads=True)
    torch.cuda.synchronize()
    ref = checker(oracle_api)
    for i, s in enumerate(views):
        img_ref, z_ref = ref.render(s, 1.0)
        g_ref = ref.grads(s, 1.0, img_ref, z_ref, 2 * (img_ref - obs[i].cpu().numpy()))
        for k in ("ij_b", "colors_b"):
            assert rel_err(g[k][i].cpu().numpy(), g_ref[k]) < 1e-8, (k, i)


def test_camera_shared_by_the_views_is_expanded():
    """extrinsic [n,3,4] with ONE intrinsic [3,3] (and the other way round): the fused projection indexes every array per view"""
    from deodr_amd.scene3d import DeviceCamera

    rng = np.random.RandomState(0)
    pts = torch.as_tensor(rng.rand(50, 3) + np.array([0, 0, 4.0]), device="cuda")
    K = np.array([[300.0, 0, 64], [0, 300.0, 48], [0, 0, 1]])
    E = np.stack([np.column_stack((np.eye(3), np.array([0.1 * i, 0, 0]))) for i in range(4)])
    shared = DeviceCamera(E, K, 96, 128)  # one intrinsic for four views
    full = DeviceCamera(E, np.stack([K] * 4), 96, 128)
    ij_a, d_a = shared.project_points(pts)
    ij_b, d_b = full.project_points(pts)
    assert ij_a.shape == (4, 50, 2) and torch.equal(ij_a, ij_b) and torch.equal(d_a, d_b)
    one_e = DeviceCamera(E[0], np.stack([K, 2 * K]), 96, 128)  # one extrinsic for two intrinsics
    ij_c, _ = one_e.project_points(pts)
    assert ij_c.shape == (2, 50, 2) and torch.allclose(ij_c[0], ij_a[0])
    with pytest.raises(ValueError):
        DeviceCamera(E, np.stack([K] * 3), 96, 128)


def test_copy_probe_copies():
    from deodr_amd import hip_renderer as hr

    a = torch.arange(1 << 20, dtype=torch.int32, device="cuda")
    b = torch.zeros_like(a)
    st = torch.cuda.current_stream().cuda_stream
    assert hr.lib().deodr_hip_copy_probe(b.data_ptr(), a.data_ptr(), a.numel() * 4, 0, 1, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert hr.lib().deodr_hip_copy_probe(b.data_ptr(), a.data_ptr(), 24, 0, 1, st) != 0  # not a multiple of 16


def _cameras(n, rng):
    from deodr_amd.scene3d import DeviceCamera

    K = np.array([[300.0, 0, 64], [0, 310.0, 48], [0, 0, 1]])
    E = np.stack([np.column_stack((scenes.roty(0.2 * i) @ scenes.rotx(0.1), np.array([0.05 * i, 0.02, 4.0 + 0.1 * i]))) for i in range(n)])
    dist = rng.randn(n, 5) * 0.01
    return DeviceCamera(E, np.stack([K] * n), 96, 128, dist, torch.device("cuda"))


@pytest.mark.parametrize("n", [1, 5, 8])
def test_views_gradient_sum_is_the_sum_of_the_camera_adjoints(n, monkeypatch):
    """deodr_hip_views_gradient_sum (what a sharded multi-view fit all-reduces: mesh_fitter.py:518-527's `vertices_b += ...` in one launch)
    against autograd through the torch formulas of the projection, view by view, and against the pose adjoint with the identity pose"""
    from deodr_amd import fronthalf

    rng = np.random.RandomState(5)
    V, Cc = 333, 3
    cam = _cameras(n, rng)
    posed = torch.as_tensor(rng.rand(n, V, 3) - 0.5, device="cuda")
    ij_b = torch.as_tensor(rng.randn(n, V, 2), device="cuda")
    depths_b = torch.as_tensor(rng.randn(n, V), device="cuda")
    colors_b = torch.as_tensor(rng.randn(n, V, Cc), device="cuda")
    vb, cs = torch.zeros(V, 3, dtype=F64, device="cuda"), torch.zeros(V, Cc, dtype=F64, device="cuda")
    fronthalf.views_gradient_sum(posed, cam, ij_b, vb, depths_b=depths_b, depths_b_scale=0.7, colors_b=colors_b, colors_sum=cs)
    p = posed.clone().requires_grad_(True)
    monkeypatch.setattr(fronthalf, "usable", lambda *a: False)  # (DeviceCamera.project_points: the torch formulas, not the fused kernel)
    ij, d = cam.project_points(p)
    monkeypatch.undo()
    ((ij * ij_b).sum() + 0.7 * (d * depths_b).sum()).backward()
    assert torch.allclose(vb, p.grad.sum(0), rtol=1e-10, atol=1e-10)
    assert torch.allclose(cs, colors_b.sum(0), rtol=1e-13, atol=1e-13)
    # the pose adjoint with the identity pose computes the same sums (bit for bit: same formulas, same order over the views)
    out, vb2, cs2 = torch.zeros(3 + 7 * n, dtype=F64, device="cuda"), torch.zeros_like(vb), torch.zeros_like(cs)
    ident = torch.tensor([[0.0, 0.0, 0.0, 1.0]] * n, dtype=F64, device="cuda")
    fronthalf.fit_pose_project_b(posed[0].contiguous(), ident, posed, cam, None, ij_b, depths_b, vb2, out, fronthalf.fit_scratch(V, n, torch.device("cuda")),
                                 depths_b_scale=0.7, colors_b=colors_b, colors_sum=cs2)  # fmt: skip
    assert torch.equal(vb, vb2) and torch.equal(cs, cs2)


@pytest.mark.parametrize("n_views", [1, 8])
def test_step_done_flag_and_wait_flag(n_views):
    """DeodrHipFitOptions::done_flag: the step stores the value when its gradients are complete; a kernel on ANOTHER stream that was queued
    behind deodr_hip_wait_flag then reads the finished gradients (no event between the streams).  One view: finalize_kernel without the vertex
    table; eight: with it.  A wait for a value that never comes gives up and says so."""
    from hip_util import device_scene
    from deodr_amd import hip_renderer as hr
    from deodr_amd.hip_renderer import HipRasterizer

    views = [scenes.sphere_scene(size=512, angle=float(a)) for a in np.linspace(-0.3, 0.3, n_views)]
    ds = device_scene(views, F64)
    r = HipRasterizer.for_scene(ds)
    obs = torch.as_tensor(np.random.RandomState(6).rand(n_views, 512, 512, 4), device=ds.device)
    image, z, g_ref = r.render_fit(ds, obs, 1.0, check_overflow=True, clear_grads=True)
    torch.cuda.synchronize()
    ref = g_ref["ij_b"].clone()
    flag = torch.zeros(1, dtype=torch.int32, device=ds.device)
    status = torch.zeros(1, dtype=torch.int32, device=ds.device)
    side = torch.cuda.Stream()
    g = ds.zero_grads()
    copies = []
    for step in range(1, 21):
        r.render_fit(ds, obs, 1.0, grads=g, clear_grads=True, check_overflow=False, done_flag=(flag, step))
        with torch.cuda.stream(side):
            hr.wait_flag(flag, step, status=status, timeout=5.0)
            copies.append(g["ij_b"].clone())  # (on the side stream: ordered behind the wait only)
        side.synchronize()  # (the next step clears the gradients: the copy must have been taken)
    torch.cuda.synchronize()
    assert int(flag.item()) == 20 and int(status.item()) == 0
    for c in copies:  # every copy was taken from complete gradients (atomics: equal to rounding, run to run)
        assert torch.allclose(c, ref, rtol=1e-9, atol=1e-9 * float(ref.abs().max()))
    # the fallback (a one-thread kernel behind everything) where finalize_kernel is not the step's last kernel: the deterministic mode
    hr.set_deterministic(True)
    try:
        r.render_fit(ds, obs, 1.0, grads=g, clear_grads=True, check_overflow=False, done_flag=(flag, 21))
        with torch.cuda.stream(side):
            hr.wait_flag(flag, 21, status=status, timeout=5.0)
            det = g["ij_b"].clone()
        torch.cuda.synchronize()
    finally:
        hr.set_deterministic(False)
    assert int(flag.item()) == 21 and int(status.item()) == 0
    assert float((det - ref).abs().max() / ref.abs().max()) < 1e-4  # (fixed point: 2^-32 per contribution, through the 3 x 3 inverses of finalize)
    # a value nobody stores: the wait gives up after its timeout and raises the status word; later waits on that word return at once
    hr.wait_flag(flag, 1000, status=status, timeout=0.05)
    hr.wait_flag(flag, 1001, status=status, timeout=30.0)
    torch.cuda.synchronize()
    assert int(status.item()) == 1


@pytest.mark.parametrize("n_views", [1, 9])
def test_tiles_of_many_edges_are_split_into_parts(oracle_api, n_views):
    """A fit step lists a tile of 17 .. 128 silhouette edges once per part of its edges (8 per part for launches of about one dispatch round,
    16 for larger ones), every copy back-propagating its own edges from a colour snapshot and the product of the later transparencies:
    a 512^2 frame (a grid with a head of the list) with ~60 edges crowded into one tile, against the two-call path and the checker;
    and the head of the work list really holds more entries per view with parts of 8 than with parts of 16."""
    from test_hip_parity import compare_fit_step
    from test_hip_parity2 import crowded_scene
    from hip_util import device_scene
    from deodr_amd.hip_renderer import HipRasterizer

    views = [crowded_scene(20, seed=3 + i, size=512) for i in range(n_views)]
    for v in views:
        v.texture = np.zeros((0, 0))  # (an untextured scene: the forward raster of a fit step back-propagates the tiles with edges itself)
    compare_fit_step(oracle_api, views, 1.0, F64)
    if n_views == 1:
        compare_fit_step(oracle_api, views, 2.5, F32)
        heads = {}
        for n in (1, 9):  # the same view n times: n = 9 makes the launch large enough for parts of 16
            ds = device_scene([views[0]] * n, F64)
            r = HipRasterizer.for_scene(ds)
            obs = torch.zeros((n, 512, 512, views[0].nb_colors), dtype=F64, device=ds.device)
            r.render_fit(ds, obs, 1.0, check_overflow=True, clear_grads=True)
            torch.cuda.synchronize()
            heads[n] = int(r.workspace[:64].view(torch.int32).cpu().numpy()[13])  # WsHeader::work_count[0] of view 0: the head of the list, copies included
        assert heads[1] > heads[9] > 0, heads


@pytest.mark.parametrize("n_tri,size", [(70, 512), (70, 24), (20, 24)])
def test_crowded_tiles_of_an_untextured_fit_step(oracle_api, n_tri, size):
    """the crowded tiles of test_many_edges_in_one_tile in an UNTEXTURED scene, where the forward raster of a fit step back-propagates them
    itself: ~200 edges in a tile (more than the staged sweep orders: the un-staged tile code inside the forward raster) in a grid with a
    head of the list and in a tiny frame, ~60 edges in a tiny frame (never split: one list)"""
    from test_hip_parity import compare_fit_step
    from test_hip_parity2 import crowded_scene

    s = crowded_scene(n_tri, seed=5, size=size)
    s.texture = np.zeros((0, 0))
    compare_fit_step(oracle_api, s, 1.0, F64)
    compare_fit_step(oracle_api, s, 1.0, F32)


def test_overlapped_views_reduction_equals_a_synchronous_one():
    """deodr_amd.distributed.OverlappedViewsReduction (what bench.py --gpus N drives): the shared gradient of every step, reduced on the
    communication stream behind the step-done flag while the next step renders, equals the same sums formed synchronously from that
    step's gradient arrays (one process: no collective; the two-rank collective is test_bench_two_ranks_on_one_gpu's)"""
    from hip_util import device_scene
    from deodr_amd import fronthalf
    from deodr_amd.distributed import OverlappedViewsReduction
    from deodr_amd.hip_renderer import HipRasterizer
    from deodr_amd.scene3d import DeviceCamera

    n, S = 4, 256
    poses = np.linspace(-0.3, 0.3, n)
    views = [scenes.sphere_scene(size=S, nu=40, n_rings=40, angle=float(a)) for a in poses]
    verts, _f = scenes.bumpy_sphere(40, 40)
    cams = [scenes.fit_camera(S, S, 60.0, verts, scenes.rotx(0.37) @ scenes.roty(0.23 + float(a))) for a in poses]
    ds = device_scene(views, F64)
    r = HipRasterizer.for_scene(ds)
    camera = DeviceCamera(np.stack([c.extrinsic for c in cams]), np.stack([c.intrinsic for c in cams]), S, S, None, ds.device)
    posed = torch.as_tensor(np.ascontiguousarray(verts), device=ds.device)[None].expand(n, -1, -1).contiguous()
    V, C = posed.shape[1], ds.nb_colors
    red = OverlappedViewsReduction(ds, camera, posed)
    rng = np.random.RandomState(4)
    r.render(ds, 1.0, check_overflow=True)
    expected, slots = [], []
    for step in range(6):
        obs = torch.as_tensor(rng.rand(n, S, S, C), device=ds.device)
        slot = red.begin()
        r.render_fit(ds, obs, 1.0, grads=slot.grads, clear_grads=True, check_overflow=False, done_flag=slot.done_flag)
        red.reduce(slot)
        if step >= 4:  # the last two steps: their sets are not rendered into again
            slots.append(slot)
    red.finish()
    for slot in slots:
        vb, cs = torch.zeros(V, 3, dtype=F64, device=ds.device), torch.zeros(V, C, dtype=F64, device=ds.device)
        fronthalf.views_gradient_sum(posed, camera, slot.grads["ij_b"], vb, colors_b=slot.grads["colors_b"], colors_sum=cs)
        torch.cuda.synchronize()
        assert float(vb.abs().max()) > 0
        assert torch.equal(slot.vertices_b, vb) and torch.equal(slot.colors_b, cs)
