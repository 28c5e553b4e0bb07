"""CPU tests of the Python HOST LAYER (deodr_amd/hip_renderer.py, the torch operator, the NumPy drop-ins, Scene3DDevice's scene
assembly) with tests/fake_hip.py standing in for libdeodr_hip.so: marshalling into ``DeodrHipScene``, accumulate-into / clear
semantics, gradient rebinding of the drop-in entry points, workspace caching, generation stamps of the forward state, exceptions.

The pixels and gradients come from the CPU checker here, so these tests say nothing about the kernels -- the `-m gpu` suite does
that through the same Python code.  What they do is keep the host logic under test on machines without a GPU."""

import numpy as np
import pytest
import torch

import fake_hip
from conftest import golden_soup
from deodr_amd import scenes


@pytest.fixture
def fake(oracle_api):
    with fake_hip.emulate(oracle_api.ref() or oracle_api.port(), oracle_api.ref(fixed=True) or oracle_api.port(fixed=True)) as lib:
        yield lib


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_the_emulation_is_scoped(oracle_api):
    """outside `emulate()` the product is untouched: no CPU path, the real library"""
    from deodr_amd import hip_renderer as hr

    with fake_hip.emulate(oracle_api.port(), oracle_api.port(fixed=True)):
        assert hr._resolve_device("cuda") == torch.device("cpu")
    with pytest.raises(RuntimeError, match="no CPU path"):
        hr._resolve_device("cpu")
    assert not isinstance(hr.lib(), fake_hip.FakeLib)


def test_scene2d_dropin_soup_fit_follows_reference_losses(fake, oracle_api):
    """Scene2D.render_compare_and_backward through renderSceneCpp / renderSceneBCpp (upload, in-place outputs, `scene.x_b = old + new`
    rebinding, clear_gradients): 12 iterations of the reference's soup fit reproduce its published loss curve"""
    for clockwise in (0, 1):
        gt, d = golden_soup(clockwise, "gt_")
        target = (oracle_api.ref() or oracle_api.port()).render(gt, 1)[0]
        s, _ = golden_soup(clockwise, "init_")
        speed, losses = np.zeros_like(s.ij), []
        for _ in range(12):
            _, _, _, loss = s.render_compare_and_backward(obs=target, sigma=1, antialiase_error=False)
            losses.append(loss)
            speed = 0.80 * speed - s.ij_b * 0.01
            s.ij = s.ij + speed
        assert np.abs(np.array(losses) - d["aa0_losses50"][:12]).max() <= 1e-12 * d["aa0_losses50"][0]
    assert fake.calls["render_scene"] == 24 and fake.calls["render_scene_b"] == 24


def test_dropin_backward_accumulates_and_checks(fake, oracle_api):
    """renderSceneBCpp adds to the scene's gradient arrays (pyx:406-410), render_backward wants culling and no perspective correction"""
    s = scenes.soup_scene(n_tri=12, width=40, height=32, seed=4, flat=False, min_area=20.0)
    image, z = s.render(1.0)
    image_b = np.random.RandomState(0).randn(*image.shape)
    s.render_backward(image_b)
    once = s.ij_b.copy()
    s.render_backward(image_b)
    assert rel(s.ij_b, 2 * once) < 1e-14
    s.clear_gradients()
    assert not s.ij_b.any()
    s.backface_culling = False
    with pytest.raises(BaseException, match="backface_culling=True"):
        s.render_backward(image_b)
    s.backface_culling, s.perspective_correct = True, True
    with pytest.raises(BaseException, match="perspective_correct"):
        s.render_backward(image_b)
    with pytest.raises(AssertionError):  # the shape contract of the boundary (dr.py:58-124)
        from deodr_amd import renderScene

        renderScene(s, 1.0, np.zeros((5, 5, 3)), np.zeros((32, 40)))


def test_rasterizer_calls_and_forward_state_stamps(fake, oracle_api):
    """HipRasterizer: render / render_backward / render_fit agree, `grads=` is accumulated into, `clear_grads` clears, a second
    forward on the same workspace makes an older adjoint recompute its state (generation stamp), wrong shapes / devices are refused"""
    from deodr_amd.hip_renderer import DeviceScene, HipRasterizer
    from hip_util import device_scene

    views = [scenes.soup_scene(n_tri=15, width=40, height=32, seed=5 + v, flat=False, textured_ratio=0.5, texture_size=8, min_area=20.0) for v in range(2)]
    views[1].textured, views[1].shaded, views[1].uv = views[0].textured, views[0].shaded, views[0].uv  # one topology, two views
    views[1].colors[views[0].textured.repeat(3)] = 0
    ds = device_scene(views, torch.float64)
    assert isinstance(ds, DeviceScene) and ds.n_views == 2 and ds.device == torch.device("cpu")
    r = HipRasterizer.for_scene(ds)
    obs = torch.rand(2, 32, 40, 3, dtype=torch.float64)
    image, z = r.render(ds, 1.0)
    stamp = r.generation
    g1 = r.render_backward(ds, residual_obs=obs)
    image_f, z_f, g2 = r.render_fit(ds, obs, 1.0)
    assert torch.equal(image, image_f) and torch.equal(z, z_f)
    for k in ("ij_b", "colors_b", "shade_b", "uv_b", "texture_b"):
        assert torch.equal(g1[k], g2[k]), k
    g3 = r.render_fit(ds, obs, 1.0, grads=g2)[2]  # accumulated into
    assert g3 is g2 and rel(g3["ij_b"], 2 * g1["ij_b"]) < 1e-14
    r.render_fit(ds, obs, 1.0, grads=g2, clear_grads=True)
    assert rel(g2["ij_b"], g1["ij_b"]) < 1e-14
    # explicit image_b == the residual formed by the caller
    g4 = r.render_backward(ds, image_b=2 * (image - obs))
    assert rel(g4["ij_b"], g1["ij_b"]) < 1e-12
    # an adjoint that belongs to an older forward: other inputs have been rendered since
    moved = ds.ij.clone()
    ds.set_views(ij=moved + 0.25)
    r.render(ds, 1.0)
    assert r.generation != stamp
    ds.set_views(ij=moved)
    g5 = r.render_backward(ds, residual_obs=obs, generation=stamp, sigma=1.0)
    assert rel(g5["ij_b"], g1["ij_b"]) < 1e-12
    with pytest.raises(ValueError, match="scene shape differs"):
        r.render(device_scene(views[:1], torch.float64), 1.0)
    with pytest.raises(ValueError, match="out= buffers"):
        r.render(ds, 1.0, out=(torch.zeros(2, 32, 40, 3), torch.zeros(2, 32, 40, dtype=torch.float64)))
    with pytest.raises(RuntimeError, match="before any render"):
        HipRasterizer.for_scene(ds).render_backward(ds, residual_obs=obs)


def test_fit_step_loss_bookkeeping(fake):
    """render_fit(loss_out=...): the background table is computed once per (observation, background) and again when either changes --
    an in-place edit of the observation included --, the loss is that of the stored frame, wrong loss tensors are refused"""
    from deodr_amd.hip_renderer import HipRasterizer
    from hip_util import device_scene

    views = [scenes.soup_scene(n_tri=12, width=43, height=29, seed=9 + v, min_area=20.0) for v in range(2)]
    views[1].textured, views[1].shaded, views[1].uv = views[0].textured, views[0].shaded, views[0].uv
    ds = device_scene(views, torch.float64)
    r = HipRasterizer.for_scene(ds)
    obs = torch.rand(2, 29, 43, 3, dtype=torch.float64)
    loss = torch.zeros(1, dtype=torch.float64)
    tables = []
    for step in range(3):
        image, _z, _g = r.render_fit(ds, obs, 1.0, clear_grads=True, loss_out=loss)
        assert abs(float(loss) - float(((image - obs) ** 2).sum())) <= 1e-12 * float(loss)
        tables.append(r._loss_cache[1])
    assert tables[0] is tables[1] is tables[2]
    obs[0, :8, :8] += 0.25  # in place: the version counter of the tensor is part of the key
    image, _z, _g = r.render_fit(ds, obs, 1.0, clear_grads=True, loss_out=loss)
    assert r._loss_cache[1] is not tables[0] and abs(float(loss) - float(((image - obs) ** 2).sum())) <= 1e-12 * float(loss)
    with pytest.raises(ValueError, match="loss_out"):
        r.render_fit(ds, obs, 1.0, loss_out=torch.zeros(1, dtype=torch.float32))
    # with a clamp: another table; loss and gradients are those of sum (clamp(image) - obs)^2
    before = r._loss_cache[1]
    image, _z, g = r.render_fit(ds, obs, 1.0, clear_grads=True, loss_out=loss, clamp=(0.3, 0.7))
    assert r._loss_cache[1] is not before
    clamped = image.clamp(0.3, 0.7)
    assert abs(float(loss) - float(((clamped - obs) ** 2).sum())) <= 1e-12 * float(loss)
    g = {k: v.clone() for k, v in g.items() if v is not None}
    r.render(ds, 1.0)
    g_ref = r.render_backward(ds, image_b=2 * (clamped - obs) * ((image >= 0.3) & (image <= 0.7)))
    for k, v in g.items():
        assert rel(v, g_ref[k]) < 1e-12, k


def test_device_scene_validation(fake):
    """checkSceneValid's index checks at construction (H.h:2700-2712), textured triangles without a texture"""
    from hip_util import device_scene

    s = scenes.soup_scene(n_tri=6, width=24, height=24, seed=1, flat=False, min_area=10.0)
    bad = scenes.soup_scene(n_tri=6, width=24, height=24, seed=1, flat=False, min_area=10.0)
    bad.faces = s.faces.copy()
    bad.faces[2, 1] = 18
    with pytest.raises(ValueError, match="scene.faces"):
        device_scene(bad)
    bad.faces, bad.faces_uv = s.faces, s.faces_uv.copy()
    bad.faces_uv[0, 0] = 999
    with pytest.raises(ValueError, match="faces_uv"):
        device_scene(bad)
    bad.faces_uv = s.faces_uv
    bad.textured, bad.shaded, bad.texture = np.ones(6, dtype=bool), np.ones(6, dtype=bool), np.zeros((0, 0))
    with pytest.raises(ValueError, match="no texture"):
        device_scene(bad)


def test_torch_operator_caches_and_reuploads(fake, oracle_api):
    """TorchDifferentiableRender2D (deodr/pytorch signature): gradients for ij and colors; the device scene and workspace are cached
    on the scene object; a replaced texture / background object is uploaded again; two renders in one graph give both gradients"""
    from types import SimpleNamespace

    from deodr_amd.pytorch import TorchDifferentiableRender2D

    s = scenes.soup_scene(n_tri=10, width=32, height=32, seed=9, flat=False, textured_ratio=0.5, texture_size=8, min_area=20.0)
    holder = SimpleNamespace(scene_2d=s)
    ref, fixed = oracle_api.ref() or oracle_api.port(), oracle_api.ref(fixed=True) or oracle_api.port(fixed=True)
    ij = torch.tensor(s.ij, requires_grad=True)
    colors = torch.tensor(s.colors, requires_grad=True)
    image = TorchDifferentiableRender2D(ij, colors, holder)
    image_ref, z_ref = ref.render(s, 1)
    assert rel(image.detach(), image_ref) < 1e-14
    seed = torch.randn_like(image)
    image.backward(seed)
    g = fixed.grads(s, 1, image_ref, z_ref, seed.numpy())
    assert rel(ij.grad, g["ij_b"]) < 1e-12 and rel(colors.grad, g["colors_b"]) < 1e-12
    state = holder.__dict__["_hip_state"]
    TorchDifferentiableRender2D(ij.detach(), colors.detach(), holder)
    assert holder.__dict__["_hip_state"] is state  # cached: same DeviceScene, same workspace
    s.texture = 1 - s.texture  # another object: uploaded again
    image2 = TorchDifferentiableRender2D(ij.detach(), colors.detach(), holder)
    assert rel(image2, ref.render(s, 1)[0]) < 1e-14 and holder.__dict__["_hip_state"]["r"] is state["r"]
    # two renders in one autograd graph: the first backward finds the workspace holding the second forward
    ij_a = torch.tensor(s.ij, requires_grad=True)
    ij_b = torch.tensor(s.ij + 0.3, requires_grad=True)
    loss = (TorchDifferentiableRender2D(ij_a, colors.detach(), holder) ** 2).sum() + (TorchDifferentiableRender2D(ij_b, colors.detach(), holder) ** 2).sum()
    loss.backward()
    img_a, z_a = ref.render(s, 1)
    assert rel(ij_a.grad, fixed.grads(s, 1, img_a, z_a, 2 * img_a)["ij_b"]) < 1e-12
    s.ij = s.ij + 0.3
    img_b, z_b = ref.render(s, 1)
    assert rel(ij_b.grad, fixed.grads(s, 1, img_b, z_b, 2 * img_b)["ij_b"]) < 1e-12


def test_textured_mesh_through_scene3d_and_the_rasterizer_objects(fake):
    """The reference's duck scene through the drop-ins AND the scene / workspace containers (Scene3D -> Scene3DDevice ->
    DeviceScene holding the texture as a tensor -> HipRasterizer -> C ABI struct): its stored test image, exactly"""
    import deodr_amd as deodr
    from test_scene3d import duck_fixture

    d, texture = duck_fixture()
    mesh = deodr.ColoredTriMesh(d["faces"], d["vertices"], clockwise=False, faces_uv=d["faces_uv"], uv=d["uv"], texture=texture, device="cpu")
    camera = deodr.default_camera(320, 240, 80, mesh.vertices, d["rot"])
    camera.distortion = np.array([-0.5, 0.5, 0, 0, 0])
    scene = deodr.Scene3D()
    scene.set_light(light_directional=0.3 * np.array([1, -1, 0]), light_ambient=0)
    scene.set_mesh(mesh)
    scene.set_background_color(np.array((0.8, 0.8, 0.8)))
    image = scene.render(camera)
    assert fake.calls["render_scene"] == 1
    assert np.abs((image * 255).astype(np.uint8).astype(int) - d["stored_u8"].astype(int)).max() == 0
    scene.clear_gradients()
    scene.render_backward(np.ones_like(image))  # the adjoint reaches vertices and lights through the same objects
    assert np.isfinite(mesh._vertices_b).all() and np.abs(mesh._vertices_b).max() > 0 and np.isfinite(scene.light_directional_b).all()


def test_the_randomised_sweep_harness_runs(fake, capsys):
    """tests/fuzz_parity.py (run by hand on the GPU box) end to end with the checker on both sides: every mode of both sweeps executes
    and reports no miss -- so a mistake in the harness is found here, not on GPU time"""
    import fuzz_parity

    assert fuzz_parity.main(8) == 0
    assert fuzz_parity.main_meshes(10) == 0
    out = capsys.readouterr().out
    assert "8 random scenes, 0 missed" in out and "10 random mesh scenes, 0 missed" in out
