"""The reference's OWN, UNMODIFIED Python running on the drop-ins ("drops into mesh_fitter.py").

Only where /root/reference exists (the build container); nothing of the reference is copied: its package is imported from where
it lies, for the duration of a test, with exactly one module replaced through ``sys.modules``:

* ``deodr.differentiable_renderer_cython`` := a module exposing ``deodr_amd.hip_renderer.renderSceneCpp / renderSceneBCpp`` -- the
  reference's ``deodr/mesh_fitter.py::MeshDepthFitter`` (NumPy ``Scene3D`` / ``Camera`` / ``ColoredTriMesh`` of the reference itself)
  then fits the depth image of its ``tests/test_depth_image_hand_fitting.py`` through OUR entry points;
* ``deodr.pytorch.differentiable_renderer_pytorch`` := ``deodr_amd.pytorch.differentiable_renderer_pytorch`` -- the reference's
  ``deodr/pytorch/mesh_fitter_pytorch.py::MeshDepthFitter`` (its own mesh, rigid energy, quaternion code, CPU tensors, ``.numpy()``
  calls) then runs on OUR ``CameraPytorch`` / ``Scene3DPytorch``.

Without a GPU the rasterizer behind the drop-ins is the CPU checker (tests/fake_hip.py: the C ABI restated over it; tests/cpu_raster.py:
the one device call of the Scene3D pipeline), so what is exercised here is the boundary itself: marshalling, shapes, in-place
contracts, gradient rebinding, devices.  Goldens: the reference's own (251.3271111... NumPy path, 251.32711067513003 PyTorch path)."""

import contextlib
import os
import sys
import types

import numpy as np
import pytest

from conftest import GOLDEN

REFERENCE = os.environ.get("DEODR_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "deodr")), reason="the reference tree is not on this machine")


@contextlib.contextmanager
def reference_package(replaced):
    """``import deodr`` resolves to the reference's source tree with the modules of `replaced` {name: module} substituted; everything is
    undone on exit (sys.path, sys.modules)."""
    stubs = {"trimesh": types.ModuleType("trimesh"), "trimesh.base": types.ModuleType("trimesh.base")}
    stubs["trimesh.base"].Trimesh = type("Trimesh", (), {})  # a type annotation in deodr/triangulated_mesh.py:10-13, 369
    stubs["trimesh"].base = stubs["trimesh.base"]
    before = {k: v for k, v in sys.modules.items() if k == "deodr" or k.startswith("deodr.") or k in stubs}
    for k in before:
        del sys.modules[k]
    sys.modules.update({k: v for k, v in stubs.items() if k not in sys.modules})
    sys.modules.update(replaced)
    sys.path.insert(0, REFERENCE)
    try:
        yield
    finally:
        sys.path.remove(REFERENCE)
        for k in [k for k in sys.modules if k == "deodr" or k.startswith("deodr.") or k in stubs]:
            del sys.modules[k]
        sys.modules.update(before)


def depth_fit_inputs():
    """the set-up of deodr/examples/depth_image_hand_fitting.py:34-60 (the example itself writes into its own folder: not imported)"""
    import deodr

    depth_image = np.fliplr(np.fromfile(os.path.join(deodr.data_path, "depth.bin"), dtype=np.float32).reshape(240, 320).astype(np.float64))
    depth_image = depth_image[20:-20, 60:-60]
    max_depth = 450
    depth_image[depth_image == 0] = max_depth
    depth_image = depth_image / max_depth
    faces, vertices = deodr.read_obj(os.path.join(deodr.data_path, "hand.obj"))
    return depth_image, faces, vertices, max_depth


def run_depth_fit(fitter_class, n_iter):
    depth_image, faces, vertices, max_depth = depth_fit_inputs()
    fitter = fitter_class(vertices, faces.copy(), np.array([0.1, 0.1, 0.1]), np.zeros(3), cregu=1000)
    fitter.set_image(depth_image, focal=241, distortion=np.array([1, 0, 0, 0, 0]))
    fitter.set_max_depth(1)
    fitter.set_depth_scale(110 / max_depth)
    energies = []
    for _ in range(n_iter):
        energy, synthetic_depth, diff_image = fitter.step()
        assert synthetic_depth.shape == depth_image.shape and diff_image.shape == depth_image.shape
        energies.append(float(energy))
    return energies


def test_reference_numpy_fitter_runs_unmodified_on_our_entry_points(oracle_api, capsys):
    """deodr/mesh_fitter.py::MeshDepthFitter, 50 iterations, golden of the reference's tests/test_depth_image_hand_fitting.py:36-42"""
    from fake_hip import emulate
    from deodr_amd import hip_renderer

    ours = types.ModuleType("deodr.differentiable_renderer_cython")
    ours.renderSceneCpp, ours.renderSceneBCpp = hip_renderer.renderSceneCpp, hip_renderer.renderSceneBCpp
    checker = oracle_api.ref() or oracle_api.port()
    with reference_package({"deodr.differentiable_renderer_cython": ours}), emulate(checker, checker) as fake:
        import deodr.differentiable_renderer as ref_dr
        from deodr.mesh_fitter import MeshDepthFitter

        assert ref_dr.differentiable_renderer_cython is ours and ref_dr.__file__.startswith(REFERENCE)
        energies = run_depth_fit(MeshDepthFitter, 50)
        assert fake.calls["render_scene"] >= 50 and fake.calls["render_scene_b"] == 50  # every frame went through the C-ABI restatement
    capsys.readouterr()  # (the reference prints every energy)
    d = np.load(os.path.join(GOLDEN, "depth_hand_fit.npz"))
    assert np.allclose(energies, d["energies"], rtol=1e-9, atol=1e-9), np.abs(np.array(energies) - d["energies"]).max()
    assert min(abs(energies[49] - g) for g in (251.32711113732933, 251.32711113730954, 251.3271111242092)) < 1e-5


def test_reference_pytorch_fitter_runs_unmodified_on_our_single_view_classes(oracle_api, capsys):
    """deodr/pytorch/mesh_fitter_pytorch.py::MeshDepthFitter on deodr_amd.pytorch.{CameraPytorch, Scene3DPytorch}: 50 iterations,
    golden of the reference's tests/test_depth_image_hand_fitting.py:18-24"""
    import cpu_raster
    import deodr_amd.pytorch.differentiable_renderer_pytorch as ours

    from deodr_amd import hip_renderer

    cy = types.ModuleType("deodr.differentiable_renderer_cython")  # (the package imports it; this fitter never reaches it)
    cy.renderSceneCpp, cy.renderSceneBCpp = hip_renderer.renderSceneCpp, hip_renderer.renderSceneBCpp
    checker = oracle_api.ref() or oracle_api.port()
    with reference_package({"deodr.differentiable_renderer_cython": cy, "deodr.pytorch.differentiable_renderer_pytorch": ours}), cpu_raster.emulate(checker):
        import torch

        saved = ours._resolve_device
        ours._resolve_device = lambda device: torch.device("cpu")  # (no GPU here: the pipeline's tensors stay where the fitter's are)
        try:
            import deodr.pytorch as ref_torch
            from deodr.pytorch.mesh_fitter_pytorch import MeshDepthFitter

            assert ref_torch.Scene3DPytorch is ours.Scene3DPytorch and ref_torch.CameraPytorch is ours.CameraPytorch
            assert ref_torch.mesh_fitter_pytorch.__file__.startswith(REFERENCE)
            energies = run_depth_fit(MeshDepthFitter, 50)
        finally:
            ours._resolve_device = saved
    capsys.readouterr()
    assert min(abs(energies[49] - g) for g in (251.32711067513003, 251.31652686512888, 251.31652686495823)) < 1e-5, energies[49]


def test_single_view_classes_have_the_reference_shapes(oracle_api):
    """render -> [H,W,C], render_depth -> [H,W,1], project_points -> ([V,2], [V]) on the device of the inputs; gradients reach
    mesh.vertices (deodr/pytorch/differentiable_renderer_pytorch.py:13-38, 84-109; callers mesh_fitter_pytorch.py:279-283, 458)"""
    import cpu_raster
    import torch

    import deodr_amd.pytorch.differentiable_renderer_pytorch as ours
    from deodr_amd.scene3d import DeviceMesh

    d = np.load(os.path.join(GOLDEN, "hand_mesh.npz"))
    vertices, faces = d["vertices"], d["faces"]
    checker = oracle_api.ref() or oracle_api.port()
    with cpu_raster.emulate(checker):
        saved = ours._resolve_device
        ours._resolve_device = lambda device: torch.device("cpu")
        try:
            center = vertices.mean(axis=0)
            rot = np.array([[1.0, 0, 0], [0, -1, 0], [0, 0, -1]])
            cam_center = center + np.array([0, 0, 9.0]) * np.max(np.std(vertices, axis=0))
            cam = ours.CameraPytorch(np.column_stack((rot, -rot.T.dot(cam_center))), np.array([[128.0, 0, 32], [0, 128.0, 24], [0, 0, 1]]), 48, 64)
            v = torch.tensor(vertices, requires_grad=True)
            ij, depths = cam.project_points(v)
            assert ij.shape == (len(vertices), 2) and depths.shape == (len(vertices),) and not ij.is_cuda
            scene = ours.Scene3DPytorch()
            mesh = DeviceMesh(faces, v, colors=np.random.RandomState(0).rand(len(vertices), 3), device="cpu")
            scene.set_mesh(mesh)
            scene.set_light(np.array([-0.1, -0.5, -0.4]), 0.6)
            scene.set_background_color([0.5, 0.6, 0.7])
            image, z = scene.render(cam, return_z_buffer=True)
            assert image.shape == (48, 64, 3) and z.shape == (48, 64)
            scene.batched.background_color = None
            scene.set_background_color([1.0])
            depth = scene.render_depth(cam, depth_scale=0.1)
            assert depth.shape == (48, 64, 1)
            (torch.sum(depth**2) + torch.sum(image**2)).backward()  # two renders of one scene in one graph
            assert v.grad is not None and bool(torch.isfinite(v.grad).all()) and float(v.grad.abs().max()) > 0
        finally:
            ours._resolve_device = saved


def test_pytorch_package_exports_the_reference_names():
    """every public name of deodr/pytorch/*.py has a counterpart in deodr_amd.pytorch (differentiable_renderer_pytorch.py:13, 41, 84;
    laplacian_rigid_energy_pytorch.py:25; mesh_fitter_pytorch.py:26, 34, 124, 177, 334; triangulated_mesh_pytorch.py:20, 55), and the
    adjacency class computes the reference's normals"""
    import torch

    import deodr_amd.pytorch as ours

    for name in ("CameraPytorch", "TorchDifferentiableRenderer2DFunc", "Scene3DPytorch", "LaplacianRigidEnergyPytorch", "qrot", "MeshDepthFitterEnergy",
                 "MeshDepthFitterPytorchOptim", "MeshDepthFitter", "MeshRGBFitterWithPose", "TriMeshAdjacenciesPytorch", "ColoredTriMeshPytorch"):  # fmt: skip
        assert hasattr(ours, name), name
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hand_mesh.npz"))
    adjacency = ours.TriMeshAdjacenciesPytorch(d["faces"].astype(np.int64), clockwise=False, device="cpu")
    vertices = torch.tensor(d["vertices"])
    face_normals = adjacency.compute_face_normals(vertices)
    vertex_normals = adjacency.compute_vertex_normals(face_normals)
    cy = types.ModuleType("deodr.differentiable_renderer_cython")  # (the reference's package imports its compiled module; never reached here)
    with reference_package({"deodr.differentiable_renderer_cython": cy}):
        from deodr.triangulated_mesh import TriMeshAdjacencies

        ref = TriMeshAdjacencies(d["faces"].astype(np.int64))
        fr = ref.compute_face_normals(d["vertices"])
        assert np.abs(face_normals.numpy() - fr).max() < 1e-14 and np.abs(vertex_normals.numpy() - ref.compute_vertex_normals(fr)).max() < 1e-14
        flags_ref = ref.edge_on_silhouette(d["vertices"][:, :2])
    assert np.array_equal(adjacency.edge_on_silhouette(vertices[:, :2]).numpy().astype(bool), flags_ref)
