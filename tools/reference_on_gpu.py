"""The reference's OWN, UNMODIFIED fitters on the REAL library: tests/test_reference_dropin.py without the
fake_hip / cpu_raster swaps.  Needs the reference's Python package next to the repository for ONE GPU call (never committed):

    cp -r /root/reference/deodr _scratch_reference/deodr        (in the build container; _scratch_reference/ is git-ignored)
    gpurun -- 'DEODR_REFERENCE=$GRAFT_REPO_ROOT/_scratch_reference python tools/reference_on_gpu.py'

1. deodr/mesh_fitter.py::MeshDepthFitter (NumPy Scene3D / Camera / ColoredTriMesh of the reference) with
   deodr.differentiable_renderer_cython := deodr_amd.hip_renderer.renderSceneCpp / renderSceneBCpp on libdeodr_hip.so, 50 iterations,
   golden of the reference's tests/test_depth_image_hand_fitting.py:36-42 (251.3271111...);
2. deodr/pytorch/mesh_fitter_pytorch.py::MeshDepthFitter with deodr.pytorch.differentiable_renderer_pytorch :=
   deodr_amd.pytorch.differentiable_renderer_pytorch (single-view classes, tensors moved to the GPU inside), golden :18-24."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

assert torch.cuda.is_available(), "this tool is for the GPU box"
import test_reference_dropin as T  # noqa: E402  (reference_package, run_depth_fit: the same code the CPU suite runs)

assert os.path.isdir(os.path.join(T.REFERENCE, "deodr")), f"no reference package under {T.REFERENCE}"
from deodr_amd import hip_renderer  # noqa: E402

print("library:", hip_renderer.LIB_PATH, "ABI", hip_renderer.lib().deodr_hip_abi_version(), "device:", torch.cuda.get_device_name(0))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "depth_hand_fit.npz"))["energies"]

ours = types.ModuleType("deodr.differentiable_renderer_cython")
ours.renderSceneCpp, ours.renderSceneBCpp = hip_renderer.renderSceneCpp, hip_renderer.renderSceneBCpp
with T.reference_package({"deodr.differentiable_renderer_cython": ours}):
    import deodr.differentiable_renderer as ref_dr
    from deodr.mesh_fitter import MeshDepthFitter

    assert ref_dr.differentiable_renderer_cython is ours and ref_dr.__file__.startswith(T.REFERENCE)
    print("1. reference deodr/mesh_fitter.py::MeshDepthFitter from", ref_dr.__file__.rsplit("/", 1)[0], "on renderSceneCpp / renderSceneBCpp of libdeodr_hip.so")
    out = sys.stdout
    sys.stdout = open(os.devnull, "w")  # (the reference prints every energy)
    try:
        energies = T.run_depth_fit(MeshDepthFitter, 50)
    finally:
        sys.stdout = out
err = np.abs(np.array(energies) - GOLD)
print(f"   energy after 50 iterations: {energies[49]!r} (reference golden 251.32711113732933); max |curve - reference's own curve| = {err.max():.3e}")
assert min(abs(energies[49] - g) for g in (251.32711113732933, 251.32711113730954, 251.3271111242092)) < 1e-5
assert np.allclose(energies, GOLD, rtol=1e-7, atol=1e-7)

import deodr_amd.pytorch.differentiable_renderer_pytorch as ours_t  # noqa: E402

cy = types.ModuleType("deodr.differentiable_renderer_cython")
cy.renderSceneCpp, cy.renderSceneBCpp = hip_renderer.renderSceneCpp, hip_renderer.renderSceneBCpp
with T.reference_package({"deodr.differentiable_renderer_cython": cy, "deodr.pytorch.differentiable_renderer_pytorch": ours_t}):
    import deodr.pytorch as ref_torch
    from deodr.pytorch.mesh_fitter_pytorch import MeshDepthFitter as TorchFitter

    assert ref_torch.Scene3DPytorch is ours_t.Scene3DPytorch and ref_torch.mesh_fitter_pytorch.__file__.startswith(T.REFERENCE)
    print("2. reference deodr/pytorch/mesh_fitter_pytorch.py::MeshDepthFitter on deodr_amd.pytorch (HIP rasterizer on", ours_t._resolve_device("cuda"), ")")
    out = sys.stdout
    sys.stdout = open(os.devnull, "w")
    try:
        energies_t = T.run_depth_fit(TorchFitter, 50)
    finally:
        sys.stdout = out
print(f"   energy after 50 iterations: {energies_t[49]!r} (reference goldens 251.32711067513003 / 251.31652686512888)")
assert min(abs(energies_t[49] - g) for g in (251.32711067513003, 251.31652686512888, 251.31652686495823)) < 1e-5, energies_t[49]
print("OK: both unmodified reference fitters reproduce their goldens on the MI355X through libdeodr_hip.so")
