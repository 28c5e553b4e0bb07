"""PyTorch layer, source-compatible with ``deodr.pytorch`` (deodr/pytorch/__init__.py): the rasterizer op and, on top of it, the
camera / scene / mesh / energy / fitter classes -- all of them device-resident here (ROCm tensors end to end; the reference's
versions convert to NumPy around every render, deodr/pytorch/differentiable_renderer_pytorch.py:52-54, and evaluate the rigid
energy with SciPy on the host, deodr/pytorch/laplacian_rigid_energy_pytorch.py:38-46)."""

from ..mesh_fitter import (  # noqa: F401
    MeshDepthFitter,
    MeshDepthFitterEnergy,
    MeshDepthFitterPytorchOptim,
    MeshRGBFitterWithPose,
    MeshRGBFitterWithPoseMultiFrame,
)
from ..scene3d import DeviceCamera, DeviceMesh, LaplacianRigidEnergyDevice, Scene3DDevice  # noqa: F401  (the batched classes, n views per call)
from .differentiable_renderer_pytorch import (  # noqa: F401
    CameraPytorch,  # one view, the reference's shapes (over DeviceCamera)
    Scene3DPytorch,  # one view, the reference's shapes (over Scene3DDevice)
    TorchDifferentiableRender2D,
    TorchDifferentiableRenderer2DFunc,
    TorchDifferentiableRenderViews,
    TorchDifferentiableRenderViewsFunc,
    TorchRenderViewsL2Loss,
    TorchRenderViewsL2LossFunc,
)


def ColoredTriMeshPytorch(faces, vertices, clockwise=False, faces_uv=None, uv=None, texture=None, colors=None, device="cuda"):
    """argument order of deodr/pytorch/triangulated_mesh_pytorch.py:57-76"""
    return DeviceMesh(faces, vertices, clockwise=clockwise, colors=colors, uv=uv, faces_uv=faces_uv, texture=texture, device=device)


class LaplacianRigidEnergyPytorch:
    """``evaluate(vertices) -> (energy, gradient, approximate hessian)`` like deodr/pytorch/laplacian_rigid_energy_pytorch.py:24-50 (the
    hessian, a host SciPy matrix in the reference that no PyTorch fitter uses, is not materialised: None)"""

    def __init__(self, mesh, vertices, cregu):
        self._dev = LaplacianRigidEnergyDevice(mesh.topology, vertices, cregu)

    def evaluate(self, vertices):
        energy, grad = self._dev.evaluate(vertices)
        return energy, grad, None
