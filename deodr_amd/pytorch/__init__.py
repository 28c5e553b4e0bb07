"""PyTorch layer, source-compatible with ``deodr.pytorch`` (deodr/pytorch/__init__.py): the rasterizer op and, on top of it, the
camera / scene / mesh / energy / fitter classes -- all of them device-resident here (ROCm tensors end to end; the reference's
versions convert to NumPy around every render, deodr/pytorch/differentiable_renderer_pytorch.py:52-54, and evaluate the rigid
energy with SciPy on the host, deodr/pytorch/laplacian_rigid_energy_pytorch.py:38-46)."""

import torch

from ..mesh_fitter import (  # noqa: F401
    MeshDepthFitter,
    MeshDepthFitterEnergy,
    MeshDepthFitterPytorchOptim,
    MeshRGBFitterWithPose,
    MeshRGBFitterWithPoseMultiFrame,
)
from ..mesh_fitter import qrot  # noqa: F401  (deodr/pytorch/mesh_fitter_pytorch.py:26-31)
from ..scene3d import DeviceCamera, DeviceMesh, LaplacianRigidEnergyDevice, MeshTopology, Scene3DDevice  # noqa: F401  (the batched classes, n views per call)
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


class TriMeshAdjacenciesPytorch(MeshTopology):
    """``TriMeshAdjacenciesPytorch(faces, clockwise)`` with the reference's three methods (deodr/pytorch/triangulated_mesh_pytorch.py:20-52)
    over the device topology: flat index arrays instead of SciPy / torch sparse matrices, tensors in and out (``edge_on_silhouette``
    returns a uint8 tensor on the device of its input where the reference returns a NumPy bool array after a host round trip)."""

    def __init__(self, faces, clockwise=False, device="cuda"):
        super().__init__(faces, None, clockwise, device)

    def compute_face_normals(self, vertices):
        return self.face_normals(vertices)

    def compute_vertex_normals(self, face_normals):
        acc = torch.zeros((self.nb_vertices, 3), dtype=face_normals.dtype, device=face_normals.device)
        acc = acc.index_add(0, self.faces.reshape(-1), face_normals.repeat_interleave(3, dim=0))
        return acc / acc.norm(dim=-1, keepdim=True)
