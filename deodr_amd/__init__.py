"""deodr_amd -- MI355X-native (HIP / gfx950) implementation of DEODR's differentiable rasterizer hot path.

The path ``renderScene`` / ``renderScene_B`` and the callers either side of it (section 8f: camera,
lighting, silhouette flags, normals, rigid energy, fitters -- all device-resident); the Python surface
mirrors ``deodr.differentiable_renderer`` (Scene2D, renderScene, renderSceneB), the Cython entry points
(``renderSceneCpp`` / ``renderSceneBCpp`` in :mod:`deodr_amd.hip_renderer`) and ``deodr.pytorch``'s
``TorchDifferentiableRenderer2DFunc``.  The HIP shared library is required: nothing falls back to the CPU.
"""

from .differentiable_renderer import Scene2D, Scene2DBase, renderScene, renderSceneB  # noqa: F401

_LAZY = {  # the 3-D level (needs torch + the device): imported on first use, `import deodr_amd` stays light
    "Camera": "scene3d_compat", "PerspectiveCamera": "scene3d_compat", "default_camera": "scene3d_compat", "Scene3D": "scene3d_compat",
    "ColoredTriMesh": "triangulated_mesh", "TriMesh": "triangulated_mesh", "TriMeshAdjacencies": "triangulated_mesh",
    "LaplacianRigidEnergy": "laplacian_rigid_energy", "read_obj": "obj", "save_obj": "obj",
}  # fmt: skip


def __getattr__(name):
    if name in _LAZY:
        import importlib

        return getattr(importlib.import_module("." + _LAZY[name], __name__), name)
    raise AttributeError(name)

__version__ = "0.1.0"
