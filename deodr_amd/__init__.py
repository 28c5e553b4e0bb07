"""deodr_amd -- MI355X-native (HIP / gfx950) implementation of DEODR's differentiable rasterizer hot path.

Only the path ``renderScene`` / ``renderScene_B`` is implemented; the Python surface
mirrors ``deodr.differentiable_renderer`` (Scene2D, renderScene, renderSceneB), the Cython entry points
(``renderSceneCpp`` / ``renderSceneBCpp`` in :mod:`deodr_amd.hip_renderer`) and ``deodr.pytorch``'s
``TorchDifferentiableRenderer2DFunc``.  The HIP shared library is required: nothing falls back to the CPU.
"""

from .differentiable_renderer import Scene2D, Scene2DBase, renderScene, renderSceneB  # noqa: F401

__version__ = "0.1.0"
