"""PyTorch operator layer, source-compatible with ``deodr.pytorch`` for the rasterizer op."""

from .differentiable_renderer_pytorch import (  # noqa: F401
    TorchDifferentiableRender2D,
    TorchDifferentiableRenderer2DFunc,
    TorchDifferentiableRenderViews,
    TorchDifferentiableRenderViewsFunc,
    TorchRenderViewsL2Loss,
    TorchRenderViewsL2LossFunc,
)
