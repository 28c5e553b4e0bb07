"""``torch.autograd.Function`` wrappers of the HIP rasterizer.

``TorchDifferentiableRenderer2DFunc`` keeps the signature of the reference's class
(deodr/pytorch/differentiable_renderer_pytorch.py:41-81): ``forward(ctx, ij, colors, scene)`` where ``scene`` carries a
``scene_2d`` (a ``Scene2D``), ``backward`` returns ``(ij_b, colors_b, None)``; like the reference it renders with sigma = 1
unless ``scene.sigma_2d`` is set.  Unlike the reference nothing goes through NumPy: ``ij`` / ``colors`` may live on the ROCm
device (the image then stays there) or on the CPU (results are copied back, reference behaviour).  The incoming
``grad_output`` is never mutated.

``TorchDifferentiableRenderViewsFunc`` is the batched form for ``n_views`` views of one mesh on a prepared
:class:`deodr_amd.hip_renderer.DeviceScene` -- the unit the multi-GPU path shards.
"""

import numpy as np
import torch

from ..hip_renderer import DeviceScene, HipRasterizer


def _device_state(scene, device, pixel_dtype):
    """DeviceScene + HipRasterizer for ``scene.scene_2d``, cached on the scene object and rebuilt when the topology,
    the image size or the flags change."""
    s = scene.scene_2d
    key = (id(s.faces), s.height, s.width, int(np.shape(s.colors)[1]), np.shape(s.faces)[0], s.clockwise, s.backface_culling,
           s.strict_edge, s.perspective_correct, s.integer_pixel_centers, str(device), pixel_dtype)  # fmt: skip
    st = scene.__dict__.get("_hip_state")
    if st is None or st[0] != key:
        to_np = lambda a: a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
        bgi = None if s.background_image is None else to_np(s.background_image)[None]
        ds = DeviceScene(
            to_np(s.faces), to_np(s.faces_uv), to_np(s.textured), to_np(s.shaded), to_np(s.uv), to_np(s.ij)[None], to_np(s.depths)[None],
            to_np(s.colors)[None], to_np(s.shade)[None], to_np(s.edgeflags)[None], s.height, s.width,
            texture=to_np(s.texture) if np.size(s.texture) else None,
            background_color=None if s.background_color is None else to_np(s.background_color), background_image=bgi,
            clockwise=s.clockwise, backface_culling=s.backface_culling, strict_edge=s.strict_edge,
            perspective_correct=s.perspective_correct, integer_pixel_centers=s.integer_pixel_centers, vertex_dtype=torch.float64,
            pixel_dtype=pixel_dtype, device=device,
        )  # fmt: skip
        st = (key, ds, HipRasterizer.for_scene(ds))
        scene.__dict__["_hip_state"] = st
    return st[1], st[2]


class TorchDifferentiableRenderer2DFunc(torch.autograd.Function):
    """Differentiable 2.5-D rendering: (ij, colors) -> image, gradients w.r.t. ij and colors."""

    @staticmethod
    def forward(ctx, ij, colors, scene):
        s = scene.scene_2d
        on_device = ij.is_cuda
        device = ij.device if on_device else torch.device("cuda")
        pixel_dtype = torch.float32 if (on_device and colors.dtype == torch.float32) else torch.float64
        ds, r = _device_state(scene, device, pixel_dtype)
        to_t = lambda a: a.detach() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))
        ds.set_views(ij=ij.detach()[None], colors=colors.detach()[None], depths=to_t(s.depths)[None], shade=to_t(s.shade)[None],
                     edgeflags=to_t(s.edgeflags)[None])  # fmt: skip
        sigma = getattr(scene, "sigma_2d", 1)  # the reference hard-codes 1 (deodr/pytorch/differentiable_renderer_pytorch.py:55)
        image, z_buffer = r.render(ds, sigma)
        ctx.ds, ctx.r, ctx.on_device, ctx.in_dtypes = ds, r, on_device, (ij.dtype, colors.dtype)
        ctx.z_buffer = z_buffer[0]
        out = image[0]
        return out if on_device else out.to(device="cpu", dtype=torch.float64)

    @staticmethod
    def backward(ctx, image_b):
        g = ctx.r.render_backward(ctx.ds, image_b=image_b)
        ij_b, colors_b = g["ij_b"][0], g["colors_b"][0]
        if not ctx.on_device:
            ij_b, colors_b = ij_b.cpu(), colors_b.cpu()
        return ij_b.to(ctx.in_dtypes[0]), colors_b.to(ctx.in_dtypes[1]), None


TorchDifferentiableRender2D = TorchDifferentiableRenderer2DFunc.apply


class TorchDifferentiableRenderViewsFunc(torch.autograd.Function):
    """n_views views in one launch: (ij [n,V,2], colors [n,V,C]) -> image [n,H,W,C] on ``device_scene``'s GPU."""

    @staticmethod
    def forward(ctx, ij, colors, device_scene, rasterizer, sigma):
        device_scene.set_views(ij=ij.detach(), colors=colors.detach())
        image, _ = rasterizer.render(device_scene, sigma)
        ctx.ds, ctx.r, ctx.in_dtypes = device_scene, rasterizer, (ij.dtype, colors.dtype)
        return image

    @staticmethod
    def backward(ctx, image_b):
        g = ctx.r.render_backward(ctx.ds, image_b=image_b)
        return g["ij_b"].to(ctx.in_dtypes[0]), g["colors_b"].to(ctx.in_dtypes[1]), None, None, None


def TorchDifferentiableRenderViews(ij, colors, device_scene, rasterizer, sigma=1.0):
    return TorchDifferentiableRenderViewsFunc.apply(ij, colors, device_scene, rasterizer, sigma)


class TorchRenderViewsL2LossFunc(torch.autograd.Function):
    """sum((render(ij, colors) - obs)**2) over ``n_views`` views as ONE op: (ij [n,V,2], colors [n,V,C]) -> scalar loss.

    What the reference's fitters write as ``image = render(...); loss = ((image - obs) ** 2).sum(); loss.backward()``
    (deodr/pytorch/mesh_fitter_pytorch.py, dr.py:701-740): here the forward is one ``deodr_hip_render_scene_fit`` call that
    renders AND back-propagates the residual (the gradient of the loss w.r.t. the image is known as soon as a pixel is
    resolved), so ``backward`` only scales the stored gradients.  ``image`` and ``z_buffer`` of the last call are kept on
    the context owner (``rasterizer.last_fit``) for display."""

    @staticmethod
    def forward(ctx, ij, colors, obs, device_scene, rasterizer, sigma):
        device_scene.set_views(ij=ij.detach(), colors=colors.detach())
        image, z, g = rasterizer.render_fit(device_scene, obs, sigma, clear_grads=False)
        rasterizer.last_fit = (image, z)
        ctx.save_for_backward(g["ij_b"].to(ij.dtype), g["colors_b"].to(colors.dtype))
        return ((image.double() - obs.to(image.device).double()) ** 2).sum()

    @staticmethod
    def backward(ctx, loss_b):
        ij_b, colors_b = ctx.saved_tensors
        return loss_b.to(ij_b.dtype) * ij_b, loss_b.to(colors_b.dtype) * colors_b, None, None, None, None


def TorchRenderViewsL2Loss(ij, colors, obs, device_scene, rasterizer, sigma=1.0):
    return TorchRenderViewsL2LossFunc.apply(ij, colors, obs, device_scene, rasterizer, sigma)
