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

from ..hip_renderer import DeviceScene, HipRasterizer, _count, _resolve_device


def _to_np(a):
    return a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)


def _same_small(a, b):
    """Topology-sized arrays are compared by content: the reference's Scene3D.render rebuilds them for every frame
    (``faces.astype(np.uint32)``, dr.py:919), so object identity says nothing."""
    if a is b:
        return True
    a, b = _to_np(a), _to_np(b)
    return a.shape == b.shape and bool(np.array_equal(a, b))


def _device_state(scene, device, pixel_dtype):
    """DeviceScene + HipRasterizer for ``scene.scene_2d``, cached on the scene object.

    * dims, flags, device or dtype changed, or the topology (faces, faces_uv, textured, shaded -- compared by CONTENT) ->
      a new DeviceScene; the HipRasterizer (the workspace) is kept as long as the dims are the same;
    * uv, texture, background (compared by identity: they are large) replaced by another object -> that array is uploaded
      again.  Mutating one of them IN PLACE is not seen (nor is it by the reference's own pytorch layer, which keeps numpy
      copies)."""
    s = scene.scene_2d
    nb_colors = int(np.shape(s.colors)[1])
    dims = (int(np.shape(s.faces)[0]), int(s.height), int(s.width), nb_colors, str(device), pixel_dtype)
    flags = (bool(s.clockwise), bool(s.backface_culling), bool(s.strict_edge), bool(s.perspective_correct), bool(s.integer_pixel_centers))
    st = scene.__dict__.get("_hip_state")
    topo = ("faces", "faces_uv", "textured", "shaded")
    big = ("uv", "texture", "background_image", "background_color")
    rebuild = st is None or st["dims"] != dims or st["flags"] != flags or not all(_same_small(getattr(s, k), st["src"][k]) for k in topo)
    if rebuild:
        bgi = None if s.background_image is None else _to_np(s.background_image)[None]
        ds = DeviceScene(
            _to_np(s.faces), _to_np(s.faces_uv), _to_np(s.textured), _to_np(s.shaded), _to_np(s.uv), _to_np(s.ij)[None],
            _to_np(s.depths)[None], _to_np(s.colors)[None], _to_np(s.shade)[None], _to_np(s.edgeflags)[None], s.height, s.width,
            texture=_to_np(s.texture) if _count(s.texture) else None,
            background_color=None if s.background_color is None else _to_np(s.background_color), background_image=bgi,
            clockwise=s.clockwise, backface_culling=s.backface_culling, strict_edge=s.strict_edge,
            perspective_correct=s.perspective_correct, integer_pixel_centers=s.integer_pixel_centers, vertex_dtype=torch.float64,
            pixel_dtype=pixel_dtype, device=device,
        )  # fmt: skip
        r = st["r"] if st is not None and st["dims"] == dims else HipRasterizer.for_scene(ds)
        st = dict(dims=dims, flags=flags, ds=ds, r=r, src={k: getattr(s, k) for k in topo + big})
        scene.__dict__["_hip_state"] = st
        return ds, r
    ds = st["ds"]
    for k in big:
        new = getattr(s, k)
        if new is st["src"][k]:
            continue
        st["src"][k] = new
        if k == "uv":
            ds.uv = torch.as_tensor(_to_np(new)).to(device=ds.device, dtype=ds.vertex_dtype).reshape(-1, 2).contiguous()
        elif k == "texture":
            ds.texture = torch.as_tensor(_to_np(new)).to(device=ds.device, dtype=pixel_dtype).contiguous() if _count(new) else None
        elif k == "background_color":
            ds.background_color = None if new is None else torch.as_tensor(_to_np(new)).to(device=ds.device, dtype=pixel_dtype).reshape(-1)
        else:
            ds.background_image = None if new is None else torch.as_tensor(_to_np(new)).to(device=ds.device, dtype=pixel_dtype).reshape(
                1, s.height, s.width, nb_colors).contiguous()  # fmt: skip
    return ds, st["r"]


class TorchDifferentiableRenderer2DFunc(torch.autograd.Function):
    """Differentiable 2.5-D rendering: (ij, colors) -> image, gradients w.r.t. ij and colors."""

    @staticmethod
    def forward(ctx, ij, colors, scene):
        s = scene.scene_2d
        on_device = ij.is_cuda
        device = ij.device if on_device else _resolve_device("cuda")
        pixel_dtype = torch.float32 if (on_device and colors.dtype == torch.float32) else torch.float64
        ds, r = _device_state(scene, device, pixel_dtype)
        to_t = lambda a: a.detach() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))
        per_view = dict(depths=to_t(s.depths)[None], shade=to_t(s.shade)[None], edgeflags=to_t(s.edgeflags)[None])
        ds.set_views(ij=ij.detach()[None], colors=colors.detach()[None], **per_view)
        sigma = getattr(scene, "sigma_2d", 1)  # the reference hard-codes 1 (deodr/pytorch/differentiable_renderer_pytorch.py:55)
        image, z_buffer = r.render(ds, sigma)
        ctx.ds, ctx.r, ctx.on_device, ctx.in_dtypes = ds, r, on_device, (ij.dtype, colors.dtype)
        # the workspace and `ds` are shared by every render of this scene: the stamp tells backward whether they still hold THIS
        # forward (two renders in one graph); if not, the forward state is rebuilt from the saved inputs
        ctx.generation, ctx.per_view, ctx.sigma = r.generation, per_view, sigma
        ctx.save_for_backward(ij, colors)
        ctx.z_buffer = z_buffer[0]
        out = image[0]
        return out if on_device else out.to(device="cpu", dtype=torch.float64)

    @staticmethod
    def backward(ctx, image_b):
        ds, r = ctx.ds, ctx.r
        if r.generation != ctx.generation:
            ij, colors = ctx.saved_tensors
            ds.set_views(ij=ij.detach()[None], colors=colors.detach()[None], **ctx.per_view)
        g = r.render_backward(ds, image_b=image_b, generation=ctx.generation, sigma=ctx.sigma)
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
        ctx.ds, ctx.r, ctx.in_dtypes, ctx.generation, ctx.sigma = device_scene, rasterizer, (ij.dtype, colors.dtype), rasterizer.generation, sigma
        ctx.save_for_backward(ij, colors)
        return image

    @staticmethod
    def backward(ctx, image_b):
        if ctx.r.generation != ctx.generation:  # another forward has used the scene / workspace since: restore the inputs
            ij, colors = ctx.saved_tensors
            ctx.ds.set_views(ij=ij.detach(), colors=colors.detach())
        g = ctx.r.render_backward(ctx.ds, image_b=image_b, generation=ctx.generation, sigma=ctx.sigma)
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
