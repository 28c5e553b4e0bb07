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


# ---------------------------------------------------------------------------------------------------------------------
# single-view classes with the reference's shapes (deodr/pytorch/differentiable_renderer_pytorch.py:13-38, 84-109)


class CameraPytorch:
    """ONE pinhole camera (+ OpenCV distortion) with the signatures and shapes of the reference's ``CameraPytorch``:
    ``project_points(points_3d [V,3]) -> (ij [V,2], depths [V])``, ``world_to_camera([V,3]) -> [V,3]``.

    A thin view of the batched :class:`deodr_amd.scene3d.DeviceCamera` (kept as ``.batched``): the algebra runs on the ROCm
    device and the results come back on the device of the input tensor, so that code written for the reference's CPU tensors
    (deodr/pytorch/mesh_fitter_pytorch.py:106-114, 279-283) runs unchanged."""

    def __init__(self, extrinsic, intrinsic, height, width, distortion=None):
        self.extrinsic, self.intrinsic = np.asarray(extrinsic, dtype=np.float64), np.asarray(intrinsic, dtype=np.float64)
        self.height, self.width = int(height), int(width)
        self.distortion = None if distortion is None else np.asarray(distortion, dtype=np.float64)
        self._batched = None

    def batched(self, device):
        from ..scene3d import DeviceCamera

        if self._batched is None or self._batched.device != torch.device(device):
            self._batched = DeviceCamera(self.extrinsic, self.intrinsic, self.height, self.width, self.distortion, device)
        return self._batched

    def world_to_camera(self, points_3d):
        assert isinstance(points_3d, torch.Tensor)
        dev = _compute_device(points_3d)
        return self.batched(dev).world_to_camera(points_3d.to(dev))[0].to(points_3d.device)

    def project_points(self, points_3d, return_depths=True):
        assert isinstance(points_3d, torch.Tensor)
        dev = _compute_device(points_3d)
        ij, depths = self.batched(dev).project_points(points_3d.to(dev))
        ij, depths = ij[0].to(points_3d.device), depths[0].to(points_3d.device)
        return (ij, depths) if return_depths else ij

    def get_center(self):
        return -self.extrinsic[:3, :3].T.dot(self.extrinsic[:, 3])


def _compute_device(t):
    """where the work of a call happens: the tensor's own device when it already lives on a ROCm device, the current one otherwise"""
    return t.device if t.is_cuda else _resolve_device("cuda")


class Scene3DPytorch:
    """ONE view per call with the reference's shapes (``render -> [H,W,C]``, ``render_depth -> [H,W,1]``), over the batched
    :class:`deodr_amd.scene3d.Scene3DDevice` (kept as ``.batched``).

    ``set_mesh`` takes a :class:`deodr_amd.scene3d.DeviceMesh` or any object with the attributes of the reference's
    ``ColoredTriMeshPytorch`` (``faces``, ``vertices`` -- a tensor that may require grad --, ``clockwise``, ``vertices_colors``,
    ``uv`` / ``faces_uv`` / ``texture``): its connectivity is analysed once, its current vertices / colours are picked up at every
    render, and gradients flow back to them through autograd.  Images come back on the device of ``mesh.vertices`` (the
    reference's fitters work on CPU tensors and call ``.numpy()`` on the results)."""

    def __init__(self, sigma=1.0, perspective_correct=False, integer_pixel_centers=True):
        from ..scene3d import Scene3DDevice

        self.batched = Scene3DDevice(sigma=sigma, perspective_correct=perspective_correct, integer_pixel_centers=integer_pixel_centers)
        self.mesh = None
        self._dev_mesh = None
        self.light_directional, self.light_ambient = None, 0.0

    sigma = property(lambda self: self.batched.sigma, lambda self, v: setattr(self.batched, "sigma", float(v)))

    def set_mesh(self, mesh):
        self.mesh, self._dev_mesh = mesh, None

    def set_light(self, light_directional, light_ambient):
        self.light_directional, self.light_ambient = light_directional, light_ambient

    def set_background_color(self, background_color):
        self.batched.set_background_color(background_color)

    def set_background_image(self, background_image):
        self.batched.set_background_image(background_image)

    def _sync(self):
        """the device-side twin of ``self.mesh`` with the mesh's CURRENT vertices and colours -> (twin, device of the results)"""
        from ..scene3d import DeviceMesh

        m = self.mesh
        assert m is not None, "You need to provide a mesh first."
        if isinstance(m, DeviceMesh):
            d, out_dev = m, m.vertices.device
        else:
            v = m.vertices if torch.is_tensor(m.vertices) else torch.as_tensor(np.asarray(m.vertices, dtype=np.float64))
            dev, out_dev = _compute_device(v), v.device
            if self._dev_mesh is None or self._dev_mesh.device != dev:
                uv, tex = getattr(m, "uv", None), getattr(m, "texture", None)
                self._dev_mesh = DeviceMesh(np.asarray(m.faces), v.detach(), clockwise=bool(getattr(m, "clockwise", False)), uv=uv,
                                            faces_uv=getattr(m, "faces_uv", None) if uv is not None else None, texture=tex, device=dev)  # fmt: skip
            d = self._dev_mesh
            d.set_vertices(v.to(device=dev, dtype=d.dtype))  # (differentiable: the gradient returns to m.vertices, wherever it lives)
            colors = getattr(m, "vertices_colors", None)
            if colors is not None:
                c = colors if torch.is_tensor(colors) else torch.as_tensor(np.asarray(colors, dtype=np.float64))
                d.set_vertices_colors(c.to(device=dev, dtype=d.dtype))
        self.batched.set_mesh(d)
        ld, la = self.light_directional, self.light_ambient
        self.batched.light_directional = None if ld is None else (ld if torch.is_tensor(ld) else torch.as_tensor(np.asarray(ld, dtype=np.float64))).to(d.device)
        self.batched.light_ambient = la.to(d.device) if torch.is_tensor(la) else la
        return d, out_dev

    def render(self, camera, return_z_buffer=False, backface_culling=True):
        """-> image [H,W,C] (and z_buffer [H,W]); dr.py:896-983"""
        d, out_dev = self._sync()
        image, z = self.batched.render(camera.batched(d.device), return_z_buffer=True, backface_culling=backface_culling)
        image, z = image[0].to(out_dev), z[0].to(out_dev)
        return (image, z) if return_z_buffer else image

    def render_depth(self, camera, depth_scale=1.0, backface_culling=True):
        """-> depth image [H,W,1]: the depth of every vertex rendered as its colour (dr.py:1001-1036)"""
        d, out_dev = self._sync()
        return self.batched.render_depth(camera.batched(d.device), depth_scale=depth_scale, backface_culling=backface_culling)[0].to(out_dev)
