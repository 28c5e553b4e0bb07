"""TEST harness: run the device-resident pipeline (deodr_amd.scene3d / mesh_fitter) on CPU tensors with the CPU checker (oracle/)
standing in for the HIP rasterizer.

The fitters and the Scene3D front half are plain tensor algebra up to ONE call, ``Scene3DDevice._rasterize``; ``emulate()``
replaces that call by an autograd function that hands every view to the checker's renderScene / renderScene_B.  Nothing of this
is reachable from the product (tests only: the product has no CPU path); it lets the CPU suite run whole fits against the
reference's energy curves, and check fitter changes without a GPU."""

import contextlib

import numpy as np
import torch

from deodr_amd import Scene2D
from deodr_amd.scene3d import Scene3DDevice


def _scene2d(static, ij, depths, colors, shade, flags):
    T, V = len(static["faces"]), len(depths)
    textured = static["textured"]
    return Scene2D(
        faces=static["faces"], faces_uv=static["faces_uv"], ij=ij, depths=depths, textured=np.full(T, textured), uv=static["uv"] if textured else np.zeros((V, 2)),
        shade=shade, colors=colors, shaded=np.full(T, textured), edgeflags=flags.astype(bool), height=static["height"], width=static["width"],
        nb_colors=colors.shape[1], texture=static["texture"] if textured else np.zeros((0, 0)), background_image=static["background_image"],
        background_color=static["background_color"], clockwise=static["clockwise"], backface_culling=static["backface_culling"], strict_edge=True,
        perspective_correct=static["perspective_correct"], integer_pixel_centers=static["integer_pixel_centers"],
    )  # fmt: skip


class CheckerRenderViews(torch.autograd.Function):
    """(ij [n,V,2], colors [n,V,C], shade [n,V]) -> (image [n,H,W,C], z [n,H,W]) through the checker, view by view"""

    @staticmethod
    def forward(ctx, ij, colors, shade, depths, flags, static, checker, sigma):
        views, images, zs = [], [], []
        for i in range(ij.shape[0]):
            bgi = static["background_image"]
            st = dict(static, background_image=None if bgi is None else (bgi[i] if bgi.ndim == 4 else bgi))
            s = _scene2d(st, ij[i].detach().numpy(), depths[i].detach().numpy(), colors[i].detach().numpy(), shade[i].detach().numpy(), flags[i].numpy())
            image, z = checker.render(s, sigma)
            views.append((s, image, z))
            images.append(image)
            zs.append(z)
        ctx.views, ctx.checker, ctx.sigma = views, checker, sigma
        z = torch.as_tensor(np.stack(zs))
        ctx.mark_non_differentiable(z)
        return torch.as_tensor(np.stack(images)), z

    @staticmethod
    def backward(ctx, image_b, _z_b):
        g = [ctx.checker.grads(s, ctx.sigma, image, z, image_b[i].numpy()) for i, (s, image, z) in enumerate(ctx.views)]
        stack = lambda k: torch.as_tensor(np.stack([x[k] for x in g]))
        return stack("ij_b"), stack("colors_b"), stack("shade_b"), None, None, None, None, None


def _rasterize_with_checker(checker):
    def _rasterize(self, camera, ij, depths, colors, shade, textured, backface_culling):
        if (self.background_image is None) == (self.background_color is None):
            raise BaseException("You need to provide either a background image or background color")
        m, n = self.mesh, camera.n_views
        flags = m.topology.edge_on_silhouette(ij) if self.sigma > 0 else torch.zeros((n, m.nb_faces, 3), dtype=torch.uint8)
        self.last = dict(ij=ij, depths=depths, edgeflags=flags, colors=colors, shade=shade)
        static = dict(
            faces=m.faces_np, faces_uv=m.faces_uv_np if textured else m.faces_np, textured=bool(textured), uv=None if m.uv is None else m.uv.numpy(),
            texture=None if m.texture is None else m.texture.numpy(), height=camera.height, width=camera.width,
            background_color=None if self.background_color is None else np.asarray(self.background_color, dtype=np.float64),
            background_image=None if self.background_image is None else np.asarray(self.background_image, dtype=np.float64), clockwise=m.clockwise,
            backface_culling=bool(backface_culling), perspective_correct=self.perspective_correct, integer_pixel_centers=self.integer_pixel_centers,
        )  # fmt: skip
        return CheckerRenderViews.apply(ij, colors, shade, depths.detach(), flags, static, checker, self.sigma)

    return _rasterize


@contextlib.contextmanager
def emulate(checker):
    """within the block Scene3DDevice rasterizes with `checker` (an oracle.api renderer) on CPU tensors, and the NumPy-level drop-ins
    of deodr_amd.scene3d_compat put their tensors on the CPU"""
    from deodr_amd import scene3d_compat

    def _rasterize_l2(self, camera, ij, depths, colors, shade, textured, backface_culling, obs):
        # (the one-call fit step of the product = render + L2 loss + adjoint; here: the emulated render under autograd)
        image, _z = self._rasterize(camera, ij, depths, colors, shade, textured, backface_culling)
        return ((image.to(torch.float64) - obs.to(torch.float64)) ** 2).sum(), image.detach()

    saved = Scene3DDevice._rasterize, Scene3DDevice._rasterize_l2, scene3d_compat._device
    Scene3DDevice._rasterize, Scene3DDevice._rasterize_l2 = _rasterize_with_checker(checker), _rasterize_l2
    scene3d_compat._device = lambda: torch.device("cpu")
    try:
        yield
    finally:
        Scene3DDevice._rasterize, Scene3DDevice._rasterize_l2, scene3d_compat._device = saved
