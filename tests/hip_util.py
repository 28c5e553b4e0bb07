"""Helpers shared by the GPU parity tests: run a Scene2D through the HIP path and through a CPU checker."""

import numpy as np
import torch

from deodr_amd.hip_renderer import DeviceScene, HipRasterizer


def device_scene(scenes, pixel_dtype=torch.float32, vertex_dtype=torch.float64):
    """One DeviceScene holding `scenes` (list of Scene2D sharing topology / texture / flags) as views."""
    if not isinstance(scenes, (list, tuple)):
        scenes = [scenes]
    s0 = scenes[0]
    stack = lambda name: np.stack([np.asarray(getattr(s, name)) for s in scenes])
    bgi = None if s0.background_image is None else stack("background_image")
    return DeviceScene(
        s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"),
        stack("edgeflags"), s0.height, s0.width, texture=s0.texture, background_color=s0.background_color, background_image=bgi,
        clockwise=s0.clockwise, backface_culling=s0.backface_culling, strict_edge=s0.strict_edge,
        perspective_correct=s0.perspective_correct, integer_pixel_centers=s0.integer_pixel_centers, vertex_dtype=vertex_dtype,
        pixel_dtype=pixel_dtype,
    )  # fmt: skip


def hip_render(scenes, sigma, pixel_dtype=torch.float32, antialiase_error=False, obs=None, rasterizer=None, pool_pairs=0):
    ds = device_scene(scenes, pixel_dtype)
    r = rasterizer or HipRasterizer.for_scene(ds, pool_pairs=pool_pairs)
    obs_t = None if obs is None else torch.as_tensor(np.asarray(obs)).reshape(ds.n_views, ds.height, ds.width, ds.nb_colors)
    out = r.render(ds, sigma, antialiase_error, obs_t, check_overflow=True)
    torch.cuda.synchronize()
    return ds, r, [o.cpu().numpy().astype(np.float64) for o in out]


def hip_grads(ds, r, image_b=None, err_buffer_b=None):
    n = ds.n_views
    if image_b is not None:
        g = r.render_backward(ds, image_b=torch.as_tensor(np.asarray(image_b)).reshape(n, ds.height, ds.width, ds.nb_colors))
    else:
        g = r.render_backward(ds, err_buffer_b=torch.as_tensor(np.asarray(err_buffer_b)).reshape(n, ds.height, ds.width))
    torch.cuda.synchronize()
    return {k: (None if v is None else v.cpu().numpy().astype(np.float64)) for k, v in g.items()}


def rel_err(a, b):
    """max |a - b| relative to max |b| (the tolerance convention of the north star: 1e-4 on gradients)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(float(np.abs(b).max()) if b.size else 0.0, 1e-30)
    return float(np.abs(a - b).max() / scale) if b.size else 0.0


def image_report(image, image_ref, z, z_ref, tol):
    """(max abs error over pixels whose owner agrees, number of pixels whose coverage / depth test flipped)."""
    finite = np.isfinite(z_ref)
    flipped = (np.isfinite(z) != finite) | (finite & np.isfinite(z) & (np.abs(np.where(finite, z, 0) - np.where(finite, z_ref, 0)) > 1e-3 * (1 + np.abs(np.where(finite, z_ref, 0)))))
    err = np.abs(image - image_ref).max(axis=-1)
    return float(err[~flipped].max()) if (~flipped).any() else 0.0, int(flipped.sum())
