"""2.5-D scene container and dispatch, source-compatible with ``deodr.differentiable_renderer``.

Mirrors the reference's L2 layer (deodr/differentiable_renderer.py:16-250 ``Scene2DBase`` / ``renderScene`` /
``renderSceneB`` and :525-734 ``Scene2D``): same constructor keywords, same methods, same in-place buffer
semantics, same exceptions -- but the rasterizer underneath is the HIP library (``deodr_amd.hip_renderer``)
instead of the Cython module.  There is no CPU fallback: without the HIP extension and a GPU every render call
raises.

Array conventions (reference H.h:56-90): ``ij[:,0]`` is x (column), ``ij[:,1]`` is y (row);
``image[y, x, c]``; ``faces`` / ``faces_uv`` are ``uint32 [T,3]``; gradients ``*_b`` are *accumulated into*.
"""

from dataclasses import field, make_dataclass
from typing import Optional

import numpy as np

# The structure the rasterizer consumes (reference dr.py:16-45 / H.h:56-90), generated from three tables: the arrays every scene
# has, the optional arrays (backgrounds, adjoints) and the flags with their defaults.
_REQUIRED = ("faces", "faces_uv", "ij", "depths", "textured", "uv", "shade", "colors", "shaded", "edgeflags")
_SIZES = ("height", "width", "nb_colors")
_OPTIONAL = ("background_image", "background_color", "uv_b", "ij_b", "shade_b", "colors_b", "texture_b")
_FLAGS = {"clockwise": False, "backface_culling": True, "strict_edge": True, "perspective_correct": False, "integer_pixel_centers": True}
_GRADIENTS = tuple(n for n in _OPTIONAL if n.endswith("_b"))

Scene2DBase = make_dataclass(
    "Scene2DBase",
    [(n, np.ndarray) for n in _REQUIRED] + [(n, int) for n in _SIZES] + [("texture", np.ndarray)]
    + [(n, Optional[np.ndarray], field(default=None)) for n in _OPTIONAL] + [(n, bool, field(default=v)) for n, v in _FLAGS.items()],
)  # fmt: skip
Scene2DBase.__doc__ = "Field for field the scene structure of the operator boundary (reference dr.py:16-45 / H.h:56-90)."


def _shape_is(array, *dims):
    return tuple(array.shape[: len(dims)]) == tuple(dims)


def check_scene(scene, image, z_buffer, backward=False, image_b=None, antialiase_error=False, obs=None, err_buffer=None):
    """The shape / dtype contract of the operator boundary (reference dr.py:58-124 and :141-237, pyx:61-114): an
    AssertionError wherever the reference's Python layer raises one."""
    assert image is not None and z_buffer is not None, "image and z_buffer are caller-allocated"
    H, W, C = image.shape[:3]
    T, V, Vuv = scene.faces.shape[0], scene.depths.shape[0], scene.uv.shape[0]
    assert scene.faces.dtype == np.uint32, "faces must be uint32 (pyx:75)"
    assert scene.faces_uv.shape[0] == T
    assert np.all(np.asarray(scene.faces) < V) and np.all(np.asarray(scene.faces_uv) < Vuv), "face index out of range"
    expected = {"ij": (V, 2), "uv": (Vuv, 2), "colors": (V, C), "shade": (V,), "edgeflags": (T, 3), "textured": (T,), "shaded": (T,)}
    for name, shape in expected.items():
        a = getattr(scene, name)
        assert a.ndim == len(shape) and _shape_is(a, *shape), f"scene.{name}: expected shape {shape}, got {tuple(a.shape)}"
    has_image, has_color = scene.background_image is not None, scene.background_color is not None
    assert has_image != has_color, "You need to provide either background_image or background_color"
    if has_image:
        assert _shape_is(scene.background_image, H, W, C)
    else:
        assert scene.background_color.shape[0] == C
    textured_scene = np.size(scene.texture) > 0
    if textured_scene:
        assert scene.texture.ndim == 3 and min(scene.texture.shape[:2]) > 0 and scene.texture.shape[2] == C
    assert _shape_is(z_buffer, H, W)
    if backward:
        for name, like in (("uv_b", "uv"), ("ij_b", "ij"), ("shade_b", "shade"), ("colors_b", "colors")):
            g = getattr(scene, name)
            assert g is not None and tuple(g.shape) == tuple(getattr(scene, like).shape), name
        if textured_scene:
            assert scene.texture_b is not None and tuple(scene.texture_b.shape) == tuple(scene.texture.shape)
    if antialiase_error:
        assert err_buffer is not None, "You need to provide err_buffer"
        assert obs is not None, "You need to provide obs"
        assert _shape_is(err_buffer, H, W) and _shape_is(obs, H, W)
        assert backward or obs.shape[2] == C
    elif backward:
        assert image_b is not None and _shape_is(image_b, H, W)


def renderScene(scene, sigma, image, z_buffer, antialiase_error=False, obs=None, err_buffer=None, check_valid=True):
    """Forward render into caller-owned ``image`` / ``z_buffer`` (/ ``err_buffer``). Reference dr.py:48-126."""
    from . import hip_renderer

    if check_valid:
        check_scene(scene, image, z_buffer, False, None, antialiase_error, obs, err_buffer)
    hip_renderer.renderSceneCpp(scene, sigma, image, z_buffer, antialiase_error, obs, err_buffer, check_valid=False)


def renderSceneB(
    scene, sigma, image, z_buffer, image_b=None, antialiase_error=False, obs=None, err_buffer=None, err_buffer_b=None,
    check_valid=True,
):  # fmt: skip
    """Adjoint pass; accumulates into ``scene.{ij,colors,uv,shade,texture}_b``. Reference dr.py:129-249."""
    from . import hip_renderer

    if check_valid:
        check_scene(scene, image, z_buffer, True, image_b, antialiase_error, obs, err_buffer)
    hip_renderer.renderSceneBCpp(scene, sigma, image, z_buffer, image_b, antialiase_error, obs, err_buffer, err_buffer_b, check_valid=False)


class Scene2D(Scene2DBase):
    """A 2.5-D scene: 2-D vertices with depths, triangles, per-vertex colours or a Gouraud-shaded texture.

    Same constructor and methods as the reference class (dr.py:525-734).  Pixel-centre convention: with
    ``integer_pixel_centers=True`` (default) pixel (col, row) is sampled at (col, row); otherwise at
    (col + 0.5, row + 0.5)."""

    def __init__(
        self, faces, faces_uv, ij, depths, textured, uv, shade, colors, shaded, edgeflags, height, width, nb_colors, texture,
        background_image=None, background_color=None, clockwise=False, backface_culling=False, strict_edge=True,
        perspective_correct=False, integer_pixel_centers=True,
    ):  # fmt: skip
        given = dict(locals())
        for name in _REQUIRED + _SIZES + ("texture", "background_image", "background_color") + tuple(_FLAGS):
            setattr(self, name, given[name])
        for name in _GRADIENTS:  # adjoints of uv / ij / shade / colors / texture, accumulated into by the backward calls
            setattr(self, name, np.zeros(np.shape(getattr(self, name[:-2]))))
        self.store_backward = ()  # what the last forward left for its adjoint (dr.py:618-627)

    def clear_gradients(self) -> None:
        for name in _GRADIENTS:
            getattr(self, name).fill(0)

    def _frame(self):
        return np.zeros((self.height, self.width, self.nb_colors)), np.zeros((self.height, self.width))

    def render(self, sigma: float = 1):
        """-> (image, z_buffer); dr.py:612-627"""
        image, z_buffer = self._frame()
        renderScene(self, sigma, image, z_buffer)
        self.store_backward = (sigma, image, z_buffer)
        return image, z_buffer

    def render_error(self, obs, sigma: float = 1):
        """-> (image, z_buffer, err_buffer) with the squared residual antialiased instead of the image; dr.py:629-663"""
        image, z_buffer = self._frame()
        err_buffer = np.empty(z_buffer.shape)
        renderScene(self, sigma, image, z_buffer, antialiase_error=True, obs=obs, err_buffer=err_buffer)
        self.store_backward = (sigma, obs, image, z_buffer, err_buffer)
        return image, z_buffer, err_buffer

    def _require_differentiable(self):
        """the two pre-flight checks of the reference's backward methods (dr.py:666-672), same exception type"""
        if self.perspective_correct:
            raise BaseException("perspective_correct not supported yet for gradient back propagation")
        if not self.backface_culling:
            raise BaseException("use backface_culling=True if you use gradient backpropagation to get valid gradient through edge anti-aliasing.")

    def render_backward(self, image_b, make_copies: bool = True) -> None:
        """dr.py:665-699.  (make_copies=False let the reference un-antialiase `image` in place; this adjoint mutates nothing)"""
        self._require_differentiable()
        sigma, image, z_buffer = self.store_backward
        renderSceneB(self, sigma, image.copy() if make_copies else image, z_buffer, image_b=image_b)

    def render_error_backward(self, err_buffer_b, make_copies: bool = True) -> None:
        self._require_differentiable()
        sigma, obs, image, z_buffer, err_buffer = self.store_backward
        renderSceneB(self, sigma, image, z_buffer, antialiase_error=True, obs=obs, err_buffer=err_buffer.copy() if make_copies else err_buffer,
                     err_buffer_b=err_buffer_b)  # fmt: skip

    def render_compare_and_backward(
        self, obs, sigma: float = 1, antialiase_error: bool = False, mask=None, clear_gradients: bool = True,
        make_copies: bool = True,
    ):  # fmt: skip
        """Render, compare with ``obs`` (sum of squares, optionally masked) and back-propagate -> (image, z_buffer, err_buffer, err);
        dr.py:701-734."""
        if self.perspective_correct:
            raise BaseException("perspective_correct not supported yet for gradient back propagation")
        weights = np.ones(obs.shape[:2]) if mask is None else mask
        forward = self.render_error(obs, sigma) if antialiase_error else self.render(sigma)
        image, z_buffer = forward[0], forward[1]
        if clear_gradients:
            self.clear_gradients()
        if antialiase_error:
            err_buffer = forward[2] * weights
            self.render_error_backward(np.array(weights, copy=True), make_copies=make_copies)
        else:
            residual = (image - obs) * weights[:, :, None]
            err_buffer = residual**2
            self.render_backward(2 * residual, make_copies=make_copies)
        return image, z_buffer, err_buffer, float(np.sum(err_buffer))


# the 3-D level of the reference's module (Camera, Scene3D: dr.py:250-522, 735-1174) lives in scene3d_compat (adapters over the
# device-resident pipeline of deodr_amd.scene3d); re-exported here so that `from deodr_amd.differentiable_renderer import ...`
# reads like the reference's import
def __getattr__(name):
    if name in ("Camera", "PerspectiveCamera", "default_camera", "Scene3D"):
        from . import scene3d_compat

        return getattr(scene3d_compat, name)
    raise AttributeError(name)
