"""2.5-D scene container and dispatch, source-compatible with ``deodr.differentiable_renderer``.

Mirrors the reference's L2 layer (deodr/differentiable_renderer.py:16-250 ``Scene2DBase`` / ``renderScene`` /
``renderSceneB`` and :525-734 ``Scene2D``): same constructor keywords, same methods, same in-place buffer
semantics, same exceptions -- but the rasterizer underneath is the HIP library (``deodr_amd.hip_renderer``)
instead of the Cython module.  There is no CPU fallback: without the HIP extension and a GPU every render call
raises.

Array conventions (reference H.h:56-90): ``ij[:,0]`` is x (column), ``ij[:,1]`` is y (row);
``image[y, x, c]``; ``faces`` / ``faces_uv`` are ``uint32 [T,3]``; gradients ``*_b`` are *accumulated into*.
"""

import copy
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np


@dataclass
class Scene2DBase:
    """Field-for-field the structure the rasterizer consumes (reference dr.py:16-45 / H.h:56-90)."""

    faces: np.ndarray
    faces_uv: np.ndarray
    ij: np.ndarray
    depths: np.ndarray
    textured: np.ndarray
    uv: np.ndarray
    shade: np.ndarray
    colors: np.ndarray
    shaded: np.ndarray
    edgeflags: np.ndarray
    height: int
    width: int
    nb_colors: int
    texture: np.ndarray
    background_image: Optional[np.ndarray] = None
    background_color: Optional[np.ndarray] = None
    uv_b: Optional[np.ndarray] = None
    ij_b: Optional[np.ndarray] = None
    shade_b: Optional[np.ndarray] = None
    colors_b: Optional[np.ndarray] = None
    texture_b: Optional[np.ndarray] = None
    clockwise: bool = False
    backface_culling: bool = True
    strict_edge: bool = True
    perspective_correct: bool = False
    integer_pixel_centers: bool = True


def check_scene(scene, image, z_buffer, backward=False, image_b=None, antialiase_error=False, obs=None, err_buffer=None):
    """The shape/dtype contract of the operator boundary (reference dr.py:58-124 and :141-237, pyx:61-114).

    Raises AssertionError exactly where the reference's Python layer would."""
    assert image is not None
    assert z_buffer is not None
    height, width, nb_colors = image.shape[0], image.shape[1], image.shape[2]
    nb_triangles = scene.faces.shape[0]
    nb_vertices = scene.depths.shape[0]
    nb_uv = scene.uv.shape[0]
    assert scene.faces_uv.shape[0] == nb_triangles
    assert scene.faces.dtype == np.uint32
    assert np.all(np.asarray(scene.faces) < nb_vertices)
    assert np.all(np.asarray(scene.faces_uv) < nb_uv)
    for name, ndim in (("colors", 2), ("uv", 2), ("ij", 2), ("shade", 1), ("edgeflags", 2), ("textured", 1), ("shaded", 1)):
        assert getattr(scene, name).ndim == ndim, name
    assert scene.uv.shape[1] == 2
    assert tuple(scene.ij.shape) == (nb_vertices, 2)
    assert scene.shade.shape[0] == nb_vertices
    assert tuple(scene.colors.shape) == (nb_vertices, nb_colors)
    assert tuple(scene.edgeflags.shape) == (nb_triangles, 3)
    assert scene.textured.shape[0] == nb_triangles
    assert scene.shaded.shape[0] == nb_triangles
    assert (scene.background_image is not None) != (
        scene.background_color is not None
    ), "You need to provide either background_image or background_color"
    if scene.background_image is not None:
        assert tuple(scene.background_image.shape) == (height, width, nb_colors)
    else:
        assert scene.background_color.shape[0] == nb_colors
    if np.size(scene.texture) > 0:
        assert scene.texture.ndim == 3
        assert scene.texture.shape[0] > 0 and scene.texture.shape[1] > 0
        assert scene.texture.shape[2] == nb_colors
    assert tuple(z_buffer.shape[:2]) == (height, width)
    if backward:
        for name in ("uv_b", "ij_b", "shade_b", "colors_b"):
            assert getattr(scene, name) is not None, name
        assert tuple(scene.uv_b.shape) == (nb_uv, 2)
        assert tuple(scene.ij_b.shape) == (nb_vertices, 2)
        assert scene.shade_b.shape[0] == nb_vertices
        assert tuple(scene.colors_b.shape) == (nb_vertices, nb_colors)
        if np.size(scene.texture) > 0:
            assert scene.texture_b is not None
            assert tuple(scene.texture_b.shape) == tuple(scene.texture.shape)
    if antialiase_error:
        assert err_buffer is not None, "You need to provide err_buffer"
        assert obs is not None, "You need to provide obs"
        assert tuple(err_buffer.shape[:2]) == (height, width)
        assert tuple(obs.shape[:2]) == (height, width)
        if not backward:
            assert obs.shape[2] == nb_colors
    elif backward:
        assert image_b is not None
        assert tuple(image_b.shape[:2]) == (height, width)


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
        self.faces, self.faces_uv = faces, faces_uv
        self.ij, self.depths = ij, depths
        self.textured, self.uv, self.shade = textured, uv, shade
        self.colors, self.shaded, self.edgeflags = colors, shaded, edgeflags
        self.height, self.width, self.nb_colors = height, width, nb_colors
        self.texture = texture
        self.background_image, self.background_color = background_image, background_color
        self.clockwise, self.backface_culling = clockwise, backface_culling
        self.strict_edge, self.perspective_correct = strict_edge, perspective_correct
        self.integer_pixel_centers = integer_pixel_centers
        for name in ("uv", "ij", "shade", "colors", "texture"):
            setattr(self, name + "_b", np.zeros(np.shape(getattr(self, name))))
        self.store_backward: Tuple = ()

    def clear_gradients(self) -> None:
        for name in ("uv_b", "ij_b", "shade_b", "colors_b", "texture_b"):
            grad = getattr(self, name)
            assert grad is not None
            grad.fill(0)

    def _new_buffers(self):
        return np.zeros((self.height, self.width, self.nb_colors)), np.zeros((self.height, self.width))

    def render(self, sigma: float = 1):
        image, z_buffer = self._new_buffers()
        renderScene(self, sigma, image, z_buffer, False, None, None)
        self.store_backward = (sigma, image, z_buffer)
        return image, z_buffer

    def render_error(self, obs, sigma: float = 1):
        image, z_buffer = self._new_buffers()
        err_buffer = np.empty((self.height, self.width))
        renderScene(self, sigma, image, z_buffer, True, obs, err_buffer)
        self.store_backward = (sigma, obs, image, z_buffer, err_buffer)
        return image, z_buffer, err_buffer

    def _check_differentiable(self):
        if self.perspective_correct:
            raise BaseException("perspective_correct not supported yet for gradient back propagation")
        if not self.backface_culling:
            raise BaseException(
                "use backface_culling=True if you use gradient backpropagation"
                " to get valid gradient through edge anti-aliasing."
            )

    def render_backward(self, image_b, make_copies: bool = True) -> None:
        self._check_differentiable()
        sigma, image, z_buffer = self.store_backward
        # make_copies=False lets the adjoint un-antialiase `image` in place, as the reference does (dr.py:675-699)
        renderSceneB(self, sigma, image.copy() if make_copies else image, z_buffer, image_b, False, None, None, None)

    def render_error_backward(self, err_buffer_b, make_copies: bool = True) -> None:
        self._check_differentiable()
        sigma, obs, image, z_buffer, err_buffer = self.store_backward
        renderSceneB(self, sigma, image, z_buffer, None, True, obs, err_buffer.copy() if make_copies else err_buffer, err_buffer_b)

    def render_compare_and_backward(
        self, obs, sigma: float = 1, antialiase_error: bool = False, mask=None, clear_gradients: bool = True,
        make_copies: bool = True,
    ):  # fmt: skip
        """Render, compare with ``obs`` (sum of squares, optionally masked) and back-propagate. dr.py:701-734."""
        if self.perspective_correct:
            raise BaseException("perspective_correct not supported yet for gradient back propagation")
        if mask is None:
            mask = np.ones((obs.shape[0], obs.shape[1]))
        if antialiase_error:
            image, z_buffer, err_buffer = self.render_error(obs, sigma)
        else:
            image, z_buffer = self.render(sigma)
        if clear_gradients:
            self.clear_gradients()
        if antialiase_error:
            err_buffer = err_buffer * mask
            err = float(np.sum(err_buffer))
            self.render_error_backward(copy.copy(mask), make_copies=make_copies)
        else:
            diff_image = (image - obs) * mask[:, :, None]
            err_buffer = diff_image**2
            err = float(np.sum(err_buffer))
            self.render_backward(2 * diff_image, make_copies=make_copies)
        return image, z_buffer, err_buffer, err


# the 3-D level of the reference's module (Camera, Scene3D: dr.py:250-522, 735-1174) lives in scene3d_compat (adapters over the
# device-resident pipeline of deodr_amd.scene3d); re-exported here so that `from deodr_amd.differentiable_renderer import ...`
# reads like the reference's import
def __getattr__(name):
    if name in ("Camera", "PerspectiveCamera", "default_camera", "Scene3D"):
        from . import scene3d_compat

        return getattr(scene3d_compat, name)
    raise AttributeError(name)
