"""ctypes front-end shared by the two CPU checkers (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

The flat struct mirrors ``struct Scene`` of the reference (C++/DifferentiableRenderer.h:56-90) with
``int`` flags; marshalling follows deodr/differentiable_renderer_cython.pyx:117-172 (private
contiguous float64 / uint32 / uint8 copies of every scene array, gradients copied back after the call,
pyx:406-410).
"""

import ctypes as C
import os
import subprocess
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("DEODR_REFERENCE", "/root/reference")

_dp = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)


class SceneFlat(C.Structure):
    _fields_ = [
        ("faces", _u32p),
        ("faces_uv", _u32p),
        ("depths", _dp),
        ("uv", _dp),
        ("ij", _dp),
        ("shade", _dp),
        ("colors", _dp),
        ("edgeflags", _u8p),
        ("textured", _u8p),
        ("shaded", _u8p),
        ("texture", _dp),
        ("background_image", _dp),
        ("background_color", _dp),
        ("uv_b", _dp),
        ("ij_b", _dp),
        ("shade_b", _dp),
        ("colors_b", _dp),
        ("texture_b", _dp),
        ("nb_triangles", C.c_int),
        ("nb_vertices", C.c_int),
        ("nb_uv", C.c_int),
        ("height", C.c_int),
        ("width", C.c_int),
        ("nb_colors", C.c_int),
        ("texture_height", C.c_int),
        ("texture_width", C.c_int),
        ("clockwise", C.c_int),
        ("backface_culling", C.c_int),
        ("strict_edge", C.c_int),
        ("perspective_correct", C.c_int),
        ("integer_pixel_centers", C.c_int),
    ]


def _f64(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


class _Marshalled:
    """Private contiguous copies of a duck-typed scene (pyx:117-134) + the ctypes struct."""

    def __init__(self, scene, nb_colors, with_grads):
        k = self.keep = SimpleNamespace()
        k.faces = np.ascontiguousarray(np.asarray(scene.faces), dtype=np.uint32).reshape(-1, 3)
        k.faces_uv = np.ascontiguousarray(np.asarray(scene.faces_uv), dtype=np.uint32).reshape(-1, 3)
        k.depths = _f64(scene.depths).reshape(-1)
        k.uv = _f64(scene.uv).reshape(-1, 2)
        k.ij = _f64(scene.ij).reshape(-1, 2)
        k.shade = _f64(scene.shade).reshape(-1)
        k.colors = _f64(scene.colors).reshape(k.depths.shape[0], -1)
        k.edgeflags = np.ascontiguousarray(np.asarray(scene.edgeflags), dtype=np.uint8)
        k.textured = np.ascontiguousarray(np.asarray(scene.textured), dtype=np.uint8)
        k.shaded = np.ascontiguousarray(np.asarray(scene.shaded), dtype=np.uint8)
        tex = _f64(scene.texture)
        if tex.size == 0:
            tex = np.zeros((1, 1, max(nb_colors, 1)))  # never sampled; keeps the pointer non-NULL (H.h:2686)
            th, tw = 0, 0
        else:
            th, tw = tex.shape[0], tex.shape[1]
        k.texture = tex
        s = self.c = SceneFlat()
        s.faces = _ptr(k.faces, _u32p)
        s.faces_uv = _ptr(k.faces_uv, _u32p)
        s.depths = _ptr(k.depths, _dp)
        s.uv = _ptr(k.uv, _dp)
        s.ij = _ptr(k.ij, _dp)
        s.shade = _ptr(k.shade, _dp)
        s.colors = _ptr(k.colors, _dp)
        s.edgeflags = _ptr(k.edgeflags, _u8p)
        s.textured = _ptr(k.textured, _u8p)
        s.shaded = _ptr(k.shaded, _u8p)
        s.texture = _ptr(k.texture, _dp)
        bgi = getattr(scene, "background_image", None)
        bgc = getattr(scene, "background_color", None)
        assert (bgi is None) != (bgc is None)
        if bgi is not None:
            k.bgi = _f64(bgi)
            s.background_image = _ptr(k.bgi, _dp)
        else:
            k.bgc = _f64(bgc).reshape(-1)
            assert k.bgc.shape[0] == nb_colors
            s.background_color = _ptr(k.bgc, _dp)
        if with_grads:
            k.uv_b = _f64(scene.uv_b).reshape(k.uv.shape).copy()
            k.ij_b = _f64(scene.ij_b).reshape(k.ij.shape).copy()
            k.shade_b = _f64(scene.shade_b).reshape(k.shade.shape).copy()
            k.colors_b = _f64(scene.colors_b).reshape(k.colors.shape).copy()
            tb = _f64(scene.texture_b)
            k.texture_b = tb.copy() if tb.size else np.zeros_like(k.texture)
            s.uv_b = _ptr(k.uv_b, _dp)
            s.ij_b = _ptr(k.ij_b, _dp)
            s.shade_b = _ptr(k.shade_b, _dp)
            s.colors_b = _ptr(k.colors_b, _dp)
            s.texture_b = _ptr(k.texture_b, _dp)
        s.nb_triangles = k.faces.shape[0]
        s.nb_vertices = k.depths.shape[0]
        s.nb_uv = k.uv.shape[0]
        s.height = int(scene.height)
        s.width = int(scene.width)
        s.nb_colors = nb_colors
        s.texture_height = th
        s.texture_width = tw
        s.clockwise = int(bool(scene.clockwise))
        s.backface_culling = int(bool(scene.backface_culling))
        s.strict_edge = int(bool(scene.strict_edge))
        s.perspective_correct = int(bool(scene.perspective_correct))
        s.integer_pixel_centers = int(bool(scene.integer_pixel_centers))


class CpuRenderer:
    """``renderSceneCpp`` / ``renderSceneBCpp`` semantics (pyx:50-57, 206-215) on a CPU checker library."""

    def __init__(self, path, prefix):
        self.path = path
        self.lib = C.CDLL(path)
        self._fwd = getattr(self.lib, prefix + "_render_scene")
        self._bwd = getattr(self.lib, prefix + "_render_scene_b")
        self._err = getattr(self.lib, prefix + "_last_error")
        self._err.restype = C.c_char_p
        self._fwd.restype = C.c_int
        self._bwd.restype = C.c_int
        self._fwd.argtypes = [C.POINTER(SceneFlat), _dp, _dp, C.c_double, C.c_int, _dp, _dp]
        self._bwd.argtypes = [C.POINTER(SceneFlat), _dp, _dp, _dp, C.c_double, C.c_int, _dp, _dp, _dp]

    # in-place semantics exactly like the Cython entry points -------------------------------------
    def renderSceneCpp(self, scene, sigma, image, z_buffer, antialiase_error=False, obs=None, err_buffer=None):
        assert image.dtype == np.float64 and image.flags.c_contiguous and image.ndim == 3
        assert z_buffer.dtype == np.float64 and z_buffer.flags.c_contiguous
        m = _Marshalled(scene, image.shape[2], with_grads=False)
        assert image.shape[:2] == (m.c.height, m.c.width) == z_buffer.shape
        obs_p = err_p = None
        if antialiase_error:
            assert obs.dtype == np.float64 and obs.flags.c_contiguous and obs.shape == image.shape
            assert err_buffer.dtype == np.float64 and err_buffer.flags.c_contiguous
            obs_p, err_p = _ptr(obs, _dp), _ptr(err_buffer, _dp)
        rc = self._fwd(C.byref(m.c), _ptr(image, _dp), _ptr(z_buffer, _dp), float(sigma), int(antialiase_error), obs_p, err_p)
        if rc:
            raise RuntimeError(self._err().decode())

    def renderSceneBCpp(
        self, scene, sigma, image, z_buffer, image_b=None, antialiase_error=False, obs=None, err_buffer=None, err_buffer_b=None
    ):
        """Mutates image / image_b / err_buffer / err_buffer_b like the reference and rebinds scene.*_b (pyx:406-410)."""
        assert image.dtype == np.float64 and image.flags.c_contiguous
        m = _Marshalled(scene, image.shape[2], with_grads=True)
        ib = ob = eb = ebb = None
        if antialiase_error:
            for a in (obs, err_buffer, err_buffer_b):
                assert a.dtype == np.float64 and a.flags.c_contiguous
            ob, eb, ebb = _ptr(obs, _dp), _ptr(err_buffer, _dp), _ptr(err_buffer_b, _dp)
        else:
            assert image_b.dtype == np.float64 and image_b.flags.c_contiguous and image_b.shape == image.shape
            ib = _ptr(image_b, _dp)
        rc = self._bwd(C.byref(m.c), _ptr(image, _dp), _ptr(z_buffer, _dp), ib, float(sigma), int(antialiase_error), ob, eb, ebb)
        if rc:
            raise RuntimeError(self._err().decode())
        k = m.keep
        scene.uv_b = k.uv_b.reshape(np.shape(scene.uv_b))
        scene.ij_b = k.ij_b.reshape(np.shape(scene.ij_b))
        scene.shade_b = k.shade_b.reshape(np.shape(scene.shade_b))
        scene.colors_b = k.colors_b.reshape(np.shape(scene.colors_b))
        if np.size(scene.texture_b):
            scene.texture_b = k.texture_b.reshape(np.shape(scene.texture_b))

    # functional conveniences used by the tests ----------------------------------------------------
    def render(self, scene, sigma, antialiase_error=False, obs=None):
        nb_colors = int(scene.nb_colors) if getattr(scene, "nb_colors", None) else np.shape(scene.colors)[1]
        image = np.zeros((scene.height, scene.width, nb_colors))
        z = np.zeros((scene.height, scene.width))
        err = np.zeros((scene.height, scene.width)) if antialiase_error else None
        obs = _f64(obs) if obs is not None else None
        self.renderSceneCpp(scene, sigma, image, z, antialiase_error, obs, err)
        return (image, z, err) if antialiase_error else (image, z)

    def grads(self, scene, sigma, image, z_buffer, image_b=None, antialiase_error=False, obs=None, err_buffer=None, err_buffer_b=None):
        """Fresh-zero gradients of one backward call; inputs are copied so nothing the caller holds is mutated."""
        sc = SimpleNamespace(**{k: getattr(scene, k) for k in _SCENE_KEYS if hasattr(scene, k)})
        sc.uv_b = np.zeros(np.shape(scene.uv))
        sc.ij_b = np.zeros(np.shape(scene.ij))
        sc.shade_b = np.zeros(np.shape(scene.shade))
        sc.colors_b = np.zeros(np.shape(scene.colors))
        sc.texture_b = np.zeros(np.shape(scene.texture))
        cp = lambda a: None if a is None else _f64(a).copy()
        self.renderSceneBCpp(sc, sigma, cp(image), cp(z_buffer), cp(image_b), antialiase_error, cp(obs), cp(err_buffer), cp(err_buffer_b))
        return {"ij_b": sc.ij_b, "colors_b": sc.colors_b, "uv_b": sc.uv_b, "shade_b": sc.shade_b, "texture_b": sc.texture_b}


_SCENE_KEYS = [
    "faces", "faces_uv", "ij", "depths", "textured", "uv", "shade", "colors", "shaded", "edgeflags", "height", "width",
    "nb_colors", "texture", "background_image", "background_color", "clockwise", "backface_culling", "strict_edge",
    "perspective_correct", "integer_pixel_centers",
]  # fmt: skip


def _make(target=None):
    cmd = ["make", "-C", HERE] + ([target] if target else [])
    subprocess.run(cmd, check=True, capture_output=True)


def build_port():
    _make()
    return os.path.join(HERE, "libdeodr_oracle.so")


def build_ref():
    """(Re)build oracle/_ref from the reference sources where they lie; no-op when /root/reference is absent."""
    if os.path.isdir(REFERENCE):
        _make("ref")
    return os.path.join(HERE, "_ref", "libdeodr_ref.so")


_cache = {}


def port(fixed=False):
    """Our C restatement.  fixed=False reproduces the reference as shipped (defects D1, D2 included);
    fixed=True repairs them.  The switch is a process-wide global of the library, set on every call here."""
    if "port" not in _cache:
        _cache["port"] = CpuRenderer(build_port(), "deodr_oracle")
    _cache["port"].lib.deodr_oracle_set_reference_defects(0 if fixed else 1)
    return _cache["port"]


def min_abs_T(reset=True):
    """Smallest |T| the restatement's adjoint divided by since the last reset (diagnostic: see deodr_oracle_min_abs_T)."""
    lib = port().lib
    lib.deodr_oracle_min_abs_T.restype = C.c_double
    lib.deodr_oracle_min_abs_T.argtypes = [C.c_int]
    return float(lib.deodr_oracle_min_abs_T(int(reset)))


def ref(fixed=False):
    """The real reference (or None when oracle/_ref was never built and /root/reference is absent).

    fixed=True: the build with the two adjoint defects repaired (oracle/Makefile D1, D2)."""
    key = "ref_fixed" if fixed else "ref"
    if key not in _cache:
        path = os.path.join(HERE, "_ref", "libdeodr_ref_fixed.so" if fixed else "libdeodr_ref.so")
        if not os.path.exists(path) and os.path.isdir(REFERENCE):
            build_ref()
        _cache[key] = CpuRenderer(path, "deodr_ref") if os.path.exists(path) else None
    return _cache[key]
