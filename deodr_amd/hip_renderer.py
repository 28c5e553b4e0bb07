"""Host side of the HIP rasterizer: ctypes binding of ``libdeodr_hip.so`` (C ABI in ``include/deodr_hip.h``).

Two levels, both without any CPU fallback (a missing library or a missing GPU raises):

* :class:`DeviceScene` + :class:`HipRasterizer` -- the device-resident path: PyTorch-ROCm tensors in, PyTorch-ROCm tensors
  out, nothing crosses PCIe, calls are asynchronous on the current torch stream, ``n_views`` views per launch.
* :func:`renderSceneCpp` / :func:`renderSceneBCpp` -- drop-in replacements of the reference's Cython entry points
  (deodr/differentiable_renderer_cython.pyx:50-57 and :206-215): duck-typed ``scene`` with NumPy (or CPU torch) arrays,
  caller-owned float64 output buffers written in place, ``scene.*_b`` rebound to ``old + new`` (pyx:406-410).  They copy to
  the GPU, run the same kernels with float64 storage and copy back; the call is complete on return.

torch is used for device memory and streams only.
"""

import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdeodr_hip.so")
ABI_VERSION = 12
ERR_FACES, ERR_FACES_UV, ERR_NO_TEXTURE, ERR_INTERNAL, ERR_DET_RANGE = 1, 2, 4, 8, 16  # include/deodr_hip.h DEODR_HIP_ERR_*
_STATUS_NEEDED, _STATUS_ERRORS = 11, 12  # words of the 64-byte status block at the start of the workspace


class _SceneC(C.Structure):
    _fields_ = (
        [(n, C.c_void_p) for n in ("faces", "faces_uv", "textured", "shaded", "depths", "ij", "shade", "colors", "edgeflags", "uv")]
        + [(n, C.c_void_p) for n in ("texture", "background_image", "background_color")]
        + [(n, C.c_void_p) for n in ("uv_b", "ij_b", "shade_b", "colors_b", "texture_b")]
        + [(n, C.c_int) for n in ("nb_triangles", "nb_vertices", "nb_uv", "height", "width", "nb_colors", "texture_height", "texture_width")]
        + [(n, C.c_int) for n in ("clockwise", "backface_culling", "strict_edge", "perspective_correct", "integer_pixel_centers")]
        + [(n, C.c_int) for n in ("n_views", "vertex_dtype", "pixel_dtype", "deterministic")]
    )


_lib = None


class _FitOptionsC(C.Structure):
    """include/deodr_hip.h::DeodrHipFitOptions"""

    _fields_ = [("tile_loss", C.c_void_p), ("loss", C.c_void_p), ("loss_scratch", C.c_void_p), ("clamp", C.c_int), ("clamp_lo", C.c_double),
                ("clamp_hi", C.c_double), ("done_flag", C.c_void_p), ("done_value", C.c_uint32)]  # fmt: skip


def lib():
    """The HIP library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). deodr_amd has no CPU fallback."
            )
        L = C.CDLL(LIB_PATH)
        L.deodr_hip_abi_version.restype = C.c_int
        if L.deodr_hip_abi_version() != ABI_VERSION:
            raise ImportError("libdeodr_hip.so ABI version mismatch; rebuild it")
        L.deodr_hip_last_error.restype = C.c_char_p
        L.deodr_hip_workspace_bytes.restype = C.c_size_t
        L.deodr_hip_workspace_bytes.argtypes = [C.c_int] * 5 + [C.c_size_t]
        L.deodr_hip_render_scene.restype = C.c_int
        L.deodr_hip_render_scene.argtypes = [C.POINTER(_SceneC), C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_size_t, C.c_void_p]  # fmt: skip
        L.deodr_hip_render_scene_b.restype = C.c_int
        L.deodr_hip_render_scene_b.argtypes = [C.POINTER(_SceneC), C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]  # fmt: skip
        L.deodr_hip_render_scene_fit.restype = C.c_int
        L.deodr_hip_render_scene_fit.argtypes = [C.POINTER(_SceneC), C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_void_p,
                                                 C.c_size_t, C.c_void_p]  # fmt: skip
        L.deodr_hip_fit_loss_bytes.restype, L.deodr_hip_fit_loss_bytes.argtypes = C.c_size_t, [C.c_int] * 3
        L.deodr_hip_background_loss.restype = C.c_int
        L.deodr_hip_background_loss.argtypes = [C.POINTER(_SceneC), C.c_void_p, C.POINTER(_FitOptionsC), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.deodr_hip_render_scene_fit_ex.restype = C.c_int
        L.deodr_hip_render_scene_fit_ex.argtypes = [C.POINTER(_SceneC), C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.POINTER(_FitOptionsC),
                                                    C.c_void_p, C.c_size_t, C.c_void_p]  # fmt: skip
        L.deodr_hip_wait_flag.restype = C.c_int
        L.deodr_hip_wait_flag.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_double, C.c_void_p]
        L.deodr_hip_workspace_status.restype = C.c_int
        L.deodr_hip_workspace_status.argtypes = [C.POINTER(_SceneC), C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int),
                                                 C.POINTER(C.c_ulonglong), C.POINTER(C.c_int)]  # fmt: skip
        L.deodr_hip_workspace_census.restype = C.c_int
        L.deodr_hip_workspace_census.argtypes = [C.POINTER(_SceneC), C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_ulonglong),
                                                 C.POINTER(C.c_ulonglong)]  # fmt: skip
        L.deodr_hip_workspace_pool_pairs.restype = C.c_int
        L.deodr_hip_workspace_pool_pairs.argtypes = [C.POINTER(_SceneC), C.c_size_t, C.POINTER(C.c_ulonglong)]
        L.deodr_hip_profile_stamps.restype, L.deodr_hip_profile_stamps.argtypes = C.c_int, [C.c_void_p, C.c_int]
        L.deodr_hip_copy_probe.restype = C.c_int
        L.deodr_hip_copy_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def set_deterministic(on):
    """``deodr_hip_set_deterministic``: integer accumulation on the un-staged kernels -- gradients bit-identical from run to run (slow;
    for tests and for debugging an optimiser).  Process-wide; ``DeviceScene(deterministic=True)`` asks for it per scene."""
    lib().deodr_hip_set_deterministic(int(bool(on)))


def wait_flag(flag, value, status=None, timeout=1.0, stream=None):
    """``deodr_hip_wait_flag``: the current stream of ``flag``'s device (or ``stream``) waits until ``flag`` (a 4-byte integer tensor a fit step
    was given as ``done_flag``) has reached ``value``.  The step must have been queued before this call.  ``status``: a 4-byte tensor set to 1 by a
    wait that gave up after ``timeout`` seconds (check it where you synchronise)."""
    dev = flag.device
    with torch.cuda.device(dev):
        _check(lib().deodr_hip_wait_flag(_ptr(flag), int(value) & 0xFFFFFFFF, _ptr(status), float(timeout),
                                         C.c_void_p(stream.cuda_stream) if stream is not None else _stream(dev)))  # fmt: skip


def force_generic(on):
    """Test hook (``deodr_hip_force_generic``): route every call through the un-staged kernels.  Process-wide."""
    lib().deodr_hip_force_generic(int(bool(on)))


def tile_census(rasterizer, ds):
    """(tiles with a primitive, tiles with silhouette edges) of the last forward on `rasterizer`, over all views (synchronises)."""
    sc = ds.c_struct()
    a, b = C.c_ulonglong(0), C.c_ulonglong(0)
    with torch.cuda.device(rasterizer.device):
        _check(lib().deodr_hip_workspace_census(C.byref(sc), _ptr(rasterizer.workspace), rasterizer.nbytes, _stream(rasterizer.device),
                                                C.byref(a), C.byref(b)))  # fmt: skip
    return int(a.value), int(b.value)


def scene_error_message(bits):
    what = []
    if bits & ERR_FACES:
        what.append("an entry of scene.faces is >= the number of vertices")
    if bits & ERR_FACES_UV:
        what.append("an entry of scene.faces_uv is >= the number of uv vertices")
    if bits & ERR_NO_TEXTURE:
        what.append("a triangle is textured and shaded but the scene has no texture")
    if bits & ERR_INTERNAL:
        what.append("internal: a finalize workgroup of a fit step gave up waiting for the tile walkers (gradients incomplete)")
    det = ("deterministic mode: a gradient contribution or running sum of the last deterministic adjoint left the fixed-point range +- 2^31 "
           "(the gradients of that call are wrong; the bit is cleared when the next deterministic adjoint starts)")
    if bits & ERR_DET_RANGE and not what:
        return det  # (not a statement about the scene)
    if bits & ERR_DET_RANGE:
        what.append(det)
    return "invalid scene (checkSceneValid): " + "; ".join(what)


def _check(rc):
    if rc:
        raise RuntimeError("deodr_hip: " + lib().deodr_hip_last_error().decode())


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _resolve_device(device):
    """the ROCm device a scene or a workspace lives on (there is no CPU path: anything else is refused)"""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("deodr_amd needs a ROCm device; there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device()) if dev.index is None else dev


def _count(a):
    """number of elements of an array OR a tensor (np.size of a tensor is its bound `size` method, not a number)"""
    return 0 if a is None else int(a.numel()) if torch.is_tensor(a) else int(np.size(a))


def _on(t, device, dtype, shape, what):
    """`t` as a contiguous tensor of `dtype` on `device` with `shape` (no copy when it already is one)."""
    if not torch.is_tensor(t):
        t = torch.as_tensor(np.asarray(t))
    t = t.to(device=device, dtype=dtype)
    if tuple(t.shape) != tuple(shape):
        if t.numel() != int(np.prod(shape)):
            raise ValueError(f"{what}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
        t = t.reshape(shape)
    return t.contiguous()


class DeviceScene:
    """The arrays of ``struct Scene`` (reference H.h:56-90) as contiguous ROCm tensors, for ``n_views`` views of one mesh.

    Shapes: faces / faces_uv ``[T,3] int32|uint32``; textured / shaded ``[T] uint8|bool``; uv ``[Vuv,2]``;
    per view (leading dim ``n_views``, optional for one view): ij ``[n,V,2]``, depths ``[n,V]``, colors ``[n,V,C]``,
    shade ``[n,V]``, edgeflags ``[n,T,3]``; texture ``[Ht,Wt,C]`` or None; background_color ``[C]`` or
    background_image ``[n,H,W,C]``.  Vertex arrays share one float dtype, pixel arrays another."""

    def __init__(self, faces, faces_uv, textured, shaded, uv, ij, depths, colors, shade, edgeflags, height, width, texture=None,
                 background_color=None, background_image=None, clockwise=False, backface_culling=True, strict_edge=True,
                 perspective_correct=False, integer_pixel_centers=True, vertex_dtype=torch.float64, pixel_dtype=torch.float32,
                 device="cuda", validate=True, deterministic=False):  # fmt: skip
        dev = _resolve_device(device)
        self.device, self.vertex_dtype, self.pixel_dtype = dev, vertex_dtype, pixel_dtype
        # integer accumulation for the calls on THIS scene (DeodrHipScene::deterministic): gradients bit-identical from run to run, several times
        # slower; may be switched at any time (it is read when a call is made).  set_deterministic() is the process-wide switch.
        self.deterministic = bool(deterministic)
        as_t = lambda a, dt: torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).to(device=dev, dtype=dt).contiguous()
        self.faces = as_t(np.asarray(faces).astype(np.int64) if not torch.is_tensor(faces) else faces, torch.int32)
        self.faces_uv = as_t(np.asarray(faces_uv).astype(np.int64) if not torch.is_tensor(faces_uv) else faces_uv, torch.int32)
        self.textured = as_t(textured, torch.uint8)
        self.shaded = as_t(shaded, torch.uint8)
        self.uv = as_t(uv, vertex_dtype).reshape(-1, 2)
        self.height, self.width = int(height), int(width)
        self.flags = dict(clockwise=bool(clockwise), backface_culling=bool(backface_culling), strict_edge=bool(strict_edge),
                          perspective_correct=bool(perspective_correct), integer_pixel_centers=bool(integer_pixel_centers))  # fmt: skip
        self.texture = None
        if _count(texture) > 0:
            self.texture = as_t(texture, pixel_dtype)
        self.background_color = None if background_color is None else as_t(background_color, pixel_dtype).reshape(-1)
        self.background_image = None if background_image is None else as_t(background_image, pixel_dtype)
        self.set_views(ij, depths, colors, shade, edgeflags)
        if self.background_image is not None:
            self.background_image = self.background_image.reshape(self.n_views, self.height, self.width, self.nb_colors).contiguous()
        if validate:
            self.validate()

    def validate(self):
        """checkSceneValid's index checks (reference H.h:2700-2712), once per topology (synchronises).  The set-up kernel
        makes the same checks on every forward and raises the workspace's sticky error word; this one fails early."""
        V, Vuv = int(self.depths.shape[1]), int(self.uv.shape[0])
        if self.nb_triangles:
            # int32 storage of uint32 indices: a negative value is an index >= 2^31
            if int(self.faces.min()) < 0 or int(self.faces.max()) >= V:
                raise ValueError(scene_error_message(ERR_FACES))
            if int(self.faces_uv.min()) < 0 or int(self.faces_uv.max()) >= Vuv:
                raise ValueError(scene_error_message(ERR_FACES_UV))
            if self.texture is None and bool((self.textured.bool() & self.shaded.bool()).any()):
                raise ValueError(scene_error_message(ERR_NO_TEXTURE))

    def set_views(self, ij=None, depths=None, colors=None, shade=None, edgeflags=None):
        """Replace per-view arrays (tensors are used as they are when already contiguous on the device)."""
        dev, vd = self.device, self.vertex_dtype
        conv = lambda a, dt: (a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))).to(device=dev, dtype=dt).contiguous()
        if depths is not None:
            d = conv(depths, vd)
            self.depths = d.reshape(1, -1) if d.dim() == 1 else d
        n, V = self.depths.shape
        self.n_views = n
        if ij is not None:
            self.ij = conv(ij, vd).reshape(n, V, 2)
        if colors is not None:
            self.colors = conv(colors, vd).reshape(n, V, -1)
        if shade is not None:
            self.shade = conv(shade, vd).reshape(n, V)
        if edgeflags is not None:
            self.edgeflags = conv(edgeflags, torch.uint8).reshape(n, -1, 3)
        self.nb_colors = int(self.colors.shape[2])

    @property
    def nb_triangles(self):
        return int(self.faces.shape[0])

    def zero_grads(self):
        vd, pd, dev = self.vertex_dtype, self.pixel_dtype, self.device
        g = dict(
            ij_b=torch.zeros_like(self.ij), colors_b=torch.zeros_like(self.colors), shade_b=torch.zeros_like(self.shade),
            uv_b=torch.zeros_like(self.uv), texture_b=None if self.texture is None else torch.zeros_like(self.texture),
        )  # fmt: skip
        return g

    def c_struct(self, grads=None):
        s = _SceneC()
        for name in ("faces", "faces_uv", "textured", "shaded", "depths", "ij", "shade", "colors", "edgeflags", "uv", "texture",
                     "background_image", "background_color"):  # fmt: skip
            setattr(s, name, _ptr(getattr(self, name)))
        if grads is not None:
            for name in ("uv_b", "ij_b", "shade_b", "colors_b", "texture_b"):
                setattr(s, name, _ptr(grads[name]))
        s.nb_triangles, s.nb_vertices, s.nb_uv = self.nb_triangles, int(self.depths.shape[1]), int(self.uv.shape[0])
        s.height, s.width, s.nb_colors = self.height, self.width, self.nb_colors
        if self.texture is not None:
            s.texture_height, s.texture_width = int(self.texture.shape[0]), int(self.texture.shape[1])
        for k, v in self.flags.items():
            setattr(s, k, int(v))
        s.n_views = self.n_views
        s.vertex_dtype = 1 if self.vertex_dtype == torch.float64 else 0
        s.pixel_dtype = 1 if self.pixel_dtype == torch.float64 else 0
        s.deterministic = 1 if self.deterministic else 0
        return s


class HipRasterizer:
    """Owns the device workspace of one scene shape and runs renderScene / renderScene_B on it.

    The workspace keeps the forward state (per-primitive records, tile lists, per-pixel owner ids) between
    :meth:`render` and :meth:`render_backward`, like ``Scene2D.store_backward`` does in the reference (dr.py:618-627).
    Every forward is stamped with a generation number (``self.generation``); :meth:`render_backward` reuses the state only
    when the caller's stamp is still the current one and recomputes it otherwise.

    Spill-pool overflow and invalid scene indices are detected WITHOUT synchronising: after a forward the 64-byte status block
    of the workspace is copied asynchronously to pinned memory (every ``poll_every`` forwards, and after each of the first
    two) and inspected at the next call.  An overflow found that way means that frames rendered since the poll were
    incomplete: the workspace is regrown and a RuntimeError says so.  ``check_overflow=True`` on a call checks synchronously
    (and regrows / repeats transparently)."""

    def __init__(self, nb_triangles, height, width, nb_colors, n_views=1, device="cuda", pool_pairs=0, poll_every=8):
        self.dims = (int(nb_triangles), int(height), int(width), int(nb_colors), int(n_views))
        self.device = _resolve_device(device)
        self.poll_every = int(poll_every)
        self.generation = 0
        self._alloc(pool_pairs)

    def _alloc(self, pool_pairs):
        self.pool_pairs = int(pool_pairs)
        nbytes = lib().deodr_hip_workspace_bytes(*self.dims, self.pool_pairs)
        if nbytes == 0:
            raise ValueError("invalid scene dimensions")
        with torch.cuda.device(self.device):
            self.workspace = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)  # must start zero-filled
        self.nbytes = nbytes
        self._status_words = self.workspace[:64].view(torch.int32)
        self._status_host = torch.zeros(16, dtype=torch.int32).pin_memory()
        self._status_event = None  # recorded after the last asynchronous copy of the status block
        self._forwards = 0
        self._checked = False
        self._pool_cap = None
        self._last = None
        self._loss_cache = None  # (the background-loss table belongs to one (observation, background, clamp) of one workspace)
        self.alloc_count = getattr(self, "alloc_count", 0) + 1  # a captured HIP graph holds the OLD workspace address: see GraphedStep

    @classmethod
    def for_scene(cls, ds, pool_pairs=0):
        return cls(ds.nb_triangles, ds.height, ds.width, ds.nb_colors, ds.n_views, ds.device, pool_pairs)

    # ---- status ------------------------------------------------------------------------------------------------------

    def _capacity(self, sc):
        if self._pool_cap is None:
            cap = C.c_ulonglong(0)
            _check(lib().deodr_hip_workspace_pool_pairs(C.byref(sc), self.nbytes, C.byref(cap)))
            self._pool_cap = int(cap.value)
        return self._pool_cap

    def _inspect_poll(self, sc):
        """Look at the last completed asynchronous copy of the status block (never waits)."""
        ev = self._status_event
        if ev is None or torch.cuda.is_current_stream_capturing() or not ev.query():
            return
        self._status_event = None
        needed, errors = int(self._status_host[_STATUS_NEEDED]) & 0xFFFFFFFF, int(self._status_host[_STATUS_ERRORS])
        if errors:
            raise RuntimeError("deodr_hip: " + scene_error_message(errors))
        if needed > self._capacity(sc):
            self._alloc(max(2 * needed, 1024))
            raise RuntimeError(
                f"deodr_hip: the spill pool of the workspace overflowed ({needed} pairs needed): frames rendered since the last "
                "check were incomplete; the workspace has been regrown, render again"
            )

    def _poll(self):
        """Queue an asynchronous copy of the status block behind the forward that was just launched."""
        self._forwards += 1
        if torch.cuda.is_current_stream_capturing():
            return
        if self._status_event is None and (self._forwards <= 2 or self._forwards % self.poll_every == 0):
            self._status_host.copy_(self._status_words, non_blocking=True)
            self._status_event = torch.cuda.Event()
            self._status_event.record()

    def poll_status(self):
        """For callers that launch this workspace's kernels without going through :meth:`render` & co (a captured HIP graph being
        replayed): look at the last asynchronous copy of the status block, then queue the next one.  Never waits; raises when a
        forward since the previous look overflowed the spill pool (the workspace is regrown: capture again) or met invalid indices."""
        if self._last is None:
            return
        self._inspect_poll(self._last[0].c_struct())
        self._polls = getattr(self, "_polls", 0) + 1
        if self._status_event is None and (self._polls - 1) % max(self.poll_every, 1) == 0:  # (a copy + event per replay is ~13 us of a ~200 us iteration)
            self._forwards = max(self._forwards, 2)
            self._status_host.copy_(self._status_words, non_blocking=True)
            self._status_event = torch.cuda.Event()
            self._status_event.record()

    def status(self, ds):
        """Synchronous check: -> (overflowed, needed_pairs, scene_error_bits)."""
        sc = ds.c_struct()
        over, need, errs = C.c_int(0), C.c_ulonglong(0), C.c_int(0)
        with torch.cuda.device(self.device):
            _check(lib().deodr_hip_workspace_status(C.byref(sc), _ptr(self.workspace), self.nbytes, _stream(self.device), C.byref(over),
                                                    C.byref(need), C.byref(errs)))  # fmt: skip
        return bool(over.value), int(need.value), int(errs.value)

    def _check_scene(self, ds):
        n, H, W, Cc = ds.n_views, ds.height, ds.width, ds.nb_colors
        if (ds.nb_triangles, H, W, Cc, n) != self.dims:
            raise ValueError("scene shape differs from the workspace shape")
        if ds.device != self.device:
            raise ValueError(f"scene lives on {ds.device}, the workspace on {self.device}")

    def _frame(self, ds, out):
        n, H, W, Cc = ds.n_views, ds.height, ds.width, ds.nb_colors
        pd = ds.pixel_dtype
        if out is None:
            return torch.empty((n, H, W, Cc), dtype=pd, device=ds.device), torch.empty((n, H, W), dtype=pd, device=ds.device)
        image, z = out
        for t, shape in ((image, (n, H, W, Cc)), (z, (n, H, W))):
            if t.device != ds.device or t.dtype != pd or tuple(t.shape) != shape or not t.is_contiguous():
                raise ValueError("out= buffers must be contiguous pixel-dtype tensors [n,H,W,C] / [n,H,W] on the scene's device")
        return image, z

    def _run_checked(self, ds, launch, check_overflow):
        """Launch a forward; with a synchronous check, regrow the workspace and repeat until nothing spills."""
        sync = check_overflow is True or (check_overflow is None and not self._checked)
        for _attempt in range(16):
            launch()
            if not sync:
                self._poll()
                return
            over, need, errs = self.status(ds)
            self._checked = True
            if errs:
                raise RuntimeError("deodr_hip: " + scene_error_message(errs))
            if not over:
                self._forwards += 1
                return
            self._alloc(max(2 * need, 1024))  # regrow (zero-filled) and render again
            self._checked = True
        # the pool doubles every time: this is not a scene that needs more room, something is wrong
        raise RuntimeError("deodr_hip: the spill pool still overflows after 16 regrows")

    # ---- calls -------------------------------------------------------------------------------------------------------

    def render(self, ds, sigma=1.0, antialiase_error=False, obs=None, out=None, check_overflow=None):
        """-> (image [n,H,W,C], z_buffer [n,H,W][, err_buffer [n,H,W]]) as pixel-dtype device tensors.

        ``check_overflow``: True = synchronise and make sure no tile list spilled past the pool (regrow + repeat if one did);
        None (default) = do that on the first call only, afterwards poll asynchronously; False = only poll."""
        self._check_scene(ds)
        n, H, W, Cc = ds.n_views, ds.height, ds.width, ds.nb_colors
        pd = ds.pixel_dtype
        with torch.cuda.device(self.device):
            sc = ds.c_struct()
            self._inspect_poll(sc)
            image, z = self._frame(ds, out)
            err = obs_t = None
            if antialiase_error:
                obs_t = _on(obs, ds.device, pd, (n, H, W, Cc), "obs")
                err = torch.empty((n, H, W), dtype=pd, device=ds.device)

            def launch():
                _check(lib().deodr_hip_render_scene(C.byref(sc), _ptr(image), _ptr(z), float(sigma), int(antialiase_error), _ptr(obs_t),
                                                    _ptr(err), _ptr(self.workspace), self.nbytes, _stream(self.device)))  # fmt: skip

            self._run_checked(ds, launch, check_overflow)
        self.generation += 1
        self._last = (ds, float(sigma), bool(antialiase_error), obs_t, image, err, self.generation, False)
        return (image, z, err) if antialiase_error else (image, z)

    def _loss_table(self, ds, sc, obs_t, options):
        """the background-loss table of (obs, background, clamp) for the loss of a fit step, computed once (deodr_hip_background_loss)"""
        key = (obs_t.data_ptr(), obs_t._version, tuple(obs_t.shape), None if ds.background_color is None else (ds.background_color.data_ptr(), ds.background_color._version),
               None if ds.background_image is None else (ds.background_image.data_ptr(), ds.background_image._version),
               (options.clamp, options.clamp_lo, options.clamp_hi))  # fmt: skip
        cache = getattr(self, "_loss_cache", None)
        if cache is None or cache[0] != key:
            L = lib()
            n = int(L.deodr_hip_fit_loss_bytes(ds.height, ds.width, ds.n_views)) // 8
            table, scratch = torch.empty(n, dtype=torch.float64, device=self.device), torch.empty(n, dtype=torch.float64, device=self.device)
            _check(L.deodr_hip_background_loss(C.byref(sc), _ptr(obs_t), C.byref(options), _ptr(table), _ptr(self.workspace), self.nbytes,
                                               _stream(self.device)))  # fmt: skip
            # (every tensor whose address is part of the key is kept alive by the cache: a new tensor cannot land on a keyed address)
            self._loss_cache = cache = (key, table, scratch, (obs_t, ds.background_color, ds.background_image))
        return cache[1], cache[2]

    def render_fit(self, ds, obs, sigma=1.0, grads=None, out=None, check_overflow=None, clear_grads=False, loss_out=None, clamp=None, done_flag=None):
        """One fit step in one call: render ``ds`` and back-propagate ``sum((image - obs)**2)``; -> (image, z_buffer, grads).

        Same results as :meth:`render` followed by ``render_backward(residual_obs=obs)`` (what the reference's
        ``Scene2D.render_compare_and_backward`` does with ``antialiase_error=False``), but the forward raster already
        back-propagates through every tile without silhouette edges, so the frame is traversed once.  ``clear_grads``: zero
        ``grads`` first, inside the same kernel launches (otherwise they are accumulated into).  ``loss_out``: a float64 device
        tensor of one element that receives ``sum((image - obs)**2)`` -- from the same launches, without a pass over the frame
        (``deodr_hip_render_scene_fit_ex``; the table it needs is computed at the first call with this observation).  ``clamp`` =
        (lo, hi): the loss is ``sum((image.clamp(lo, hi) - obs)**2)``, the depth fitter's data term (deodr/mesh_fitter.py:108-123);
        the returned image is the un-clamped rendering.  ``done_flag`` = (int32 / uint32 device tensor of one element, value): the step
        stores ``value`` there when its gradients are complete -- what a consumer on another stream waits for with :func:`wait_flag`
        instead of an event (``DeodrHipFitOptions::done_flag``)."""
        self._check_scene(ds)
        n, H, W, Cc = ds.n_views, ds.height, ds.width, ds.nb_colors
        pd = ds.pixel_dtype
        with torch.cuda.device(self.device):
            image, z = self._frame(ds, out)
            obs_t = obs if torch.is_tensor(obs) else torch.as_tensor(np.asarray(obs))
            obs_t = obs_t.to(device=ds.device, dtype=pd)
            if tuple(obs_t.shape) != (n, H, W, Cc) or not obs_t.is_contiguous():  # pass [n,H,W,C] to avoid this copy
                obs_t = obs_t.expand(n, H, W, Cc).contiguous()
            if check_overflow or (check_overflow is None and not self._checked):
                self.render(ds, sigma, out=(image, z), check_overflow=True)  # sizes the spill pool once (synchronises)
            if grads is None:
                grads = ds.zero_grads()
            sc = ds.c_struct(grads)
            self._inspect_poll(sc)
            if done_flag is not None and (done_flag[0].element_size() != 4 or done_flag[0].numel() != 1 or done_flag[0].device != ds.device):
                raise ValueError("done_flag must be (a 4-byte integer tensor of one element on the scene's device, value)")
            if loss_out is None and clamp is None and done_flag is None:
                _check(lib().deodr_hip_render_scene_fit(C.byref(sc), _ptr(image), _ptr(z), float(sigma), _ptr(obs_t), int(bool(clear_grads)),
                                                        _ptr(self.workspace), self.nbytes, _stream(self.device)))  # fmt: skip
            else:
                options = _FitOptionsC()
                if done_flag is not None:
                    options.done_flag, options.done_value = done_flag[0].data_ptr(), int(done_flag[1]) & 0xFFFFFFFF
                if clamp is not None:
                    options.clamp, options.clamp_lo, options.clamp_hi = 1, float(clamp[0]), float(clamp[1])
                if loss_out is not None:
                    if loss_out.dtype != torch.float64 or loss_out.device != ds.device or loss_out.numel() != 1:
                        raise ValueError("loss_out must be a float64 tensor of one element on the scene's device")
                    table, scratch = self._loss_table(ds, sc, obs_t, options)
                    options.tile_loss, options.loss, options.loss_scratch = table.data_ptr(), loss_out.data_ptr(), scratch.data_ptr()
                _check(lib().deodr_hip_render_scene_fit_ex(C.byref(sc), _ptr(image), _ptr(z), float(sigma), _ptr(obs_t), int(bool(clear_grads)),
                                                           C.byref(options), _ptr(self.workspace), self.nbytes, _stream(self.device)))  # fmt: skip
            self._poll()
        self.generation += 1
        self._last = (ds, float(sigma), False, obs_t, image, None, self.generation, True)
        return image, z, grads

    def render_backward(self, ds, image_b=None, err_buffer_b=None, grads=None, have_forward_state=True, residual_obs=None,
                        generation=None, sigma=None):  # fmt: skip
        """Adjoint of the last :meth:`render` of ``ds``; returns the dict of gradient tensors (accumulated into ``grads``
        when given, fresh zeros otherwise).  Nothing passed in is mutated.

        ``residual_obs`` (instead of ``image_b``): propagate the gradient of ``sum((image - residual_obs)**2)`` where
        ``image`` is the output of the last render; ``2 (image - obs)`` is formed inside the kernel.
        ``generation``: the value of ``self.generation`` right after the forward this adjoint belongs to; when another forward
        has run on the workspace since (two renders in one autograd graph), the forward state is recomputed from ``ds``
        instead of being trusted (pass that forward's ``sigma`` too)."""
        self._check_scene(ds)
        if self._last is None:
            raise RuntimeError("deodr_hip: render_backward called before any render on this workspace")
        last_ds, last_sigma, aa, obs_t, image, _err, gen, fused = self._last
        sigma = last_sigma if sigma is None else float(sigma)
        n, H, W, Cc = ds.n_views, ds.height, ds.width, ds.nb_colors
        pd = ds.pixel_dtype
        with torch.cuda.device(self.device):
            if grads is None:
                grads = ds.zero_grads()
            sc = ds.c_struct(grads)
            ib = eb = None
            if aa:
                eb = _on(err_buffer_b, ds.device, pd, (n, H, W), "err_buffer_b")
            elif residual_obs is not None:
                obs_t = residual_obs if torch.is_tensor(residual_obs) else torch.as_tensor(np.asarray(residual_obs))
                obs_t = obs_t.to(device=ds.device, dtype=pd)
                if tuple(obs_t.shape) != (n, H, W, Cc) or not obs_t.is_contiguous():  # pass [n,H,W,C] to avoid this copy
                    obs_t = obs_t.expand(n, H, W, Cc).contiguous()
            else:
                ib = _on(image_b, ds.device, pd, (n, H, W, Cc), "image_b")
            state = have_forward_state and last_ds is ds and not fused and (generation is None or generation == gen)
            if not state and not aa and residual_obs is not None:
                # residual mode forms 2 (image - obs) inside the kernels from the frame of THIS forward; the frame at hand belongs to
                # another one (a later forward used the workspace, or the last call was a fit step): render again, then the state --
                # and the frame -- are this scene's
                image, _z = self.render(ds, sigma, check_overflow=False)
                state = True
            _check(lib().deodr_hip_render_scene_b(C.byref(sc), _ptr(image), None, _ptr(ib), sigma, int(aa), _ptr(obs_t), None, _ptr(eb),
                                                  _ptr(self.workspace), self.nbytes, int(state), _stream(self.device)))  # fmt: skip
            if not state:
                # a forward ran inside the call: the workspace now holds the state of THIS (ds, sigma), stamped anew so that
                # any other pending adjoint sees that its own forward state is gone
                self._poll()
                self.generation += 1
                self._last = (ds, sigma, aa, obs_t, image, _err, self.generation, False)
        return grads


# ---------------------------------------------------------------------------------------------------------------------
# drop-in NumPy entry points (reference pyx:50-57, 206-215)


def _np(a, dtype=None):
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=dtype)


_ctx_cache = {}


def _device_scene(scene, nb_colors):
    tex = _np(scene.texture, np.float64)
    bgi = getattr(scene, "background_image", None)
    bgc = getattr(scene, "background_color", None)
    return DeviceScene(
        faces=_np(scene.faces), faces_uv=_np(scene.faces_uv), textured=_np(scene.textured, np.uint8), shaded=_np(scene.shaded, np.uint8),
        uv=_np(scene.uv, np.float64), ij=_np(scene.ij, np.float64)[None], depths=_np(scene.depths, np.float64)[None],
        colors=_np(scene.colors, np.float64).reshape(1, -1, nb_colors), shade=_np(scene.shade, np.float64)[None],
        edgeflags=_np(scene.edgeflags, np.uint8)[None], height=scene.height, width=scene.width,
        texture=tex if tex.size else None, background_color=None if bgc is None else _np(bgc, np.float64),
        background_image=None if bgi is None else _np(bgi, np.float64)[None], clockwise=scene.clockwise,
        backface_culling=scene.backface_culling, strict_edge=scene.strict_edge, perspective_correct=scene.perspective_correct,
        integer_pixel_centers=scene.integer_pixel_centers, vertex_dtype=torch.float64, pixel_dtype=torch.float64,
    )  # fmt: skip


def _rasterizer_for(ds):
    key = (ds.nb_triangles, ds.height, ds.width, ds.nb_colors, str(ds.device))
    if key not in _ctx_cache:
        if len(_ctx_cache) > 8:
            _ctx_cache.clear()
        _ctx_cache[key] = HipRasterizer.for_scene(ds)
    return _ctx_cache[key]


def renderSceneCpp(scene, sigma, image, z_buffer, antialiase_error=False, obs=None, err_buffer=None, check_valid=True):
    """Same contract as the reference's Cython ``renderSceneCpp``: fills ``image`` / ``z_buffer`` (/ ``err_buffer``) in place."""
    if check_valid:
        from .differentiable_renderer import check_scene

        check_scene(scene, image, z_buffer, False, None, antialiase_error, obs, err_buffer)
    ds = _device_scene(scene, image.shape[2])
    r = _rasterizer_for(ds)
    out = r.render(ds, sigma, bool(antialiase_error), None if obs is None else torch.as_tensor(_np(obs, np.float64)), check_overflow=True)
    image[...] = out[0][0].cpu().numpy()
    z_buffer[...] = out[1][0].cpu().numpy()
    if antialiase_error:
        err_buffer[...] = out[2][0].cpu().numpy()


def renderSceneBCpp(scene, sigma, image, z_buffer, image_b=None, antialiase_error=False, obs=None, err_buffer=None, err_buffer_b=None,
                    check_valid=True):  # fmt: skip
    """Same contract as the reference's Cython ``renderSceneBCpp``: ``scene.{uv,ij,shade,colors,texture}_b`` are rebound to
    ``old + new`` (pyx:406-410).  Stateless like the reference: the forward state is recomputed on the device.  The
    reference's in-place side effects on ``image`` / ``image_b`` / ``err_buffer`` are NOT reproduced."""
    if check_valid:
        from .differentiable_renderer import check_scene

        check_scene(scene, image, z_buffer, True, image_b, antialiase_error, obs, err_buffer)
    if scene.perspective_correct:
        raise RuntimeError("backward gradient propagation not supported yet with perspective_correct=True")
    nb_colors = image.shape[2]
    ds = _device_scene(scene, nb_colors)
    r = _rasterizer_for(ds)
    dev = ds.device
    img_t = torch.as_tensor(_np(image, np.float64)).to(dev)[None]
    obs_t = None if obs is None else torch.as_tensor(_np(obs, np.float64)).to(dev)[None].contiguous()
    for _attempt in range(16):
        r.generation += 1
        r._last = (ds, float(sigma), bool(antialiase_error), obs_t, img_t, None, r.generation, False)
        if antialiase_error:
            g = r.render_backward(ds, err_buffer_b=torch.as_tensor(_np(err_buffer_b, np.float64)), have_forward_state=False)
        else:
            g = r.render_backward(ds, image_b=torch.as_tensor(_np(image_b, np.float64)), have_forward_state=False)
        over, need, errs = r.status(ds)  # the call is synchronous anyway: the stateless forward must not have spilled
        if errs:
            raise RuntimeError("deodr_hip: " + scene_error_message(errs))
        if not over:
            break
        r._alloc(max(2 * need, 1024))
    else:
        raise RuntimeError("deodr_hip: the spill pool still overflows after 16 regrows")
    for name in ("uv_b", "ij_b", "shade_b", "colors_b", "texture_b"):
        new = g[name]
        old = getattr(scene, name, None)
        if new is None or old is None or _count(old) == 0:
            continue
        setattr(scene, name, _np(old, np.float64) + new.cpu().numpy().reshape(np.shape(old)))
