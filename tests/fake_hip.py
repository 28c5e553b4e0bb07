"""TEST harness: the Python host layer of deodr_amd (hip_renderer.py, the torch operator, the NumPy drop-ins, Scene3DDevice) on CPU
tensors, with a restatement of libdeodr_hip.so's C ABI over the CPU checker (oracle/) in place of the library.

``emulate()`` swaps ``hip_renderer.lib`` for :class:`FakeLib` -- same entry points, same struct, same accumulate-into / clear
semantics, rendering and back-propagating with the checker view by view -- and stubs the handful of torch.cuda calls the host layer
makes (device context, stream handle, events, pinned memory).  What runs under it is the HOST LOGIC: marshalling, caching,
generation stamps, gradient rebinding, exceptions.  The numbers come from the checker, so nothing here says anything about the
kernels (that is what the `-m gpu` tests are for); nothing of this is reachable from the product."""

import contextlib
import ctypes as C
import types

import numpy as np
import torch

from deodr_amd import Scene2D, hip_renderer


def _view(ptr, shape, dtype):
    """writable NumPy view of `shape` / `dtype` at a raw address (None for NULL)"""
    if ptr is None:
        return None
    addr = ptr.value if isinstance(ptr, C.c_void_p) else int(ptr)
    if not addr:
        return None
    count = int(np.prod(shape))
    if count == 0:
        return np.zeros(shape, dtype=dtype)
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class FakeLib:
    def __init__(self, checker, checker_repaired):
        self.checker, self.repaired = checker, checker_repaired
        self.error = b""
        self.calls = dict(render_scene=0, render_scene_b=0, render_scene_fit=0)
        self.generic = 0

    # ---- the small entry points ------------------------------------------------------------------------------------------------
    def deodr_hip_abi_version(self):
        return hip_renderer.ABI_VERSION

    def deodr_hip_last_error(self):
        return self.error

    def deodr_hip_workspace_bytes(self, T, H, W, Cc, n_views, pool_pairs):
        return 0 if min(H, W, Cc, n_views) <= 0 or T < 0 else 4096

    def deodr_hip_workspace_pool_pairs(self, sc, nbytes, cap):
        cap._obj.value = 1 << 40
        return 0

    def deodr_hip_workspace_status(self, sc, ws, nbytes, stream, over, need, errs):
        over._obj.value, need._obj.value, errs._obj.value = 0, 0, 0
        return 0

    def deodr_hip_workspace_census(self, sc, ws, nbytes, stream, a, b):
        a._obj.value, b._obj.value = 0, 0
        return 0

    def deodr_hip_set_deterministic(self, on):
        return 0  # (the checker is single-threaded: always deterministic)

    def deodr_hip_force_generic(self, on):
        self.generic = int(on)

    def deodr_hip_profile_enable(self, every):
        return 0

    # ---- scene marshalling -----------------------------------------------------------------------------------------------------
    def _scene(self, sc_ref):
        sc = sc_ref._obj
        vd = np.float64 if sc.vertex_dtype == 1 else np.float32
        pd = np.float64 if sc.pixel_dtype == 1 else np.float32
        n, T, V, Vuv, Cc, H, W = sc.n_views, sc.nb_triangles, sc.nb_vertices, sc.nb_uv, sc.nb_colors, sc.height, sc.width
        tex_shape = (sc.texture_height, sc.texture_width, Cc)
        a = dict(
            faces=_view(sc.faces, (T, 3), np.int32), faces_uv=_view(sc.faces_uv, (T, 3), np.int32), textured=_view(sc.textured, (T,), np.uint8),
            shaded=_view(sc.shaded, (T,), np.uint8), depths=_view(sc.depths, (n, V), vd), ij=_view(sc.ij, (n, V, 2), vd), shade=_view(sc.shade, (n, V), vd),
            colors=_view(sc.colors, (n, V, Cc), vd), edgeflags=_view(sc.edgeflags, (n, T, 3), np.uint8), uv=_view(sc.uv, (Vuv, 2), vd),
            texture=_view(sc.texture, tex_shape, pd), background_image=_view(sc.background_image, (n, H, W, Cc), pd),
            background_color=_view(sc.background_color, (Cc,), pd),
            uv_b=_view(sc.uv_b, (Vuv, 2), vd), ij_b=_view(sc.ij_b, (n, V, 2), vd), shade_b=_view(sc.shade_b, (n, V), vd),
            colors_b=_view(sc.colors_b, (n, V, Cc), vd), texture_b=_view(sc.texture_b, tex_shape, pd),
        )  # fmt: skip
        return sc, a, pd

    def _view_scene(self, sc, a, i):
        f64 = lambda x: None if x is None else np.ascontiguousarray(x, dtype=np.float64)
        return Scene2D(
            faces=a["faces"].astype(np.uint32), faces_uv=a["faces_uv"].astype(np.uint32), ij=f64(a["ij"][i]), depths=f64(a["depths"][i]),
            textured=a["textured"].astype(bool), uv=f64(a["uv"]), shade=f64(a["shade"][i]), colors=f64(a["colors"][i]), shaded=a["shaded"].astype(bool),
            edgeflags=a["edgeflags"][i].astype(bool), height=sc.height, width=sc.width, nb_colors=sc.nb_colors,
            texture=np.zeros((0, 0)) if a["texture"] is None else f64(a["texture"]),
            background_image=None if a["background_image"] is None else f64(a["background_image"][i]), background_color=f64(a["background_color"]),
            clockwise=bool(sc.clockwise), backface_culling=bool(sc.backface_culling), strict_edge=bool(sc.strict_edge),
            perspective_correct=bool(sc.perspective_correct), integer_pixel_centers=bool(sc.integer_pixel_centers),
        )  # fmt: skip

    def _fail(self, message):
        self.error = message.encode()
        return 1

    def _check(self, sc, a, backward):
        if sc.nb_triangles and (a["faces"].min() < 0 or a["faces"].max() >= sc.nb_vertices or a["faces_uv"].min() < 0 or a["faces_uv"].max() >= sc.nb_uv):
            return self._fail("invalid scene indices")
        if (a["background_image"] is None) == (a["background_color"] is None):
            return self._fail("exactly one of scene.background_image / scene.background_color must be given")
        if backward and not sc.backface_culling:
            return self._fail("You have to use backface_culling true if you ant to compute gradients")
        if backward and sc.perspective_correct:
            return self._fail("backward gradient propagation not supported yet with perspective_correct=True")
        return 0

    # ---- the three calls of the path --------------------------------------------------------------------------------------------
    def deodr_hip_render_scene(self, sc_ref, image, z_buffer, sigma, antialiase_error, obs, err_buffer, ws, nbytes, stream):
        self.calls["render_scene"] += 1
        sc, a, pd = self._scene(sc_ref)
        if self._check(sc, a, False):
            return 1
        n, H, W, Cc = sc.n_views, sc.height, sc.width, sc.nb_colors
        image, z_buffer = _view(image, (n, H, W, Cc), pd), _view(z_buffer, (n, H, W), pd)
        obs, err_buffer = _view(obs, (n, H, W, Cc), pd), _view(err_buffer, (n, H, W), pd)
        for i in range(n):
            out = self.checker.render(self._view_scene(sc, a, i), sigma, bool(antialiase_error), None if obs is None else obs[i].astype(np.float64))
            image[i], z_buffer[i] = out[0], out[1]
            if antialiase_error:
                err_buffer[i] = out[2]
        return 0

    def _adjoint(self, sc, a, pd, sigma, antialiase_error, image_b, obs, err_buffer_b, image=None):
        n = sc.n_views
        for i in range(n):
            s = self._view_scene(sc, a, i)
            ob = None if obs is None else obs[i].astype(np.float64)
            out = self.checker.render(s, sigma, bool(antialiase_error), ob if antialiase_error else None)
            if antialiase_error:
                g = self.repaired.grads(s, sigma, out[0], out[1], None, True, ob, out[2], err_buffer_b[i].astype(np.float64))
            else:
                # residual mode: 2 (image - obs) from the frame the CALLER hands over (the library reads that pointer), as the real ABI
                frame = out[0] if image is None else image[i].astype(np.float64)
                seed = 2 * (frame - ob) if image_b is None else image_b[i].astype(np.float64)
                g = self.repaired.grads(s, sigma, out[0], out[1], seed)
            # (where the reference divides by an edge transparency of 0 its gradient is NaN; the library returns finite numbers there)
            g = {k: np.nan_to_num(v) for k, v in g.items()}
            a["ij_b"][i] += g["ij_b"]
            a["colors_b"][i] += g["colors_b"]
            a["shade_b"][i] += g["shade_b"]
            a["uv_b"] += g["uv_b"]
            if a["texture_b"] is not None:
                a["texture_b"] += g["texture_b"]

    def deodr_hip_render_scene_b(self, sc_ref, image, z_buffer, image_b, sigma, antialiase_error, obs, err_buffer, err_buffer_b, ws, nbytes,
                                 have_forward_state, stream):  # fmt: skip
        self.calls["render_scene_b"] += 1
        sc, a, pd = self._scene(sc_ref)
        if self._check(sc, a, True):
            return 1
        if any(a[k] is None for k in ("uv_b", "ij_b", "shade_b", "colors_b")):
            return self._fail("scene gradient array == NULL")
        n, H, W, Cc = sc.n_views, sc.height, sc.width, sc.nb_colors
        image_b, obs, err_buffer_b = _view(image_b, (n, H, W, Cc), pd), _view(obs, (n, H, W, Cc), pd), _view(err_buffer_b, (n, H, W), pd)
        if antialiase_error and (obs is None or err_buffer_b is None):
            return self._fail("antialiase_error needs obs and err_buffer_b")
        if not antialiase_error and image_b is None and obs is None:
            return self._fail("image_b == NULL (or, for the residual mode, image and obs)")
        self._adjoint(sc, a, pd, sigma, antialiase_error, image_b, obs, err_buffer_b, _view(image, (n, H, W, Cc), pd))
        return 0

    def deodr_hip_render_scene_fit(self, sc_ref, image, z_buffer, sigma, obs, clear_gradients, ws, nbytes, stream):
        self.calls["render_scene_fit"] += 1
        sc, a, pd = self._scene(sc_ref)
        if self._check(sc, a, True):
            return 1
        if clear_gradients:
            for k in ("uv_b", "ij_b", "shade_b", "colors_b", "texture_b"):
                if a[k] is not None:
                    a[k][...] = 0
        rc = self.deodr_hip_render_scene(sc_ref, image, z_buffer, sigma, 0, None, None, ws, nbytes, stream)
        self.calls["render_scene"] -= 1
        if rc:
            return rc
        n, H, W, Cc = sc.n_views, sc.height, sc.width, sc.nb_colors
        self._adjoint(sc, a, pd, sigma, 0, None, _view(obs, (n, H, W, Cc), pd), None)
        return 0


    # ---- the loss of a fit step: same table semantics as the library (tests/test_hip_round3.py checks the kernels themselves)
    @staticmethod
    def _tiles(H, W):
        return (W + 7) // 8, (H + 7) // 8

    def deodr_hip_fit_loss_bytes(self, H, W, n_views):
        tx, ty = self._tiles(H, W)
        return 8 * max(1 + n_views * tx * ty, 256 + 16, n_views * 256)

    @staticmethod
    def _tile_sums(per_pixel, tx, ty):
        """[n,H,W] -> [n, ty * tx]: sums over the 8 x 8 tiles (ragged last row / column)"""
        n, H, W = per_pixel.shape
        padded = np.zeros((n, ty * 8, tx * 8))
        padded[:, :H, :W] = per_pixel
        return padded.reshape(n, ty, 8, tx, 8).sum(axis=(2, 4)).reshape(n, ty * tx)

    @staticmethod
    def _clamped(values, options):
        o = None if options is None else options._obj
        return np.clip(values, o.clamp_lo, o.clamp_hi) if o is not None and o.clamp else values

    def deodr_hip_background_loss(self, sc_ref, obs, options, table, ws, nbytes, stream):
        sc, a, pd = self._scene(sc_ref)
        n, H, W, Cc = sc.n_views, sc.height, sc.width, sc.nb_colors
        tx, ty = self._tiles(H, W)
        background = a["background_image"] if a["background_image"] is not None else np.broadcast_to(a["background_color"], (n, H, W, Cc))
        per_pixel = ((self._clamped(background.astype(np.float64), options) - _view(obs, (n, H, W, Cc), pd).astype(np.float64)) ** 2).sum(axis=-1)
        out = _view(table, (1 + n * tx * ty,), np.float64)
        out[1:] = self._tile_sums(per_pixel, tx, ty).reshape(-1)
        out[0] = out[1:].sum()
        return 0

    def deodr_hip_render_scene_fit_ex(self, sc_ref, image, z_buffer, sigma, obs, clear_gradients, options, ws, nbytes, stream):
        o = options._obj
        sc, a, pd = self._scene(sc_ref)
        n, H, W, Cc = sc.n_views, sc.height, sc.width, sc.nb_colors
        if not o.clamp:
            rc = self.deodr_hip_render_scene_fit(sc_ref, image, z_buffer, sigma, obs, clear_gradients, ws, nbytes, stream)
        else:  # the adjoint of sum (clamp(image) - obs)^2: image_b = 2 (clamp(image) - obs) where the clamp passes
            if self._check(sc, a, True):
                return 1
            if clear_gradients:
                for k in ("uv_b", "ij_b", "shade_b", "colors_b", "texture_b"):
                    if a[k] is not None:
                        a[k][...] = 0
            rc = self.deodr_hip_render_scene(sc_ref, image, z_buffer, sigma, 0, None, None, ws, nbytes, stream)
            self.calls["render_scene"] -= 1
            self.calls["render_scene_fit"] += 1
            if not rc:
                img, ob = _view(image, (n, H, W, Cc), pd).astype(np.float64), _view(obs, (n, H, W, Cc), pd).astype(np.float64)
                image_b = np.where((img >= o.clamp_lo) & (img <= o.clamp_hi), 2 * (img - ob), 0.0).astype(pd)
                self._adjoint(sc, a, pd, sigma, 0, image_b, None, None)
        if not rc and o.done_flag:  # (the step-done flag: everything is synchronous here)
            _view(o.done_flag, (1,), np.uint32)[0] = o.done_value
        if rc or not o.loss:
            return rc
        tx, ty = self._tiles(H, W)
        table = _view(o.tile_loss, (1 + n * tx * ty,), np.float64)
        rendered = ((self._clamped(_view(image, (n, H, W, Cc), pd).astype(np.float64), options) - _view(obs, (n, H, W, Cc), pd).astype(np.float64)) ** 2).sum(axis=-1)
        tiles = self._tile_sums(rendered, tx, ty).reshape(-1)
        changed = tiles != table[1:]  # (the library only visits the tiles that hold primitives: the others ARE their background)
        _view(o.loss, (1,), np.float64)[0] = table[0] + (tiles[changed] - table[1:][changed]).sum()
        return 0


class _Event:
    def record(self, *a):
        pass

    def query(self):
        return True

    def synchronize(self):
        pass

    def wait(self, *a):
        pass


@contextlib.contextmanager
def emulate(checker, checker_repaired):
    """-> the FakeLib in use.  Within the block every scene / workspace "device" is the CPU."""
    from deodr_amd import scene3d_compat

    fake = FakeLib(checker, checker_repaired)
    cpu = torch.device("cpu")
    patches = [
        (hip_renderer, "lib", lambda: fake), (hip_renderer, "_resolve_device", lambda device: cpu), (scene3d_compat, "_device", lambda: cpu),
        (torch.cuda, "device", lambda device: contextlib.nullcontext()), (torch.cuda, "current_stream", lambda device=None: types.SimpleNamespace(cuda_stream=0)),
        (torch.cuda, "Event", _Event), (torch.cuda, "is_current_stream_capturing", lambda: False), (torch.cuda, "synchronize", lambda *a: None),
        (torch.cuda, "current_device", lambda: 0), (torch.Tensor, "pin_memory", lambda self: self),
    ]  # fmt: skip
    import deodr_amd.pytorch.differentiable_renderer_pytorch as wrapper

    patches.append((wrapper, "_resolve_device", lambda device: cpu))
    saved = [(obj, name, getattr(obj, name)) for obj, name, _ in patches]
    saved_cache = dict(hip_renderer._ctx_cache)
    hip_renderer._ctx_cache.clear()
    for obj, name, new in patches:
        setattr(obj, name, new)
    try:
        yield fake
    finally:
        for obj, name, old in saved:
            setattr(obj, name, old)
        hip_renderer._ctx_cache.clear()
        hip_renderer._ctx_cache.update(saved_cache)
