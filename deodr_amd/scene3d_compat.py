"""NumPy-level ``Camera`` / ``Scene3D`` with the interface of ``deodr.differentiable_renderer`` (dr.py:250-522, 735-1174).

These are adapters, not a port: each call moves its NumPy inputs to the device, runs the batched device pipeline of
:mod:`deodr_amd.scene3d` (projection, distortion, lighting, silhouette flags, HIP rasterizer) under autograd, and hands NumPy
back.  The ``*_backward`` methods of the reference -- hand-written adjoints chained through ``store_backward`` dictionaries --
become one ``torch.autograd`` call on the graph the forward left behind; the result attributes keep the reference's names
(``mesh._vertices_b``, ``mesh.vertices_colors_b``, ``light_directional_b``, ``light_ambient_b``, ``scene_2d.ij_b`` ...), so
fitter code written against DEODR's ``Scene3D`` (deodr/mesh_fitter.py) runs unchanged on top.

For fit loops that should never leave the device use :class:`deodr_amd.scene3d.Scene3DDevice` and the fitters of
:mod:`deodr_amd.mesh_fitter` directly.
"""

from types import SimpleNamespace

import numpy as np
import torch

from .scene3d import DeviceCamera, DeviceMesh, Scene3DDevice


def _device():
    return torch.device("cuda", torch.cuda.current_device())


class Camera:
    """Pinhole camera with OpenCV's distortion parameters (dr.py:250-438)."""

    def __init__(self, extrinsic, intrinsic, height, width, distortion=None, checks=True, tol=1e-6):
        if checks:
            assert extrinsic.shape == (3, 4) and intrinsic.shape == (3, 3)
            assert np.all(intrinsic[2, :] == [0, 0, 1])
            r = extrinsic[:3, :3]
            assert np.linalg.norm(r.T.dot(r) - np.eye(3)) < tol
            if distortion is not None:
                distortion = np.array(distortion)
                assert distortion.shape == (5,)
        self.extrinsic, self.intrinsic, self.distortion = extrinsic, intrinsic, distortion
        self.height, self.width = height, width

    def on_device(self):
        return DeviceCamera(np.asarray(self.extrinsic, dtype=np.float64), np.asarray(self.intrinsic, dtype=np.float64), self.height, self.width,
                            None if self.distortion is None else np.asarray(self.distortion, dtype=np.float64), _device())  # fmt: skip

    def world_to_camera(self, points_3d):
        return points_3d.dot(self.extrinsic[:3, :3].T) + self.extrinsic[:3, 3]

    def get_center(self):
        return -self.extrinsic[:3, :3].T.dot(self.extrinsic[:, 3])

    # small host-side helpers of the reference class (dr.py:280-310, 443-451): exporters and viewers call them
    def _fov(self, axis, size):
        assert self.intrinsic[axis, 2] == size / 2, "the principal point must be the image centre"
        return float(np.degrees(2 * np.arctan(size / (2 * self.intrinsic[axis, axis]))))

    xfov = property(lambda self: self._fov(0, self.width), doc="horizontal field of view in degrees")
    yfov = property(lambda self: self._fov(1, self.height), doc="vertical field of view in degrees")

    def camera_to_world_mtx_4x4(self):
        m = np.eye(4)
        m[:3, :3], m[:3, 3] = self.extrinsic[:, :3].T, self.get_center()
        return m

    def left_mul_intrinsic(self, projected):
        assert projected.ndim == 2 and projected.shape[-1] == 2
        return projected.dot(self.intrinsic[:2, :2].T) + self.intrinsic[:2, 2]

    def column_stack(self, values):
        return np.column_stack(values)

    def __repr__(self):
        fields = ("width", "height", "extrinsic", "intrinsic", "distortion")
        return "<Camera>\n" + "".join(f"{name}:\n{getattr(self, name)}\n" for name in fields)

    def project_points(self, points_3d, return_depths=True, store_backward=None):
        pts = torch.as_tensor(np.asarray(points_3d, dtype=np.float64), device=_device()).requires_grad_(store_backward is not None)
        ij, depths = self.on_device().project_points(pts)
        if store_backward is not None:
            store_backward["project_points"] = (pts, ij, depths)  # the autograd graph IS the stored state
        ij_np, d_np = ij[0].detach().cpu().numpy(), depths[0].detach().cpu().numpy()
        return (ij_np, d_np) if return_depths else ij_np

    def project_points_backward(self, projected_image_coordinates_b, store_backward, depths_b=None):
        pts, ij, depths = store_backward["project_points"]
        outs, grads = [ij], [torch.as_tensor(np.asarray(projected_image_coordinates_b, dtype=np.float64), device=ij.device)[None]]
        if depths_b is not None:
            outs.append(depths)
            grads.append(torch.as_tensor(np.asarray(depths_b, dtype=np.float64), device=ij.device)[None])
        (g,) = torch.autograd.grad(outs, [pts], grads, retain_graph=True)
        return g.cpu().numpy()


class PerspectiveCamera(Camera):
    """width x height pixels, horizontal field of view ``fov`` in degrees, x_cam = rot (x_world - camera_center) (dr.py:441-499)"""

    def __init__(self, width, height, fov, camera_center, rot=None, distortion=None):
        rot = np.eye(3) if rot is None else rot
        assert camera_center.shape == (3,) and rot.shape == (3, 3)
        assert np.allclose(rot.T.dot(rot), np.eye(3), 1e-6) and np.linalg.det(rot) > 0
        focal = 0.5 * width / np.tan(0.5 * np.deg2rad(fov))
        intrinsic = np.array([[focal, 0, width / 2], [0, focal, height / 2], [0, 0, 1]])
        extrinsic = np.column_stack((rot, -rot.T.dot(camera_center)))
        super().__init__(extrinsic=extrinsic, intrinsic=intrinsic, distortion=distortion, width=width, height=height)


def default_camera(width, height, fov, vertices, rot, distortion=None):
    """camera far enough on -z (in the frame ``rot``) for the whole mesh to be in view (dr.py:502-522)"""
    cam = vertices.dot(rot.T)
    lo, hi = cam.min(axis=0), cam.max(axis=0)
    size, t = hi - lo, np.tan(0.5 * np.deg2rad(fov))
    distance = max(0.5 * size[0] / t, 0.5 * size[1] * (width / height) / t) + 0.5 * size[2]
    return PerspectiveCamera(width, height, fov, rot.T.dot(0.5 * (lo + hi) + np.array([0, 0, -distance])), rot, distortion)


class Scene3D:
    """One mesh + a directional and an ambient light, rendered through the device pipeline (dr.py:735-1174).

    Same methods and result attributes as the reference; ``scene_2d`` exposes the 2.5-D arrays (and, after a backward, their
    adjoints) of the last render as NumPy for code that inspects them.
    Caching: the device twin of the mesh (connectivity analysis, uploaded texture and uv) is kept between renders and rebuilt when
    ``mesh.faces`` / ``mesh.uv`` / ``mesh.texture`` (or the background image) are REPLACED by other objects.  Editing one of those
    arrays in place is not seen -- rebind it (``mesh.texture = mesh.texture - step``, as the reference's fitters do) or call
    ``scene.set_mesh(mesh)`` again; vertices, colours and lights are read at every render."""

    def __init__(self, sigma=1, perspective_correct=False, integer_pixel_centers=True):
        self.mesh = None
        self.light_directional, self.light_ambient = None, 0
        self.sigma, self.perspective_correct, self.integer_pixel_centers = sigma, perspective_correct, integer_pixel_centers
        self.background_image, self.background_color = None, None
        self.scene_2d = None
        self._dev = Scene3DDevice(sigma, perspective_correct, integer_pixel_centers)
        self._graph = None
        self._dmesh_key, self._dmesh = None, None
        self.light_directional_b, self.light_ambient_b, self.vertex_normals_b = None, None, None

    # ---- configuration ---------------------------------------------------------------------------------------------------

    def set_light(self, light_directional, light_ambient):
        self.light_directional = None if light_directional is None else np.array(light_directional)
        self.light_ambient = light_ambient

    def set_mesh(self, mesh):
        self.mesh = mesh
        self._dmesh_key, self._dmesh = None, None  # (the device twin is rebuilt at the next render: see "Caching" above)

    def set_background_image(self, background_image):
        if self.background_color is not None:
            raise BaseException("you cannot provide both background image and background color")
        background_image = np.asanyarray(background_image)
        assert background_image.dtype == np.double and background_image.ndim == 3
        self.background_image = background_image

    set_background = set_background_image

    def set_background_color(self, background_color):
        if self.background_image is not None:
            raise BaseException("you cannot provide both background image and background color")
        background_color = np.asanyarray(background_color, dtype=np.float64)
        assert background_color.ndim == 1
        self.background_color = background_color

    def clear_gradients(self):
        assert self.mesh is not None
        self.mesh._vertices_b = np.zeros((self.mesh.nb_vertices, 3))

    # ---- lighting on its own (dr.py:814-850) ------------------------------------------------------------------------------

    def compute_vertices_luminosity(self):
        """max(0, -n . l) + ambient per vertex from ``mesh.vertex_normals``, through the device op (its graph is kept for
        :meth:`compute_vertices_luminosity_backward`)"""
        assert self.mesh is not None
        dev = _device()
        normals = torch.as_tensor(np.asarray(self.mesh.vertex_normals, dtype=np.float64), device=dev).requires_grad_(True)
        ambient = torch.as_tensor(float(self.light_ambient), dtype=torch.float64, device=dev).requires_grad_(True)
        light = None
        if self.light_directional is None:
            out = torch.zeros(normals.shape[0], dtype=torch.float64, device=dev) + ambient
        else:
            light = torch.as_tensor(np.asarray(self.light_directional, dtype=np.float64), device=dev).requires_grad_(True)
            out = torch.relu(-(normals * light).sum(-1)) + ambient
        self._luminosity_graph = (out, normals, light, ambient)
        return out.detach().cpu().numpy()

    def compute_vertices_luminosity_backward(self, vertices_luminosity_b):
        """-> ``light_directional_b``, ``vertex_normals_b``, ``light_ambient_b`` (same attributes as the reference)"""
        out, normals, light, ambient = self._luminosity_graph
        seed = torch.as_tensor(np.asarray(vertices_luminosity_b, dtype=np.float64), device=out.device)
        leaves = [ambient] + ([normals, light] if light is not None else [])
        grads = torch.autograd.grad([out], leaves, [seed], retain_graph=True)
        self.light_ambient_b = float(grads[0])
        if light is not None:
            self.vertex_normals_b, self.light_directional_b = grads[1].cpu().numpy(), grads[2].cpu().numpy()

    # ---- forward ---------------------------------------------------------------------------------------------------------

    def _device_mesh(self):
        m = self.mesh
        key = (id(m.faces), m.clockwise, id(m.uv), id(m.faces_uv), id(m.texture))
        if self._dmesh_key != key:
            self._dmesh = DeviceMesh(m.faces, np.asarray(m.vertices, dtype=np.float64), m.clockwise, uv=m.uv, faces_uv=m.faces_uv, texture=m.texture,
                                     device=_device())  # fmt: skip
            self._dmesh_key = key
        return self._dmesh

    def _prepare(self, with_lights):
        assert self.mesh is not None, "You need to provide a mesh first."
        dm = self._device_mesh()
        dev = dm.device
        leaf = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev).requires_grad_(True)
        leaves = {"vertices": leaf(self.mesh.vertices)}
        dm.set_vertices(leaves["vertices"])
        d = self._dev
        d.sigma, d.perspective_correct, d.integer_pixel_centers = float(self.sigma), self.perspective_correct, self.integer_pixel_centers
        d.background_color, d.background_image = self.background_color, self.background_image
        d.set_mesh(dm)
        if with_lights:
            if self.mesh.uv is None:
                leaves["colors"] = leaf(self.mesh.vertices_colors)
                dm.set_vertices_colors(leaves["colors"])
            leaves["ambient"] = leaf(self.light_ambient)
            if self.light_directional is not None:
                leaves["directional"] = leaf(self.light_directional)
            d.light_directional, d.light_ambient = leaves.get("directional"), leaves["ambient"]
        return leaves

    def _publish(self, camera, nb_colors):
        last = self._dev.last
        np_ = lambda t: t[0].detach().cpu().numpy()
        self.scene_2d = SimpleNamespace(ij=np_(last["ij"]), depths=np_(last["depths"]), edgeflags=np_(last["edgeflags"]).astype(bool),
                                        colors=np_(last["colors"]), shade=np_(last["shade"]), height=camera.height, width=camera.width,
                                        nb_colors=nb_colors, ij_b=None, colors_b=None, shade_b=None)  # fmt: skip

    def render(self, camera, return_z_buffer=False, backface_culling=True):
        if (self.background_image is None) == (self.background_color is None):
            raise BaseException("You need to provide either a background image or background color")
        leaves = self._prepare(True)
        image, z = self._dev.render(camera.on_device(), True, backface_culling)
        self._graph = ("render", leaves, image, None)
        self._publish(camera, int(image.shape[-1]))
        image_np = image[0].detach().cpu().numpy()
        return (image_np, z[0].cpu().numpy()) if return_z_buffer else image_np

    def render_depth(self, camera, depth_scale=1, backface_culling=True):
        leaves = self._prepare(False)
        image = self._dev.render_depth(camera.on_device(), depth_scale, backface_culling)
        self._graph = ("render_depth", leaves, image, depth_scale)
        self._publish(camera, 1)
        return image[0].detach().cpu().numpy()

    # ---- adjoints: one autograd call on the stored graph -----------------------------------------------------------------

    def _backward(self, kind, image_b):
        if self.perspective_correct:
            raise BaseException("perspective_correct not supported yet for gradient back propagation")
        assert self._graph is not None and self._graph[0] == kind, f"call {kind}() first"
        _, leaves, image, _scale = self._graph
        last = self._dev.last
        names = list(leaves)
        inner = [(n + "_b", last[n]) for n in ("ij", "colors", "shade") if last[n].requires_grad]  # the 2.5-D arrays' adjoints, for inspection
        seed = torch.as_tensor(np.asarray(image_b, dtype=np.float64), device=image.device).reshape(image.shape)
        grads = torch.autograd.grad([image], [leaves[k] for k in names] + [t for _, t in inner], [seed], retain_graph=True, allow_unused=True)
        out = dict(zip(names, grads[: len(names)]))
        for (name, _), g in zip(inner, grads[len(names) :]):
            setattr(self.scene_2d, name, None if g is None else g[0].cpu().numpy())
        np_ = lambda g, like: np.zeros(np.shape(like)) if g is None else g.cpu().numpy()
        self.mesh._vertices_b = np_(out.get("vertices"), self.mesh.vertices)
        return out, np_

    def render_backward(self, image_b):
        out, np_ = self._backward("render", image_b)
        if "colors" in out:
            self.mesh.vertices_colors_b = np_(out["colors"], self.mesh.vertices_colors)
        self.light_ambient_b = float(out["ambient"]) if out.get("ambient") is not None else 0.0
        if self.light_directional is not None:
            self.light_directional_b = np_(out.get("directional"), self.light_directional)

    def render_depth_backward(self, depth_b):
        self._backward("render_depth", depth_b)

    # ---- deferred shading buffers (dr.py:1053-1174): forward only, sigma must be 0 --------------------------------------

    def render_deferred(self, camera, depth_scale=1, color=True, depth=True, face_id=True, barycentric=True, normal=True, luminosity=True, uv=True,
                        xyz=True, backface_culling=True):  # fmt: skip
        """-> dict of per-pixel buffers: every requested attribute is one group of channels of ONE triangle-soup render"""
        from .differentiable_renderer import Scene2DBase, renderScene

        m = self.mesh
        assert m is not None, "You need to provide a mesh first"
        if self.sigma > 0:
            raise BaseException("Antialiasing is not supposed to be used when using deferred rendering, please use sigma==0")
        points_2d, depths = camera.project_points(m.vertices)
        T = m.nb_faces
        soup = lambda a: np.asarray(a)[m.faces].reshape(3 * T, -1)
        channels = {}
        if depth:
            channels["depth"] = soup(depths) * depth_scale
        if face_id:
            channels["face_id"] = np.repeat(np.arange(T), 3)[:, None].astype(np.float64)
        if barycentric:
            channels["barycentric"] = np.tile(np.eye(3), (T, 1))
        if normal or luminosity:
            m.compute_vertex_normals()
        if normal:
            channels["normal"] = soup(m.vertex_normals)
        if luminosity:
            lum = np.zeros(m.nb_vertices) if self.light_directional is None else np.maximum(0, -m.vertex_normals.dot(self.light_directional))
            channels["luminosity"] = soup(lum + self.light_ambient)
        if xyz:
            channels["xyz"] = soup(m.vertices)
        if m.uv is None:
            if color:
                channels["color"] = soup(m.vertices_colors)
        elif uv:
            channels["uv"] = np.asarray(m.uv)[m.faces_uv].reshape(3 * T, 2)
        ranges, offset = {}, 0
        for k, v in channels.items():
            ranges[k] = (offset, offset + v.shape[1])
            offset += v.shape[1]
        colors = np.column_stack(list(channels.values()))
        nb_colors = colors.shape[1]
        background = np.zeros((camera.height, camera.width, nb_colors))
        if "depth" in channels:
            background[:, :, ranges["depth"][0] : ranges["depth"][1]] = depths.max()
        scene_2d = Scene2DBase(
            faces=np.arange(3 * T, dtype=np.uint32).reshape(T, 3), faces_uv=np.arange(3 * T, dtype=np.uint32).reshape(T, 3), ij=soup(points_2d),
            depths=soup(depths)[:, 0], textured=np.zeros(T, dtype=bool), uv=np.zeros((3 * T, 2)), shade=np.zeros(3 * T), colors=colors,
            shaded=np.zeros(T, dtype=bool), edgeflags=np.zeros((T, 3), dtype=bool), height=camera.height, width=camera.width, nb_colors=nb_colors,
            texture=np.zeros((0, 0)), background_image=background, background_color=None, backface_culling=backface_culling,
            integer_pixel_centers=self.integer_pixel_centers,
        )  # fmt: skip
        buffers, z_buffer = np.empty((camera.height, camera.width, nb_colors)), np.empty((camera.height, camera.width))
        renderScene(scene_2d, 0, buffers, z_buffer)
        return {k: buffers[:, :, a:b] for k, (a, b) in ranges.items()}
