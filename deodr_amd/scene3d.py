"""Device-resident front half of the rasterizer: camera, lighting, silhouette flags, mesh normals, rigid energy.

The reference assembles the 2.5-D scene of every frame on the host with NumPy + SciPy (projection and distortion
``deodr/differentiable_renderer.py:341-438``, luminosity ``:814-850``, Scene2D assembly ``:896-983``, silhouette edge flags
``deodr/triangulated_mesh.py:153-166``, vertex normals ``:113-151``, Laplacian energy ``deodr/laplacian_rigid_energy.py``) and
its PyTorch layer round-trips through ``.numpy()`` (``deodr/pytorch/differentiable_renderer_pytorch.py:52-54``).  Once the
rasterizer takes tens of microseconds that glue is the whole iteration.  Here every array lives on the ROCm device from the
mesh vertices to the loss: the O(V) algebra is a handful of batched torch ops over ``n_views`` views (differentiated by
autograd), the rasterizer is the HIP library behind one autograd Function, and nothing visits the host inside a fit loop.

Same math as the reference (float64 by default), not the same code: batched over views, index_add instead of SciPy sparse
products, adjacency as flat index arrays built once per topology.
"""

import numpy as np
import torch

from .hip_renderer import DeviceScene, HipRasterizer


def _t(a, device, dtype=torch.float64):
    return (a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))).to(device=device, dtype=dtype)


class DeviceCamera:
    """``n`` pinhole cameras with OpenCV's distortion model (k1, k2, p1, p2, k3), as device tensors.

    ``extrinsic`` [n,3,4] (or [3,4]) world -> camera, ``intrinsic`` [n,3,3] (or [3,3]), ``distortion`` [n,5] / [5] / None.
    Reference: ``Camera`` deodr/differentiable_renderer.py:250-438."""

    def __init__(self, extrinsic, intrinsic, height, width, distortion=None, device="cuda", dtype=torch.float64):
        self.device, self.dtype = torch.device(device), dtype
        e, k = _t(extrinsic, self.device, dtype), _t(intrinsic, self.device, dtype)
        self.extrinsic = e[None] if e.dim() == 2 else e
        self.intrinsic = k[None] if k.dim() == 2 else k
        # one camera matrix shared by n views (extrinsic [n,3,4] with one intrinsic [3,3], or the other way round): every array is
        # expanded to n views HERE -- the fused kernels index all three per view (`intrinsic[9 b + i]`, `extrinsic[12 b + i]`)
        self.n_views = max(int(self.extrinsic.shape[0]), int(self.intrinsic.shape[0]))
        for name in ("extrinsic", "intrinsic"):
            a = getattr(self, name)
            if a.shape[0] not in (1, self.n_views):
                raise ValueError(f"DeviceCamera: {name} has {a.shape[0]} views, expected 1 or {self.n_views}")
            setattr(self, name, a.expand(self.n_views, *a.shape[1:]))
        self.height, self.width = int(height), int(width)
        self.distortion = None
        if distortion is not None:
            d = _t(distortion, self.device, dtype)
            self.distortion = (d[None] if d.dim() == 1 else d).expand(self.n_views, 5).contiguous()
        self.extrinsic, self.intrinsic = self.extrinsic.contiguous(), self.intrinsic.contiguous()

    @classmethod
    def stack(cls, cameras, device="cuda", dtype=torch.float64):
        """One batched camera from a list of cameras with ``extrinsic`` / ``intrinsic`` / ``distortion`` / ``height`` / ``width``
        attributes (e.g. the reference's ``Camera`` objects or this module's NumPy-level mirror)."""
        c0 = cameras[0]
        dist = None
        if any(getattr(c, "distortion", None) is not None for c in cameras):
            dist = np.stack([np.zeros(5) if c.distortion is None else np.asarray(c.distortion, dtype=np.float64) for c in cameras])
        return cls(np.stack([np.asarray(c.extrinsic) for c in cameras]), np.stack([np.asarray(c.intrinsic) for c in cameras]), c0.height,
                   c0.width, dist, device, dtype)  # fmt: skip

    def world_to_camera(self, points_3d):
        """[V,3] (shared by the views) or [n,V,3] -> [n,V,3]"""
        p = points_3d if points_3d.dim() == 3 else points_3d[None].expand(self.n_views, -1, -1)
        return p @ self.extrinsic[:, :, :3].transpose(1, 2) + self.extrinsic[:, None, :, 3]

    def project_points(self, points_3d):
        """-> (image coordinates [n,V,2] with x = column first, depths [n,V]); differentiable (dr.py:341-395)."""
        from . import fronthalf

        batch_ok = points_3d.dim() == 2 or int(points_3d.shape[0]) == self.n_views  # (else: the torch path below broadcasts, or raises)
        if batch_ok and fronthalf.usable(points_3d, self.extrinsic, self.intrinsic) and (self.distortion is None or fronthalf.usable(self.distortion)):
            # one kernel (two with its adjoint) instead of ~15 + ~25 torch kernels
            p = points_3d if points_3d.dim() == 3 else points_3d[None].expand(self.n_views, -1, -1)
            return fronthalf.ProjectPointsFunc.apply(p, self.extrinsic, self.intrinsic, self.distortion)
        pc = self.world_to_camera(points_3d)
        depths = pc[..., 2]
        xy = pc[..., :2] / depths[..., None]
        if self.distortion is not None:
            k1, k2, p1, p2, k3 = (self.distortion[:, i, None] for i in range(5))
            x, y = xy[..., 0], xy[..., 1]
            x2, y2 = x * x, y * y
            r2 = x2 + y2
            r4 = r2 * r2
            radial = 1 + k1 * r2 + k2 * r4 + k3 * (r2 * r4)
            xd = x * radial + (2 * p1 * x * y + p2 * (r2 + 2 * x2))
            yd = y * radial + (p1 * (r2 + 2 * y2) + 2 * p2 * x * y)
            xy = torch.stack((xd, yd), dim=-1)
        ij = xy @ self.intrinsic[:, :2, :2].transpose(1, 2) + self.intrinsic[:, None, :2, 2]
        return ij, depths


class MeshTopology:
    """Static connectivity of a triangle mesh as flat device index arrays, built once on the host.

    ``face_edge`` [T,3]: id of the edge (v0,v1), (v1,v2), (v2,v0) of every face -- the same slot order as the rasterizer's
    ``edgeflags`` (reference ``TriMeshAdjacencies`` deodr/triangulated_mesh.py:17-110, which keeps SciPy sparse matrices)."""

    def __init__(self, faces, nb_vertices=None, clockwise=False, device="cuda"):
        f = np.asarray(faces).astype(np.int64)
        assert f.ndim == 2 and f.shape[1] == 3
        self.device = torch.device(device)
        self.clockwise = bool(clockwise)
        self.nb_faces = int(f.shape[0])
        self.nb_vertices = int(nb_vertices) if nb_vertices is not None else int(f.max()) + 1
        e = np.concatenate((f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]))  # slot-major: [3T,2]
        key = np.minimum(e[:, 0], e[:, 1]) * self.nb_vertices + np.maximum(e[:, 0], e[:, 1])
        _, edge_id, counts = np.unique(key, return_inverse=True, return_counts=True)
        self.nb_edges = int(edge_id.max()) + 1 if len(edge_id) else 0
        self.is_manifold = bool(np.all(counts <= 2))
        self.is_closed = bool(self.is_manifold and np.all(counts == 2))
        self.faces = torch.as_tensor(f, device=self.device)
        self.face_edge = torch.as_tensor(edge_id.reshape(3, -1).T.copy(), device=self.device)  # [T,3]
        # for the fused silhouette kernel (manifold meshes): the face across every edge slot, 0xffffffff on a boundary
        self._edge_faces = None
        if self.is_manifold and self.nb_faces:
            slot_face = np.tile(np.arange(self.nb_faces), 3)  # face of slot-major entry i of `e`
            order = np.argsort(edge_id, kind="stable")
            sorted_edge, sorted_face = edge_id[order], slot_face[order]
            other = np.full(3 * self.nb_faces, 0xFFFFFFFF, dtype=np.int64)
            same = sorted_edge[1:] == sorted_edge[:-1]  # (at most two slots per edge)
            other[order[:-1][same]] = sorted_face[1:][same]
            other[order[1:][same]] = sorted_face[:-1][same]
            self._edge_faces = torch.as_tensor(other.reshape(3, -1).T.astype(np.uint32).view(np.int32).copy(), device=self.device)
            self._faces_u32 = torch.as_tensor(f.astype(np.uint32).view(np.int32), device=self.device).contiguous()
        # graph Laplacian over "shares a face" adjacency, and M = L^T L as a COO list (dr: laplacian_rigid_energy.py:18-22)
        a = np.zeros((0, 2), dtype=np.int64)
        if self.nb_faces:
            pairs = np.concatenate((f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]], f[:, [1, 0]], f[:, [2, 1]], f[:, [0, 2]]))
            a = np.unique(pairs, axis=0)
        import scipy.sparse as sp

        adj = sp.coo_matrix((np.ones(len(a)), (a[:, 0], a[:, 1])), shape=(self.nb_vertices, self.nb_vertices)).tocsr()
        lap = sp.diags(np.asarray(adj.sum(axis=1)).ravel()) - adj
        m = (lap.T @ lap).tocoo()
        u32 = lambda a: torch.as_tensor(np.ascontiguousarray(a).astype(np.uint32).view(np.int32), device=self.device)
        mc = m.tocsr()
        mc.sort_indices()
        self._m_csr = (u32(mc.indptr), u32(mc.indices), torch.as_tensor(mc.data.astype(np.float64), device=self.device))  # rows of L^T L
        # vertex -> the slots 3 f + corner it occupies in `faces` (the gathers of the shading adjoint, no atomics)
        corner_vertex = f.reshape(-1)
        self._vf_corners = u32(np.argsort(corner_vertex, kind="stable"))
        self._vf_offsets = u32(np.concatenate(([0], np.cumsum(np.bincount(corner_vertex, minlength=self.nb_vertices)))))
        if self._edge_faces is None and self.nb_faces:
            self._faces_u32 = u32(f)
        self._m_rows = torch.as_tensor(m.row.astype(np.int64), device=self.device)
        self._m_cols = torch.as_tensor(m.col.astype(np.int64), device=self.device)
        self._m_vals = torch.as_tensor(m.data.astype(np.float64), device=self.device)
        self.n_components = int(sp.csgraph.connected_components(adj, directed=False)[0]) if self.nb_vertices else 0

    # ---- differentiable geometry (batched: vertices [..., V, 3]) ---------------------------------------------------------

    def face_normals(self, vertices):
        tri = vertices[..., self.faces, :]  # [..., T, 3, 3]
        n = torch.cross(tri[..., 1, :] - tri[..., 0, :], tri[..., 2, :] - tri[..., 0, :], dim=-1)
        if self.clockwise:
            n = -n
        return n / n.norm(dim=-1, keepdim=True)

    def vertex_normals(self, vertices):
        """normalised sum of the unit normals of the faces around each vertex (triangulated_mesh.py:113-151)"""
        fn = self.face_normals(vertices)
        acc = torch.zeros_like(vertices)
        idx = self.faces.reshape(-1)
        acc = acc.index_add(-2, idx, fn.repeat_interleave(3, dim=-2))
        return acc / acc.norm(dim=-1, keepdim=True)

    def edge_on_silhouette(self, ij):
        """[..., V, 2] image coordinates -> uint8 [..., T, 3]: the edge has exactly one front-facing incident face in the image
        (triangulated_mesh.py:153-166).  No gradient (the flags select which edges are antialiased)."""
        from . import fronthalf

        if self._edge_faces is not None and ij.dim() == 3 and fronthalf.usable(ij):
            return fronthalf.silhouette_flags(ij, self._faces_u32, self._edge_faces, self.clockwise)  # one kernel instead of eight
        with torch.no_grad():
            tri = ij[..., self.faces, :]
            u, v = tri[..., 1, :] - tri[..., 0, :], tri[..., 2, :] - tri[..., 0, :]
            cr = u[..., 0] * v[..., 1] - u[..., 1] * v[..., 0]
            visible = (cr > 0) if self.clockwise else (cr < 0)  # [..., T]
            count = torch.zeros(ij.shape[:-2] + (self.nb_edges,), dtype=torch.int32, device=ij.device)
            count = count.index_add(-1, self.face_edge.reshape(-1), visible.to(torch.int32).repeat_interleave(3, dim=-1))
            return (count[..., self.face_edge] == 1).to(torch.uint8)

    def laplacian_quadratic(self, diff):
        """M diff with M = L^T L, diff [..., V, 3] (one index_add over the non-zeros of M)"""
        out = torch.zeros_like(diff)
        return out.index_add(-2, self._m_rows, self._m_vals[:, None] * diff[..., self._m_cols, :])


class LaplacianRigidEnergyDevice:
    """As-rigid-as-possible energy 0.5 c (V - V_ref)^T (L^T L x I3) (V - V_ref) and its gradient, on the device
    (reference deodr/laplacian_rigid_energy.py:15-41; its PyTorch twin falls back to SciPy on the host, :38-46)."""

    def __init__(self, topology, vertices_ref, cregu):
        if topology.n_components > 1:
            raise BaseException("You have more than one connected component in your mesh.")
        self.topology, self.cregu = topology, float(cregu)
        self.vertices_ref = _t(vertices_ref, topology.device).clone()

    def evaluate(self, vertices):
        """-> (energy, gradient [V,3]); the energy is differentiable too (autograd sees plain tensor ops)"""
        from . import fronthalf

        if fronthalf.usable(vertices, self.vertices_ref) and vertices.dim() == 2:
            return fronthalf.RigidEnergyFunc.apply(vertices, self.vertices_ref, self.topology, self.cregu)  # one kernel, deterministic
        diff = vertices - self.vertices_ref
        grad = self.cregu * self.topology.laplacian_quadratic(diff)
        return 0.5 * (diff * grad).sum(), grad


class RenderViewsFunc(torch.autograd.Function):
    """(ij [n,V,2], colors [n,V,C], shade [n,V]) -> image [n,H,W,C]: the HIP rasterizer with gradients for all three.

    ``depths`` [n,V] and ``edgeflags`` [n,T,3] are inputs without gradient (dr.py:1017: the z buffer is not differentiated; the
    flags select which edges are antialiased).  ALL five per-view arrays are saved: the DeviceScene / workspace are shared by
    every render of a Scene3DDevice, and when another render has used them since (two cameras, or two vertex sets, in one loss)
    the adjoint rebuilds this forward's state from its own inputs, not from whatever the scene holds now."""

    @staticmethod
    def forward(ctx, ij, colors, shade, depths, edgeflags, device_scene, rasterizer, sigma):
        device_scene.set_views(ij=ij.detach(), colors=colors.detach(), shade=shade.detach(), depths=depths.detach(), edgeflags=edgeflags)
        image, z = rasterizer.render(device_scene, sigma)
        ctx.ds, ctx.r, ctx.sigma, ctx.generation = device_scene, rasterizer, sigma, rasterizer.generation
        ctx.save_for_backward(ij, colors, shade, depths, edgeflags)
        ctx.mark_non_differentiable(z)
        return image, z

    @staticmethod
    def backward(ctx, image_b, _z_b):
        ij, colors, shade, depths, edgeflags = ctx.saved_tensors
        if ctx.r.generation != ctx.generation:  # another forward used the scene since: restore this one's inputs
            ctx.ds.set_views(ij=ij.detach(), colors=colors.detach(), shade=shade.detach(), depths=depths.detach(), edgeflags=edgeflags)
        g = ctx.r.render_backward(ctx.ds, image_b=image_b, generation=ctx.generation, sigma=ctx.sigma)
        ctx.uv_b, ctx.texture_b = g["uv_b"], g["texture_b"]
        return g["ij_b"].to(ij.dtype), g["colors_b"].to(colors.dtype), g["shade_b"].to(shade.dtype), None, None, None, None, None


class RenderViewsL2Func(torch.autograd.Function):
    """(ij, colors, shade) -> (sum over the views of sum (image - obs)^2, image): ONE ``deodr_hip_render_scene_fit`` call renders and
    back-propagates the residual (the forward raster knows dL/dimage of a pixel the moment the pixel is resolved), so the backward
    of this op only scales the gradients the forward left.  What the reference's colour fitters write as render, subtract, square,
    sum, render_backward (deodr/mesh_fitter.py:296-318, 533-548) -- half the rasterizer time of the two-call path."""

    @staticmethod
    def forward(ctx, ij, colors, shade, depths, edgeflags, obs, device_scene, rasterizer, sigma):
        device_scene.set_views(ij=ij.detach(), colors=colors.detach(), shade=shade.detach(), depths=depths.detach(), edgeflags=edgeflags)
        loss = torch.empty(1, dtype=torch.float64, device=ij.device)  # sum (image - obs)^2, from the same launches (no pass over the frame)
        image, z, g = rasterizer.render_fit(device_scene, obs, sigma, clear_grads=False, loss_out=loss)
        ctx.save_for_backward(g["ij_b"].to(ij.dtype), g["colors_b"].to(colors.dtype), g["shade_b"].to(shade.dtype))
        ctx.uv_b, ctx.texture_b = g["uv_b"], g["texture_b"]
        ctx.mark_non_differentiable(image)
        return loss[0], image

    @staticmethod
    def backward(ctx, loss_b, _image_b):
        ij_b, colors_b, shade_b = ctx.saved_tensors
        return loss_b.to(ij_b.dtype) * ij_b, loss_b.to(colors_b.dtype) * colors_b, loss_b.to(shade_b.dtype) * shade_b, None, None, None, None, None, None


class DeviceMesh:
    """A coloured (or textured) triangle mesh on the device: topology + per-vertex attributes.

    The counterpart of the reference's ``ColoredTriMesh`` (deodr/triangulated_mesh.py:302-360) for the attributes the renderer
    consumes.  ``vertices`` may be a leaf tensor that requires grad."""

    def __init__(self, faces, vertices, clockwise=False, colors=None, uv=None, faces_uv=None, texture=None, device="cuda", dtype=torch.float64):
        self.device, self.dtype = torch.device(device), dtype
        self.topology = MeshTopology(faces, int(np.shape(vertices)[-2]), clockwise, self.device)
        self.faces_np = np.asarray(faces).astype(np.uint32)
        self.clockwise = bool(clockwise)
        self.vertices = _t(vertices, self.device, dtype)
        self.vertices_colors = None if colors is None else _t(colors, self.device, dtype)
        self.uv = None if uv is None else _t(uv, self.device, dtype)
        self.faces_uv_np = None if faces_uv is None else np.asarray(faces_uv).astype(np.uint32)
        self.texture = None if texture is None else _t(texture, self.device, dtype)

    @property
    def nb_vertices(self):
        return self.topology.nb_vertices

    @property
    def nb_faces(self):
        return self.topology.nb_faces

    def set_vertices(self, vertices):
        self.vertices = vertices

    def set_vertices_colors(self, colors):
        self.vertices_colors = colors


class Scene3DDevice:
    """One mesh, one directional + one ambient light, ``n`` cameras: everything up to the image stays on the device.

    Mirrors what ``Scene3D.render`` / ``render_depth`` compute in the reference (dr.py:764-1051) with the per-view work
    batched: project -> silhouette flags -> vertex colours (or shade) -> rasterize.  Gradients reach ``mesh.vertices``,
    ``mesh.vertices_colors``, the light tensors and anything upstream through autograd."""

    def __init__(self, sigma=1.0, perspective_correct=False, integer_pixel_centers=True, pixel_dtype=torch.float64):
        self.sigma, self.perspective_correct, self.integer_pixel_centers = float(sigma), bool(perspective_correct), bool(integer_pixel_centers)
        self.pixel_dtype = pixel_dtype
        self.mesh = None
        self.light_directional, self.light_ambient = None, 0.0
        self.background_color, self.background_image = None, None
        self._state = None  # (key, DeviceScene, HipRasterizer)
        self.last = {}

    def set_mesh(self, mesh):
        self.mesh = mesh

    def set_light(self, light_directional, light_ambient):
        self.light_directional = None if light_directional is None else _t(light_directional, self.mesh.device if self.mesh else "cuda")
        self.light_ambient = light_ambient

    def set_background_color(self, background_color):
        if self.background_image is not None:
            raise BaseException("you cannot provide both background image and background color")
        self.background_color = np.asarray(background_color, dtype=np.float64).reshape(-1)

    def set_background_image(self, background_image):
        if self.background_color is not None:
            raise BaseException("you cannot provide both background image and background color")
        self.background_image = background_image

    # ---- pieces ------------------------------------------------------------------------------------------------------

    def vertices_luminosity(self, vertices):
        """max(0, -n . l) + ambient per vertex (dr.py:814-822); vertices [..., V, 3]"""
        amb = self.light_ambient
        if self.light_directional is None:
            return torch.zeros(vertices.shape[:-1], dtype=vertices.dtype, device=vertices.device) + amb
        from . import fronthalf

        light = self.light_directional
        if fronthalf.usable(vertices, light) and self.mesh.nb_faces and (not torch.is_tensor(amb) or fronthalf.usable(amb)):
            amb = amb if torch.is_tensor(amb) else torch.full((), float(amb), dtype=torch.float64, device=vertices.device)
            lum = fronthalf.VertexLuminosityFunc.apply(vertices if vertices.dim() == 3 else vertices[None], light, amb, self.mesh.topology)  # 1 kernel
            return lum if vertices.dim() == 3 else lum[0]
        normals = self.mesh.topology.vertex_normals(vertices)
        return torch.relu(-(normals * light).sum(-1)) + amb

    def _rasterizer(self, n_views, height, width, nb_colors, textured, backface_culling):
        m = self.mesh
        key = (id(m.topology), n_views, height, width, nb_colors, textured, backface_culling, self.perspective_correct, self.integer_pixel_centers,
               None if self.background_color is None else self.background_color.tobytes(), id(self.background_image), id(m.texture))  # fmt: skip
        if self._state is None or self._state[0] != key:
            V, T = m.nb_vertices, m.nb_faces
            dev, vd = m.device, m.dtype
            zeros = lambda *s: torch.zeros(s, dtype=vd, device=dev)
            bgi = self.background_image
            if bgi is not None:
                bgi = _t(bgi, dev, self.pixel_dtype)
                bgi = bgi[None].expand(n_views, -1, -1, -1) if bgi.dim() == 3 else bgi
            ds = DeviceScene(
                m.faces_np, m.faces_uv_np if textured else m.faces_np, np.full(T, textured, dtype=np.uint8), np.full(T, textured, dtype=np.uint8),
                m.uv if textured else zeros(V, 2), zeros(n_views, V, 2), torch.ones(n_views, V, dtype=vd, device=dev), zeros(n_views, V, nb_colors),
                zeros(n_views, V), torch.zeros(n_views, T, 3, dtype=torch.uint8, device=dev), height, width,
                texture=m.texture if textured else None, background_color=self.background_color, background_image=bgi,
                clockwise=m.clockwise, backface_culling=backface_culling, strict_edge=True, perspective_correct=self.perspective_correct,
                integer_pixel_centers=self.integer_pixel_centers, vertex_dtype=vd, pixel_dtype=self.pixel_dtype, device=dev,
            )  # fmt: skip
            keep = self._state[2] if self._state is not None and self._state[2].dims == (T, height, width, nb_colors, n_views) else None
            self._state = (key, ds, keep or HipRasterizer.for_scene(ds))
        return self._state[1], self._state[2]

    def _rasterize(self, camera, ij, depths, colors, shade, textured, backface_culling):
        if (self.background_image is None) == (self.background_color is None):
            raise BaseException("You need to provide either a background image or background color")
        n = camera.n_views
        ds, r = self._rasterizer(n, camera.height, camera.width, int(colors.shape[-1]), textured, backface_culling)
        flags = self.mesh.topology.edge_on_silhouette(ij) if self.sigma > 0 else torch.zeros((n, self.mesh.nb_faces, 3), dtype=torch.uint8, device=ij.device)
        self.last = dict(ij=ij, depths=depths, edgeflags=flags, colors=colors, shade=shade)
        image, z = RenderViewsFunc.apply(ij, colors, shade, depths.detach(), flags, ds, r, self.sigma)
        return image, z

    def _rasterize_l2(self, camera, ij, depths, colors, shade, textured, backface_culling, obs):
        """-> (sum (image - obs)^2 over all views, image [n,H,W,C]) through the one-call fit step"""
        if (self.background_image is None) == (self.background_color is None):
            raise BaseException("You need to provide either a background image or background color")
        n = camera.n_views
        ds, r = self._rasterizer(n, camera.height, camera.width, int(colors.shape[-1]), textured, backface_culling)
        flags = self.mesh.topology.edge_on_silhouette(ij) if self.sigma > 0 else torch.zeros((n, self.mesh.nb_faces, 3), dtype=torch.uint8, device=ij.device)
        self.last = dict(ij=ij, depths=depths, edgeflags=flags, colors=colors, shade=shade)
        return RenderViewsL2Func.apply(ij, colors, shade, depths.detach(), flags, obs, ds, r, self.sigma)

    # ---- the reference's entry points, batched over the camera's views -----------------------------------------------

    def render_l2(self, camera, obs, backface_culling=True):
        """-> (sum over views and pixels of (render(camera) - obs)^2 as a differentiable scalar, the rendered images [n,H,W,C]).
        ``obs`` [n,H,W,C] in the scene's pixel dtype, contiguous (anything else is converted at every call)."""
        m = self.mesh
        assert m is not None, "You need to provide a mesh first."
        ij, depths = camera.project_points(m.vertices)
        n, V = ij.shape[0], m.nb_vertices
        obs = obs.to(device=ij.device, dtype=self.pixel_dtype)
        lum = self.vertices_luminosity(m.vertices)
        lum = lum[None].expand(n, -1) if lum.dim() == 1 else lum
        if m.uv is not None:
            assert m.texture is not None
            colors = torch.zeros((n, V, int(m.texture.shape[2])), dtype=m.dtype, device=m.device)
            return self._rasterize_l2(camera, ij, depths, colors, lum, True, backface_culling, obs.expand(n, -1, -1, -1).contiguous())
        vc = m.vertices_colors if m.vertices_colors.dim() == 3 else m.vertices_colors[None].expand(n, -1, -1)
        shade = torch.zeros((n, V), dtype=m.dtype, device=m.device)
        return self._rasterize_l2(camera, ij, depths, vc * lum[..., None], shade, False, backface_culling, obs.expand(n, -1, -1, -1).contiguous())

    def render(self, camera, return_z_buffer=False, backface_culling=True):
        """-> image [n,H,W,C] (and z_buffer [n,H,W]); dr.py:896-983"""
        m = self.mesh
        assert m is not None, "You need to provide a mesh first."
        ij, depths = camera.project_points(m.vertices)
        n, V = ij.shape[0], m.nb_vertices
        lum = self.vertices_luminosity(m.vertices)
        lum = lum[None].expand(n, -1) if lum.dim() == 1 else lum
        if m.uv is not None:
            assert m.texture is not None
            colors = torch.zeros((n, V, int(m.texture.shape[2])), dtype=m.dtype, device=m.device)
            image, z = self._rasterize(camera, ij, depths, colors, lum, True, backface_culling)
        else:
            vc = m.vertices_colors if m.vertices_colors.dim() == 3 else m.vertices_colors[None].expand(n, -1, -1)
            shade = torch.zeros((n, V), dtype=m.dtype, device=m.device)
            image, z = self._rasterize(camera, ij, depths, vc * lum[..., None], shade, False, backface_culling)
        return (image, z) if return_z_buffer else image

    def render_depth(self, camera, depth_scale=1.0, backface_culling=True):
        """-> depth image [n,H,W,1]: the depth of every vertex rendered as its colour (dr.py:1001-1036)"""
        m = self.mesh
        ij, depths = camera.project_points(m.vertices)
        shade = torch.zeros_like(depths)
        image, _ = self._rasterize(camera, ij, depths, depths[..., None] * depth_scale, shade, False, backface_culling)
        return image
