"""NumPy-level mesh container with the interface of ``deodr.triangulated_mesh`` that the renderer and the fitters use.

``ColoredTriMesh`` keeps the reference's attribute and method names (deodr/triangulated_mesh.py:170-360: ``vertices``, ``faces``,
``set_vertices``, ``compute_vertex_normals``, ``vertex_normals``, ``edge_on_silhouette``, ``_vertices_b``, ``vertices_colors``,
``vertices_colors_b``, ``uv`` / ``faces_uv`` / ``texture``, ``clockwise``, ``adjacencies``) so that ``Scene3D`` and fitter code
written against DEODR runs on it -- but every computation is delegated to the device implementation in
:mod:`deodr_amd.scene3d` (index arrays + batched torch ops on the ROCm device; the reference uses SciPy sparse matrices on the
host).  File I/O, trimesh conversion and Loop subdivision are out of scope.
"""

import numpy as np
import torch

from .scene3d import MeshTopology


class TriMeshAdjacencies:
    """Connectivity queries of the reference's class of the same name, answered by :class:`MeshTopology`."""

    def __init__(self, faces, clockwise=False, nb_vertices=None, device="cuda"):
        self.faces = np.asarray(faces)
        self.clockwise = clockwise
        self.topology = MeshTopology(self.faces, nb_vertices, clockwise, device)
        self.nb_faces, self.nb_vertices, self.nb_edges = self.topology.nb_faces, self.topology.nb_vertices, self.topology.nb_edges
        self.is_manifold, self.is_closed = self.topology.is_manifold, self.topology.is_closed

    def _dev(self, a):
        return torch.as_tensor(np.asarray(a, dtype=np.float64), device=self.topology.device)

    def compute_face_normals(self, vertices):
        return self.topology.face_normals(self._dev(vertices)).cpu().numpy()

    def compute_vertex_normals_from_vertices(self, vertices):
        return self.topology.vertex_normals(self._dev(vertices)).cpu().numpy()

    def edge_on_silhouette(self, vertices_2d):
        return self.topology.edge_on_silhouette(self._dev(vertices_2d)).cpu().numpy().astype(bool)


class TriMesh:
    def __init__(self, faces, vertices, clockwise=False, compute_adjacencies=True, device="cuda"):
        faces = np.array(faces)
        assert np.issubdtype(faces.dtype, np.integer) and faces.ndim == 2 and faces.shape[1] == 3 and np.all(faces >= 0)
        self._faces = faces
        self.nb_vertices, self.nb_faces = int(vertices.shape[0]), int(faces.shape[0])
        self.clockwise = clockwise
        self.device = device
        self._vertices_b = np.zeros((self.nb_vertices, 3))
        self._adjacencies = None
        self.set_vertices(vertices)
        if compute_adjacencies:
            self.compute_adjacencies()

    def compute_adjacencies(self):
        self._adjacencies = TriMeshAdjacencies(self._faces, self.clockwise, self.nb_vertices, self.device)

    @property
    def faces(self):
        return self._faces

    @property
    def vertices(self):
        return self._vertices

    @property
    def adjacencies(self):
        if self._adjacencies is None:
            self.compute_adjacencies()
        return self._adjacencies

    def set_vertices(self, vertices):
        self._vertices = vertices
        self._vertex_normals = None
        self._normals_graph = None

    def compute_vertex_normals(self):
        """(keeps the autograd graph so that compute_vertex_normals_backward needs no hand-written adjoint)"""
        topo = self.adjacencies.topology
        v = torch.as_tensor(np.asarray(self._vertices, dtype=np.float64), device=topo.device).requires_grad_(True)
        n = topo.vertex_normals(v)
        self._normals_graph = (v, n)
        self._vertex_normals = n.detach().cpu().numpy()

    @property
    def vertex_normals(self):
        if self._vertex_normals is None:
            self.compute_vertex_normals()
        return self._vertex_normals

    def compute_vertex_normals_backward(self, vertex_normals_b):
        """adds the pull-back of ``vertex_normals_b`` to ``_vertices_b`` (triangulated_mesh.py:290-293)"""
        if self._normals_graph is None:
            self.compute_vertex_normals()
        v, n = self._normals_graph
        (g,) = torch.autograd.grad(n, v, torch.as_tensor(np.asarray(vertex_normals_b, dtype=np.float64), device=n.device), retain_graph=True)
        self._vertices_b = self._vertices_b + g.cpu().numpy()

    def edge_on_silhouette(self, points_2d):
        return self.adjacencies.edge_on_silhouette(points_2d)


class ColoredTriMesh(TriMesh):
    def __init__(self, faces, vertices=None, clockwise=False, faces_uv=None, uv=None, texture=None, colors=None, nb_colors=None,
                 compute_adjacencies=True, device="cuda"):  # fmt: skip
        super().__init__(faces, vertices=vertices, clockwise=clockwise, compute_adjacencies=compute_adjacencies, device=device)
        self.faces_uv, self.uv, self.texture = faces_uv, uv, texture
        self.vertices_colors = colors
        self.textured = texture is not None
        self.nb_colors = nb_colors
        if nb_colors is None:
            if texture is not None:
                self.nb_colors = texture.shape[2]
            elif colors is not None:
                self.nb_colors = colors.shape[1]
        self.vertices_colors_b = None

    def set_vertices_colors(self, colors):
        self.vertices_colors = colors

    def subdivise(self, n_iter):
        if n_iter:
            raise NotImplementedError("Loop subdivision is outside the scope of deodr_amd (SURVEY.md section 2)")
        return self
