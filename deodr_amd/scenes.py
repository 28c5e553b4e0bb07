"""Seeded synthetic 2.5-D scenes for the BASELINE configurations.

Everything here is plain NumPy producing :class:`deodr_amd.differentiable_renderer.Scene2D` inputs; it plays
the role the reference's `Scene3D.render` front half plays (project a mesh, light it, flag silhouette edges:
deodr/differentiable_renderer.py:896-983, deodr/triangulated_mesh.py:153-166) but only as a *generator of
inputs* for tests and benchmarks -- the rasterizer never depends on it.

  soup_scene      configs[0]: 256x256, 200 flat-colour soup triangles (generator modelled on
                  deodr/examples/triangle_soup_fitting.py:18-97, own RNG stream)
  deferred_scene  the frame of Scene3D.render_deferred: the soup of the sphere mesh, 15 channels, no silhouette edge, background image
  sphere_scene    configs[2] / configs[4]: bumpy UV sphere, 2*nu*n_rings triangles (100x100 -> 20 000 tris /
                  10 002 verts; 224x224 -> 100 352 tris / 50 178 verts), perspective camera, Gouraud colours
                  (+ depth channel) or planar-UV texture
  mesh_scene      configs[1] / configs[3]: any (vertices, faces) mesh, e.g. tests/golden/hand_mesh.npz
"""

from types import SimpleNamespace

import numpy as np

from .differentiable_renderer import Scene2D

# ------------------------------------------------------------------------------------------------------------------
# small geometry helpers


def perspective_camera(width, height, fov_deg, center, rot=None):
    """3x4 extrinsic and 3x3 intrinsic of a pinhole camera (x_cam = rot @ (x - center))."""
    rot = np.eye(3) if rot is None else np.asarray(rot, dtype=np.float64)
    focal = 0.5 * width / np.tan(0.5 * np.deg2rad(fov_deg))
    intrinsic = np.array([[focal, 0, width / 2], [0, focal, height / 2], [0, 0, 1.0]])
    extrinsic = np.column_stack((rot, -rot @ np.asarray(center, dtype=np.float64)))
    return SimpleNamespace(extrinsic=extrinsic, intrinsic=intrinsic, width=width, height=height)


def fit_camera(width, height, fov_deg, vertices, rot=None):
    """Camera on the -z side of the mesh far enough to see all of it (cf. reference dr.py:502-522)."""
    rot = np.eye(3) if rot is None else np.asarray(rot, dtype=np.float64)
    cam = vertices @ rot.T
    lo, hi = cam.min(axis=0), cam.max(axis=0)
    size = hi - lo
    t = np.tan(0.5 * np.deg2rad(fov_deg))
    dist = max(0.5 * size[0] / t, 0.5 * size[1] * (width / height) / t) + 0.5 * size[2]
    center = rot.T @ (0.5 * (lo + hi) + np.array([0, 0, -dist]))
    return perspective_camera(width, height, fov_deg, center, rot)


def project(camera, vertices):
    p = vertices @ camera.extrinsic[:, :3].T + camera.extrinsic[:, 3]
    depths = p[:, 2].copy()
    ij = (p[:, :2] / depths[:, None]) @ camera.intrinsic[:2, :2].T + camera.intrinsic[:2, 2]
    return np.ascontiguousarray(ij), depths


def vertex_normals(vertices, faces, clockwise):
    tri = vertices[faces]
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    if clockwise:
        n = -n
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-300)
    vn = np.zeros_like(vertices)
    for k in range(3):
        np.add.at(vn, faces[:, k], n)
    vn /= np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-300)
    return vn


def silhouette_edgeflags(ij, faces, clockwise):
    """[T,3] bool: edge n of face f (n=0:(v0,v1), 1:(v1,v2), 2:(v2,v0)) has exactly one front-facing
    incident face in the image (the rule of deodr/triangulated_mesh.py:153-166)."""
    faces = np.asarray(faces, dtype=np.int64)
    tri = ij[faces]
    u, v = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    cr = u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]
    visible = (cr > 0) if clockwise else (cr < 0)
    nv = int(faces.max()) + 1
    e = np.stack((faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]), axis=1)  # [T,3,2]
    key = np.minimum(e[..., 0], e[..., 1]) * nv + np.maximum(e[..., 0], e[..., 1])
    _, inv = np.unique(key.reshape(-1), return_inverse=True)
    inv = inv.reshape(key.shape)
    nvis = np.zeros(inv.max() + 1)
    np.add.at(nvis, inv, np.broadcast_to(visible[:, None], inv.shape).astype(np.float64))
    return nvis[inv] == 1


def smooth_texture(height, width, channels, seed, passes=5):
    rs = np.random.RandomState(seed)
    t = rs.rand(height, width, channels)
    for _ in range(passes):
        t = (t + np.roll(t, 1, 0) + np.roll(t, -1, 0) + np.roll(t, 1, 1) + np.roll(t, -1, 1)) / 5
    t -= t.min()
    return t / t.max()


def roty(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def rotx(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


# ------------------------------------------------------------------------------------------------------------------
# meshes


def bumpy_sphere(nu=100, n_rings=100, bump=0.1):
    """UV sphere r = 1 + bump*sin(5 theta)cos(4 phi): nu*n_rings + 2 vertices, 2*nu*n_rings triangles, outward CCW."""
    theta = np.pi * (np.arange(1, n_rings + 1) / (n_rings + 1))  # polar angle of the rings
    phi = 2 * np.pi * np.arange(nu) / nu
    th, ph = np.meshgrid(theta, phi, indexing="ij")
    r = 1 + bump * np.sin(5 * th) * np.cos(4 * ph)
    ring = np.stack((r * np.sin(th) * np.cos(ph), r * np.cos(th), r * np.sin(th) * np.sin(ph)), axis=-1).reshape(-1, 3)
    vertices = np.vstack((ring, [[0, 1.0, 0]], [[0, -1.0, 0]]))
    north, south = nu * n_rings, nu * n_rings + 1
    idx = np.arange(nu * n_rings).reshape(n_rings, nu)
    nxt = np.roll(idx, -1, axis=1)
    a, b, c, d = idx[:-1], nxt[:-1], idx[1:], nxt[1:]
    quads = np.concatenate((np.stack((a, b, c), -1).reshape(-1, 3), np.stack((b, d, c), -1).reshape(-1, 3)))
    top = np.stack((np.full(nu, north), nxt[0], idx[0]), -1)
    bot = np.stack((np.full(nu, south), idx[-1], nxt[-1]), -1)
    faces = np.concatenate((quads, top, bot)).astype(np.uint32)
    return vertices, faces


def mesh_scene(
    vertices, faces, width, height, nb_colors=3, rot=None, fov=60.0, camera=None, seed=1, depth_channel=False,
    textured=False, texture_size=256, sigma_edges=True, background_color=None, light=(-0.1, -0.5, -0.4), ambient=0.6,
    clockwise=None,
):  # fmt: skip
    """Project + light a mesh and assemble the Scene2D the rasterizer consumes.

    Untextured: colors = per-vertex RNG colour x luminosity (+ last channel = depth when depth_channel).
    Textured: planar (x,y) UVs scaled to the texture, shade = luminosity, seeded blurred texture."""
    vertices = np.asarray(vertices, dtype=np.float64)
    faces = np.asarray(faces, dtype=np.uint32)
    if camera is None:
        camera = fit_camera(width, height, fov, vertices, rot)
    ij, depths = project(camera, vertices)
    if clockwise is None:  # pick the winding flag that makes the NEAR side of a closed mesh front-facing
        tri = ij[faces.astype(np.int64)]
        u, v = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
        cr = u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]
        zf = depths[faces.astype(np.int64)].mean(axis=1)
        # the faces of one winding sign are on average nearer than those of the other: that sign is "front" (a count of
        # faces weighted by nearness -- round 1's rule -- is nearly balanced on a sphere in perspective and flipped with the pose)
        clockwise = bool(zf[cr > 0].mean() < zf[cr < 0].mean()) if (cr > 0).any() and (cr < 0).any() else bool((cr > 0).any())
    normals = vertex_normals(vertices, faces.astype(np.int64), clockwise)
    luminosity = np.maximum(0, -normals @ np.asarray(light, dtype=np.float64)) + ambient
    nv, nt = vertices.shape[0], faces.shape[0]
    rs = np.random.RandomState(seed)
    if textured:
        xy = vertices[:, :2]
        uv = (xy - xy.min(0)) / (xy.max(0) - xy.min(0)) * (texture_size - 1)
        texture = smooth_texture(texture_size, texture_size, nb_colors, seed)
        colors = np.zeros((nv, nb_colors))
        shade = luminosity
        flag = np.ones(nt, dtype=bool)
    else:
        uv = np.zeros((nv, 2))
        texture = np.zeros((0, 0))
        base = rs.rand(nv, nb_colors - 1 if depth_channel else nb_colors)
        colors = base * luminosity[:, None]
        if depth_channel:
            colors = np.column_stack((colors, depths / depths.max()))
        shade = np.zeros(nv)
        flag = np.zeros(nt, dtype=bool)
    edgeflags = silhouette_edgeflags(ij, faces, clockwise) if sigma_edges else np.zeros((nt, 3), dtype=bool)
    if background_color is None:
        background_color = np.array([0.5, 0.6, 0.7, 0.0][:nb_colors] + [0.0] * max(0, nb_colors - 4))
    return Scene2D(
        faces=faces, faces_uv=faces.copy(), ij=ij, depths=depths, textured=flag, uv=uv, shade=shade,
        colors=np.ascontiguousarray(colors), shaded=flag.copy(), edgeflags=edgeflags, height=height, width=width,
        nb_colors=nb_colors, texture=texture, background_color=np.asarray(background_color, dtype=np.float64),
        clockwise=clockwise, backface_culling=True, strict_edge=True, perspective_correct=False,
        integer_pixel_centers=True,
    )  # fmt: skip


def sphere_scene(size=1024, nu=100, n_rings=100, nb_colors=4, depth_channel=True, textured=False, texture_size=256, angle=0.0, seed=1):
    """BASELINE configs[2] (defaults) and configs[4] (size=2048, nu=n_rings=224, nb_colors=3, textured, 1024 texture).

    The sphere is tilted by a generic rotation: seen along an axis of symmetry, mirrored triangles have EXACTLY equal
    depth sums, and the blending order of their silhouette edges is then whatever the reference's unstable std::sort
    (H.h:2781) happens to produce -- undefined behaviour we do not want in a parity scene."""
    vertices, faces = bumpy_sphere(nu, n_rings)
    return mesh_scene(
        vertices, faces, size, size, nb_colors=nb_colors, rot=rotx(0.37) @ roty(0.23 + angle), seed=seed, depth_channel=depth_channel and not textured,
        textured=textured, texture_size=texture_size,
    )  # fmt: skip


def deferred_scene(size=1024, channels=15, nu=100, n_rings=100, angle=0.0, width=None, height=None, background_image=True, seed=2):
    """The 2.5-D scene `Scene3D.render_deferred` hands to renderScene (deodr/differentiable_renderer.py:1053-1174): the triangle SOUP of a mesh (three
    vertices per face, so that face ids / barycentrics / discontinuous uv can ride in the colour channels), `channels` interpolated channels (the
    reference's default set is 15: depth, face id, 3 barycentrics, 3 normal, luminosity, 3 xyz, 3 colour), no silhouette edge (it asserts sigma = 0),
    a background image (depth channel = far), back-face culling.  Here: the bumpy sphere of configs[2] with random channel values."""
    width, height = width or size, height or size
    m = sphere_scene(size=width, nu=nu, n_rings=n_rings, nb_colors=channels, depth_channel=True, angle=angle)
    f = m.faces.astype(np.int64).reshape(-1)
    rs = np.random.RandomState(seed)
    return Scene2D(
        faces=np.arange(f.size, dtype=np.uint32).reshape(-1, 3), faces_uv=np.arange(f.size, dtype=np.uint32).reshape(-1, 3),
        ij=m.ij[f] * [1.0, height / width], depths=m.depths[f], textured=m.textured, uv=np.zeros((f.size, 2)), shade=np.zeros(f.size),
        colors=np.ascontiguousarray(m.colors[f]), shaded=m.shaded, edgeflags=np.zeros_like(m.edgeflags), height=height, width=width, nb_colors=channels,
        texture=np.zeros((0, 0)), background_image=rs.rand(height, width, channels) if background_image else None,
        background_color=None if background_image else rs.rand(channels), clockwise=m.clockwise, backface_culling=True, strict_edge=True,
        perspective_correct=False, integer_pixel_centers=True,
    )  # fmt: skip


def soup_scene(n_tri=200, width=256, height=256, seed=2, clockwise=False, textured_ratio=0.0, flat=True, min_area=None, texture_size=64):
    """BASELINE configs[0]: a triangle soup with constant depth per triangle and all 3 edges flagged."""
    rs = np.random.RandomState(seed)
    ij = np.zeros((n_tri, 3, 2))
    if min_area is None:
        min_area = 0.0045 * width * height  # ~300 px^2 at 256x256 (the reference example rejects slivers too)
    for t in range(n_tri):
        for _attempt in range(10000):
            c = rs.rand(2) * [width, height]
            p = c + (rs.rand(3, 2) - 0.5) * 0.5 * [width, height]
            u, v = p[1] - p[0], p[2] - p[0]
            area2 = u[0] * v[1] - u[1] * v[0]
            if abs(area2) > 2 * min_area:
                break
        front_sign = 1.0 if clockwise else -1.0  # signedArea convention of H.h:391-398
        if area2 * front_sign < 0:
            p = p[::-1]
        ij[t] = p
    depths = np.repeat(rs.rand(n_tri), 3)
    textured = rs.rand(n_tri) < textured_ratio
    colors = np.repeat(rs.rand(n_tri, 3), 3, axis=0) if flat else rs.rand(3 * n_tri, 3)
    colors[np.repeat(textured, 3)] = 0
    shade = np.where(np.repeat(textured, 3), rs.rand(3 * n_tri), 0.0)
    uv = np.where(np.repeat(textured, 3)[:, None], rs.rand(3 * n_tri, 2) * (texture_size - 1), 0.0)
    faces = np.arange(3 * n_tri, dtype=np.uint32).reshape(-1, 3)
    return Scene2D(
        faces=faces, faces_uv=faces.copy(), ij=ij.reshape(-1, 2), depths=depths, textured=textured, uv=uv, shade=shade,
        colors=colors, shaded=textured.copy(), edgeflags=np.ones((n_tri, 3), dtype=bool), height=height, width=width,
        nb_colors=3, texture=smooth_texture(texture_size, texture_size, 3, seed + 100),
        background_image=np.tile(np.array([0.3, 0.5, 0.7])[None, None, :], (height, width, 1)), clockwise=clockwise,
        backface_culling=True, strict_edge=True, perspective_correct=False, integer_pixel_centers=True,
    )  # fmt: skip


def load_hand_mesh(path):
    """(vertices, faces) of the reference's hand.obj as committed in tests/golden/hand_mesh.npz, centred."""
    d = np.load(path)
    v = d["vertices"] - d["vertices"].mean(axis=0)
    return v, d["faces"].astype(np.uint32)


def hand_scene(mesh_path, size=1024, angle=0.0, textured=True, nb_colors=3, seed=0):
    """BASELINE configs[1] (textured Gouraud + 256^2 texture) and one view of configs[3] (untextured, angle)."""
    vertices, faces = load_hand_mesh(mesh_path)
    rot = np.diag([1.0, -1.0, -1.0]) @ roty(angle)  # camera orientation of deodr/mesh_fitter.py:258-282
    return mesh_scene(vertices, faces, size, size, nb_colors=nb_colors, rot=rot, fov=2 * np.rad2deg(np.arctan(0.25)), seed=seed, textured=textured, texture_size=256)
