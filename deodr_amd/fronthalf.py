"""Fused kernels for the O(V) algebra of a fit iteration (``deodr_amd/csrc/dr_fronthalf.h``, C ABI in ``include/deodr_hip.h``):
rigid transform, camera projection (+ distortion), silhouette flags and the momentum update, each one launch (two with its adjoint)
instead of the 10 - 50 torch kernels the same formulas take (``deodr_amd/scene3d.py`` keeps those formulas: they are what runs on
tensors that are not float64 ROCm tensors -- the CPU suite -- and what the kernels are tested against)."""

import ctypes as C

import torch

from .hip_renderer import _check, _stream, lib


def usable(*tensors):
    """the kernels take contiguous float64 tensors on a ROCm device"""
    return all(t is not None and t.is_cuda and t.dtype == torch.float64 for t in tensors)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


_bound = False


def _lib():
    global _bound
    L = lib()
    if not _bound:
        vp, i, d = C.c_void_p, C.c_int, C.c_double
        L.deodr_hip_rigid_transform.argtypes = [vp, vp, vp, vp, i, i, vp]
        L.deodr_hip_rigid_transform_b.argtypes = [vp, vp, vp, vp, vp, i, i, vp]
        L.deodr_hip_project_points.argtypes = [vp] * 6 + [i, i, vp]
        L.deodr_hip_project_points_b.argtypes = [vp] * 7 + [i, i, vp]
        L.deodr_hip_silhouette_flags.argtypes = [vp, vp, vp, vp, i, i, i, i, vp]
        L.deodr_hip_momentum_update.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, vp, d, d, vp, vp, vp, vp, vp, d, vp, C.c_size_t, vp]
        L.deodr_hip_fit_front.argtypes = [vp] * 4 + [i] + [vp] * 6 + [i] + [vp] * 7 + [d, vp, vp, vp, C.c_size_t, i, i, i, vp]
        L.deodr_hip_fit_scratch_bytes.argtypes, L.deodr_hip_fit_scratch_bytes.restype = [i, i], C.c_size_t
        L.deodr_hip_fit_pose_project.argtypes = [vp] * 11 + [d, i, i, vp]
        L.deodr_hip_fit_pose_project_b.argtypes = [vp] * 9 + [d, vp, vp, vp, C.c_size_t, i, i, vp, i, vp, vp]
        L.deodr_hip_views_gradient_sum.argtypes = [vp] * 6 + [d, vp, i, i, vp, i, vp, vp]
        L.deodr_hip_vertex_shade.argtypes = [vp] * 7 + [i, vp, vp, i, i, i, vp]
        L.deodr_hip_vertex_shade_b.argtypes = [vp] * 7 + [i, vp, vp, vp, vp, vp, C.c_size_t, i, i, i, vp]
        L.deodr_hip_rigid_energy.argtypes = [vp] * 5 + [d, vp, vp, vp, d, vp, C.c_size_t, i, vp]
        L.deodr_hip_l2_loss.argtypes = [vp, vp, i, C.c_size_t, vp, vp, C.c_size_t, vp]
        L.deodr_hip_depth_residual.argtypes = [vp, i, vp, d, C.c_size_t, vp, vp, vp, vp, vp, C.c_size_t, vp]
        for f in ("rigid_transform", "rigid_transform_b", "project_points", "project_points_b", "silhouette_flags", "momentum_update", "fit_pose_project",
                  "fit_pose_project_b", "views_gradient_sum", "vertex_shade", "vertex_shade_b", "rigid_energy", "l2_loss", "depth_residual", "fit_front"):  # fmt: skip
            getattr(L, "deodr_hip_" + f).restype = i
        _bound = True
    return L


class RigidTransformFunc(torch.autograd.Function):
    """(vertices [V,3], unit quaternions [n,4] = (x, y, z, w), translations [n,3]) -> [n,V,3]: qrot(q, v) + t (deodr/tools.py:8-35)"""

    @staticmethod
    def forward(ctx, vertices, quaternions, translations):
        v, q, t = vertices.contiguous(), quaternions.contiguous(), translations.contiguous()
        n, V = q.shape[0], v.shape[0]
        out = torch.empty((n, V, 3), dtype=torch.float64, device=v.device)
        with torch.cuda.device(v.device):
            _check(_lib().deodr_hip_rigid_transform(_p(v), _p(q), _p(t), _p(out), V, n, _stream(v.device)))
        ctx.save_for_backward(v, q)
        return out

    @staticmethod
    def backward(ctx, out_b):
        v, q = ctx.saved_tensors
        n, V = q.shape[0], v.shape[0]
        out_b = out_b.contiguous()
        v_b = torch.empty_like(v)
        pose_b = torch.empty(7 * n, dtype=torch.float64, device=v.device)
        with torch.cuda.device(v.device):
            _check(_lib().deodr_hip_rigid_transform_b(_p(v), _p(q), _p(out_b), _p(v_b), _p(pose_b), V, n, _stream(v.device)))
        return v_b, pose_b[: 4 * n].view(n, 4), pose_b[4 * n :].view(n, 3)


def _check_cameras(n, extrinsic, intrinsic, distortion):
    """The kernels read `extrinsic[12 b + i]`, `intrinsic[9 b + i]`, `distortion[5 b + i]` of view b < n: one contiguous matrix PER VIEW
    (a camera shared by the views must have been expanded, as DeviceCamera does) -- anything else would be read out of bounds."""
    for name, a, shape in (("extrinsic", extrinsic, (3, 4)), ("intrinsic", intrinsic, (3, 3)), ("distortion", distortion, (5,))):
        if a is None:
            continue
        if tuple(a.shape) != (n,) + shape or not a.is_contiguous():
            raise ValueError(f"{name}: expected a contiguous [{n}, {', '.join(map(str, shape))}] tensor (one per view), got {tuple(a.shape)}")


class ProjectPointsFunc(torch.autograd.Function):
    """(points [n,V,3]; extrinsic [n,3,4], intrinsic [n,3,3], distortion [n,5] | None: constants) -> (ij [n,V,2], depths [n,V]);
    Camera.project_points / project_points_backward (deodr/differentiable_renderer.py:341-438)"""

    @staticmethod
    def forward(ctx, points, extrinsic, intrinsic, distortion):
        pts = points.contiguous()
        n, V = pts.shape[0], pts.shape[1]
        _check_cameras(n, extrinsic, intrinsic, distortion)
        ij = torch.empty((n, V, 2), dtype=torch.float64, device=pts.device)
        depths = torch.empty((n, V), dtype=torch.float64, device=pts.device)
        with torch.cuda.device(pts.device):
            _check(_lib().deodr_hip_project_points(_p(pts), _p(extrinsic), _p(intrinsic), _p(distortion), _p(ij), _p(depths), V, n, _stream(pts.device)))
        ctx.save_for_backward(pts, extrinsic, intrinsic)
        ctx.distortion = distortion
        return ij, depths

    @staticmethod
    def backward(ctx, ij_b, depths_b):
        pts, extrinsic, intrinsic = ctx.saved_tensors
        n, V = pts.shape[0], pts.shape[1]
        ij_b = ij_b.contiguous()
        depths_b = None if depths_b is None else depths_b.contiguous()
        pts_b = torch.empty_like(pts)
        with torch.cuda.device(pts.device):
            _check(_lib().deodr_hip_project_points_b(_p(pts), _p(extrinsic), _p(intrinsic), _p(ctx.distortion), _p(ij_b), _p(depths_b), _p(pts_b), V, n,
                                                     _stream(pts.device)))  # fmt: skip
        return pts_b, None, None, None


def silhouette_flags(ij, faces_u32, edge_faces_u32, clockwise, out=None):
    """ij [n,V,2] -> uint8 [n,T,3] (TriMeshAdjacencies.edge_on_silhouette, deodr/triangulated_mesh.py:153-166); no gradient"""
    ij = ij.detach().contiguous()
    n, V, T = ij.shape[0], ij.shape[1], faces_u32.shape[0]
    flags = torch.empty((n, T, 3), dtype=torch.uint8, device=ij.device) if out is None else out
    with torch.cuda.device(ij.device):
        _check(_lib().deodr_hip_silhouette_flags(_p(ij), _p(faces_u32), _p(edge_faces_u32), _p(flags), T, V, n, int(bool(clockwise)), _stream(ij.device)))
    return flags


def momentum_update(entries, inertia, damping, scratch=None, energy=None, data_energy=None, data_weight=1.0):
    """entries: [(x, speed, grad, grad2 | None, factor, step_max | None, normalize_rows[, grad_scale, grad_mean | None, mean_out | None])];
    x and speed are updated IN PLACE: s = (1 - damping)(inertia s + (1 - inertia) clamp(-factor (grad_scale (grad - grad_mean) + grad2))),
    x += s (deodr/mesh_fitter.py:153-190); mean_out [3] receives the column mean of the updated [.,3] tensor (needs ``scratch``);
    with ``energy`` [2] and ``data_energy`` [1]: energy[1] = data_weight * data_energy[0] + energy[0] on the way"""
    k = len(entries)
    assert 0 < k <= 8
    entries = [tuple(e) + (1.0, None, None)[len(e) - 7 :] for e in entries]
    ptrs = lambda j: (C.c_void_p * k)(*[None if e[j] is None else e[j].data_ptr() for e in entries])
    factor = (C.c_double * k)(*[float(e[4]) for e in entries])
    step_max = (C.c_double * k)(*[0.0 if e[5] is None else float(e[5]) for e in entries])
    count = (C.c_int * k)(*[int(e[0].numel()) for e in entries])
    rows = (C.c_int * k)(*[int(e[6]) for e in entries])
    scale = (C.c_double * k)(*[float(e[7]) for e in entries])
    dev = entries[0][0].device
    for e in entries:
        assert e[0].is_contiguous() and e[1].is_contiguous() and e[2].is_contiguous() and (e[3] is None or e[3].is_contiguous())
    with torch.cuda.device(dev):
        _check(_lib().deodr_hip_momentum_update(k, ptrs(0), ptrs(1), ptrs(2), ptrs(3), factor, step_max, count, rows, float(inertia), float(damping), scale,
                                                ptrs(8), ptrs(9), _p(energy), _p(data_energy), float(data_weight), _p(scratch),
                                                0 if scratch is None else scratch.numel(), _stream(dev)))  # fmt: skip


# ---- one fit iteration without an autograd graph (deodr_amd/csrc/dr_fititer.h) ----------------------------------------------------


def fit_scratch(V, n, device):
    """zero-filled scratch of the kernels below (their counter words stay zero between launches)"""
    return torch.zeros(int(_lib().deodr_hip_fit_scratch_bytes(int(V), int(n))), dtype=torch.uint8, device=device)


def _topology_scratch(topology, n):
    """one scratch per (topology, number of views), for the autograd wrappers (the kernels of one stream run one after the other)"""
    cache = topology.__dict__.setdefault("_fit_scratch", {})
    if n not in cache:
        cache[n] = fit_scratch(topology.nb_vertices, n, topology.device)
    return cache[n]


def fit_pose_project(vertices, vertices_mean, quaternions, translations, camera, posed, ij, depths, depth_colors=None, depth_scale=1.0):
    """centre ``vertices`` [V,3] in place (when a mean [3] is given), pose them with every view's quaternion (normalised inside) and
    translation, project them with every view's camera -> posed [n,V,3], ij [n,V,2], depths [n,V] (all written)"""
    n, V = posed.shape[0], posed.shape[1]
    _check_cameras(n, camera.extrinsic, camera.intrinsic, camera.distortion)
    with torch.cuda.device(posed.device):
        _check(_lib().deodr_hip_fit_pose_project(_p(vertices), _p(vertices_mean), _p(quaternions), _p(translations), _p(camera.extrinsic), _p(camera.intrinsic),
                                                 _p(camera.distortion), _p(posed), _p(ij), _p(depths), _p(depth_colors), float(depth_scale), V, n, _stream(posed.device)))  # fmt: skip


def fit_pose_project_b(vertices, quaternions, posed, camera, posed_b, ij_b, depths_b, vertices_b, out, scratch, depths_b_scale=1.0, colors_b=None,
                       colors_sum=None):
    """adjoint of :func:`fit_pose_project`: -> vertices_b [V,3]; out [3 + 7n] = mean of vertices_b over the vertices, quaternion adjoints
    [n,4] (raw quaternions), translation adjoints [n,3]; colors_sum [V,C] (optional) = colors_b [n,V,C] summed over the views"""
    n, V = posed.shape[0], posed.shape[1]
    _check_cameras(n, camera.extrinsic, camera.intrinsic, camera.distortion)
    with torch.cuda.device(posed.device):
        _check(_lib().deodr_hip_fit_pose_project_b(_p(vertices), _p(quaternions), _p(posed), _p(camera.extrinsic), _p(camera.intrinsic), _p(camera.distortion),
                                                   _p(posed_b), _p(ij_b), _p(depths_b), float(depths_b_scale), _p(vertices_b), _p(out), _p(scratch), scratch.numel(), V, n,
                                                   _p(colors_b), 0 if colors_b is None else int(colors_b.shape[-1]), _p(colors_sum), _stream(posed.device)))  # fmt: skip


def views_gradient_sum(posed, camera, ij_b, vertices_b, depths_b=None, depths_b_scale=1.0, colors_b=None, colors_sum=None, validate=True):
    """What the views of a multi-view fit share (mesh_fitter.py:518-527): vertices_b [V,3] (written) = the adjoint of every view's camera
    projection applied to ij_b [n,V,2] (and depths_b [n,V]), summed over the n views; colors_sum [V,C] (written, optional) = colors_b
    [n,V,C] summed over the views.  One wide launch -- the packed buffer a sharded fit all-reduces."""
    n, V = posed.shape[0], posed.shape[1]
    if validate:  # (validate=False: a caller that launches the same, already validated, tensors every step -- OverlappedViewsReduction)
        _validate_views_gradient_sum(posed, camera, ij_b, vertices_b, depths_b, colors_b, colors_sum)
    with torch.cuda.device(posed.device):
        _check(_lib().deodr_hip_views_gradient_sum(_p(posed), _p(camera.extrinsic), _p(camera.intrinsic), _p(camera.distortion), _p(ij_b), _p(depths_b),
                                                   float(depths_b_scale), _p(vertices_b), V, n, _p(colors_b), 0 if colors_b is None else int(colors_b.shape[-1]),
                                                   _p(colors_sum), _stream(posed.device)))  # fmt: skip


def _validate_views_gradient_sum(posed, camera, ij_b, vertices_b, depths_b, colors_b, colors_sum):
    n, V = posed.shape[0], posed.shape[1]
    _check_cameras(n, camera.extrinsic, camera.intrinsic, camera.distortion)
    # the kernel reads every pointer as contiguous float64 of exactly these shapes: anything else (a float32 vertex_dtype, a strided
    # view) would be read out of bounds and go into the collective as garbage without an error
    expected = [("posed", posed, (n, V, 3)), ("ij_b", ij_b, (n, V, 2)), ("vertices_b", vertices_b, (V, 3))]
    if depths_b is not None:
        expected.append(("depths_b", depths_b, (n, V)))
    if colors_b is not None or colors_sum is not None:
        if colors_b is None or colors_sum is None:
            raise ValueError("views_gradient_sum: colors_b and colors_sum go together")
        expected += [("colors_b", colors_b, (n, V, int(colors_b.shape[-1]))), ("colors_sum", colors_sum, (V, int(colors_b.shape[-1])))]
    for name, t, shape in expected:
        if not usable(t) or not t.is_contiguous() or tuple(t.shape) != shape or t.device != posed.device:
            raise ValueError(f"views_gradient_sum: {name} must be a contiguous float64 ROCm tensor of shape {shape} on {posed.device}, got "
                             f"{tuple(t.shape)} {t.dtype} {t.device}{'' if t.is_contiguous() else ' (not contiguous)'}")


def vertex_shade(posed, topology, light, ambient, color=None, luminosity=None, colors=None):
    """luminosity [n,V] = max(0, -normal . light) + ambient and / or colors [n,V,C] = color [C] * luminosity (written)"""
    n, V = posed.shape[0], posed.shape[1]
    with torch.cuda.device(posed.device):
        _check(_lib().deodr_hip_vertex_shade(_p(posed), _p(topology._faces_u32), _p(topology._vf_offsets), _p(topology._vf_corners), _p(light), _p(ambient),
                                             _p(color), 0 if color is None else color.numel(), _p(luminosity), _p(colors), V, n, int(topology.clockwise),
                                             _stream(posed.device)))  # fmt: skip


def fit_front(topology, n, scratch, ij=None, flags=None, posed=None, light=None, ambient=None, color=None, luminosity=None, colors=None, vertices=None,
              vertices_ref=None, cregu=0.0, gradient=None, energy=None):
    """:func:`silhouette_flags` (``flags`` given), :func:`vertex_shade` (``luminosity`` or ``colors`` given) and :func:`rigid_energy`
    (``gradient`` given; energy[0] only) in ONE launch -- the three do not depend on one another.  Same results bit for bit."""
    off, cols, vals = topology._m_csr if gradient is not None else (None, None, None)
    dev = scratch.device
    with torch.cuda.device(dev):
        _check(_lib().deodr_hip_fit_front(_p(ij), _p(topology._faces_u32), _p(topology._edge_faces), _p(flags), topology.nb_faces, _p(posed), _p(topology._vf_offsets),
                                          _p(topology._vf_corners), _p(light), _p(ambient), _p(color), 0 if color is None else color.numel(), _p(luminosity), _p(colors),
                                          _p(vertices), _p(vertices_ref), _p(off), _p(cols), _p(vals), float(cregu), _p(gradient), _p(energy), _p(scratch),
                                          scratch.numel(), topology.nb_vertices, int(n), int(topology.clockwise), _stream(dev)))  # fmt: skip


def vertex_shade_b(posed, topology, light, ambient, color, luminosity_b, colors_b, posed_b, out, scratch):
    """adjoint of :func:`vertex_shade`: -> posed_b [n,V,3] (written), out [4 + C] = light_b, ambient_b, color_b"""
    n, V = posed.shape[0], posed.shape[1]
    with torch.cuda.device(posed.device):
        _check(_lib().deodr_hip_vertex_shade_b(_p(posed), _p(topology._faces_u32), _p(topology._vf_offsets), _p(topology._vf_corners), _p(light), _p(ambient),
                                               _p(color), 0 if color is None else color.numel(), _p(luminosity_b), _p(colors_b), _p(posed_b), _p(out), _p(scratch),
                                               scratch.numel(), V, n, int(topology.clockwise), _stream(posed.device)))  # fmt: skip


def rigid_energy(vertices, vertices_ref, topology, cregu, gradient, energy, scratch, data_energy=None, data_weight=1.0):
    """energy[0] = 0.5 c d^T (L^T L) d, gradient [V,3] = c (L^T L) d, d = vertices - vertices_ref (both written); with ``data_energy`` [1]
    also energy[1] = data_weight * data_energy[0] + energy[0]"""
    off, cols, vals = topology._m_csr
    with torch.cuda.device(vertices.device):
        _check(_lib().deodr_hip_rigid_energy(_p(vertices), _p(vertices_ref), _p(off), _p(cols), _p(vals), float(cregu), _p(gradient), _p(energy), _p(data_energy),
                                             float(data_weight), _p(scratch), scratch.numel(), vertices.shape[0], _stream(vertices.device)))  # fmt: skip


def l2_loss(image, obs, out, scratch):
    """out[0] = sum (image - obs)^2, image and obs contiguous tensors of one pixel dtype (float32 / float64) and one shape"""
    assert image.dtype == obs.dtype and image.shape == obs.shape and image.is_contiguous() and obs.is_contiguous()
    with torch.cuda.device(image.device):
        _check(_lib().deodr_hip_l2_loss(_p(image), _p(obs), 1 if image.dtype == torch.float64 else 0, image.numel(), _p(out), _p(scratch), scratch.numel(),
                                        _stream(image.device)))  # fmt: skip


def depth_residual(image, obs, max_depth, depth, diff, image_b, loss, scratch):
    """the depth fitter's data term in one kernel: depth = clamp(image, 0, max_depth), diff = (depth - obs)^2 (both float64, written),
    image_b = d sum(diff) / d image in the pixel dtype, loss[0] = sum diff (deodr/mesh_fitter.py:108-123)"""
    assert obs.dtype == torch.float64 and depth.dtype == torch.float64 and diff.dtype == torch.float64 and image_b.dtype == image.dtype
    assert all(t.is_contiguous() and t.numel() == image.numel() for t in (image, obs, depth, diff, image_b))
    with torch.cuda.device(image.device):
        _check(_lib().deodr_hip_depth_residual(_p(image), 1 if image.dtype == torch.float64 else 0, _p(obs), float(max_depth), image.numel(), _p(depth), _p(diff),
                                               _p(image_b), _p(loss), _p(scratch), scratch.numel(), _stream(image.device)))  # fmt: skip


class VertexLuminosityFunc(torch.autograd.Function):
    """(posed [n,V,3], light [3], ambient []) -> luminosity [n,V]: vertex normals + max(0, -n.l) + ambient in one kernel, two for the adjoint
    (deodr/triangulated_mesh.py:113-151, deodr/differentiable_renderer.py:814-822)"""

    @staticmethod
    def forward(ctx, posed, light, ambient, topology):
        posed, light, ambient = posed.contiguous(), light.contiguous(), ambient.contiguous()
        lum = torch.empty(posed.shape[:2], dtype=torch.float64, device=posed.device)
        vertex_shade(posed, topology, light, ambient, luminosity=lum)
        ctx.save_for_backward(posed, light, ambient)
        ctx.topology = topology
        return lum

    @staticmethod
    def backward(ctx, lum_b):
        posed, light, ambient = ctx.saved_tensors
        posed_b = torch.empty_like(posed)
        out = torch.empty(4, dtype=torch.float64, device=posed.device)
        vertex_shade_b(posed, ctx.topology, light, ambient, None, lum_b.contiguous(), None, posed_b, out, _topology_scratch(ctx.topology, posed.shape[0]))
        return posed_b, out[:3], out[3].reshape(ambient.shape), None


class RigidEnergyFunc(torch.autograd.Function):
    """vertices [V,3] -> (energy, gradient [V,3]) of the as-rigid-as-possible energy (deodr/laplacian_rigid_energy.py:15-41) in one kernel;
    the energy is differentiable (its adjoint is the gradient the same launch produced)"""

    @staticmethod
    def forward(ctx, vertices, vertices_ref, topology, cregu):
        v = vertices.contiguous()
        grad, energy = torch.empty_like(v), torch.empty(1, dtype=torch.float64, device=v.device)
        rigid_energy(v, vertices_ref, topology, cregu, grad, energy, _topology_scratch(topology, 1))
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(grad)
        return energy[0], grad

    @staticmethod
    def backward(ctx, energy_b, _grad_b):
        (grad,) = ctx.saved_tensors
        return energy_b * grad, None, None, None
