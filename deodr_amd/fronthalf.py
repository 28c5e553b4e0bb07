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
        L.deodr_hip_momentum_update.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, vp, d, d, vp]
        for f in ("rigid_transform", "rigid_transform_b", "project_points", "project_points_b", "silhouette_flags", "momentum_update"):
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


class ProjectPointsFunc(torch.autograd.Function):
    """(points [n,V,3]; extrinsic [n,3,4], intrinsic [n,3,3], distortion [n,5] | None: constants) -> (ij [n,V,2], depths [n,V]);
    Camera.project_points / project_points_backward (deodr/differentiable_renderer.py:341-438)"""

    @staticmethod
    def forward(ctx, points, extrinsic, intrinsic, distortion):
        pts = points.contiguous()
        n, V = pts.shape[0], pts.shape[1]
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


def silhouette_flags(ij, faces_u32, edge_faces_u32, clockwise):
    """ij [n,V,2] -> uint8 [n,T,3] (TriMeshAdjacencies.edge_on_silhouette, deodr/triangulated_mesh.py:153-166); no gradient"""
    ij = ij.detach().contiguous()
    n, V, T = ij.shape[0], ij.shape[1], faces_u32.shape[0]
    flags = torch.empty((n, T, 3), dtype=torch.uint8, device=ij.device)
    with torch.cuda.device(ij.device):
        _check(_lib().deodr_hip_silhouette_flags(_p(ij), _p(faces_u32), _p(edge_faces_u32), _p(flags), T, V, n, int(bool(clockwise)), _stream(ij.device)))
    return flags


def momentum_update(entries, inertia, damping):
    """entries: [(x, speed, grad, grad2 | None, factor, step_max | None, normalize_rows)]; x and speed are updated IN PLACE
    (deodr/mesh_fitter.py:153-190: s = (1 - damping)(inertia s + (1 - inertia) clamp(-factor (grad + grad2))), x += s)"""
    k = len(entries)
    assert 0 < k <= 8
    ptrs = lambda j: (C.c_void_p * k)(*[None if e[j] is None else e[j].data_ptr() for e in entries])
    factor = (C.c_double * k)(*[float(e[4]) for e in entries])
    step_max = (C.c_double * k)(*[0.0 if e[5] is None else float(e[5]) for e in entries])
    count = (C.c_int * k)(*[int(e[0].numel()) for e in entries])
    rows = (C.c_int * k)(*[int(e[6]) for e in entries])
    dev = entries[0][0].device
    for e in entries:
        assert e[0].is_contiguous() and e[1].is_contiguous() and e[2].is_contiguous() and (e[3] is None or e[3].is_contiguous())
    with torch.cuda.device(dev):
        _check(_lib().deodr_hip_momentum_update(k, ptrs(0), ptrs(1), ptrs(2), ptrs(3), factor, step_max, count, rows, float(inertia), float(damping),
                                                _stream(dev)))  # fmt: skip
