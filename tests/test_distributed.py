"""world_size-2 gloo test (CPU) of the multi-GPU layer: view sharding and the single packed all-reduce."""

import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deodr_amd.distributed import PackedGradients, allreduce_shared_gradients, shard_views


def test_shard_views_partition():
    for n, w in [(8, 1), (8, 2), (8, 8), (7, 4), (3, 8), (64, 8)]:
        got = [i for r in range(w) for i in shard_views(n, r, w)]
        assert got == list(range(n))
        sizes = [len(shard_views(n, r, w)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    V, C, n_views = 11, 3, 5
    rs = np.random.RandomState(0)
    per_view_ij = rs.randn(n_views, V, 2)
    per_view_col = rs.randn(n_views, V, C)
    mine = list(shard_views(n_views, rank, world))
    ij_b = torch.as_tensor(per_view_ij[mine].sum(0)) if mine else torch.zeros(V, 2, dtype=torch.float64)
    col_b = torch.as_tensor(per_view_col[mine].sum(0)) if mine else torch.zeros(V, C, dtype=torch.float64)
    packed = PackedGradients([(V, 2), (V, C)])
    tot_ij, tot_col = allreduce_shared_gradients(packed, [ij_b, col_b])
    ok = np.allclose(tot_ij.numpy(), per_view_ij.sum(0)) and np.allclose(tot_col.numpy(), per_view_col.sum(0))
    with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
        f.write(str(int(ok)))
    dist.destroy_process_group()


def test_packed_allreduce_gloo_world2(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert [open(tmp_path / f"ok{r}").read() for r in range(2)] == ["1", "1"]


def test_pack_unpack_roundtrip():
    a, b = torch.arange(6.0, dtype=torch.float64).reshape(3, 2), torch.arange(4.0, dtype=torch.float64)
    p = PackedGradients([a.shape, b.shape])
    x, y = allreduce_shared_gradients(p, [a, b])  # no process group: plain pack / unpack
    assert torch.equal(x, a) and torch.equal(y, b)


# --------------------------------------------------------------------------------------------- textured scenes: texture_b and uv_b join the collective


def _textured_views(n_views, size=48):
    from deodr_amd import scenes

    angles = np.linspace(-0.3, 0.3, n_views)
    views = [scenes.sphere_scene(size=size, nu=12, n_rings=10, nb_colors=3, textured=True, texture_size=8, angle=float(a)) for a in angles]
    verts, _faces = scenes.bumpy_sphere(12, 10)
    cams = [scenes.fit_camera(size, size, 60.0, verts, scenes.rotx(0.37) @ scenes.roty(0.23 + float(a))) for a in angles]
    for cam, v in zip(cams, views):
        assert np.abs(scenes.project(cam, verts)[0] - v.ij).max() < 1e-9  # the cameras the views were made with
    return views, verts, cams


def _camera_adjoint(cam, verts, ij_b):
    """project_points_backward of a pinhole camera (deodr/differentiable_renderer.py:397-438) by autograd on CPU tensors"""
    p = torch.tensor(verts, requires_grad=True)
    q = p @ torch.tensor(cam.extrinsic[:, :3]).T + torch.tensor(cam.extrinsic[:, 3])
    ij = (q[:, :2] / q[:, 2:3]) @ torch.tensor(cam.intrinsic[:2, :2]).T + torch.tensor(cam.intrinsic[:2, 2])
    (g,) = torch.autograd.grad(ij, p, torch.as_tensor(ij_b))
    return g


def _textured_worker(rank, world, port, out_dir):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fake_hip
    from hip_util import device_scene
    from deodr_amd.distributed import SharedGradientBuffer
    from deodr_amd.hip_renderer import HipRasterizer
    from oracle import api

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_views = 3
    views, verts, cams = _textured_views(n_views)
    H, W, Cc = views[0].height, views[0].width, views[0].nb_colors
    obs = np.random.RandomState(4).rand(n_views, H, W, Cc)
    mine = list(shard_views(n_views, rank, world))
    ref, fixed = api.ref() or api.port(), api.ref(fixed=True) or api.port(fixed=True)
    V, n_uv = verts.shape[0], views[0].uv.shape[0]
    buf = SharedGradientBuffer(V, Cc, n_uv, views[0].texture.shape, torch.float64, "cpu")
    with fake_hip.emulate(ref, fixed):  # (the host layer on CPU tensors, the checker behind the C ABI: this rank's views in one call)
        ds = device_scene([views[i] for i in mine], torch.float64)
        grads = ds.zero_grads()
        grads["texture_b"], grads["uv_b"] = buf.texture_b, buf.uv_b  # the library accumulates its views' taps straight into the packed buffer
        _, _, g = HipRasterizer.for_scene(ds).render_fit(ds, torch.as_tensor(obs[mine]), 1.0, grads=grads, clear_grads=True)
    for j, i in enumerate(mine):
        buf.vertices_b += _camera_adjoint(cams[i], verts, g["ij_b"][j].numpy())
    buf.colors_b += g["colors_b"].sum(0)
    buf.all_reduce()
    # the same sums from one call of the checker per view, all views, on this process
    want = SharedGradientBuffer(V, Cc, n_uv, views[0].texture.shape, torch.float64, "cpu")
    for i, s in enumerate(views):
        image, z = ref.render(s, 1.0)
        gr = fixed.grads(s, 1.0, image, z, 2 * (image - obs[i]))  # (texture_b accumulates in the repaired build: defect D1)
        want.vertices_b += _camera_adjoint(cams[i], verts, gr["ij_b"])
        want.colors_b += torch.as_tensor(gr["colors_b"])
        want.uv_b += torch.as_tensor(gr["uv_b"])
        want.texture_b += torch.as_tensor(gr["texture_b"])
    err = float((buf.flat - want.flat).abs().max() / want.flat.abs().max())
    ok = err < 1e-12 and float(want.texture_b.abs().max()) > 0 and float(want.uv_b.abs().max()) > 0 and len(mine) == (2 if rank == 0 else 1)
    with open(os.path.join(out_dir, f"ok{rank}"), "w") as f:
        f.write(f"{int(ok)} {err:.3e}")
    dist.destroy_process_group()


def test_textured_shared_gradient_gloo_world2(tmp_path):
    """Three views of a textured mesh over two ranks: the packed buffer [texture_b | vertices_b | colors_b | uv_b] after ONE all-reduce
    equals the sum over all views of per-view calls of the checker (deodr/mesh_fitter.py:518-527 with the texture the reference's
    struct Scene shares between the views, H.h:56-90, 563-631)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_textured_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [open(tmp_path / f"ok{r}").read() for r in range(2)]
    assert all(g.startswith("1 ") for g in got), got
