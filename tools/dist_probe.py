"""Where does the reduction of the shared gradients cost the render loop its time?  One rank, the bench workload; variants:
   0 render only | 1 + event record on the render stream | 2 + the reduction kernels on the communication stream |
   3 + all_reduce (async) | 4 all of it on the render stream (no overlap) | 5 as 3 with the library's one-kernel reduction |
   6 the one-kernel reduction on the render stream, the all_reduce on the communication stream | 7 as 5 without the all_reduce
   python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 tools/dist_probe.py"""
import os, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deodr_amd import scenes
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

dist.init_process_group("nccl", rank=0, world_size=1) if "RANK" in os.environ else None
B, S = 8, 1024
dev = torch.device("cuda:0")
views = [scenes.sphere_scene(size=S, angle=float(a)) for a in np.linspace(-0.5, 0.5, B)]
s0 = views[0]
stack = lambda n: np.stack([np.asarray(getattr(v, n)) for v in views])
ds = DeviceScene(s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"), stack("edgeflags"), S, S,
                 texture=None, background_color=s0.background_color, clockwise=s0.clockwise, vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev)
r = HipRasterizer.for_scene(ds)
V, C = ds.ij.shape[1], ds.nb_colors
obs = torch.rand((B, S, S, C), dtype=torch.float32, device=dev)
image, z = torch.empty((B, S, S, C), dtype=torch.float32, device=dev), torch.empty((B, S, S), dtype=torch.float32, device=dev)
grads = [ds.zero_grads(), ds.zero_grads()]
shared = [torch.zeros(V * (3 + C), dtype=torch.float64, device=dev) for _ in range(2)]
jac = torch.rand((B, V, 2, 3), dtype=torch.float64, device=dev)
comm = torch.cuda.Stream(device=dev)
from deodr_amd import fronthalf
from deodr_amd.scene3d import DeviceCamera
verts, _f = scenes.bumpy_sphere(100, 100)
cams = [scenes.fit_camera(S, S, 60.0, verts, scenes.rotx(0.37) @ scenes.roty(0.23 + float(a))) for a in np.linspace(-0.5, 0.5, B)]
camera = DeviceCamera(np.stack([c.extrinsic for c in cams]), np.stack([c.intrinsic for c in cams]), S, S, None, dev)
wv = torch.as_tensor(np.ascontiguousarray(verts, dtype=np.float64), device=dev)
posed, ident = wv[None].expand(B, -1, -1).contiguous(), torch.tensor([[0.0, 0.0, 0.0, 1.0]] * B, dtype=torch.float64, device=dev)
pose_out, scratch = torch.zeros(3 + 7 * B, dtype=torch.float64, device=dev), fronthalf.fit_scratch(V, B, dev)
r.render(ds, 1.0, out=(image, z), check_overflow=True)


def run(variant, steps=200):
    reads_done, pending, it = [None, None], [None, None], 0

    def reduce_into(i, g):
        if variant >= 5:
            fronthalf.fit_pose_project_b(wv, ident, posed, camera, None, g["ij_b"], None, shared[i][: 3 * V].view(V, 3), pose_out, scratch, colors_b=g["colors_b"],
                                         colors_sum=shared[i][3 * V :].view(V, C))
            return
        torch.sum(g["ij_b"][..., None] * jac, dim=(0, 2), out=shared[i][: 3 * V].view(V, 3))
        torch.sum(g["colors_b"], dim=0, out=shared[i][3 * V :].view(V, C))

    def step():
        nonlocal it
        i = it % 2
        it += 1
        g = grads[i]
        if variant in (2, 3, 5, 7) and reads_done[i] is not None:
            reads_done[i].synchronize()
        r.render_fit(ds, obs, 1.0, grads=g, out=(image, z), check_overflow=False, clear_grads=True)
        if variant == 0:
            return
        if variant == 4:
            reduce_into(i, g)
            dist.all_reduce(shared[i])
            return
        if variant == 6:
            reduce_into(i, g)
        ev = torch.cuda.Event()
        ev.record()
        if variant == 1:
            return
        if variant == 6:
            with torch.cuda.stream(comm):
                comm.wait_event(ev)
                pending[i] = dist.all_reduce(shared[i], async_op=True)
            return
        with torch.cuda.stream(comm):
            comm.wait_event(ev)
            if pending[i] is not None:
                pending[i].wait()
            reduce_into(i, g)
            reads_done[i] = torch.cuda.Event()
            reads_done[i].record()
            if variant in (3, 5):
                pending[i] = dist.all_reduce(shared[i], async_op=True)

    for _ in range(60):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    for pnd in pending:
        if pnd is not None:
            pnd.wait()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


names = {5: "one-kernel reduction on comm + async all_reduce", 6: "one-kernel reduction on the render stream + async all_reduce", 7: "one-kernel reduction on comm, no all_reduce", 0: "render only", 1: "+ event record", 2: "+ reduction kernels on the comm stream", 3: "+ async all_reduce", 4: "all on the render stream"}
for rep in range(2):
    for v in ([0, 1, 3, 5, 6, 7] if dist.is_initialized() else [0, 1, 2]):
        print(f"variant {v} ({names[v]}): {run(v) * 1e3:.4f} ms / step")
