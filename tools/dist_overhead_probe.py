"""What the shared-gradient reduction of a sharded fit costs ONE rank per step (bench.py --force-dist), piece by piece.  One process,
a one-rank RCCL group, the bench workload (configs[2], 8 views).  Run on the GPU box:
    python tools/dist_overhead_probe.py [--steps 200] [--modes plain,cur,...]
modes
  plain      the fit step alone
  record     + an event recorded on the render stream after every step (nothing waits for it)
  cur        bench.py's pipeline: event -> communication stream: camera adjoint + packing kernel, all_reduce(async_op=True)
  cur_sync   the same with all_reduce(async_op=False) issued under the communication stream
  kernel2    event -> communication stream: the camera adjoint kernel only (no collective)
  serial     camera adjoint kernel + all_reduce on the render stream itself (no second stream, no overlap)
  serial_k   camera adjoint kernel on the render stream, nothing else
  wide_k     the wide views_gradient_sum kernel on the render stream, nothing else
  wide_comm  event -> communication stream: the wide kernel + all_reduce(async_op=False)
  comm_empty / comm_wide / comm_ar   wide_comm with nothing / the kernel only / the collective only on the communication stream
  wide_main  the wide kernel on the render stream, event, all_reduce on the communication stream
  flagged    the fit step stores a step-done flag (DeodrHipFitOptions::done_flag: finalize_kernel's last wavefront); the communication stream
             waits for it with deodr_hip_wait_flag, then the wide kernel + all_reduce there: NO event on the render stream
  flagged_sleepN   flagged, and the communication stream idles N us more before the reduction kernel
  flag_only  the fit step stores the flag, nothing waits for it (what the flag costs finalize_kernel)
  (measured once and removed from the library, profiles/r04v_*: a hipEvent recorded through finalize_kernel's completion signal --
   hipExtLaunchKernelGGL's stopEvent -- instead of hipEventRecord: 1.7 us alone, but 25 us once another stream waits for it)
  flag       camera adjoint kernel on the render stream, then hipStreamWriteValue32 there; the communication stream waits with
             hipStreamWaitValue32 and runs the collective (no hipEvent on the render stream)
"""
import sys, os, time, ctypes
import numpy as np, torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deodr_amd import scenes, fronthalf, hip_renderer as hr
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer
from deodr_amd.scene3d import DeviceCamera

arg = lambda name, default: type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default
B, S, steps = arg("--views", 8), 1024, arg("--steps", 200)
modes = arg("--modes", "plain,record,cur_sync,wide_k,comm_empty,wide_comm,flag_only,flagged,plain,flagged,wide_comm").split(",")
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
dist.init_process_group(backend="nccl", device_id=dev)
dist.all_reduce(torch.zeros(1, device=dev))
torch.cuda.synchronize()

poses = np.linspace(-0.5, 0.5, B)
views = [scenes.sphere_scene(size=S, angle=float(a)) for a in poses]
s0 = views[0]
stack = lambda n: np.stack([np.asarray(getattr(v, n)) for v in views])
ds = DeviceScene(s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"),
                 stack("edgeflags"), S, S, texture=None, background_color=s0.background_color, clockwise=s0.clockwise,
                 vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev)  # fmt: skip
r = HipRasterizer.for_scene(ds)
Cc, V = ds.nb_colors, int(ds.depths.shape[1])
obs = torch.rand((B, S, S, Cc), dtype=torch.float32, device=dev)
image = torch.empty((B, S, S, Cc), dtype=torch.float32, device=dev)
z = torch.empty((B, S, S), dtype=torch.float32, device=dev)
PRIO = arg("--priority", 0)  # of the communication stream: -1 = high (a hardware queue of its own class)
NBUF = arg("--nbuf", 2)  # sets of gradient buffers: the host runs at most NBUF steps ahead of the reduction
grads_pp = [ds.zero_grads() for _ in range(NBUF)]
shared_pp = [torch.zeros(V * (3 + Cc), dtype=torch.float64, device=dev) for _ in range(NBUF)]
verts, _f = scenes.bumpy_sphere(100, 100)
cams = [scenes.fit_camera(S, S, 60.0, verts, scenes.rotx(0.37) @ scenes.roty(0.23 + float(a))) for a in poses]
camera = DeviceCamera(np.stack([c.extrinsic for c in cams]), np.stack([c.intrinsic for c in cams]), S, S, None, dev)
wv = torch.as_tensor(np.ascontiguousarray(verts, dtype=np.float64), device=dev)
posed = wv[None].expand(B, -1, -1).contiguous()
ident = torch.tensor([[0.0, 0.0, 0.0, 1.0]] * B, dtype=torch.float64, device=dev)
pose_out = torch.zeros(3 + 7 * B, dtype=torch.float64, device=dev)
scratch = fronthalf.fit_scratch(V, B, dev)
r.render(ds, 1.0, out=(image, z), check_overflow=True)

hip = ctypes.CDLL("libamdhip64.so")
hip.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
hip.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
flag_ptr = ctypes.c_void_p()
HIP_MALLOC_SIGNAL_MEMORY = 0x2
have_flag = hip.hipExtMallocWithFlags(ctypes.byref(flag_ptr), 8, HIP_MALLOC_SIGNAL_MEMORY) == 0
if have_flag:
    hip.hipMemset(flag_ptr, 0, 8)
torch.cuda.synchronize()


def adjoint(i):
    g = grads_pp[i]
    fronthalf.fit_pose_project_b(wv, ident, posed, camera, None, g["ij_b"], None, shared_pp[i][: 3 * V].view(V, 3), pose_out, scratch,
                                 colors_b=g["colors_b"], colors_sum=shared_pp[i][3 * V :].view(V, Cc))  # fmt: skip


def wide(i):
    g = grads_pp[i]
    fronthalf.views_gradient_sum(posed, camera, g["ij_b"], shared_pp[i][: 3 * V].view(V, 3), colors_b=g["colors_b"], colors_sum=shared_pp[i][3 * V :].view(V, Cc))  # fmt: skip


L = hr.lib()
step_flag = torch.zeros(1, dtype=torch.int32, device=dev)
seq = [0]  # step numbers of the flag: increasing over the whole run
wait_status = torch.zeros(1, dtype=torch.int32, device=dev)
# the wide kernel against the pose adjoint with the identity pose
r.render_fit(ds, obs, 1.0, grads=grads_pp[0], out=(image, z), check_overflow=False, clear_grads=True)
adjoint(0)
ref = shared_pp[0].clone()
shared_pp[0].zero_()
wide(0)
torch.cuda.synchronize()
print("wide kernel vs fit_pose_project_b (identity pose): max abs diff", float((shared_pp[0] - ref).abs().max()), "of", float(ref.abs().max()), flush=True)


def run(mode):
    comm = torch.cuda.Stream(device=dev, priority=PRIO)  # (streams of one priority share a few hardware queues, dealt round-robin)
    reads_done, pending, it = [None] * NBUF, [None] * NBUF, [0]

    def step():
        i = it[0] % NBUF
        it[0] += 1
        if reads_done[i] is not None:
            reads_done[i].synchronize()
        if mode in ("flagged", "flag_only") or mode.startswith("flagged_sleep"):
            seq[0] += 1
            r.render_fit(ds, obs, 1.0, grads=grads_pp[i], out=(image, z), check_overflow=False, clear_grads=True, done_flag=(step_flag, seq[0]))
            if mode != "flag_only":
                with torch.cuda.stream(comm):
                    hr.wait_flag(step_flag, seq[0], status=wait_status, timeout=1.0)
                    if mode.startswith("flagged_sleep"):  # flagged_sleepN: the reduction starts N us later (beside the forward raster, not beside set-up)
                        torch.cuda._sleep(int(float(mode[len("flagged_sleep"):]) * 2100))
                    wide(i)
                    dist.all_reduce(shared_pp[i])
                    reads_done[i] = torch.cuda.Event()
                    reads_done[i].record()
            return
        r.render_fit(ds, obs, 1.0, grads=grads_pp[i], out=(image, z), check_overflow=False, clear_grads=True)
        if mode == "plain":
            return
        if mode == "wide_k":
            wide(i)
            return
        if mode == "wide_main":
            wide(i)
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(comm):
                comm.wait_event(ev)
                dist.all_reduce(shared_pp[i])
                reads_done[i] = torch.cuda.Event()
                reads_done[i].record()
            return
        if mode in ("wide_comm", "comm_empty", "comm_wide", "comm_ar"):
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(comm):
                comm.wait_event(ev)
                if mode in ("wide_comm", "comm_wide"):
                    wide(i)
                if mode in ("wide_comm", "comm_ar"):
                    dist.all_reduce(shared_pp[i])
                reads_done[i] = torch.cuda.Event()
                reads_done[i].record()
            return
        if mode == "record":
            e = torch.cuda.Event()
            e.record()
            return
        if mode in ("serial", "serial_k"):
            adjoint(i)
            if mode == "serial":
                dist.all_reduce(shared_pp[i])
            return
        if mode == "flag":
            adjoint(i)
            seq[0] += 1
            main = torch.cuda.current_stream(dev)
            assert hip.hipStreamWriteValue32(ctypes.c_void_p(main.cuda_stream), flag_ptr, seq[0], 0) == 0
            with torch.cuda.stream(comm):
                assert hip.hipStreamWaitValue32(ctypes.c_void_p(comm.cuda_stream), flag_ptr, seq[0], 0, 0xFFFFFFFF) == 0  # flags 0: >=
                dist.all_reduce(shared_pp[i])
                reads_done[i] = torch.cuda.Event()
                reads_done[i].record()
            return
        rendered = torch.cuda.Event()
        rendered.record()
        with torch.cuda.stream(comm):
            comm.wait_event(rendered)
            if pending[i] is not None:
                pending[i].wait()
            adjoint(i)
            reads_done[i] = torch.cuda.Event()
            reads_done[i].record()
            if mode == "cur":
                pending[i] = dist.all_reduce(shared_pp[i], async_op=True)
            elif mode == "cur_sync":
                dist.all_reduce(shared_pp[i])

    def drain():
        for k in range(NBUF):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
        comm.synchronize()
        torch.cuda.synchronize()

    for _ in range(60):
        step()
    best, host = 1e9, 0.0
    for _rep in range(3):
        drain()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        t1 = time.perf_counter()
        drain()
        t2 = time.perf_counter()
        if (t2 - t0) / steps < best:
            best, host = (t2 - t0) / steps, (t1 - t0) / steps
    if mode.startswith("flagged"):  # the reduction of the last step against the same thing done synchronously
        last = (it[0] - 1) % NBUF
        got = shared_pp[last].clone()
        wide(last)
        torch.cuda.synchronize()
        print(f"   flagged: wait status {int(wait_status.item())} (0 = no wait timed out), flag {int(step_flag.item())} of {seq[0]}, "
              f"reduced buffer equals a synchronous one: {bool(torch.equal(got, shared_pp[last]))}, |max| {float(got.abs().max()):.4g}", flush=True)
    print(f"{mode:9s} {best * 1e3:.4f} ms / step   (host loop {host * 1e3:.4f} ms / step)", flush=True)
    return best


base = None
for m in modes:
    if m == "flag" and not have_flag:
        print("flag      hipExtMallocWithFlags(hipMallocSignalMemory) failed: skipped")
        continue
    try:
        t = run(m)
    except Exception as e:  # noqa: BLE001
        print(f"{m:9s} failed: {e!r}")
        continue
    if m == "plain" and base is None:
        base = t
dist.destroy_process_group()
