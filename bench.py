"""bench.py -- Mpixels/s forward+backward of the HIP rasterizer on BASELINE.json's metric configuration.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (config.workload): BASELINE configs[2], the configuration the metric is quoted on -- 1024x1024, 20 000-triangle
bumpy sphere, C = 4 channels (RGB + depth), sigma = 1, ~33 % coverage, ~650 drawn silhouette edges -- rendered as
`--views` (default 8) poses per GPU per step.  One step = renderScene + renderScene_B for the loss sum (image - obs)^2 over
the whole view batch -- by default through the one-call fit step (deodr_hip_render_scene_fit: same outputs, the forward raster
back-propagates through the tiles without silhouette edges itself), with --two-pass as two calls -- with all inputs resident
in HBM (+, for N > 1, ONE RCCL all-reduce of the shared-parameter gradient, the reduction the reference's
multi-view fitter does on the host at deodr/mesh_fitter.py:518-527).  Views shard across ranks with no data-path
collective, per-GPU work is fixed as N grows ("weak").

The JSON line carries two extra objects:
  roofline      dominant kernel (largest summed time): achieved = algorithmic bytes per launch (float32
                buffers) / average launch duration measured with hipEvents on the launch stream inside the timed region
  cpu_baseline  the reference's own CPU path (oracle/_ref = unmodified header, g++ -O2, one thread) on the host of the GPU
                box, timed on a bounded sample of the same workload (rank 0, N = 1 only)
"""

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
KERNELS = ["setup_bin_kernel", "raster_fwd_kernel", "raster_bwd_kernel", "finalize_kernel"]


def algorithmic_bytes(H, W, C, T, V, n_views, fused):
    """Per LAUNCH algorithmic HBM bytes of each kernel, float32 buffers (untextured, colour background).

    B_fwd = 4 [H W (C+1) + V (2+1+C) + 3T]      B_bwd = 4 [H W C + H W + V (2+1+C) + 3T + V (2+C)]
    split by the kernel that has to move them.  In the fused fit step the forward raster also does the frame-sized part of
    the adjoint (it reads the observation where the two-pass adjoint reads image_b), so it is charged both frame terms and
    the adjoint's edge kernel, which only revisits the ~3 % of tiles that hold silhouette edges, none."""
    px = H * W
    per_view = {
        "setup_bin_kernel": 4 * (V * (3 + C) + 3 * T),
        "raster_fwd_kernel": 4 * px * (C + 1) * (2 if fused else 1),
        "raster_bwd_kernel": 0 if fused else 4 * px * (C + 1),
        "finalize_kernel": 4 * (V * (3 + C) + 3 * T + V * (2 + C)),
    }
    return {k: v * n_views for k, v in per_view.items()}


def cpu_baseline(scene, image_b, budget_s=12.0):
    """The reference CPU path on this host: median-free throughput over a bounded sample (about `budget_s` of CPU work)."""
    from oracle import api

    ref = api.ref()
    kind = "reference"
    if ref is None:
        ref, kind = api.port(), "port"
    H, W, Cc = scene.height, scene.width, scene.nb_colors
    image, z = np.zeros((H, W, Cc)), np.zeros((H, W))
    ref.renderSceneCpp(scene, 1.0, image, z)  # warm-up (page-faults the buffers)
    reps, t_used = 0, 0.0
    while t_used < budget_s and reps < 2000:
        scene.clear_gradients()
        t0 = time.perf_counter()
        ref.renderSceneCpp(scene, 1.0, image, z)
        ref.renderSceneBCpp(scene, 1.0, image, z, image_b.copy())  # the adjoint un-antialiases `image` in place, as the reference
        t_used += time.perf_counter() - t0
        reps += 1
    return {
        "value": reps * H * W / t_used / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": kind,
        "sample": f"{reps} x (renderScene + renderScene_B) of ONE view of the same workload ({t_used:.1f} s, "
                  f"oracle/_ref = unmodified reference header, g++ -O2, single thread; host has {os.cpu_count()} logical cores)",
    }  # fmt: skip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--views", type=int, default=8, help="views rendered per GPU per step")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--sigma", type=float, default=1.0, help="edge-overdraw width (the metric configuration uses 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--time-every", type=int, default=4, help="steps between two steps whose kernels are timed with hipEvents")
    ap.add_argument("--two-pass", action="store_true", help="render and render_backward as two calls (default: the fused fit step, same outputs)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even for one rank (exercises the RCCL path)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL writes a banner to the C stdout when the communicator is created; send it to stderr so that the JSON result
        # stays the only (and last) line on stdout
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="nccl", device_id=dev)  # RCCL over xGMI
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)  # creates the communicator now
            torch.cuda.synchronize()
        finally:
            C.CDLL(None).fflush(None)
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    import __graft_entry__ as g

    g.build_hip()
    from deodr_amd import scenes
    from deodr_amd import hip_renderer as hr
    from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

    # ---- synthetic inputs: `views` poses of the same mesh per rank (distinct poses on every rank)
    B, S = args.views, args.size
    poses = np.linspace(-0.5, 0.5, B * world)[rank * B : (rank + 1) * B]
    views = [scenes.sphere_scene(size=S, angle=float(a)) for a in poses]
    s0 = views[0]
    stack = lambda name: np.stack([np.asarray(getattr(v, name)) for v in views])
    ds = DeviceScene(
        s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"),
        stack("edgeflags"), S, S, texture=None, background_color=s0.background_color, clockwise=s0.clockwise,
        vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev,
    )  # fmt: skip
    T, V, Cc = ds.nb_triangles, int(ds.depths.shape[1]), ds.nb_colors
    rs = np.random.RandomState(3)
    obs = torch.as_tensor(rs.rand(S, S, Cc).astype(np.float32), device=dev)
    r = HipRasterizer.for_scene(ds)
    image = torch.empty((B, S, S, Cc), dtype=torch.float32, device=dev)
    z = torch.empty((B, S, S), dtype=torch.float32, device=dev)
    grads = ds.zero_grads()
    shared = torch.zeros(V * (2 + Cc), dtype=torch.float64, device=dev)  # packed shared-parameter gradient

    obs_views = obs.expand(B, S, S, Cc).contiguous()  # one observation per view (here the same synthetic image)
    pending = [None]

    def step():
        if args.two_pass:
            grads["ij_b"].zero_()
            grads["colors_b"].zero_()
            r.render(ds, args.sigma, out=(image, z), check_overflow=False)
            # adjoint of L = sum (image - obs)^2: dL/dimage = 2 (image - obs) is formed inside the adjoint kernel (residual mode)
            r.render_backward(ds, residual_obs=obs_views, grads=grads)
        else:
            # same outputs in one call: the forward raster back-propagates through the tiles without silhouette edges itself
            # (and zeroes the gradient arrays of the previous step on the way)
            r.render_fit(ds, obs_views, args.sigma, grads=grads, out=(image, z), check_overflow=False, clear_grads=True)
        if dist is not None:
            if pending[0] is not None:
                pending[0].wait()  # the previous step's all-reduce overlapped this step's rendering
            torch.cat((grads["ij_b"].sum(0).reshape(-1), grads["colors_b"].sum(0).reshape(-1)), out=shared)
            pending[0] = dist.all_reduce(shared, async_op=True)

    # first call checks the spill pool once (synchronises), then nothing in the loop does
    r.render(ds, args.sigma, out=(image, z), check_overflow=True)
    for _ in range(args.warmup):
        step()

    def barrier():
        if dist is not None:
            if pending[0] is not None:
                pending[0].wait()
                pending[0] = None
            dist.barrier()
        torch.cuda.synchronize()

    # per-kernel hipEvents on every 4th step of the timed region (an event pair takes ~3 us of stream time: timing all
    # five launches of every step would add ~10 % to the step being measured)
    hr.lib().deodr_hip_profile_enable(0 if os.environ.get("DEODR_BENCH_NO_EVENTS") else args.time_every)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    hr.lib().deodr_hip_profile_enable(0)
    ms_sum = (C.c_double * 4)()
    launches = (C.c_ulonglong * 4)()
    hr.lib().deodr_hip_profile_read(ms_sum, launches)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # spill pool never overflowed during the run (deferred check, outside the timed region)
    sc = ds.c_struct()
    over, need = C.c_int(0), C.c_ulonglong(0)
    hr.lib().deodr_hip_workspace_status(C.byref(sc), C.c_void_p(r.workspace.data_ptr()), r.nbytes, None, C.byref(over), C.byref(need))
    assert not over.value, "spill pool overflowed during the benchmark"

    if rank == 0:
        px = world * B * S * S * args.steps
        alg = algorithmic_bytes(S, S, Cc, T, V, B, fused=not args.two_pass)
        per_kernel = {}
        for i, k in enumerate(KERNELS):
            n = max(int(launches[i]), 1)
            avg_ms = ms_sum[i] / n
            per_kernel[k] = {"avg_ms": avg_ms, "launches": int(launches[i]), "alg_bytes": alg[k],
                             "GBps": alg[k] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 and alg[k] else None}  # fmt: skip
        dom = max(KERNELS, key=lambda k: per_kernel[k]["avg_ms"])
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):  # HBM bytes per launch from the last PMC run (tools/profile_round.sh), gfx950 correction applied
            traffic = (json.load(open(tpath)).get(dom) or {}).get("bytes_per_launch")
        kernel_ms = sum(v["avg_ms"] for v in per_kernel.values())
        out = {
            "metric": "Mpixels/s forward+backward, 1024^2 20k-tri scene", "value": px / dt / 1e6, "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: {S}x{S}, {T}-triangle bumpy sphere, C={Cc} (RGB+depth), sigma=1, "
                                   f"{B} views per GPU per step, float32 pixel buffers / float64 vertex arrays, all-double arithmetic",
                       "step": "renderScene + renderScene_B (two calls)" if args.two_pass else "deodr_hip_render_scene_fit (forward + adjoint of sum (image - obs)^2, one call)",
                       "views_per_gpu": B, "global_views": B * world,
                       "parallelism": f"views sharded {B}/GPU" + (", 1 RCCL all-reduce of the shared gradient per step" if world > 1 else "")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": per_kernel[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (per_kernel[dom]["GBps"] or 0) / HBM_PEAK_GBS, "traffic": traffic,
                         "whole_step_alg_GBps": sum(alg.values()) / (dt / args.steps) / 1e9,
                         "kernel_time_fraction_of_step": kernel_ms / (dt / args.steps * 1e3), "per_kernel": per_kernel},
        }  # fmt: skip
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(views[0], 2 * (image[0] - obs).cpu().numpy().astype(np.float64))
        result_line = json.dumps(out)
    else:
        result_line = None
    if dist is not None:
        dist.destroy_process_group()
    if result_line is not None:  # the ONE JSON line, last thing on stdout
        sys.stdout.flush()
        print(result_line, flush=True)


if __name__ == "__main__":
    main()
