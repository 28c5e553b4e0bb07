"""Time of ONE ITERATION of the device fitters (the reason the front half lives on the device), its rasterizer
share (hipEvent times of the library's kernels inside the step) and, with `--count`, the number of kernel launches per step
(run under `rocprofv3 --kernel-trace`: tools/fit_round.sh).  GPU box.

    depth    MeshDepthFitter, 200 x 200 depth image of the hand (the reference's tests/test_depth_image_hand_fitting.py workload)
    rgb      MeshRGBFitterWithPose, 1024 x 1024 colour image of the hand mesh (configs[1]-like, untextured colour fit)
    multi8   MeshRGBFitterWithPoseMultiFrame, 8 views of 1024 x 1024 (configs[3]), one batched launch per step
"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deodr_amd import hip_renderer as hr, scenes
from deodr_amd.mesh_fitter import MeshDepthFitter, MeshRGBFitterWithPose, MeshRGBFitterWithPoseMultiFrame

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["depth", "rgb", "multi8"]
steps = 30
hand = np.load(os.path.join(GOLD, "hand_mesh.npz"))
vertices, faces = hand["vertices"], hand["faces"].astype(np.int64)
pixel_dtype = torch.float32 if "--f32" in sys.argv else torch.float64


def synthetic_view(size, angle):
    """a rendered view of the hand as the observation (colour x luminosity, as rgb_image_hand_fitting does with a photograph)"""
    s = scenes.hand_scene(os.path.join(GOLD, "hand_mesh.npz"), size=size, angle=angle, textured=False)
    return np.asarray(s.render(1.0)[0] if hasattr(s, "render") else np.zeros((size, size, 3)))


def build(name):
    if name == "depth":
        d = np.load(os.path.join(GOLD, "depth_hand_fit.npz"))
        depth = d["depth_raw_f32"].astype(np.float64)
        depth[depth == 0] = float(d["max_depth"])
        f = MeshDepthFitter(vertices, faces, d["euler_init"], d["translation_init"], cregu=1000, pixel_dtype=pixel_dtype)
        f.set_image(depth / float(d["max_depth"]), focal=241, distortion=d["distortion"])
        f.set_max_depth(1)
        f.set_depth_scale(float(d["depth_scale"]))
        return f, 1, 200 * 200
    v0 = vertices - vertices.mean(axis=0)
    n = 1 if name == "rgb" else 8
    color, light, ambient, bg = np.array([0.8, 0.6, 0.5]), np.array([0.1, 0.5, 0.4]), 0.6, np.array([0.5, 0.6, 0.7])
    eul = np.stack([np.array([0, a, 0]) for a in (np.linspace(-0.5, 0.5, 8) if n > 1 else [0.0])])

    def make(euler, color):
        if name == "rgb":
            f = MeshRGBFitterWithPose(v0, faces, euler[0], np.zeros(3), color, light, ambient, cregu=1000, pixel_dtype=pixel_dtype)
            f.set_image(np.zeros((1024, 1024, 3)))
        else:
            f = MeshRGBFitterWithPoseMultiFrame(v0, faces, euler, np.zeros((8, 3)), color, light, ambient, cregu=2000, pixel_dtype=pixel_dtype)
            f.set_images([np.zeros((1024, 1024, 3))] * 8)
        f.set_background_color(bg)
        return f

    # the target: the same mesh a little further round and of another colour, rendered by the fitter's own scene (a fit that converges:
    # with a noise image as the target the mesh blows up over the timed iterations and the rasterizer's share with it)
    target = make(eul + np.array([0.04, 0.06, -0.03]), np.array([0.7, 0.65, 0.55])).render().detach().cpu().numpy()
    f = make(eul, color)
    f.set_image(target[0]) if name == "rgb" else f.set_images(list(target))
    return f, n, n * 1024 * 1024


def timed(run, count):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(count):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / count


for name in which:
    f, n, px = build(name)
    if "--graph-only" in sys.argv:  # (under rocprofv3 --kernel-trace: the kernels of the replayed step and nothing else)
        from deodr_amd.mesh_fitter import GraphedStep

        g = GraphedStep(f)
        for _ in range(50):
            g.step_device()
        torch.cuda.synchronize()
        continue
    # every measurement on a FRESH fitter over the same iterations (5 .. 5 + steps), so that the scene is the same in all of them
    for _ in range(5):
        f.step_device()
    dt = timed(f.step_device, steps)  # energies stay on the device: nothing synchronises inside the loop
    f, n, px = build(name)
    for _ in range(5):
        f.step_device()
    dt_host = timed(f.step, steps)  # the reference's protocol: float energy + NumPy images every step
    from deodr_amd.mesh_fitter import GraphedStep
    dt_graph = float("nan")
    try:
        g = GraphedStep(build(name)[0], warmup=3)  # (iterations 0 .. 4)
        dt_graph = timed(g.step_device, steps)
    except Exception as e:
        print(f"{name}: graph capture failed: {e!r}")
    f, n, px = build(name)
    for _ in range(5):
        f.step_device()
    hr.lib().deodr_hip_profile_enable(1)
    for _ in range(steps):
        f.step_device()
    torch.cuda.synchronize()
    hr.lib().deodr_hip_profile_enable(0)
    ms, ln = (ctypes.c_double * 4)(), (ctypes.c_ulonglong * 4)()
    hr.lib().deodr_hip_profile_read(ms, ln)
    raster = sum(ms[i] for i in range(4)) / steps
    print(f"{name}: {n} view(s), {px} pixels: step_device {dt*1e3:.3f} ms, step (float energy + NumPy images) {dt_host*1e3:.3f} ms, ONE HIP-GRAPH REPLAY per step {dt_graph*1e3:.3f} ms; rasterizer kernels "
          f"{raster:.3f} ms per step: the graphed iteration is {dt_graph*1e3/raster:.2f} x the rasterizer; launches of the library's rasterizer per step: {sum(ln[i] for i in range(4))/steps:.1f}")
