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
    rs = np.random.RandomState(0)
    v0 = vertices - vertices.mean(axis=0)
    n = 1 if name == "rgb" else 8
    img = [np.clip(0.5 + 0.2 * rs.randn(1024, 1024, 3), 0, 1) for _ in range(n)]  # (what is timed does not depend on the picture)
    if name == "rgb":
        f = MeshRGBFitterWithPose(v0, faces, np.zeros(3), np.zeros(3), np.array([0.8, 0.6, 0.5]), np.array([0.1, 0.5, 0.4]), 0.6, cregu=1000, pixel_dtype=pixel_dtype)
        f.set_image(img[0])
    else:
        eul = np.stack([np.array([0, a, 0]) for a in np.linspace(-0.5, 0.5, 8)])
        f = MeshRGBFitterWithPoseMultiFrame(v0, faces, eul, np.zeros((8, 3)), np.array([0.8, 0.6, 0.5]), np.array([0.1, 0.5, 0.4]), 0.6, cregu=2000, pixel_dtype=pixel_dtype)
        f.set_images(img)
    f.set_background_color(np.array([0.5, 0.6, 0.7]))
    return f, n, n * 1024 * 1024


for name in which:
    f, n, px = build(name)
    if "--graph-only" in sys.argv:  # (under rocprofv3 --kernel-trace: the kernels of the replayed step and nothing else)
        from deodr_amd.mesh_fitter import GraphedStep

        g = GraphedStep(f)
        for _ in range(50):
            g.step_device()
        torch.cuda.synchronize()
        continue
    for _ in range(5):
        f.step_device()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        f.step_device()  # energies stay on the device: nothing synchronises inside the loop
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    for _ in range(steps):
        f.step()  # the reference's protocol: float energy + NumPy images every step
    torch.cuda.synchronize()
    dt_host = (time.perf_counter() - t0) / steps
    from deodr_amd.mesh_fitter import GraphedStep
    dt_graph = float("nan")
    try:
        g = GraphedStep(f)
        for _ in range(3):
            g.step_device()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.step_device()
        torch.cuda.synchronize()
        dt_graph = (time.perf_counter() - t0) / steps
    except Exception as e:
        print(f"{name}: graph capture failed: {e!r}")
    hr.lib().deodr_hip_profile_enable(1)
    for _ in range(8):
        f.step_device()
    torch.cuda.synchronize()
    hr.lib().deodr_hip_profile_enable(0)
    ms, ln = (ctypes.c_double * 4)(), (ctypes.c_ulonglong * 4)()
    hr.lib().deodr_hip_profile_read(ms, ln)
    raster = sum(ms[i] for i in range(4)) / 8
    print(f"{name}: {n} view(s), {px} pixels: step_device {dt*1e3:.3f} ms, step (float energy + NumPy images) {dt_host*1e3:.3f} ms, ONE HIP-GRAPH REPLAY per step {dt_graph*1e3:.3f} ms; rasterizer kernels "
          f"{raster:.3f} ms per step = {100*raster/(dt*1e3):.0f} % of step_device; launches of the library per step: {sum(ln[i] for i in range(4))/8:.1f}")
