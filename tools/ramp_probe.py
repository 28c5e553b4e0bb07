"""How the fit-step time of the bench workload settles after the GPU was idle: device time stamps (deodr_hip_profile_stamps) of every step of
a run of N steps, started (a) after a second of idleness, (b) right after the copy-bandwidth probe, (c) right after 300 single-view steps.
    python tools/ramp_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deodr_amd import hip_renderer as hr
from deodr_amd import scenes
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

dev = torch.device("cuda:0")
S = 1024


def make(B):
    views = [scenes.sphere_scene(size=S, angle=float(a)) for a in np.linspace(-0.5, 0.5, B)]
    s0 = views[0]
    stack = lambda n: np.stack([np.asarray(getattr(v, n)) for v in views])
    ds = DeviceScene(s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"),
                     stack("edgeflags"), S, S, texture=None, background_color=s0.background_color, clockwise=s0.clockwise,
                     vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev)  # fmt: skip
    r = HipRasterizer.for_scene(ds)
    C = ds.nb_colors
    obs = torch.rand((B, S, S, C), dtype=torch.float32, device=dev)
    image = torch.empty((B, S, S, C), dtype=torch.float32, device=dev)
    z = torch.empty((B, S, S), dtype=torch.float32, device=dev)
    grads = ds.zero_grads()
    r.render(ds, 1.0, out=(image, z), check_overflow=True)
    return lambda: r.render_fit(ds, obs, 1.0, grads=grads, out=(image, z), check_overflow=False, clear_grads=True)


fit8, fit1 = make(8), make(1)


def run(label, n=400):
    stamps = torch.zeros((n + 1, 4), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    hr.lib().deodr_hip_profile_stamps(stamps.data_ptr(), n + 1)
    t0 = time.perf_counter()
    for _ in range(n):
        fit8()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    fit8()
    torch.cuda.synchronize()
    hr.lib().deodr_hip_profile_stamps(None, 0)
    st = stamps.cpu().numpy()
    step = (st[1 : n + 1, 0] - st[:n, 0]) * 1e-2  # us
    setup, fwd, fin = (st[:n, 1] - st[:n, 0]) * 1e-2, (st[:n, 2] - st[:n, 1]) * 1e-2, (st[1 : n + 1, 0] - st[:n, 2]) * 1e-2
    pick = [0, 1, 2, 3, 5, 8, 12, 20, 30, 50, 100, 200, n - 1]
    print(f"{label}: wall {wall * 1e3:.4f} ms / step; device step (us) at step k:")
    print("   " + "  ".join(f"{k}:{step[k]:.1f}({setup[k]:.0f}/{fwd[k]:.0f}/{fin[k]:.0f})" for k in pick))
    print(f"   mean of steps 5..24: {step[5:25].mean():.1f}   25..99: {step[25:100].mean():.1f}   200..: {step[200:].mean():.1f}", flush=True)


time.sleep(1.0)
run("after 1 s idle")
time.sleep(1.0)
a = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
for _ in range(60):
    b.copy_(a)
torch.cuda.synchronize()
run("after 60 x 1 GiB copies")
time.sleep(1.0)
for _ in range(300):
    fit1()
torch.cuda.synchronize()
run("after 300 single-view steps")
time.sleep(0.05)
run("after 50 ms idle")
time.sleep(0.005)
run("after 5 ms idle")
