import sys, os, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from deodr_amd import scenes
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer
dev = torch.device("cuda:0")
B, S = 8, 1024
views = [scenes.sphere_scene(size=S, angle=float(a)) for a in np.linspace(-0.5, 0.5, B)]
s0 = views[0]
stack = lambda n: np.stack([np.asarray(getattr(v, n)) for v in views])
ds = DeviceScene(s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"),
                 stack("edgeflags"), S, S, texture=None, background_color=s0.background_color, clockwise=s0.clockwise,
                 vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev)
r = HipRasterizer.for_scene(ds)
C = ds.nb_colors
obs = torch.rand((B, S, S, C), dtype=torch.float32, device=dev)
image = torch.empty((B, S, S, C), dtype=torch.float32, device=dev)
z = torch.empty((B, S, S), dtype=torch.float32, device=dev)
grads = ds.zero_grads()
fit = lambda: r.render_fit(ds, obs, 1.0, grads=grads, out=(image, z), check_overflow=False, clear_grads=True)
r.render(ds, 1.0, out=(image, z), check_overflow=True)
for _ in range(1000):
    fit()
torch.cuda.synchronize()
def region(K, mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fit()
    if mode == "spin":
        e = torch.cuda.Event(); e.record()
        while not e.query():
            pass
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K
for K in (20, 200):
    for mode in ("sync", "spin", "sync", "spin"):
        ts = [region(K, mode) for _ in range(7)]
        print(K, mode, "min %.4f med %.4f ms" % (min(ts) * 1e3, sorted(ts)[3] * 1e3), flush=True)
# host issue time alone
t0 = time.perf_counter()
for _ in range(200):
    fit()
print("host issue per step %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
torch.cuda.synchronize()
