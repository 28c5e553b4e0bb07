"""Fit-step time of the bench workload (configs[2], 8 views) with per-kernel hipEvent times, for A/B runs of library variants
on ONE box:  python tools/step_time.py [--lib tools/variants/libdeodr_hip_fwd4.so] [--views 8] [--steps 40]"""
import sys, os, time, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deodr_amd import scenes, hip_renderer as hr
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

arg = lambda name, default: type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default
if "--lib" in sys.argv:
    hr.LIB_PATH = os.path.abspath(arg("--lib", ""))
B, S, steps = arg("--views", 8), arg("--size", 1024), arg("--steps", 40)
dev = torch.device("cuda:0")
views = [scenes.sphere_scene(size=S, angle=float(a)) for a in np.linspace(-0.5, 0.5, B)]
s0 = views[0]
stack = lambda n: np.stack([np.asarray(getattr(v, n)) for v in views])
ds = DeviceScene(s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"),
                 stack("edgeflags"), S, S, texture=None, background_color=s0.background_color, clockwise=s0.clockwise,
                 vertex_dtype=torch.float32 if "--vtx32" in sys.argv else torch.float64, pixel_dtype=torch.float32, device=dev)
r = HipRasterizer.for_scene(ds)
C = ds.nb_colors
obs = torch.rand((B, S, S, C), dtype=torch.float32, device=dev)
image = torch.empty((B, S, S, C), dtype=torch.float32, device=dev)
z = torch.empty((B, S, S), dtype=torch.float32, device=dev)
grads = ds.zero_grads()
fit = lambda: r.render_fit(ds, obs, 1.0, grads=grads, out=(image, z), check_overflow=False, clear_grads=True)
r.render(ds, 1.0, out=(image, z), check_overflow=True)
for _ in range(5):
    fit()
best = 1e9
for _rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fit()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / steps)
hr.lib().deodr_hip_profile_enable(1)
for _ in range(8):
    fit()
torch.cuda.synchronize()
hr.lib().deodr_hip_profile_enable(0)
ms, ln = (ctypes.c_double * 4)(), (ctypes.c_ulonglong * 4)()
hr.lib().deodr_hip_profile_read(ms, ln)
per = [ms[i] / max(ln[i], 1) for i in range(4)]
print(f"{os.path.basename(hr.LIB_PATH)}: {B} views {S}x{S}: {best*1e3:.4f} ms / fit step = {B*S*S/best/1e6:.0f} Mpixel/s"
      f"   [set-up {per[0]*1e3:.1f}, forward {per[1]*1e3:.1f}, edge tiles {per[2]*1e3:.1f}, finalize {per[3]*1e3:.1f} us]  census {hr.tile_census(r, ds)}")
