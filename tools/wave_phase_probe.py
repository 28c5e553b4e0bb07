"""Reads the per-wave timers of a debug build of the library (a scratch build, see profiles/README.md round 5): mean time from the start of
a tile walker to the arrival of its first work entry, and to its end, per walker instance.   python tools/wave_phase_probe.py --lib <dbg.so>"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deodr_amd import hip_renderer as hr, scenes
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer
hr.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
B, S = 8, 1024
dev = torch.device("cuda:0")
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
for _ in range(200):
    fit()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 24)()
hr.lib().deodr_hip_debug_read(out, 1)
N = 50
for _ in range(N):
    fit()
torch.cuda.synchronize()
hr.lib().deodr_hip_debug_read(out, 1)
a = np.array(list(out), dtype=np.float64).reshape(3, 8)
for mode, name in ((1, "head walkers (edge-capable instance)"), (2, "other walkers (pairs, edge-free tiles)")):
    w, t1, t2, n, got = a[mode, 0], a[mode, 1], a[mode, 3], a[mode, 4], a[mode, 5]
    print(f"{name}: {w / N:.0f} walkers per step, {n / N:.0f} entries ({n / max(w, 1):.2f} per walker; {got / max(w, 1):.2f} of the walkers had one); "
          f"start -> first entry known {t1 / max(got, 1) * 0.01:.2f} us; life {t2 / max(w, 1) * 0.01:.2f} us; slot-time per step {t2 / N * 0.01 / 5120:.1f} us of 5 120 slots")
