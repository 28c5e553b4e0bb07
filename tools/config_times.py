"""Fit-step time of the other BASELINE configurations (sanity check for pathologies: texture gradients, big frames).  GPU box."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:  # a variant of the library (tools/build_variants.sh): trace builds, other occupancies
    import deodr_amd.hip_renderer as _hr

    _hr.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from deodr_amd import scenes
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

dev = torch.device("cuda:0")
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hand_mesh.npz")


def run(name, views, steps=20):
    s0 = views[0]
    stack = lambda n: np.stack([np.asarray(getattr(v, n)) for v in views])
    tex = s0.texture if np.size(s0.texture) else None
    ds = DeviceScene(s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"),
                     stack("edgeflags"), s0.height, s0.width, texture=tex, background_color=s0.background_color,
                     background_image=None if s0.background_image is None else stack("background_image"), clockwise=s0.clockwise,
                     vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev)
    r = HipRasterizer.for_scene(ds)
    n, H, W, C = ds.n_views, ds.height, ds.width, ds.nb_colors
    obs = torch.rand((n, H, W, C), dtype=torch.float32, device=dev)
    image = torch.empty((n, H, W, C), dtype=torch.float32, device=dev)
    z = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    grads = ds.zero_grads()
    for _ in range(3):
        r.render_fit(ds, obs, 1.0, grads=grads, out=(image, z), clear_grads=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render_fit(ds, obs, 1.0, grads=grads, out=(image, z), clear_grads=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    import ctypes
    from deodr_amd import hip_renderer as hr
    hr.lib().deodr_hip_profile_enable(1)
    for _ in range(4):
        r.render_fit(ds, obs, 1.0, grads=grads, out=(image, z), clear_grads=True)
    torch.cuda.synchronize()
    hr.lib().deodr_hip_profile_enable(0)
    ms, ln = (ctypes.c_double * 4)(), (ctypes.c_ulonglong * 4)()
    hr.lib().deodr_hip_profile_read(ms, ln)
    per = [ms[i] / max(ln[i], 1) for i in range(4)]
    print(f"{name}: {n} view(s) {W}x{H} C={C} T={ds.nb_triangles} texture={'yes' if tex is not None else 'no'}: {dt*1e3:.3f} ms / fit step, {n*H*W/dt/1e6:.0f} Mpixel/s"
          f"   [set-up {per[0]:.3f}, forward {per[1]:.3f}, edge tiles {per[2]:.3f}, finalize {per[3]:.3f} ms]")


ONLY = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""  # e.g. --only "configs[4] shape, 8" (profiling one configuration)
_run = run
def run(name, views, steps=20):
    if ONLY in name:
        _run(name, views() if callable(views) else views, steps)


run("configs[1] hand, textured", lambda: [scenes.hand_scene(GOLD, size=1024, angle=0.2, textured=True)])
run("configs[3] hand, 8 views", lambda: [scenes.hand_scene(GOLD, size=1024, angle=float(a), textured=False) for a in np.linspace(-0.5, 0.5, 8)])
run("configs[2] sphere 20k", lambda: [scenes.sphere_scene(size=1024, angle=float(a)) for a in np.linspace(-0.5, 0.5, 8)])
big = dict(size=2048, nu=224, n_rings=224, nb_colors=3, textured=True, texture_size=1024)
run("configs[4] shape, 1 view", lambda: [scenes.sphere_scene(**big)])
for nv in (2, 4):
    run(f"configs[4] shape, {nv} views", lambda: [scenes.sphere_scene(angle=float(a), **big) for a in np.linspace(-0.5, 0.5, nv)], steps=10)
run("configs[4] shape, 8 views", lambda: [scenes.sphere_scene(angle=float(a), **big) for a in np.linspace(-0.5, 0.5, 8)], steps=10)
