"""Per-kernel times (hipEvents) of what runs on the un-staged kernels: antialiase_error (render + render_backward), many channels.
    python tools/slow_times.py [--lib variant.so]"""
import sys, os, time, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deodr_amd import scenes, hip_renderer as hr
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

if "--lib" in sys.argv:
    hr.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
dev = torch.device("cuda:0")


def dscene(s, pixel=torch.float32):
    return DeviceScene(s.faces, s.faces_uv, s.textured, s.shaded, s.uv, s.ij[None], s.depths[None], s.colors[None], s.shade[None], s.edgeflags[None],
                       s.height, s.width, texture=None, background_color=getattr(s, "background_color", None),
                       background_image=None if getattr(s, "background_image", None) is None else s.background_image[None], clockwise=s.clockwise,
                       vertex_dtype=torch.float64, pixel_dtype=pixel, device=dev)  # fmt: skip


def timed(step, n=20):
    for _ in range(3):
        step()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    hr.lib().deodr_hip_profile_enable(1)
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    hr.lib().deodr_hip_profile_enable(0)
    ms, ln = (ctypes.c_double * 4)(), (ctypes.c_ulonglong * 4)()
    hr.lib().deodr_hip_profile_read(ms, ln)
    per = [ms[i] / max(ln[i], 1) * 1e3 for i in range(4)]
    return best, per


def report(name, best, per):
    print(f"{name}: {best*1e3:.4f} ms / step   [set-up {per[0]:.1f}, forward raster {per[1]:.1f}, adjoint raster {per[2]:.1f}, finalize {per[3]:.1f} us]")


sphere = scenes.sphere_scene(size=1024, angle=0.0)
soup = scenes.soup_scene(n_tri=200, width=256, height=256, seed=2)
for name, s in (("sphere 1024^2 1 view", sphere), ("soup 256^2 200 triangles", soup)):
    ds = dscene(s)
    r = HipRasterizer.for_scene(ds)
    H, W, Cc = s.height, s.width, s.nb_colors
    obs = torch.rand((1, H, W, Cc), dtype=torch.float32, device=dev)
    ones = torch.ones((1, H, W), dtype=torch.float32, device=dev)
    grads = ds.zero_grads()
    image, z = torch.empty((1, H, W, Cc), dtype=torch.float32, device=dev), torch.empty((1, H, W), dtype=torch.float32, device=dev)
    r.render(ds, 1.0, out=(image, z), check_overflow=True)

    def aa():
        r.render(ds, 1.0, antialiase_error=True, obs=obs, out=(image, z), check_overflow=False)
        r.render_backward(ds, err_buffer_b=ones, grads=grads)

    def two_call():
        r.render(ds, 1.0, out=(image, z), check_overflow=False)
        r.render_backward(ds, residual_obs=obs, grads=grads)

    def fit():
        r.render_fit(ds, obs, 1.0, grads=grads, out=(image, z), check_overflow=False, clear_grads=True)

    report(name + ", antialiase_error two calls", *timed(aa))
    report(name + ", image mode two calls (staged)", *timed(two_call))
    report(name + ", image mode fit step (staged, fused)", *timed(fit))
    hr.force_generic(True)
    report(name + ", image mode two calls (un-staged)", *timed(two_call))
    hr.force_generic(False)

# the frame of Scene3D.render_deferred (dr.py:1053-1174): triangle soup of the mesh (3 vertices per face), 15 channels, sigma = 0, background image, forward only
for C in (15, 4):
    s = scenes.deferred_scene(size=1024, channels=C)
    ds = dscene(s)
    r = HipRasterizer.for_scene(ds)
    image, z = torch.empty((1, s.height, s.width, C), dtype=torch.float32, device=dev), torch.empty((1, s.height, s.width), dtype=torch.float32, device=dev)
    r.render(ds, 0.0, out=(image, z), check_overflow=True)
    report(f"render_deferred shape, C = {C}, sigma = 0, forward only", *timed(lambda: r.render(ds, 0.0, out=(image, z), check_overflow=False)))
    if C == 15:
        hr.force_generic(True)
        report(f"render_deferred shape, C = {C}, sigma = 0, forward only (un-staged)", *timed(lambda: r.render(ds, 0.0, out=(image, z), check_overflow=False)))
        hr.force_generic(False)
