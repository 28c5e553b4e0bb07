"""Does running the views of a batch as independent groups on separate streams pay?  (experiment behind the view-group pipelining:
the latency-bound kernels of one group overlap the raster of another)  python tools/overlap_probe.py"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deodr_amd import scenes
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

dev = torch.device("cuda:0")
B, S = 8, 1024
views = [scenes.sphere_scene(size=S, angle=float(a)) for a in np.linspace(-0.5, 0.5, B)]


def make(vs):
    s0 = vs[0]
    stack = lambda n: np.stack([np.asarray(getattr(v, n)) for v in vs])
    ds = DeviceScene(s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"),
                     stack("edgeflags"), S, S, texture=None, background_color=s0.background_color, clockwise=s0.clockwise,
                     vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev)
    r = HipRasterizer.for_scene(ds)
    n, C = len(vs), ds.nb_colors
    obs = torch.rand((n, S, S, C), dtype=torch.float32, device=dev)
    image = torch.empty((n, S, S, C), dtype=torch.float32, device=dev)
    z = torch.empty((n, S, S), dtype=torch.float32, device=dev)
    grads = ds.zero_grads()
    r.render(ds, 1.0, out=(image, z), check_overflow=True)
    return lambda: r.render_fit(ds, obs, 1.0, grads=grads, out=(image, z), check_overflow=False, clear_grads=True)


for ng in (1, 2, 4, 8):
    per = B // ng
    fits = [make(views[g * per:(g + 1) * per]) for g in range(ng)]
    streams = [torch.cuda.Stream() for _ in range(ng)]
    main = torch.cuda.current_stream()

    def step():
        ev = torch.cuda.Event()
        ev.record(main)
        for st, f in zip(streams, fits):
            st.wait_event(ev)
            with torch.cuda.stream(st):
                f()
        for st in streams:
            main.wait_stream(st)

    for _ in range(5):
        step()
    best = 1e9
    for _rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 30)
    print(f"{ng} group(s) of {per} view(s) on {ng} stream(s): {best*1e3:.4f} ms / step = {B*S*S/best/1e6:.0f} Mpixel/s")
