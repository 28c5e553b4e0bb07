"""Would splitting the views of a fit step over several streams pay?  8 views in one call vs G groups of 8/G views, each group
with its own workspace on its own stream (no library change: plain concurrent calls).  Run on the GPU box."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deodr_amd import scenes
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

dev = torch.device("cuda:0")
S, B = 1024, 8
views = [scenes.sphere_scene(size=S, angle=float(a)) for a in np.linspace(-0.5, 0.5, B)]
s0 = views[0]


def make(vs):
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


for G in (1, 2, 4, 8):
    fits = [make(views[i * B // G:(i + 1) * B // G]) for i in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]

    JOIN = "--join" in sys.argv  # every step forks from and joins back to the main stream (what a fit loop needs)

    def step():
        main = torch.cuda.current_stream()
        for f, st in zip(fits, streams):
            if JOIN:
                st.wait_stream(main)
            with torch.cuda.stream(st):
                f()
        if JOIN:
            for st in streams:
                main.wait_stream(st)

    for _ in range(5):
        step()
    best = 1e9
    for _rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 40)
    # host-side cost of issuing the calls alone
    t0 = time.perf_counter()
    for _ in range(40):
        step()
    host = (time.perf_counter() - t0) / 40
    torch.cuda.synchronize()
    if JOIN and "--graph" in sys.argv:
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        for _ in range(5):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            graph.replay()
        torch.cuda.synchronize()
        print(f"   as one HIP graph per step: {(time.perf_counter() - t0) / 40 * 1e3:.4f} ms / step")
    print(f"{G} group(s) of {B // G} views: {best * 1e3:.4f} ms / step ({B * S * S / best / 1e6:.0f} Mpixel/s), host issue time {host * 1e3:.4f} ms")
