"""SURVEY.md 8(e) "single huge frame: split the screen into row bands per GPU (replicated mesh), same gradient all-reduce" -- what a band of ONE frame
costs on one GPU, as a fit step of its own (the rank of a row-band split would run exactly this: the whole mesh, `height / n` rows of the frame), next to
the whole frame: the band with the longest step bounds the split's step.  The band is rendered through the existing boundary: vertex rows shifted by the
band's first row (an integer: the pixel grid does not move), frame height = the band's, observation sliced.  Checked against the whole frame's rows
(image 1e-5; the band gradients summed over the bands against the whole frame's gradient).
    python tools/band_probe.py [--size 1024] [--bands 2,4,8]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deodr_amd import scenes
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

arg = lambda name, default: type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default
S, steps = arg("--size", 1024), arg("--steps", 200)
bands = [int(b) for b in arg("--bands", "2,4,8").split(",")]
dev = torch.device("cuda:0")
s0 = scenes.sphere_scene(size=S, angle=0.0)
obs_full = torch.rand((1, S, S, s0.nb_colors), dtype=torch.float32, device=dev)


def band_step(row0, rows):
    ij = np.asarray(s0.ij, dtype=np.float64).copy()
    ij[:, 1] -= row0
    ds = DeviceScene(s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, ij[None], np.asarray(s0.depths)[None], np.asarray(s0.colors)[None],
                     np.asarray(s0.shade)[None], np.asarray(s0.edgeflags)[None], rows, S, texture=None, background_color=s0.background_color,
                     clockwise=s0.clockwise, vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev)
    r = HipRasterizer.for_scene(ds)
    obs = obs_full[:, row0:row0 + rows].contiguous()
    image = torch.empty((1, rows, S, ds.nb_colors), dtype=torch.float32, device=dev)
    z = torch.empty((1, rows, S), dtype=torch.float32, device=dev)
    grads = ds.zero_grads()
    fit = lambda: r.render_fit(ds, obs, 1.0, grads=grads, out=(image, z), check_overflow=False, clear_grads=True)
    r.render(ds, 1.0, out=(image, z), check_overflow=True)
    for _ in range(20):
        fit()
    best = 1e9
    for _rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fit()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    g = {k: v.detach().double().cpu().numpy().copy() for k, v in grads.items() if torch.is_tensor(v)} if isinstance(grads, dict) else None
    return best, image.cpu().numpy()[0], g


t_full, image_full, g_full = band_step(0, S)
print(f"{S}x{S}, one view of the benchmark mesh: whole frame {t_full*1e3:.4f} ms per fit step")
for n in bands:
    rows = S // n
    times, err, gsum = [], 0.0, None
    for k in range(n):
        t, im, g = band_step(k * rows, rows)
        times.append(t)
        err = max(err, float(np.abs(im - image_full[k * rows:(k + 1) * rows]).max()))
        if g is not None:
            gsum = g if gsum is None else {kk: gsum[kk] + g[kk] for kk in g}
    gerr = max(float(np.abs(gsum[k] - g_full[k]).max() / max(np.abs(g_full[k]).max(), 1e-30)) for k in gsum) if gsum else float("nan")
    print(f"  {n} bands of {rows} rows: {' '.join(f'{t*1e3:.4f}' for t in times)} ms -> the split's step {max(times)*1e3:.4f} ms = {t_full/max(times):.2f} x the whole frame's rate"
          f" (+ one all-reduce of the shared gradient);  image vs the whole frame's rows {err:.1e}, summed gradients {gerr:.1e}")
