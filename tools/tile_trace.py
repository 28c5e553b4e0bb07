"""Per-tile cycle trace of the adjoint's edge kernel (needs a library built with -DDR_TILE_TRACE: the kernel then writes
eight counters over the first row of each edge tile in the rendered image).  Run on the GPU box."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:  # a variant of the library (tools/build_variants.sh): trace builds, other occupancies
    import deodr_amd.hip_renderer as _hr

    _hr.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from deodr_amd import scenes
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

dev = torch.device("cuda:0")
S, B = 1024, 8
poses = np.linspace(-0.5, 0.5, B)
views = [scenes.sphere_scene(size=S, angle=float(a)) for a in poses]
s0 = views[0]
stack = lambda name: np.stack([np.asarray(getattr(v, name)) for v in views])
ds = DeviceScene(s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"),
                 stack("edgeflags"), S, S, texture=None, background_color=s0.background_color, clockwise=s0.clockwise,
                 vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev)
r = HipRasterizer.for_scene(ds)
image = torch.empty((B, S, S, ds.nb_colors), dtype=torch.float32, device=dev)
z = torch.empty((B, S, S), dtype=torch.float32, device=dev)
obs = torch.rand((B, S, S, ds.nb_colors), dtype=torch.float32, device=dev)
grads = ds.zero_grads()
for _ in range(3):
    r.render(ds, 1.0, out=(image, z))
    r.render_backward(ds, residual_obs=obs, grads=grads)
torch.cuda.synchronize()
zi = image.cpu().numpy().view(np.uint32)
v, yy, xx = np.nonzero(zi[..., 0] == 0x7FC0BEEF)
rows = np.concatenate([zi[v, yy, xx, :], zi[v, yy, xx + 1, :]], 1).astype(np.int64)
print("edge tiles:", len(rows), "per view", np.bincount(v))
ne = rows[:, 1]
print("edges/tile: mean %.1f  p50 %d  p90 %d  p99 %d  max %d" % (ne.mean(), *np.percentile(ne, [50, 90, 99]), ne.max()))
names = ["gather", "passA", "passB", "owner+flush"]
t = rows[:, 2:6]
d = np.diff(np.concatenate([np.zeros((len(t), 1), np.int64), t], 1), axis=1)
for i, n in enumerate(names):
    print("%-12s cycles: mean %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f" % (n, d[:, i].mean(), *np.percentile(d[:, i], [50, 90]), d[:, i].max()))
print("total        cycles: mean %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f" % (t[:, 3].mean(), *np.percentile(t[:, 3], [50, 90]), t[:, 3].max()))
for lo, hi in [(1, 2), (3, 4), (5, 8), (9, 16), (17, 32), (33, 64)]:
    m = (ne >= lo) & (ne <= hi)
    if m.any():
        print("n_edges %2d-%2d: %5d tiles, total mean %8.0f, passB mean %8.0f" % (lo, hi, m.sum(), t[m, 3].mean(), d[m, 2].mean()))
