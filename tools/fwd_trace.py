"""Per-tile phase timing of the fused forward raster (library built with -DDR_FWD_TRACE: the kernel writes eight counters
over the first row of each non-empty tile in the z buffer).  Run on the GPU box."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:  # a variant of the library (tools/build_variants.sh): trace builds, other occupancies
    import deodr_amd.hip_renderer as _hr

    _hr.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from deodr_amd import scenes
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

dev = torch.device("cuda:0")
S, B = 1024, (int(sys.argv[sys.argv.index('--views') + 1]) if '--views' in sys.argv else 8)
TEXTURED = "--textured" in sys.argv  # the shape of BASELINE configs[4]: 2048^2, 100 k triangles, 1024^2 texture, C = 3
if TEXTURED:
    S = 2048
    views = [scenes.sphere_scene(size=S, nu=224, n_rings=224, nb_colors=3, textured=True, texture_size=1024, angle=float(a)) for a in np.linspace(-0.5, 0.5, B)]
else:
    views = [scenes.sphere_scene(size=S, angle=float(a)) for a in np.linspace(-0.5, 0.5, B)]
s0 = views[0]
stack = lambda name: np.stack([np.asarray(getattr(v, name)) for v in views])
ds = DeviceScene(s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"),
                 stack("edgeflags"), S, S, texture=s0.texture if TEXTURED else None, background_color=s0.background_color, clockwise=s0.clockwise,
                 vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev)
r = HipRasterizer.for_scene(ds)
obs = torch.rand((B, S, S, ds.nb_colors), dtype=torch.float32, device=dev)
grads = ds.zero_grads()
for _ in range(3):
    image, z, _g = r.render_fit(ds, obs, 1.0, grads=grads, clear_grads=True)
torch.cuda.synchronize()
zi = z.cpu().numpy().view(np.uint32)
v, yy, xx = np.nonzero(zi == 0x7FC0F00D)
rows = np.stack([zi[v, yy + i // 8, xx + i % 8] for i in range(16)], 1).astype(np.int64)
ntri, nedge = rows[:, 1] & 0xFFFF, rows[:, 1] >> 16
print("non-empty tiles:", len(rows), " with edges:", int((nedge > 0).sum()), " triangles/tile mean %.1f p50 %d p90 %d max %d" % (ntri.mean(), *np.percentile(ntri, [50, 90]), ntri.max()))
names = ["prologue + counters", "pass 1 (stage, spans, z)", "resolve + edges", "frame stores", "adjoint of pass 1"]
t = rows[:, 2:7]
d = np.diff(np.concatenate([np.zeros((len(t), 1), np.int64), t], 1), axis=1)
sel = nedge == 0
for i, n in enumerate(names):
    print("%-26s cycles (tiles without edges): mean %7.0f  p50 %7.0f  p90 %7.0f" % (n, d[sel, i].mean(), *np.percentile(d[sel, i], [50, 90])))
print("%-26s cycles: mean %7.0f  p50 %7.0f  p90 %7.0f  max %7.0f" % ("whole tile", t[sel, 4].mean(), *np.percentile(t[sel, 4], [50, 90]), t[sel, 4].max()))
sub = rows[:, 8:11]
for nm, a, b in (("  entry -> records staged", t[:, 0], sub[:, 0]), ("  spans", sub[:, 0], sub[:, 1]), ("  depth test", sub[:, 1], sub[:, 2]), ("  shading of the winner", sub[:, 2], t[:, 1])):
    dd = (b - a)[sel & (ntri <= 16)]
    print("%-26s cycles (first batch, tiles without edges): mean %7.0f  p50 %7.0f  p90 %7.0f" % (nm, dd.mean(), *np.percentile(dd, [50, 90])))
for lo, hi in [(1, 4), (5, 8), (9, 16), (17, 32), (33, 1000)]:
    m = sel & (ntri >= lo) & (ntri <= hi)
    if m.any():
        print("ntri %3d-%3d: %6d tiles, pass 1 mean %7.0f, whole tile mean %7.0f" % (lo, hi, m.sum(), d[m, 1].mean(), t[m, 4].mean()))
# tiles with silhouette edges (a fit step runs their adjoint in the same wavefront): whole-tile cycles by size, and the longest
e = nedge > 0
if e.any():
    whole, fwd = t[:, 4], t[:, 3]
    print("tiles with edges: whole tile cycles mean %7.0f p50 %7.0f p90 %7.0f max %7.0f;  up to the frame stores mean %7.0f  (adjoint: the difference)" % (
        whole[e].mean(), *np.percentile(whole[e], [50, 90]), whole[e].max(), fwd[e].mean()))
    for lo, hi in [(1, 2), (3, 8), (9, 16), (17, 1000)]:
        m = e & (nedge >= lo) & (nedge <= hi)
        if m.any():
            print("nedge %3d-%3d: %6d tiles, ntri mean %5.1f, forward part mean %7.0f, adjoint part mean %7.0f, whole max %7.0f" % (
                lo, hi, m.sum(), ntri[m].mean(), fwd[m].mean(), (whole - fwd)[m].mean(), whole[m].max()))
    top = np.argsort(-whole)[:12]
    print("longest tiles (ntri, nedge, forward part, whole):", [(int(ntri[i]), int(nedge[i]), int(fwd[i]), int(whole[i])) for i in top])
    d2 = np.diff(np.concatenate([np.zeros((len(t), 1), np.int64), t], 1), axis=1)
    for i in top[:4]:
        print("   phases of (%d tri, %d edges):" % (ntri[i], nedge[i]), d2[i].tolist())
