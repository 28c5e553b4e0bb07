"""Reads the per-class timers of a debug build of the library's forward raster (a scratch build: profiles/README.md, round 5): when the
workgroups of each class -- background fill, head walkers (tiles with silhouette edges / many triangles), other walkers (pairs, edge-free
tiles) -- start and end, relative to the start of the tile scan kernel in front of the forward raster.
    python tools/wave_phase_probe.py --lib <dbg.so>"""
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
for _ in range(300):
    fit()
torch.cuda.synchronize()
stamps = torch.zeros((4, 4), dtype=torch.int64, device=dev)
hr.lib().deodr_hip_profile_stamps(stamps.data_ptr(), 4)
for _ in range(3):
    fit()  # the records of the LAST step survive (row 2 of the stamps)
torch.cuda.synchronize()
NREC = 1 << 16
out = (ctypes.c_ulonglong * (2 * NREC))()
hr.lib().deodr_hip_debug_read(out, NREC)
fit()
torch.cuda.synchronize()
hr.lib().deodr_hip_profile_stamps(None, 0)
st = stamps.cpu().numpy()
print("stamps of the recorded step (us from the scan kernel's start): set-up started %.1f, scan 0, finalize %.1f, next set-up %.1f" % (
    (st[2, 0] - st[2, 1]) * 0.01, (st[2, 2] - st[2, 1]) * 0.01, (st[3, 0] - st[2, 1]) * 0.01))
rec = np.array(list(out), dtype=np.uint64).reshape(NREC, 2)
rec = rec[rec[:, 1] > 0]
cls = (rec[:, 0] & np.uint64(3)).astype(int)
start = (rec[:, 0] >> np.uint64(2)).astype(np.float64) * 0.01
end = rec[:, 1].astype(np.float64) * 0.01
for c, name in ((0, "background fill workgroups"), (1, "head walkers (edge-capable instance)"), (2, "other walkers (pairs, edge-free tiles)")):
    m = cls == c
    if not m.any():
        continue
    life = end[m] - start[m]
    print(f"{name}: {m.sum()} workgroups; start first {start[m].min():.1f} mean {start[m].mean():.1f} last {start[m].max():.1f} us; end last {end[m].max():.1f} us; "
          f"life mean {life.mean():.2f} p50 {np.percentile(life, 50):.2f} p90 {np.percentile(life, 90):.2f} max {life.max():.2f} us; slot-time {life.sum() / 5120:.1f} us of 5 120 slots")
    print("   starts per 4 us:", " ".join(str(int(x)) for x in np.histogram(start[m], bins=np.arange(0, 100, 4))[0]))
    print("   ends   per 4 us:", " ".join(str(int(x)) for x in np.histogram(end[m], bins=np.arange(0, 100, 4))[0]))
inflight = [(int(((start <= t) & (end > t)).sum())) for t in np.arange(2, 90, 4)]
print("workgroups in flight at 2, 6, 10, ... us:", inflight)
late = np.argsort(-end)[:8]
print("last to end (class, start, life):", [(int(cls[i]), round(float(start[i]), 1), round(float(end[i] - start[i]), 1)) for i in late])
