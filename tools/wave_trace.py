"""Timeline of the waves of setup_bin_kernel / finalize_kernel (library built with -DDR_WAVE_TRACE).  Run on the GPU box."""
import sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:  # a variant of the library (tools/build_variants.sh): trace builds, other occupancies
    import deodr_amd.hip_renderer as _hr

    _hr.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from deodr_amd import scenes, hip_renderer as hr
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

dev = torch.device("cuda:0")
S, B = 1024, (int(sys.argv[sys.argv.index('--views') + 1]) if '--views' in sys.argv else 8)
views = [scenes.sphere_scene(size=S, angle=float(a)) for a in np.linspace(-0.5, 0.5, B)]
s0 = views[0]
stack = lambda name: np.stack([np.asarray(getattr(v, name)) for v in views])
ds = DeviceScene(s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"),
                 stack("edgeflags"), S, S, texture=None, background_color=s0.background_color, clockwise=s0.clockwise,
                 vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev)
r = HipRasterizer.for_scene(ds)
obs = torch.rand((B, S, S, ds.nb_colors), dtype=torch.float32, device=dev)
grads = ds.zero_grads()
for _ in range(4):
    r.render_fit(ds, obs, 1.0, grads=grads, clear_grads=True)
torch.cuda.synchronize()
buf = np.zeros((3, 1 << 18, 2), dtype=np.uint64)
L = hr.lib()
L.deodr_hip_debug_wave_trace.argtypes = [C.c_void_p, C.c_size_t]
assert L.deodr_hip_debug_wave_trace(buf.ctypes.data, buf.nbytes) == 0
T = ds.nb_triangles
for which, name in enumerate(("setup_bin_kernel", "finalize_kernel (or the finalize workgroups of the forward raster)", "raster_fwd_fast_kernel (walkers)")):
    t = buf[which].astype(np.int64)
    ok = t[:, 1] > 0
    t = t[ok]
    if len(t) == 0:
        continue
    t = t[t[:, 0] > t[:, 0].max() - 30000]  # (entries of earlier, larger grids stay in the buffer: keep the last launch -- 300 us)
    if which == 1 and (buf[2][:, 1] > 0).any():  # same clock: finalize workgroups inside the forward raster are shown on the walkers' time axis
        t0_fwd = buf[2].astype(np.int64)[buf[2][:, 1] > 0][:, 0].min()
        print(f"   (first of these wavefronts starts {(t[:, 0].min() - t0_fwd) * 0.01:.1f} us after the first walker, the last one ends {(t[:, 1].max() - t0_fwd) * 0.01:.1f} us after it)")
    t0 = t[:, 0].min()
    start, end = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01  # microseconds
    dur = end - start
    print(f"{name}: {len(t)} waves, kernel span {end.max():.1f} us; wave start p50 {np.percentile(start,50):.1f} p90 {np.percentile(start,90):.1f} max {start.max():.1f};"
          f" wave life mean {dur.mean():.2f} p50 {np.percentile(dur,50):.2f} p90 {np.percentile(dur,90):.2f} p99 {np.percentile(dur,99):.2f} max {dur.max():.2f}")
    hist, edges = np.histogram(end, bins=10, range=(0, end.max()))
    print("   waves ending per tenth of the span:", hist.tolist())
    hist, edges = np.histogram(start, bins=10, range=(0, end.max()))
    print("   waves starting per tenth of the span:", hist.tolist())
    long = np.argsort(-dur)[:5]
    print("   longest waves (start, life):", [(round(float(start[i]), 1), round(float(dur[i]), 1)) for i in long])
    ts = np.linspace(0, end.max(), 41)[1:-1]  # waves in flight over time
    print("   waves in flight at 39 instants:", [int(((start <= x) & (end > x)).sum()) for x in ts])
    late = np.argsort(-end)[:8]
    print("   last waves to end (start, life):", [(round(float(start[i]), 1), round(float(dur[i]), 1)) for i in late])

# phases inside the wavefronts that work on flagged edges (set-up: 1 flags compacted, 2 inputs loaded, 3 record computed, 4 record
# stored, 5 binning done; finalize: 1 flags compacted, 2 inputs arrived, 3 arithmetic done, 4 accumulators cleared)
ph = np.zeros((4, 1 << 16, 8), dtype=np.uint64)
L.deodr_hip_debug_wave_phase.argtypes = [C.c_void_p, C.c_size_t]
if L.deodr_hip_debug_wave_phase(ph.ctypes.data, ph.nbytes) == 0:
    for which, name, last in ((0, "setup_bin_kernel edge", 5), (1, "finalize_kernel edge", 4), (2, "setup_bin_kernel tri", 4), (3, "finalize_kernel tri", 3)):
        t = ph[which].astype(np.int64)
        ok = np.all(t[:, :last + 1] > 0, axis=1)
        t = t[ok]
        if len(t) == 0:
            continue
        d = np.diff(t[:, :last + 1], axis=1) * 0.01
        print(f"{name}: {len(t)} wavefronts that went all the way; phase durations (us) mean / p90:",
              [(round(float(d[:, i].mean()), 2), round(float(np.percentile(d[:, i], 90)), 2)) for i in range(last)], " whole mean %.2f" % ((t[:, last] - t[:, 0]).mean() * 0.01))

# --slots: where the wavefronts of the forward raster ran (HW_ID / XCC_ID of each, recorded by the trace build) -- how full the SIMDs are while the
# grid still has workgroups to hand out, and how long a SIMD waits between the end of one wavefront and the start of the next
if "--slots" in sys.argv:
    hw = np.zeros((3, 1 << 18, 2), dtype=np.uint32)
    L.deodr_hip_debug_wave_hw.argtypes = [C.c_void_p, C.c_size_t]
    assert L.deodr_hip_debug_wave_hw(hw.ctypes.data, hw.nbytes) == 0
    for which, name, cap in ((0, "setup_bin_kernel", None), (1, "finalize_kernel", None), (2, "raster_fwd_fast_kernel", None)):
        t = buf[which].astype(np.int64)
        ok = t[:, 1] > 0
        ok &= t[:, 0] > t[ok, 0].max() - 30000
        t, h = t[ok], hw[which][ok]
        if len(t) == 0:
            continue
        t0 = t[:, 0].min()
        start, end = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01
        role = h[:, 0] >> 31
        simd = (h[:, 1].astype(np.int64) & 0xf) << 16 | (h[:, 0].astype(np.int64) & 0xff30)  # XCC | SE, SH, CU, SIMD (PIPE and WAVE bits masked out)
        ids = np.unique(simd)
        last_start = start.max()
        print(f"{name}: {len(t)} wavefronts ({int(role.sum())} of them fill workgroups: life mean {end[role == 1].mean() - start[role == 1].mean() if role.any() else 0:.2f} us) on {len(ids)} SIMDs;"
              f" the last one starts at {last_start:.1f} us, the kernel ends at {end.max():.1f}")
        peak, occ, gaps, idle_at_end = [], [], [], []
        horizon = np.percentile(start, 99.5)  # the grid has workgroups left to hand out until about here
        for s in ids:
            m = simd == s
            ev = np.concatenate([np.stack([start[m], np.ones(m.sum())], 1), np.stack([end[m], -np.ones(m.sum())], 1)])
            ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
            n = np.cumsum(ev[:, 1])
            peak.append(n.max())
            tt = ev[:, 0]
            sel = tt[:-1] < horizon
            occ.append(float((n[:-1][sel] * np.diff(tt)[sel]).sum() / max(horizon - tt[0], 1e-9)))
            # a wavefront ends while the SIMD is at its peak: how long until the next one starts there
            e_sorted, s_sorted = np.sort(end[m]), np.sort(start[m])
            for e in e_sorted[e_sorted < horizon]:
                j = np.searchsorted(s_sorted, e, side="left")
                if j < len(s_sorted):
                    gaps.append(s_sorted[j] - e)
        gaps = np.array(gaps)
        print(f"   waves per SIMD at once: peak p50 {np.percentile(peak, 50):.0f} max {max(peak):.0f}; mean occupancy until {horizon:.1f} us (99.5 % of the starts): {np.mean(occ):.2f} waves per SIMD"
              f" = {np.mean(occ) * len(ids):.0f} in flight")
        print(f"   from the end of a wavefront to the next start on the same SIMD: mean {gaps.mean():.2f} p50 {np.percentile(gaps, 50):.2f} p90 {np.percentile(gaps, 90):.2f} us ({len(gaps)} ends)")
