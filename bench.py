"""bench.py -- Mpixels/s forward+backward of the HIP rasterizer on BASELINE.json's metric configuration.

    python bench.py --gpus N --steps K --warmup W [--config 2|4] [--scaling weak|strong]
    (N > 1: one rank per GPU under torch.distributed.run -- started by the driver's launcher, or by bench.py itself when it is run as plain
    `python bench.py --gpus N`: it re-executes itself under `torch.distributed.run --nproc-per-node N` on 127.0.0.1)

Workload (config.workload): BASELINE configs[2], the configuration the metric is quoted on -- 1024x1024, 20 000-triangle
bumpy sphere, C = 4 channels (RGB + depth), sigma = 1, ~33 % coverage, ~650 drawn silhouette edges -- rendered as
`--views` (default 8) poses per GPU per step.  One step = renderScene + renderScene_B for the loss sum (image - obs)^2 over
the whole view batch -- by default through the one-call fit step (deodr_hip_render_scene_fit: same outputs, the forward raster
back-propagates through the tiles without silhouette edges itself), with --two-pass as two calls -- with all inputs resident
in HBM (+, for N > 1, ONE RCCL all-reduce of the shared-parameter gradient, the reduction the reference's
multi-view fitter does on the host at deodr/mesh_fitter.py:518-527).  Views shard across ranks with no data-path
collective, per-GPU work is fixed as N grows ("weak"; `--scaling strong` deals `--views` views of the whole job to the ranks instead:
BASELINE configs[3], one view per GPU at N = 8).  `--config 4` runs BASELINE configs[4] (2048^2, 100k triangles, 1024^2 texture): texture_b
and uv_b -- per-scene arrays that every view of a rank adds into -- are part of the ONE all-reduced buffer (float32, 12.6 MB + the vertex sums).
`--warmup W` is honoured to the step; `cold_start` / `steady_state` say what the same K steps read from an idle GPU and after 1 000 more.

The JSON line carries, besides the driver's contract:
  roofline      dominant kernel group (largest average duration, from device time stamps of EVERY step of the timed region; "raster_fwd_kernel" = tile scan + forward raster -- which in a
                fit step also back-propagates every tile and streams two thirds of the background fill --, "finalize_kernel" =
                finalize + the last third of the fill): achieved = SURVEY.md 8d bytes that group moves (float32 buffers) / its
                average duration, measured with hipEvents on the launch stream inside the timed region; `peak` = 8 TB/s (spec),
                `peak_measured` / `frac_of_measured` = against the copy ceiling of THIS box (hbm_probe); `whole_step` = all of
                SURVEY 8d's bytes / the step time (the number the north star's 40 % is about) and `frac_moved_bytes` = only the
                bytes somebody actually moves
  parity        views 0 and last of the TIMED launch compared with oracle/_ref after the timed region (image 1e-5, gradients
                1e-4 of the largest reference entry): the number belongs to a correct result
  reduction_check  (N > 1 or --force-dist) the all-reduced shared gradient of the LAST timed step against the same reduction done
                again synchronously from that step's gradient arrays (relative error, asserted < 1e-12)
  hbm_probe     device-to-device copy / write-only / read-only bandwidth of this box (1 GiB buffers)
  single_view   the same fit step for ONE view (latency case): eager and replayed from a captured HIP graph
  batch_sweep   the same fit step with 16 and 32 views per launch (where the library saturates; the headline stays at --views)
  other_configs fit-step time of BASELINE configs[1], [3], [4] (outside the timed headline; skipped with --no-other-configs)
  cpu_baseline  the reference's own CPU path (oracle/_ref = unmodified header, g++ -O2) on the host of the GPU box: one
                thread, and one process per view on min(views, cores) cores; bounded samples (rank 0, N = 1 only)
"""

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
KERNELS = ["setup_bin_kernel", "raster_fwd_kernel", "raster_bwd_kernel", "finalize_kernel"]


FILL_W_FIN, FILL_W_FWD = 1, 2  # background fill of a fit step: words dealt finalize : forward raster (dr_forward.h DR_FILL_W_*)


def algorithmic_bytes(H, W, C, T, V, n_views, fused, nonempty_frac=None):
    """Per LAUNCH algorithmic HBM bytes of each kernel group, float32 buffers (untextured, colour background).

    B_fwd = 4 [H W (C+1) + V (2+1+C) + 3T]      B_bwd = 4 [H W C + H W + V (2+1+C) + 3T + V (2+C)]
    split by the kernel that moves them.  Two-call path: forward raster = frame term of B_fwd, adjoint raster = frame term of
    B_bwd.  Fit step: the forward raster does both frame terms for the pixels of the NON-EMPTY tiles (it writes image + z and
    reads the observation where the two-pass adjoint reads image_b; `nonempty_frac` = their share of the frame, counted from the
    tile bitmap of the run) and back-propagates them, tiles with silhouette edges included; the image + z of the empty tiles
    (background, depth = inf) are streamed by extra workgroups of the forward raster (2/3) and of finalize (1/3); the adjoint's
    frame term of the empty tiles is moved by nobody (no owner: nothing to back-propagate) and is reported as `not_moved`."""
    px = H * W
    frame = 4 * px * (C + 1)
    setup, fin = 4 * (V * (3 + C) + 3 * T), 4 * (V * (3 + C) + 3 * T + V * (2 + C))
    if not fused:
        per_view = {"setup_bin_kernel": setup, "raster_fwd_kernel": frame, "raster_bwd_kernel": frame, "finalize_kernel": fin}
    else:
        f = 1.0 if nonempty_frac is None else nonempty_frac
        wf, wn = FILL_W_FWD / (FILL_W_FWD + FILL_W_FIN), FILL_W_FIN / (FILL_W_FWD + FILL_W_FIN)
        per_view = {"setup_bin_kernel": setup, "raster_fwd_kernel": 2 * frame * f + wf * frame * (1 - f), "raster_bwd_kernel": 0,
                    "finalize_kernel": fin + wn * frame * (1 - f), "not_moved": frame * (1 - f)}  # fmt: skip
    return {k: v * n_views for k, v in per_view.items()}


GUIDE_COPY_GBS = 6290.0  # float4 copy measured on MI355X in MI355X_MICROARCH.md: the practical ceiling of a read + write stream


def survey_8d_bytes(H, W, C, T, V, n_views, Vuv=0, tex_hw=None, bg_image=False):
    """SURVEY.md section 8d, algorithmic bytes of forward + adjoint of n_views views (float32 buffers), textured terms included:
    B_fwd = 4 [H W (C+1) + (bg image: H W C) + V (3+C) + 3T (+ 3T + 2 Vuv + V + Ht Wt C)]
    B_bwd = 4 [H W C + H W + V (3+C) + 3T + V (2+C) (+ 3T + 2 Vuv + V + Ht Wt C read + 2 Vuv + V + Ht Wt C gradient write)]"""
    px = H * W
    fwd = px * (C + 1) + (px * C if bg_image else 0) + V * (3 + C) + 3 * T
    bwd = px * C + px + V * (3 + C) + 3 * T + V * (2 + C)
    if tex_hw is not None:
        tex = tex_hw[0] * tex_hw[1] * C
        fwd += 3 * T + 2 * Vuv + V + tex
        bwd += 3 * T + 2 * Vuv + V + tex + 2 * Vuv + V + tex
    return 4 * (fwd + bwd) * n_views


def hbm_probe(dev, nbytes=1 << 30, reps=10):
    """Copy / write-only / read-only bandwidth of this box (GB/s): the ceilings SURVEY.md section 8d asks to report next to the
    8 TB/s of the data sheet.  `lib_*`: the library's own 16-byte non-temporal streaming kernel (deodr_hip_copy_probe: the access
    pattern of its frame stores and background fill); `copy / write / read`: torch device kernels.  1 GiB buffers, hipEvent
    timing, best of `reps`.  `best_copy_GBps` (the larger of the two copies) is what `frac_of_measured` is quoted against."""
    import deodr_amd.hip_renderer as hr

    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    st = torch.cuda.current_stream(dev).cuda_stream

    def lib_call(mode):
        rc = hr.lib().deodr_hip_copy_probe(b.data_ptr(), a.data_ptr(), nbytes, mode, 1, st)
        assert rc == 0, hr.lib().deodr_hip_last_error()

    out = {}
    cases = (("copy", lambda: b.copy_(a), 2 * nbytes), ("write", lambda: b.fill_(1.0), nbytes), ("read", lambda: a.sum(), nbytes),
             ("lib_copy", lambda: lib_call(0), 2 * nbytes), ("lib_write", lambda: lib_call(1), nbytes), ("lib_read", lambda: lib_call(2), nbytes))  # fmt: skip
    for name, fn, moved in cases:
        fn()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e-3)
        out[name + "_GBps"] = moved / best / 1e9
    out["best_copy_GBps"] = max(out["copy_GBps"], out["lib_copy_GBps"])
    out["guide_copy_GBps"] = GUIDE_COPY_GBS
    del a, b
    return out


def parity_check(views, which, image, z, grads, obs, sigma=1.0, sum_views=False):
    """Views `which` of the timed launch against oracle/_ref (the reference's own code): max |image - ref|, gradients relative to the
    largest reference entry -- the tolerances of the north star (1e-5 / 1e-4, float32 pixel buffers).  Per-view gradients: ij_b, colors_b
    (untextured triangles), shade_b (textured ones); uv_b / texture_b are sums over the views of a launch and are compared when the launch
    has ONE view, or -- `sum_views` -- against the sum of per-view checker calls over every view of the launch (a few seconds of CPU per 2048^2 view)."""
    from oracle import api

    ref = api.ref() or api.port()
    fixed = api.ref(fixed=True) or api.port(fixed=True)  # (texture_b: the reference overwrites instead of accumulating, defect D1 of DESIGN.md section 6)
    out = {"checker": "oracle/_ref (unmodified reference header)" if api.ref() is not None else "oracle/deodr_oracle.c", "views": list(which)}
    worst = {"image": 0.0, "z": 0.0, "ij_b": 0.0, "colors_b": 0.0, "shade_b": 0.0}
    rel = lambda a, r: float(np.abs(a - r).max() / max(np.abs(r).max(), 1e-30)) if np.abs(r).max() > 0 else float(np.abs(a).max())
    single = len(views) == 1 and image.shape[0] == 1
    for i in which:
        s = views[i]
        im_ref, z_ref = ref.render(s, sigma)
        im = image[i].cpu().numpy().astype(np.float64)
        zz = z[i].cpu().numpy().astype(np.float64)
        image_b = 2 * (im - obs[i].cpu().numpy().astype(np.float64))
        g_ref = ref.grads(s, sigma, im_ref, z_ref, image_b)
        fin = np.isfinite(z_ref)
        worst["image"] = max(worst["image"], float(np.abs(im - im_ref).max()))
        worst["z"] = max(worst["z"], float(np.abs(zz[fin] - z_ref[fin]).max()) if fin.any() else 0.0, float((np.isfinite(zz) != fin).sum()))
        for k in ("ij_b", "colors_b", "shade_b"):
            worst[k] = max(worst[k], rel(grads[k][i].cpu().numpy(), g_ref[k]))
        if single and grads.get("texture_b") is not None and np.size(s.texture):
            worst["uv_b"] = rel(grads["uv_b"].cpu().numpy(), g_ref["uv_b"])
            worst["texture_b"] = rel(grads["texture_b"].cpu().numpy(), fixed.grads(s, sigma, im_ref, z_ref, image_b)["texture_b"])
    if sum_views and not single and grads.get("texture_b") is not None and np.size(views[0].texture):
        # uv_b / texture_b of a multi-view launch are SUMS over its views (one array per scene, H.h:56-90): against the sum of per-view checker
        # calls over ALL views of the timed launch, each with the residual of that view's own frame of the launch
        uv_sum, tex_sum = 0.0, 0.0
        for i, s in enumerate(views):
            im_ref, z_ref = ref.render(s, sigma)
            image_b = 2 * (image[i].cpu().numpy().astype(np.float64) - obs[i].cpu().numpy().astype(np.float64))
            uv_sum = uv_sum + ref.grads(s, sigma, im_ref, z_ref, image_b)["uv_b"]
            tex_sum = tex_sum + fixed.grads(s, sigma, im_ref, z_ref, image_b)["texture_b"]
        worst["uv_b"] = rel(grads["uv_b"].cpu().numpy(), uv_sum)
        worst["texture_b"] = rel(grads["texture_b"].cpu().numpy(), tex_sum)
        out["summed_over_views"] = {"arrays": ["uv_b", "texture_b"], "views": len(views)}
    out.update({"max_abs_err_image": worst["image"], "max_abs_err_z": worst["z"], "tolerances": {"image": 1e-5, "gradients": 1e-4}})
    out.update({"rel_err_" + k: v for k, v in worst.items() if k not in ("image", "z")})
    out["ok"] = bool(worst["image"] < 1e-5 and worst["z"] < 1e-3 and all(v < 1e-4 for k, v in worst.items() if k not in ("image", "z")))
    return out


def other_configs(dev):
    """Fit-step time of the other BASELINE configurations (same code path: deodr_hip_render_scene_fit, float32 pixel buffers)."""
    from deodr_amd import scenes
    from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

    gold = os.path.join(ROOT, "tests", "golden", "hand_mesh.npz")
    big = dict(size=2048, nu=224, n_rings=224, nb_colors=3, textured=True, texture_size=1024)
    cases = [
        ("configs[1] 1024^2 hand mesh textured, 1 view", lambda: [scenes.hand_scene(gold, size=1024, angle=0.2, textured=True)], 30),
        ("configs[3] 1024^2 hand mesh, 8 views", lambda: [scenes.hand_scene(gold, size=1024, angle=float(a), textured=False) for a in np.linspace(-0.5, 0.5, 8)], 30),
        ("configs[4] 2048^2 100k-triangle shape, 1024^2 texture, 8 views", lambda: [scenes.sphere_scene(angle=float(a), **big) for a in np.linspace(-0.5, 0.5, 8)], 10),
    ]  # fmt: skip
    out = []
    for name, make, steps in cases:
        views = make()
        s0 = views[0]
        stack = lambda n: np.stack([np.asarray(getattr(v, n)) for v in views])
        ds = DeviceScene(
            s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"), stack("edgeflags"),
            s0.height, s0.width, texture=s0.texture if np.size(s0.texture) else None, background_color=s0.background_color,
            background_image=None if s0.background_image is None else stack("background_image"), clockwise=s0.clockwise,
            vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev,
        )  # fmt: skip
        r = HipRasterizer.for_scene(ds)
        n, H, W, Cc = ds.n_views, ds.height, ds.width, ds.nb_colors
        obs = torch.rand((n, H, W, Cc), dtype=torch.float32, device=dev)
        image = torch.empty((n, H, W, Cc), dtype=torch.float32, device=dev)
        z = torch.empty((n, H, W), dtype=torch.float32, device=dev)
        grads = ds.zero_grads()
        fit = lambda: r.render_fit(ds, obs, 1.0, grads=grads, out=(image, z), check_overflow=False, clear_grads=True)
        r.render(ds, 1.0, out=(image, z), check_overflow=True)
        for _ in range(5):
            fit()
        # (best of three short regions: one host-side stall of tens of milliseconds -- seen once in round 6: configs[4] read 4.57 ms instead of
        # 0.81 -- otherwise decides the average of ten steps; the headline is not treated that way: its region is the driver's)
        dt = min(timed_steps(fit, steps) for _ in range(3))
        tex_hw = (ds.texture.shape[0], ds.texture.shape[1]) if ds.texture is not None else None
        alg = survey_8d_bytes(H, W, Cc, ds.nb_triangles, int(ds.depths.shape[1]), n, Vuv=int(ds.uv.shape[0]), tex_hw=tex_hw,
                              bg_image=ds.background_image is not None)  # fmt: skip
        # the timed launch against the checker (view 0; a one-view launch: uv_b and texture_b too)
        parity = parity_check(views, [0], image, z, grads, obs, sum_views=True)
        assert parity["ok"], f"bench: {name}: the timed launch does not match the checker: {parity}"
        out.append({"config": name, "views": n, "ms_per_step": dt * 1e3, "Mpixels_s": n * H * W / dt / 1e6, "parity": parity, "parity_checked": parity["ok"],
                    "roofline": {"alg_bytes": alg, "GBps": alg / dt / 1e9, "frac": alg / dt / 1e9 / HBM_PEAK_GBS, "peak": HBM_PEAK_GBS,
                                 "note": "SURVEY 8d bytes of the whole step / step time (whole-step fraction, as roofline.whole_step)"}})  # fmt: skip
        del ds, r, obs, image, z, grads
    return out


def slow_family(dev, fused_one_view_ms=None):
    """What has no LDS-staged kernel, timed and checked: `antialiase_error` (rasterize_edge_*_error[_B], H.h:2067-2618: the mode of two
    of the reference's four soup-fit goldens, tests/test_triangle_soup_fitting.py:50-67) and more than four channels (Scene3D.render_deferred's
    15-channel frame, dr.py:1053-1174) ran on the un-staged raster_fwd_kernel / raster_bwd_kernel until round 6 (now: the AA instances of the staged
    forward + raster_bwd_edge_err_kernel, and fwd_manyc_tile for a many-channel frame without edges); and the NumPy drop-ins renderSceneCpp /
    renderSceneBCpp (pyx:50-57, 206-215; float64 host arrays over PCIe, stateless adjoint), SURVEY.md section 8d's "report separately"."""
    from deodr_amd import scenes
    from deodr_amd.hip_renderer import DeviceScene, HipRasterizer, renderSceneBCpp, renderSceneCpp
    from oracle import api

    ref, fixed = api.ref() or api.port(), api.ref(fixed=True) or api.port(fixed=True)
    rel = lambda a, r: float(np.abs(a - r).max() / max(np.abs(r).max(), 1e-30))
    out = []

    def device_scene(s):
        return DeviceScene(s.faces, s.faces_uv, s.textured, s.shaded, s.uv, s.ij[None], s.depths[None], s.colors[None], s.shade[None], s.edgeflags[None],
                           s.height, s.width, texture=None, background_color=getattr(s, "background_color", None),
                           background_image=None if getattr(s, "background_image", None) is None else s.background_image[None], clockwise=s.clockwise,
                           vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev)  # fmt: skip

    def entry(name, s, dt, parity, extra_frame_terms, note):
        H, W, Cc, T, V = s.height, s.width, s.nb_colors, len(s.faces), len(s.depths)
        alg = survey_8d_bytes(H, W, Cc, T, V, 1, bg_image=getattr(s, "background_image", None) is not None) + 4 * H * W * extra_frame_terms
        e = {"config": name, "views": 1, "ms_per_step": dt * 1e3, "Mpixels_s": H * W / dt / 1e6, "parity": parity, "parity_checked": parity["ok"],
             "roofline": {"alg_bytes": alg, "GBps": alg / dt / 1e9, "frac": alg / dt / 1e9 / HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "note": note}}  # fmt: skip
        if fused_one_view_ms and (H, W) == (1024, 1024):
            e["times_the_fused_one_view_step"] = dt * 1e3 / fused_one_view_ms
        assert parity["ok"], f"bench: {name}: the timed launch does not match the checker: {parity}"
        out.append(e)

    # ---- antialiase_error = True: Scene2D.render_compare_and_backward's other branch (dr.py:700-724): render with err_buffer, adjoint of sum(err_buffer)
    soup = scenes.soup_scene(n_tri=200, width=256, height=256, seed=2)
    sphere = scenes.sphere_scene(size=1024, angle=0.0)
    for name, s, steps in (("configs[0] 256^2 200-triangle soup, antialiase_error=True (render + render_backward)", soup, 50),
                           ("configs[2] scene, 1 view, antialiase_error=True (render + render_backward)", sphere, 30)):  # fmt: skip
        ds = device_scene(s)
        r = HipRasterizer.for_scene(ds)
        H, W, Cc = s.height, s.width, s.nb_colors
        obs = torch.rand((1, H, W, Cc), dtype=torch.float32, device=dev)
        ones = torch.ones((1, H, W), dtype=torch.float32, device=dev)
        grads = ds.zero_grads()
        image = torch.empty((1, H, W, Cc), dtype=torch.float32, device=dev)
        z = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        state = {}

        def step():
            for g in ("ij_b", "colors_b"):
                grads[g].zero_()
            state["out"] = r.render(ds, 1.0, antialiase_error=True, obs=obs, out=(image, z), check_overflow=False)
            r.render_backward(ds, err_buffer_b=ones, grads=grads)

        r.render(ds, 1.0, out=(image, z), check_overflow=True)
        for _ in range(3):
            step()
        dt = min(timed_steps(step, steps) for _ in range(2))
        o = obs[0].cpu().numpy().astype(np.float64)
        im_ref, z_ref, err_ref = ref.render(s, 1.0, True, o)
        # (the adjoint of the error buffer: against the REPAIRED reference -- defect D2 of DESIGN.md section 6, H.h:2595)
        g_ref = fixed.grads(s, 1.0, im_ref, z_ref, None, True, o, err_ref, np.ones((H, W)))
        err = state["out"][2][0].cpu().numpy().astype(np.float64)
        worst = {"max_abs_err_image": float(np.abs(image[0].cpu().numpy() - im_ref).max()), "rel_err_err_buffer": rel(err, err_ref),
                 "rel_err_ij_b": rel(grads["ij_b"][0].cpu().numpy(), g_ref["ij_b"]), "rel_err_colors_b": rel(grads["colors_b"][0].cpu().numpy(), g_ref["colors_b"])}  # fmt: skip
        worst["ok"] = bool(worst["max_abs_err_image"] < 1e-5 and worst["rel_err_err_buffer"] < 1e-5 and worst["rel_err_ij_b"] < 1e-4 and worst["rel_err_colors_b"] < 1e-4)
        worst["checker"] = "oracle/_ref (adjoint: the build with defect D2 repaired)"
        entry(name, s, dt, worst, Cc + 2, "SURVEY 8d bytes of forward + adjoint + the observation (read) and the error buffer (written, its adjoint read) / step time")
        del ds, r, obs, ones, grads, image, z
    # ---- 15 channels: the frame of Scene3D.render_deferred (dr.py:1053-1174: depth, face ids, barycentrics, normals, luminosity, xyz, colours in ONE
    # forward render of the mesh's triangle soup, sigma = 0 by its own assert, a background image; there is no adjoint of it in the reference)
    s = scenes.deferred_scene(size=1024, channels=15)
    ds = device_scene(s)
    r = HipRasterizer.for_scene(ds)
    H, W, Cc = s.height, s.width, s.nb_colors
    image = torch.empty((1, H, W, Cc), dtype=torch.float32, device=dev)
    z = torch.empty((1, H, W), dtype=torch.float32, device=dev)
    r.render(ds, 0.0, out=(image, z), check_overflow=True)
    fwd15 = lambda: r.render(ds, 0.0, out=(image, z), check_overflow=False)
    for _ in range(3):
        fwd15()
    dt = min(timed_steps(fwd15, 30) for _ in range(2))
    im_ref, z_ref = ref.render(s, 0.0)
    fin = np.isfinite(z_ref)
    worst = {"max_abs_err_image": float(np.abs(image[0].cpu().numpy() - im_ref).max()), "max_abs_err_z": float(np.abs(z[0].cpu().numpy()[fin] - z_ref[fin]).max()),
             "pixels_with_another_owner": int((np.isfinite(z[0].cpu().numpy()) != fin).sum()), "checker": "oracle/_ref"}  # fmt: skip
    worst["ok"] = bool(worst["max_abs_err_image"] < 1e-5 and worst["max_abs_err_z"] < 1e-5 and worst["pixels_with_another_owner"] == 0)
    T, V = len(s.faces), len(s.depths)
    fwd_bytes = 4 * (H * W * (Cc + 1) + H * W * Cc + V * (3 + Cc) + 3 * T)  # SURVEY 8d's forward term with a background image
    out.append({"config": "the frame of Scene3D.render_deferred: 1024^2, the 20 000-triangle mesh as a triangle soup, 15 channels, sigma = 0, background image, FORWARD only (staged forward, fwd_manyc_tile)",
                "views": 1, "ms_per_step": dt * 1e3, "Mpixels_s": H * W / dt / 1e6, "parity": worst, "parity_checked": worst["ok"],
                "roofline": {"alg_bytes": fwd_bytes, "GBps": fwd_bytes / dt / 1e9, "frac": fwd_bytes / dt / 1e9 / HBM_PEAK_GBS, "peak": HBM_PEAK_GBS,
                             "note": "SURVEY 8d forward bytes (C = 15, background image read) / time of the forward call"}})  # fmt: skip
    assert worst["ok"], f"bench: render_deferred frame: the timed launch does not match the checker: {worst}"
    del ds, r, image, z
    # ---- the NumPy drop-ins of the reference's entry points: float64 host arrays in and out, everything over PCIe, the adjoint stateless
    s = scenes.sphere_scene(size=1024, angle=0.0)
    H, W, Cc = s.height, s.width, s.nb_colors
    image, zb = np.zeros((H, W, Cc)), np.zeros((H, W))
    image_b = np.random.RandomState(5).rand(H, W, Cc) - 0.5
    best = 1e9
    for rep in range(4):
        s.clear_gradients()
        t0 = time.perf_counter()
        renderSceneCpp(s, 1.0, image, zb)
        renderSceneBCpp(s, 1.0, image, zb, image_b)
        if rep:  # (the first pair allocates the workspace)
            best = min(best, time.perf_counter() - t0)
    im_ref, z_ref = ref.render(s, 1.0)
    g_ref = ref.grads(s, 1.0, im_ref, z_ref, image_b)
    worst = {"max_abs_err_image": float(np.abs(image - im_ref).max()), "rel_err_ij_b": rel(s.ij_b, g_ref["ij_b"]), "rel_err_colors_b": rel(s.colors_b, g_ref["colors_b"]),
             "checker": "oracle/_ref", "tolerances": {"image": 1e-9, "gradients": 1e-8, "note": "float64 buffers"}}  # fmt: skip
    worst["ok"] = bool(worst["max_abs_err_image"] < 1e-9 and worst["rel_err_ij_b"] < 1e-8 and worst["rel_err_colors_b"] < 1e-8)
    host_bytes = 8 * (2 * H * W * Cc + 2 * H * W + H * W * Cc)  # frame out, frame + image_b in, z out and in (float64)
    entry("configs[2] scene, 1 view, NumPy drop-ins renderSceneCpp + renderSceneBCpp (float64 host arrays, PCIe included, wall clock)", s, best, worst, 0,
          "SURVEY 8d bytes / wall time of the two calls: PCIe-bound by construction -- never part of `value`")
    out[-1]["host_bytes_over_pcie"] = host_bytes
    out[-1]["pcie_GBps_if_nothing_else"] = host_bytes / best / 1e9
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_fit_loop(scene, image_b, budget_s, copies):
    """reps x (renderScene + renderScene_B) of one view on the reference CPU path -> (reps, seconds)."""
    from oracle import api

    ref = api.ref() or api.port()
    H, W, Cc = scene.height, scene.width, scene.nb_colors
    image, z = np.zeros((H, W, Cc)), np.zeros((H, W))
    ref.renderSceneCpp(scene, 1.0, image, z)  # warm-up (page-faults the buffers)
    reps, t_used = 0, 0.0
    while t_used < budget_s and reps < 2000:
        scene.clear_gradients()
        t0 = time.perf_counter()
        ref.renderSceneCpp(scene, 1.0, image, z)
        # the adjoint un-antialiases `image` and rescales `image_b` in place: the reference's Scene2D.render_backward hands it
        # copies (make_copies=True, dr.py:665-699)
        ref.renderSceneBCpp(scene, 1.0, image.copy() if copies else image, z, image_b.copy() if copies else image_b)
        t_used += time.perf_counter() - t0
        reps += 1
    return reps, t_used


def _cpu_worker(args):
    angle, size, budget_s = args
    sys.path.insert(0, ROOT)
    from deodr_amd import scenes

    scene = scenes.sphere_scene(size=size, angle=angle)
    image_b = np.random.RandomState(3).rand(size, size, scene.nb_colors) - 0.5
    return _cpu_fit_loop(scene, image_b, budget_s, True)


def cpu_baseline(scene, image_b, poses, size):
    """The reference CPU path on this host, bounded samples: one thread (the reference is single-threaded) with and without
    the Python-side copies, then one process per view on min(views, cores) cores."""
    from oracle import api

    kind = "reference" if api.ref() is not None else "port"
    H, W = scene.height, scene.width
    reps, t = _cpu_fit_loop(scene, image_b, 8.0, True)
    reps_nc, t_nc = _cpu_fit_loop(scene, image_b.copy(), 3.0, False)
    out = {
        "value": reps * H * W / t / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": kind,
        "sample": f"{reps} x (renderScene + renderScene_B) of ONE view of the same workload ({t:.1f} s, oracle/_ref = unmodified "
                  f"reference header, g++ -O2 -fwrapv, single thread, with the image / image_b copies of Scene2D.render_backward)",
        "without_copies_Mpixels_s": reps_nc * H * W / t_nc / 1e6,
        "cpu_model": cpu_model(), "logical_cores": os.cpu_count(),
    }  # fmt: skip
    try:
        import multiprocessing as mp

        nproc = max(1, min(len(poses), os.cpu_count() or 1))
        with mp.get_context("spawn").Pool(nproc) as pool:
            res = pool.map(_cpu_worker, [(float(a), size, 5.0) for a in poses[:nproc]])
        out["n_process"] = {
            "value": sum(r for r, _ in res) * H * W / max(t for _, t in res) / 1e6, "unit": "Mpixels/s", "cores": nproc,
            "sample": f"one process per view, {nproc} processes x ~5 s, views = the poses of the GPU batch",
        }  # fmt: skip
    except Exception as e:  # the single-thread figure is the contract; say why the other one is missing
        out["n_process"] = {"value": None, "error": repr(e)}
    return out


def timed_steps(step, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def batch_sweep(scenes_mod, DeviceScene, HipRasterizer, dev, S, sigma, batches=(16, 32), cases=None):
    """The same fit step with more views per launch (the headline is quoted at `--views`, default 8): where the library saturates.
    `cases` = [(views, frame size)]: the same mesh on larger frames (`frame_sweep`: more pixels per triangle -- the per-triangle work of set-up and
    finalize and the latency chain of a step are what keeps the 1024^2 / 20 000-triangle step under the frame's bandwidth bound, not the pixel kernels)."""
    out = []
    for B, S in (cases if cases is not None else [(b, S) for b in batches]):
        views = [scenes_mod.sphere_scene(size=S, angle=float(a)) for a in np.linspace(-0.5, 0.5, B)]
        s0 = views[0]
        stack = lambda n: np.stack([np.asarray(getattr(v, n)) for v in views])
        ds = DeviceScene(
            s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"), stack("edgeflags"),
            S, S, texture=None, background_color=s0.background_color, clockwise=s0.clockwise, vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev,
        )  # fmt: skip
        r = HipRasterizer.for_scene(ds)
        Cc = ds.nb_colors
        obs = torch.rand((B, S, S, Cc), dtype=torch.float32, device=dev)
        image = torch.empty((B, S, S, Cc), dtype=torch.float32, device=dev)
        z = torch.empty((B, S, S), dtype=torch.float32, device=dev)
        grads = ds.zero_grads()
        fit = lambda: r.render_fit(ds, obs, sigma, grads=grads, out=(image, z), check_overflow=False, clear_grads=True)
        r.render(ds, sigma, out=(image, z), check_overflow=True)
        for _ in range(5):
            fit()
        dt = min(timed_steps(fit, 20) for _ in range(3))
        alg = sum(v for k, v in algorithmic_bytes(S, S, Cc, ds.nb_triangles, int(ds.depths.shape[1]), B, True).items() if k != "not_moved")
        out.append({"views": B, "size": S, "ms_per_step": dt * 1e3, "Mpixels_s": B * S * S / dt / 1e6, "whole_step_frac": alg / dt / 1e9 / HBM_PEAK_GBS})
        del ds, r, obs, image, z, grads
    return out


def single_view_latency(scenes_mod, DeviceScene, HipRasterizer, dev, S, sigma, obs):
    """One view per call: the latency case (a fit loop on a single image).  Eager launches and one HIP-graph replay per step."""
    s = scenes_mod.sphere_scene(size=S, angle=0.0)
    ds = DeviceScene(
        s.faces, s.faces_uv, s.textured, s.shaded, s.uv, s.ij[None], s.depths[None], s.colors[None], s.shade[None],
        s.edgeflags[None], S, S, texture=None, background_color=s.background_color, clockwise=s.clockwise,
        vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev,
    )  # fmt: skip
    Cc = ds.nb_colors
    r = HipRasterizer.for_scene(ds)
    image = torch.empty((1, S, S, Cc), dtype=torch.float32, device=dev)
    z = torch.empty((1, S, S), dtype=torch.float32, device=dev)
    grads = ds.zero_grads()
    obs1 = obs[None].contiguous()
    fit = lambda: r.render_fit(ds, obs1, sigma, grads=grads, out=(image, z), check_overflow=False, clear_grads=True)
    r.render(ds, sigma, out=(image, z), check_overflow=True)
    for _ in range(5):
        fit()
    eager = min(timed_steps(fit, 50) for _ in range(3))  # (best of three: one host-side stall of tens of ms otherwise decides the average)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fit()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):  # (N > 1: RCCL's watchdog thread queries events while this thread captures)
        fit()
    for _ in range(5):
        graph.replay()
    replay = min(timed_steps(graph.replay, 50) for _ in range(3))
    T, V = ds.nb_triangles, int(ds.depths.shape[1])
    alg = sum(algorithmic_bytes(S, S, Cc, T, V, 1, True).values())
    best = min(eager, replay)
    return {
        "workload": f"1 view {S}x{S} of the same scene per call (deodr_hip_render_scene_fit)", "ms_eager": eager * 1e3,
        "ms_graph_replay": replay * 1e3, "Mpixels_s": S * S / best / 1e6, "alg_GBps": alg / best / 1e9,
        "frac_of_hbm_peak": alg / best / 1e9 / HBM_PEAK_GBS,
    }  # fmt: skip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--views", type=int, default=8, help="views rendered per GPU per step (--scaling strong: views of the whole job per step)")
    ap.add_argument("--size", type=int, default=0, help="frame side (default: the configuration's, 1024 / 2048)")
    ap.add_argument("--config", type=int, choices=[2, 4], default=2,
                    help="BASELINE configs[2] (the metric's configuration: 1024^2, 20k triangles, RGB + depth) or configs[4] (2048^2, 100k triangles, "
                         "1024^2 texture: the batch BASELINE labels '8-view batch on 8 x MI355X'; texture_b and uv_b join the all-reduced buffer)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --views per GPU whatever N; strong: --views in all, dealt to the N ranks (BASELINE configs[3]: 8 views, one per GPU at N = 8)")
    ap.add_argument("--sigma", type=float, default=1.0, help="edge-overdraw width (the metric configuration uses 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-view", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--time-every", type=int, default=20, help="0: no per-kernel timing (otherwise: device time stamps on every step, hipEvents on 6 steps after the timed region)")
    ap.add_argument("--two-pass", action="store_true", help="render and render_backward as two calls (default: the fused fit step, same outputs)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even for one rank (exercises the RCCL path)")
    ap.add_argument("--test-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: for the regression test of the multi-rank path on a ONE-GPU box (tests/test_hip_round3.py): the ranks share "
                         "the visible GPUs and all-reduce over gloo.  Never a measurement: the line says so in config.env_overrides")
    args = ap.parse_args()

    # Nothing outside the command line may change what is timed: the library reads no environment variable, and a variable
    # that LOOKS like one of its former tuning knobs (or a library override used by tools/) is refused rather than ignored.
    overrides = sorted(k for k in os.environ if k.startswith("DEODR_HIP_"))
    assert not overrides, f"unset {overrides}: bench.py times the library as built, without overrides"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: become the launcher (one rank per GPU, torch.distributed.run on 127.0.0.1) -- the
        # ranks print nothing but rank 0's JSON line, so stdout of this process is still that one line
        import socket

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]  # fmt: skip
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.test_backend == "gloo":
        overrides = overrides + ["--test-backend gloo: ranks share the GPUs of the box, NOT a measurement"]
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL writes a banner to the C stdout when the communicator is created; send it to stderr so that the JSON result
        # stays the only (and last) line on stdout
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.test_backend == "gloo":
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=dev)  # RCCL over xGMI
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)  # creates the communicator now
            torch.cuda.synchronize()
        finally:
            C.CDLL(None).fflush(None)
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    import __graft_entry__ as g

    if dist is None or local_rank == 0:
        g.build_hip()  # a no-op when the in-tree library is up to date; never by several ranks at once
    if dist is not None:
        dist.barrier()
    from deodr_amd import scenes
    from deodr_amd import hip_renderer as hr
    from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

    # ---- synthetic inputs: `views` poses of the same mesh per rank (distinct poses on every rank)
    textured = args.config == 4
    S = args.size or (2048 if textured else 1024)
    if args.scaling == "strong":
        assert args.views % world == 0, f"--scaling strong deals --views {args.views} to {world} ranks evenly"
        B, global_views = args.views // world, args.views
    else:
        B, global_views = args.views, args.views * world
    poses = np.linspace(-0.5, 0.5, global_views)[rank * B : (rank + 1) * B]
    mesh = dict(nu=224, n_rings=224) if textured else dict(nu=100, n_rings=100)
    shape = dict(size=S, nb_colors=3, textured=True, texture_size=1024, **mesh) if textured else dict(size=S)
    views = [scenes.sphere_scene(angle=float(a), **shape) for a in poses]
    s0 = views[0]
    stack = lambda name: np.stack([np.asarray(getattr(v, name)) for v in views])
    ds = DeviceScene(
        s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"),
        stack("edgeflags"), S, S, texture=s0.texture if textured else None, background_color=s0.background_color, clockwise=s0.clockwise,
        vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev,
    )  # fmt: skip
    T, V, Cc = ds.nb_triangles, int(ds.depths.shape[1]), ds.nb_colors
    rs = np.random.RandomState(3)
    obs = torch.as_tensor(rs.rand(S, S, Cc).astype(np.float32), device=dev)
    r = HipRasterizer.for_scene(ds)
    image = torch.empty((B, S, S, Cc), dtype=torch.float32, device=dev)
    z = torch.empty((B, S, S), dtype=torch.float32, device=dev)
    grads = ds.zero_grads()
    # Multi-GPU: what the ranks share is the mesh -- 3-D vertex positions and vertex colours -- so the all-reduced buffer is
    # [vertices_b (V x 3), colors_b summed over the views (V x C)].  vertices_b = sum over the views of the adjoint of the view's
    # camera projection (deodr's Camera.project_points_backward, dr.py:397-438) applied to ij_b: ONE launch
    # (deodr_hip_views_gradient_sum), which also adds the colour gradients up over the views.
    reduction = None
    if world > 1 or args.force_dist:
        from deodr_amd import fronthalf
        from deodr_amd.scene3d import DeviceCamera

        verts, _faces = scenes.bumpy_sphere(mesh["nu"], mesh["n_rings"])
        cams = [scenes.fit_camera(S, S, 60.0, verts, scenes.rotx(0.37) @ scenes.roty(0.23 + float(a))) for a in poses]
        assert np.abs(scenes.project(cams[-1], verts)[0] - views[-1].ij).max() < 1e-9  # same cameras as the rendered views
        camera = DeviceCamera(np.stack([c.extrinsic for c in cams]), np.stack([c.intrinsic for c in cams]), S, S, None, dev)
        world_vertices = torch.as_tensor(np.ascontiguousarray(verts, dtype=np.float64), device=dev)
        # deodr_amd.distributed.OverlappedViewsReduction: a communication stream, two alternating sets of gradient buffers, the hand-over
        # through a word of device memory (DeodrHipFitOptions::done_flag / deodr_hip_wait_flag: an event that a second queue waits for costs
        # the render stream ~8 us per step, tools/dist_overhead_probe.py), deodr_hip_views_gradient_sum, the collective
        from deodr_amd.distributed import OverlappedViewsReduction

        reduction = OverlappedViewsReduction(ds, camera, world_vertices[None].expand(B, -1, -1).contiguous(), always_collective=True)

    obs_views = obs.expand(B, S, S, Cc).contiguous()  # one observation per view (here the same synthetic image)
    # The reduction of step k (camera adjoint, packing, RCCL all-reduce) runs on a communication stream while step k + 1 renders:
    # two sets of gradient buffers alternate, and a set is rendered into again only when the reduction that read it has finished.
    it = [0]
    last_slot = [None]

    def step():
        it[0] += 1
        slot = reduction.begin() if reduction is not None else None
        g = slot.grads if slot is not None else grads
        if args.two_pass:
            g["ij_b"].zero_()
            g["colors_b"].zero_()
            r.render(ds, args.sigma, out=(image, z), check_overflow=False)
            # adjoint of L = sum (image - obs)^2: dL/dimage = 2 (image - obs) is formed inside the adjoint kernel (residual mode)
            r.render_backward(ds, residual_obs=obs_views, grads=g)
        else:
            # same outputs in one call: the forward raster back-propagates through the tiles without silhouette edges itself
            # (and zeroes the gradient arrays of the previous step on the way)
            r.render_fit(ds, obs_views, args.sigma, grads=g, out=(image, z), check_overflow=False, clear_grads=True,
                         done_flag=slot.done_flag if slot is not None else None)
        if slot is not None:
            event = None
            if args.two_pass:  # (the two-call step stores no done flag: an event)
                event = torch.cuda.Event()
                event.record()
            reduction.reduce(slot, event=event)
            last_slot[0] = slot

    # first call checks the spill pool once (synchronises), then nothing in the loop does
    r.render(ds, args.sigma, out=(image, z), check_overflow=True)
    def barrier():
        if reduction is not None:
            reduction.finish()
            dist.barrier()
        torch.cuda.synchronize()

    # --warmup is honoured to the step (round 4 ran at least 1 000 whatever it said: the driver flagged the mismatch).  What those steps
    # bought is the core clock: the forward raster is bound by instruction issue and takes 76 us per step in the first ~10 ms after the
    # GPU was idle or streaming memory, 70 - 71 us from ~20 ms of raster work on (tools/ramp_probe.py, profiles/r05e_ramp.txt: device time
    # stamps of every step; set-up and finalize do not move).  A timed region of 20 steps is 2.4 ms, so what it reads is decided by what the
    # GPU did just before.  The line therefore carries three figures: `cold_start` -- W + K steps as the first thing the process does;
    # the headline `ms_per_step` -- W + K steps right after the line's other raster measurements (`single_view`, `batch_sweep`: the same
    # kernels, ~0.2 s of them, on every rank), i.e. at the clocks a fit loop runs at; `steady_state` -- K steps after 1 000 more untimed ones.
    cold_dt = None
    if args.test_backend != "gloo":
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        cold_dt = time.perf_counter() - t0
    probe = hbm_probe(dev) if (rank == 0 and args.test_backend != "gloo") else None
    single_view = sweep = frames = None
    if not args.no_single_view and not textured and args.test_backend != "gloo":
        single_view = single_view_latency(scenes, DeviceScene, HipRasterizer, dev, S, args.sigma, obs)
        # (the larger frames first: what runs right before the headline's warm-up is the same kernels on the same scene -- 16 and 32 views per launch)
        frames = batch_sweep(scenes, DeviceScene, HipRasterizer, dev, S, args.sigma, cases=[(1, 2 * S), (8, 2 * S), (1, 4 * S)]) if rank == 0 and S <= 1024 else None
        sweep = batch_sweep(scenes, DeviceScene, HipRasterizer, dev, S, args.sigma)
    if dist is not None:
        dist.barrier()
    for _ in range(args.warmup):
        step()

    # Per-kernel durations of the timed region: device time stamps (deodr_hip_profile_stamps: the first thread of set-up / tile scan /
    # finalize writes the 100 MHz realtime counter; kernels of one stream run back to back, so consecutive stamps bracket the kernels
    # between them).  EVERY step of the timed region is sampled and nothing is inserted between the launches -- a hipEvent pair per
    # kernel costs the step it brackets ~ 36 us: with 5 of 20 steps bracketed the same code read 0.1368 ms per step instead of 0.1277.
    # hipEvents are still used, AFTER the timed region, as a cross-check of the stamps (`avg_ms_events`).
    stamps = torch.zeros((args.steps + 1, 4), dtype=torch.int64, device=dev) if args.time_every > 0 else None
    barrier()
    if stamps is not None:
        hr.lib().deodr_hip_profile_stamps(stamps.data_ptr(), args.steps + 1)
    # (no gc.collect() / gc.disable() here: tried at the end of round 5 -- the collection idles the GPU for some tens of milliseconds right before the
    # timed region, the core clock falls back and the 200 steps read 0.1136 - 0.1144 ms instead of 0.1101 - 0.1102, profiles/r05z3_ab_bench_gc.txt)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ms_sum = (C.c_double * 4)()
    launches = (C.c_ulonglong * 4)()
    stamp_ms = None
    if stamps is not None:
        step()  # (one more, untimed: its set-up stamp is the end of the last timed finalize)
        barrier()
        hr.lib().deodr_hip_profile_stamps(None, 0)
        st = stamps.cpu().numpy().astype(np.int64)
        K = args.steps
        nxt = st[1 : K + 1, 0]
        ok = (st[:K, 0] > 0) & (st[:K, 1] > 0) & (st[:K, 2] > 0) & (nxt > 0)
        if K > 1:
            ok[K - 1] = False  # (the end of the last step is the set-up stamp of the extra step, launched behind the barrier: host time, not finalize's)
        if args.two_pass or not ok.any():
            stamp_ms = None  # (the two-call step runs set-up twice per step: the rows do not line up with steps; hipEvents below)
        else:
            tick = 1e-5  # ms per tick of the 100 MHz counter
            stamp_ms = {"setup_bin_kernel": float(((st[:K, 1] - st[:K, 0])[ok]).mean() * tick), "raster_fwd_kernel": float(((st[:K, 2] - st[:K, 1])[ok]).mean() * tick),
                        "raster_bwd_kernel": 0.0, "finalize_kernel": float(((nxt - st[:K, 2])[ok]).mean() * tick), "samples": int(ok.sum()),
                        "step_ms_by_stamps": float(((nxt - st[:K, 0])[ok]).mean() * tick),
                        # (the GPU's own view of the region: a step the host was late for -- the queue starts EMPTY behind the barrier, a host pause
                        # longer than its lead shows as one long step -- moves the mean and the maximum, not the median)
                        "step_ms_by_stamps_median": float(np.median((nxt - st[:K, 0])[ok]) * tick), "step_ms_by_stamps_max": float(((nxt - st[:K, 0])[ok]).max() * tick),
                        # [step, ms] of the steps that took more than 1.25 x the median (at most 12 of them): where in the region the GPU waited for the host
                        "late_steps": [[int(i), round(float((nxt - st[:K, 0])[i] * tick), 4)] for i in np.nonzero(ok & ((nxt - st[:K, 0]) > 1.25 * np.median((nxt - st[:K, 0])[ok])))[0][:12]]}  # fmt: skip
        # cross-check with hipEvents on a few more steps (outside the timed region: they perturb what they measure)
        hr.lib().deodr_hip_profile_enable(1)
        for _ in range(6):
            step()
        barrier()
        hr.lib().deodr_hip_profile_enable(0)
        hr.lib().deodr_hip_profile_read(ms_sum, launches)
    # the same K steps once more in the steady state (1 000 untimed steps in between): what --warmup is worth on this box
    steady_dt = None
    if args.test_backend != "gloo":
        for _ in range(1000):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        steady_dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt, steady_dt or 0.0, cold_dt or 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, steady_dt, cold_dt = float(tt[0].item()), (float(tt[1].item()) if steady_dt is not None else None), (float(tt[2].item()) if cold_dt is not None else None)

    # the shared gradient the pipeline all-reduced in the last timed step against the same reduction done again, step by step and
    # synchronously, from that step's gradient arrays (every rank takes part: it is a collective)
    reduction_check = None
    if dist is not None:
        from deodr_amd import fronthalf

        from deodr_amd.distributed import SharedGradientBuffer

        slot = last_slot[0]
        again = SharedGradientBuffer(V, Cc, slot.buffer.n_uv, slot.buffer.texture_shape, slot.shared.dtype, dev)
        if textured:
            # texture_b / uv_b were all-reduced in place (the library accumulates them straight into the packed buffer): the step is
            # rendered again, synchronously, into a fresh set -- equal up to the order of the float32 texture atomics
            g = ds.zero_grads()
            r.render_fit(ds, obs_views, args.sigma, grads=g, out=(image, z), check_overflow=False, clear_grads=True)
            again.texture_b.copy_(g["texture_b"])
            again.uv_b.copy_(g["uv_b"])
            tol = 1e-4
        else:
            g, tol = slot.grads, 1e-12
        vb, cs = torch.zeros(V, 3, dtype=torch.float64, device=dev), torch.zeros(V, Cc, dtype=torch.float64, device=dev)
        fronthalf.views_gradient_sum(reduction.posed, reduction.camera, g["ij_b"], vb, colors_b=g["colors_b"], colors_sum=cs)
        again.vertices_b.copy_(vb)
        again.colors_b.copy_(cs)
        local_max = float(again.flat.abs().max())
        dist.all_reduce(again.flat)
        torch.cuda.synchronize()
        err = float((slot.shared - again.flat).abs().max() / again.flat.abs().max())
        timed_out = int(reduction.wait_status.item())
        reduction_check = {"ranks": world, "rel_err": err, "tolerance": tol, "ok": bool(err < tol and local_max > 0 and not timed_out), "values": int(again.flat.numel()),
                           "bytes": int(again.flat.numel() * again.flat.element_size()), "dtype": str(again.flat.dtype).replace("torch.", ""),
                           "parts": "texture_b | vertices_b | colors_b | uv_b" if textured else "vertices_b | colors_b",
                           "flag_wait_timed_out": bool(timed_out), "sync": "event" if args.two_pass else "done flag (deodr_hip_wait_flag)"}  # fmt: skip
        assert reduction_check["ok"], f"bench: the all-reduced shared gradient of the timed loop differs from a synchronous reduction: {reduction_check}"

    # spill pool never overflowed and the scene was valid during the run (deferred check, outside the timed region)
    over, need, errs = r.status(ds)
    assert not over and not errs, "spill pool overflowed (or invalid scene) during the benchmark"

    if rank == 0:
        px = global_views * S * S * args.steps
        # which tiles held primitives / silhouette edges in this run (from the workspace: tile bitmap, saved edge counts)
        nonempty = edge_tiles = None
        try:
            nonempty, edge_tiles = hr.tile_census(r, ds)
        except Exception as e:
            print(f"bench: tile census unavailable ({e!r})", file=sys.stderr)
        fused = not args.two_pass
        ntiles = B * ((S + 7) // 8) ** 2
        alg_8d = algorithmic_bytes(S, S, Cc, T, V, B, fused)  # every frame byte of the step
        alg = algorithmic_bytes(S, S, Cc, T, V, B, fused, nonempty / ntiles if (fused and nonempty is not None) else None)
        if textured:
            # configs[4]: SURVEY 8d's textured terms for the whole step; no split by kernel group (a textured fit step is four launches since round 5 -- the
            # tiles with silhouette edges are back-propagated inside the forward raster, as in untextured scenes --, but the byte model of the groups
            # has no texture terms)
            alg_8d = {"whole": survey_8d_bytes(S, S, Cc, T, V, B, Vuv=int(ds.uv.shape[0]), tex_hw=(int(ds.texture.shape[0]), int(ds.texture.shape[1])))}
            alg = dict(alg_8d, **{k: None for k in KERNELS})
        per_kernel = {}
        for i, k in enumerate(KERNELS):
            n = max(int(launches[i]), 1)
            ev_ms = ms_sum[i] / n
            avg_ms, samples = (stamp_ms[k], stamp_ms["samples"]) if stamp_ms is not None else (ev_ms, int(launches[i]))
            per_kernel[k] = {"avg_ms": avg_ms, "launches": samples, "alg_bytes": alg.get(k), "GBps": alg[k] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 and alg.get(k) else None,
                             "avg_ms_events": ev_ms, "event_launches": int(launches[i])}  # fmt: skip
        dom = max(KERNELS, key=lambda k: per_kernel[k]["avg_ms"])
        traffic = traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):  # HBM bytes per launch from the last PMC run (tools/profile_round.sh), gfx950 correction applied
            tjson = json.load(open(tpath))
            traffic = (tjson.get(dom) or {}).get("bytes_per_launch")
            # NOT measured in this run (PMC passes need rocprofv3 around the process): which profile the figure is from
            traffic_source = dict(tjson.get("_source") or {}, file="profiles/traffic_latest.json", measured_in_this_run=False)
        kernel_ms = sum(v["avg_ms"] for v in per_kernel.values())
        step_s = dt / args.steps
        whole = sum(v for k, v in alg_8d.items() if k != "not_moved")
        moved = sum(v for k, v in alg.items() if k != "not_moved" and v is not None)
        if probe is None:  # (--test-backend gloo: a functional test, nothing is measured)
            probe = {"best_copy_GBps": GUIDE_COPY_GBS, "note": "not measured (--test-backend gloo)"}
        peak_meas = probe["best_copy_GBps"]
        dom_GBps = per_kernel[dom]["GBps"] if not textured else whole / step_s / 1e9  # (configs[4]: no per-group bytes -> the whole step)
        single_floor = ("one view per GPU at N = 8: a step cannot be shorter than the single-view fit step (`single_view` of the N = 1 line: ~0.058 ms "
                        "against ~0.117 ms for 8 views on one GPU, i.e. the strong-scaling curve of this batch tops out near 2 x)")
        out = {
            "metric": "Mpixels/s forward+backward, 1024^2 20k-tri scene" if not textured else "Mpixels/s forward+backward, 2048^2 100k-tri textured scene (BASELINE configs[4])",
            "value": px / dt / 1e6, "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_s * 1e3,
            # the same K steps timed again after 1 000 more untimed ones: what a longer --warmup would have read on this box
            "steady_state": None if steady_dt is None else {"ms_per_step": steady_dt / args.steps * 1e3, "value": px / steady_dt / 1e6, "untimed_steps_after_the_headline": 1000 + (7 if stamps is not None else 0)},
            # the same W + K steps as the first thing the process did (GPU idle before: the core clock has not ramped up yet)
            "cold_start": None if cold_dt is None else {"ms_per_step": cold_dt / args.steps * 1e3, "value": px / cold_dt / 1e6},
            "order": "cold_start (W + K steps) first; then hbm_probe, single_view, frame_sweep (rank 0), batch_sweep (on every rank); then the W warm-up and K timed steps of the headline; "
                     "steady_state, parity check, other configurations and the CPU baseline after it",
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[2]: {S}x{S}, {T}-triangle bumpy sphere, C={Cc} (RGB+depth), sigma=1, " if not textured else
                                    f"BASELINE configs[4]: {S}x{S}, {T}-triangle bumpy sphere, textured Gouraud, {ds.texture.shape[1]}x{ds.texture.shape[0]} texture, C={Cc}, sigma=1, ")
                                   + f"{B} views per GPU per step, float32 pixel buffers / float64 vertex arrays, all-double arithmetic",
                       "step": "renderScene + renderScene_B (two calls)" if args.two_pass else "deodr_hip_render_scene_fit (forward + adjoint of sum (image - obs)^2, one call)",
                       "views_per_gpu": B, "global_views": global_views, "env_overrides": overrides,
                       "nonempty_tiles": nonempty, "edge_tiles": edge_tiles, "tiles": ntiles,
                       "parallelism": f"views sharded {B}/GPU" + (", 1 RCCL all-reduce of the shared gradient per step" if world > 1 else "")
                                      + (" (texture_b | vertices_b | colors_b | uv_b in one float32 buffer)" if textured and world > 1 else ""),
                       **({"strong_scaling_floor": single_floor} if args.scaling == "strong" else {})},
            # dominant kernel group (largest hipEvent time).  achieved / frac: the SURVEY 8d bytes THIS group moves / its average
            # duration, against the 8 TB/s of the data sheet; frac_of_measured: against the copy bandwidth measured on this box.
            # whole_step: all of SURVEY 8d's bytes / the step time (the north star's 40 % is about this number); frac_moved_bytes:
            # only the bytes somebody moves (the adjoint's frame term of the empty tiles is moved by nobody).
            "roofline": {"bound": "hbm", "kernel": dom if not textured else "whole step (four launches; no per-group byte split for textured scenes)",
                         "achieved": dom_GBps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (dom_GBps or 0) / HBM_PEAK_GBS, "traffic": traffic if not textured else None,
                         "traffic_source": traffic_source if not textured else None,
                         "peak_measured": peak_meas, "frac_of_measured": (dom_GBps or 0) / peak_meas,
                         "timing": ("device time stamps of every step of the timed region (deodr_hip_profile_stamps); avg_ms_events = hipEvents on 6 steps after it"
                                    if stamp_ms is not None else "hipEvents on steps after the timed region"),
                         "step_ms_by_stamps": None if stamp_ms is None else stamp_ms["step_ms_by_stamps"],
                         "step_ms_by_stamps_median": None if stamp_ms is None else stamp_ms["step_ms_by_stamps_median"],
                         "step_ms_by_stamps_max": None if stamp_ms is None else stamp_ms["step_ms_by_stamps_max"],
                         "late_steps": None if stamp_ms is None else stamp_ms["late_steps"],
                         "whole_step": {"alg_bytes": whole, "GBps": whole / step_s / 1e9, "frac": whole / step_s / 1e9 / HBM_PEAK_GBS,
                                        "frac_of_measured": whole / step_s / 1e9 / peak_meas,
                                        "frac_of_guide_copy": whole / step_s / 1e9 / GUIDE_COPY_GBS, "moved_bytes": moved,
                                        "frac_moved_bytes": moved / step_s / 1e9 / HBM_PEAK_GBS},
                         "not_moved_bytes": alg.get("not_moved"),
                         "kernel_time_fraction_of_step": kernel_ms / (step_s * 1e3), "per_kernel": per_kernel},
            "hbm_probe": probe,
        }  # fmt: skip
        if reduction_check is not None:
            out["reduction_check"] = reduction_check
        if not args.no_parity_check:
            # the launch that was timed (its last step's outputs are still in image / z / the gradient set it wrote), views 0 and last
            last = last_slot[0].grads if last_slot[0] is not None else grads
            out["parity"] = parity_check(views, sorted({0, B - 1}), image, z, last, obs_views, sigma=args.sigma)
            out["parity_checked"] = out["parity"]["ok"]
            assert out["parity"]["ok"], f"bench: the timed launch does not match the checker: {out['parity']}"
        if world == 1 and not args.no_other_configs and not textured:
            out["other_configs"] = other_configs(dev)
            out["other_configs"] += slow_family(dev, single_view["ms_eager"] if single_view is not None else None)
        if single_view is not None:
            out["single_view"], out["batch_sweep"] = single_view, sweep
            # (the north star's 40 % is asked of the forward + backward pass of this scene: where more views per launch take the same code)
            out["roofline"]["whole_step"]["saturated"] = max(out["batch_sweep"], key=lambda e: e["whole_step_frac"])
            if frames is not None:
                # the same 20 000-triangle mesh on larger frames (views, size): where the pixel kernels take the step once the per-triangle work is amortised
                out["frame_sweep"] = frames
        if world == 1 and not args.no_cpu_baseline and not textured:
            out["cpu_baseline"] = cpu_baseline(views[0], 2 * (image[0] - obs).cpu().numpy().astype(np.float64), poses, S)
        result_line = json.dumps(out)
    else:
        result_line = None
    if dist is not None:
        dist.barrier()  # rank 0 has finished its read-backs before any rank tears the communicator down
        dist.destroy_process_group()
    if result_line is not None:  # the ONE JSON line, last thing on stdout
        sys.stdout.flush()
        print(result_line, flush=True)


if __name__ == "__main__":
    main()
