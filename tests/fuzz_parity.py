"""Randomised parity sweep (not collected by pytest; run on the GPU box):  python tests/fuzz_parity.py [n_scenes [soups|meshes|both|large]]

Random triangle soups of random sizes / densities / flags / view counts, and random views of bumpy spheres (shared vertices,
silhouette edges, 1-6 channels, zoomed past the frame): forward-only call, one-call fit step, two-call adjoint, antialiase_error
forward + adjoint and perspective-correct forward of the HIP library against the CPU checker (oracle/).  Prints the worst errors;
exits non-zero on a miss."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from deodr_amd import scenes  # noqa: E402
from deodr_amd.hip_renderer import HipRasterizer  # noqa: E402
from hip_util import device_scene, rel_err  # noqa: E402
from oracle import api  # noqa: E402


def draw_scene(it, rs, replay=False):
    """Scene number `it` of the sweep: (views, sigma, pixel dtype, description).  With replay=True the stream `rs` is advanced
    through scenes 0 .. it-1 first, so that a single scene of a run can be re-created."""
    for skip in range(it if replay else 0):
        draw_scene(skip, rs)
    H, W = int(rs.choice([24, 40, 61, 96, 128, 200])), int(rs.choice([24, 48, 77, 96, 160, 256]))
    n_tri = int(rs.choice([1, 3, 12, 40, 150, 400]))
    n_views = int(rs.choice([1, 1, 2, 3, 5]))
    sigma = float(rs.choice([0.0, 0.7, 1.0, 2.5]))
    dt = torch.float64 if rs.rand() < 0.5 else torch.float32
    textured = float(rs.choice([0.0, 0.0, 0.5, 1.0]))
    tex_size, min_area = int(rs.choice([8, 16, 33])), float(rs.choice([2.0, 30.0, 300.0]))
    # (an area the frame cannot hold makes soup_scene burn its 10 000 attempts per triangle: capped since r02i -- the scene numbers of
    # the r02i logs under profiles/ were drawn without the cap; what they found is pinned in tests/test_hip_parity2.py and test_sim.py)
    min_area = min(min_area, H * W / 40.0)
    rounded, strict = it % 4 == 1, it % 5 != 2
    if rounded and not strict:
        strict = True  # integer vertices under the non-strict rule: the reference's adjoint counts the middle-vertex row twice
    background_color = None  # one colour for all the views of a scene (the scene holds one)
    views = []
    for v in range(n_views):
        s = scenes.soup_scene(n_tri=n_tri, width=W, height=H, seed=1000 * it + 7, clockwise=bool(it & 1), textured_ratio=textured, flat=False,
                              texture_size=tex_size, min_area=min_area)
        if v:  # other views: the same mesh, vertices moved
            s.ij = s.ij + rs.randn(*s.ij.shape) * 2.0
        if rounded:
            s.ij = np.round(s.ij)  # integer vertices: every tie of the fill rule
        if it % 3 == 0:
            colour = rs.rand(3)
            background_color = colour if background_color is None else background_color
            s.background_image, s.background_color = None, background_color
        s.strict_edge, s.integer_pixel_centers, s.backface_culling = bool(strict), bool(it % 7 != 3), True
        if textured > 0.0 and it % 3 == 1:
            # textured && !shaded triangles (H.h:2798, 2813, 2868-2895): out of pass 1, their silhouette edges drawn interpolated from
            # the vertex colours (a generator of its own: the stream `rs` of the scenes drawn before round 5 stays what it was)
            rs_mixed = np.random.RandomState(500000 + it)
            unshaded = np.flatnonzero(s.textured)[rs_mixed.rand(int(s.textured.sum())) < 0.4]
            s.shaded = s.shaded.copy()
            s.shaded[unshaded] = False
            s.colors[s.faces[unshaded].ravel()] = np.random.RandomState(600000 + 10 * it + v).rand(3 * unshaded.size, 3)
        if textured == 0.0 and it % 2 == 0:
            s.texture = np.zeros((0, 0))  # a scene WITHOUT a texture: the fit step's forward raster back-propagates the tiles with edges itself
        views.append(s)
    desc = (f"H={H} W={W} n_tri={n_tri} views={n_views} sigma={sigma} dt={dt} textured={textured} unshaded={int((views[0].textured & ~views[0].shaded).sum())} no_texture={np.size(views[0].texture) == 0} tex={tex_size} min_area={min_area} round={rounded} "
            f"bgcolor={background_color is not None} strict={strict} intpix={views[0].integer_pixel_centers} cw={bool(it & 1)}")
    obs = rs.rand(n_views, H, W, 3)
    return views, sigma, dt, desc, obs


def main(n):
    ref = api.ref() or api.port()
    fixed = api.ref(fixed=True) or api.port(fixed=True)
    worst = dict(image=0.0, ij_b=0.0, colors_b=0.0, uv_b=0.0, texture_b=0.0, flips=0)
    rs = np.random.RandomState(12345)
    misses = on_the_line = 0
    for it in range(n):
        views, sigma, dt, desc, obs_host = draw_scene(it, rs)
        n_views, H, W = len(views), views[0].height, views[0].width
        textured = views[0].textured.any()
        ds = device_scene(views, dt)
        r = HipRasterizer.for_scene(ds)
        obs = torch.as_tensor(obs_host, device=ds.device, dtype=dt)
        image, z, g = r.render_fit(ds, obs, sigma, check_overflow=True, clear_grads=True)
        image2, z2 = r.render(ds, sigma)
        g2 = r.render_backward(ds, residual_obs=obs)
        torch.cuda.synchronize()
        assert torch.equal(image, image2) and torch.equal(z, z2), (it, "fit frame != forward-only frame")
        tol_img, tol = (1e-9, 1e-8) if dt == torch.float64 else (1e-5, 1e-4)
        uv_ref = tex_ref = 0
        unchecked = False
        before = dict(worst)
        for i, s in enumerate(views):
            img_ref, z_ref = ref.render(s, sigma)
            e = np.abs(image[i].cpu().numpy() - img_ref).max()
            worst["image"] = max(worst["image"], e / tol_img)
            worst["flips"] += int((np.isinf(z[i].cpu().numpy()) != np.isinf(z_ref)).sum())
            image_b = 2 * (image[i].cpu().numpy().astype(np.float64) - obs[i].cpu().numpy().astype(np.float64))
            if "round=True" in desc and sigma > 0:
                # a pixel centre exactly on the line of a silhouette edge: the reference un-blends with a division by T = 0 (NaN) or
                # T ~ 1e-15 (rounding noise / T) -- its gradient is not a checker there; the HIP adjoint must still be finite
                api.min_abs_T()
                api.port().grads(s, sigma, img_ref.copy(), z_ref, image_b.copy())
                if api.min_abs_T() < 1e-9:
                    assert all(bool(torch.isfinite(g[k][i]).all()) for k in ("ij_b", "colors_b")), (it, i, "non-finite gradient")
                    on_the_line += 1
                    unchecked = True
                    continue
            g_ref, g_fix = ref.grads(s, sigma, img_ref, z_ref, image_b), fixed.grads(s, sigma, img_ref, z_ref, image_b)
            for k in ("ij_b", "colors_b"):
                worst[k] = max(worst[k], rel_err(g[k][i].cpu().numpy(), g_ref[k]) / tol, rel_err(g2[k][i].cpu().numpy(), g_ref[k]) / tol)
            uv_ref = uv_ref + g_ref["uv_b"]
            tex_ref = tex_ref + g_fix["texture_b"]
        if textured and not unchecked:
            worst["uv_b"] = max(worst["uv_b"], rel_err(g["uv_b"].cpu().numpy(), uv_ref) / tol)
            if g["texture_b"] is not None and np.abs(tex_ref).max() > 0:
                worst["texture_b"] = max(worst["texture_b"], rel_err(g["texture_b"].cpu().numpy(), tex_ref) / tol)
        if any(worst[k] > 1 and worst[k] > before[k] for k in worst if k != "flips") or worst["flips"] > before["flips"]:
            print(f"MISS it={it} {desc}: " + str({k: (round(float(worst[k]), 2)) for k in worst}), flush=True)
            misses += 1
            worst.update(before)
    print(f"({on_the_line} views with a pixel centre on an edge line: gradients only checked for finiteness)")
    print(f"{n} random scenes, {misses} missed; worst error / tolerance of the others:", {k: (round(float(v), 3) if k != "flips" else v) for k, v in worst.items()})
    return misses


def draw_mesh_scene(it):
    """Scene number `it` of the mesh sweep: views of a bumpy sphere (shared vertices, silhouette edges only), zoomed so that it may
    leave the frame; -> (views, sigma, pixel dtype, mode, description)."""
    rs = np.random.RandomState(7000 + it)
    H, W = int(rs.choice([32, 64, 100, 150, 256])), int(rs.choice([32, 64, 100, 150, 256]))
    nu, n_rings = [(6, 5), (12, 10), (24, 16), (40, 30)][rs.randint(4)]
    nb_colors = int(rs.choice([1, 2, 3, 4, 6]))
    textured = bool(rs.rand() < 0.4)
    n_views = int(rs.choice([1, 2, 4]))
    sigma = float(rs.choice([0.0, 0.5, 1.0, 2.0]))
    dt = torch.float64 if rs.rand() < 0.5 else torch.float32
    mode = ["image", "image", "error", "persp", "noculling"][rs.randint(5)]  # the last two: forward only, like the reference
    zoom, fov = float(rs.choice([0.7, 1.0, 1.5])), float(rs.choice([30.0, 60.0]))
    vertices, faces = scenes.bumpy_sphere(nu, n_rings, bump=float(rs.rand() * 0.3))
    tilt, views, clockwise, texture_size = rs.rand(2), [], None, int(rs.choice([8, 32]))
    for v in range(n_views):
        rot = scenes.rotx(0.2 + tilt[0]) @ scenes.roty(0.1 + tilt[1] + 0.4 * v)
        s = scenes.mesh_scene(vertices, faces, W, H, nb_colors=nb_colors, rot=rot, fov=fov, seed=it, textured=textured, texture_size=texture_size,
                              clockwise=clockwise)
        clockwise = s.clockwise
        centre = np.array([W / 2.0, H / 2.0])
        s.ij = centre + (s.ij - centre) * zoom
        s.integer_pixel_centers = bool(it % 3 != 1)
        s.perspective_correct = mode == "persp"
        s.backface_culling = mode != "noculling"
        if textured and it % 3 == 2:  # a third of the textured meshes: bands of textured && !shaded triangles (H.h:2798, 2868-2895), coloured vertices
            s.shaded = s.shaded.copy()
            s.shaded[(np.arange(s.shaded.size) // 7) % 3 == 0] = False
            s.colors = np.random.RandomState(800000 + 10 * it + v).rand(*s.colors.shape)
        views.append(s)
    desc = (f"mesh H={H} W={W} sphere={nu}x{n_rings} C={nb_colors} textured={textured} unshaded={int((views[0].textured & ~views[0].shaded).sum())} views={n_views} sigma={sigma} dt={dt} mode={mode} zoom={zoom} "
            f"fov={fov} intpix={views[0].integer_pixel_centers}")
    return views, sigma, dt, mode, desc


def main_meshes(n):
    """image mode: as main(); error mode: antialiase_error forward + adjoint against the repaired checker; persp / noculling modes
    (perspective_correct=True / backface_culling=False): forward only, the reference has no adjoint for them"""
    ref = api.ref() or api.port()
    fixed = api.ref(fixed=True) or api.port(fixed=True)
    worst = dict(image=0.0, err_buffer=0.0, ij_b=0.0, colors_b=0.0, shade_b=0.0, uv_b=0.0, texture_b=0.0, flips=0)
    misses = 0
    for it in range(n):
        views, sigma, dt, mode, desc = draw_mesh_scene(it)
        rs = np.random.RandomState(9000 + it)
        n_views, H, W, C = len(views), views[0].height, views[0].width, views[0].nb_colors
        ds = device_scene(views, dt)
        r = HipRasterizer.for_scene(ds)
        obs_host = rs.rand(n_views, H, W, C)
        err_b_host = rs.rand(n_views, H, W)
        obs = torch.as_tensor(obs_host, device=ds.device, dtype=dt)
        tol_img, tol = (1e-9, 1e-8) if dt == torch.float64 else (1e-5, 1e-4)
        before = dict(worst)
        sums = dict(uv_b=0, texture_b=0)
        if mode == "image":
            image, z, g = r.render_fit(ds, obs, sigma, check_overflow=True, clear_grads=True)
            g2 = r.render_backward(ds, residual_obs=obs)
        elif mode == "error":
            image, z, err = r.render(ds, sigma, True, obs, check_overflow=True)
            g = g2 = r.render_backward(ds, err_buffer_b=torch.as_tensor(err_b_host, device=ds.device, dtype=dt))
        else:
            image, z = r.render(ds, sigma, check_overflow=True)
            g = g2 = None
        torch.cuda.synchronize()
        for i, s in enumerate(views):
            out_ref = ref.render(s, sigma, mode == "error", obs_host[i] if mode == "error" else None)
            img_ref, z_ref = out_ref[0], out_ref[1]
            worst["image"] = max(worst["image"], np.abs(image[i].cpu().numpy() - img_ref).max() / tol_img)
            worst["flips"] += int((np.isinf(z[i].cpu().numpy()) != np.isinf(z_ref)).sum())
            if mode in ("persp", "noculling"):
                continue
            if mode == "error":
                worst["err_buffer"] = max(worst["err_buffer"], np.abs(err[i].cpu().numpy() - out_ref[2]).max() / (10 * tol_img * max(1.0, out_ref[2].max())))
                g_fix = fixed.grads(s, sigma, img_ref, z_ref, None, True, obs_host[i], out_ref[2], err_b_host[i])
                g_ref = g_fix  # defect D2 touches colors_b: every gradient against the repaired checker
            else:
                image_b = 2 * (image[i].cpu().numpy().astype(np.float64) - obs[i].cpu().numpy().astype(np.float64))
                g_ref, g_fix = ref.grads(s, sigma, img_ref, z_ref, image_b), fixed.grads(s, sigma, img_ref, z_ref, image_b)
            for k in ("ij_b", "colors_b", "shade_b"):
                for gg in (g, g2):
                    if np.abs(g_ref[k]).max() > 0 or float(gg[k][i].abs().max()) > 0:
                        worst[k] = max(worst[k], rel_err(gg[k][i].cpu().numpy(), g_ref[k]) / tol)
            sums["uv_b"] = sums["uv_b"] + g_ref["uv_b"]
            sums["texture_b"] = sums["texture_b"] + g_fix["texture_b"]
        if mode in ("image", "error") and views[0].textured.any():
            for k in ("uv_b", "texture_b"):
                if g[k] is not None and np.abs(sums[k]).max() > 0:
                    worst[k] = max(worst[k], rel_err(g[k].cpu().numpy(), sums[k]) / tol)
        if any(worst[k] > 1 and worst[k] > before[k] for k in worst if k != "flips") or worst["flips"] > before["flips"]:
            print(f"MISS it={it} {desc}: " + str({k: (round(float(worst[k]), 2)) for k in worst}), flush=True)
            misses += 1
            worst.update(before)
    print(f"{n} random mesh scenes, {misses} missed; worst error / tolerance of the others:", {k: (round(float(v), 3) if k != "flips" else v) for k, v in worst.items()})
    return misses


def draw_large_scene(it):
    """Scene number `it` of the large sweep: thousands of small soup triangles (all edges flagged) on frames of 512 .. 1024 pixels,
    up to 8 views: deep tile lists, every class of edge list, sweep slots, spill pools.  -> (views, sigma, dtype, description)"""
    rs = np.random.RandomState(11000 + it)
    H, W = int(rs.choice([512, 777, 1024])), int(rs.choice([512, 640, 1024]))
    n_tri = int(rs.choice([2000, 8000, 20000]))
    n_views = int(rs.choice([1, 3, 8]))
    sigma = float(rs.choice([0.5, 1.0, 3.0]))
    dt = torch.float64 if rs.rand() < 0.5 else torch.float32
    textured = float(rs.choice([0.0, 0.5]))
    shrink = float(rs.choice([0.04, 0.1, 0.3]))
    strict = bool(rs.rand() < 0.7)
    base = scenes.soup_scene(n_tri=n_tri, width=W, height=H, seed=it, clockwise=bool(it & 1), textured_ratio=textured, flat=False, texture_size=32, min_area=50.0)
    tri = base.ij.reshape(-1, 3, 2)
    base.ij = (tri.mean(axis=1, keepdims=True) + (tri - tri.mean(axis=1, keepdims=True)) * shrink).reshape(-1, 2)
    views = []
    for v in range(n_views):
        s = scenes.soup_scene(n_tri=1, width=W, height=H, seed=it)  # a container of the right type, filled from `base` below
        for name in ("faces", "faces_uv", "depths", "textured", "uv", "shade", "colors", "shaded", "edgeflags", "texture", "background_image", "clockwise"):
            setattr(s, name, getattr(base, name))
        s.ij = base.ij + (rs.randn(*base.ij.shape) * 1.5 if v else 0.0)
        s.strict_edge = strict
        for name in ("uv_b", "ij_b", "shade_b", "colors_b", "texture_b"):
            setattr(s, name, np.zeros(np.shape(getattr(s, name[:-2]))))
        views.append(s)
    if textured == 0.0:
        for s in views:
            s.texture = s.texture_b = np.zeros((0, 0))  # (untextured: fused edge tiles, split tiles, tile pairs -- these frames have a head of the list)
    desc = f"large H={H} W={W} n_tri={n_tri} shrink={shrink} views={n_views} sigma={sigma} dt={dt} textured={textured} strict={strict} cw={bool(it & 1)}"
    return views, sigma, dt, desc


def main_large(n):
    ref = api.ref() or api.port()
    fixed = api.ref(fixed=True) or api.port(fixed=True)
    worst = dict(image=0.0, ij_b=0.0, colors_b=0.0, shade_b=0.0, uv_b=0.0, texture_b=0.0, flips=0)
    misses = 0
    for it in range(n):
        views, sigma, dt, desc = draw_large_scene(it)
        n_views, H, W = len(views), views[0].height, views[0].width
        ds = device_scene(views, dt)
        r = HipRasterizer.for_scene(ds)
        obs_host = np.random.RandomState(12000 + it).rand(n_views, H, W, 3)
        obs = torch.as_tensor(obs_host, device=ds.device, dtype=dt)
        image, z, g = r.render_fit(ds, obs, sigma, check_overflow=True, clear_grads=True)
        image2, z2 = r.render(ds, sigma)
        g2 = r.render_backward(ds, residual_obs=obs)
        torch.cuda.synchronize()
        assert torch.equal(image, image2) and torch.equal(z, z2), (it, "fit frame != forward-only frame")
        tol_img, tol = (1e-9, 1e-8) if dt == torch.float64 else (1e-5, 1e-4)
        before = dict(worst)
        sums = dict(uv_b=0, texture_b=0)
        for i, s in enumerate(views):
            img_ref, z_ref = ref.render(s, sigma)
            hip_image, hip_z = image[i].cpu().numpy().astype(np.float64), z[i].cpu().numpy()
            # float32 frames: a pixel whose two nearest triangles differ in depth by less than the rounding of the stored z may
            # change owner (float32 note); those are counted, the rest is compared
            flipped = np.isinf(hip_z) != np.isinf(z_ref)
            worst["flips"] += int(flipped.sum())
            # (relative to the value where it is large: a sliver left by the perturbation of a shrunken triangle extrapolates its colours to +- 1 000 over
            # the pixels it still covers, and a float32 frame holds those to 6e-5 -- scene 91, tests/fuzz_debug_large.py)
            worst["image"] = max(worst["image"], (np.abs(hip_image - img_ref) / np.maximum(1.0, np.abs(img_ref))).max() / tol_img)
            image_b = 2 * (hip_image - obs[i].cpu().numpy().astype(np.float64))
            g_ref, g_fix = ref.grads(s, sigma, img_ref, z_ref, image_b), fixed.grads(s, sigma, img_ref, z_ref, image_b)
            for k in ("ij_b", "colors_b", "shade_b"):
                for gg in (g, g2):
                    if np.abs(g_ref[k]).max() > 0 or float(gg[k][i].abs().max()) > 0:
                        worst[k] = max(worst[k], rel_err(gg[k][i].cpu().numpy(), g_ref[k]) / tol)
            sums["uv_b"] = sums["uv_b"] + g_ref["uv_b"]
            sums["texture_b"] = sums["texture_b"] + g_fix["texture_b"]
        if views[0].textured.any():
            for k in ("uv_b", "texture_b"):
                worst[k] = max(worst[k], rel_err(g[k].cpu().numpy(), sums[k]) / tol)
        if any(worst[k] > 1 and worst[k] > before[k] for k in worst if k != "flips") or worst["flips"] > before["flips"]:
            print(f"MISS it={it} {desc}: " + str({k: (round(float(worst[k]), 2)) for k in worst}), flush=True)
            misses += 1
            worst.update(before)
        else:
            print(f"ok   it={it} {desc}", flush=True)
    print(f"{n} large scenes, {misses} missed; worst error / tolerance of the others:", {k: (round(float(v), 3) if k != "flips" else v) for k, v in worst.items()})
    return misses


if __name__ == "__main__":
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    which = sys.argv[2] if len(sys.argv) > 2 else "both"
    missed = (main(count) if which in ("both", "soups") else 0) + (main_meshes(count) if which in ("both", "meshes") else 0)
    missed += main_large(count) if which == "large" else 0
    sys.exit(1 if missed else 0)
