"""Shrinks a failing scene of tests/fuzz_parity.py (GPU box):  python tests/fuzz_debug.py <it> [<it> ...]

Re-creates scene number `it` of the sweep (first view), reports where the HIP frame leaves the CPU checker's, tries the same scene
with single switches changed, and delta-debugs the triangle list down to a small failing subset that it prints as a literal."""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from deodr_amd.hip_renderer import HipRasterizer  # noqa: E402
from fuzz_parity import draw_scene  # noqa: E402
from hip_util import device_scene  # noqa: E402
from oracle import api  # noqa: E402

ref = api.ref() or api.port()


def subset(s, idx):
    t = copy.copy(s)
    for name in ("faces", "faces_uv", "textured", "shaded", "edgeflags"):
        setattr(t, name, np.ascontiguousarray(getattr(s, name)[idx]))
    return t


def frames(s, sigma, dt):
    ds = device_scene([s], dt)
    image, z = HipRasterizer.for_scene(ds).render(ds, sigma)
    torch.cuda.synchronize()
    return image[0].cpu().numpy().astype(np.float64), z[0].cpu().numpy().astype(np.float64)


def wrong_pixels(s, sigma, dt):
    image, z = frames(s, sigma, dt)
    image_ref, z_ref = ref.render(s, sigma)
    tol = 1e-9 if dt == torch.float64 else 1e-5
    bad = (np.abs(image - image_ref).max(axis=-1) > tol) | (np.isinf(z) != np.isinf(z_ref))
    return bad, image, z, image_ref, z_ref


def adjoint_report(views, sigma, dt):
    """per view, alone: worst gradient entries of the fit step against the checker, with switches changed one at a time"""
    from hip_util import rel_err

    def grads(s, sg, edges=None):
        t = copy.copy(s)
        if edges is not None:
            t.edgeflags = np.zeros_like(s.edgeflags)
            t.edgeflags[:, edges] = s.edgeflags[:, edges]
        ds = device_scene([t], dt)
        r = HipRasterizer.for_scene(ds)
        obs = torch.as_tensor(np.random.RandomState(1).rand(1, t.height, t.width, 3), device=ds.device, dtype=dt)
        image, z, g = r.render_fit(ds, obs, sg, clear_grads=True)
        g = {k: v.cpu().numpy().astype(np.float64) for k, v in g.items() if v is not None}
        g2 = {k: v.cpu().numpy().astype(np.float64) for k, v in r.render_backward(ds, residual_obs=obs).items() if v is not None}
        torch.cuda.synchronize()
        image_ref, z_ref = ref.render(t, sg)
        image_b = 2 * (image[0].cpu().numpy().astype(np.float64) - obs[0].cpu().numpy().astype(np.float64))
        return g, g2, ref.grads(t, sg, image_ref, z_ref, image_b), api.port().grads(t, sg, image_ref, z_ref, image_b)

    for i, s in enumerate(views):
        g, g2, g_ref, g_port = grads(s, sigma)
        e = rel_err(g["ij_b"][0], g_ref["ij_b"])
        print(f" view {i}: ij_b rel err one-call {e:.3g}, two-call {rel_err(g2['ij_b'][0], g_ref['ij_b']):.3g}, restatement vs reference build "
              f"{rel_err(g_port['ij_b'], g_ref['ij_b']):.3g}; colors_b {rel_err(g['colors_b'][0], g_ref['colors_b']):.3g}")
        if e < 1e-7:
            continue
        np.set_printoptions(precision=10, suppress=True)
        print("   ij", s.ij[s.faces[0]].tolist() if len(s.faces) == 1 else "(many)")
        worst = np.argsort(-np.abs(g["ij_b"][0] - g_ref["ij_b"]).max(axis=1))[:3]
        for v in worst:
            print(f"   vertex {v} at {s.ij[v]}: hip {g['ij_b'][0][v]} ref {g_ref['ij_b'][v]}")
        for name, sg, edges in (("sigma=0", 0.0, None), ("edge 0 only", sigma, [0]), ("edge 1 only", sigma, [1]), ("edge 2 only", sigma, [2])):
            g, g2, g_ref, _ = grads(s, sg, edges)
            print(f"   {name}: ij_b rel err {rel_err(g['ij_b'][0], g_ref['ij_b']):.3g}  (hip {g['ij_b'][0][worst[0]]} ref {g_ref['ij_b'][worst[0]]})")
        t = copy.copy(s)
        t.ij = s.ij + 1e-3 * np.random.RandomState(2).randn(*s.ij.shape)
        g, g2, g_ref, _ = grads(t, sigma)
        print(f"   vertices moved by 1e-3: ij_b rel err {rel_err(g['ij_b'][0], g_ref['ij_b']):.3g}")


for it in [int(a) for a in sys.argv[1:]]:
    views, sigma, dt, desc, _ = draw_scene(it, np.random.RandomState(12345), replay=True)
    s = views[0]
    print(f"== it={it} {desc}")
    for sg in sorted({0.0, sigma}):
        bad, image, z, image_ref, z_ref = wrong_pixels(s, sg, dt)
        print(f" sigma={sg}: {int(bad.sum())} wrong pixels")
        for y, x in np.argwhere(bad)[:6]:
            print(f"   (y={y}, x={x}) hip z={z[y, x]:.6g} ref z={z_ref[y, x]:.6g} hip {np.round(image[y, x], 4)} ref {np.round(image_ref[y, x], 4)}")
    for name, change in (("strict_edge=True", dict(strict_edge=True)), ("untextured", dict(textured=np.zeros_like(s.textured), shaded=np.zeros_like(s.shaded))),
                         ("float64 frames", None), ("float32 frames", None)):
        t = copy.copy(s)
        for k, v in (change or {}).items():
            setattr(t, k, v)
        d = {"float64 frames": torch.float64, "float32 frames": torch.float32}.get(name, dt)
        print(f" {name}: {int(wrong_pixels(t, sigma, d)[0].sum())} wrong pixels")
    # delta debugging on the triangle list (forward frame only)
    idx = np.arange(len(s.faces))
    fails = lambda ii: len(ii) > 0 and wrong_pixels(subset(s, ii), sigma, dt)[0].any()
    if not fails(idx):
        print(" (the forward frame is right: the miss is in the adjoint)")
        adjoint_report(views, sigma, dt)
        continue
    chunk = max(len(idx) // 2, 1)
    renders = 0
    while chunk >= 1 and renders < 400:
        shrunk = False
        for a in range(0, len(idx), chunk):
            trial = np.concatenate([idx[:a], idx[a + chunk:]])
            renders += 1
            if fails(trial):
                idx, shrunk = trial, True
                break
        if not shrunk:
            chunk //= 2
    t = subset(s, idx)
    bad, image, z, image_ref, z_ref = wrong_pixels(t, sigma, dt)
    print(f" smallest failing subset: triangles {idx.tolist()}  ({int(bad.sum())} wrong pixels, {renders} renders)")
    for y, x in np.argwhere(bad)[:6]:
        print(f"   (y={y}, x={x}) hip z={z[y, x]:.17g} ref z={z_ref[y, x]:.17g} hip {image[y, x]} ref {image_ref[y, x]}")
    np.set_printoptions(precision=17)
    for k in idx:
        f = s.faces[k]
        print(f"   tri {k}: ij={s.ij[f].tolist()} depths={s.depths[f].tolist()} textured={bool(s.textured[k])} shaded={bool(s.shaded[k])}")
        print(f"          uv={s.uv[s.faces_uv[k]].tolist()} shade={s.shade[f].tolist()}")
    print(f"   H={s.height} W={s.width} sigma={sigma} clockwise={s.clockwise} strict={s.strict_edge} ipc={s.integer_pixel_centers} tex={s.texture.shape}")
