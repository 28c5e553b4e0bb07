"""Gradients of one fit step by the fused (staged) path, the un-staged path (deodr_hip_force_generic) and the deterministic mode against one
another, for 1 / 8 views of the bumpy sphere at 512^2 / 256^2: which path is off when two disagree (e.g. a library variant given with --lib).
Run on the GPU box:  python tools/diag_paths.py [--lib tools/variants/libdeodr_hip_x.so]"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from deodr_amd import scenes, hip_renderer as hr
if "--lib" in sys.argv:
    hr.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from hip_util import device_scene, rel_err
from deodr_amd.hip_renderer import HipRasterizer
F64 = torch.float64
for n_views, size in ((1, 512), (8, 512), (8, 256)):
    views = [scenes.sphere_scene(size=size, angle=float(a)) for a in np.linspace(-0.3, 0.3, n_views)]
    ds = device_scene(views, F64)
    r = HipRasterizer.for_scene(ds)
    obs = torch.as_tensor(np.random.RandomState(6).rand(n_views, size, size, 4), device=ds.device)
    res = {}
    _, _, g = r.render_fit(ds, obs, 1.0, check_overflow=True, clear_grads=True); torch.cuda.synchronize()
    res["fused"] = {k: v.clone() for k, v in g.items() if v is not None}
    hr.force_generic(True)
    _, _, g = r.render_fit(ds, obs, 1.0, check_overflow=True, clear_grads=True); torch.cuda.synchronize()
    res["generic"] = {k: v.clone() for k, v in g.items() if v is not None}
    hr.force_generic(False)
    hr.set_deterministic(True)
    _, _, g = r.render_fit(ds, obs, 1.0, check_overflow=True, clear_grads=True); torch.cuda.synchronize()
    res["det"] = {k: v.clone() for k, v in g.items() if v is not None}
    hr.set_deterministic(False)
    for a, b in (("fused", "generic"), ("det", "generic"), ("fused", "det")):
        for k in ("ij_b", "colors_b"):
            d = (res[a][k] - res[b][k]).abs()
            print(n_views, size, a, "vs", b, k, "rel", float(d.max() / res[b][k].abs().max()), "abs", float(d.max()), "at view", int(d.flatten(1).max(1)[0].argmax()) if d.dim() == 3 else -1)
