import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from deodr_amd import scenes
from hip_util import device_scene
from deodr_amd.hip_renderer import HipRasterizer
from oracle import api
big = dict(size=256, nu=40, n_rings=40, nb_colors=3, textured=True, texture_size=64)
views = [scenes.sphere_scene(angle=float(a), **big) for a in np.linspace(-0.5, 0.5, 8)]
ref = api.ref()
for mode in ("batch", "single"):
    for i, s in enumerate(views):
        if mode == "batch" and i > 0: break
        ds = device_scene(views if mode == "batch" else s, torch.float32)
        r = HipRasterizer.for_scene(ds)
        image, z = r.render(ds, 1.0)
        torch.cuda.synchronize()
        for j in range(ds.n_views):
            sj = views[j] if mode == "batch" else s
            img_ref, z_ref = ref.render(sj, 1.0)
            d = np.abs(image[j].cpu().numpy() - img_ref).max(-1)
            bad = np.argwhere(d > 1e-5)
            print(mode, i, j, "max err", d.max(), "bad px", len(bad), bad[:5].tolist(), "zdiff", np.abs(np.where(np.isfinite(z_ref), z[j].cpu().numpy() - z_ref, 0)).max(), "finite mismatch", (np.isfinite(z[j].cpu().numpy()) != np.isfinite(z_ref)).sum())
            if len(bad):
                y, x = bad[0]
                print("   at", y, x, "got", image[j, y, x].cpu().numpy(), "ref", img_ref[y, x], "z", float(z[j, y, x]), z_ref[y, x])
