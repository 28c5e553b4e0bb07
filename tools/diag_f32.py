"""fit step of the float32 instances against the two-call path (double owner adjoint): worst gradient differences (a library variant with --lib)"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from deodr_amd import scenes, hip_renderer as hr
if "--lib" in sys.argv:
    hr.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from hip_util import device_scene
from deodr_amd.hip_renderer import HipRasterizer
for n_views, size, nu in ((1, 1024, 100), (2, 1024, 100), (1, 256, 40), (1, 200, 30)):
    views = [scenes.sphere_scene(size=size, nu=nu, n_rings=nu, angle=float(a)) for a in np.linspace(-0.3, 0.3, n_views)]
    for v in views:
        v.texture = np.zeros((0, 0))
    ds = device_scene(views, torch.float32)
    r = HipRasterizer.for_scene(ds)
    obs = torch.as_tensor(np.random.RandomState(6).rand(n_views, size, size, 4), device=ds.device, dtype=torch.float32)
    _, _, g = r.render_fit(ds, obs, 1.0, check_overflow=True, clear_grads=True)
    g = {k: v.clone() for k, v in g.items() if v is not None}
    r.render(ds, 1.0)
    g2 = r.render_backward(ds, residual_obs=obs)
    torch.cuda.synchronize()
    for k in ("ij_b", "colors_b"):
        d = (g[k] - g2[k]).abs()
        bad = (d > 1e-5 * g2[k].abs().max()).any(-1)
        print(os.path.basename(hr.LIB_PATH), n_views, size, k, "rel", float(d.max() / g2[k].abs().max()), "vertices off:", int(bad.sum()), "of", bad.numel(), flush=True)
