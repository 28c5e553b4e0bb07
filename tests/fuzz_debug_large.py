"""Where a scene of the LARGE sweep of tests/fuzz_parity.py leaves the checker (GPU box):  python tests/fuzz_debug_large.py <it> [--lib other.so]
Per view: the worst pixels of the fit step's frame against the CPU checker, their tile, the checker's owner there and the number of triangles / flagged
edges whose bounding boxes (grown by sigma) reach the pixel; the same frame from the forward-only call and from the un-staged kernels."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import deodr_amd.hip_renderer as hr  # noqa: E402

if "--lib" in sys.argv:
    hr.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from deodr_amd.hip_renderer import HipRasterizer  # noqa: E402
from fuzz_parity import draw_large_scene  # noqa: E402
from hip_util import device_scene  # noqa: E402
from oracle import api  # noqa: E402

ref = api.ref() or api.port()
it = int(sys.argv[1])
views, sigma, dt, desc = draw_large_scene(it)
print(desc, "lib", os.path.basename(hr.LIB_PATH))
n_views, H, W = len(views), views[0].height, views[0].width
ds = device_scene(views, dt)
r = HipRasterizer.for_scene(ds)
obs = torch.as_tensor(np.random.RandomState(12000 + it).rand(n_views, H, W, 3), device=ds.device, dtype=dt)
image, z, g = r.render_fit(ds, obs, sigma, check_overflow=True, clear_grads=True)
hr.force_generic(True)
try:
    image_g, z_g = r.render(ds, sigma)
finally:
    hr.force_generic(False)
torch.cuda.synchronize()
tol = 1e-9 if dt == torch.float64 else 1e-5
for i, s in enumerate(views):
    img_ref, z_ref = ref.render(s, sigma)
    a = image[i].cpu().numpy().astype(np.float64)
    b = image_g[i].cpu().numpy().astype(np.float64)
    d = np.abs(a - img_ref).max(axis=2)
    dg = np.abs(b - img_ref).max(axis=2)
    print(f"view {i}: staged fit step max {d.max() / tol:.2f} tol at {np.unravel_index(d.argmax(), d.shape)}, {int((d > tol).sum())} pixels over; un-staged forward max {dg.max() / tol:.2f} tol, {int((dg > tol).sum())} over")
    ij = np.asarray(s.ij).reshape(-1, 2)
    for y, x in zip(*np.nonzero(d > tol)):
        tri = ij[np.asarray(s.faces).reshape(-1, 3)]  # [T, 3, 2] (x, y)
        lo, hi = tri.min(axis=1) - sigma - 1, tri.max(axis=1) + sigma + 1
        near = np.nonzero((lo[:, 0] <= x) & (x <= hi[:, 0]) & (lo[:, 1] <= y) & (y <= hi[:, 1]))[0]
        print(f"   pixel (row {y}, col {x}) tile ({y // 8}, {x // 8}): staged {a[y, x]} checker {img_ref[y, x]} un-staged {b[y, x]}; z staged {float(z[i, y, x]):.6f} checker {z_ref[y, x]:.6f};"
              f" {len(near)} triangles near: {near[:12].tolist()}")
        if len(near) <= 6:
            for k in near:
                print("      triangle", int(k), "vertices", tri[k].round(4).tolist(), "depths", np.asarray(s.depths)[np.asarray(s.faces).reshape(-1, 3)[k]].round(4).tolist())
