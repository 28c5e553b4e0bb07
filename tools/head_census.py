"""Entries of the forward's work list of view 0 -- head (many-primitive tiles / tiles with edges, walked by one workgroup in heavy_share) and
the rest -- for the bench workload and for configs[4]: how many entries a head walker walks one after the other.  GPU box.
    python tools/head_census.py [--lib path]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deodr_amd import hip_renderer as hr
from deodr_amd import scenes
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer

if "--lib" in sys.argv:
    hr.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
dev = torch.device("cuda:0")


def census(name, views):
    s0 = views[0]
    stack = lambda n: np.stack([np.asarray(getattr(v, n)) for v in views])
    tex = s0.texture if np.size(s0.texture) else None
    ds = DeviceScene(s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"),
                     stack("edgeflags"), s0.height, s0.width, texture=tex, background_color=s0.background_color, clockwise=s0.clockwise,
                     vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev)  # fmt: skip
    r = HipRasterizer.for_scene(ds)
    n, H, W, C = ds.n_views, ds.height, ds.width, ds.nb_colors
    obs = torch.rand((n, H, W, C), dtype=torch.float32, device=dev)
    for fit in (False, True):
        if fit:
            r.render_fit(ds, obs, 1.0, clear_grads=True)
        else:
            r.render(ds, 1.0, check_overflow=True)
        torch.cuda.synchronize()
        words = r.workspace.view(torch.uint8)[:64].cpu().numpy().view(np.uint32)  # WsHeader of view 0 (dr_workspace.h)
        print(f"{name}, {n} view(s), {'fit step' if fit else 'render'}: head entries {words[13]}, other entries {words[14]}, tiles {(H // 8) * (W // 8)}, census {hr.tile_census(r, ds)}", flush=True)


for nv in (1, 8):
    census("configs[2] sphere 20k", [scenes.sphere_scene(size=1024, angle=float(a)) for a in np.linspace(-0.5, 0.5, nv)])
big = dict(size=2048, nu=224, n_rings=224, nb_colors=3, textured=True, texture_size=1024)
for nv in (1, 8):
    census("configs[4] shape", [scenes.sphere_scene(angle=float(a), **big) for a in np.linspace(-0.5, 0.5, nv)])
