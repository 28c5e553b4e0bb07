"""The fit step of the bench workload on two builds of the library: images, depths and gradients of one against the other (for measurement builds whose
results must not change):  python tools/lib_compare.py --lib A.so --out a.npz ; python tools/lib_compare.py --lib B.so --out b.npz ; python tools/lib_compare.py a.npz b.npz"""
import sys, os
import numpy as np
if len(sys.argv) == 3 and sys.argv[1].endswith(".npz"):
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    worst = 0.0
    for k in a.files:
        d = float(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max() / max(float(np.abs(a[k]).max()), 1e-30))
        worst = max(worst, d)
        print(f"{k}: max difference / max value {d:.2e}")
    sys.exit(0 if worst < 1e-5 else 1)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deodr_amd import scenes, hip_renderer as hr
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer
arg = lambda name, default: type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default
if "--lib" in sys.argv:
    hr.LIB_PATH = os.path.abspath(arg("--lib", ""))
B, S = arg("--views", 8), arg("--size", 1024)
dev = torch.device("cuda:0")
views = [scenes.sphere_scene(size=S, angle=float(a)) for a in np.linspace(-0.5, 0.5, B)]
s0 = views[0]
stack = lambda n: np.stack([np.asarray(getattr(v, n)) for v in views])
ds = DeviceScene(s0.faces, s0.faces_uv, s0.textured, s0.shaded, s0.uv, stack("ij"), stack("depths"), stack("colors"), stack("shade"),
                 stack("edgeflags"), S, S, texture=None, background_color=s0.background_color, clockwise=s0.clockwise,
                 vertex_dtype=torch.float64, pixel_dtype=torch.float32, device=dev)
r = HipRasterizer.for_scene(ds)
C = ds.nb_colors
obs = torch.from_numpy(np.random.RandomState(0).rand(B, S, S, C).astype(np.float32)).to(dev)
image = torch.empty((B, S, S, C), dtype=torch.float32, device=dev)
z = torch.empty((B, S, S), dtype=torch.float32, device=dev)
grads = ds.zero_grads()
for _ in range(3):  # (the third step: counters and lists of the steps before it have been re-used)
    r.render_fit(ds, obs, 1.0, grads=grads, out=(image, z), check_overflow=True, clear_grads=True)
torch.cuda.synchronize()
zz = z.cpu().numpy()
np.savez(arg("--out", "out.npz"), image=image.cpu().numpy(), z=np.where(np.isfinite(zz), zz, 0.0), **{k: v.cpu().numpy() for k, v in grads.items() if torch.is_tensor(v)})
print("saved", arg("--out", "out.npz"))
