"""Speed of the background fill alone: forward-only render of a frame whose only triangle covers one tile, so that the call is the
fill kernel (+ trivial set-up / scan / raster).  python tools/fill_probe.py [--lib variant.so]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deodr_amd import hip_renderer as hr
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer
if "--lib" in sys.argv:
    hr.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
dev = torch.device("cuda:0")
for S, B, C in ((1024, 8, 4), (1024, 8, 3), (2048, 8, 3), (1024, 1, 4)):
    faces = np.array([[0, 1, 2]]); ij = np.tile(np.array([[2.0, 2.0], [6.0, 2.5], [3.0, 6.0]]), (B, 1, 1))
    ds = DeviceScene(faces, faces, np.zeros(1), np.zeros(1), np.zeros((3, 2)), ij, np.ones((B, 3)), np.random.rand(B, 3, C), np.ones((B, 3)),
                     np.zeros((B, 1, 3)), S, S, background_color=np.linspace(0.1, 0.9, C), clockwise=False, pixel_dtype=torch.float32, device=dev)
    r = HipRasterizer.for_scene(ds)
    image = torch.empty((B, S, S, C), dtype=torch.float32, device=dev); z = torch.empty((B, S, S), dtype=torch.float32, device=dev)
    r.render(ds, 1.0, out=(image, z), check_overflow=True)
    for _ in range(5):
        r.render(ds, 1.0, out=(image, z), check_overflow=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        r.render(ds, 1.0, out=(image, z), check_overflow=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    nbytes = B * S * S * (C + 2) * 4  # image + z + owner ids
    print(f"{os.path.basename(hr.LIB_PATH)} {B} x {S}^2 C={C}: {dt*1e6:.1f} us per forward-only call, {nbytes/dt/1e9:.0f} GB/s of frame stores;"
          f" bg ok: {bool((image[0, -1, -1].cpu().numpy() == np.linspace(0.1, 0.9, C).astype(np.float32)).all())}, z inf: {bool(torch.isinf(z[0, -1, -1]))}")
