"""The reference's own micro-benchmark (tests/benchmark_rendering.py:11-26: median forward time of a 500 x 500, 200-triangle
untextured soup, sigma = 0, 1000 repetitions) on this stack: through the drop-in `renderSceneCpp` (NumPy buffers in and out: the
PCIe copies are part of the call, as the host copies are part of the reference's) and device-resident (`HipRasterizer.render`).
Run on the GPU box:  python tools/benchmark_rendering.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deodr_amd import hip_renderer as hr  # noqa: E402
from deodr_amd import scenes  # noqa: E402
from deodr_amd.hip_renderer import DeviceScene, HipRasterizer  # noqa: E402

s = scenes.soup_scene(n_tri=200, width=500, height=500, seed=2, clockwise=True, textured_ratio=0.0, flat=False)
image, z_buffer = np.zeros((s.height, s.width, s.nb_colors)), np.zeros((s.height, s.width))
hr.renderSceneCpp(s, 0, image, z_buffer)
durations = []
for _ in range(200):
    image.fill(0)
    z_buffer.fill(0)
    t0 = time.perf_counter_ns()
    hr.renderSceneCpp(s, 0, image, z_buffer)
    durations.append(time.perf_counter_ns() - t0)
print(f"drop-in renderSceneCpp (float64 NumPy buffers, host <-> device copies included): median {np.median(durations) / 1e6:.3f} ms")

ds = DeviceScene(s.faces, s.faces_uv, s.textured, s.shaded, s.uv, s.ij[None], s.depths[None], s.colors[None], s.shade[None], s.edgeflags[None],
                 s.height, s.width, background_image=s.background_image[None], clockwise=True, pixel_dtype=torch.float32)
r = HipRasterizer.for_scene(ds)
out = r.render(ds, 0.0, check_overflow=True)
durations = []
for _ in range(1000):
    torch.cuda.synchronize()
    t0 = time.perf_counter_ns()
    r.render(ds, 0.0, out=out, check_overflow=False)
    torch.cuda.synchronize()
    durations.append(time.perf_counter_ns() - t0)
print(f"device-resident HipRasterizer.render (float32 frame, synchronised every call): median {np.median(durations) / 1e6:.4f} ms")
