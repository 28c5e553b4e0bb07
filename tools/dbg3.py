import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from deodr_amd.mesh_fitter import MeshDepthFitter
from test_scene3d import depth_inputs, hand
d, depth_image = depth_inputs()
vertices, faces = hand()
for rep in range(12):
    fitter = MeshDepthFitter(vertices, faces, d["euler_init"], d["translation_init"], cregu=1000)
    fitter.set_image(depth_image, focal=241, distortion=d["distortion"]); fitter.set_max_depth(1); fitter.set_depth_scale(float(d["depth_scale"]))
    e = np.array([fitter.step()[0] for _ in range(50)])
    dev = np.abs(e - d["energies"])
    print(rep, "final", e[49], "max dev", dev.max(), "at", int(dev.argmax()), "first it with dev>1e-8:", int(np.argmax(dev > 1e-8)) if (dev > 1e-8).any() else -1)
