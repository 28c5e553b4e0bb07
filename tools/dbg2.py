import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from deodr_amd.mesh_fitter import MeshRGBFitterWithPose, MeshRGBFitterWithPoseMultiFrame
d = np.load("tests/golden/rgb_hand_fit.npz"); faces = np.load("tests/golden/hand_mesh.npz")["faces"].astype(np.int64)
image_obs = d["image_u8"].astype(np.float64) / 255
args = (d["default_color"], d["default_light_directional"], float(d["default_light_ambient"]))
n = 3
multi = MeshRGBFitterWithPoseMultiFrame(d["vertices_centered"], faces, np.zeros((n, 3)), np.tile(d["translation_init"], (n, 1)), *args, cregu=1000)
multi.set_background_color(d["background_color"]); multi.set_images([image_obs] * n)
single = MeshRGBFitterWithPose(d["vertices_centered"], faces, np.zeros(3), d["translation_init"], *args, cregu=1000)
single.set_background_color(d["background_color"]); single.set_image(image_obs)
for f, name in ((multi, "multi"), (single, "single")):
    leaves = f._leaves(f._appearance_leaves())
    image = f.render()
    e, _ = f._data_energy(image)
    g = torch.autograd.grad(e, leaves)
    print(name, float(e), [float(x.abs().sum()) for x in g])
    if name == "multi": gm = g
    else: gs = g
dv = (gm[0] - 3 * gs[0]).abs()
print("g_v diff max", float(dv.max()), "at", int(dv.max(dim=1).values.argmax()), "scale", float(gs[0].abs().max()))
lm, ls = multi.scene.last, single.scene.last
print("ij equal", float((lm["ij"][0] - ls["ij"][0]).abs().max()), float((lm["ij"][2] - ls["ij"][0]).abs().max()), "flags", int((lm["edgeflags"][1] != ls["edgeflags"][0]).sum()))
