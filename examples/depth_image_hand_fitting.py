"""Fit the deformable hand mesh to a depth image -- the reference's deodr/examples/depth_image_hand_fitting.py:34-110 with everything
on the MI355X: parameters, camera (with distortion), silhouette flags, rasterizer, rigid energy and the momentum update; one
HIP-graph replay per iteration.

    python examples/depth_image_hand_fitting.py [--iterations 100] [--eager] [--save out.npz]
"""
import argparse

import numpy as np

from _common import golden, hand_mesh, run


def main(iterations=100, graph=True, save=None):
    from deodr_amd.mesh_fitter import GraphedStep, MeshDepthFitter

    d = golden("depth_hand_fit.npz")  # depth.bin of the reference cropped as its example does, camera and initial pose of the example
    depth = d["depth_raw_f32"].astype(np.float64)
    max_depth = float(d["max_depth"])
    depth[depth == 0] = max_depth
    vertices, faces = hand_mesh()
    fitter = MeshDepthFitter(vertices, faces, d["euler_init"], d["translation_init"], cregu=1000)
    fitter.set_image(depth / max_depth, focal=241, distortion=d["distortion"])
    fitter.set_max_depth(1)
    fitter.set_depth_scale(float(d["depth_scale"]))
    stepper = GraphedStep(fitter) if graph else fitter  # (GraphedStep runs iterations 0 .. 4 eagerly while it sets itself up)
    energies = run(lambda: stepper.step_device()[0], iterations, max(iterations // 10, 1), "depth fit")
    if save:
        _e, depth_image, diff_image = stepper.step_device()
        np.savez(save, energies=energies, vertices=fitter.vertices.cpu().numpy(), depth=depth_image.cpu().numpy(), diff=diff_image.cpu().numpy())
    return energies


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=100)
    ap.add_argument("--eager", action="store_true", help="launch the kernels of every iteration from the host instead of replaying a HIP graph")
    ap.add_argument("--save", default=None)
    a = ap.parse_args()
    main(a.iterations, not a.eager, a.save)
