"""Fit the hand mesh, its pose, one colour and a directional + ambient light to a photograph -- the reference's
deodr/examples/rgb_image_hand_fitting.py:22-101 on the device.

    python examples/rgb_image_hand_fitting.py [--iterations 100] [--eager] [--save out.npz]
"""
import argparse

import numpy as np

from _common import golden, hand_mesh, run


def main(iterations=100, graph=True, save=None):
    from deodr_amd.mesh_fitter import GraphedStep, MeshRGBFitterWithPose

    r = golden("rgb_hand_fit.npz")  # hand.png of the reference, the constants of its example
    _vertices, faces = hand_mesh()
    fitter = MeshRGBFitterWithPose(r["vertices_centered"], faces, np.zeros(3), r["translation_init"], r["default_color"], r["default_light_directional"],
                                   float(r["default_light_ambient"]), cregu=1000)  # fmt: skip
    fitter.set_image(r["image_u8"].astype(np.float64) / 255)
    fitter.set_background_color(r["background_color"])
    stepper = GraphedStep(fitter) if graph else fitter
    energies = run(lambda: stepper.step_device()[0], iterations, max(iterations // 10, 1), "colour fit")
    if save:
        _e, image = stepper.step_device()
        np.savez(save, energies=energies, vertices=fitter.vertices.cpu().numpy(), image=image[0].cpu().numpy(), color=fitter.mesh_color.cpu().numpy(),
                 light_directional=fitter.light_directional.cpu().numpy(), light_ambient=float(fitter.light_ambient))  # fmt: skip
    return energies


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=100)
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--save", default=None)
    a = ap.parse_args()
    main(a.iterations, not a.eager, a.save)
