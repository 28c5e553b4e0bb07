"""Fit the image coordinates of a triangle soup to an image with the NumPy-level drop-in of the reference's Scene2D -- the loop of
deodr/examples/triangle_soup_fitting.py:100-180 unchanged but for the import (`deodr_amd.differentiable_renderer` for
`deodr.differentiable_renderer`); the renders and adjoints run on the GPU, the arrays live on the host as in the reference.

    python examples/triangle_soup_fitting.py [--iterations 50] [--antialiase-error]
"""
import argparse
import copy
import time

import numpy as np

from _common import ROOT  # noqa: F401  (puts the repository on sys.path)


def main(iterations=50, antialiase_error=False):
    from deodr_amd import scenes

    scene_gt = scenes.soup_scene(n_tri=30, width=200, height=200, seed=2, textured_ratio=0.5, flat=False)
    sigma = 1
    image_target, _z = scene_gt.render(sigma)
    rs = np.random.RandomState(2)
    n_vertices = len(scene_gt.depths)
    scene_iter = copy.deepcopy(scene_gt)
    scene_iter.ij = scene_gt.ij + rs.randn(n_vertices, 2) * 10
    alpha_ij, beta_ij = 0.01, 0.80
    speed_ij = np.zeros((n_vertices, 2))
    losses = []
    t0 = time.perf_counter()
    for niter in range(iterations):
        _image, _zb, _loss_image, loss = scene_iter.render_compare_and_backward(sigma=sigma, antialiase_error=antialiase_error, obs=image_target)
        losses.append(loss)
        speed_ij = beta_ij * speed_ij - scene_iter.ij_b * alpha_ij
        scene_iter.ij = scene_iter.ij + speed_ij
        if niter % max(iterations // 10, 1) == 0:
            print(f"soup fit: iteration {niter:4d}  loss {loss:.6f}")
    print(f"soup fit: iteration {iterations - 1:4d}  loss {losses[-1]:.6f}   ({(time.perf_counter() - t0) / iterations * 1e3:.3f} ms per iteration, host arrays)")
    return losses


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=50)
    ap.add_argument("--antialiase-error", action="store_true")
    a = ap.parse_args()
    main(a.iterations, a.antialiase_error)
