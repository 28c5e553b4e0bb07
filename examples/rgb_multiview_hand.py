"""One deformable hand, one colour and light, one pose per photograph: the reference's deodr/examples/rgb_multiview_hand.py:21-110 with
all views rendered by ONE batched launch per iteration (and, under torch.distributed, sharded over the GPUs: one all-reduce of the
shared gradients per iteration).

    python examples/rgb_multiview_hand.py [--iterations 100] [--eager]
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 examples/rgb_multiview_hand.py      # views 0, 1 | view 2
"""
import argparse
import os

import numpy as np

from _common import golden, hand_mesh, run


def main(iterations=100, graph=True):
    import torch

    if "RANK" in os.environ:  # one process per GPU
        import torch.distributed as dist

        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        dist.init_process_group("nccl")
    from deodr_amd.mesh_fitter import GraphedStep, MeshRGBFitterWithPoseMultiFrame

    d = golden("rgb_multiview_fit.npz")  # the three photographs of the reference's example, its initial poses and constants
    _vertices, faces = hand_mesh()
    fitter = MeshRGBFitterWithPoseMultiFrame(d["vertices_centered"], faces, d["euler_init"], d["translation_init"], d["default_color"],
                                             d["default_light_directional"], float(d["default_light_ambient"]), cregu=2000,
                                             device=torch.device("cuda", torch.cuda.current_device()))  # fmt: skip
    fitter.set_images([im.astype(np.float64) / 255 for im in d["images_u8"]])
    fitter.set_background_color(np.zeros(3))
    # (a captured graph cannot hold a collective of another process group's stream: the sharded fit steps eagerly -- 13 launches)
    stepper = GraphedStep(fitter) if graph and fitter.world == 1 else fitter
    label = f"multi-view fit, views {fitter.my_views}"
    return run(lambda: stepper.step_device()[0], iterations, max(iterations // 10, 1), label)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=100)
    ap.add_argument("--eager", action="store_true")
    a = ap.parse_args()
    main(a.iterations, not a.eager)
