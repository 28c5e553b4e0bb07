"""Generate tests/golden/*.npz from the REAL reference (its own Python + Cython build).

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py

The reference tree is copied to a temp dir, built there with its own setup.py
(`build_ext --inplace`), three import stubs are injected (trimesh.base.Trimesh is only a type
annotation in deodr/triangulated_mesh.py:10-13,369; cv2 / imageio are imported by the examples and
replaced by PIL), the reference's own drivers are run, and inputs + outputs are dumped as small
fixtures.  Nothing of the reference's source is written into this repository; the fixtures are data.

Fixtures
  soup30_cw{0,1}.npz   deodr/examples/triangle_soup_fitting.py `run()` set-up (np.random.seed(2),
                       create_example_scene(clockwise), perturbed scene_init): all Scene2D inputs,
                       the texture as uint8 (material = imread/255 is reproduced exactly), SHA-256 of
                       the float64 image / z_buffer of scene_gt and of scene_init for both
                       antialiase_error modes, every gradient array of the first
                       render_compare_and_backward, and the 50-iteration loss curves whose last values are
                       the goldens of tests/test_triangle_soup_fitting.py:29-108.
  hand_mesh.npz        vertices / faces of deodr/data/hand.obj read with deodr/obj.py (input data for
                       BASELINE configs 2 and 4).
"""

import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

REFERENCE = os.environ.get("DEODR_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

STUBS = {
    "trimesh/__init__.py": "from . import base\n",
    "trimesh/base.py": "class Trimesh:\n    pass\n",
    "cv2.py": "def imshow(*a, **k):\n    pass\n\n\ndef waitKey(*a, **k):\n    return 0\n",
    "imageio/__init__.py": "from . import v3\n",
    "imageio/v3.py": (
        "import numpy as np\nfrom PIL import Image\n\n\n"
        "def imread(path):\n    return np.asarray(Image.open(path))\n\n\n"
        "def imwrite(path, arr):\n    Image.fromarray(np.asarray(arr)).save(path)\n"
    ),
}


def build_reference(tmp):
    for name in ("deodr", "C++", "setup.py", "readme.md"):
        src = os.path.join(REFERENCE, name)
        dst = os.path.join(tmp, name)
        shutil.copytree(src, dst) if os.path.isdir(src) else shutil.copy(src, dst)
    subprocess.run([sys.executable, "setup.py", "build_ext", "--inplace"], cwd=tmp, check=True, capture_output=True)
    stubs = os.path.join(tmp, "_stubs")
    for rel, text in STUBS.items():
        path = os.path.join(stubs, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)
    sys.path.insert(0, stubs)
    sys.path.insert(0, tmp)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def scene_inputs(s, prefix):
    keys = ["faces", "faces_uv", "ij", "depths", "textured", "uv", "shade", "colors", "shaded", "edgeflags"]
    return {prefix + k: np.asarray(getattr(s, k)) for k in keys}


def soup(clockwise):
    import copy

    import deodr
    from deodr import differentiable_renderer_cython as drc
    from deodr.examples.triangle_soup_fitting import create_example_scene, run
    from PIL import Image

    np.random.seed(2)
    scene_gt = create_example_scene(clockwise=clockwise)
    sigma = 1
    image_target = np.zeros((scene_gt.height, scene_gt.width, scene_gt.nb_colors))
    z_target = np.zeros((scene_gt.height, scene_gt.width))
    drc.renderSceneCpp(scene_gt, sigma, image_target, z_target)
    n_vertices = len(scene_gt.depths)
    scene_init = copy.deepcopy(scene_gt)  # as triangle_soup_fitting.py:131-136
    scene_init.ij = scene_gt.ij + np.random.randn(n_vertices, 2) * 10
    scene_init.uv = scene_gt.uv + np.random.randn(n_vertices, 2) * 0
    max_uv = np.array(scene_gt.texture.shape[:2]) - 1
    scene_init.uv = np.minimum(np.maximum(scene_init.uv, 0), max_uv)
    scene_init.colors = scene_gt.colors + np.random.randn(n_vertices, 3) * 0

    tex_u8 = np.asarray(Image.open(os.path.join(deodr.data_path, "trefle.jpg")))
    assert np.array_equal(tex_u8.astype(np.float64) / 255, scene_gt.texture)

    out = {}
    out.update(scene_inputs(scene_gt, "gt_"))
    out.update(scene_inputs(scene_init, "init_"))
    out["texture_u8"] = tex_u8
    out["background_rgb"] = np.array([0.3, 0.5, 0.7])
    out["height"], out["width"], out["clockwise"] = scene_gt.height, scene_gt.width, clockwise
    out["gt_image_sha256"], out["gt_z_sha256"] = sha(image_target), sha(z_target)
    out["gt_image_mean"] = image_target.mean()
    for aa in (False, True):
        sc = copy.deepcopy(scene_init)
        image, z, err_buffer, err = sc.render_compare_and_backward(sigma=sigma, antialiase_error=aa, obs=image_target)
        tag = f"aa{int(aa)}_"
        out[tag + "image_sha256"], out[tag + "z_sha256"], out[tag + "err_buffer_sha256"] = sha(image), sha(z), sha(err_buffer)
        out[tag + "loss"] = err
        for g in ("ij_b", "colors_b", "uv_b", "shade_b"):
            out[tag + g] = getattr(sc, g)
        out[tag + "texture_b_sha256"] = sha(sc.texture_b)  # stock reference: overwrite bug of H.h:621-624 included
        out[tag + "texture_b_sum"] = sc.texture_b.sum()
        losses, hashes = run(nb_max_iter=50, display=False, clockwise=clockwise, antialiase_error=aa)
        out[tag + "losses50"] = np.array(losses)
        out[tag + "hash_iter0"], out[tag + "hash_iter1"] = hashes[0], hashes[1]
        assert hashes[0] == out[tag + "image_sha256"]
    np.savez_compressed(os.path.join(OUT, f"soup30_cw{int(clockwise)}.npz"), **out)
    print(f"soup30_cw{int(clockwise)}: final losses", out["aa0_losses50"][-1], out["aa1_losses50"][-1])


def hand():
    import deodr
    from deodr.obj import read_obj

    faces, vertices = read_obj(os.path.join(deodr.data_path, "hand.obj"))
    np.savez_compressed(os.path.join(OUT, "hand_mesh.npz"), faces=np.asarray(faces, dtype=np.uint32), vertices=np.asarray(vertices, dtype=np.float64))
    print("hand mesh", np.shape(vertices), np.shape(faces))


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as tmp:
        build_reference(tmp)
        soup(False)
        soup(True)
        hand()
