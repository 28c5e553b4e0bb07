"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads and exports what include/*.h declares."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_lib():
    import __graft_entry__ as g

    return ctypes.CDLL(g.build_hip())


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "deodr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(deodr_hip_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported(hip_lib):
    names = declared_symbols()
    assert {"deodr_hip_render_scene", "deodr_hip_render_scene_b", "deodr_hip_workspace_bytes"} <= set(names)
    for n in names:
        assert hasattr(hip_lib, n), n


def test_abi_version_and_workspace_query(hip_lib):
    from deodr_amd import hip_renderer as hr

    assert hip_lib.deodr_hip_abi_version() == hr.ABI_VERSION
    hip_lib.deodr_hip_workspace_bytes.restype = ctypes.c_size_t
    hip_lib.deodr_hip_workspace_bytes.argtypes = [ctypes.c_int] * 5 + [ctypes.c_size_t]
    one = hip_lib.deodr_hip_workspace_bytes(20000, 1024, 1024, 4, 1, 0)
    assert 20e6 < one < 200e6  # tens of MB per 1024^2 / 20k-triangle view
    assert hip_lib.deodr_hip_workspace_bytes(20000, 1024, 1024, 4, 8, 0) == 8 * one
    assert hip_lib.deodr_hip_workspace_bytes(20000, 1024, 1024, 4, 1, 1 << 22) > one
    assert hip_lib.deodr_hip_workspace_bytes(-1, 1024, 1024, 4, 1, 0) == 0


def test_python_struct_matches_header():
    """Field order of the ctypes mirror == field order of DeodrHipScene in the header."""
    from deodr_amd.hip_renderer import _SceneC

    text = open(os.path.join(ROOT, "include", "deodr_hip.h")).read()
    body = text[text.index("typedef struct DeodrHipScene") : text.index("} DeodrHipScene;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = re.sub(r"^(const\s+)?(void|uint32_t|uint8_t|int)\s*", "", decl)
        fields += [n.strip().lstrip("*").strip() for n in names.split(",")]
    assert fields == [f[0] for f in _SceneC._fields_]


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from deodr_amd import scenes

    s = scenes.soup_scene(n_tri=3, width=16, height=16, min_area=10.0)
    with pytest.raises(Exception):
        s.render(1.0)  # must fail loudly, never fall back to a CPU path


def test_scene_checks_of_the_boundary_reject_before_any_launch():
    """The host-side part of checkSceneValid (H.h:2664-2715, 2924, 810) at the C ABI: every bad argument is refused with a message,
    before any HIP call -- the pointers below are never dereferenced, no GPU is needed."""
    from deodr_amd import hip_renderer as hr

    L = hr.lib()
    fake = 0x1000  # non-NULL, never read
    arrays = ("faces", "faces_uv", "textured", "shaded", "depths", "ij", "shade", "colors", "edgeflags", "uv")

    def scene(**changes):
        sc = hr._SceneC()
        for n in arrays + ("background_color", "uv_b", "ij_b", "shade_b", "colors_b"):
            setattr(sc, n, fake)
        sc.nb_triangles, sc.nb_vertices, sc.nb_uv, sc.height, sc.width, sc.nb_colors, sc.n_views = 10, 30, 30, 64, 64, 3, 1
        sc.backface_culling, sc.strict_edge, sc.integer_pixel_centers = 1, 1, 1
        sc.vertex_dtype, sc.pixel_dtype = 1, 0  # DEODR_HIP_F64 vertices, DEODR_HIP_F32 pixels (include/deodr_hip.h)
        for k, v in changes.items():
            setattr(sc, k, v)
        return sc

    def forward(sc, workspace=fake, nbytes=1 << 30, aa=0, obs=None, err=None):
        rc = L.deodr_hip_render_scene(ctypes.byref(sc) if sc is not None else None, fake, fake, 1.0, aa, obs, err, workspace, nbytes, None)
        return rc, L.deodr_hip_last_error().decode()

    def backward(sc, image_b=fake):
        rc = L.deodr_hip_render_scene_b(ctypes.byref(sc), fake, fake, image_b, 1.0, 0, None, None, None, fake, 1 << 30, 1, None)
        return rc, L.deodr_hip_last_error().decode()

    text = open(os.path.join(ROOT, "include", "deodr_hip.h")).read()
    assert re.search(r"DEODR_HIP_F32\s*=?\s*0", text) and re.search(r"DEODR_HIP_F64\s*=?\s*1", text)
    assert forward(None) == (1, "scene == NULL")
    for n in arrays:
        assert forward(scene(**{n: None})) == (1, "scene array == NULL"), n
    assert forward(scene(background_image=fake))[1].startswith("exactly one of scene.background_image / scene.background_color")
    assert forward(scene(background_color=None))[1].startswith("exactly one of")
    for bad in (dict(nb_triangles=-1), dict(nb_vertices=0), dict(nb_uv=0), dict(height=0), dict(width=-3), dict(n_views=0)):
        assert forward(scene(**bad)) == (1, "invalid scene dimensions"), bad
    assert forward(scene(nb_colors=0))[1] == "nb_colors out of range"
    assert forward(scene(height=40000))[1].startswith("image larger than 32767 pixels")
    assert forward(scene(pixel_dtype=7))[1] == "unknown dtype tag"
    assert forward(scene(texture=fake, texture_height=1, texture_width=8))[1] == "texture must be at least 2 x 2"
    assert forward(scene(), workspace=None)[1] == "workspace == NULL"
    assert forward(scene(), nbytes=1024)[1].startswith("workspace too small")
    assert backward(scene(backface_culling=0))[1].startswith("You have to use backface_culling true")  # the reference's message, H.h:2924
    assert backward(scene(perspective_correct=1))[1].startswith("backward gradient propagation not supported yet with perspective_correct")
    assert backward(scene(ij_b=None))[1] == "scene gradient array == NULL"
    assert backward(scene(texture=fake, texture_height=8, texture_width=8))[1].startswith("scene.texture_b == NULL")
    assert backward(scene(), image_b=None)[1].startswith("image_b == NULL")
