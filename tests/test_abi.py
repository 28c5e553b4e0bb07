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
