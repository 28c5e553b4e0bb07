"""ctypes front-end of tests/sim/libtile_sim.so (host instantiation of the kernels' per-primitive code; tests only)."""

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "sim", "tile_sim.cpp")
LIB = os.path.join(HERE, "sim", "libtile_sim.so")


class SimScene(C.Structure):
    _fields_ = (
        [(n, C.c_void_p) for n in ("faces", "faces_uv", "textured", "shaded", "edgeflags", "depths", "ij", "shade", "colors", "uv")]
        + [(n, C.c_int) for n in ("T", "V", "Vuv", "H", "W", "C", "tex_h", "tex_w", "clockwise", "culling", "strict", "persp", "ipc")]
        + [("sigma", C.c_double)]
    )


def lib():
    deps = [SRC] + [os.path.join(HERE, "..", "deodr_amd", "csrc", h) for h in ("dr_math.h", "dr_prims.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", LIB, SRC], check=True)
    L = C.CDLL(LIB)
    L.sim_bin_counts.argtypes = [C.POINTER(SimScene), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.sim_tri_coverage.argtypes = [C.POINTER(SimScene), C.c_int, C.c_void_p]
    L.sim_edge_coverage.argtypes = [C.POINTER(SimScene), C.c_int, C.c_int, C.c_void_p]
    return L


def sim_scene(s, sigma=1.0):
    keep = dict(
        faces=np.ascontiguousarray(s.faces, np.uint32), faces_uv=np.ascontiguousarray(s.faces_uv, np.uint32),
        textured=np.ascontiguousarray(s.textured, np.uint8), shaded=np.ascontiguousarray(s.shaded, np.uint8),
        edgeflags=np.ascontiguousarray(s.edgeflags, np.uint8), depths=np.ascontiguousarray(s.depths, np.float64),
        ij=np.ascontiguousarray(s.ij, np.float64), shade=np.ascontiguousarray(s.shade, np.float64),
        colors=np.ascontiguousarray(s.colors, np.float64), uv=np.ascontiguousarray(s.uv, np.float64),
    )  # fmt: skip
    c = SimScene()
    for k, v in keep.items():
        setattr(c, k, v.ctypes.data)
    c.T, c.V, c.Vuv = len(keep["faces"]), len(keep["depths"]), len(keep["uv"])
    c.H, c.W, c.C = s.height, s.width, keep["colors"].shape[1]
    c.tex_h, c.tex_w = s.texture.shape[:2] if np.size(s.texture) else (0, 0)
    c.clockwise, c.culling, c.strict = int(s.clockwise), int(s.backface_culling), int(s.strict_edge)
    c.persp, c.ipc, c.sigma = int(s.perspective_correct), int(s.integer_pixel_centers), sigma
    return c, keep


def bin_counts(s, sigma=1.0, tile=8, exact=True):
    c, keep = sim_scene(s, sigma)
    nt = ((s.width + tile - 1) // tile) * ((s.height + tile - 1) // tile)
    tc, ec = np.zeros(nt, np.uint32), np.zeros(nt, np.uint32)
    lib().sim_bin_counts(C.byref(c), tile, int(exact), tc.ctypes.data, ec.ctypes.data)
    return tc, ec
