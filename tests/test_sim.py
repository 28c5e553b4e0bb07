"""CPU checks of the kernels' per-primitive arithmetic (deodr_amd/csrc/dr_math.h, dr_prims.h instantiated on the host by
tests/sim/tile_sim.cpp): coverage of single primitives through the SAME span functions the HIP kernels use must equal the
coverage the CPU oracle draws -- including degenerate slopes where the reference falls back to its incremental search."""

import numpy as np
import pytest

import sim_util
from deodr_amd.differentiable_renderer import Scene2D


def one_triangle_scene(ij, W=40, H=32, strict=True, edgeflags=(True, True, True)):
    ij = np.asarray(ij, dtype=np.float64)
    cr = (ij[1, 0] - ij[0, 0]) * (ij[2, 1] - ij[0, 1]) - (ij[1, 1] - ij[0, 1]) * (ij[2, 0] - ij[0, 0])
    return Scene2D(
        faces=np.array([[0, 1, 2]], dtype=np.uint32), faces_uv=np.array([[0, 1, 2]], dtype=np.uint32), ij=ij,
        depths=np.array([1.0, 1.0, 1.0]), textured=np.zeros(1, dtype=bool), uv=np.zeros((3, 2)), shade=np.zeros(3),
        colors=np.ones((3, 1)), shaded=np.zeros(1, dtype=bool), edgeflags=np.array([edgeflags], dtype=bool), height=H, width=W,
        nb_colors=1, texture=np.zeros((0, 0)), background_color=np.zeros(1), clockwise=bool(cr > 0), backface_culling=True,
        strict_edge=strict,
    )  # fmt: skip


CASES = [
    [[3.2, 4.1], [30.7, 6.3], [12.4, 27.9]],
    [[5, 5], [30, 5], [18, 25]],  # exactly horizontal edge: the reference's b == 0 fallback
    [[5, 5], [5, 28], [31, 17]],  # exactly vertical edge
    [[2, 10], [38, 10.0000001], [20, 30]],  # almost horizontal: quotient beyond the 16-bit range
    [[10, 2], [10.0000001, 30], [30, 16]],  # almost vertical
    [[-20, -10], [70, 5], [10, 60]],  # larger than the image
    [[7, 7], [9, 7.5], [8, 9]],  # tiny
]


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_span_functions_match_oracle_coverage(oracle_api, case, strict):
    import ctypes as C

    s = one_triangle_scene(CASES[case], strict=strict)
    lib = sim_util.lib()
    c, keep = sim_util.sim_scene(s, sigma=1.5)
    rnd = oracle_api.ref() or oracle_api.port()
    # triangle coverage == pixels whose z-buffer the oracle wrote
    image, z = rnd.render(s, 0.0)
    mask = np.zeros((s.height, s.width), dtype=np.uint8)
    lib.sim_tri_coverage(C.byref(c), 0, mask.ctypes.data)
    assert np.array_equal(mask.astype(bool), np.isfinite(z))
    # edge-band coverage == pixels the oracle's edge pass changes (colour 1 blended over background 0), edge by edge
    for n in range(3):
        s1 = one_triangle_scene(CASES[case], strict=strict, edgeflags=tuple(i == n for i in range(3)))
        image1, z1 = rnd.render(s1, 1.5)
        band = (np.abs(image1[:, :, 0] - image[:, :, 0]) > 0) & ~np.isfinite(z)
        emask = np.zeros((s.height, s.width), dtype=np.uint8)
        lib.sim_edge_coverage(C.byref(c), 0, n, emask.ctypes.data)
        # the band also extends over the triangle itself only where Z_edge < z (never here: same depth), so compare outside
        assert np.array_equal(emask.astype(bool) & ~np.isfinite(z), band), n


@pytest.mark.parametrize("strict", [True, False])
def test_binning_keeps_every_tile_a_triangle_draws_into(oracle_api, strict):
    """The half-plane rejection of setup_bin_kernel must never drop a tile the reference's spans reach -- including the pixel of
    column x_max that the non-strict rule draws OUTSIDE the left edge (ceil_div clamps to x_max, H.h:895): near the rightmost
    vertex, and along the right border for a triangle that leaves the frame there."""
    import ctypes as C

    rs = np.random.RandomState(5)
    lib = sim_util.lib()
    tile, W, H = 8, 40, 32
    clamp_cases = 0
    tris = [[[30.3, 20.7], [48.5, 9.1], [39.9, -2.3]], [[20.2, 14.7], [17.9, 10.4], [31.6, 6.7]]]  # both draw column x_max outside the left edge
    tris += [(rs.rand(2) * [W, H] + (rs.rand(3, 2) - 0.5) * [W, H] * rs.choice([0.3, 1.0, 2.5])).tolist() for _ in range(300)]
    for ij in tris:
        s = one_triangle_scene(ij, W=W, H=H, strict=strict)
        c, keep = sim_util.sim_scene(s, sigma=1.0)
        mask = np.zeros((H, W), dtype=np.uint8)
        lib.sim_tri_coverage(C.byref(c), 0, mask.ctypes.data)
        tri_cnt, _ = sim_util.bin_counts(s, sigma=1.0, tile=tile, exact=True)
        tri_cnt = tri_cnt.reshape((H + tile - 1) // tile, (W + tile - 1) // tile)
        ys, xs = np.nonzero(mask)
        assert np.all(tri_cnt[ys // tile, xs // tile] == 1), ij
        # how often the clamp artefact occurs in this sample: drawn pixels left of the left edge by geometry
        e = np.asarray(ij, dtype=np.float64)
        inside = np.ones(len(xs), dtype=bool)
        for a, b in ((0, 1), (1, 2), (2, 0)):
            cr = (e[b, 0] - e[a, 0]) * (ys - e[a, 1]) - (e[b, 1] - e[a, 1]) * (xs - e[a, 0])
            inside &= (cr >= 0) if s.clockwise else (cr <= 0)
        clamp_cases += int((~inside).sum() > 0)
    if not strict:
        assert clamp_cases >= 2  # the sample does exercise the artefact
    else:
        assert clamp_cases == 0


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("sigma", [0.5, 1.0, 3.0])
def test_binning_keeps_every_tile_an_edge_band_draws_into(oracle_api, strict, sigma):
    """Same property for the silhouette edges: the four half-planes of the band (bary0 > 0, bary1 > 0, 0 < T < 1) must not
    reject a tile that get_edge_xrange_from_ineq (H.h:2620-2648) reaches, including bands leaving the frame on every side and
    integer vertex coordinates (pixel centres exactly on the band's border lines)."""
    import ctypes as C

    rs = np.random.RandomState(11)
    lib = sim_util.lib()
    tile, W, H = 8, 40, 32
    tris = [(rs.rand(2) * [W, H] + (rs.rand(3, 2) - 0.5) * [W, H] * rs.choice([0.3, 1.0, 2.5])) for _ in range(200)]
    tris += [np.round(t) for t in tris[:60]]
    drawn = 0
    for ij in tris:
        for n in range(3):
            s = one_triangle_scene(ij, W=W, H=H, strict=strict, edgeflags=tuple(i == n for i in range(3)))
            c, keep = sim_util.sim_scene(s, sigma=sigma)
            mask = np.zeros((H, W), dtype=np.uint8)
            lib.sim_edge_coverage(C.byref(c), 0, n, mask.ctypes.data)
            _, edge_cnt = sim_util.bin_counts(s, sigma=sigma, tile=tile, exact=True)
            edge_cnt = edge_cnt.reshape((H + tile - 1) // tile, (W + tile - 1) // tile)
            ys, xs = np.nonzero(mask)
            drawn += len(ys)
            assert np.all(edge_cnt[ys // tile, xs // tile] == 1), (np.asarray(ij).tolist(), n)
    assert drawn > 1000


@pytest.mark.parametrize("strict", [True, False])
def test_span_functions_match_oracle_coverage_random(oracle_api, strict):
    """test_span_functions_match_oracle_coverage over 300 random triangles (a third with integer vertices, many leaving the frame):
    triangle coverage and the band of every edge, pixel for pixel against what the checker draws"""
    import ctypes as C

    rs = np.random.RandomState(21)
    lib = sim_util.lib()
    rnd = oracle_api.ref() or oracle_api.port()
    W, H, sigma = 40, 32, 1.5
    tris = [(rs.rand(2) * [W, H] + (rs.rand(3, 2) - 0.5) * [W, H] * rs.choice([0.3, 1.0, 2.5])) for _ in range(200)]
    tris += [np.round(t) for t in tris[:100]]
    for ij in tris:
        s = one_triangle_scene(ij, W=W, H=H, strict=strict)
        if abs(np.linalg.det(np.column_stack((np.asarray(ij), np.ones(3))))) < 1e-9:
            continue  # three integer vertices on a line
        c, keep = sim_util.sim_scene(s, sigma=sigma)
        image, z = rnd.render(s, 0.0)
        mask = np.zeros((H, W), dtype=np.uint8)
        lib.sim_tri_coverage(C.byref(c), 0, mask.ctypes.data)
        assert np.array_equal(mask.astype(bool), np.isfinite(z)), np.asarray(ij).tolist()
        for n in range(3):
            s1 = one_triangle_scene(ij, W=W, H=H, strict=strict, edgeflags=tuple(i == n for i in range(3)))
            image1, _ = rnd.render(s1, sigma)
            band = (np.abs(image1[:, :, 0] - image[:, :, 0]) > 0) & ~np.isfinite(z)
            emask = np.zeros((H, W), dtype=np.uint8)
            lib.sim_edge_coverage(C.byref(c), 0, n, emask.ctypes.data)
            assert np.array_equal(emask.astype(bool) & ~np.isfinite(z), band), (np.asarray(ij).tolist(), n)
