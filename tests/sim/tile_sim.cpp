// tests/sim/tile_sim.cpp -- TEST / ANALYSIS TOOL, never part of the product.
//
// Host instantiation of the per-primitive code the HIP kernels run (deodr_amd/csrc/dr_math.h, dr_prims.h are
// `__host__ __device__`): set-up of every triangle and silhouette edge of a view and the binning decisions of
// setup_bin_kernel, executed sequentially.  Used (a) to study tile-list statistics of a scene without a GPU and (b) by
// tests/test_sim.py to check the device set-up arithmetic (stencils, spans, planes) against the CPU oracle.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../deodr_amd/csrc/dr_prims.h"

using namespace dr;

extern "C" {

struct SimScene
{
	const uint32_t *faces, *faces_uv;
	const uint8_t *textured, *shaded, *edgeflags;
	const double *depths, *ij, *shade, *colors, *uv;
	int T, V, Vuv, H, W, C, tex_h, tex_w;
	int clockwise, culling, strict, persp, integer_pixel_centers;
	double sigma;
};

static SceneView view_of(const SimScene *s)
{
	SceneView v;
	v.faces = s->faces;
	v.faces_uv = s->faces_uv;
	v.textured = s->textured;
	v.shaded = s->shaded;
	v.edgeflags = s->edgeflags;
	v.depths = s->depths;
	v.ij = s->ij;
	v.shade = s->shade;
	v.colors = s->colors;
	v.uv = s->uv;
	v.T = s->T;
	v.V = s->V;
	v.Vuv = s->Vuv;
	v.H = s->H;
	v.W = s->W;
	v.C = s->C;
	v.P = planes_per_prim(s->C);
	v.tex_h = s->tex_h;
	v.tex_w = s->tex_w;
	v.has_texture = s->tex_h > 0 && s->tex_w > 0;
	v.clockwise = s->clockwise;
	v.culling = s->culling;
	v.strict = s->strict;
	v.persp = s->persp;
	v.vtx_f64 = true;
	v.offset = s->integer_pixel_centers ? 0.0 : 0.5;
	v.sigma = s->sigma;
	return v;
}

static bool tile_outside(const double *eq, int n, int tx, int ty, int tile)
{ // same test as tile_outside_halfplanes in dr_kernels.hip
	const double xa = tx * tile, xb = tx * tile + (tile - 1), ya = ty * tile, yb = ty * tile + (tile - 1);
	for (int k = 0; k < n; k++)
	{
		const double a = eq[3 * k], b = eq[3 * k + 1], c = eq[3 * k + 2];
		const double emax = a * (a > 0 ? xb : xa) + b * (b > 0 ? yb : ya) + c;
		const double scale = fabs(a) * xb + fabs(b) * yb + fabs(c);
		if (emax < -1e-9 * scale - 1e-12)
			return true;
	}
	return false;
}

// per-tile triangle / edge counts exactly as setup_bin_kernel bins them (exact = with the half-plane rejection)
void sim_bin_counts(const SimScene *s, int tile, int exact, uint32_t *tri_cnt, uint32_t *edge_cnt)
{
	SceneView v = view_of(s);
	const int tiles_x = (v.W + tile - 1) / tile, tiles_y = (v.H + tile - 1) / tile;
	memset(tri_cnt, 0, sizeof(uint32_t) * tiles_x * tiles_y);
	memset(edge_cnt, 0, sizeof(uint32_t) * tiles_x * tiles_y);
	std::vector<double> planes(3 * v.P), eplanes(9 * v.P);
	for (int k = 0; k < v.T; k++)
	{
		TriRec rec;
		EdgeRec erec[3];
		setup_triangle(v, k, rec, planes.data(), erec, eplanes.data());
		if (rec.kind != KIND_NONE)
		{
			int x0 = rec.x_min < 0 ? 0 : rec.x_min, x1 = rec.x_max > v.W - 1 ? v.W - 1 : rec.x_max;
			int y0 = rec.y_begin[0] < 0 ? 0 : rec.y_begin[0], y1 = rec.y_end[1] > v.H - 1 ? v.H - 1 : rec.y_end[1];
			if (x0 <= x1 && y0 <= y1)
				for (int ty = y0 / tile; ty <= y1 / tile; ty++)
					for (int tx = x0 / tile; tx <= x1 / tile; tx++)
						if (!exact || !tile_outside(&rec.eq[0][0], 3, tx, ty, tile) || (!v.strict && tx == x1 / tile))
							tri_cnt[ty * tiles_x + tx]++; // (the column of x_max under the non-strict rule: see setup_bin_kernel)
		}
		for (int n = 0; n < 3; n++)
		{
			const EdgeRec &e = erec[n];
			if (e.kind == KIND_NONE || e.x_begin > e.x_end || e.y_begin > e.y_end)
				continue;
			const double band[12] = {e.x2b[0], e.x2b[1], e.x2b[2], e.x2b[3], e.x2b[4], e.x2b[5], e.x2t[0], e.x2t[1], e.x2t[2], -e.x2t[0], -e.x2t[1], 1 - e.x2t[2]};
			for (int ty = e.y_begin / tile; ty <= e.y_end / tile; ty++)
				for (int tx = e.x_begin / tile; tx <= e.x_end / tile; tx++)
					if (!exact || !tile_outside(band, 4, tx, ty, tile))
						edge_cnt[ty * tiles_x + tx]++;
		}
	}
}

// coverage mask of one triangle / edge on the whole image through the span functions the kernels use
void sim_tri_coverage(const SimScene *s, int k, uint8_t *mask)
{
	SceneView v = view_of(s);
	std::vector<double> planes(3 * v.P), eplanes(9 * v.P);
	TriRec rec;
	EdgeRec erec[3];
	setup_triangle(v, k, rec, planes.data(), erec, eplanes.data());
	for (int y = 0; y < v.H; y++)
		for (int x = 0; x < v.W; x++)
			mask[y * v.W + x] = rec.kind != KIND_NONE && tri_covers(rec, x, y, v.W, v.H, v.strict);
}

void sim_edge_coverage(const SimScene *s, int k, int n, uint8_t *mask)
{
	SceneView v = view_of(s);
	std::vector<double> planes(3 * v.P), eplanes(9 * v.P);
	TriRec rec;
	EdgeRec erec[3];
	setup_triangle(v, k, rec, planes.data(), erec, eplanes.data());
	for (int y = 0; y < v.H; y++)
		for (int x = 0; x < v.W; x++)
			mask[y * v.W + x] = erec[n].kind != KIND_NONE && edge_covers(erec[n], x, y, v.W);
}
}
