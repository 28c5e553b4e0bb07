// deodr_amd/csrc/dr_prims.h -- per-primitive set-up (forward) and finalize (adjoint) of the rasterizer.
//
// One call handles ONE triangle (and the up to three silhouette-edge slots it owns) of ONE view: it is the body of
// one thread of `setup_bin_kernel` / `finalize_kernel` in dr_kernels.hip.  Like dr_math.h the code is
// `__host__ __device__` so that tests/sim/tile_sim.cpp can run the identical arithmetic sequentially on the CPU.
//
// Layout of the per-primitive plane arrays (P = max(nb_colors, 3) planes of 3 doubles each):
//   KIND_INTERP    plane c          = colour channel c                              (H.h:779-785, 1579-1585)
//   KIND_TEXTURED  plane 0, 1       = texture coordinates u, v ; plane 2 = shade L  (H.h:1079-1087, 1825-1834)
// and of the adjoint accumulators filled by the backward raster kernel: the same P planes, each holding the three
// image moments  sum g [x, y, 1]  of the plane's adjoint; edges carry one more plane (index P) for the transparency T.
#pragma once
#include "dr_math.h"

namespace dr
{

struct SceneView // one view of the scene: plain device (or host, in the simulator) pointers
{
	const uint32_t *faces, *faces_uv;
	const uint8_t *textured, *shaded, *edgeflags;
	const void *depths, *ij, *shade, *colors, *uv; // vertex dtype
	int T, V, Vuv, H, W, C, P;
	int tex_h, tex_w;
	bool clockwise, culling, strict, persp;
	bool vtx_f64;
	bool has_texture; // scene.texture != NULL
	double offset; // 0 (integer pixel centres) or 0.5, H.h:2783
	double sigma;
};

DR_HD double ldv(const void *p, size_t i, bool f64) { return f64 ? ((const double *)p)[i] : (double)((const float *)p)[i]; }

static const int LIST_SUB[3][2] = {{1, 0}, {2, 1}, {0, 2}}; // H.h:2822: edge n joins vertices LIST_SUB[n]

DR_HD int planes_per_prim(int nb_colors) { return nb_colors < 3 ? 3 : nb_colors; }

// Per-triangle prologue of renderScene / renderScene_B (H.h:2751-2779): depth sum (the far->near sort key of its
// edges) and signed area (0 when a vertex is behind the camera).
DR_HD void tri_cull(const SceneView &s, int k, double &sum_depth, double &area)
{
	const uint32_t *face = s.faces + 3 * (size_t)k;
	sum_depth = 0;
	bool front = true;
	double V[3][2];
	for (int i = 0; i < 3; i++)
	{
		double d = ldv(s.depths, face[i], s.vtx_f64);
		if (d < 0)
			front = false;
		sum_depth += d;
		V[i][0] = ldv(s.ij, 2 * (size_t)face[i], s.vtx_f64);
		V[i][1] = ldv(s.ij, 2 * (size_t)face[i] + 1, s.vtx_f64);
	}
	area = front ? signed_area(V, s.clockwise) : 0.0;
}

// attribute planes of a primitive with nv vertices; vid / uvid are the vertex indices of its corners
DR_HD void attr_planes(const SceneView &s, int kind, int nv, const uint32_t vid[3], const uint32_t uvid[3], const double Zv[3],
					   const double *x2b, double *planes)
{
	double w[3] = {1, 1, 1};
	if (s.persp)
		for (int i = 0; i < nv; i++)
			w[i] = 1 / Zv[i];
	if (kind == KIND_TEXTURED)
	{
		for (int c = 0; c < 2; c++)
		{
			double a[3];
			for (int i = 0; i < nv; i++)
			{
				a[i] = ldv(s.uv, 2 * (size_t)uvid[i] + c, s.vtx_f64);
				if (s.persp)
					a[i] = a[i] * w[i];
			}
			for (int j = 0; j < 3; j++)
				planes[3 * c + j] = plane_coef(nv, a, x2b, j);
		}
		double a[3];
		for (int i = 0; i < nv; i++)
		{
			a[i] = ldv(s.shade, vid[i], s.vtx_f64);
			if (s.persp)
				a[i] = w[i] * a[i];
		}
		for (int j = 0; j < 3; j++)
			planes[6 + j] = plane_coef(nv, a, x2b, j);
	}
	else
		for (int c = 0; c < s.C; c++)
		{
			double a[3];
			for (int i = 0; i < nv; i++)
			{
				a[i] = ldv(s.colors, (size_t)vid[i] * s.C + c, s.vtx_f64);
				if (s.persp)
					a[i] = a[i] * w[i];
			}
			for (int j = 0; j < 3; j++)
				planes[3 * c + j] = plane_coef(nv, a, x2b, j);
		}
}

// attribute planes from attributes already in registers: att[v][c] (c < C colour channels, or u, v, shade for textured)
DR_HD void attr_planes_reg(const SceneView &s, int kind, int nv, const double att[3][4], const double Zv[3], const double *x2b,
						   double *planes)
{
	const int np = kind == KIND_TEXTURED ? 3 : s.C;
	// fully unrolled with a guard: a loop with a run-time trip count would index att[][c] dynamically, which sends the
	// caller's register-resident inputs to scratch memory
#pragma unroll
	for (int c = 0; c < 4; c++)
		if (c < np)
		{
			double a[3];
#pragma unroll
			for (int i = 0; i < 3; i++)
			{
				a[i] = 0;
				if (i < nv)
				{
					a[i] = att[i][c];
					if (s.persp)
					{
						const double w = 1 / Zv[i];
						a[i] = (kind == KIND_TEXTURED && c == 2) ? w * a[i] : a[i] * w;
					}
				}
			}
#pragma unroll
			for (int j = 0; j < 3; j++)
				planes[3 * c + j] = plane_coef(nv, a, x2b, j);
		}
}

// Inputs of one triangle, fetched up front (two dependent memory round trips: indices, then vertex data; each further
// round trip in the middle of the thread's long dependent chain would cost the whole kernel its latency).
struct TriInputs
{
	uint32_t f[3], fuv[3];
	bool tex, both;
	double Vraw[3][2], Zv[3];
	double att[3][4]; // colours (C <= 4) or (u, v, shade)
	double sum_depth, area;
};

// The index checks of checkSceneValid (H.h:2700-2712), made where the indices are read: an out-of-range entry of faces /
// faces_uv (or a textured triangle in a scene without texture) must never be dereferenced.
enum SceneError : uint32_t
{
	SCENE_ERR_FACES = 1,	  // faces[k][i] >= nb_vertices
	SCENE_ERR_FACES_UV = 2,	  // faces_uv[k][i] >= nb_uv
	SCENE_ERR_NO_TEXTURE = 4, // textured[k] && shaded[k] but scene.texture == NULL
	SCENE_ERR_DET_RANGE = 16, // deterministic mode: a contribution or a sum left the fixed-point range (det_add, dr_workspace.h)
};

// -> 0, or the SceneError bits of triangle k (then nothing was gathered through its indices and the caller must drop it)
DR_HD uint32_t load_triangle(const SceneView &s, int k, TriInputs &t, bool with_attributes)
{
	const uint32_t *face = s.faces + 3 * (size_t)k, *face_uv = s.faces_uv + 3 * (size_t)k;
	uint32_t bad = 0;
	for (int i = 0; i < 3; i++)
	{
		t.f[i] = face[i];
		t.fuv[i] = face_uv[i];
		bad |= t.f[i] >= (uint32_t)s.V ? (uint32_t)SCENE_ERR_FACES : 0u;
		bad |= t.fuv[i] >= (uint32_t)s.Vuv ? (uint32_t)SCENE_ERR_FACES_UV : 0u;
	}
	t.tex = s.textured[k] != 0;
	t.both = t.tex && s.shaded[k] != 0;
	bad |= (t.both && !s.has_texture) ? (uint32_t)SCENE_ERR_NO_TEXTURE : 0u;
	if (bad)
	{
		t.sum_depth = 0;
		t.area = 0;
		return bad;
	}
	for (int i = 0; i < 3; i++)
	{
		t.Vraw[i][0] = ldv(s.ij, 2 * (size_t)t.f[i], s.vtx_f64);
		t.Vraw[i][1] = ldv(s.ij, 2 * (size_t)t.f[i] + 1, s.vtx_f64);
		t.Zv[i] = ldv(s.depths, t.f[i], s.vtx_f64);
	}
	for (int i = 0; i < 3; i++)
		for (int c = 0; c < 4; c++)
			t.att[i][c] = 0;
	if (with_attributes)
	{
		if (t.both)
			for (int i = 0; i < 3; i++)
			{
				t.att[i][0] = ldv(s.uv, 2 * (size_t)t.fuv[i], s.vtx_f64);
				t.att[i][1] = ldv(s.uv, 2 * (size_t)t.fuv[i] + 1, s.vtx_f64);
				t.att[i][2] = ldv(s.shade, t.f[i], s.vtx_f64);
			}
		else if (s.C <= 4)
			for (int i = 0; i < 3; i++)
				for (int c = 0; c < 4; c++)
					if (c < s.C)
						t.att[i][c] = ldv(s.colors, (size_t)t.f[i] * s.C + c, s.vtx_f64);
	}
	// prologue of renderScene (H.h:2751-2779): depth sum (sort key of the edges) and signed area (0 behind the camera)
	t.sum_depth = 0;
	bool front = true;
	for (int i = 0; i < 3; i++)
	{
		if (t.Zv[i] < 0)
			front = false;
		t.sum_depth += t.Zv[i];
	}
	t.area = front ? signed_area(t.Vraw, s.clockwise) : 0.0;
	return 0;
}

// Triangle part of the set-up in two steps, so that the device kernel can request its tile slots between them (everything
// binning needs -- edge equations and bounds -- comes out of the first step; the second one is independent arithmetic that
// then overlaps the round trip of the slot requests).  Pass-1 kind follows H.h:2785-2819.
// Step 1: stencil (edge equations, bounds, barycentric frame x2b) and kind.  -> false: culled, nothing else to do.
DR_HD bool setup_tri_geometry(const SceneView &s, const TriInputs &t, TriRec &rec, double x2b[9])
{
	if (s.culling && !(t.area > 0))
	{ // culled: neither pass 1 (H.h:2786) nor pass 2 (H.h:2847) nor the adjoint (H.h:3063) touches it -- no stencil needed
		rec.kind = KIND_NONE;
		rec.front = 0;
		return false;
	}
	double V[3][2];
	for (int i = 0; i < 3; i++)
	{
		V[i][0] = t.Vraw[i][0] - s.offset;
		V[i][1] = t.Vraw[i][1] - s.offset;
	}
	tri_stencil(V, s.strict, rec, x2b);
	rec.front = t.area > 0;
	rec.kind = KIND_NONE;
	if (t.area > 0 || !s.culling)
		rec.kind = t.both ? KIND_TEXTURED : (t.tex ? KIND_NONE : KIND_INTERP);
	return true;
}

// Step 2: the Z plane of the record and the attribute planes.
DR_HD void setup_tri_attributes(const SceneView &s, const TriInputs &t, TriRec &rec, const double x2b[9], double *tri_planes /*[3P]*/)
{
	double zz[3];
	for (int i = 0; i < 3; i++)
		zz[i] = s.persp ? 1 / t.Zv[i] : t.Zv[i];
	for (int j = 0; j < 3; j++)
		rec.xZ[j] = plane_coef(3, zz, x2b, j);
	if (rec.kind != KIND_NONE)
	{
		if (t.both || s.C <= 4)
			attr_planes_reg(s, rec.kind, 3, t.att, t.Zv, x2b, tri_planes);
		else
			attr_planes(s, rec.kind, 3, t.f, t.fuv, t.Zv, x2b, tri_planes);
	}
}

DR_HD void setup_tri_only(const SceneView &s, const TriInputs &t, TriRec &rec, double *tri_planes /*[3P]*/)
{
	double x2b[9];
	if (setup_tri_geometry(s, t, rec, x2b))
		setup_tri_attributes(s, t, rec, x2b, tri_planes);
}

template <class T>
DR_HD T pick3(T v0, T v1, T v2, int i)
{
	return i == 0 ? v0 : (i == 1 ? v1 : v2);
}

// Edge n of triangle k: eligibility H.h:2847-2853, kind H.h:2868-2895, stencil H.h:1366-1460.
DR_HD void setup_edge_only(const SceneView &s, const TriInputs &t, int k, int n, EdgeRec &e, double *ep /*[3P]*/, EdgeFin *fin = nullptr)
{
	e.kind = KIND_NONE;
	if (!(s.sigma > 0) || !(t.area > 0) || !s.edgeflags[3 * (size_t)k + n])
		return;
	// vertices LIST_SUB[n] = {1,0}, {2,1}, {0,2} of the triangle, picked with selects: indexing the register-resident
	// TriInputs with a run-time n would push the whole structure to scratch memory
	const int a = n == 0 ? 1 : (n == 1 ? 2 : 0), b = n == 0 ? 0 : (n == 1 ? 1 : 2);
	// (pick3 takes VALUES: `i == 0 ? arr[0] : arr[1]` on lvalues selects an address and loads through it afterwards)
#define DR_PICK(arr, i) pick3(arr[0], arr[1], arr[2], i)
#define DR_PICK2(arr, i, c) pick3(arr[0][c], arr[1][c], arr[2][c], i)
	double EV[2][2], EZ[3] = {DR_PICK(t.Zv, a), DR_PICK(t.Zv, b), 0};
	EV[0][0] = DR_PICK2(t.Vraw, a, 0) - s.offset;
	EV[0][1] = DR_PICK2(t.Vraw, a, 1) - s.offset;
	EV[1][0] = DR_PICK2(t.Vraw, b, 0) - s.offset;
	EV[1][1] = DR_PICK2(t.Vraw, b, 1) - s.offset;
	edge_stencil(EV, s.H, s.W, s.sigma, s.clockwise, e);
	e.kind = t.both ? KIND_TEXTURED : KIND_INTERP; // textured && !shaded edges are drawn interpolated (H.h:2884)
	e.key = t.sum_depth;
	double zz[2] = {s.persp ? 1 / EZ[0] : EZ[0], s.persp ? 1 / EZ[1] : EZ[1]};
	for (int j = 0; j < 3; j++)
		e.xZ[j] = plane_coef(2, zz, e.x2b, j);
	if (fin)
	{ // inputs of the edge's finalize step (finalize_edge_fin)
		for (int i = 0; i < 2; i++)
			for (int d = 0; d < 2; d++)
				fin->V[i][d] = EV[i][d];
		fin->vid[0] = DR_PICK(t.f, a);
		fin->vid[1] = DR_PICK(t.f, b);
		fin->uvid[0] = DR_PICK(t.fuv, a);
		fin->uvid[1] = DR_PICK(t.fuv, b);
		fin->has_att = (t.both || (s.C <= 4 && !t.tex)) ? 1u : 0u;
		fin->pad[0] = fin->pad[1] = fin->pad[2] = 0;
		for (int c = 0; c < 4; c++)
		{
			fin->att[0][c] = DR_PICK2(t.att, a, c);
			fin->att[1][c] = DR_PICK2(t.att, b, c);
		}
	}
	if (t.both || (s.C <= 4 && !t.tex))
	{
		double eatt[3][4];
		for (int c = 0; c < 4; c++)
		{
			eatt[0][c] = DR_PICK2(t.att, a, c);
			eatt[1][c] = DR_PICK2(t.att, b, c);
			eatt[2][c] = 0;
		}
		attr_planes_reg(s, e.kind, 2, eatt, EZ, e.x2b, ep);
	}
	else
	{ // many channels, or a textured-but-unshaded triangle whose edges use the vertex colours
		uint32_t vid[3] = {DR_PICK(t.f, a), DR_PICK(t.f, b), 0}, uvid[3] = {DR_PICK(t.fuv, a), DR_PICK(t.fuv, b), 0};
		attr_planes(s, e.kind, 2, vid, uvid, EZ, e.x2b, ep);
	}
#undef DR_PICK
#undef DR_PICK2
}

// Whole set-up of triangle k (record, planes, three edge slots) in one call: used by the host simulator.
DR_HD void setup_triangle(const SceneView &s, int k, TriRec &rec, double *tri_planes /*[3P]*/, EdgeRec erec[3],
						  double *edge_planes /*[3][3P]*/)
{
	TriInputs t;
	if (load_triangle(s, k, t, true))
	{ // invalid indices: dropped (the device kernel also raises the scene's sticky error word)
		rec.kind = KIND_NONE;
		rec.front = 0;
		for (int n = 0; n < 3; n++)
			erec[n].kind = KIND_NONE;
		return;
	}
	setup_tri_only(s, t, rec, tri_planes);
	for (int n = 0; n < 3; n++)
		setup_edge_only(s, t, k, n, erec[n], edge_planes + (size_t)n * 3 * s.P);
}

// ------------------------------------------------------------------------------------------------------- finalize

struct GradView // adjoint arrays of one view (vertex dtype), accumulated into
{
	void *ij_b, *colors_b, *shade_b, *uv_b;
};

// `Add` is a functor  add(void* array, size_t index, bool f64, double value)  (atomicAdd on the device)
// Where finalize_triangle sends the adjoint of corner i of the triangle (i and the channel / coordinate index are compile-time
// constants at every call): the device kernel either adds them straight to the gradient arrays (atomics) or collects them in
// registers to merge the contributions of the triangles of a wavefront that share a vertex first.
//   sink.color(i, c, v)   d/d colors[face[i]][c]        sink.shade(i, v)   d/d shade[face[i]]
//   sink.uv(i, c, v)      d/d uv[face_uv[i]][c]          sink.ij(i, d, v)   d/d ij[face[i]][d]
//
// SMALL: the caller knows that nb_colors <= 4 (the loop over a run-time channel count is compiled out, so that `acc` may be a
// register-resident copy of the accumulators: no dynamic indexing)
template <bool SMALL = false, class Sink>
DR_HD void finalize_triangle(const SceneView &s, int k, int kind, const double *acc /*[3P]*/, Sink &sink)
{ // the caller has checked that the triangle is front-facing (the adjoint only visits those, H.h:3063) and of a drawable kind
	const uint32_t *face = s.faces + 3 * (size_t)k, *face_uv = s.faces_uv + 3 * (size_t)k;
	double V[3][2];
	for (int i = 0; i < 3; i++)
	{
		V[i][0] = ldv(s.ij, 2 * (size_t)face[i], s.vtx_f64) - s.offset;
		V[i][1] = ldv(s.ij, 2 * (size_t)face[i] + 1, s.vtx_f64) - s.offset;
	}
	double b2x[9], x2b[9], x2b_B[9], b2x_B[9];
	bary_frame(V, b2x);
	inv3(b2x, x2b);
	for (int i = 0; i < 9; i++)
		x2b_B[i] = b2x_B[i] = 0;
	if (kind == KIND_TEXTURED)
	{ // H.h:1138-1148
#pragma unroll
		for (int c = 0; c < 2; c++)
		{
			double a[3], a_B[3] = {0, 0, 0};
			for (int i = 0; i < 3; i++)
				a[i] = ldv(s.uv, 2 * (size_t)face_uv[i] + c, s.vtx_f64);
			plane_adjoint(3, acc + 3 * c, a, a_B, x2b, x2b_B);
#pragma unroll
			for (int i = 0; i < 3; i++)
				sink.uv(i, c, a_B[i]);
		}
		double a[3], a_B[3] = {0, 0, 0};
		for (int i = 0; i < 3; i++)
			a[i] = ldv(s.shade, face[i], s.vtx_f64);
		plane_adjoint(3, acc + 6, a, a_B, x2b, x2b_B);
#pragma unroll
		for (int i = 0; i < 3; i++)
			sink.shade(i, a_B[i]);
	}
	else if (SMALL || s.C <= 4)
	{ // same as below, unrolled under a guard: every colour load is in flight before the first one is used (a rolled loop
	  // pays one memory round trip per channel)
		double a[4][3];
#pragma unroll
		for (int c = 0; c < 4; c++)
			for (int i = 0; i < 3; i++)
				a[c][i] = c < s.C ? ldv(s.colors, (size_t)face[i] * s.C + c, s.vtx_f64) : 0.0;
#pragma unroll
		for (int c = 0; c < 4; c++)
			if (c < s.C)
			{
				double a_B[3] = {0, 0, 0};
				plane_adjoint(3, acc + 3 * c, a[c], a_B, x2b, x2b_B);
#pragma unroll
				for (int i = 0; i < 3; i++)
					sink.color(i, c, a_B[i]);
			}
	}
	else if (!SMALL)
		for (int c = 0; c < s.C; c++)
		{ // H.h:841-851
			double a[3], a_B[3] = {0, 0, 0};
			for (int i = 0; i < 3; i++)
				a[i] = ldv(s.colors, (size_t)face[i] * s.C + c, s.vtx_f64);
			plane_adjoint(3, acc + 3 * c, a, a_B, x2b, x2b_B);
#pragma unroll
			for (int i = 0; i < 3; i++)
				sink.color(i, c, a_B[i]);
		}
	inv3_adjoint(b2x, b2x_B, x2b_B); // H.h:854-858
#pragma unroll
	for (int v = 0; v < 3; v++)
#pragma unroll
		for (int d = 0; d < 2; d++)
			sink.ij(v, d, b2x_B[3 * d + v]);
}

template <class Add>
DR_HD void finalize_edge(const SceneView &s, const GradView &g, int k, int n, const EdgeRec &e, const double *acc /*[3P+3]*/, Add add)
{
	if (e.kind == KIND_NONE)
		return;
	const uint32_t *face = s.faces + 3 * (size_t)k, *face_uv = s.faces_uv + 3 * (size_t)k;
	const int ia = n == 0 ? 1 : (n == 1 ? 2 : 0), ib = n == 0 ? 0 : (n == 1 ? 1 : 2); // LIST_SUB[n]
	const uint32_t vid[2] = {face[ia], face[ib]}, uvid[2] = {face_uv[ia], face_uv[ib]};
	double V[2][2];
	for (int i = 0; i < 2; i++)
	{
		V[i][0] = ldv(s.ij, 2 * (size_t)vid[i], s.vtx_f64) - s.offset;
		V[i][1] = ldv(s.ij, 2 * (size_t)vid[i] + 1, s.vtx_f64) - s.offset;
	}
	double x2b_B[6] = {0, 0, 0, 0, 0, 0};
	if (e.kind == KIND_TEXTURED)
	{ // H.h:2045-2056
		for (int c = 0; c < 2; c++)
		{
			double a[3], a_B[3] = {0, 0, 0};
			for (int i = 0; i < 2; i++)
				a[i] = ldv(s.uv, 2 * (size_t)uvid[i] + c, s.vtx_f64);
			plane_adjoint(2, acc + 3 * c, a, a_B, e.x2b, x2b_B);
			for (int i = 0; i < 2; i++)
				add(g.uv_b, 2 * (size_t)uvid[i] + c, s.vtx_f64, a_B[i]);
		}
		double a[3], a_B[3] = {0, 0, 0};
		for (int i = 0; i < 2; i++)
			a[i] = ldv(s.shade, vid[i], s.vtx_f64);
		plane_adjoint(2, acc + 6, a, a_B, e.x2b, x2b_B);
		for (int i = 0; i < 2; i++)
			add(g.shade_b, vid[i], s.vtx_f64, a_B[i]);
	}
	else if (s.C <= 4)
	{ // unrolled under a guard, loads first (see finalize_triangle)
		double a[4][3];
#pragma unroll
		for (int c = 0; c < 4; c++)
		{
			for (int i = 0; i < 2; i++)
				a[c][i] = c < s.C ? ldv(s.colors, (size_t)vid[i] * s.C + c, s.vtx_f64) : 0.0;
			a[c][2] = 0;
		}
#pragma unroll
		for (int c = 0; c < 4; c++)
			if (c < s.C)
			{
				double a_B[3] = {0, 0, 0};
				plane_adjoint(2, acc + 3 * c, a[c], a_B, e.x2b, x2b_B);
				for (int i = 0; i < 2; i++)
					add(g.colors_b, (size_t)vid[i] * s.C + c, s.vtx_f64, a_B[i]);
			}
	}
	else
		for (int c = 0; c < s.C; c++)
		{ // H.h:1758-1767
			double a[3], a_B[3] = {0, 0, 0};
			for (int i = 0; i < 2; i++)
				a[i] = ldv(s.colors, (size_t)vid[i] * s.C + c, s.vtx_f64);
			plane_adjoint(2, acc + 3 * c, a, a_B, e.x2b, x2b_B);
			for (int i = 0; i < 2; i++)
				add(g.colors_b, (size_t)vid[i] * s.C + c, s.vtx_f64, a_B[i]);
		}
	// transparency plane: moments [x, y, 1] -> xy1_to_transp_B (H.h:1754-1755, 1771)
	const double *x2t_B = acc + 3 * (size_t)s.P;
	double V_B[2][2] = {{0, 0}, {0, 0}};
	edge_stencil_adjoint(V, V_B, s.sigma, x2b_B, x2t_B, s.clockwise);
	for (int i = 0; i < 2; i++)
		for (int d = 0; d < 2; d++)
			add(g.ij_b, 2 * (size_t)vid[i] + d, s.vtx_f64, V_B[i][d]);
}

// The same with the edge's inputs as the set-up step left them in an EdgeFin (no gather through faces / vertex arrays).  The
// caller checks fin.has_att (otherwise: finalize_edge).
template <class Add>
DR_HD void finalize_edge_fin(const SceneView &s, const GradView &g, int kind, const double x2b[6], const EdgeFin &fin, const double *acc /*[12]*/,
							 const double x2t_B[3], Add add)
{
	if (kind == KIND_NONE)
		return;
	double x2b_B[6] = {0, 0, 0, 0, 0, 0};
	if (kind == KIND_TEXTURED)
	{ // H.h:2045-2056
#pragma unroll
		for (int c = 0; c < 2; c++)
		{
			double a[3] = {fin.att[0][c], fin.att[1][c], 0}, a_B[3] = {0, 0, 0};
			plane_adjoint(2, acc + 3 * c, a, a_B, x2b, x2b_B);
			for (int i = 0; i < 2; i++)
				add(g.uv_b, 2 * (size_t)fin.uvid[i] + c, s.vtx_f64, a_B[i]);
		}
		double a[3] = {fin.att[0][2], fin.att[1][2], 0}, a_B[3] = {0, 0, 0};
		plane_adjoint(2, acc + 6, a, a_B, x2b, x2b_B);
		for (int i = 0; i < 2; i++)
			add(g.shade_b, fin.vid[i], s.vtx_f64, a_B[i]);
	}
	else
	{ // H.h:1758-1767
#pragma unroll
		for (int c = 0; c < 4; c++)
			if (c < s.C)
			{
				double a[3] = {fin.att[0][c], fin.att[1][c], 0}, a_B[3] = {0, 0, 0};
				plane_adjoint(2, acc + 3 * c, a, a_B, x2b, x2b_B);
				for (int i = 0; i < 2; i++)
					add(g.colors_b, (size_t)fin.vid[i] * s.C + c, s.vtx_f64, a_B[i]);
			}
	}
	double V_B[2][2] = {{0, 0}, {0, 0}};
	edge_stencil_adjoint(fin.V, V_B, s.sigma, x2b_B, x2t_B, s.clockwise);
	for (int i = 0; i < 2; i++)
		for (int d = 0; d < 2; d++)
			add(g.ij_b, 2 * (size_t)fin.vid[i] + d, s.vtx_f64, V_B[i][d]);
}

// ------------------------------------------------------------------------------------------------- per-pixel helpers

// texel fetch as double from a pixel-typed texture
template <class PixT>
DR_HD double ldp(const PixT *p, size_t i)
{
	return (double)p[i];
}

// colour channel c of the fragment of a drawn triangle at pixel (x, y); Z is the fragment depth (only used when
// perspective_correct).  For KIND_TEXTURED the caller prepares the tap once per pixel with `textured_tap`.
DR_HD double interp_channel(const double *planes, int c, double x, double y, bool persp, double Z)
{
	double a = plane_at(planes + 3 * c, x, y);
	return persp ? a * Z : a;
}

DR_HD void textured_tap(const double *planes, double x, double y, bool persp, double Z, int tex_w, int tex_h, int nc, Tap &tap, double &L,
						double UV[2])
{
	L = plane_at(planes + 6, x, y);
	UV[0] = plane_at(planes, x, y);
	UV[1] = plane_at(planes + 3, x, y);
	if (persp)
	{
		L = L * Z;
		UV[0] = UV[0] * Z;
		UV[1] = UV[1] * Z;
	}
	bilinear_tap(tex_w, tex_h, UV[0], UV[1], nc, tap);
}

// The four texels of a footprint, every channel, in the pixel type.  Three float32 channels (RGB: the usual texture) are 12 consecutive
// bytes: ONE load per texel instead of three -- the compiler merges the three single loads at one of the kernels' sites and not at the
// others (a lane's 12 separate gathers per footprint were 768 requests per wavefront where 256 do).
template <class PixT>
DR_HD void tap_texels(const PixT *texture, const Tap &tap, int C, PixT (&t)[4][4])
{
#if defined(__HIPCC__)
	if (C == 3 && sizeof(PixT) == 4)
	{
		typedef PixT V3 __attribute__((ext_vector_type(3), aligned(4)));
#pragma unroll
		for (int q = 0; q < 4; q++)
		{
			const V3 v = *(const V3 *)(texture + tap.idx[q]);
			t[q][0] = v.x, t[q][1] = v.y, t[q][2] = v.z, t[q][3] = 0;
		}
		return;
	}
#endif
#pragma unroll
	for (int q = 0; q < 4; q++)
#pragma unroll
		for (int c = 0; c < 4; c++)
			t[q][c] = c < C ? texture[tap.idx[q] + c] : (PixT)0;
}

template <class PixT>
DR_HD double textured_channel(const PixT *texture, const Tap &tap, int c)
{
	return bilinear_mix(tap, ldp(texture, tap.idx[0] + c), ldp(texture, tap.idx[1] + c), ldp(texture, tap.idx[2] + c), ldp(texture, tap.idx[3] + c));
}

} // namespace dr
