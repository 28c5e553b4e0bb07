// deodr_amd/csrc/dr_math.h -- per-primitive and per-pixel math of the rasterizer, shared by every HIP kernel.
//
// Everything here is branch-light scalar double arithmetic on ONE primitive or ONE pixel: the kernels in
// dr_kernels.hip are the wave-level plumbing around it (tile lists, LDS staging, cross-lane reductions, atomics).
// The functions are `__host__ __device__` so the same code can be instantiated by g++ in tests/sim/tile_sim.cpp, a
// sequential emulation of the tile pipeline that lets the algorithm be checked against the oracle on a machine
// without a GPU.  The shipped library never runs any of this on the host.
//
// Decision arithmetic (coverage spans, depth test) is kept in IEEE double with separately rounded multiply/add
// (build with -ffp-contract=off) and in the operation order of the reference, so that WHICH pixels a primitive
// covers and WHICH triangle owns a pixel are identical to the reference's single-threaded scanline code;
// only the stored image / z values are rounded (to float, when the pixel buffers are float).
//
// Reference: /root/reference/C++/DifferentiableRenderer.h, cited as H.h:<lines>.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define DR_HD __host__ __device__ __forceinline__
#else
#define DR_HD inline
#endif

namespace dr
{

// -------------------------------------------------------------------------------------------------- records in HBM

enum PrimKind : uint8_t
{
	KIND_NONE = 0,	   // not rasterized in pass 1 (culled, or textured && !shaded: H.h:2798/2813)
	KIND_INTERP = 1,   // vertex colours interpolated linearly (H.h:742-794)
	KIND_TEXTURED = 2, // bilinear texture x Gouraud shade (H.h:1043-1092)
};

struct alignas(16) TriRec // 128 bytes, one per triangle and view
{
	double eq[3][3];			  // edge equations a x + b y + c of edges (v0,v1) (v1,v2) (v2,v0), H.h:657-659
	double xZ[3];				  // plane of Z (of 1/Z when perspective_correct)
	int16_t x_min, x_max;		  // unclipped column bounds, H.h:677-686
	int16_t y_begin[2], y_end[2]; // unclipped row bounds of the upper / lower half, H.h:688-711
	uint8_t left[2], right[2];	  // which equation bounds each half on the left / right, H.h:715-738
	uint8_t kind;				  // PrimKind in pass 1
	uint8_t front;				  // signedArea > 0: owns silhouette edges and takes part in the adjoint (H.h:2847, 3063)
	uint8_t pad0[2];
	float pad1[3];
};
static_assert(sizeof(TriRec) == 128, "TriRec must stay 128 bytes (8 lanes x 16 B staging)");

struct alignas(16) EdgeRec // 128 bytes, slot 3*triangle + n
{
	double x2b[6]; // rows: barycentric coordinate along the edge of vertex 0 and of vertex 1
	double x2t[3]; // transparency T = distance to the edge / sigma
	double xZ[3];
	double key;	 // depth sum of the owning triangle: edges are blended far -> near (H.h:2781, 2841-2843)
	int32_t y_begin, y_end;
	int32_t x_begin, x_end; // conservative column bounds used only for binning
	uint8_t kind;			// KIND_NONE (slot unused), KIND_INTERP or KIND_TEXTURED
	uint8_t pad0[7];
};
static_assert(sizeof(EdgeRec) == 128, "EdgeRec must stay 128 bytes");

// What the finalize step of a drawn silhouette edge reads besides its EdgeRec (slot 3*triangle + n): the two vertices of the
// edge as the set-up step saw them, so that the adjoint algebra starts after ONE memory round trip.
struct alignas(16) EdgeFin // 128 bytes
{
	double V[2][2];	  // vertex positions, pixel-centre offset removed
	double att[2][4]; // per vertex: colour channels (nb_colors <= 4), or u, v, shade of a textured edge
	uint32_t vid[2], uvid[2];
	uint32_t has_att; // 0: attributes not stored (nb_colors > 4, or a textured-but-unshaded triangle): finalize gathers them
	uint32_t pad[3];
};
static_assert(sizeof(EdgeFin) == 128, "EdgeFin must stay 128 bytes");

// ----------------------------------------------------------------------------------------------------- 3x3 algebra

// cofactor m = s (S[a] S[b] - S[c] S[d]); row order = order of the reference's adjoint sweep (H.h:172-231)
struct Cof
{
	int8_t m, s, a, b, c, d;
};
#define DR_COF_TABLE                                                                                                         \
	{                                                                                                                        \
		{0, 1, 4, 8, 7, 5}, {3, -1, 3, 8, 6, 5}, {6, 1, 3, 7, 6, 4}, {1, -1, 1, 8, 7, 2}, {4, 1, 0, 8, 6, 2},                  \
			{7, -1, 0, 7, 6, 1}, {2, 1, 1, 5, 4, 2}, {5, -1, 0, 5, 3, 2}, {8, 1, 0, 4, 3, 1}                                   \
	}

DR_HD double cofactors3(const double S[9], double Tp[9])
{
	const Cof cof[9] = DR_COF_TABLE;
#pragma unroll
	for (int n = 0; n < 9; n++)
	{
		double v = S[cof[n].a] * S[cof[n].b] - S[cof[n].c] * S[cof[n].d];
		Tp[cof[n].m] = cof[n].s > 0 ? v : -v;
	}
	return 1 / (S[0] * Tp[0] + S[1] * Tp[3] + S[2] * Tp[6]);
}

DR_HD void inv3(const double S[9], double T[9]) // H.h:92-117
{
	double inv_det = cofactors3(S, T);
#pragma unroll
	for (int k = 0; k < 9; k++)
		T[k] *= inv_det;
}

DR_HD void inv3_adjoint(const double S[9], double S_B[9], const double T_B[9]) // H.h:124-232; S_B accumulated into
{
	const Cof cof[9] = DR_COF_TABLE;
	double Tp[9], Tp_B[9];
	double inv_det = cofactors3(S, Tp);
	double inv_det_b = 0;
#pragma unroll
	for (int k = 0; k < 9; k++)
	{
		inv_det_b += Tp[k] * T_B[k];
		Tp_B[k] = inv_det * T_B[k];
	}
	double t_B = inv_det_b * (-inv_det * inv_det);
#pragma unroll
	for (int k = 0; k < 3; k++)
	{
		S_B[k] += Tp[3 * k] * t_B;
		Tp_B[3 * k] += S[k] * t_B;
	}
#pragma unroll
	for (int n = 0; n < 9; n++)
	{
		double g = Tp_B[cof[n].m], s = cof[n].s;
		S_B[cof[n].a] += (s * S[cof[n].b]) * g;
		S_B[cof[n].b] += (s * S[cof[n].a]) * g;
		S_B[cof[n].c] += (-s * S[cof[n].d]) * g;
		S_B[cof[n].d] += (-s * S[cof[n].c]) * g;
	}
}

// value at column 0 of scanline y of the plane p = [px, py, p1]; summation order of the reference's row setup
// (dot with t = {0, y, 1}: H.h:929-934, 1596-1598)
// (the reference's dot product also adds p[0] * 0 in front: for finite coefficients that term is +0 and changes nothing but
// the sign of an exact zero, so it is not evaluated -- three double operations per plane evaluation)
DR_HD double row0(const double p[3], double y) { return p[1] * y + p[2]; }
DR_HD double plane_at(const double p[3], double x, double y) { return row0(p, y) + p[0] * x; }

// plane coefficient j of an attribute given at nv vertices: sum_k a[k] x2b[3k + j]   (H.h:779-788, 1577-1585)
DR_HD double plane_coef(int nv, const double a[3], const double *x2b, int j)
{
	double s = 0;
	for (int k = 0; k < nv; k++)
		s += a[k] * x2b[3 * k + j];
	return s;
}

// ------------------------------------------------------------------------------------------- robust integer division

#define DR_SHRT_MAX 32767

// The reference's fallback for |a / b| beyond the 16-bit range (or b == 0) walks x upward from x_min while a predicate
// P(x) holds (H.h:459-477, 499-517): up to `width` iterations.  P(x) compares the correctly rounded product (x + 1) * b
// with a, which is monotone in x, so the first x where P fails is found by bisection with the identical result -- a
// 1000-iteration data-dependent loop in one lane would stall its whole wavefront.
template <class Pred>
DR_HD int first_failing(int x_min, int x_max, Pred holds)
{
	int lo = x_min, hi = x_max; // the walk stops at x_max whatever P says there
	while (lo < hi)
	{
		const int mid = lo + ((hi - lo) >> 1);
		if (holds(mid))
			lo = mid + 1;
		else
			hi = mid;
	}
	return lo;
}

// the reference's slow walks (never taken by a usual frame): ONE out-of-line copy per kernel instead of one inlined at every call site of the
// raster kernels (DR_SLOW_DIV_OUTLINE = 0 builds the neighbour)
#ifndef DR_SLOW_DIV_OUTLINE
#define DR_SLOW_DIV_OUTLINE 1
#endif
#if defined(__HIPCC__) && DR_SLOW_DIV_OUTLINE
#define DR_SLOW __host__ __device__ __attribute__((noinline))
#else
#define DR_SLOW DR_HD
#endif
DR_SLOW int div_slow_walk(double a, double b, int x_min, int x_max, int ceil_mode)
{
	if (x_min >= x_max)
		return x_min;
	if (!ceil_mode)
		return b > 0 ? first_failing(x_min, x_max, [=](int t) { return (t + 1) * b <= a; }) : first_failing(x_min, x_max, [=](int t) { return (t + 1) * b >= a; });
	return b > 0 ? first_failing(x_min, x_max, [=](int t) { return (t + 1) * b < a; }) : first_failing(x_min, x_max, [=](int t) { return (t + 1) * b > a; });
}

DR_HD int floor_div(double a, double b, int x_min, int x_max) // H.h:440-479
{
	int x;
	if (fabs(b) * DR_SHRT_MAX > fabs(a) + fabs(b))
	{
		x = (int)(int16_t)floor(a / b);
		if (x < x_min)
			x = x_min;
		if (x > x_max)
			x = x_max;
	}
	else
		x = div_slow_walk(a, b, x_min, x_max, 0);
	return x;
}

DR_HD int ceil_div(double a, double b, int x_min, int x_max) // H.h:481-519
{
	int x;
	if (fabs(b) * DR_SHRT_MAX > fabs(a) + fabs(b))
	{
		x = (int)(int16_t)ceil(a / b);
		if (x < x_min)
			x = x_min;
		if (x > x_max)
			x = x_max;
	}
	else
		x = div_slow_walk(a, b, x_min, x_max, 1);
	return x;
}

// floor_div / ceil_div WITHOUT the IEEE division, for the usual quotient (round 6): branch-free, and `ok` is cleared when the result is
// not PROVEN equal to the exact function's -- the caller then repeats the whole span pass with the exact functions (a wave-uniform
// branch around a second copy of the pass: a per-call fallback inside these functions put a division behind a branch at every one of
// the ~40 inlined call sites of the forward raster, whose edge-free walkers then spilled registers: 0.1119 -> 0.1175 ms per step).
// q~ = a * r with r = 1 / b from the hardware reciprocal (v_rcp_f64: relative error < 2^-24, tools/probes/rcp_probe.hip; 2^-20 is
// assumed here) and ONE Newton step, so r is good to 2^-39 and, as |a / b| < 2^15 behind the reference's own guard,
// |q~ - a / b| < 2^-23.  fl(a / b) is within half an ulp (< 2^-38) of a / b.  When q~ lies at least 2^-20 away from both neighbouring
// integers, a / b and fl(a / b) lie strictly between the same two integers: floor(fl(a / b)) = floor(q~) and ceil(fl(a / b)) =
// floor(q~) + 1, bit for bit.  Otherwise (integer vertex coordinates do that: the quotient IS an integer, and fl() of a quotient just
// below one may round up to it), or when the guard fails (the reference's slow walk), or with a NaN / infinite intermediate (both
// comparisons fail): ok = false.  On the device a division is ~14 double instructions behind a quarter-rate reciprocal, twice per
// (triangle, row) lane of a span pass and four times per (edge, row).
DR_HD double quick_floor_quotient(double a, double b, bool &ok)
{
#if defined(__HIP_DEVICE_COMPILE__)
	double r = __builtin_amdgcn_rcp(b); // v_rcp_f64
#else
	double r = 1.0 / b;
#endif
	r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
	const double q = a * r, fl = floor(q), d = q - fl; // (d: exact)
	ok = ok & (fabs(b) * DR_SHRT_MAX > fabs(a) + fabs(b)) & (d > 0x1p-20) & (d < 1.0 - 0x1p-20);
	return fl;
}
DR_HD int clamp_i16(double v, int x_min, int x_max)
{ // (the conversions of the reference's fast path; a garbage v of a lane whose `ok` is false is never used)
	int x = (int)(int16_t)(int)v;
	x = x < x_min ? x_min : x;
	return x > x_max ? x_max : x;
}
DR_HD int floor_div_quick(double a, double b, int x_min, int x_max, bool &ok) { return clamp_i16(quick_floor_quotient(a, b, ok), x_min, x_max); }
DR_HD int ceil_div_quick(double a, double b, int x_min, int x_max, bool &ok) { return clamp_i16(quick_floor_quotient(a, b, ok) + 1, x_min, x_max); }

// ------------------------------------------------------------------------------------------------- triangle stencil

DR_HD double signed_area(const double V[3][2], bool clockwise) // H.h:391-398
{
	double ux = V[1][0] - V[0][0], uy = V[1][1] - V[0][1];
	double vx = V[2][0] - V[0][0], vy = V[2][1] - V[0][1];
	return 0.5 * (ux * vy - vx * uy) * (clockwise ? 1 : -1);
}

DR_HD void edge_equation(double e[3], const double v1[2], const double v2[2], bool clockwise) // H.h:373-389
{
	if (clockwise)
	{
		e[0] = (v1[1] - v2[1]);
		e[1] = (v2[0] - v1[0]);
	}
	else
	{
		e[0] = (v2[1] - v1[1]);
		e[1] = (v1[0] - v2[0]);
	}
	e[2] = -0.5 * (e[0] * (v1[0] + v2[0]) + e[1] * (v1[1] + v2[1]));
}

DR_HD void sort3(const double v[3], double sv[3], int order[3]) // H.h:400-426
{
	for (int k = 0; k < 3; k++)
	{
		sv[k] = v[k];
		order[k] = k;
	}
#define DR_CSWAP(a, b)                                                                                                       \
	if (sv[a] > sv[b])                                                                                                       \
	{                                                                                                                        \
		double tv = sv[a];                                                                                                   \
		sv[a] = sv[b];                                                                                                       \
		sv[b] = tv;                                                                                                          \
		int ti = order[a];                                                                                                   \
		order[a] = order[b];                                                                                                 \
		order[b] = ti;                                                                                                       \
	}
	DR_CSWAP(0, 1)
	DR_CSWAP(0, 2)
	DR_CSWAP(1, 2)
#undef DR_CSWAP
}

DR_HD void bary_frame(const double V[3][2], double b2x[9]) // bary_to_xy1, H.h:645-649
{
	for (int v = 0; v < 3; v++)
	{
		b2x[v] = V[v][0];
		b2x[3 + v] = V[v][1];
		b2x[6 + v] = 1;
	}
}

// get_triangle_stencil_equations, H.h:633-739.  V already has the pixel-centre offset removed.
DR_HD void tri_stencil(const double V[3][2], bool strict, TriRec &r, double x2b[9])
{
	double b2x[9];
	bary_frame(V, b2x);
	inv3(b2x, x2b);
	bool cw = signed_area(V, true) > 0;
	edge_equation(r.eq[0], V[0], V[1], cw);
	edge_equation(r.eq[1], V[1], V[2], cw);
	edge_equation(r.eq[2], V[2], V[0], cw);
	double xs[3] = {V[0][0], V[1][0], V[2][0]}, ys[3] = {V[0][1], V[1][1], V[2][1]}, sx[3], sy[3];
	int ox[3], oy[3];
	sort3(xs, sx, ox);
	sort3(ys, sy, oy);
	r.x_min = strict ? (int16_t)floor(sx[0]) : (int16_t)ceil(sx[0]);
	r.x_max = (int16_t)floor(sx[2]);
	r.y_begin[0] = strict ? (int16_t)((int16_t)floor(sy[0]) + 1) : (int16_t)ceil(sy[0]);
	r.y_end[0] = (int16_t)floor(sy[1]);
	r.y_begin[1] = strict ? (int16_t)((int16_t)floor(sy[1]) + 1) : (int16_t)ceil(sy[1]);
	r.y_end[1] = (int16_t)floor(sy[2]);
	// (selects instead of r.eq[id][0]: a record kept in registers must not be indexed dynamically)
	int id = oy[0];
	const double ea0 = r.eq[0][0], ea1 = r.eq[1][0], ea2 = r.eq[2][0]; // values, not lvalues: see pick3 in dr_prims.h
	if ((id == 0 ? ea0 : (id == 1 ? ea1 : ea2)) > 0)
	{
		r.right[0] = (uint8_t)((id + 2) % 3);
		r.left[0] = (uint8_t)(id % 3);
	}
	else
	{
		r.right[0] = (uint8_t)(id % 3);
		r.left[0] = (uint8_t)((id + 2) % 3);
	}
	id = oy[2];
	if ((id == 0 ? ea0 : (id == 1 ? ea1 : ea2)) < 0)
	{
		r.right[1] = (uint8_t)(id % 3);
		r.left[1] = (uint8_t)((id + 2) % 3);
	}
	else
	{
		r.right[1] = (uint8_t)((id + 2) % 3);
		r.left[1] = (uint8_t)(id % 3);
	}
}

// Columns [xb, xe] of scanline y covered by one half of the triangle: get_xrange, H.h:864-906 (left edge exclusive
// when strict, right edge inclusive).  Returns an empty span (xb > xe) when the row is outside the half.
// QUICK: the divisions by reciprocal (floor_div_quick); *ok is cleared when a result is not proven exact -- the caller repeats the pass.
template <bool QUICK = false>
DR_HD void tri_half_span(const TriRec &r, int half, int y, int width, int height, bool strict, int &xb, int &xe, bool *ok = nullptr)
{
	int yb = r.y_begin[half] < 0 ? 0 : r.y_begin[half];
	int ye = r.y_end[half] > height - 1 ? height - 1 : r.y_end[half];
	xb = 1;
	xe = 0;
	if (y < yb || y > ye)
		return;
	int x_min = r.x_min < 0 ? 0 : r.x_min;
	int x_max = r.x_max > width - 1 ? width - 1 : r.x_max;
	const double *left = r.eq[r.left[half]], *right = r.eq[r.right[half]];
	xb = x_min;
	xe = x_max;
	double num = -(left[1] * y + left[2]);
	int t;
	if (QUICK)
		t = strict ? 1 + floor_div_quick(num, left[0], x_min - 1, x_max, *ok) : ceil_div_quick(num, left[0], x_min - 1, x_max, *ok);
	else
		t = strict ? 1 + floor_div(num, left[0], x_min - 1, x_max) : ceil_div(num, left[0], x_min - 1, x_max);
	if (t > xb)
		xb = t;
	num = -(right[1] * y + right[2]);
	t = QUICK ? floor_div_quick(num, right[0], x_min - 1, x_max, *ok) : floor_div(num, right[0], x_min - 1, x_max);
	if (t < xe)
		xe = t;
}

// true when pixel (x, y) is rasterized by the triangle (either half; the halves only overlap on the middle-vertex row
// in non-strict mode, where both give the same answer up to the reference's own double draw)
DR_HD bool tri_covers(const TriRec &r, int x, int y, int width, int height, bool strict)
{
	for (int half = 0; half < 2; half++)
	{
		int xb, xe;
		tri_half_span(r, half, y, width, height, strict, xb, xe);
		if (x >= xb && x <= xe)
			return true;
	}
	return false;
}

// ----------------------------------------------------------------------------------------------------- edge stencil

DR_HD void edge_normal(const double V[2][2], bool clockwise, double nt[2], double &inv_norm) // H.h:1383-1393
{
	if (clockwise)
	{
		nt[0] = V[0][1] - V[1][1];
		nt[1] = V[1][0] - V[0][0];
	}
	else
	{
		nt[0] = V[1][1] - V[0][1];
		nt[1] = V[0][0] - V[1][0];
	}
	inv_norm = 1 / sqrt(nt[0] * nt[0] + nt[1] * nt[1]);
}

DR_HD void edge_frame(const double V[2][2], const double n[2], double e2x[9]) // edge_to_xy1, H.h:1397-1404
{
	for (int v = 0; v < 2; v++)
	{
		e2x[v] = V[v][0];
		e2x[3 + v] = V[v][1];
		e2x[6 + v] = 1;
	}
	e2x[2] = n[0];
	e2x[5] = n[1];
	e2x[8] = 0;
}

// get_edge_stencil_equations, H.h:1366-1460 (+ a conservative column range for binning)
DR_HD void edge_stencil(const double V[2][2], int height, int width, double sigma, bool clockwise, EdgeRec &r)
{
	double nt[2], inv_norm, n[2], e2x[9], x2e[9];
	edge_normal(V, clockwise, nt, inv_norm);
	n[0] = nt[0] * inv_norm;
	n[1] = nt[1] * inv_norm;
	edge_frame(V, n, e2x);
	inv3(e2x, x2e);
	for (int k = 0; k < 6; k++)
		r.x2b[k] = x2e[k];
	for (int k = 0; k < 3; k++)
		r.x2t[k] = (1 / sigma) * x2e[6 + k];
	int y_begin = height - 1;
	for (int k = 0; k < 2; k++)
		if (V[k][1] - sigma < y_begin)
			y_begin = (int)floor(V[k][1] - sigma) + 1;
	if (y_begin < 0)
		y_begin = 0;
	int y_end = 0;
	for (int k = 0; k < 2; k++)
		if (V[k][1] + sigma > y_end)
			y_end = (int)floor(V[k][1] + sigma);
	if (y_end > height - 1)
		y_end = height - 1;
	r.y_begin = y_begin;
	r.y_end = y_end;
	// the band is the parallelogram V0, V1, V0 + sigma n, V1 + sigma n; one extra pixel absorbs rounding
	double xlo = fmin(V[0][0], V[1][0]) - sigma - 1, xhi = fmax(V[0][0], V[1][0]) + sigma + 1;
	r.x_begin = xlo < 0 ? 0 : (xlo > width ? width : (int)floor(xlo));
	r.x_end = xhi > width - 1 ? width - 1 : (xhi < -1 ? -1 : (int)ceil(xhi));
}

// get_edge_xrange_from_ineq, H.h:2620-2648.  The four half-planes are bary0 > 0, bary1 > 0, T > 0, 1 - T > 0
// (rows built at H.h:1418-1435).
template <bool QUICK = false> // (see tri_half_span)
DR_HD void edge_row_span(const EdgeRec &r, int y, int width, int &xb, int &xe, bool *ok = nullptr)
{
	xb = 0;
	xe = width - 1;
#pragma unroll
	for (int k = 0; k < 4; k++)
	{
		double a, b, c;
		if (k < 2)
		{
			a = r.x2b[3 * k];
			b = r.x2b[3 * k + 1];
			c = r.x2b[3 * k + 2];
		}
		else if (k == 2)
		{
			a = r.x2t[0];
			b = r.x2t[1];
			c = r.x2t[2];
		}
		else
		{
			a = -r.x2t[0];
			b = -r.x2t[1];
			c = (1 - r.x2t[2]);
		}
		const double num = -(b * y + c);
		// one division whatever the sign of a (on the GPU both sides of a branch run)
		const int t = QUICK ? floor_div_quick(num, a, xb - 1, xe + 1, *ok) : floor_div(num, a, xb - 1, xe + 1);
		if (a < 0)
			xe = t < xe ? t : xe;
		else
			xb = t + 1 > xb ? t + 1 : xb;
	}
}

DR_HD bool edge_covers(const EdgeRec &r, int x, int y, int width)
{
	if (y < r.y_begin || y > r.y_end)
		return false;
	int xb, xe;
	edge_row_span(r, y, width, xb, xe);
	return x >= xb && x <= xe;
}

// adjoint of the edge frame: get_edge_stencil_equations_B, H.h:1462-1539.  V_B is accumulated into.
DR_HD void edge_stencil_adjoint(const double V[2][2], double V_B[2][2], double sigma, const double x2b_B[6], const double x2t_B[3],
								bool clockwise)
{
	double nt[2], inv_norm, n[2], e2x[9], e2x_B[9], x2e_B[9];
	edge_normal(V, clockwise, nt, inv_norm);
	n[0] = nt[0] * inv_norm;
	n[1] = nt[1] * inv_norm;
	edge_frame(V, n, e2x);
	for (int k = 0; k < 9; k++)
		e2x_B[k] = 0;
	for (int k = 0; k < 6; k++)
		x2e_B[k] = x2b_B[k];
	for (int k = 0; k < 3; k++)
		x2e_B[6 + k] = x2t_B[k] * (1 / sigma);
	inv3_adjoint(e2x, e2x_B, x2e_B);
	for (int v = 0; v < 2; v++)
		for (int d = 0; d < 2; d++)
			V_B[v][d] += e2x_B[3 * d + v];
	double n_B[2] = {e2x_B[2], e2x_B[5]}, nt_B[2] = {0, 0}, inv_norm_B = 0;
	for (int k = 0; k < 2; k++)
	{
		nt_B[k] += n_B[k] * inv_norm;
		inv_norm_B += n_B[k] * nt[k];
	}
	double nor_B = -inv_norm_B * (inv_norm * inv_norm);
	double nor_s_B = nor_B * 0.5 * inv_norm;
	nt_B[0] += 2 * nt[0] * nor_s_B;
	nt_B[1] += 2 * nt[1] * nor_s_B;
	double sgn = clockwise ? 1.0 : -1.0;
	V_B[0][1] += sgn * nt_B[0];
	V_B[1][1] += -sgn * nt_B[0];
	V_B[1][0] += sgn * nt_B[1];
	V_B[0][0] += -sgn * nt_B[1];
}

// -------------------------------------------------------------------------------------------------- texture sampling

struct Tap
{
	int idx[4]; // texel offsets 00 10 01 11, already multiplied by the channel count
	double e[2];
	bool out[2];
};

// H.h:527-556: clamp-to-edge bilinear footprint; u runs along the texture width, v along its height
DR_HD void bilinear_tap(int tex_w, int tex_h, double u, double v, int nc, Tap &t)
{
	const int size[2] = {tex_w, tex_h};
	const double p[2] = {u, v};
	int fp[2];
	for (int k = 0; k < 2; k++)
	{
		fp[k] = (int)floor(p[k]);
		t.e[k] = p[k] - fp[k];
		t.out[k] = false;
		if (fp[k] < 0)
		{
			t.out[k] = true;
			fp[k] = 0;
			t.e[k] = 0;
		}
		if (fp[k] > size[k] - 2)
		{
			t.out[k] = true;
			fp[k] = size[k] - 2;
			t.e[k] = 1;
		}
	}
	t.idx[0] = nc * (fp[0] + tex_w * fp[1]);
	t.idx[1] = nc * (fp[0] + 1 + tex_w * fp[1]);
	t.idx[2] = nc * (fp[0] + tex_w * (fp[1] + 1));
	t.idx[3] = nc * (fp[0] + 1 + tex_w * (fp[1] + 1));
}

DR_HD double bilinear_mix(const Tap &t, double i00, double i10, double i01, double i11) // H.h:559
{
	return ((1 - t.e[0]) * i00 + t.e[0] * i10) * (1 - t.e[1]) + ((1 - t.e[0]) * i01 + t.e[0] * i11) * t.e[1];
}

// one channel of bilinear_sample_B (H.h:607-625): weights of the four texels and the footprint adjoint
DR_HD void bilinear_mix_adjoint(const Tap &t, double a_b, double i00, double i10, double i01, double i11, double w[4], double e_B[2])
{
	double t1 = ((1 - t.e[0]) * i00 + t.e[0] * i10);
	double t2 = ((1 - t.e[0]) * i01 + t.e[0] * i11);
	e_B[1] += -a_b * t1;
	e_B[1] += a_b * t2;
	double t1_B = a_b * (1 - t.e[1]);
	double t2_B = a_b * t.e[1];
	e_B[0] += t1_B * (i10 - i00);
	e_B[0] += t2_B * (i11 - i01);
	w[0] = (1 - t.e[0]) * (1 - t.e[1]) * a_b;
	w[1] = t.e[0] * (1 - t.e[1]) * a_b;
	w[2] = (1 - t.e[0]) * t.e[1] * a_b;
	w[3] = t.e[0] * t.e[1] * a_b;
}

// ------------------------------------------------------------------------------------- per-primitive adjoint (finalize)
//
// Every per-pixel gradient contribution of a primitive is linear in [x, y, 1]; the raster kernels therefore only
// accumulate, per plane, the three image moments  M = sum g [x, y, 1]  (exactly what the reference builds row by row in
// xy1_to_A_B, H.h:1029-1037).  These helpers turn the accumulated plane adjoints into vertex adjoints.

// adjoint of  plane[j] = sum_k a[k] x2b[3k+j]   (H.h:841-851): a_B[k] += ..., x2b_B += ...
DR_HD void plane_adjoint(int nv, const double plane_B[3], const double a[3], double a_B[3], const double *x2b, double *x2b_B)
{
	for (int j = 0; j < 3; j++)
		for (int k = 0; k < nv; k++)
		{
			a_B[k] += plane_B[j] * x2b[3 * k + j];
			x2b_B[3 * k + j] += a[k] * plane_B[j];
		}
}

} // namespace dr
