// deodr_amd/csrc/dr_backward_generic.h -- part of the single translation unit dr_kernels.hip (device code, gfx950 / wave64).
// raster_bwd_kernel: the adjoint raster without LDS staging (nb_colors > 4, antialiase_error).
#pragma once

#include "dr_forward.h"

using namespace dr;

namespace
{

// ------------------------------------------------------------------------------------------------ backward raster

// adds  sum over the wave of  v * [x, y, 1]  to acc[0..2]
// (det: KParams::det, the deterministic mode of the un-staged kernels -- a compile-time false wherever this is inlined into a staged kernel)
__device__ __forceinline__ void add_moments(double *acc, double v, double x, double y, int lane, uint32_t *det = nullptr)
{
	double mx = wave_sum(v * x), my = wave_sum(v * y), m1 = wave_sum(v);
	if (lane == 0)
	{
		if (mx != 0)
			acc_add(acc + 0, mx, det);
		if (my != 0)
			acc_add(acc + 1, my, det);
		if (m1 != 0)
			acc_add(acc + 2, m1, det);
	}
}

template <class PixT>
__device__ __forceinline__ void texture_scatter(PixT *texture_b, const Tap &tap, int c, const double wgt[4], long long *det_texture = nullptr, uint32_t *det_err = nullptr)
{
#pragma unroll
	for (int q = 0; q < 4; q++)
		if (wgt[q] != 0)
		{
			if (det_texture)
				det_add(det_texture + tap.idx[q] + c, wgt[q], det_err);
			else
				unsafeAtomicAdd(texture_b + tap.idx[q] + c, (PixT)wgt[q]);
		}
}

// adjoint of one tile, any channel count / edge count / mode; `order` is a per-wave LDS array of MAX_SORTED entries.
// LEAN: the instance inlined into raster_bwd_edge_kernel for the (pathological) tiles with more than EMAX edges: at most CH
// channels and no antialiase_error, which the compiler can then drop.
template <class PixT, bool LEAN, bool TEX = true>
__device__ __forceinline__ void bwd_tile_generic_impl(const KParams &p, int view, int tx, int ty, int lane, volatile uint32_t *order)
{
	const ViewPtrs w = view_ptrs(p, view);
	const int W = p.W, H = p.H, C = p.C, P = p.L.P;
	const bool aa_err = !LEAN && p.aa_err;
	uint32_t *const det = (!LEAN && p.det) ? p.det_err : nullptr; // (the deterministic mode runs on raster_bwd_kernel only)
	long long *const det_tex = det ? p.det_texture : nullptr;
	const PixT *texture = (const PixT *)p.texture;
	PixT *texture_b = (PixT *)p.texture_b;
	const int tile = ty * p.L.tiles_x + tx;
	const int px = tx * TILE + (lane & 7), py = ty * TILE + (lane >> 3);
	const bool inb = px < W && py < H;
	const size_t pix = (size_t)py * W + px;
	const size_t vpix = (size_t)view * H * W + pix;
	const double x = px, y = py;
	const int nedge = uniform((int)(w.edge_saved[tile] & ~SWEEP_SAVED));
	int owner = -1, kind = KIND_NONE;
	if (inb)
		unpack_owner(w.face_id[pix], owner, kind);
	if (__ballot(owner >= 0) == 0 && nedge == 0)
		return;

	// what pass 1 left at this pixel
	const double *planes = nullptr;
	double zown = INFINITY;
	Tap tap;
	double L = 0, UV[2] = {0, 0};
	if (owner >= 0)
	{
		const TriRec &r = w.tri_rec[owner];
		planes = w.tri_planes + (size_t)owner * 3 * P;
		zown = plane_at(r.xZ, x, y);
		if (kind == KIND_TEXTURED && TEX)
			textured_tap(planes, x, y, false, zown, p.tex_w, p.tex_h, C, tap, L, UV);
	}
	auto base_channel = [&](int c) -> double { // un-antialiased colour of the pixel
		if (owner < 0)
			return inb ? background_channel<PixT>(p, view, pix, c) : 0.0;
		if (kind == KIND_TEXTURED && TEX)
			return textured_channel(texture, tap, c) * L;
		return interp_channel(planes, c, x, y, false, zown);
	};

	// edge order + which edges touch this pixel
	uint32_t edge_spill_n = 0;
	if (nedge > K_EDGE)
	{
		edge_spill_n = w.hdr->edge_spill[w.hdr->cur];
		if (edge_spill_n > p.L.edge_pool_cap)
			edge_spill_n = p.L.edge_pool_cap;
	}
	const bool cached = nedge <= MAX_SORTED;
	unsigned long long touched = 0;
	int n_sorted = nedge;
	if (nedge > 0 && cached)
	{
		EdgeCursor cur = {0, 0};
		for (int r = 0; r < nedge; r++)
		{
			EdgeCursor f;
			uint32_t slot = next_edge(w, tile, nedge, edge_spill_n, r == 0, cur, false, lane, f);
			if (slot == 0xffffffffu)
			{
				n_sorted = r;
				break;
			}
			if (lane == 0)
				order[r] = slot;
			cur = f;
			if (edge_touches(w.edge_rec[slot], px, py, W, false, zown, inb))
				touched |= 1ull << r;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
	// r-th edge of the tile in blending order (cached in LDS, or searched when the tile has more than MAX_SORTED edges)
	auto edge_at = [&](int r) -> uint32_t {
		if (cached)
			return (uint32_t)uniform((int)order[r]);
		EdgeCursor cur = {0, 0}, f;
		uint32_t slot = 0;
		for (int i = 0; i <= r; i++)
		{
			slot = next_edge(w, tile, nedge, edge_spill_n, i == 0, cur, false, lane, f);
			cur = f;
		}
		return (uint32_t)uniform((int)slot);
	};
	auto is_touched = [&](int r, uint32_t slot) -> bool {
		if (cached)
			return (touched >> r) & 1ull;
		return slot != 0xffffffffu && edge_touches(w.edge_rec[slot], px, py, W, false, zown, inb);
	};

	// per-pixel scalar adjoints that sum over channels (textured owner): accumulated across the channel chunks
	double own_L_B = 0, own_e_B[2] = {0, 0};

	// ---- antialiase_error mode: the edges blended the squared residual err_buffer, not the image (H.h:2200-2368, 2481-2618)
	double eb = 0; // running adjoint of err_buffer at this pixel
	if (aa_err)
	{
		const PixT *obs = (const PixT *)p.obs + vpix * C;
		eb = inb ? (double)((const PixT *)p.err_b)[vpix] : 0.0;
		if (nedge > 0)
		{
			double err0 = 0; // residual before any edge: sum_c (image - obs)^2 with the un-antialiased image (H.h:2824-2837)
			if (inb)
				for (int c = 0; c < C; c++)
				{
					double d = base_channel(c) - (double)obs[c];
					err0 += d * d;
				}
			// squared distance between the colour an edge would paint here and the observation
			auto edge_err = [&](const EdgeRec &e, const double *ep, const Tap &etap, double eL) -> double {
				double Err = 0;
				for (int c = 0; c < C; c++)
				{
					double d = edge_channel<PixT, TEX>(e, ep, texture, etap, eL, c, x, y, false, 0.0) - (double)obs[c];
					Err += d * d;
				}
				return Err;
			};
			for (int r = n_sorted - 1; r >= 0; r--)
			{
				const uint32_t slot = edge_at(r);
				if (slot == 0xffffffffu)
					continue;
				const bool hit = is_touched(r, slot);
				if (__ballot(hit) == 0)
					continue;
				const EdgeRec &e = w.edge_rec[slot];
				const double *ep = w.edge_planes + (size_t)slot * 3 * P;
				double *eacc = w.edge_acc + (size_t)slot * (3 * P + 3);
				double prev = err0; // err_buffer before this edge: replay of the earlier edges
				for (int q = 0; q < r; q++)
				{
					const uint32_t sq = edge_at(q);
					if (sq == 0xffffffffu || !is_touched(q, sq))
						continue;
					const EdgeRec &eq = w.edge_rec[sq];
					const double *qp = w.edge_planes + (size_t)sq * 3 * P;
					const double Tq = plane_at(eq.x2t, x, y);
					Tap qtap;
					double qL = 0, qUV[2];
					if (eq.kind == KIND_TEXTURED && TEX)
						textured_tap(qp, x, y, false, 0.0, p.tex_w, p.tex_h, C, qtap, qL, qUV);
					prev *= Tq;
					prev += (1 - Tq) * edge_err(eq, qp, qtap, qL);
				}
				const double Tr = plane_at(e.x2t, x, y);
				Tap etap;
				double eL = 0, eUV[2] = {0, 0};
				if (e.kind == KIND_TEXTURED && TEX && hit)
					textured_tap(ep, x, y, false, 0.0, p.tex_w, p.tex_h, C, etap, eL, eUV);
				double T_B = 0, L_B = 0, e_B[2] = {0, 0}, Err_B = 0;
				if (hit)
				{
					const double Err = edge_err(e, ep, etap, eL);
					T_B = eb * (prev - Err);
					Err_B = (1 - Tr) * eb;
					eb *= Tr;
				}
				for (int c = 0; c < C; c++)
				{
					double A_B = 0;
					if (hit)
					{
						if (e.kind == KIND_TEXTURED && TEX)
						{ // H.h:2315-2326
							const double i00 = ldp(texture, etap.idx[0] + c), i10 = ldp(texture, etap.idx[1] + c);
							const double i01 = ldp(texture, etap.idx[2] + c), i11 = ldp(texture, etap.idx[3] + c);
							const double A = bilinear_mix(etap, i00, i10, i01, i11);
							const double diff_B = 2 * (A * eL - (double)obs[c]) * Err_B;
							L_B += diff_B * A;
							double wgt[4];
							bilinear_mix_adjoint(etap, diff_B * eL, i00, i10, i01, i11, wgt, e_B);
							if (texture_b)
								texture_scatter(texture_b, etap, c, wgt, det_tex, det);
						}
						else // H.h:2579-2588, with the row fold the reference forgot (defect D2) restored
							A_B = 2 * (interp_channel(ep, c, x, y, false, 0.0) - (double)obs[c]) * Err_B;
					}
					if (e.kind != KIND_TEXTURED || !TEX)
						add_moments(eacc + 3 * c, A_B, x, y, lane, det);
				}
				if (e.kind == KIND_TEXTURED && TEX)
				{
					add_moments(eacc + 0, (hit && !etap.out[0]) ? e_B[0] : 0.0, x, y, lane, det);
					add_moments(eacc + 3, (hit && !etap.out[1]) ? e_B[1] : 0.0, x, y, lane, det);
					add_moments(eacc + 6, L_B, x, y, lane, det);
				}
				add_moments(eacc + 3 * P, T_B, x, y, lane, det);
			}
		}
	}
	{
		for (int c0 = 0; c0 < (LEAN ? 1 : C); c0 += CH)
		{
			double g[CH], base[CH];
#pragma unroll
			for (int j = 0; j < CH; j++)
			{
				g[j] = 0;
				base[j] = 0;
				if (c0 + j < C && inb)
				{
					if (aa_err) // image_b = -2 (obs - image) err_buffer_b, H.h:3054-3060
						g[j] = -2 * ((double)((const PixT *)p.obs)[vpix * C + c0 + j] - base_channel(c0 + j)) * eb;
					else
						g[j] = p.image_b ? (double)((const PixT *)p.image_b)[vpix * C + c0 + j]
										 : fit_residual<true>(p, (double)((const PixT *)p.image_in)[vpix * C + c0 + j], (double)((const PixT *)p.obs)[vpix * C + c0 + j]);
				}
			}
			if (nedge > 0 && !aa_err)
			{
#pragma unroll
				for (int j = 0; j < CH; j++)
					if (c0 + j < C)
						base[j] = base_channel(c0 + j);
				// antialiased colour of the pixel: one forward sweep over the edges that touch it
				double aa[CH];
#pragma unroll
				for (int j = 0; j < CH; j++)
					aa[j] = base[j];
				for (int q = 0; q < n_sorted; q++)
				{
					const uint32_t sq = edge_at(q);
					if (sq == 0xffffffffu || !is_touched(q, sq))
						continue;
					const EdgeRec &eq = w.edge_rec[sq];
					const double *qp = w.edge_planes + (size_t)sq * 3 * P;
					const double Tq = plane_at(eq.x2t, x, y);
					Tap qtap;
					double qL = 0, qUV[2];
					if (eq.kind == KIND_TEXTURED && TEX)
						textured_tap(qp, x, y, false, 0.0, p.tex_w, p.tex_h, C, qtap, qL, qUV);
#pragma unroll
					for (int j = 0; j < CH; j++)
						if (c0 + j < C)
						{
							aa[j] *= Tq;
							aa[j] += (1 - Tq) * edge_channel<PixT, TEX>(eq, qp, texture, qtap, qL, c0 + j, x, y, false, 0.0);
						}
				}
				// adjoint of pass 2: near -> far (H.h:2961-3052)
				for (int r = n_sorted - 1; r >= 0; r--)
				{
					const uint32_t slot = edge_at(r);
					if (slot == 0xffffffffu)
						continue; // only when the spill pool overflowed (the host then repeats the call)
					const bool hit = is_touched(r, slot);
					if (__ballot(hit) == 0)
						continue;
					const EdgeRec &e = w.edge_rec[slot];
					const double *ep = w.edge_planes + (size_t)slot * 3 * P;
					double *eacc = w.edge_acc + (size_t)slot * (3 * P + 3);
					double prev[CH];
#pragma unroll
					for (int j = 0; j < CH; j++)
						prev[j] = base[j];
					// colour before this edge: un-blend the running antialiased colour like the reference (H.h:1738) when T
					// is safely away from 0, otherwise replay the earlier edges from the un-antialiased colour
					const double Tr_here = hit ? plane_at(e.x2t, x, y) : 1.0;
					const bool need_replay = hit && !(Tr_here > 1e-6);
					if (hit && !need_replay)
					{
						Tap utap;
						double uL = 0, uUV[2];
						if (e.kind == KIND_TEXTURED && TEX)
							textured_tap(ep, x, y, false, 0.0, p.tex_w, p.tex_h, C, utap, uL, uUV);
#pragma unroll
						for (int j = 0; j < CH; j++)
							if (c0 + j < C)
							{
								prev[j] = (aa[j] - (1 - Tr_here) * edge_channel<PixT, TEX>(e, ep, texture, utap, uL, c0 + j, x, y, false, 0.0)) / Tr_here;
								aa[j] = prev[j];
							}
					}
					if (__ballot(need_replay))
					for (int q = 0; q < r; q++)
					{
						const uint32_t sq = edge_at(q);
						if (!need_replay || sq == 0xffffffffu || !is_touched(q, sq))
							continue;
						const EdgeRec &eq = w.edge_rec[sq];
						const double *qp = w.edge_planes + (size_t)sq * 3 * P;
						const double Tq = plane_at(eq.x2t, x, y);
						Tap qtap;
						double qL = 0, qUV[2];
						if (eq.kind == KIND_TEXTURED && TEX)
							textured_tap(qp, x, y, false, 0.0, p.tex_w, p.tex_h, C, qtap, qL, qUV);
#pragma unroll
						for (int j = 0; j < CH; j++)
							if (c0 + j < C)
							{
								prev[j] *= Tq;
								prev[j] += (1 - Tq) * edge_channel<PixT, TEX>(eq, qp, texture, qtap, qL, c0 + j, x, y, false, 0.0);
							}
					}
					if (need_replay)
					{
#pragma unroll
						for (int j = 0; j < CH; j++)
							aa[j] = prev[j];
					}
					const double Tr = plane_at(e.x2t, x, y);
					Tap etap;
					double eL = 0, eUV[2] = {0, 0};
					if (e.kind == KIND_TEXTURED && TEX && hit)
						textured_tap(ep, x, y, false, 0.0, p.tex_w, p.tex_h, C, etap, eL, eUV);
					double T_B = 0, L_B = 0, e_B[2] = {0, 0};
#pragma unroll
					for (int j = 0; j < CH; j++)
					{
						const int c = c0 + j;
						if (c >= C)
							continue;
						double A_B = 0;
						if (hit)
						{
							if (e.kind == KIND_TEXTURED && TEX)
							{ // H.h:2006-2021
								const double i00 = ldp(texture, etap.idx[0] + c), i10 = ldp(texture, etap.idx[1] + c);
								const double i01 = ldp(texture, etap.idx[2] + c), i11 = ldp(texture, etap.idx[3] + c);
								const double A = bilinear_mix(etap, i00, i10, i01, i11);
								T_B += g[j] * (prev[j] - A * eL);
								const double a_b = eL * (1 - Tr) * g[j];
								L_B += g[j] * (1 - Tr) * A;
								double wgt[4];
								bilinear_mix_adjoint(etap, a_b, i00, i10, i01, i11, wgt, e_B);
								if (texture_b)
									texture_scatter(texture_b, etap, c, wgt, det_tex, det);
							}
							else
							{ // H.h:1726-1746
								const double A = interp_channel(ep, c, x, y, false, 0.0);
								T_B += g[j] * (prev[j] - A);
								A_B = (1 - Tr) * g[j];
							}
							g[j] *= Tr;
						}
						if (e.kind != KIND_TEXTURED || !TEX)
							add_moments(eacc + 3 * c, A_B, x, y, lane, det);
					}
					if (e.kind == KIND_TEXTURED && TEX)
					{
						add_moments(eacc + 0, (hit && !etap.out[0]) ? e_B[0] : 0.0, x, y, lane, det);
						add_moments(eacc + 3, (hit && !etap.out[1]) ? e_B[1] : 0.0, x, y, lane, det);
						add_moments(eacc + 6, L_B, x, y, lane, det);
					}
					add_moments(eacc + 3 * P, T_B, x, y, lane, det);
				}
			}
			// adjoint of pass 1: what is left of g belongs to the triangle that owns the pixel (H.h:1024-1037, 1320-1353)
			if (kind == KIND_TEXTURED && TEX)
			{
#pragma unroll
				for (int j = 0; j < CH; j++)
				{
					const int c = c0 + j;
					if (c >= C)
						continue;
					const double i00 = ldp(texture, tap.idx[0] + c), i10 = ldp(texture, tap.idx[1] + c);
					const double i01 = ldp(texture, tap.idx[2] + c), i11 = ldp(texture, tap.idx[3] + c);
					const double A = bilinear_mix(tap, i00, i10, i01, i11);
					own_L_B += g[j] * A;
					double wgt[4];
					bilinear_mix_adjoint(tap, g[j] * L, i00, i10, i01, i11, wgt, own_e_B);
					if (texture_b)
						texture_scatter(texture_b, tap, c, wgt, det_tex, det);
				}
			}
			// segmented wave reduction over the distinct interpolated owners of the tile
			unsigned long long rem = __ballot(owner >= 0 && kind == KIND_INTERP);
			while (rem)
			{
				const int l = __ffsll((long long)rem) - 1;
				const int cur = __shfl(owner, l, 64);
				const bool mine = owner == cur;
				rem &= ~__ballot(mine);
				double *acc = w.tri_acc + (size_t)cur * 3 * P;
#pragma unroll
				for (int j = 0; j < CH; j++)
					if (c0 + j < C)
						add_moments(acc + 3 * (c0 + j), mine ? g[j] : 0.0, x, y, lane, det);
			}
		}
	}
	// textured owners: the channel sums are complete, reduce the UV and shade plane adjoints
	unsigned long long rem = __ballot(owner >= 0 && kind == KIND_TEXTURED && TEX);
	while (rem)
	{
		const int l = __ffsll((long long)rem) - 1;
		const int cur = __shfl(owner, l, 64);
		const bool mine = owner == cur && kind == KIND_TEXTURED && TEX;
		rem &= ~__ballot(owner == cur);
		double *acc = w.tri_acc + (size_t)cur * 3 * P;
		add_moments(acc + 0, (mine && !tap.out[0]) ? own_e_B[0] : 0.0, x, y, lane, det);
		add_moments(acc + 3, (mine && !tap.out[1]) ? own_e_B[1] : 0.0, x, y, lane, det);
		add_moments(acc + 6, mine ? own_L_B : 0.0, x, y, lane, det);
	}
}

template <class PixT>
__device__ __noinline__ void bwd_tile_generic(const KParams &p, int view, int tx, int ty, int lane, volatile uint32_t *order)
{
	bwd_tile_generic_impl<PixT, false>(p, view, tx, ty, lane, order);
}

template <class PixT>
__global__ __launch_bounds__(256) void raster_bwd_kernel(KParams p)
{
	__shared__ volatile uint32_t s_order[4][MAX_SORTED];
	const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
	const int strips_x = (p.L.tiles_x + 3) / 4;
	const int b = xcd_band(blockIdx.x, gridDim.x);
	const int ty = xcd_strip_row(b / strips_x, p.L.tiles_y, p.row_group), tx = (b % strips_x) * 4 + wave;
	if (tx < p.L.tiles_x)
		bwd_tile_generic<PixT>(p, blockIdx.y, tx, ty, lane, s_order[wave]);
}

} // namespace
