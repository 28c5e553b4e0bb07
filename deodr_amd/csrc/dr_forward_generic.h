// deodr_amd/csrc/dr_forward_generic.h -- part of the single translation unit dr_kernels.hip (device code, gfx950 / wave64).
// Blending order of a tile's edges, and raster_fwd_kernel: the forward raster without LDS staging (nb_colors > 4, antialiase_error).
#pragma once

#include "dr_setup.h"

using namespace dr;

namespace
{

// ------------------------------------------------------------------------------------------- tile-level edge ordering

// Edges are blended far -> near: descending depth sum of the owning triangle, ties by slot (= 3 * triangle + n), which
// is the order of the reference's loops (H.h:2841-2853) with a stable sort.  `next_edge` returns the first edge of the
// tile strictly after (last_key, last_slot) in that order, scanning the inline list and, if the tile spilled, the pool.
struct EdgeCursor
{
	double key;
	uint32_t slot;
};

__device__ __forceinline__ bool edge_before(double ka, uint32_t sa, double kb, uint32_t sb) { return ka > kb || (ka == kb && sa < sb); }

__device__ __forceinline__ uint32_t next_edge(const ViewPtrs &w, int tile, int nedge, uint32_t spill_n, bool first, EdgeCursor last, bool reverse, int lane,
							  EdgeCursor &found)
{
	// per-lane best candidate
	double bk = 0;
	uint32_t bs = 0xffffffffu;
	auto consider = [&](uint32_t slot) {
		double key = w.edge_rec[slot].key;
		bool after = first || (reverse ? edge_before(key, slot, last.key, last.slot) : edge_before(last.key, last.slot, key, slot));
		if (!after)
			return;
		bool better = bs == 0xffffffffu || (reverse ? edge_before(bk, bs, key, slot) : edge_before(key, slot, bk, bs));
		if (better)
		{
			bk = key;
			bs = slot;
		}
	};
	int n_inline = nedge < K_EDGE ? nedge : K_EDGE;
	if (lane < n_inline)
		consider(w.edge_list[(size_t)tile * K_EDGE + lane]);
	if (nedge > K_EDGE)
		for (uint32_t i = lane; i < spill_n; i += 64)
		{
			uint2 pr = w.edge_pool[i];
			if ((int)pr.x == tile)
				consider(pr.y);
		}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
	{
		double ok = __shfl_xor(bk, o, 64);
		uint32_t os = (uint32_t)__shfl_xor((int)bs, o, 64);
		bool take = os != 0xffffffffu && (bs == 0xffffffffu || (reverse ? edge_before(bk, bs, ok, os) : edge_before(ok, os, bk, bs)));
		if (take)
		{
			bk = ok;
			bs = os;
		}
	}
	found.key = bk;
	found.slot = bs;
	return bs;
}

// per-pixel evaluation of one edge: is the pixel in the sigma band in front of what pass 1 left there?
__device__ __forceinline__ bool edge_touches(const EdgeRec &e, int x, int y, int W, bool persp, double zbest, bool inb)
{
	if (!inb || !edge_covers(e, x, y, W))
		return false;
	double Z = plane_at(e.xZ, (double)x, (double)y);
	if (persp)
		Z = 1 / Z;
	return Z < zbest;
}

template <class PixT, bool TEX = true>
__device__ __forceinline__ double edge_channel(const EdgeRec &e, const double *planes, const PixT *texture, const Tap &tap, double L, int c, double x,
											   double y, bool persp, double Z)
{
	if (TEX && e.kind == KIND_TEXTURED)
		return textured_channel(texture, tap, c) * L;
	return interp_channel(planes, c, x, y, persp, Z);
}

template <class PixT>
__device__ __forceinline__ double background_channel(const KParams &p, int view, size_t pix, int c)
{
	if (p.bg_image)
		return (double)((const PixT *)p.bg_image)[((size_t)view * p.H * p.W + pix) * p.C + c];
	return (double)((const PixT *)p.bg_color)[c];
}

// ------------------------------------------------------------------------------------------------- forward raster

// One thread per view closes the epoch of a forward (nobody else reads `epoch` or `needed_max` during the forward raster): the
// sticky spill high-water mark, and -- in the header of view 0 -- the maximum / union over the views that the host polls with
// ONE 64-byte copy (deodr_hip_workspace_status, HipRasterizer's deferred check).
__device__ __forceinline__ void close_epoch(const KParams &p, const ViewPtrs &w, bool fused)
{
	const uint32_t cur = w.hdr->cur;
	const uint32_t a = w.hdr->tri_spill[cur], bq = w.hdr->edge_spill[cur];
	uint32_t m = a > bq ? a : bq;
	if (m > w.hdr->needed_max)
		w.hdr->needed_max = m;
	else
		m = w.hdr->needed_max;
	w.hdr->owners_partial = fused ? 1u : 0u;
	WsHeader *all = (WsHeader *)(p.ws + p.L.hdr);
	if (m > all->all_needed_max) // monotone: a stale read only costs a redundant atomic
		atomicMax(&all->all_needed_max, m);
	const uint32_t errs = w.hdr->scene_errors;
	if (errs)
		atomicOr(&all->all_scene_errors, errs);
	w.hdr->epoch = w.hdr->epoch + 1;
}

template <class PixT>
__global__ __launch_bounds__(256) void raster_fwd_kernel(KParams p)
{
	__shared__ volatile uint32_t s_order[4][MAX_SORTED];
	const int view = blockIdx.y;
	const int wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63;
	const ViewPtrs w = view_ptrs(p, view);
	const int strips_x = (p.L.tiles_x + 3) / 4;
	const int b = xcd_band(blockIdx.x, gridDim.x);
	const int ty = xcd_strip_row(b / strips_x, p.L.tiles_y, p.row_group), tx = (b % strips_x) * 4 + wave;
	const int W = p.W, H = p.H, C = p.C, P = p.L.P;
	const bool persp = p.persp, strict = p.strict;
	const PixT *texture = (const PixT *)p.texture;

	if (tx < p.L.tiles_x)
	{
		const int tile = ty * p.L.tiles_x + tx;
		const int px = tx * TILE + (lane & 7), py = ty * TILE + (lane >> 3);
		const bool inb = px < W && py < H;
		const size_t pix = (size_t)py * W + px;
		const size_t vpix = (size_t)view * H * W + pix;
		const int ntri = uniform((int)w.tri_cnt[tile]);
		const int nedge = uniform((int)w.edge_cnt[tile]);
		if (lane == 0)
		{ // self-cleaning tile counters; the adjoint finds the edge count in edge_saved
			w.tri_cnt[tile] = 0;
			w.edge_cnt[tile] = 0;
			w.edge_saved[tile] = (uint32_t)nedge;
		}
		// ---- pass 1: visibility.  winner = min (Z, triangle index): identical to the reference's index-order loop
		//      with the strict test Z < z_buffer (H.h:961)
		double zbest = INFINITY;
		int kbest = -1;
		auto try_triangle = [&](int k) {
			const TriRec &r = w.tri_rec[k];
			if (inb && tri_covers(r, px, py, W, H, strict))
			{
				double Z = plane_at(r.xZ, (double)px, (double)py);
				if (persp)
					Z = 1 / Z;
				if (Z < zbest || (Z == zbest && k < kbest))
				{
					zbest = Z;
					kbest = k;
				}
			}
		};
		const int n_inline = ntri < K_TRI ? ntri : K_TRI;
		for (int i = 0; i < n_inline; i++)
			try_triangle(uniform((int)w.tri_list[(size_t)tile * K_TRI + i]));
		if (ntri > K_TRI)
		{ // the tile spilled: pick its pairs out of the pool
			uint32_t spill_n = w.hdr->tri_spill[w.hdr->cur];
			if (spill_n > p.L.tri_pool_cap)
				spill_n = p.L.tri_pool_cap;
			for (uint32_t i0 = 0; i0 < spill_n; i0 += 64)
			{
				uint2 pr = (i0 + lane < spill_n) ? w.tri_pool[i0 + lane] : make_uint2(0xffffffffu, 0u);
				unsigned long long m = __ballot((int)pr.x == tile);
				while (m)
				{
					int l = __ffsll((long long)m) - 1;
					m &= m - 1;
					try_triangle(__shfl((int)pr.y, l, 64));
				}
			}
		}
		// ---- edge order of the tile (shared by all channel chunks)
		uint32_t edge_spill_n = 0;
		if (nedge > K_EDGE)
		{
			edge_spill_n = w.hdr->edge_spill[w.hdr->cur];
			if (edge_spill_n > p.L.edge_pool_cap)
				edge_spill_n = p.L.edge_pool_cap;
		}
		const bool cached = nedge <= MAX_SORTED;
		int n_sorted = nedge; // edges actually retrievable (fewer than nedge only when the spill pool overflowed)
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
					s_order[wave][r] = slot;
				cur = f;
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
		}
		// owner's kind and planes
		int kind = KIND_NONE;
		const double *planes = nullptr;
		Tap tap;
		double L = 0, UV[2];
		if (kbest >= 0)
		{
			kind = w.tri_rec[kbest].kind;
			planes = w.tri_planes + (size_t)kbest * 3 * P;
			if (kind == KIND_TEXTURED)
				textured_tap(planes, (double)px, (double)py, persp, zbest, p.tex_w, p.tex_h, C, tap, L, UV);
		}
		double err_acc = 0;
		for (int c0 = 0; c0 < C; c0 += CH)
		{
			double col[CH];
#pragma unroll
			for (int j = 0; j < CH; j++)
			{
				const int c = c0 + j;
				col[j] = 0;
				if (c < C && inb)
				{
					if (kbest < 0)
						col[j] = background_channel<PixT>(p, view, pix, c);
					else if (kind == KIND_TEXTURED)
						col[j] = textured_channel(texture, tap, c) * L;
					else
						col[j] = interp_channel(planes, c, (double)px, (double)py, persp, zbest);
				}
			}
			if (p.aa_err)
			{ // err_buffer initialisation, H.h:2824-2837 (the image itself stays un-antialiased in this mode)
#pragma unroll
				for (int j = 0; j < CH; j++)
					if (c0 + j < C && inb)
					{
						double d = col[j] - (double)((const PixT *)p.obs)[vpix * C + c0 + j];
						err_acc += d * d;
					}
			}
			else if (nedge > 0)
			{ // ---- pass 2: discontinuity-edge overdraw, far -> near (H.h:1629-1644, 1865-1904)
				EdgeCursor cur = {0, 0};
				for (int r = 0; r < n_sorted; r++)
				{
					uint32_t slot;
					if (cached)
						slot = s_order[wave][r];
					else
					{
						EdgeCursor f;
						slot = next_edge(w, tile, nedge, edge_spill_n, r == 0, cur, false, lane, f);
						cur = f;
					}
					slot = (uint32_t)uniform((int)slot);
					if (slot == 0xffffffffu)
						break;
					const EdgeRec &e = w.edge_rec[slot];
					if (edge_touches(e, px, py, W, persp, zbest, inb))
					{
						const double *ep = w.edge_planes + (size_t)slot * 3 * P;
						double Ze = plane_at(e.xZ, (double)px, (double)py);
						if (persp)
							Ze = 1 / Ze;
						const double Tr = plane_at(e.x2t, (double)px, (double)py);
						Tap etap;
						double eL = 0, eUV[2];
						if (e.kind == KIND_TEXTURED)
							textured_tap(ep, (double)px, (double)py, persp, Ze, p.tex_w, p.tex_h, C, etap, eL, eUV);
#pragma unroll
						for (int j = 0; j < CH; j++)
							if (c0 + j < C)
							{
								double A = edge_channel(e, ep, texture, etap, eL, c0 + j, (double)px, (double)py, persp, Ze);
								col[j] *= Tr;
								col[j] += (1 - Tr) * A;
							}
					}
				}
			}
			if (p.image && inb)
			{
				PixT *out = (PixT *)p.image + vpix * C + c0;
#pragma unroll
				for (int j = 0; j < CH; j++)
					if (c0 + j < C)
						out[j] = (PixT)col[j];
			}
		}
		if (p.aa_err)
		{ // edges antialiase the squared residual instead of the image (H.h:2441-2472, 2154-2193)
			double err = err_acc;
			EdgeCursor cur = {0, 0};
			for (int r = 0; r < n_sorted; r++)
			{
				uint32_t slot;
				if (cached)
					slot = s_order[wave][r];
				else
				{
					EdgeCursor f;
					slot = next_edge(w, tile, nedge, edge_spill_n, r == 0, cur, false, lane, f);
					cur = f;
				}
				slot = (uint32_t)uniform((int)slot);
				if (slot == 0xffffffffu)
					break;
				const EdgeRec &e = w.edge_rec[slot];
				if (edge_touches(e, px, py, W, persp, zbest, inb))
				{
					const double *ep = w.edge_planes + (size_t)slot * 3 * P;
					double Ze = plane_at(e.xZ, (double)px, (double)py);
					if (persp)
						Ze = 1 / Ze;
					const double Tr = plane_at(e.x2t, (double)px, (double)py);
					Tap etap;
					double eL = 0, eUV[2];
					if (e.kind == KIND_TEXTURED)
						textured_tap(ep, (double)px, (double)py, persp, Ze, p.tex_w, p.tex_h, C, etap, eL, eUV);
					double Err = 0;
					for (int c = 0; c < C; c++)
					{
						double d = edge_channel(e, ep, texture, etap, eL, c, (double)px, (double)py, persp, Ze) - (double)((const PixT *)p.obs)[vpix * C + c];
						Err += d * d;
					}
					err *= Tr;
					err += (1 - Tr) * Err;
				}
			}
			if (p.err && inb)
				((PixT *)p.err)[vpix] = (PixT)err;
		}
		if (inb)
		{
			if (p.zbuf)
				((PixT *)p.zbuf)[vpix] = (PixT)zbest;
			w.face_id[pix] = pack_owner(kbest, kind);
		}
	}
	if (blockIdx.x == 0 && threadIdx.x == 0)
		close_epoch(p, w, false);
}

} // namespace
