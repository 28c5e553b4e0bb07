// deodr_amd/csrc/dr_backward.h -- part of the single translation unit dr_kernels.hip (device code, gfx950 / wave64).
// The LDS-staged adjoint: raster_bwd_fast_kernel (pass 1) and raster_bwd_edge_kernel (pass 2, persistent waves over the edge tiles).
#pragma once

#include "dr_backward_generic.h"

using namespace dr;

namespace
{

// ---------------------------------------------------------------------------------- backward raster, LDS-staged fast path
//
// nb_colors <= 4, no antialiase_error, at most K_EDGE edges in the tile (other tiles call bwd_tile_generic).
// Differences from the generic tile: edges are staged / ranked / span-tested exactly as in raster_fwd_fast_kernel, and the
// segmented reductions "sum over the pixels of a primitive" are a head-flag scan over the 8 lanes of every pixel row (DPP, no LDS),
// a table of run totals in LDS, and ONE global atomic per (primitive, moment) issued by 64 lanes in parallel -- instead of a
// 64-lane butterfly per moment and primitive (owner_adjoint).  The texture gradient alone goes through LDS atomics (ds_add_f64).

constexpr int NMOM = 12; // moments per owner slot: 3 per channel (or 9 for a textured owner)

struct alignas(16) BwdLds
{
	RecSlot rec[TB]; // (the layout of WaveLds: edge_reverse_sweep and stage_edge_batch take either)
	double planes[TB * 12];
	uint32_t ids[TB];
	uint8_t cover[TILE][TB];
	uint32_t order[TB];
};

__device__ __forceinline__ void lds_add(double *slot, double v)
{ // ds_add_f64 through the address-space-3 builtin (unsafeAtomicAdd on a generic pointer asks "is it shared?" first)
	if (v != 0)
		__builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double *)slot, v);
}
// the native LDS float add of gfx950 (ds_add_f32, no return): through the address-space-3 builtin, so that the compiler cannot expand it
// into a compare-and-swap loop (what round 4's atomicAdd on a generic float * became)
__device__ __forceinline__ void lds_add_f32(float *slot, float v)
{
	__builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
}

// The texture-gradient window of owner_adjoint holds doubles.  A float32 window for float32 frames (twice the texels in the same 3 KB)
// was built twice -- round 4 through atomicAdd (a compare-and-swap loop), round 5 through the native ds_add_f32 -- and is slower both
// times: BASELINE configs[4], forward raster 0.64 -> 0.97 / 0.99 ms.  LDS float32 adds are slow on this part where float64 adds are
// not (the same showed in the slot table of owner_adjoint_slots: 24 ds_add_f32 per pair of tiles cost 17 us per step).
__device__ __forceinline__ void win_add(double *slot, double v) { lds_add(slot, v); }

constexpr int RUNS = 32; // run totals flushed per pass: 32 x 12 doubles fit in the (by then idle) record staging area of the wave
static_assert(RUNS * NMOM * sizeof(double) <= sizeof(WaveLds::rec) + sizeof(WaveLds::planes) && RUNS * 4 <= sizeof(WaveLds::cover), "LDS reuse");

// Adjoint of pass 1 for one tile: g = dL/d(colour written by pass 1) of this lane's pixel, owned by triangle `owner`.
// tab (RUNS * NMOM doubles) and own (RUNS words) are LDS scratch of this wave.  All 64 lanes must call it.
template <class PixT, bool TEX>
__device__ __forceinline__ void owner_adjoint(const KParams &p, const ViewPtrs &w, int lane, double x, double y, int owner, int kind, const double *g,
											  const Tap &tap, double L, double *tab, uint32_t *own, int win_cap)
{ // win_cap: entries of `tab` the texture-gradient window may use (at least RUNS * NMOM: what the run totals need afterwards)
	const int C = p.C, P = p.L.P;
	const PixT *texture = (const PixT *)p.texture;
	PixT *texture_b = (PixT *)p.texture_b;
	// per-pixel adjoint of the owner's (up to four) attribute planes; its moments  sum v * [x, y, 1]  over the owner's pixels
	// are what the per-triangle finalize needs
	double val[CH] = {0, 0, 0, 0};
	// Texture gradient.  The taps of the 64 pixels of a tile fall into a small window of texels (a magnified texture: a
	// dozen texels for 768 contributions), and atomics to one address serialise in the L2 at ~80 ns each: when the window
	// fits the LDS scratch, the contributions are summed there (LDS adds) and each touched texel leaves with ONE global
	// atomic -- "per-tile LDS partials before a single atomicAdd".
	// Round 5: a tile whose window does not fit -- a MINIFIED texture: towards the limb of a curved surface the texels per pixel grow
	// without bound; on BASELINE configs[4] a fifth of the tiles, and they scattered their 768 taps straight to memory -- is taken as
	// four 4 x 4-pixel quadrants, each with a window of its own (quadrant bounds are the same butterfly stopped two steps early);
	// only a quadrant whose own window does not fit scatters.
	typedef double WinT;
	const int WIN_CAP = win_cap;
	WinT *const win = tab;
	const bool textured = kind == KIND_TEXTURED && TEX;
	int fu = 0, fv = 0;
	int mode = 0; // 0: no window (scatter, or no texture gradient asked for), 1: one window for the tile, 2: one per quadrant
	int full_u0 = 0, full_v0 = 0, full_w = 0, full_h = 0;
	uint32_t q_lo = 0xffffffffu, q_hi = 0; // this lane's quadrant: packed (u, v) bounds of the taps' first texels
	if (texture_b && __ballot(textured))
	{
		if (textured)
		{
			const int t0 = tap.idx[0] / C;
			fv = t0 / p.tex_w;
			fu = t0 - fv * p.tex_w;
		}
		if (p.tex_w <= 0xffff && p.tex_h <= 0xffff)
		{ // (u, v) packed into one word: two exchanges per step (v_pk_min_u16 / v_pk_max_u16)
			typedef unsigned short U2 __attribute__((ext_vector_type(2)));
			U2 lo = textured ? U2{(unsigned short)fu, (unsigned short)fv} : U2{0xffff, 0xffff};
			U2 hi = textured ? U2{(unsigned short)fu, (unsigned short)fv} : U2{0, 0};
			auto step = [&](int d) {
				const int lo_o = __shfl_xor(__builtin_bit_cast(int, lo), d, 64), hi_o = __shfl_xor(__builtin_bit_cast(int, hi), d, 64);
				lo = __builtin_elementwise_min(lo, __builtin_bit_cast(U2, lo_o));
				hi = __builtin_elementwise_max(hi, __builtin_bit_cast(U2, hi_o));
			};
			step(1), step(2), step(8), step(16); // lanes (x & 4, y & 4) alike: a 4 x 4-pixel quadrant
			q_lo = __builtin_bit_cast(uint32_t, lo), q_hi = __builtin_bit_cast(uint32_t, hi);
			step(4), step(32);
			full_u0 = lo.x, full_v0 = lo.y, full_w = hi.x - lo.x + 2, full_h = hi.y - lo.y + 2;
			mode = full_w * full_h * C <= WIN_CAP ? 1 : 2;
		}
		else
		{
			int lo_u = textured ? fu : 0x7fffffff, lo_v = textured ? fv : 0x7fffffff, hi_u = textured ? fu : -1, hi_v = textured ? fv : -1;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1)
			{
				lo_u = min(lo_u, __shfl_xor(lo_u, d, 64));
				lo_v = min(lo_v, __shfl_xor(lo_v, d, 64));
				hi_u = max(hi_u, __shfl_xor(hi_u, d, 64));
				hi_v = max(hi_v, __shfl_xor(hi_v, d, 64));
			}
			full_u0 = lo_u, full_v0 = lo_v, full_w = hi_u - lo_u + 2, full_h = hi_v - lo_v + 2;
			mode = full_w * full_h * C <= WIN_CAP ? 1 : 0;
		}
	}
	auto window_zero = [&](int ww, int wh) {
		lds_sync();
		for (int i = lane; i < ww * wh * C; i += 64)
			win[i] = 0;
		lds_sync();
	};
	// The window leaves row by row: a row of it is ww * C CONSECUTIVE elements of texture_b (x fastest, channels inside), so a lane
	// needs no division to find its texel and the atomics of one instruction fall into one or two cache lines.  Several rows per pass
	// when a row is shorter than half a wavefront (lane / row length once, by a float reciprocal: exact for these sizes).
	auto window_flush = [&](int u0, int v0, int ww, int wh) {
		lds_sync();
		const int rowlen = ww * C;
		const int per_pass = rowlen < 64 ? 64 / rowlen : 1; // rows per pass (uniform)
		const int lr = rowlen < 64 ? (int)(((float)lane + 0.5f) * (1.0f / (float)rowlen)) : 0, li = lane - lr * rowlen;
		for (int jv0 = 0; jv0 < wh; jv0 += per_pass)
		{
			const int jv = jv0 + lr;
			if (lr < per_pass && jv < wh)
				for (int i = li; i < rowlen; i += 64)
				{
					const WinT v = win[jv * rowlen + i];
					if (v != 0)
						unsafeAtomicAdd(texture_b + (size_t)C * (u0 + (size_t)p.tex_w * (v0 + jv)) + i, (PixT)v);
				}
		}
		lds_sync(); // the table is reused
	};
	// the texels of this lane's four taps, channel by channel: the adjoint of the bilinear mix (H.h:1320-1353) and the texture gradient's
	// four contributions per channel -- added to the window whose first texel is (u0, v0) (`add`), or scattered (`scatter`)
	double L_B = 0, e_B[2] = {0, 0};
	auto channels = [&](bool add, bool scatter, int u0, int v0, int ww, bool adjoint) {
		const int wbase = ((fv - v0) * ww + (fu - u0)) * C;
		PixT tx[4][4];
		tap_texels(texture, tap, C, tx);
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
			if (cc < C)
			{
				const double i00 = (double)tx[0][cc], i10 = (double)tx[1][cc];
				const double i01 = (double)tx[2][cc], i11 = (double)tx[3][cc];
				if (adjoint)
					L_B += g[cc] * bilinear_mix(tap, i00, i10, i01, i11);
				double wgt[4], e_tmp[2] = {0, 0};
				bilinear_mix_adjoint(tap, g[cc] * L, i00, i10, i01, i11, wgt, adjoint ? e_B : e_tmp);
				if (add)
				{
					win_add(&win[wbase + cc], wgt[0]);
					win_add(&win[wbase + C + cc], wgt[1]);
					win_add(&win[wbase + ww * C + cc], wgt[2]);
					win_add(&win[wbase + ww * C + C + cc], wgt[3]);
				}
				else if (scatter)
					texture_scatter(texture_b, tap, cc, wgt);
			}
	};
	if (mode == 1)
		window_zero(full_w, full_h);
	if (textured)
		channels(mode == 1, mode == 0 && texture_b != nullptr, full_u0, full_v0, full_w, true);
	if (mode == 1)
		window_flush(full_u0, full_v0, full_w, full_h);
	if (mode == 2)
	{
		const int my_q = ((lane >> 2) & 1) | ((lane >> 4) & 2); // (x & 4) | (y & 4): lane = y * 8 + x
#pragma unroll 1
		for (int qd = 0; qd < 4; qd++)
		{
			const int rep = (qd & 1) * 4 + (qd >> 1) * 32; // a lane of the quadrant
			const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)q_lo, rep), hi = (uint32_t)__builtin_amdgcn_readlane((int)q_hi, rep);
			if (lo == 0xffffffffu)
				continue; // no textured pixel in this quadrant
			const int u0 = (int)(lo & 0xffffu), v0 = (int)(lo >> 16), ww = (int)(hi & 0xffffu) - u0 + 2, wh = (int)(hi >> 16) - v0 + 2;
			const bool fits = ww * wh * C <= WIN_CAP;
			const bool mine = textured && my_q == qd;
			if (fits)
				window_zero(ww, wh);
			if (mine)
				channels(fits, !fits, u0, v0, ww, false);
			if (fits)
				window_flush(u0, v0, ww, wh);
		}
	}
	if (textured)
	{
		val[0] = tap.out[0] ? 0.0 : e_B[0];
		val[1] = tap.out[1] ? 0.0 : e_B[1];
		val[2] = L_B;
	}
	if (kind == KIND_INTERP)
	{ // H.h:1024-1037
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
			if (cc < C)
				val[cc] = g[cc];
	}
	// a run lies in one pixel row, so its y moment is y times its plain sum: two scanned values per plane, not three
	constexpr int NSCAN = 2 * CH;
	double sc[NSCAN];
#pragma unroll
	for (int q = 0; q < CH; q++)
	{
		sc[2 * q] = val[q] * x;
		sc[2 * q + 1] = val[q];
	}
	// Segmented reduction over the pixels of each owner.  Inside a pixel row a triangle's pixels are runs of consecutive
	// lanes, so: head-flag segmented inclusive scan over the 8 lanes of every row (3 DPP steps on the VALU, no LDS),
	// then the last lane of each run adds the run total to the owner's accumulator (one global atomic per moment and run).
	const int nm = 3 * P; // moments per owner in the global accumulator (P = max(C, 3) planes)
	const int lx = lane & 7;
	const int oid = (owner >= 0 && kind != KIND_NONE) ? owner : -1;
	const int left_oid = dpp_i<0x111>(oid); // evaluated by ALL lanes: a DPP move under a divergent branch reads 0 from disabled lanes
	const bool head = (lx == 0) | (left_oid != oid);
	int f = head ? 1 : 0;
#define DR_SEG_STEP(CTRL)                                                                                                    \
	{                                                                                                                        \
		const int tf = dpp_i<CTRL>(f);                                                                                       \
		double t[NSCAN];                                                                                                     \
		_Pragma("unroll") for (int i = 0; i < NSCAN; i++) t[i] = dpp_d<CTRL>(sc[i]);                                         \
		/* the DPP moves above run with every lane enabled (a disabled source lane reads as 0); only the adds are masked */   \
		if (!f)                                                                                                              \
		{                                                                                                                    \
			_Pragma("unroll") for (int i = 0; i < NSCAN; i++) sc[i] += t[i];                                                 \
		}                                                                                                                    \
		f = f ? f : tf;                                                                                                      \
	}
	DR_SEG_STEP(0x111)
	DR_SEG_STEP(0x112)
	DR_SEG_STEP(0x114)
#undef DR_SEG_STEP
	const int right_head = dpp_i<0x101>(head ? 1 : 0);
	const bool tail = (lx == 7) | (right_head != 0);
	// Run totals go through LDS so that the global atomics are issued moment-major by 64 lanes at once: the cost of an atomic
	// instruction is per distinct cache line it touches, and the 3P moments of one owner are contiguous.
	const bool emit = tail && oid >= 0;
	unsigned long long emask = __ballot(emit);
	while (emask)
	{
		const int my_run = __popcll(emask & ((1ull << lane) - 1ull));
		const bool sel = ((emask >> lane) & 1ull) && my_run < RUNS;
		const int total = __popcll(emask);
		const int nrun = total < RUNS ? total : RUNS;
		lds_sync();
		if (sel)
		{
			own[my_run] = (uint32_t)oid;
#pragma unroll
			for (int q = 0; q < CH; q++)
			{
				tab[my_run * NMOM + 3 * q] = sc[2 * q];
				tab[my_run * NMOM + 3 * q + 1] = sc[2 * q + 1] * y;
				tab[my_run * NMOM + 3 * q + 2] = sc[2 * q + 1];
			}
		}
		lds_sync();
		// Runs of the same owner (one per pixel row it crosses) are merged before they leave the tile: lane 12 j + m sums moment m
		// over the runs of the j-th distinct owner, five owners per atomic instruction.
		const uint32_t own_l = lane < nrun ? own[lane] : 0xffffffffu;
		uint32_t rem = (uint32_t)__ballot(lane < nrun);
		while (rem)
		{
			constexpr int G = 5;
			uint32_t gid[G], gmask[G];
#pragma unroll
			for (int j = 0; j < G; j++)
			{
				gid[j] = 0;
				gmask[j] = 0;
				if (rem)
				{
					const int lead = __ffs((int)rem) - 1;
					gid[j] = (uint32_t)__builtin_amdgcn_readlane((int)own_l, lead);
					gmask[j] = (uint32_t)__ballot(own_l == gid[j]) & rem;
					rem &= ~gmask[j];
				}
			}
			const int j = lane / NMOM, m = lane - j * NMOM;
			uint32_t o = 0, mask = 0;
#pragma unroll
			for (int q = 0; q < G; q++)
			{
				o = j == q ? gid[q] : o;
				mask = j == q ? gmask[q] : mask;
			}
			double acc = 0;
			while (mask)
			{
				const int r = __ffs((int)mask) - 1;
				mask &= mask - 1;
				acc += tab[r * NMOM + m];
			}
			if (m < nm && acc != 0)
				atomic_add_f64(w.tri_acc + (size_t)o * nm + m, acc);
		}
		emask &= ~__ballot(sel);
	}
}

// ---- round 5: the adjoint of pass 1 for the float32 instances of an UNTEXTURED fit step, keyed by the owner's SLOT in the staged batch
//
// owner_adjoint above knows an owner by its triangle index, so the run totals of a tile have to be grouped by owner before they can leave
// (readlane / ballot rounds over the distinct owners, a loop over the runs of each), and it scans in double: two v_mov_dpp + one
// v_add_f64 + the masking per value and step.  In a tile whose triangles all sit in ONE staged batch (every paired tile, nine single
// tiles in ten) the owner of a pixel is a slot number below 16, which IS a table index:
//   * per pixel and channel the two sums  g  and  g (x - x0)  -- coordinates relative to the tile, so that float32 carries them: the
//     absolute moments are formed in double at the very end,  sum g x = x0 sum g + sum g (x - x0)  -- are scanned over the 8 lanes of a
//     pixel row with ONE v_fmac_f32 row_shr per value and step (the segment mask is a float factor 0 / 1 that is scanned along);
//   * the tail lane of every run STORES its 2 C run totals into entry [slot][row] of a float table in LDS (two 16-byte stores): a convex
//     triangle meets a pixel row in one span, so an entry has one writer -- unless another triangle cuts the span in two (occlusion): a
//     run whose owner already ended a run further left in the row is told apart by an OR-scan of the tails' slot bits and ADDS
//     (ds_add_f32, the native instruction) after the stores.  (The table as [slot][moment] with every run added atomically was built
//     first: 24 ds_add_f32 per pair of tiles cost the forward raster 9 us per step -- LDS float atomics are slow, plain stores are not.)
//   * lane 4 j + c then sums the eight rows of slot j, channel c -- no grouping, no loop over runs --, makes the three moments absolute in
//     double and issues the global atomics; the 3 P moments of an owner are contiguous, as before.
// A pair of tiles (two pixels per lane) goes through ONE table, ONE zeroing and ONE flush.  Sums in float32: the contributions are
// residuals of float32 pixels, a tile adds at most 64 of them per moment, and what leaves the tile is accumulated in double as before
// (gradients within 1e-6 of the all-double path; the float64-pixel instances keep owner_adjoint).
constexpr int SLOT_ROW = 2 * CH;				  // floats of an entry [slot][row]: (sum g (x - x0), sum g) per channel
constexpr int SLOT_STRIDE = TILE * SLOT_ROW + 8; // floats per slot; + 8: the flush lanes of slots j, j + 1, ... start eight banks apart
constexpr int SLOT_MOM = 12;					  // floats per slot of the moment table behind it: 3 per channel
static_assert(TB * (SLOT_STRIDE + SLOT_MOM) * sizeof(float) <= sizeof(WaveLds) + sizeof(EdgeSort), "the slot table lives in the (by then idle) LDS of the wavefront");
static_assert(CH == 4 && TILE == 8 && TB * CH == 64, "one flush lane per (slot, channel)");

// One step of the segmented scan over 2 CH floats: v[i] += keep * v[i] of the lane `SHR` to the left, keep *= keep of that lane -- one
// v_fmac_f32 with the DPP modifier per value (the compiler emits v_mov_b32_dpp + v_fmac_f32: its DPP combiner does not fold into an
// instruction whose destination is also a source).  Hand-written, so the hazards are ours: a DPP read of a register needs two wait
// states behind the VALU write of it -- s_nop 1 covers whatever the compiler placed in front; inside the block every register was
// last written at least eight instructions earlier.  Lanes without a source lane in their 16-lane row read 0 (bound_ctrl).
#define DR_DPP_F32_STEP(SHR)                                                                                                              \
	asm("s_nop 1\n\t"                                                                                                                     \
		"v_fmac_f32_dpp %0, %0, %8 " SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                                    \
		"v_fmac_f32_dpp %1, %1, %8 " SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                                    \
		"v_fmac_f32_dpp %2, %2, %8 " SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                                    \
		"v_fmac_f32_dpp %3, %3, %8 " SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                                    \
		"v_fmac_f32_dpp %4, %4, %8 " SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                                    \
		"v_fmac_f32_dpp %5, %5, %8 " SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                                    \
		"v_fmac_f32_dpp %6, %6, %8 " SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                                    \
		"v_fmac_f32_dpp %7, %7, %8 " SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                                    \
		"v_mul_f32_dpp %8, %8, %8 " SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1"                                                          \
		: "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(keep))

// slot[k] / g[k][]: owner slot (-1: none) and dL/d(colour) of pixel k of this lane (pixel k lies 8 k columns right of lane & 7);
// id_of_slot: lane j holds the triangle index of slot j; tab: TB * SLOT_STRIDE floats of LDS scratch.  All 64 lanes must call it.
template <int NPIX>
__device__ __forceinline__ void owner_adjoint_slots(const KParams &p, const ViewPtrs &w, int lane, int x0, int y0, const int (&slot)[NPIX],
													const float (&g)[NPIX][CH], uint32_t id_of_slot, int nslots, float *tab)
{
	typedef float F4 __attribute__((ext_vector_type(4)));
	const int C = p.C, nm = 3 * p.L.P;
	const int lx = lane & 7, ly = lane >> 3;
	lds_sync();
	for (int i = lane * 4; i < nslots * SLOT_STRIDE; i += 256)
		*(F4 *)(tab + i) = F4{0.0f, 0.0f, 0.0f, 0.0f};
	lds_sync();
	// which lanes may look 1 / 2 / 4 lanes to their left without leaving their pixel row (two pixel rows share a 16-lane DPP row)
	const uint32_t in1 = lx >= 1 ? 0xffffffffu : 0u, in2 = lx >= 2 ? 0xffffffffu : 0u, in4 = lx >= 4 ? 0xffffffffu : 0u;
#pragma unroll
	for (int k = 0; k < NPIX; k++)
	{
		const int oid = slot[k];
		const int left = dpp_i<0x111>(oid); // (all lanes: a DPP move reads 0 from a disabled lane)
		const bool head = (lx == 0) | (left != oid);
		float keep = head ? 0.0f : 1.0f; // 1 while the segment of this lane reaches further left than the scan has looked
		const float xr = (float)(lx + 8 * k);
		float v[2 * CH];
#pragma unroll
		for (int q = 0; q < CH; q++)
		{
			v[2 * q] = q < C ? g[k][q] * xr : 0.0f;
			v[2 * q + 1] = q < C ? g[k][q] : 0.0f;
		}
		DR_DPP_F32_STEP("row_shr:1");
		DR_DPP_F32_STEP("row_shr:2");
		DR_DPP_F32_STEP("row_shr:4");
		const int right_head = dpp_i<0x101>(head ? 1 : 0);
		const bool tail = ((lx == 7) | (right_head != 0)) && oid >= 0;
		// A span cut in two needs three runs in one pixel row (owner, occluder, owner): whether ANY row of the tile has three tails is
		// a few scalar operations on the ballot (per-byte population counts); only then the vector unit looks which runs they are --
		// slots whose run ended further left in the row: an exclusive OR-scan of the tails' slot bits
		bool again = false;
		{
			unsigned long long t = __ballot(tail);
			t = t - ((t >> 1) & 0x5555555555555555ull);
			t = (t & 0x3333333333333333ull) + ((t >> 2) & 0x3333333333333333ull);
			t = (t + (t >> 4)) & 0x0f0f0f0f0f0f0f0full; // tails per pixel row, one byte each
			if ((t + 0x0505050505050505ull) & 0x0808080808080808ull)
			{
				const uint32_t bit = tail ? 1u << oid : 0u;
				uint32_t seen = (uint32_t)dpp_i<0x111>((int)bit) & in1;
				seen |= (uint32_t)dpp_i<0x111>((int)seen) & in1; // (bits of the lanes lx - 1, lx - 2; then - 3, - 4; then - 5 .. - 8)
				seen |= (uint32_t)dpp_i<0x112>((int)seen) & in2;
				seen |= (uint32_t)dpp_i<0x114>((int)seen) & in4;
				again = tail && ((seen >> oid) & 1u);
			}
		}
		float *entry = tab + oid * SLOT_STRIDE + ly * SLOT_ROW;
		if (tail && !again)
		{
			*(F4 *)entry = F4{v[0], v[1], v[2], v[3]};
			*(F4 *)(entry + 4) = F4{v[4], v[5], v[6], v[7]};
		}
		if (__ballot(again))
		{ // a span cut in two by a nearer triangle: the later piece adds to the entry the first one has stored
			if (again)
			{
#pragma unroll
				for (int i = 0; i < 2 * CH; i++)
					if (i < 2 * C)
						lds_add_f32(entry + i, v[i]);
			}
		}
	}
	lds_sync();
	// lane 4 j + c: the three moments of channel c of slot j
	{
		const int j = lane >> 2, c = lane & 3;
		const bool live = j < nslots && c < C;
		const float *col = tab + (live ? j : 0) * SLOT_STRIDE + 2 * c;
		float T = 0.0f, S = 0.0f, Y = 0.0f;
#pragma unroll
		for (int r = 0; r < TILE; r++)
		{
			const float2 ts = *(const float2 *)(col + r * SLOT_ROW);
			T += ts.x;
			S += ts.y;
			Y = fmaf(ts.y, (float)r, Y);
		}
		// The atomics leave MOMENT-major -- a memory-side atomic instruction is paid per cache line it touches, and the 3 P moments of an
		// owner are 96 contiguous bytes: one instruction per 64 entries, not three per lane (measured: three instructions of 4 x 8 bytes
		// per owner took the forward raster from 69 to 82 us per step, profiles/r05g_*) -- so the sums go through LDS once more:
		// [slot][12] floats behind the table's last slot row (the table itself is being read by the other lanes).
		float *const mom = tab + TB * SLOT_STRIDE;
		lds_sync();
		if (live)
		{
			mom[j * SLOT_MOM + 3 * c] = T;
			mom[j * SLOT_MOM + 3 * c + 1] = Y;
			mom[j * SLOT_MOM + 3 * c + 2] = S;
		}
		else if (j < nslots)
			mom[j * SLOT_MOM + 3 * c] = mom[j * SLOT_MOM + 3 * c + 1] = mom[j * SLOT_MOM + 3 * c + 2] = 0.0f;
	}
	lds_sync();
	const float *const mom = tab + TB * SLOT_STRIDE;
	const int entries = nslots * SLOT_MOM;
	for (int i0 = 0; i0 < entries; i0 += 64)
	{
		const int i = i0 + lane, j = i / SLOT_MOM, m = i - j * SLOT_MOM, k = m % 3;
		const bool live = i < entries;
		const float val = live ? mom[i] : 0.0f, sum = live ? mom[i + 2 - k] : 0.0f;
		const uint32_t id = (uint32_t)__shfl((int)id_of_slot, live ? j : 0, 64); // (by every lane: not under the branch below)
		if (live && m < nm && (val != 0.0f || sum != 0.0f))
		{
			const double origin = k == 0 ? (double)x0 : (k == 1 ? (double)y0 : 0.0);
			atomic_add_f64(w.tri_acc + (size_t)id * nm + m, (double)val + origin * (double)sum);
		}
	}
}
#undef DR_DPP_F32_STEP

// Adjoint of pass 2 for the batches b_hi .. b_lo of a tile's blending order (near -> far, H.h:2961-3052), shared by the edge-tile
// kernel of the two-call path and by the fused forward of a fit step.  On entry: es.sorted = the blending order, tm[b] = mask of the
// edges of batch b drawn over this lane's pixel, cur = the pixel's colour after batch b_hi, g = dL/d(that colour); on exit g is the
// gradient that reaches the colour before batch b_lo, cur that colour.  top_staged: the records of batch b_hi are still in S.
// pixel_base(base) fills the un-antialiased colour of the pixel when a replay needs it (have_base says whether it already did).
template <class PixT, bool TEX, int NBATCH, class Lds, class BaseFn>
__device__ __forceinline__ void edge_reverse_sweep(const KParams &p, const ViewPtrs &w, Lds &S, const EdgeSort *es, int lane, double x, double y, int n_edges,
												   int b_hi, int b_lo, bool top_staged, const uint32_t (&tm)[NBATCH], double (&cur)[CH], double (&g)[CH],
												   double (&base)[CH], bool &have_base, BaseFn pixel_base, int r_lo, int r_hi)
{ // r_lo .. r_hi: the edges of each batch that are swept (a split tile of the fused forward: one part of one batch; everybody else: all)
	const int C = p.C, P = p.L.P;
	const PixT *texture = (const PixT *)p.texture;
	PixT *texture_b = (PixT *)p.texture_b;
	const RecSlot *erec = &S.rec[0]; // (the staging area holds EdgeRec and TriRec alike)
	// pass B, near -> far (H.h:2961-3052)
	for (int b = b_hi; b >= b_lo; b--)
	{
		const int first = b * TB, nb = n_edges - first < TB ? n_edges - first : TB;
		if (b < b_hi || !top_staged) // (the records of the batch the caller swept last may still be in LDS)
		{
			lds_sync();
			if (lane < nb)
				S.ids[lane] = es->sorted[first + lane];
			lds_sync();
			stage_batch(*(WaveLds *)&S, w.edge_rec, w.edge_planes, P, nb, lane);
			lds_sync();
		}
		uint32_t tmb = 0;
#pragma unroll
		for (int bb = 0; bb < NBATCH; bb++)
			tmb = bb == b ? tm[bb] : tmb;
		for (int r = (nb - 1 < r_hi ? nb - 1 : r_hi); r >= r_lo; r--)
		{
			const bool hit = (tmb >> r) & 1u;
			if (__ballot(hit) == 0)
				continue;
			const EdgeRec &e = erec[r].edge();
			const double *ep = &S.planes[r * 12];
			// colour before this edge: un-blend like the reference (H.h:1738) when T is safely away from 0, otherwise
			// replay every earlier edge from the un-antialiased colour (the reference yields inf / NaN there)
			double prev[CH];
			const double Tr_here = hit ? plane_at(e.x2t, x, y) : 1.0;
			const bool need_replay = hit && !(Tr_here > 1e-6);
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				prev[cc] = base[cc];
			if (hit && !need_replay)
			{
				Tap utap;
				double uL = 0, uUV[2];
				if (e.kind == KIND_TEXTURED && TEX)
					textured_tap(ep, x, y, false, 0.0, p.tex_w, p.tex_h, C, utap, uL, uUV);
				const double inv_T = 1 / Tr_here; // one division for the C channels (the reference divides each: 1 ulp apart)
#pragma unroll
				for (int cc = 0; cc < CH; cc++)
					if (cc < C)
					{
						prev[cc] = (cur[cc] - (1 - Tr_here) * edge_channel<PixT, TEX>(e, ep, texture, utap, uL, cc, x, y, false, 0.0)) * inv_T;
						cur[cc] = prev[cc];
					}
			}
			if (__ballot(need_replay))
			{ // measure-zero event (pixel centre within 1e-6 sigma of the edge line): records straight from memory
				if (!have_base)
				{
					pixel_base();
					have_base = true;
				}
#pragma unroll
				for (int cc = 0; cc < CH; cc++)
					prev[cc] = need_replay ? base[cc] : prev[cc];
				const int upto = first + r;
				for (int q = 0; q < upto; q++)
				{
					uint32_t tq = 0;
#pragma unroll
					for (int bb = 0; bb < NBATCH; bb++)
						tq = bb == (q / TB) ? tm[bb] : tq;
					if (!need_replay || !((tq >> (q % TB)) & 1u))
						continue;
					const uint32_t sq = es->sorted[q];
					const EdgeRec &eq = w.edge_rec[sq];
					const double *qp = w.edge_planes + (size_t)sq * 3 * P;
					const double Tq = plane_at(eq.x2t, x, y);
					Tap qtap;
					double qL = 0, qUV[2];
					if (eq.kind == KIND_TEXTURED && TEX)
						textured_tap(qp, x, y, false, 0.0, p.tex_w, p.tex_h, C, qtap, qL, qUV);
#pragma unroll
					for (int cc = 0; cc < CH; cc++)
						if (cc < C)
						{
							prev[cc] *= Tq;
							prev[cc] += (1 - Tq) * edge_channel<PixT, TEX>(eq, qp, texture, qtap, qL, cc, x, y, false, 0.0);
						}
				}
				if (need_replay)
				{
#pragma unroll
					for (int cc = 0; cc < CH; cc++)
						cur[cc] = prev[cc];
				}
			}
			// per-pixel plane adjoints of this edge (0 where it does not touch the pixel) ...
			double pb[5] = {0, 0, 0, 0, 0}; // planes 0..3 (colours, or u, v, shade) and the transparency plane
			if (hit)
			{
				const double Tr = Tr_here;
				double T_B = 0;
				if (e.kind == KIND_TEXTURED && TEX)
				{ // H.h:2006-2021
					Tap etap;
					double eL, eUV[2], L_B = 0, e_B[2] = {0, 0};
					textured_tap(ep, x, y, false, 0.0, p.tex_w, p.tex_h, C, etap, eL, eUV);
#pragma unroll
					for (int cc = 0; cc < CH; cc++)
						if (cc < C)
						{
							const double i00 = ldp(texture, etap.idx[0] + cc), i10 = ldp(texture, etap.idx[1] + cc);
							const double i01 = ldp(texture, etap.idx[2] + cc), i11 = ldp(texture, etap.idx[3] + cc);
							const double A = bilinear_mix(etap, i00, i10, i01, i11);
							T_B += g[cc] * (prev[cc] - A * eL);
							L_B += g[cc] * (1 - Tr) * A;
							double wgt[4];
							bilinear_mix_adjoint(etap, eL * (1 - Tr) * g[cc], i00, i10, i01, i11, wgt, e_B);
							if (texture_b)
								texture_scatter(texture_b, etap, cc, wgt);
							g[cc] *= Tr;
						}
					pb[0] = etap.out[0] ? 0.0 : e_B[0];
					pb[1] = etap.out[1] ? 0.0 : e_B[1];
					pb[2] = L_B;
				}
				else
				{ // H.h:1726-1746
#pragma unroll
					for (int cc = 0; cc < CH; cc++)
						if (cc < C)
						{
							const double A = interp_channel(ep, cc, x, y, false, 0.0);
							T_B += g[cc] * (prev[cc] - A);
							pb[cc] = (1 - Tr) * g[cc];
							g[cc] *= Tr;
						}
				}
				pb[4] = T_B;
			}
			// ... reduced over the tile on the VALU (DPP), then ONE atomic instruction (15 lanes) per edge and tile
			double *eacc = w.edge_acc + (size_t)S.ids[r] * (3 * P + 3);
			double mv[16]; // lane 3 * pl + m ends up with moment m of plane pl
#pragma unroll
			for (int pl = 0; pl < 5; pl++)
			{
				mv[3 * pl] = pb[pl] * x;
				mv[3 * pl + 1] = pb[pl] * y;
				mv[3 * pl + 2] = pb[pl];
			}
			mv[15] = 0;
			const double esum = wave_sum16(mv, lane);
			if (lane < 15 && esum != 0 && (lane >= 12 || lane < 3 * P))
			{
				const int pl = lane / 3, m = lane - 3 * pl;
				atomic_add_f64(eacc + (pl == 4 ? 3 * P : 3 * pl) + m, esum);
			}
		}
	}
}

// One tile of the adjoint.  EDGES = false: tiles without silhouette edges (the edge code is compiled out: half the
// registers, twice the resident waves to hide the memory latency); EDGES = true: the tiles that have some.
template <class PixT, bool EDGES, bool TEX>
__device__ __forceinline__ void bwd_fast_tile(const KParams &p, const ViewPtrs &w, int view, int tx, int ty, int lane, BwdLds &S, EdgeSort *es, // es: only for EDGES
											  int chunk = -1)
{ // chunk >= 0: this wavefront is one of CHUNKS that may share the reverse sweep of a many-edged tile (batch `chunk` of it)
	const int W = p.W, H = p.H, C = p.C, P = p.L.P;
	const PixT *texture = (const PixT *)p.texture;
	PixT *texture_b = (PixT *)p.texture_b;
	const int tile = ty * p.L.tiles_x + tx;
	const int x0 = tx * TILE, y0 = ty * TILE;
	const int px = x0 + (lane & 7), py = y0 + (lane >> 3);
	const bool inb = px < W && py < H;
	const size_t pix = (size_t)py * W + px;
	const size_t vpix = (size_t)view * H * W + pix;
	const double x = px, y = py;
	// the owner ids are requested together with the tile's edge count (one memory round trip instead of two)
	const int32_t raw_owner = inb ? w.face_id[pix] : -1;
	const uint32_t raw_nedge = (uint32_t)uniform((int)w.edge_saved[tile]);
	const uint32_t sweep_slot = EDGES ? (uint32_t)uniform((int)w.edge_slot[tile]) : 0u;
	const int nedge = (int)(raw_nedge & ~SWEEP_SAVED);
	const bool sweep_saved = EDGES && (raw_nedge & SWEEP_SAVED) && sweep_slot;
	if ((nedge > 0) != EDGES)
		return; // the other kernel's tile
	// batches of the reverse sweep this wavefront runs: all of them, or -- when the forward saved the colour after every batch
	// -- only batch `chunk`
	const int nbatch_all = (nedge + TB - 1) / TB;
	uint32_t snap = 0;
	if (EDGES && sweep_saved && chunk >= 0 && nbatch_all > 1)
		snap = (uint32_t)uniform((int)*(const uint32_t *)(w.edge_sweep + (size_t)(sweep_slot - 1) * SWEEP_BYTES + SWEEP_SNAP));
	const bool chunked = snap != 0;
	if (EDGES && (chunked ? chunk >= nbatch_all : chunk > 0))
		return; // nothing for this wavefront: the tile has fewer batches, or its sweep is not shared
	const int b_hi = chunked ? chunk : nbatch_all - 1, b_lo = chunked ? chunk : 0;
	int n_edges = 0;
	if (EDGES && sweep_saved)
	{ // the forward saved the blending order with its sweep
		const uint32_t *order = (const uint32_t *)(w.edge_sweep + (size_t)(sweep_slot - 1) * SWEEP_BYTES + SWEEP_ORDER);
		lds_sync();
		for (int i = lane; i < nedge; i += 64)
			es->sorted[i] = order[i];
		lds_sync();
		n_edges = nedge;
	}
	else if (EDGES)
		n_edges = gather_sorted_edges(*es, w, p, tile, nedge, lane);
	if (EDGES && n_edges < 0)
	{ // more than EMAX edges in one tile (or pool overflow): the un-staged code, right here (pathological and slow, but no
	  // queue and no extra launch for the tiles that never exist in a real scene)
		lds_sync();
		bwd_tile_generic_impl<PixT, true, TEX>(p, view, tx, ty, lane, (volatile uint32_t *)es->sorted);
		lds_sync();
		return;
	}
	int owner = -1, kind = KIND_NONE;
	unpack_owner(raw_owner, owner, kind);
	if (__ballot(owner >= 0) == 0 && nedge == 0)
		return;

	double g[CH];
	{
		if (p.aa_err)
		{ // antialiase_error, a tile without silhouette edges: image_b = -2 (obs - image) err_buffer_b (H.h:3054-3060; `image` is the un-antialiased frame of this mode)
			const PixT *im = (const PixT *)p.image_in + vpix * C, *ob = (const PixT *)p.obs + vpix * C;
			const double eb = inb ? (double)((const PixT *)p.err_b)[vpix] : 0.0;
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				g[cc] = (cc < C && inb) ? -2 * ((double)ob[cc] - (double)im[cc]) * eb : 0.0;
		}
		else if (p.image_b)
		{
			const PixT *gin = (const PixT *)p.image_b + vpix * C;
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				g[cc] = (cc < C && inb) ? (double)gin[cc] : 0.0;
		}
		else
		{ // residual mode: dL/dimage of L = sum (image - obs)^2 formed on the fly from the rendered image and the observation
			const PixT *im = (const PixT *)p.image_in + vpix * C, *ob = (const PixT *)p.obs + vpix * C;
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				g[cc] = (cc < C && inb) ? fit_residual<true>(p, (double)im[cc], (double)ob[cc]) : 0.0;
		}
	}
	// what pass 1 left at this pixel
	const double *planes = nullptr;
	double zown = INFINITY;
	Tap tap;
	double L = 0, UV[2] = {0, 0};
	if (owner >= 0)
	{
		planes = w.tri_planes + (size_t)owner * 3 * P;
		if (kind == KIND_TEXTURED && TEX)
			textured_tap(planes, x, y, false, 0.0, p.tex_w, p.tex_h, C, tap, L, UV);
	}

	// ---- adjoint of pass 2 (near -> far), TB staged edges at a time
	if (EDGES && n_edges > 0)
	{
		// depth and un-antialiased colour of the pixel (only needed by the forward sweep and by the replay fallback)
		double base[CH] = {0, 0, 0, 0};
		auto pixel_base = [&]() {
			if (owner >= 0)
			{
				zown = plane_at(w.tri_rec[owner].xZ, x, y);
#pragma unroll
				for (int cc = 0; cc < CH; cc++)
					if (cc < C)
						base[cc] = kind == KIND_TEXTURED && TEX ? textured_channel(texture, tap, cc) * L : interp_channel(planes, cc, x, y, false, 0.0);
			}
			else if (inb)
			{
#pragma unroll
				for (int cc = 0; cc < CH; cc++)
					if (cc < C)
						base[cc] = background_channel<PixT>(p, view, pix, cc);
			}
		};
		// pass A, far -> near: which edges are drawn over this pixel (bit j of tm[b] = edge 16 b + j in blending order)
		// and the antialiased colour they leave -- read back when the forward raster saved its own sweep of this tile
		uint32_t tm[EMAX / TB] = {0, 0, 0, 0, 0, 0, 0, 0};
		static_assert(EMAX / TB == 8, "tm[] initialiser");
		double cur[CH] = {0, 0, 0, 0};
		const int nbatch = (n_edges + TB - 1) / TB;
		bool have_base = !sweep_saved;
		if (sweep_saved)
		{
			const char *slot = w.edge_sweep + (size_t)(sweep_slot - 1) * SWEEP_BYTES;
			// the colour after the last batch this wavefront un-blends: the tile's final colour, or a snapshot
			const double *after = (chunked && b_hi < nbatch - 1)
									  ? (const double *)(w.edge_snap + (size_t)(snap - 1) * SNAP_BYTES) + (size_t)b_hi * CH * 64
									  : (const double *)slot;
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				cur[cc] = after[cc * 64 + lane];
#pragma unroll
			for (int q = 0; q < EMAX / TB; q++)
				tm[q] = q < nbatch ? ((const uint16_t *)(slot + CH * 64 * sizeof(double)))[q * 64 + lane] : 0u;
		}
		else
		{
			pixel_base();
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				cur[cc] = base[cc];
		}
		for (int b = 0; b < nbatch && !sweep_saved; b++)
		{
			const int first = b * TB, nb = n_edges - first < TB ? n_edges - first : TB;
			const uint32_t ecov = stage_edge_batch(*(WaveLds *)&S, *es, w, P, first, nb, lane, x0, y0, W, inb);
			uint32_t tmb = 0;
			for (int j = 0; j < nb; j++)
			{
				const bool c = (ecov >> j) & 1u;
				if (__ballot(c) == 0)
					continue;
				const EdgeRec &eq = S.rec[j].edge();
				if (c && plane_at(eq.xZ, x, y) < zown)
				{
					tmb |= 1u << j;
					const double *qp = &S.planes[j * 12];
					const double Tq = plane_at(eq.x2t, x, y);
					Tap qtap;
					double qL = 0, qUV[2];
					if (eq.kind == KIND_TEXTURED && TEX)
						textured_tap(qp, x, y, false, 0.0, p.tex_w, p.tex_h, C, qtap, qL, qUV);
#pragma unroll
					for (int cc = 0; cc < CH; cc++)
						if (cc < C)
						{
							cur[cc] *= Tq;
							cur[cc] += (1 - Tq) * edge_channel<PixT, TEX>(eq, qp, texture, qtap, qL, cc, x, y, false, 0.0);
						}
				}
			}
#pragma unroll
			for (int bb = 0; bb < EMAX / TB; bb++)
				tm[bb] = bb == b ? tmb : tm[bb];
		}
		if (chunked && b_hi < nbatch - 1)
		{ // the gradient that reaches batch b_hi has been attenuated by every nearer edge drawn over the pixel.  The transparency
		  // planes of those edges are gathered into the (still idle) staging area with ONE round of loads: read from memory inside
		  // the loop they were a dependent round trip per edge -- 34 of them for the first batch of a 50-edge tile, the longest
		  // wavefront of the kernel (tools/tile_trace.py: 41 k cycles)
			const int r0 = (b_hi + 1) * TB;
			double *xt = (double *)&S.rec[0];
			static_assert(sizeof(S.rec) + sizeof(S.planes) >= 3 * sizeof(double) * EMAX, "room for the transparency planes of a tile's edges");
			lds_sync();
			for (int i = lane; i < n_edges - r0; i += 64)
			{
				const EdgeRec &eq = w.edge_rec[es->sorted[r0 + i]];
				xt[3 * i] = eq.x2t[0];
				xt[3 * i + 1] = eq.x2t[1];
				xt[3 * i + 2] = eq.x2t[2];
			}
			lds_sync();
			for (int r = r0; r < n_edges; r++)
			{
				uint32_t bits = 0;
#pragma unroll
				for (int bb = 0; bb < EMAX / TB; bb++)
					bits = bb == (r / TB) ? tm[bb] : bits;
				const double Tq = plane_at(xt + 3 * (r - r0), x, y);
				if ((bits >> (r % TB)) & 1u)
				{
#pragma unroll
					for (int cc = 0; cc < CH; cc++)
						g[cc] *= Tq;
				}
			}
		}
		// pass B, near -> far (H.h:2961-3052)
		edge_reverse_sweep<PixT, TEX>(p, w, S, es, lane, x, y, n_edges, b_hi, b_lo, !sweep_saved && b_hi == nbatch - 1, tm, cur, g, base, have_base, pixel_base);
	}

	if (EDGES && b_lo > 0)
		return; // the wavefront that ran batch 0 (the farthest edges) holds the gradient that reaches pass 1
	// ---- adjoint of pass 1: g now belongs to the triangle that owns the pixel
	owner_adjoint<PixT, TEX>(p, w, lane, x, y, owner, kind, g, tap, L, (double *)&S.rec[0], (uint32_t *)&S.cover[0][0]);
}

template <class PixT, bool TEX, int NC = 0> // NC: the channel count at compile time (0: the scene's), as for raster_fwd_fast_kernel
__global__ __launch_bounds__(64, 6) void raster_bwd_fast_kernel(KParams p)
{ // (two-call path) the forward's work list of the non-empty tiles, walked exactly as raster_fwd_fast_kernel walks it (same grid,
  // same entry of the list for the same workgroup); the tiles with silhouette edges are left to raster_bwd_edge_kernel.  One
  // wavefront per tile OF THE FRAME, which found out from the tile bitmap that two out of three had nothing to do, took ~51 us
  // per 8-view launch; this one ~43 (two-call step 0.254 -> 0.246 ms).
	__shared__ BwdLds s_lds;
	if (NC)
		p.C = NC, p.L.P = NC < 3 ? 3 : NC;
	const int lane = threadIdx.x & 63;
	const int G = p.tile_blocks;
	const long long b = blockIdx.x;
	const bool chunked = G % (8 * WORK_CHUNK) == 0;
	const int view = chunked ? (int)((b >> 3) % p.n_views) : (int)(b % p.n_views);
	const int q = chunked ? (int)((b >> 3) / p.n_views) * 8 + (int)(b & 7) : (int)(b / p.n_views);
	const ViewPtrs w = view_ptrs(p, view);
	const int Gh = chunked ? G / p.heavy_share : 0;
	const bool heavy_list = q < Gh;
	const int qq = heavy_list ? q : q - Gh, stride = heavy_list ? Gh : G - Gh;
	const bool chunk_here = chunked && stride % (8 * WORK_CHUNK) == 0;
	uint32_t rank = chunk_here ? (uint32_t)((((qq >> 3) / WORK_CHUNK) * 8 + (qq & 7)) * WORK_CHUNK + (qq >> 3) % WORK_CHUNK) : (uint32_t)qq;
	const uint32_t n_work = w.hdr->work_count[heavy_list ? 0 : 1];
	for (; rank < n_work; rank += (uint32_t)stride)
	{
		const WorkEntry &entry = w.work_list[heavy_list ? rank : (uint32_t)p.L.work_cap - 1u - rank];
		const int tile = uniform((int)entry.tile);
		if (uniform((int)entry.nedge) != 0)
			continue;
		int ln = lane;
		asm volatile("" : "+v"(ln)); // (see raster_fwd_fast_kernel: nothing lane-dependent is carried across the loop)
		bwd_fast_tile<PixT, false, TEX>(p, w, view, tile % p.L.tiles_x, tile / p.L.tiles_x, ln, s_lds, nullptr);
		lds_sync();
	}
}

#ifndef DR_EDGE_OCC
#define DR_EDGE_OCC 4 // waves per SIMD of the untextured edge kernel (3: no spills, 5: more) -- swept, 4 stays
#endif
template <class PixT, bool TEX, int NC = 0>
__global__ __launch_bounds__(64, TEX ? 2 : DR_EDGE_OCC) void raster_bwd_edge_kernel(KParams p)
{
	if (NC)
		p.C = NC, p.L.P = NC < 3 ? 3 : NC; // persistent waves over the lists of tiles that hold silhouette edges (built by tile_scan_kernel).  Grid (views, waves):
  // the first waves dispatched are wave 0 of every view, and every wave starts with the many-edged tiles -- the kernel
  // lasts as long as its slowest tile, so those must not start late.  Wave g takes the work items g, g + gridDim.y, ...
	__shared__ BwdLds s_lds;
	__shared__ EdgeSort s_es;
	const int view = blockIdx.x;
	const int lane = threadIdx.x;
	const ViewPtrs w = view_ptrs(p, view);
	// the last workgroups of the grid stream the background of this kernel's share of the empty tiles (fill_share)
	const int fill_n = fill_share(p.fill_mode, 0, p.L.nwords), fill_blocks = fill_share_blocks(fill_n);
#ifndef DR_FILL_FIRST
#define DR_FILL_FIRST 0 // measurement builds: 1 = the fill workgroups at the head of both grids instead of the tail
#endif
	const int walkers = (int)gridDim.y - fill_blocks;
	const int by = DR_FILL_FIRST ? (int)blockIdx.y - fill_blocks : (int)blockIdx.y; // index among the walkers (< 0: a fill workgroup)
	if (DR_FILL_FIRST ? by < 0 : by >= walkers)
	{
		for (int i = DR_FILL_FIRST ? (int)blockIdx.y : by - walkers; i < fill_n; i += fill_blocks)
			fill_share_word(p, 0, view, i, lane);
		return;
	}
	const uint32_t n_short = w.edge_tile_cnt[0], n_long = w.edge_tile_cnt[CNT_STRIDE], n_multi = w.edge_tile_cnt[2 * CNT_STRIDE] * CHUNKS;
	const uint32_t *shorts = w.edge_tiles, *longs = w.edge_tiles + p.L.ntiles, *multi = w.edge_tiles + 2 * (size_t)p.L.ntiles;
	// Work items: first the tiles with more than one batch of edges, each offered to CHUNKS wavefronts (one per batch of its
	// reverse sweep; those the tile has no use for return at once), then the other tiles with more than PRIO_EDGES edges, then
	// the rest.
#pragma nounroll
	for (uint32_t i = (uint32_t)by; i < n_multi + n_long + n_short; i += (uint32_t)walkers)
	{
		int tile, chunk = -1;
		if (i < n_multi)
			tile = (int)multi[i / CHUNKS], chunk = (int)(i % CHUNKS);
		else if (i < n_multi + n_long)
			tile = (int)longs[i - n_multi];
		else
			tile = (int)shorts[i - n_multi - n_long];
		tile = uniform(tile);
		bwd_fast_tile<PixT, true, TEX>(p, w, view, tile % p.L.tiles_x, tile / p.L.tiles_x, lane, s_lds, &s_es, chunk);
		lds_sync();
	}
}

// ------------------------------------------------------------------------------------------ antialiase_error: tiles with silhouette edges
//
// Round 6.  In this mode the edges blend the squared residual err_buffer instead of the image (rasterize_edge_*_error, H.h:2067-2197,
// 2371-2478; adjoint H.h:2200-2368, 2481-2618): far -> near,  err' = T err + (1 - T) Err  with  Err = sum_c (A_c - obs_c)^2  the squared
// distance between the colour the edge would paint and the observation.  The un-staged tile code replays, for every edge, all the
// edges behind it from records in memory -- O(n^2) dependent round trips, 300 us for ONE tile of 30 edges, which was the adjoint
// raster's whole duration (0.42 of the mode's 0.55 ms for one 1024^2 view of the benchmark scene).  Here: the staged forward sweep
// over the error buffer (masks of the drawn edges), then the reverse sweep -- un-blend  err = (err' - (1 - T) Err) / T  as the
// reference does, T_B = eb (err - Err), A_c_B = 2 (A_c - obs_c) (1 - T) eb, eb *= T -- with the same 15-moment butterfly and ONE atomic
// instruction per edge as edge_reverse_sweep, then the adjoint of pass 1 for image_b = -2 (obs - image) eb (H.h:3054-3060).
template <class PixT, bool TEX>
__device__ __forceinline__ void bwd_err_tile(const KParams &p, const ViewPtrs &w, int view, int tx, int ty, int lane, BwdLds &S, EdgeSort *es)
{
	const int W = p.W, H = p.H, C = p.C, P = p.L.P;
	const PixT *texture = (const PixT *)p.texture;
	PixT *texture_b = (PixT *)p.texture_b;
	const int tile = ty * p.L.tiles_x + tx;
	const int x0 = tx * TILE, y0 = ty * TILE;
	const int px = x0 + (lane & 7), py = y0 + (lane >> 3);
	const bool inb = px < W && py < H;
	const size_t pix = (size_t)py * W + px;
	const size_t vpix = (size_t)view * H * W + pix;
	const double x = px, y = py;
	const int32_t raw_owner = inb ? w.face_id[pix] : -1;
	const int nedge = (int)((uint32_t)uniform((int)w.edge_saved[tile]) & ~SWEEP_SAVED);
	if (nedge <= 0)
		return;
	const int n_edges = gather_sorted_edges(*es, w, p, tile, nedge, lane);
	if (n_edges < 0)
	{ // more than EMAX edges in one tile (or pool overflow): the un-staged code, right here
		lds_sync();
		bwd_tile_generic<PixT>(p, view, tx, ty, lane, (volatile uint32_t *)es->sorted);
		lds_sync();
		return;
	}
	int owner = -1, kind = KIND_NONE;
	unpack_owner(raw_owner, owner, kind);
	// what pass 1 left at this pixel: depth, un-antialiased colour (the image of this mode), its squared distance to the observation
	double ob[CH] = {0, 0, 0, 0}, base[CH] = {0, 0, 0, 0};
	if (inb)
	{
		const PixT *o = (const PixT *)p.obs + vpix * C;
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
			if (cc < C)
				ob[cc] = (double)o[cc];
	}
	const double *planes = nullptr;
	double zown = INFINITY;
	Tap tap;
	double L = 0, UV[2] = {0, 0};
	if (owner >= 0)
	{
		planes = w.tri_planes + (size_t)owner * 3 * P;
		zown = plane_at(w.tri_rec[owner].xZ, x, y);
		if (kind == KIND_TEXTURED && TEX)
			textured_tap(planes, x, y, false, 0.0, p.tex_w, p.tex_h, C, tap, L, UV);
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
			if (cc < C)
				base[cc] = kind == KIND_TEXTURED && TEX ? textured_channel(texture, tap, cc) * L : interp_channel(planes, cc, x, y, false, 0.0);
	}
	else if (inb)
	{
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
			if (cc < C)
				base[cc] = background_channel<PixT>(p, view, pix, cc);
	}
	double err0 = 0;
#pragma unroll
	for (int cc = 0; cc < CH; cc++)
		if (cc < C && inb)
			err0 += (base[cc] - ob[cc]) * (base[cc] - ob[cc]);
	// squared distance between the colour an edge would paint here and the observation (+ the colours themselves, for the adjoint)
	auto edge_err = [&](const EdgeRec &e, const double *ep, const Tap &etap, double eL, double (&A)[CH]) -> double {
		double Err = 0;
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
		{
			A[cc] = 0;
			if (cc < C)
			{
				A[cc] = edge_channel<PixT, TEX>(e, ep, texture, etap, eL, cc, x, y, false, 0.0);
				Err += (A[cc] - ob[cc]) * (A[cc] - ob[cc]);
			}
		}
		return Err;
	};
	// ---- pass A, far -> near: which edges are drawn over this pixel, and the error buffer they leave
	uint32_t tm[EMAX / TB] = {0, 0, 0, 0, 0, 0, 0, 0};
	double cur = err0;
	const int nbatch = (n_edges + TB - 1) / TB;
	for (int b = 0; b < nbatch; b++)
	{
		const int first = b * TB, nb = n_edges - first < TB ? n_edges - first : TB;
		const uint32_t ecov = stage_edge_batch(*(WaveLds *)&S, *es, w, P, first, nb, lane, x0, y0, W, inb);
		uint32_t tmb = 0;
		for (int j = 0; j < nb; j++)
		{
			const bool c = (ecov >> j) & 1u;
			if (__ballot(c) == 0)
				continue;
			const EdgeRec &eq = S.rec[j].edge();
			if (c && plane_at(eq.xZ, x, y) < zown)
			{
				tmb |= 1u << j;
				const double *qp = &S.planes[j * 12];
				const double Tq = plane_at(eq.x2t, x, y);
				Tap qtap;
				double qL = 0, qUV[2], A[CH];
				if (eq.kind == KIND_TEXTURED && TEX)
					textured_tap(qp, x, y, false, 0.0, p.tex_w, p.tex_h, C, qtap, qL, qUV);
				cur *= Tq;
				cur += (1 - Tq) * edge_err(eq, qp, qtap, qL, A);
			}
		}
#pragma unroll
		for (int bb = 0; bb < EMAX / TB; bb++)
			tm[bb] = bb == b ? tmb : tm[bb];
	}
	// ---- pass B, near -> far
	double eb = inb ? (double)((const PixT *)p.err_b)[vpix] : 0.0; // running adjoint of err_buffer at this pixel
	const RecSlot *erec = &S.rec[0];
	for (int b = nbatch - 1; b >= 0; b--)
	{
		const int first = b * TB, nb = n_edges - first < TB ? n_edges - first : TB;
		if (b < nbatch - 1) // (the records of the last batch of pass A are still staged)
		{
			lds_sync();
			if (lane < nb)
				S.ids[lane] = es->sorted[first + lane];
			lds_sync();
			stage_batch(*(WaveLds *)&S, w.edge_rec, w.edge_planes, P, nb, lane);
			lds_sync();
		}
		uint32_t tmb = 0;
#pragma unroll
		for (int bb = 0; bb < EMAX / TB; bb++)
			tmb = bb == b ? tm[bb] : tmb;
		for (int r = nb - 1; r >= 0; r--)
		{
			const bool hit = (tmb >> r) & 1u;
			if (__ballot(hit) == 0)
				continue;
			const EdgeRec &e = erec[r].edge();
			const double *ep = &S.planes[r * 12];
			const double Tr = hit ? plane_at(e.x2t, x, y) : 1.0;
			Tap etap;
			double eL = 0, eUV[2] = {0, 0}, A[CH] = {0, 0, 0, 0};
			if (e.kind == KIND_TEXTURED && TEX && hit)
				textured_tap(ep, x, y, false, 0.0, p.tex_w, p.tex_h, C, etap, eL, eUV);
			const double Err = hit ? edge_err(e, ep, etap, eL, A) : 0.0;
			// the error buffer before this edge: un-blend (H.h:2299, 2571) when T is safely away from 0, otherwise replay the earlier edges
			double prev = err0;
			const bool need_replay = hit && !(Tr > 1e-6);
			if (hit && !need_replay)
			{
				prev = (cur - (1 - Tr) * Err) / Tr;
				cur = prev;
			}
			if (__ballot(need_replay))
			{ // measure-zero event (pixel centre within 1e-6 sigma of the edge line): records straight from memory
				const int upto = first + r;
				for (int q = 0; q < upto; q++)
				{
					uint32_t tq = 0;
#pragma unroll
					for (int bb = 0; bb < EMAX / TB; bb++)
						tq = bb == (q / TB) ? tm[bb] : tq;
					if (!need_replay || !((tq >> (q % TB)) & 1u))
						continue;
					const uint32_t sq = es->sorted[q];
					const EdgeRec &eq = w.edge_rec[sq];
					const double *qp = w.edge_planes + (size_t)sq * 3 * P;
					const double Tq = plane_at(eq.x2t, x, y);
					Tap qtap;
					double qL = 0, qUV[2], Aq[CH];
					if (eq.kind == KIND_TEXTURED && TEX)
						textured_tap(qp, x, y, false, 0.0, p.tex_w, p.tex_h, C, qtap, qL, qUV);
					prev *= Tq;
					prev += (1 - Tq) * edge_err(eq, qp, qtap, qL, Aq);
				}
				if (need_replay)
					cur = prev;
			}
			// per-pixel plane adjoints of this edge (0 where it does not touch the pixel)
			double pb[5] = {0, 0, 0, 0, 0}; // planes 0..3 (colours, or u, v, shade) and the transparency plane
			if (hit)
			{
				const double Err_B = (1 - Tr) * eb;
				pb[4] = eb * (prev - Err);
				eb *= Tr;
				if (e.kind == KIND_TEXTURED && TEX)
				{ // H.h:2315-2326
					double L_B = 0, e_B[2] = {0, 0};
#pragma unroll
					for (int cc = 0; cc < CH; cc++)
						if (cc < C)
						{
							const double i00 = ldp(texture, etap.idx[0] + cc), i10 = ldp(texture, etap.idx[1] + cc);
							const double i01 = ldp(texture, etap.idx[2] + cc), i11 = ldp(texture, etap.idx[3] + cc);
							const double Amix = bilinear_mix(etap, i00, i10, i01, i11);
							const double diff_B = 2 * (Amix * eL - ob[cc]) * Err_B;
							L_B += diff_B * Amix;
							double wgt[4];
							bilinear_mix_adjoint(etap, diff_B * eL, i00, i10, i01, i11, wgt, e_B);
							if (texture_b)
								texture_scatter(texture_b, etap, cc, wgt);
						}
					pb[0] = etap.out[0] ? 0.0 : e_B[0];
					pb[1] = etap.out[1] ? 0.0 : e_B[1];
					pb[2] = L_B;
				}
				else
				{ // H.h:2579-2588 (with the row fold the reference forgot -- defect D2 -- as in the un-staged code)
#pragma unroll
					for (int cc = 0; cc < CH; cc++)
						if (cc < C)
							pb[cc] = 2 * (A[cc] - ob[cc]) * Err_B;
				}
			}
			// ... reduced over the tile on the VALU (DPP), then ONE atomic instruction (15 lanes) per edge and tile
			double *eacc = w.edge_acc + (size_t)S.ids[r] * (3 * P + 3);
			double mv[16]; // lane 3 * pl + m ends up with moment m of plane pl
#pragma unroll
			for (int pl = 0; pl < 5; pl++)
			{
				mv[3 * pl] = pb[pl] * x;
				mv[3 * pl + 1] = pb[pl] * y;
				mv[3 * pl + 2] = pb[pl];
			}
			mv[15] = 0;
			const double esum = wave_sum16(mv, lane);
			if (lane < 15 && esum != 0 && (lane >= 12 || lane < 3 * P))
			{
				const int pl = lane / 3, m = lane - 3 * pl;
				atomic_add_f64(eacc + (pl == 4 ? 3 * P : 3 * pl) + m, esum);
			}
		}
	}
	// ---- adjoint of pass 1: image_b = -2 (obs - image) eb (H.h:3054-3060), owned by the triangle that owns the pixel
	double g[CH];
#pragma unroll
	for (int cc = 0; cc < CH; cc++)
		g[cc] = (cc < C && inb) ? -2 * (ob[cc] - base[cc]) * eb : 0.0;
	lds_sync();
	owner_adjoint<PixT, TEX>(p, w, lane, x, y, owner, kind, g, tap, L, (double *)&S.rec[0], (uint32_t *)&S.cover[0][0], 384);
}

// persistent waves over the lists of tiles that hold silhouette edges, as raster_bwd_edge_kernel (a tile of several batches is listed once here)
template <class PixT, bool TEX>
__global__ __launch_bounds__(64, 2) void raster_bwd_edge_err_kernel(KParams p)
{
	__shared__ BwdLds s_lds;
	__shared__ EdgeSort s_es;
	const int view = blockIdx.x;
	const int lane = threadIdx.x;
	const ViewPtrs w = view_ptrs(p, view);
	const uint32_t n_short = w.edge_tile_cnt[0], n_long = w.edge_tile_cnt[CNT_STRIDE], n_multi = w.edge_tile_cnt[2 * CNT_STRIDE];
	const uint32_t *shorts = w.edge_tiles, *longs = w.edge_tiles + p.L.ntiles, *multi = w.edge_tiles + 2 * (size_t)p.L.ntiles;
#pragma nounroll
	for (uint32_t i = blockIdx.y; i < n_multi + n_long + n_short; i += gridDim.y)
	{ // (the many-edged tiles first: the kernel lasts as long as its slowest tile)
		int tile = i < n_multi ? (int)multi[i] : (i < n_multi + n_long ? (int)longs[i - n_multi] : (int)shorts[i - n_multi - n_long]);
		tile = uniform(tile);
		bwd_err_tile<PixT, TEX>(p, w, view, tile % p.L.tiles_x, tile / p.L.tiles_x, lane, s_lds, &s_es);
		lds_sync();
	}
}

} // namespace
