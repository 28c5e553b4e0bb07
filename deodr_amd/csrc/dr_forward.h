// deodr_amd/csrc/dr_forward.h -- part of the single translation unit dr_kernels.hip (device code, gfx950 / wave64).
// The LDS-staged forward: tile_scan_kernel (work lists), background fill, raster_fwd_fast_kernel (+ the fused adjoint of pass 1).
#pragma once

#include "dr_forward_generic.h"

using namespace dr;

namespace
{

// ---------------------------------------------------------------------------------- forward raster, LDS-staged fast path
//
// Same arithmetic as raster_fwd_kernel, restructured for latency: the tile's primitives are fetched with ONE batched
// load (ids -> 128-byte records + planes, 16 B per lane) into LDS instead of one dependent global round trip per
// primitive; the reference's scanline spans (two double divisions each, H.h:864-906) are computed once per
// (primitive, row) by lane = primitive_slot * 8 + row -- not once per pixel -- and exchanged as 8-bit column masks;
// the depth test and shading then read plane coefficients as LDS broadcasts.  Handles nb_colors <= 4 without
// antialiase_error; everything else runs on raster_fwd_kernel.

constexpr int TB = 16; // triangles (or edges) staged per batch: small, so that LDS never limits the number of resident waves

// A staged record: 128 bytes of TriRec or EdgeRec (same size) in a slot of 144 -- with a stride of 128 bytes (32 banks of 4 bytes) the
// same field of every record lies in the same bank, and the span lanes (eight triangles per pass) or the pixel lanes of a depth test
// (a few candidates per wavefront) were served one record after the other; 36 banks apart, eight records do not meet.
struct alignas(16) RecSlot
{
	uint4 bytes[8];
	uint4 pad;
	__device__ __forceinline__ const TriRec &tri() const { return *(const TriRec *)this; }
	__device__ __forceinline__ const EdgeRec &edge() const { return *(const EdgeRec *)this; }
};
static_assert(sizeof(TriRec) == 128 && sizeof(EdgeRec) == 128 && sizeof(RecSlot) == 144, "staged records");

struct alignas(16) WaveLds
{
	RecSlot rec[TB];		   // TriRec or EdgeRec
	double planes[TB * 12];	   // 3 * P doubles per primitive, P <= 4
	uint32_t ids[TB];
	uint8_t cover[TILE][TB];   // [row][primitive] -> bit x set when the primitive covers column x of the row
	uint32_t order[TB];
};


struct PixState
{
	double zbest;
	int kbest;
	int kind;
	int slot;	  // position of the winner in the staged batch (= in the tile's list when the tile has one batch)
	double v[CH]; // colours of the current winner (KIND_INTERP) or u, v, shade awaiting the texture fetch (KIND_TEXTURED)
};

__device__ __forceinline__ void lds_sync()
{ // the 64 lanes of a wave exchange data through LDS: order the compiler, the hardware executes DS ops in order
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t column_mask(int xb, int xe, int x0)
{
	int lo = (xb > x0 ? xb : x0) - x0, hi = (xe < x0 + TILE - 1 ? xe : x0 + TILE - 1) - x0;
	if (lo > hi)
		return 0;
	return ((1u << (hi + 1)) - 1u) & ~((1u << lo) - 1u);
}

// bit j of the result = bit `lx` of byte j of the 32-byte row `bytes` (coverage of my column by primitive j)
__device__ __forceinline__ uint32_t gather_column_bits(const uint8_t *row_bytes, int lx)
{
	uint32_t wd[TB / 4];
#pragma unroll
	for (int i = 0; i < TB / 16; i++)
	{
		const uint4 a = ((const uint4 *)row_bytes)[i];
		wd[4 * i] = a.x, wd[4 * i + 1] = a.y, wd[4 * i + 2] = a.z, wd[4 * i + 3] = a.w;
	}
	uint32_t m = 0;
#pragma unroll
	for (int i = 0; i < TB / 4; i++)
	{
		uint32_t t = (wd[i] >> lx) & 0x01010101u;
		m |= (((t * 0x01020408u) >> 24) & 0xfu) << (4 * i);
	}
	return m;
}

// stage `nb` primitives whose ids are in S.ids: records (128 B each, 8 lanes x 16 B) and planes (3P doubles each).
// Every global load is issued before the first LDS store: ONE memory round trip per batch (a rolled loop over the planes
// paid one per 64 doubles, i.e. two or three for a batch of more than five primitives).
template <class Rec, bool PLANES = true> // PLANES = false: records only (the many-channel forward shades its winner from memory)
__device__ __forceinline__ void stage_batch(WaveLds &S, const Rec *recs, const double *planes, int P, int nb, int lane)
{
	static_assert(TB == 16, "two record pieces and three plane doubles per lane");
	const int piece = lane & 7;
	const int np = 3 * P, total = PLANES ? nb * np : 0; // np = 9 or 12
	const int j0 = lane >> 3, j1 = 8 + (lane >> 3);
	const int i0 = lane, i1 = lane + 64, i2 = lane + 128;
	const int a0 = np == 12 ? i0 / 12 : i0 / 9, a1 = np == 12 ? i1 / 12 : i1 / 9, a2 = np == 12 ? i2 / 12 : i2 / 9;
	const int c0 = i0 - a0 * np, c1 = i1 - a1 * np, c2 = i2 - a2 * np;
	uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
	double v0 = 0, v1 = 0, v2 = 0;
	if (j0 < nb)
		r0 = ((const uint4 *)(recs + S.ids[j0]))[piece];
	if (j1 < nb)
		r1 = ((const uint4 *)(recs + S.ids[j1]))[piece];
	if (i0 < total)
		v0 = planes[(size_t)S.ids[a0] * np + c0];
	if (i1 < total)
		v1 = planes[(size_t)S.ids[a1] * np + c1];
	if (i2 < total)
		v2 = planes[(size_t)S.ids[a2] * np + c2];
	if (j0 < nb)
		((uint4 *)&S.rec[j0])[piece] = r0;
	if (j1 < nb)
		((uint4 *)&S.rec[j1])[piece] = r1;
	if (i0 < total)
		S.planes[a0 * 12 + c0] = v0;
	if (i1 < total)
		S.planes[a1 * 12 + c1] = v1;
	if (i2 < total)
		S.planes[a2 * 12 + c2] = v2;
}

template <bool TEX, bool SHADE = true> // SHADE = false: the winner is only remembered (index, depth, kind), not shaded
__device__ __forceinline__ void tri_batch(const KParams &p, WaveLds &S, int nb, int lane, int x0, int y0, bool inb, PixState &st)
{
	const int W = p.W, H = p.H, C = p.C;
	const bool persp = p.persp, strict = p.strict;
	// spans: lane = slot * 8 + row
#pragma unroll
	for (int q = 0; q < TB / 8; q++)
	{
		const int j = q * 8 + (lane >> 3), r = lane & 7;
		uint32_t m = 0;
		if (j < nb)
		{
			const TriRec &rec = S.rec[j].tri();
			if (rec.kind != KIND_NONE)
			{
				// A row lies in one half of the triangle (above or below its middle vertex), so one span (two divisions, not
				// four) per (triangle, row); only the non-strict fill rule puts the middle-vertex row in both halves.
				const int yy = y0 + r;
				const bool in0 = yy >= rec.y_begin[0] && yy <= rec.y_end[0], in1 = yy >= rec.y_begin[1] && yy <= rec.y_end[1];
				int xb, xe;
				tri_half_span(rec, in0 ? 0 : 1, yy, W, H, strict, xb, xe);
				m = column_mask(xb, xe, x0);
				if (__ballot(in0 && in1))
				{
					if (in0 && in1)
					{
						tri_half_span(rec, 1, yy, W, H, strict, xb, xe);
						m |= column_mask(xb, xe, x0);
					}
				}
			}
		}
		if (j < TB)
			S.cover[r][j] = (uint8_t)m;
	}
	lds_sync();
	const int lx = lane & 7, row = lane >> 3;
	uint32_t mine = gather_column_bits(&S.cover[row][0], lx);
	if (!inb)
		mine = 0;
	const double x = x0 + lx, y = y0 + row;
	// Depth test: every lane walks the triangles that cover ITS pixel (bits of `mine`), not the triangles of the batch -- with
	// back-face culling a pixel is covered by one triangle, rarely two, so the wavefront makes one or two passes instead of one
	// per triangle of the batch (the 93-triangle tile at the limb of the sphere: 96 -> ~12).  The winner is remembered by its slot
	// and shaded ONCE after the loop.  Lane-varying LDS addresses: a few distinct records per pass.
	int jbest = -1;
	uint32_t todo = mine;
	while (__ballot(todo != 0))
	{
		const bool act = todo != 0;
		const int j = act ? __ffs((int)todo) - 1 : 0;
		todo &= todo - 1;
		double Z = plane_at(S.rec[j].tri().xZ, x, y);
		if (persp)
			Z = 1 / Z;
		const int k = (int)S.ids[j];
		if (act && (Z < st.zbest || (Z == st.zbest && k < st.kbest)))
		{
			st.zbest = Z;
			st.kbest = k;
			jbest = j;
		}
	}
	if (!SHADE && jbest >= 0)
	{
		st.slot = jbest;
		st.kind = S.rec[jbest].tri().kind;
	}
	if (SHADE && jbest >= 0)
	{ // per-lane reads of the winner's record and planes (LDS, a few distinct slots per tile)
		st.slot = jbest;
		const int kind = S.rec[jbest].tri().kind;
		const double *pl = &S.planes[jbest * 12];
		const double Z = st.zbest;
		st.kind = kind;
		if (kind == KIND_TEXTURED && TEX)
		{
			st.v[0] = plane_at(pl, x, y);
			st.v[1] = plane_at(pl + 3, x, y);
			st.v[2] = plane_at(pl + 6, x, y);
			if (persp)
			{
				st.v[2] = st.v[2] * Z;
				st.v[0] = st.v[0] * Z;
				st.v[1] = st.v[1] * Z;
			}
		}
		else
		{
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				if (cc < C)
					st.v[cc] = interp_channel(pl, cc, x, y, persp, Z);
		}
	}
	lds_sync(); // the next batch overwrites the staging area
}

constexpr int EMAX = 128; // silhouette edges of one tile the staged kernels can order; more -> generic / deferred path (a
						  // single 90-edge tile in the deferred kernel took 5 ms)

struct EdgeSort
{
	double keys[EMAX];
	uint32_t ids[EMAX];
	uint32_t sorted[EMAX];
};

// All edges of the tile (inline list + its pairs in the spill pool), ordered far -> near (ties by slot) into es.sorted.
// Returns their number, or -1 when there are more than EMAX (or the pool overflowed and some are missing).
__device__ __forceinline__ int gather_sorted_edges(EdgeSort &es, const ViewPtrs &w, const KParams &p, int tile, int nedge, int lane)
{
	const int n_inline = nedge < K_EDGE ? nedge : K_EDGE;
	if (lane < n_inline)
		es.ids[lane] = w.edge_list[(size_t)tile * K_EDGE + lane];
	int fill = n_inline;
	if (nedge > K_EDGE)
	{
		if (nedge > EMAX)
			return -1;
		uint32_t spill_n = w.hdr->edge_spill[w.hdr->cur];
		if (spill_n > p.L.edge_pool_cap)
			spill_n = p.L.edge_pool_cap;
		for (uint32_t i0 = 0; i0 < spill_n; i0 += 64)
		{
			const uint2 pr = (i0 + lane < spill_n) ? w.edge_pool[i0 + lane] : make_uint2(0xffffffffu, 0u);
			const unsigned long long m = __ballot((int)pr.x == tile);
			const int cnt = __popcll(m);
			if (fill + cnt > EMAX)
				return -1;
			if ((m >> lane) & 1ull)
				es.ids[fill + __popcll(m & ((1ull << lane) - 1ull))] = pr.y;
			fill += cnt;
		}
		if (fill != nedge)
			return -1; // pairs lost to a pool overflow: the host repeats the call with a larger pool
	}
	lds_sync();
	for (int i = lane; i < fill; i += 64)
		es.keys[i] = w.edge_rec[es.ids[i]].key;
	lds_sync();
	for (int i = lane; i < fill; i += 64)
	{
		const double key = es.keys[i];
		const uint32_t slot = es.ids[i];
		int rank = 0;
		for (int j = 0; j < fill; j++)
			rank += edge_before(es.keys[j], es.ids[j], key, slot) ? 1 : 0;
		es.sorted[rank] = slot;
	}
	lds_sync();
	return fill;
}

// stage edges sorted[first .. first + nb) and turn their scanline spans into column masks; returns, per pixel, the
// 32-bit mask of the batch's edges whose band covers it
__device__ __forceinline__ uint32_t stage_edge_batch(WaveLds &S, const EdgeSort &es, const ViewPtrs &w, int P, int first, int nb, int lane, int x0,
													 int y0, int W, bool inb)
{
	lds_sync();
	if (lane < nb)
		S.ids[lane] = es.sorted[first + lane];
	lds_sync();
	stage_batch(S, w.edge_rec, w.edge_planes, P, nb, lane);
	lds_sync();
	const RecSlot *erec = S.rec;
#pragma unroll
	for (int q = 0; q < TB / 8; q++)
	{
		const int j = q * 8 + (lane >> 3), r = lane & 7;
		uint32_t m = 0;
		if (j < nb)
		{
			const EdgeRec &e = erec[j].edge();
			const int yy = y0 + r;
			if (yy >= e.y_begin && yy <= e.y_end)
			{
				int xb, xe;
				edge_row_span(e, yy, W, xb, xe);
				m = column_mask(xb, xe, x0);
			}
		}
		S.cover[r][j] = (uint8_t)m;
	}
	lds_sync();
	return inb ? gather_column_bits(&S.cover[lane >> 3][0], lane & 7) : 0u;
}

template <class PixT, bool TEX>
__device__ __forceinline__ void owner_adjoint(const KParams &p, const ViewPtrs &w, int lane, double x, double y, int owner, int kind, const double *g,
											  const Tap &tap, double L, double *tab, uint32_t *own, int win_cap = 384);

template <int NPIX>
__device__ __forceinline__ void owner_adjoint_slots(const KParams &p, const ViewPtrs &w, int lane, int x0, int y0, const int (&slot)[NPIX],
													const float (&g)[NPIX][CH], uint32_t id_of_slot, int nslots, float *tab);

// (dr_backward.h) adjoint of pass 2 for batches b_hi .. b_lo of a tile's blending order; (dr_backward_generic.h) the un-staged adjoint
template <class PixT, bool TEX, int NBATCH, class Lds, class BaseFn> // (NBATCH: batches of TB edges the caller's masks cover)
__device__ __forceinline__ void edge_reverse_sweep(const KParams &p, const ViewPtrs &w, Lds &S, const EdgeSort *es, int lane, double x, double y, int n_edges,
												   int b_hi, int b_lo, bool top_staged, const uint32_t (&tm)[NBATCH], double (&cur)[CH], double (&g)[CH],
												   double (&base)[CH], bool &have_base, BaseFn pixel_base, int r_lo = 0, int r_hi = TB - 1);
template <class PixT, bool LEAN, bool TEX>
__device__ __forceinline__ void bwd_tile_generic_impl(const KParams &p, int view, int tx, int ty, int lane, volatile uint32_t *order);

// Background of one tile that received no primitive: colour, depth = +inf, no owner (H.h:2728-2744).
template <class PixT>
__device__ __forceinline__ void fill_background_tile(const KParams &p, int view, int32_t *face_id, int tx, int ty, int lane, const double *bgc,
													 int owners)
{ // owners: 1 = also the owner ids (none), 0 = not, -1 = colour only (the caller writes depth and owners of four tiles at once)
	const int W = p.W, H = p.H, C = p.C;
	const int px = tx * TILE + (lane & 7), py = ty * TILE + (lane >> 3);
	if (px >= W || py >= H)
		return;
	const size_t pix = (size_t)py * W + px;
	const size_t vpix = (size_t)view * H * W + pix;
	if (p.image)
	{
		PixT *out = (PixT *)p.image + vpix * C;
		double col[CH];
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
			col[cc] = (cc < C && p.bg_image) ? (double)((const PixT *)p.bg_image)[vpix * C + cc] : bgc[cc];
		if (C == 4)
		{
			typedef PixT V4 __attribute__((ext_vector_type(4)));
			const V4 v = {(PixT)col[0], (PixT)col[1], (PixT)col[2], (PixT)col[3]};
			__builtin_nontemporal_store(v, (V4 *)out);
		}
		else
		{
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				if (cc < C)
					__builtin_nontemporal_store((PixT)col[cc], out + cc);
		}
	}
	if (owners < 0)
		return;
	if (p.zbuf)
		__builtin_nontemporal_store((PixT)INFINITY, (PixT *)p.zbuf + vpix);
	if (owners)
		__builtin_nontemporal_store((int32_t)-1, face_id + pix);
}

// ------------------------------------------------------------------------------------------------ tile scan
//
// Between set-up and the staged forward raster: one thread per tile turns the per-tile counters that binning left into
//   * the work list of the forward: one uint4 {tile, triangles, edges, sweep slot} per NON-EMPTY tile -- the tiles with more
//     than FIRST_PRIMS triangles or edges from the front of the array (the long poles start first), the others from the back;
//   * the tile bitmap (bit = the tile received a primitive) that the fill waves of the forward and the adjoint's owner-tile
//     kernel read;
//   * edge_saved[tile] (edge count + whether the forward will save its sweep), and the counters zeroed for the next forward.
// Two tiles out of three receive nothing: this is what lets the forward launch one wavefront per tile that HAS work instead of
// one per tile of the frame (the waves of the empty tiles used to take a third of its slot-time), and it takes the
// many-primitive-tile flags and lists (two more dependent atomics per lane) out of the set-up kernel.
#ifndef DR_WORK_CHUNK
#define DR_WORK_CHUNK 64
#endif
constexpr int SCAN_BLOCK = SCAN_TILES, WORK_CHUNK = DR_WORK_CHUNK;
#ifndef DR_DYN_WALKERS
#define DR_DYN_WALKERS 0 // 1: persistent walkers with tickets on the others' list of a many-view fit step (KParams::dyn_groups)
#endif
#ifndef DR_PAIR_TILES
#define DR_PAIR_TILES 1 // (measurement builds: 0 = one tile per wavefront everywhere, as in round 2)
#endif
constexpr uint32_t PAIR_FLAG = 0x80000000u; // in WorkEntry::tile of a pair of tiles (fwd_pair_tiles); then WorkEntry::ntri = nA | nB << 16
static_assert(FIRST_PRIMS <= 8, "a paired tile carries at most eight triangle ids (two 16-byte pieces of its inline list)");
// One tile workgroup in `heavy_share` walks the list of the many-primitive tiles (the head of the grid: dispatched first).  One in
// eight, unless the head of all views together would then take more than ~40 % of the chip's wave slots (5 120 at five waves per
// SIMD): with every slot of the first dispatch round on a 25 - 50 us tile the short tiles -- whose arithmetic hides those tiles'
// round trips -- start late.  Measured on the 8-view benchmark step: 1/8 0.183 ms, 1/12 0.1775, 1/16 0.1767, 1/24 0.1784; on one
// 2048^2 view (2 048 head workgroups at 1/8) 1/16 costs 4 %; on one 1024^2 view 1/2 0.0775, 1/4 0.0772, 1/8 0.0809, 1/16 0.090 ms.
#ifndef DR_HEAVY_SHARE
#define DR_HEAVY_SHARE 0 // measurement builds: a fixed share
#endif
__host__ inline int heavy_share_for(int n_views, int tile_blocks, bool fuse_edges)
{
	if (DR_HEAVY_SHARE)
		return DR_HEAVY_SHARE;
	// ~2 048 head workgroups over all views (40 % of the wave slots), the share a power of two between 1/4 and 1/16.  Twice as many
	// when the head also holds every tile with silhouette edges and runs their adjoint (a fit step: ~1 000 head entries per view of
	// the benchmark scene, 20 - 60 us each -- with 256 walkers per view the forward ended 14 us after its last short tile)
	const long long want = ((long long)n_views * tile_blocks + (fuse_edges ? 4095 : 2047)) / (fuse_edges ? 4096 : 2048);
	// (the final code of round 3, fit step of the benchmark scene, same-box A/B: 8 views 1/2 0.1316, 1/4 0.1305, 1/8 0.1328, 1/16 0.148 ms;
	// 16 views 1/4 0.2538, 1/16 0.2483; 32 views 1/4 0.515, 1/16 0.500: about one head entry per head walker up to 8 views of 1024^2)
	if (fuse_edges && want <= 8)
		return 4;
	int share = 4;
	while (share < 16 && share < want)
		share *= 2;
	return share;
}

#ifndef DR_SPLIT_EDGES
#define DR_SPLIT_EDGES 1 // (measurement builds: 0 = a tile is one work item whatever its number of edges)
#endif
constexpr uint32_t SPLIT_FLAG = 0x80000000u; // in WorkEntry::nedge: bits 16 .. 19 = which part of the edges this copy of the tile back-propagates
#ifndef DR_SPLIT_PART
#define DR_SPLIT_PART 0 // (measurement builds: a fixed number of edges per part)
#endif
// Edges per part (KParams::split_part): 8 when the launch is about one dispatch round of walkers (one or two 1024^2 views: the longest
// wavefront decides, 1 view 0.0607 -> 0.0585 ms), a whole batch of 16 otherwise (8 views: the repeated forward parts cost 1.5 us).
__host__ inline int split_part_for(int n_views, int tile_blocks) { return DR_SPLIT_PART ? DR_SPLIT_PART : ((long long)n_views * tile_blocks <= 8192 ? 8 : 16); }

__global__ __launch_bounds__(SCAN_BLOCK) void tile_scan_kernel(KParams p)
{
	// classes compacted by this kernel: 0 many-primitive tiles (front of the work list), 1 the other non-empty tiles (back of it),
	// 2 .. 4 the three lists of edge tiles, 5 every edge tile (its rank is the tile's slot in edge_sweep)
	constexpr int NCLS = 3 + EDGE_LISTS;
	__shared__ uint32_t s_cnt[NCLS][SCAN_BLOCK / 64];
	__shared__ uint32_t s_base[NCLS];
	__shared__ uint32_t s_req[SCAN_BLOCK / 64]; // extra copies of split tiles the wavefronts of this block ask for
	kernel_stamp(p, 1);
	const int view = blockIdx.y;
	const ViewPtrs w = view_ptrs(p, view);
	const int tile = blockIdx.x * SCAN_BLOCK + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const bool valid = tile < p.L.ntiles;
	uint32_t ntri = 0, nedge = 0;
	uint4 ida = make_uint4(0, 0, 0, 0), idb = ida, idc = ida;
	if (valid)
	{
		ntri = w.tri_cnt[tile];
		nedge = w.edge_cnt[tile];
	}
	const bool work = (ntri | nedge) != 0;
	if (work)
	{ // the head of the tile's inline list, only as far as it is filled (two tiles out of three are empty: requested with the
	  // counters, these 48 bytes per tile were 6 MB of reads per step for nothing and the kernel took 6.5 instead of 5 us)
		static_assert(ENTRY_IDS == 12 && K_TRI >= ENTRY_IDS, "three 16-byte pieces of the tile's inline list");
		const uint4 *ids = (const uint4 *)(w.tri_list + (size_t)tile * K_TRI);
		ida = ids[0];
		// (`ntri > 4 ? ids[1] : ida` became a select between two ADDRESSES, which put ida into scratch memory for every thread)
		idb = ids[ntri > 4 ? 1 : 0];
		idc = ids[ntri > 8 ? 2 : 0];
	}
	if (work)
	{ // self-cleaning counters
		w.tri_cnt[tile] = 0;
		w.edge_cnt[tile] = 0;
	}
	const unsigned long long wm = __ballot(work);
	if (lane == 0 && valid)
		w.tile_bits[tile >> 5] = (uint32_t)wm;
	if (lane == 32 && valid)
		w.tile_bits[tile >> 5] = (uint32_t)(wm >> 32);
	// ---- compaction: rank inside the wavefront, wavefront totals through LDS, ONE atomic per class and block
	// (fit step with fused edge tiles: every tile with edges is a long wavefront -- forward, reverse sweep, pass 1 -- and must be in
	// the head of the list, which the workgroups compiled for that adjoint walk)
	const bool heavy = work && p.tile_blocks % (8 * WORK_CHUNK) == 0 &&
					   (ntri > (uint32_t)FIRST_PRIMS || nedge > (uint32_t)(p.fuse_edges ? 0 : FIRST_PRIMS));
	const int elist = nedge == 0 ? -1 : (nedge <= (uint32_t)PRIO_EDGES ? 0 : (nedge <= (uint32_t)TB ? 1 : 2));
	// Pairs for fwd_pair_tiles (fit step, untextured scene, even number of tile columns so that the left tile of a pair is an even
	// lane and its right neighbour the next lane): tiles (2 i, 2 i + 1) of a tile row, both with triangles, neither with an edge nor
	// in the head of the list, at most ENTRY_IDS triangles together.  The left tile's thread lists the pair, the right one's only
	// adds its triangle ids to that entry.
	bool pair_left = false, pair_right = false;
	uint32_t ntri_right = 0;
	// (textured scenes, round 6: below DR_TEX_TWO_KERNELS views per launch only -- KParams::pair_tex, the host's rule: from 8 views on the head walkers
	// are a kernel of their own and the critical path, pairs among the others bought nothing there and cost 11 %: profiles/r06tp_*)
	if (DR_PAIR_TILES && p.fuse_edges && (!p.texture || p.pair_tex) && (p.L.tiles_x & 1) == 0 && p.tile_blocks % (8 * WORK_CHUNK) == 0)
	{
		const bool plain = work && !heavy && nedge == 0 && ntri > 0;
		const uint32_t n_next = (uint32_t)__shfl_down((int)(plain ? ntri : 0u), 1, 64), n_prev = (uint32_t)__shfl_up((int)(plain ? ntri : 0u), 1, 64);
		pair_left = plain && !(lane & 1) && n_next > 0 && ntri + n_next <= (uint32_t)ENTRY_IDS;
		pair_right = plain && (lane & 1) && n_prev > 0 && ntri + n_prev <= (uint32_t)ENTRY_IDS;
		ntri_right = n_next;
	}
	// A fit step's tile with more than one batch of silhouette edges (17 .. EMAX) is listed once per KParams::split_part edges: every copy ("part")
	// runs pass 1 and the forward sweep over all the edges, but the reverse sweep -- 1.4 k cycles per edge, the long part -- of its own
	// edges only (the first part also the adjoint of pass 1, the last one the frame stores).  One view of the benchmark scene waited 45 us
	// for ONE wavefront with 43 triangles and 37 edges.
	// The copies are entries of the work list, next to the tile's own (a walker finds them without another look at the counters, and they
	// start with their tile, early).  Each block of SCAN_BLOCK tiles may add SPLIT_BUDGET of them, granted in tile order -- the list is laid
	// out for that many and no more: a soup of 20 000 slivers at sigma = 3 has more tiles of 17+ edges than empty ones, and its copies,
	// unbounded, ran into the entries that come from the other end of the list (memory fault found by tests/fuzz_parity.py).
	const unsigned long long below = (1ull << lane) - 1ull;
	uint32_t extra = (DR_SPLIT_EDGES && heavy && p.fuse_edges && nedge > (uint32_t)TB && nedge <= (uint32_t)EMAX)
						 ? (nedge + (uint32_t)p.split_part - 1) / (uint32_t)p.split_part - 1u
						 : 0u; // 0 .. 15 copies asked for
	if (p.fuse_edges && DR_SPLIT_EDGES)
	{ // (a condition of the launch: every thread of the block takes the barrier)
		const unsigned long long r0 = __ballot(extra & 1u), r1 = __ballot(extra & 2u), r2 = __ballot(extra & 4u), r3 = __ballot(extra & 8u);
		if (lane == 0)
			s_req[wave] = (uint32_t)(__popcll(r0) + 2 * __popcll(r1) + 4 * __popcll(r2) + 8 * __popcll(r3));
		__syncthreads();
		uint32_t before = (uint32_t)(__popcll(r0 & below) + 2 * __popcll(r1 & below) + 4 * __popcll(r2 & below) + 8 * __popcll(r3 & below));
		for (int i = 0; i < wave; i++)
			before += s_req[i];
		if (before + extra > (uint32_t)SPLIT_BUDGET)
			extra = 0; // (not granted: the tile stays one work item)
	}
	const uint32_t nparts = extra + 1u;
	const unsigned long long xb0 = __ballot(extra & 1u), xb1 = __ballot(extra & 2u), xb2 = __ballot(extra & 4u), xb3 = __ballot(extra & 8u);
	static_assert(EMAX / 8 <= 16 && TB % 8 == 0, "split_part is 8 or 16: four bits of extra parts; a part lies in one batch");
	unsigned long long m[NCLS];
	m[0] = __ballot(heavy);
	m[1] = wm & ~m[0] & ~__ballot(pair_right);
#pragma unroll
	for (int c = 0; c < EDGE_LISTS; c++)
		m[2 + c] = __ballot(elist == c);
	m[2 + EDGE_LISTS] = __ballot(nedge > 0);
	if (lane < NCLS)
	{
		unsigned long long mine = 0;
#pragma unroll
		for (int c = 0; c < NCLS; c++)
			mine = lane == c ? m[c] : mine;
		s_cnt[lane][wave] = (uint32_t)__popcll(mine) + (lane == 0 ? (uint32_t)(__popcll(xb0) + 2 * __popcll(xb1) + 4 * __popcll(xb2) + 8 * __popcll(xb3)) : 0u);
	}
	__syncthreads();
	if (threadIdx.x < NCLS)
	{
		uint32_t total = 0;
#pragma unroll
		for (int i = 0; i < SCAN_BLOCK / 64; i++)
			total += s_cnt[threadIdx.x][i];
		uint32_t *counter = threadIdx.x < 2 ? &w.hdr->work_count[threadIdx.x] : &w.edge_tile_cnt[(threadIdx.x - 2) * CNT_STRIDE];
		s_base[threadIdx.x] = total ? atomicAdd(counter, total) : 0u;
	}
	__syncthreads();
	auto position = [&](int c, unsigned long long members) { // of this thread in class c (the thread must belong to it)
		uint32_t at = s_base[c];
		for (int i = 0; i < wave; i++)
			at += s_cnt[c][i];
		return at + (uint32_t)__popcll(members & below);
	};
	// the adjoint finds the edge count, and whether the forward sweep over the edges is saved, in edge_saved
	uint32_t sweep_slot = 0;
	if (nedge > 0)
	{
		const uint32_t at = position(2 + EDGE_LISTS, m[2 + EDGE_LISTS]);
		const uint32_t slot_word = at < (uint32_t)p.L.sweep_cap ? at + 1u : 0u;
		w.edge_slot[tile] = slot_word; // always written: a stale value must never be read
		sweep_slot = (nedge <= (uint32_t)EMAX && !p.persp && !p.fuse_edges && !p.aa_err) ? slot_word : 0u; // (fused: nothing is saved for a later kernel;
		// antialiase_error: the sweep runs over the error buffer, its adjoint over the un-staged code)
		static_assert(EDGE_LISTS == 3, "select below");
		w.edge_tiles[(size_t)elist * p.L.ntiles + position(2 + elist, elist == 0 ? m[2] : (elist == 1 ? m[3] : m[4]))] = (uint32_t)tile;
	}
	if (valid)
		w.edge_saved[tile] = nedge | (sweep_slot ? SWEEP_SAVED : 0u);
	// (every lane takes part in the exchange: the right tile of a pair needs the position of the left tile's entry)
	const uint32_t my_pos = (work && !heavy && !pair_right) ? (uint32_t)p.L.work_cap - 1u - position(1, m[1]) : 0u;
	const uint32_t left_pos = (uint32_t)__shfl_up((int)my_pos, 1, 64), left_ntri = (uint32_t)__shfl_up((int)ntri, 1, 64);
	if (pair_right)
	{ // this tile's ids behind the left tile's, in the left tile's entry
		uint32_t *ids = w.work_list[left_pos].ids + left_ntri;
		// (a paired tile is not in the head: at most FIRST_PRIMS = 8 triangles.  Component by component, no array: an array of the
		// eight ids sent `ida` through scratch memory, store + reload, in every thread of the kernel)
#define DR_PUT_ID(j, v)      \
	if ((uint32_t)(j) < ntri) \
	ids[j] = (v)
		DR_PUT_ID(0, ida.x);
		DR_PUT_ID(1, ida.y);
		DR_PUT_ID(2, ida.z);
		DR_PUT_ID(3, ida.w);
		DR_PUT_ID(4, idb.x);
		DR_PUT_ID(5, idb.y);
		DR_PUT_ID(6, idb.z);
		DR_PUT_ID(7, idb.w);
	}
	else if (pair_left)
	{
		WorkEntry &e = w.work_list[my_pos];
		((uint4 *)&e)[0] = make_uint4((uint32_t)tile | PAIR_FLAG, ntri | (ntri_right << 16), 0u, 0u);
		uint32_t *ids = e.ids; // (only its own ids: the right tile's thread writes behind them)
		DR_PUT_ID(0, ida.x);
		DR_PUT_ID(1, ida.y);
		DR_PUT_ID(2, ida.z);
		DR_PUT_ID(3, ida.w);
		DR_PUT_ID(4, idb.x);
		DR_PUT_ID(5, idb.y);
		DR_PUT_ID(6, idb.z);
		DR_PUT_ID(7, idb.w);
#undef DR_PUT_ID
	}
	else if (work)
	{
		// (the copies of a split tile: the entries behind its own)
		const uint32_t head_pos = heavy ? position(0, m[0]) + (uint32_t)(__popcll(xb0 & below) + 2 * __popcll(xb1 & below) + 4 * __popcll(xb2 & below) + 8 * __popcll(xb3 & below)) : 0u;
		for (uint32_t part = 0; part < nparts; part++)
		{
			WorkEntry &e = heavy ? w.work_list[head_pos + part] : w.work_list[my_pos];
			uint4 *out = (uint4 *)&e;
			out[0] = make_uint4((uint32_t)tile, ntri, nparts > 1 ? (SPLIT_FLAG | part << 16 | nedge) : nedge, sweep_slot);
			out[1] = ida;
			out[2] = idb;
			out[3] = idc;
		}
	}
}

// ------------------------------------------------------------------------------------------------ background fill
//
// The background of the tiles that received no primitive (two out of three): 110 MB of plain stores per 8-view step that
// depend on nothing but the tile bitmap.  As workgroups of the forward raster they cost it 22 us: a fill wave lives as long as
// the store queue lets it, and it holds one of the forward's (register-fat) wave slots while it does.  As a kernel of its own,
// with 24 registers per lane, launched on a side stream right after the scan, its waves fit into the registers and wave
// slots the forward / edge / finalize kernels leave unused, and the stores drain while those kernels compute.
// One wavefront per bitmap word (32 tiles).  Four consecutive empty tiles of a tile row share one 16-byte-per-lane store of
// depth (and of owner ids): 128 contiguous bytes per pixel row instead of 4 x 32.
constexpr int FILL_WAVES = 4; // wavefronts (bitmap words) per workgroup

// Background of the run of empty tiles [txa, txb) of tile row ty.  Every pixel row of the run is ONE contiguous range of the
// frame (image: (txb - txa) * 8 * C elements, depth / owner ids: (txb - txa) * 8), written in 16-byte pieces by consecutive
// lanes whatever the channel count -- per tile and per channel (three strided 4-byte stores per lane for C = 3) the fill of a
// 1024^2 x 8-view batch of the hand mesh ran at 1 TB/s.  Needs W % 8 == 0 (16-byte alignment of every piece).
template <class PixT>
__device__ __forceinline__ void fill_run(const KParams &p, int view, int32_t *face_id, int ty, int txa, int txb, int lane, const double *bgc, int owners)
{
	constexpr int E = 16 / (int)sizeof(PixT); // elements per piece
	typedef PixT VE __attribute__((ext_vector_type(E)));
	typedef int32_t I4 __attribute__((ext_vector_type(4)));
	const int W = p.W, H = p.H, C = p.C;
	const int x0 = txa * TILE, npx = (txb * TILE < W ? txb * TILE : W) - x0, y0 = ty * TILE, rows = H - y0 < TILE ? H - y0 : TILE;
	// f(row, piece) for the `rows` x n pieces of a plane, consecutive lanes on consecutive pieces; no integer division in the loop
	// (the fill waves live on store issue: every instruction between two stores counts)
	auto for_pieces = [&](int n, auto f) {
		if (n >= 64)
		{
			for (int row = 0; row < rows; row++)
				for (int piece = lane; piece < n; piece += 64)
					f(row, piece);
			return;
		}
		const float rn = 1.0f / (float)n; // rows * n <= 8 * 63: exact after one correction step
		for (int idx = lane; idx < rows * n; idx += 64)
		{
			int row = (int)((float)idx * rn), piece = idx - row * n;
			if (piece < 0)
				row--, piece += n;
			if (piece >= n)
				row++, piece -= n;
			f(row, piece);
		}
	};
	if (p.image)
	{
		PixT *img = (PixT *)p.image + ((size_t)view * H * W + (size_t)y0 * W + x0) * C;
		const PixT *bgi = p.bg_image ? (const PixT *)p.bg_image + ((size_t)view * H * W + (size_t)y0 * W + x0) * C : nullptr;
		const size_t row_stride = (size_t)W * C;
		const int n = npx * C / E; // pieces per pixel row (npx is a multiple of 8: whole pieces)
		auto pattern = [&](int piece) { // the background colour as it falls on piece `piece` of a row
			VE v;
			int ph = (C == 3 || C > CH) ? (piece * E) % C : ((piece * E) & (C - 1)); // channel of the piece's first element
#pragma unroll
			for (int j = 0; j < E; j++)
			{
				v[j] = C > CH ? ((const PixT *)p.bg_color)[ph] : (PixT)(ph == 0 ? bgc[0] : (ph == 1 ? bgc[1] : (ph == 2 ? bgc[2] : bgc[3])));
				ph = ph + 1 == C ? 0 : ph + 1;
			}
			return v;
		};
		if (n >= 64 && !bgi && C <= CH)
		{ // the usual long run of a colour background: a lane's pieces lane, lane + 64, ... of a row see the pattern with period 3
		  // (period 1 unless C = 3), so the three vectors are formed once and the loop is a store and a pointer increment
			const VE v0 = pattern(lane), v1 = pattern(lane + 64), v2 = pattern(lane + 128);
			for (int row = 0; row < rows; row++)
			{
				PixT *out = img + (size_t)row * row_stride + (size_t)lane * E;
				int piece = lane;
				for (; piece + 128 < n; piece += 192, out += 192 * E)
				{
					__builtin_nontemporal_store(v0, (VE *)out);
					__builtin_nontemporal_store(v1, (VE *)(out + 64 * E));
					__builtin_nontemporal_store(v2, (VE *)(out + 128 * E));
				}
				if (piece < n)
					__builtin_nontemporal_store(v0, (VE *)out);
				if (piece + 64 < n)
					__builtin_nontemporal_store(v1, (VE *)(out + 64 * E));
			}
		}
		else
			for_pieces(n, [&](int row, int piece) {
				const size_t at = (size_t)row * row_stride + (size_t)piece * E;
				__builtin_nontemporal_store(bgi ? *(const VE *)(bgi + at) : pattern(piece), (VE *)(img + at));
			});
	}
	if (p.zbuf)
	{
		VE inf;
#pragma unroll
		for (int j = 0; j < E; j++)
			inf[j] = (PixT)INFINITY;
		PixT *zb = (PixT *)p.zbuf + (size_t)view * H * W + (size_t)y0 * W + x0;
		for_pieces(npx / E, [&](int row, int piece) { __builtin_nontemporal_store(inf, (VE *)(zb + (size_t)row * W + piece * E)); });
	}
	if (owners)
	{
		const I4 none = {-1, -1, -1, -1};
		int32_t *own = face_id + (size_t)y0 * W + x0;
		for_pieces(npx / 4, [&](int row, int piece) { __builtin_nontemporal_store(none, (I4 *)(own + (size_t)row * W + piece * 4)); });
	}
}

template <class PixT>
__device__ __forceinline__ void fill_word(const KParams &p, int view, int wi, int lane, int owners)
{ // background of the empty tiles of bitmap word wi of the view (one wavefront)
	const ViewPtrs w = view_ptrs(p, view);
	const int base = wi * 32, valid = p.L.ntiles - base < 32 ? p.L.ntiles - base : 32;
	uint32_t empty = ~w.tile_bits[wi] & (valid == 32 ? 0xffffffffu : (1u << valid) - 1u);
	empty = (uint32_t)uniform((int)empty);
	if (!empty)
		return;
	const int C = p.C;
	if (p.aa_err && p.err)
	{ // antialiase_error: the error buffer of a background pixel, sum_c (background - obs)^2 (H.h:2824-2837), tile by tile (lane = pixel)
		for (uint32_t rest = empty; rest; rest &= rest - 1)
		{
			const int t = base + __ffs((int)rest) - 1, ty = t / p.L.tiles_x, tx = t - ty * p.L.tiles_x;
			const int px = tx * TILE + (lane & 7), py = ty * TILE + (lane >> 3);
			if (px < p.W && py < p.H)
			{
				const size_t pix = (size_t)py * p.W + px, vpix = (size_t)view * p.H * p.W + pix;
				double e = 0;
				for (int c = 0; c < C; c++)
				{
					const double d = background_channel<PixT>(p, view, pix, c) - (double)((const PixT *)p.obs)[vpix * C + c];
					e += d * d;
				}
				__builtin_nontemporal_store((PixT)e, (PixT *)p.err + vpix);
			}
		}
	}
	if (C > CH && (p.W & 7) != 0)
	{ // more than CH channels (the staged forward of a many-channel frame, fwd_manyc_tile) AND a ragged frame width: pixel by pixel, channel by
	  // channel (the usual width goes through fill_run below, whose 16-byte pieces do not care about the channel count)
		for (uint32_t rest = empty; rest; rest &= rest - 1)
		{
			const int t = base + __ffs((int)rest) - 1, ty = t / p.L.tiles_x, tx = t - ty * p.L.tiles_x;
			const int px = tx * TILE + (lane & 7), py = ty * TILE + (lane >> 3);
			if (px >= p.W || py >= p.H)
				continue;
			const size_t pix = (size_t)py * p.W + px, vpix = (size_t)view * p.H * p.W + pix;
			if (p.image)
				for (int c = 0; c < C; c++)
					__builtin_nontemporal_store((PixT)background_channel<PixT>(p, view, pix, c), (PixT *)p.image + vpix * C + c);
			if (p.zbuf)
				__builtin_nontemporal_store((PixT)INFINITY, (PixT *)p.zbuf + vpix);
			if (owners)
				__builtin_nontemporal_store((int32_t)-1, w.face_id + pix);
		}
		return;
	}
	double bgc[CH] = {0, 0, 0, 0};
	if (!p.bg_image)
	{
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
			if (cc < C)
				bgc[cc] = (double)((const PixT *)p.bg_color)[cc];
	}
	if ((p.W & 7) == 0)
	{ // maximal runs of empty tiles inside one tile row
		while (empty)
		{
			const int a = __ffs((int)empty) - 1;
			const uint32_t rest = ~(empty >> a);			   // bit i clear: tile a + i is empty
			int len = rest ? __ffs((int)rest) - 1 : 32 - a; // (all ones above a: the run goes to the end of the word)
			const int t0 = base + a, ty = t0 / p.L.tiles_x, tx = t0 - ty * p.L.tiles_x;
			if (tx + len > p.L.tiles_x)
				len = p.L.tiles_x - tx; // the rest of the run lies in the next tile row
			fill_run<PixT>(p, view, w.face_id, ty, tx, tx + len, lane, bgc, owners);
			empty &= len >= 32 ? 0u : ~(((1u << len) - 1u) << a);
		}
		return;
	}
	for (int i = 0; i < 32; i++) // ragged frame width: tile by tile
		if ((empty >> i) & 1u)
			fill_background_tile<PixT>(p, view, w.face_id, (base + i) % p.L.tiles_x, (base + i) / p.L.tiles_x, lane, bgc, owners);
}

template <class PixT>
__global__ __launch_bounds__(64 * FILL_WAVES) void fill_kernel(KParams p, int owners)
{
	const int gw = blockIdx.x * FILL_WAVES + (threadIdx.x >> 6);
	if (gw >= p.n_views * p.L.nwords)
		return;
	fill_word<PixT>(p, gw / p.L.nwords, gw % p.L.nwords, threadIdx.x & 63, owners);
}

// The background fill of a fit step rides on the step's other kernels as extra workgroups instead of being a kernel of its own on a
// forked stream -- the fork / join event packets cost the caller's stream two bubbles of ~7 us per step (rocprofv3 kernel trace: scan
// -> forward, finalize -> next set-up).  Which kernels take part is p.fill_mode (bit 0: raster_bwd_edge_kernel, bit 1: finalize_kernel,
// bit 2: the staged forward raster itself); the bitmap words of a view are dealt to them in proportion to FILL_W.  The fill is
// 110 MB of stores per 8-view step at the ~3 TB/s the store path sustains, i.e. ~37 us of bandwidth time wherever it goes: only the
// forward raster (issue-bound, ~95 us, 0.6 TB/s of stores of its own) is long enough to hide most of it; finalize_kernel (bound by
// the latency of its round trips) hides a part (all of it there: + 11 us).
#ifndef DR_FILL_W_EDGE
#define DR_FILL_W_EDGE 1
#endif
#ifndef DR_FILL_W_FIN
#define DR_FILL_W_FIN 1
#endif
#ifndef DR_FILL_W_FWD
#define DR_FILL_W_FWD 2
#endif
__host__ __device__ inline int fill_weight(int bit) { return bit == 0 ? DR_FILL_W_EDGE : (bit == 1 ? DR_FILL_W_FIN : DR_FILL_W_FWD); }
__host__ __device__ inline void fill_split(int fill_mode, int bit, int &den, int &off)
{ // of every `den` consecutive bitmap words, the fill_weight(bit) words starting at `off` are kernel `bit`'s
	den = off = 0;
	for (int k = 0; k < 3; k++)
		if (fill_mode & (1 << k))
		{
			if (k < bit)
				off += fill_weight(k);
			den += fill_weight(k);
		}
}
__host__ __device__ inline int fill_share(int fill_mode, int bit, int nwords)
{ // bitmap words per view the kernel `bit` fills
	if (!(fill_mode & (1 << bit)))
		return 0;
	int den, off;
	fill_split(fill_mode, bit, den, off);
	const int full = nwords / den, rest = nwords - full * den, w = fill_weight(bit);
	const int extra = rest - off < 0 ? 0 : (rest - off > w ? w : rest - off);
	return full * w + extra;
}
// workgroups (edge kernel: per view, along grid y, limited to 65535) that stream a share of n words: one word each up to a cap,
// beyond it (frames of more than ~4 M tiles) every workgroup takes several
__host__ __device__ inline int fill_share_blocks(int n) { return n < 32768 ? n : 32768; }
__device__ __forceinline__ void fill_share_word(const KParams &p, int bit, int view, int i, int lane)
{ // the i-th word of the share of kernel `bit`
	int den, off;
	fill_split(p.fill_mode, bit, den, off);
	const int w = fill_weight(bit);
	const int wi = (i / w) * den + off + i % w;
	if (wi >= p.L.nwords)
		return;
	if (p.pix_f64)
		fill_word<double>(p, view, wi, lane, 0);
	else
		fill_word<float>(p, view, wi, lane, 0);
}

// Grid of the staged forward (1-D, one wavefront per workgroup).  Workgroup b: view (b / 8) % n_views,
// q = (b / 8 / n_views) * 8 + b % 8 in [0, p.tile_blocks); it walks the entries rank(q), rank(q) + tile_blocks, ... of the
// view's work list (usually one or two).  rank() deals the list to the XCDs in chunks of 64 consecutive entries (workgroup b
// runs on XCD b % 8; consecutive entries are neighbouring tiles, which share triangle records and should share an L2).

// x / n_views without the division (KParams::views_magic = floor(2^32 / n) + 1, or 2^32 - 1 for one view): the estimate mulhi(x, magic) is off by at
// most one either way (n >= 2: exact or one too large, as magic * n exceeds 2^32 by at most n; n = 1: one too small), two compares settle it
__device__ __forceinline__ uint32_t div_views(const KParams &p, uint32_t x)
{
	const uint32_t n = (uint32_t)p.n_views;
	uint32_t qv = __umulhi(x, p.views_magic);
	qv -= qv * n > x ? 1u : 0u;
	qv += (qv + 1u) * n <= x ? 1u : 0u;
	return qv;
}

__host__ __device__ inline int fwd_tile_blocks(int ntiles, int n_views, bool dealt_fill)
{ // workgroups per view that walk the work list: a quarter of the tiles (about a third of a frame's tiles hold primitives, and most
  // of those pair up: about one entry per walker) -- a sixth from 8 views up in a fit step of an untextured scene (dealt_fill): with the
  // fill workgroups dealt among the walkers, fewer
  // and longer walkers win there (same-box A/B, 8 views of the benchmark scene: /4 0.1242 - 0.1252, /5 0.1208 - 0.1214, /6 0.1211 - 0.1213,
  // /8 0.1271 - 0.1280 ms; 8 views of the hand 0.103 -> 0.096; 1 / 2 / 4 views lose 1 - 3 % at /6, 16 views are level: profiles/r04n)
	const int unit = 8 * WORK_CHUNK;
#ifndef DR_TILE_DIV
#define DR_TILE_DIV 0 // (measurement builds: a fixed divisor)
#endif
	const int div = DR_TILE_DIV ? DR_TILE_DIV : ((dealt_fill && n_views >= 8) ? 6 : 4);
	const int g = ((ntiles / div + unit - 1) / unit) * unit;
	return g > 0 && g <= ntiles ? g : ntiles; // tiny frames: one workgroup per tile, plain order
}

// FUSED: the forward of a fit step.  The loss is L = sum (image - obs)^2, so dL/dimage is known the moment a pixel is
// resolved: tiles without silhouette edges back-propagate into their owners' accumulators right here (no second pass over the
// frame, no owner buffer round trip -- the owner ids of those tiles are not even written); tiles with edges are left to
// raster_bwd_edge_kernel.
// Waves per SIMD the staged forward is compiled for: without texture code it fits five (96 registers), with it four -- three (168
// registers) for the textured instances whose walkers also back-propagate the tiles with silhouette edges (TEXE: the reverse sweep with
// texture taps; at four waves it spills 414 registers.  configs[4], 1 / 8 views: four waves 0.176 / 0.886 - 0.930 ms, three 0.168 / 0.868 -
// 0.893, two 0.167 / 1.02; the instances WITHOUT the edge adjoint lose 6 - 9 % at three: profiles/r05y_ab_fused_textured_edge_tiles.txt)
// (tools/build_variants.sh builds the neighbours: -DDR_FWD_WAVES=n forces n for all).
#ifndef DR_TEXE_HEAD_WAVES
#define DR_TEXE_HEAD_WAVES 3 // waves per SIMD of the kernel of the head walkers alone (TEXE = 2)
#endif
#ifndef DR_FWD_WAVES
#define DR_FWD_WAVES (TEX ? (TEXE == 1 ? 3 : (TEXE == 2 ? DR_TEXE_HEAD_WAVES : 4)) : 5)
#endif
// Two horizontally adjacent tiles in one wavefront, two pixels per lane (lane = row * 8 + column: pixel `column` of the left tile A
// and pixel `column` of the right tile B).  For the pairs the scan kernel forms -- both tiles non-empty, no silhouette edge, at
// most ENTRY_IDS triangles together, untextured scene, fit step -- the per-TILE costs of the walker are paid once for 128 pixels:
// the work entry and the batched gather of records and planes (one memory round trip), the scanline spans (lane = slot * 8 + row:
// twelve slots still fit the two passes a single tile makes) and the exchange of the column masks.  tools/fwd_trace.py: of the
// 12.8 k cycles of a tile without edges 1.5 k are the prologue, 2.6 k the staging, 2.3 k the spans; 84 % of those tiles of the
// benchmark scene pair up.  A slot belongs to ONE of the two tiles (slots 0 .. nA - 1 to A, the others to B: binning is exact, a
// triangle listed only in A covers no pixel of B; a triangle listed in both has two slots), so coverage, depth test -- min (Z,
// index) per pixel -- and therefore every result are those of the two tiles walked one after the other.
// TEX (round 6): textured scenes pair up too -- configs[4]'s tiles hold 5.9 triangles on average (median 4), and of an edge-free textured tile's 20 k
// cycles 7.3 k are the prologue and pass 1 that a pair pays once; the winner's texels are fetched per pixel, and the adjoint of pass 1 runs twice
// (owner_adjoint with the texture-gradient window, tile A then tile B).
template <class PixT, bool CLAMP, bool TEX = false>
__device__ __forceinline__ void fwd_pair_tiles(const KParams &p, const ViewPtrs &w, WaveLds &S, int view, int lane, int tile, int nA, int nB, uint32_t my_id,
											   double *loss_at)
{ // loss_at (or NULL): this walker's partial of the loss, see tile_loss
	const int W = p.W, H = p.H, C = p.C, P = p.L.P;
	const bool strict = p.strict;
	const int tx = tile % p.L.tiles_x, ty = tile / p.L.tiles_x;
	const int x0 = tx * TILE, y0 = ty * TILE;
	const int lx = lane & 7, row = lane >> 3;
	const int py = y0 + row, pxA = x0 + lx, pxB = x0 + TILE + lx;
	const bool inbA = p.aligned || (pxA < W && py < H), inbB = p.aligned || (pxB < W && py < H);
	const size_t pixA = (size_t)py * W + pxA, pixB = pixA + TILE;
	const size_t vbase = (size_t)view * H * W;
	const double y = py, xA = pxA, xB = pxB;
	const int nb = nA + nB;
	// observation of both pixels: requested now, used after the depth test
	PixT obA[CH] = {0, 0, 0, 0}, obB[CH] = {0, 0, 0, 0};
	if (inbA)
	{
		const PixT *o = (const PixT *)p.obs + (vbase + pixA) * C;
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
			if (cc < C)
				obA[cc] = o[cc];
	}
	if (inbB)
	{
		const PixT *o = (const PixT *)p.obs + (vbase + pixB) * C;
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
			if (cc < C)
				obB[cc] = o[cc];
	}
	if (lane < nb)
		S.ids[lane] = my_id;
	lds_sync();
	stage_batch(S, w.tri_rec, w.tri_planes, P, nb, lane);
	lds_sync();
	// spans: lane = slot * 8 + row, against the columns of the slot's own tile
#pragma unroll
	for (int q = 0; q < TB / 8; q++)
	{
		const int j = q * 8 + (lane >> 3), r = lane & 7;
		uint32_t m = 0;
		if (j < nb)
		{
			const TriRec &rec = S.rec[j].tri();
			if (rec.kind != KIND_NONE)
			{
				const int yy = y0 + r, xs = j < nA ? x0 : x0 + TILE;
				const bool in0 = yy >= rec.y_begin[0] && yy <= rec.y_end[0], in1 = yy >= rec.y_begin[1] && yy <= rec.y_end[1];
				int xb, xe;
				tri_half_span(rec, in0 ? 0 : 1, yy, W, H, strict, xb, xe);
				m = column_mask(xb, xe, xs);
				if (__ballot(in0 && in1))
				{
					if (in0 && in1)
					{ // (only the non-strict fill rule puts the middle-vertex row in both halves)
						tri_half_span(rec, 1, yy, W, H, strict, xb, xe);
						m |= column_mask(xb, xe, xs);
					}
				}
			}
		}
		S.cover[r][j] = (uint8_t)m;
	}
	lds_sync();
	const uint32_t cov = gather_column_bits(&S.cover[row][0], lx) & ((1u << nb) - 1u), maskA = (1u << nA) - 1u;
	uint32_t todoA = inbA ? cov & maskA : 0u, todoB = inbB ? cov & ~maskA : 0u;
	// depth test: each lane walks the triangles that cover its two pixels; winner = min (Z, index) (H.h:961 in index order)
	double zA = INFINITY, zB = INFINITY;
	int kA = -1, kB = -1, jA = -1, jB = -1;
	while (__ballot((todoA | todoB) != 0))
	{
		{
			const bool act = todoA != 0;
			const int j = act ? __ffs((int)todoA) - 1 : 0;
			todoA &= todoA - 1;
			const double Z = plane_at(S.rec[j].tri().xZ, xA, y);
			const int k = (int)S.ids[j];
			if (act && (Z < zA || (Z == zA && k < kA)))
				zA = Z, kA = k, jA = j;
		}
		{
			const bool act = todoB != 0;
			const int j = act ? __ffs((int)todoB) - 1 : nA;
			todoB &= todoB - 1;
			const double Z = plane_at(S.rec[j].tri().xZ, xB, y);
			const int k = (int)S.ids[j];
			if (act && (Z < zB || (Z == zB && k < kB)))
				zB = Z, kB = k, jB = j;
		}
	}
	// colours (perspective_correct excludes the adjoint, hence a fit step)
	double colA[CH] = {0, 0, 0, 0}, colB[CH] = {0, 0, 0, 0};
	int kindA = KIND_NONE, kindB = KIND_NONE;
	Tap tapA, tapB;
	double LA = 0, LB = 0;
	const PixT *texture = (const PixT *)p.texture;
	// the winner's colour: interpolated, or (TEX) bilinear texture x shade (H.h:1159-1258) from the staged (u, v, shade) planes
	auto shade_winner = [&](int j, double xx, int &kind, Tap &tap, double &L, double (&col)[CH]) {
		kind = S.rec[j].tri().kind;
		const double *pl = &S.planes[j * 12];
		if (TEX && kind == KIND_TEXTURED)
		{
			bilinear_tap(p.tex_w, p.tex_h, plane_at(pl, xx, y), plane_at(pl + 3, xx, y), C, tap);
			L = plane_at(pl + 6, xx, y);
			PixT tx[4][4];
			tap_texels(texture, tap, C, tx);
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				col[cc] = cc < C ? bilinear_mix(tap, (double)tx[0][cc], (double)tx[1][cc], (double)tx[2][cc], (double)tx[3][cc]) * L : 0.0;
		}
		else
		{
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				if (cc < C)
					col[cc] = interp_channel(pl, cc, xx, y, false, 0.0);
		}
	};
	if (jA >= 0)
		shade_winner(jA, xA, kindA, tapA, LA, colA);
	else if (inbA)
	{
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
			if (cc < C)
				colA[cc] = background_channel<PixT>(p, view, pixA, cc);
	}
	if (jB >= 0)
		shade_winner(jB, xB, kindB, tapB, LB, colB);
	else if (inbB)
	{
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
			if (cc < C)
				colB[cc] = background_channel<PixT>(p, view, pixB, cc);
	}
	// one write per pixel (streaming stores)
	auto store_pixel = [&](bool inb, size_t pix, const double *col, double z) {
		if (!inb)
			return;
		if (p.image)
		{
			PixT *out = (PixT *)p.image + (vbase + pix) * C;
			if (C == 4)
			{
				typedef PixT V4 __attribute__((ext_vector_type(4)));
				const V4 v = {(PixT)col[0], (PixT)col[1], (PixT)col[2], (PixT)col[3]};
				__builtin_nontemporal_store(v, (V4 *)out);
			}
			else
			{
#pragma unroll
				for (int cc = 0; cc < CH; cc++)
					if (cc < C)
						__builtin_nontemporal_store((PixT)col[cc], out + cc);
			}
		}
		if (p.zbuf)
			__builtin_nontemporal_store((PixT)z, (PixT *)p.zbuf + vbase + pix);
	};
	store_pixel(inbA, pixA, colA, zA);
	store_pixel(inbB, pixB, colB, zB);
	if (loss_at)
	{
		double r2 = 0;
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
			if (cc < C)
			{
				const double dA = inbA ? fit_value<CLAMP>(p, (double)(PixT)colA[cc]) - (double)obA[cc] : 0.0,
							 dB = inbB ? fit_value<CLAMP>(p, (double)(PixT)colB[cc]) - (double)obB[cc] : 0.0;
				r2 += dA * dA + dB * dB;
			}
		r2 = wave_sum(r2);
		if (lane == 0)
			atomic_add_f64(loss_at, r2 - (p.loss_tile_bg[1 + (size_t)view * p.L.ntiles + tile] + p.loss_tile_bg[2 + (size_t)view * p.L.ntiles + tile]));
	}
	// adjoint of pass 1 for L = sum (image - obs)^2: the colour is rounded to the pixel type first, like the stored frame
	if constexpr (sizeof(PixT) == 4 && !TEX)
	{ // float32 frame: both tiles through one slot table (owner_adjoint_slots, dr_backward.h)
		if (__ballot(jA >= 0 || jB >= 0) != 0)
		{
			const int slot[2] = {jA, jB};
			float gs[2][CH];
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
			{
				gs[0][cc] = (cc < C && inbA && jA >= 0) ? fit_residual_f32<CLAMP>(p, (float)colA[cc], (float)obA[cc]) : 0.0f;
				gs[1][cc] = (cc < C && inbB && jB >= 0) ? fit_residual_f32<CLAMP>(p, (float)colB[cc], (float)obB[cc]) : 0.0f;
			}
			owner_adjoint_slots<2>(p, w, lane, x0, y0, slot, gs, my_id, nb, (float *)&S.rec[0]);
		}
		return;
	}
	// (a tile without edges: the whole LDS of the wavefront -- staging area and edge order -- is the texture-gradient window, as in the single-tile path)
	constexpr int WIN = TEX ? (int)((sizeof(WaveLds) + sizeof(EdgeSort)) / sizeof(double)) : 384;
	double g[CH];
	if (__ballot(kA >= 0) != 0)
	{
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
			g[cc] = (cc < C && inbA) ? fit_residual<CLAMP>(p, (double)(PixT)colA[cc], (double)obA[cc]) : 0.0;
		lds_sync();
		owner_adjoint<PixT, TEX>(p, w, lane, xA, y, kA, kA >= 0 ? kindA : (int)KIND_NONE, g, tapA, LA, (double *)&S.rec[0], (uint32_t *)&S.cover[0][0], WIN);
	}
	if (__ballot(kB >= 0) != 0)
	{
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
			g[cc] = (cc < C && inbB) ? fit_residual<CLAMP>(p, (double)(PixT)colB[cc], (double)obB[cc]) : 0.0;
		lds_sync();
		owner_adjoint<PixT, TEX>(p, w, lane, xB, y, kB, kB >= 0 ? kindB : (int)KIND_NONE, g, tapB, LB, (double *)&S.rec[0], (uint32_t *)&S.cover[0][0], WIN);
	}
}

// More than CH channels, forward only, no silhouette edge anywhere (sigma = 0), no texture: the frame of Scene3D.render_deferred (dr.py:1053-1174:
// depth, face ids, barycentrics, normals, luminosity, xyz, colours or uv -- 15 channels of a triangle soup in ONE render, sigma = 0 by its own assert).
// Pass 1 as everywhere (staged records, exact spans, winner = min (Z, index)); the planes of a triangle are 3 C doubles, too many to stage, and only the
// WINNER's are needed: each lane reads them from memory once the tile is resolved (lanes with the same winner read the same lines).  The un-staged
// kernel walked the tile's list once per chunk of four channels, a dependent record load per triangle each time: 133 -> 37 us of forward raster for one 1024^2 view
// (profiles/r06l_slow_family_after.txt).
template <class PixT>
__device__ __forceinline__ void fwd_manyc_tile(const KParams &p, const ViewPtrs &w, WaveLds &S, int view, int lane, int tile, int ntri, uint32_t ids12)
{
	const int W = p.W, H = p.H, C = p.C, P = p.L.P;
	const bool persp = p.persp;
	const int tx = tile % p.L.tiles_x, ty = tile / p.L.tiles_x;
	const int x0 = tx * TILE, y0 = ty * TILE;
	const int px = x0 + (lane & 7), py = y0 + (lane >> 3);
	const bool inb = px < W && py < H;
	const size_t pix = (size_t)py * W + px;
	const size_t vpix = (size_t)view * H * W + pix;
	const double x = px, y = py;
	const uint32_t list_entry = ntri <= ENTRY_IDS ? ids12 : w.tri_list[(size_t)tile * K_TRI + (lane & (K_TRI - 1))];
	PixState st;
	st.zbest = INFINITY;
	st.kbest = -1;
	st.kind = KIND_NONE;
	st.slot = 0;
	const int n_inline = ntri < K_TRI ? ntri : K_TRI;
	for (int base = 0; base < n_inline; base += TB)
	{
		const int nb = n_inline - base < TB ? n_inline - base : TB;
		if (lane >= base && lane < base + nb)
			S.ids[lane - base] = list_entry;
		lds_sync();
		stage_batch<TriRec, false>(S, w.tri_rec, w.tri_planes, P, nb, lane);
		lds_sync();
		tri_batch<false, false>(p, S, nb, lane, x0, y0, inb, st);
	}
	if (ntri > K_TRI)
	{ // spilled pairs of this tile: compact them out of the pool, TB at a time
		uint32_t spill_n = w.hdr->tri_spill[w.hdr->cur];
		if (spill_n > p.L.tri_pool_cap)
			spill_n = p.L.tri_pool_cap;
		int fill = 0;
		for (uint32_t i0 = 0; i0 < spill_n; i0 += 64)
		{
			const uint2 pr = (i0 + lane < spill_n) ? w.tri_pool[i0 + lane] : make_uint2(0xffffffffu, 0u);
			unsigned long long m = __ballot((int)pr.x == tile);
			while (m)
			{
				const int room = TB - fill, cnt = __popcll(m);
				const int rank = __popcll(m & ((1ull << lane) - 1ull));
				const bool sel = ((m >> lane) & 1ull) && rank < room;
				if (sel)
					S.ids[fill + rank] = pr.y;
				m &= ~__ballot(sel);
				fill += cnt < room ? cnt : room;
				if (fill == TB)
				{
					lds_sync();
					stage_batch<TriRec, false>(S, w.tri_rec, w.tri_planes, P, TB, lane);
					lds_sync();
					tri_batch<false, false>(p, S, TB, lane, x0, y0, inb, st);
					fill = 0;
				}
			}
		}
		if (fill > 0)
		{
			lds_sync();
			stage_batch<TriRec, false>(S, w.tri_rec, w.tri_planes, P, fill, lane);
			lds_sync();
			tri_batch<false, false>(p, S, fill, lane, x0, y0, inb, st);
		}
	}
	if (p.image)
	{
		const double *planes = w.tri_planes + (size_t)(st.kbest < 0 ? 0 : st.kbest) * 3 * P;
		const bool drawn = st.kbest >= 0 && st.kind == KIND_INTERP;
		// A pixel's C values are C consecutive elements of the frame, a tile row 8 C of them: written by the pixel's lane they are 4-byte stores
		// 4 C bytes apart (15 channels: every store instruction touches 64 partial lines).  Through LDS instead -- the wavefront's whole
		// allotment, staging area and edge order, idle by now -- and out as 16-byte pieces of contiguous tile rows, as many rows per pass as fit.
		// (A frame whose width is not a multiple of the tile keeps the direct stores: its last tile column has shorter rows.)
		constexpr int LDS_ELEMS = (int)((sizeof(WaveLds) + sizeof(EdgeSort)) / sizeof(PixT)), E = 16 / (int)sizeof(PixT);
		const int row_elems = TILE * C;
		const bool via_lds = (W & (TILE - 1)) == 0 && row_elems <= LDS_ELEMS;
		const int rows_pp = via_lds ? (LDS_ELEMS / row_elems < TILE ? LDS_ELEMS / row_elems : TILE) : TILE;
		PixT *const stage = (PixT *)&S;
		for (int r0 = 0; r0 < TILE; r0 += rows_pp)
		{
			const int my_row = (lane >> 3) - r0;
			const bool mine = inb && my_row >= 0 && my_row < rows_pp;
			if (via_lds)
				lds_sync();
			if (mine)
			{
				PixT *out = via_lds ? stage + (my_row * TILE + (lane & 7)) * C : (PixT *)p.image + vpix * C;
				for (int c0 = 0; c0 < C; c0 += CH)
				{ // four channels at a time: their twelve plane coefficients are requested together
					double v[CH];
#pragma unroll
					for (int j = 0; j < CH; j++)
						v[j] = c0 + j < C ? (drawn ? interp_channel(planes, c0 + j, x, y, persp, st.zbest) : background_channel<PixT>(p, view, pix, c0 + j)) : 0.0;
#pragma unroll
					for (int j = 0; j < CH; j++)
						if (c0 + j < C)
						{
							if (via_lds)
								out[c0 + j] = (PixT)v[j];
							else
								__builtin_nontemporal_store((PixT)v[j], out + c0 + j);
						}
				}
			}
			if (via_lds)
			{
				lds_sync();
				typedef PixT VE __attribute__((ext_vector_type(E)));
				const int ppr = row_elems / E; // 16-byte pieces per tile row (8 C elements: a whole number of pieces)
				for (int i = lane; i < rows_pp * ppr; i += 64)
				{
					const int row = i / ppr, piece = i - row * ppr, yy = y0 + r0 + row;
					if (r0 + row < TILE && yy < H)
						__builtin_nontemporal_store(*(const VE *)(stage + row * row_elems + piece * E),
													(VE *)((PixT *)p.image + ((size_t)view * H * W + (size_t)yy * W + x0) * C + piece * E));
				}
			}
		}
		if (via_lds)
			lds_sync(); // (the next tile of this wavefront stages its records here)
	}
	if (!inb)
		return;
	if (p.zbuf)
		__builtin_nontemporal_store((PixT)st.zbest, (PixT *)p.zbuf + vpix);
	__builtin_nontemporal_store(pack_owner(st.kbest, st.kind), w.face_id + pix);
}

// Code-generation modes of the tile walker.  A fit step (FUSED) back-propagates EVERY tile in the forward launch: the scan kernel puts
// the tiles that hold silhouette edges at the head of the work list, and the workgroups that walk the head run the instance that
// can do their adjoint too (FWD_EDGE_ADJ: reverse sweep over the tile's edges, then pass 1) while all the other workgroups run an
// instance without any edge code (FWD_NO_EDGES).  Two instances in one kernel, chosen per workgroup, instead of a branch inside one
// loop: the edge adjoint needs ~40 registers more than the rest, and as a branch of the common loop (or as a function called from
// it) it cost EVERY tile spills on its path (forward 83 -> 90 us before the first edge tile was fused); as a disjoint path its
// spills stay with the one workgroup in sixteen that walks the head.
#ifndef DR_ONE_BATCH_BODY
#define DR_ONE_BATCH_BODY 1 // (measurement builds: 0 = one body for every head tile, as until the end of round 5)
#endif
enum FwdMode
{
	FWD_PLAIN = 0,	  // forward only (or adjoint left to the two-call path): pass 2 saves its sweep for raster_bwd_edge_kernel
	FWD_EDGE_ADJ = 1, // fit step, head of the list: tiles with edges are back-propagated here as well
	FWD_NO_EDGES = 2, // fit step, rest of the list: no tile has an edge
};
// AA (round 6): antialiase_error -- the image stays un-antialiased, err_buffer = sum_c (image - obs)^2 (H.h:2824-2837) is what the edges blend
// (rasterize_edge_*_error, H.h:2067-2197, 2371-2478).  Forward-only instances (FUSED = false: the adjoint of this mode is the two-call path's).
// TEXPAIR: (TEX) the instance also walks PAIRS of textured tiles (fwd_pair_tiles<…, true>): the one-kernel form of a textured fit step only -- compiled into
// the two kernels of the 8-view form as well (where the scan kernel forms no textured pair) the code alone cost them 13 % (128 registers + spills instead of 101)
template <class PixT, bool FUSED, bool TEX, int MODE, bool CLAMP, bool AA = false, bool MANYC = false, bool TEXPAIR = false>
__device__ __forceinline__ void fwd_tiles(const KParams &p, WaveLds *s_lds, EdgeSort *s_es, const uint32_t b)
{ // b: index of this walker among the walkers of the grid (the workgroup index, unless fill workgroups are dealt among them).
  // (32-bit: a grid has fewer than 2^31 workgroups, and every wavefront pays for this arithmetic on the scalar unit before its first load --
  // as `long long` the two divisions by the number of views alone were ~250 instructions of 64-bit division emulation)
	DR_WAVE_TRACE_SCOPE(2);
	constexpr int wave = 0;
	const int lane0 = wave_lane(); // (= threadIdx.x: one-wave workgroups; see wave_lane)
	const int G = p.tile_blocks;
	int view, q;
	const bool chunked = G % (8 * WORK_CHUNK) == 0;
	if (chunked)
	{
		const uint32_t g8 = b >> 3, gq = div_views(p, g8);
		view = (int)(g8 - gq * (uint32_t)p.n_views);
		q = (int)(gq * 8 + (b & 7));
	}
	else
	{
		const uint32_t gq = div_views(p, b);
		view = (int)(b - gq * (uint32_t)p.n_views);
		q = (int)gq;
	}
	const int W = p.W, H = p.H, C = p.C, P = p.L.P;
	const bool persp = p.persp;
	const PixT *texture = (const PixT *)p.texture;
	WaveLds &S = s_lds[wave];
	int grp = 0; // (persistent walkers) the ticket group of `view` this walker draws from: its own XCD's, until that one runs dry
	for (int hop = 0;; hop++)
	{ // (one pass, unless the walker is a persistent one that moves on to another group's entries: below)
	const ViewPtrs w = view_ptrs(p, view);
	// The first G / p.heavy_share workgroups of a view walk the many-primitive tiles (front of the list), the others the rest (from
	// the back): the index of a workgroup's entry does not depend on the counts, so the counts, the entry header and the
	// entry's triangle ids are all requested at once.
	// (tiny frames -- G not a multiple of 512 -- have one class only: the scan kernel lists every tile as "other")
	const int Gh = chunked ? (int)p.fwd_heads : 0; // (= G / p.heavy_share, from the host)
	const bool heavy_list = q < Gh;
	const int qq = heavy_list ? q : q - Gh, stride = heavy_list ? Gh : G - Gh;
	const bool chunk_here = chunked && stride % (8 * WORK_CHUNK) == 0;
	uint32_t rank = chunk_here ? (uint32_t)((((qq >> 3) / WORK_CHUNK) * 8 + (qq & 7)) * WORK_CHUNK + (qq >> 3) % WORK_CHUNK) : (uint32_t)qq;
	// Persistent walkers (KParams::dyn_groups, the others' list of a many-view fit step): far fewer workgroups than entries, all resident, each taking
	// its entries by TICKET -- a wave slot then never waits for the dispatcher between two tiles (tools/wave_trace.py --slots: ~1 us from the end of a
	// one-tile workgroup to the start of the next one on its SIMD, 27 000 times per 8-view step).  One counter per (view, XCD): ticket t of group g is
	// entry (t / 8) * 64 + g * 8 + t % 8 -- runs of eight neighbouring entries stay on one XCD's L2.  The ticket of the entry after next is requested
	// (one lane, returning atomic) when a tile starts, the next entry itself a tile ahead as before: no round trip is waited for between tiles.
	const bool dyn = DR_DYN_WALKERS && MODE == FWD_NO_EDGES && p.dyn_groups > 0;
	if (hop == 0)
		grp = qq & (DYN_GROUPS - 1);
	uint32_t *const tick = &w.edge_tile_cnt[(EDGE_LISTS + 1 + grp) * CNT_STRIDE];
	auto ticket_rank = [&](uint32_t t) { return (t >> 3) * (8u * DYN_GROUPS) + (uint32_t)grp * 8u + (t & 7u); };
	uint32_t tk = 0; // (lane 0) the ticket of the entry after the one whose header is in `head`
	if (dyn)
	{
		if (lane0 == 0)
			tk = atomicAdd(tick, 2u);
	}
	// The first entry of a walker (usually its only one) is requested TOGETHER with the count it is checked against -- the position of the
	// entry does not depend on the count and lies inside the list whatever the count is (rank < stride <= tiles <= work_cap).  As the first
	// statement of the loop body the load sat behind the branch on the count: a third dependent round trip (count, entry, records) in the
	// ~8 us life of a wavefront whose arithmetic is ~2 us.
	auto entry_at = [&](uint32_t r) { return &w.work_list[heavy_list ? r : (uint32_t)p.L.work_cap - 1u - r]; };
	// (The compiler sinks loads that only the loop body uses below the branch on the count, whatever their place in the source: the empty
	// asm statement takes the three results as read-write operands, so all three loads are issued, and waited for ONCE, in front of it.)
	uint32_t n_work_v = w.hdr->work_count[heavy_list ? 0 : 1];
	if (dyn)
	{ // (the first two tickets in one request; the count travels with it)
		const uint32_t t0 = (uint32_t)uniform((int)tk);
		rank = ticket_rank(t0);
		tk = t0 + 1u;
	}
	uint4 head = *(const uint4 *)entry_at(rank); // {tile, ntri, nedge, sweep_slot}
	uint32_t ids_first = entry_at(rank)->ids[lane0 < ENTRY_IDS ? lane0 : 0];
	asm volatile("" : "+v"(head.x), "+v"(head.y), "+v"(head.z), "+v"(head.w), "+v"(ids_first), "+v"(n_work_v));
	const uint32_t n_work = (uint32_t)uniform((int)n_work_v);
	while (rank < n_work)
	{
		// (the lane index is made opaque per iteration: otherwise every lane-dependent address of the body is hoisted out of the
		// loop and kept -- spilled -- in registers across it: + 150 VGPRs for a loop that usually runs once or twice)
		int lane = wave_lane();
		asm volatile("" : "+v"(lane));
		// this iteration's entry has been requested an iteration ago (or in front of the loop); the NEXT one of this walker is requested
		// now, a whole tile ahead of its use: its position does not depend on anything this tile computes
		const uint4 cur = head;
		const uint32_t ids12 = ids_first;
		rank = dyn ? ticket_rank((uint32_t)uniform((int)tk)) : rank + (uint32_t)stride;
		if (rank < n_work)
		{
			head = *(const uint4 *)entry_at(rank);
			ids_first = entry_at(rank)->ids[lane < ENTRY_IDS ? lane : 0];
			if (dyn && lane == 0)
				tk = atomicAdd(tick, 1u);
		}
		const uint32_t e_tile = (uint32_t)uniform((int)cur.x), e_ntri = (uint32_t)uniform((int)cur.y), e_nedge = (uint32_t)uniform((int)cur.z);
		if (FUSED && (!TEX || TEXPAIR) && MODE == FWD_NO_EDGES && (e_tile & PAIR_FLAG))
		{ // two adjacent tiles, two pixels per lane (the scan kernel pairs them up: fwd_pair_tiles)
			fwd_pair_tiles<PixT, CLAMP, TEX>(p, w, S, view, lane, (int)(e_tile & ~PAIR_FLAG), (int)(e_ntri & 0xffffu), (int)(e_ntri >> 16), ids12,
								 p.loss_wave ? p.loss_wave + (size_t)view * LOSS_SLOTS + q % LOSS_SLOTS : nullptr);
			lds_sync();
			continue;
		}
		if (MANYC)
		{ // more than CH channels (forward only, no edges, no texture: the host's rule)
			fwd_manyc_tile<PixT>(p, w, S, view, lane, (int)e_tile, (int)e_ntri, ids12);
			lds_sync();
			continue;
		}
		// The rest of the body twice in the FWD_EDGE_ADJ instance: for the tiles of at most ONE batch of TB edges (one_batch: most of the head)
		// without the state of a tile of several -- eight masks instead of one, the colour and transparency a split part starts from, the
		// un-staged fallback -- that every head tile used to carry (and spill) through its pass 2 and reverse sweep.  It pays in the TEXTURED
		// instance (250 spilled registers at 168 -> 197): configs[4], 8 views 0.874 - 0.877 -> 0.833 - 0.835 ms, 4 views 0.451 -> 0.433, 2 views
		// 0.242 -> 0.235, one view level; the untextured instance is level (forward 68.2 / 68.5 against 68.8 / 68.6 us, 69.9 / 68.2 against
		// 70.0 / 67.1).  Both as a generic lambda called twice -- code generation is touchy here: the same body as text included three times left the
		// textured instance where it was (382 spilled registers, 0.870 - 0.880 ms), and the lambda with ONE call in the untextured instance cost
		// it 15 spilled registers and 11 us (profiles/r05y_ab_one_batch_body.txt, r05z4_ab_one_batch_body_textured.txt).
		auto tile_body = [&](auto one_batch_tag) __attribute__((always_inline)) {
		constexpr bool one_batch = decltype(one_batch_tag)::value;
		constexpr int NBATCH = one_batch ? 1 : EMAX / TB;
		const uint32_t nedge_word = MODE == FWD_NO_EDGES ? 0u : e_nedge;
		// (FWD_EDGE_ADJ: a tile of several batches of edges is listed once per batch, see tile_scan_kernel)
		bool split = MODE == FWD_EDGE_ADJ && !one_batch && (nedge_word & SPLIT_FLAG);
		const int part = split ? (int)((nedge_word >> 16) & 0xffu) : 0;
		const int tile = (int)e_tile, ntri = (int)e_ntri, nedge = (int)(split ? (nedge_word & 0xffffu) : nedge_word);
		const uint32_t sweep_slot = AA ? 0u : (uint32_t)uniform((int)cur.w);
		const int tx = tile % p.L.tiles_x, ty = tile / p.L.tiles_x;
		const int x0 = tx * TILE, y0 = ty * TILE;
		const int px = x0 + (lane & 7), py = y0 + (lane >> 3);
		const bool inb = p.aligned || (px < W && py < H);
		const size_t pix = (size_t)py * W + px;
		const size_t vpix = (size_t)view * H * W + pix;
		const double x = px, y = py;
		// more than ENTRY_IDS triangles: the rest of the inline list (one more round trip, one tile in ten)
		const uint32_t list_entry = ntri <= ENTRY_IDS ? ids12 : w.tri_list[(size_t)tile * K_TRI + (lane & (K_TRI - 1))];
		{
		{
		PixT ob[CH] = {0, 0, 0, 0};
		constexpr bool fuse_edges = MODE == FWD_EDGE_ADJ; // tiles with silhouette edges are back-propagated right here as well
		if ((AA && inb) || (FUSED && inb && ((nedge == 0 ? ntri > 0 : fuse_edges) || p.loss_wave)))
		{ // requested now, used after the last triangle
			const PixT *o = (const PixT *)p.obs + vpix * C;
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				if (cc < C)
					ob[cc] = o[cc];
		}
		PixState st;
		st.zbest = INFINITY;
		st.kbest = -1;
		st.kind = KIND_NONE;
		st.slot = 0;
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
			st.v[cc] = 0;
		// ---- pass 1
		if (ntri > 0)
		{
			const int n_inline = ntri < K_TRI ? ntri : K_TRI;
			for (int base = 0; base < n_inline; base += TB)
			{
				const int nb = n_inline - base < TB ? n_inline - base : TB;
				if (lane >= base && lane < base + nb)
					S.ids[lane - base] = list_entry;
				lds_sync();
				stage_batch(S, w.tri_rec, w.tri_planes, P, nb, lane);
				lds_sync();
				tri_batch<TEX>(p, S, nb, lane, x0, y0, inb, st);
			}
			if (ntri > K_TRI)
			{ // spilled pairs of this tile: compact them out of the pool, TB at a time
				uint32_t spill_n = w.hdr->tri_spill[w.hdr->cur];
				if (spill_n > p.L.tri_pool_cap)
					spill_n = p.L.tri_pool_cap;
				int fill = 0;
				for (uint32_t i0 = 0; i0 < spill_n; i0 += 64)
				{
					const uint2 pr = (i0 + lane < spill_n) ? w.tri_pool[i0 + lane] : make_uint2(0xffffffffu, 0u);
					unsigned long long m = __ballot((int)pr.x == tile);
					while (m)
					{
						const int room = TB - fill;
						const int cnt = __popcll(m);
						// lanes whose pair matches take consecutive slots; at most `room` of them this round
						const int rank = __popcll(m & ((1ull << lane) - 1ull));
						const bool sel = ((m >> lane) & 1ull) && rank < room;
						if (sel)
							S.ids[fill + rank] = pr.y;
						const unsigned long long taken = __ballot(sel);
						m &= ~taken;
						fill += cnt < room ? cnt : room;
						if (fill == TB)
						{
							lds_sync();
							stage_batch(S, w.tri_rec, w.tri_planes, P, TB, lane);
							lds_sync();
							tri_batch<TEX>(p, S, TB, lane, x0, y0, inb, st);
							fill = 0;
						}
					}
				}
				if (fill > 0)
				{
					lds_sync();
					stage_batch(S, w.tri_rec, w.tri_planes, P, fill, lane);
					lds_sync();
					tri_batch<TEX>(p, S, fill, lane, x0, y0, inb, st);
				}
			}
		}
		// ---- resolve the winner's colour
		double col[CH];
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
			col[cc] = st.v[cc];
		if (st.kbest < 0)
		{
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				col[cc] = (cc < C && inb) ? background_channel<PixT>(p, view, pix, cc) : 0.0;
		}
		Tap tap;
		double L = 0;
		if (st.kbest >= 0 && st.kind == KIND_TEXTURED && TEX)
		{
			bilinear_tap(p.tex_w, p.tex_h, st.v[0], st.v[1], C, tap);
			L = st.v[2];
			PixT tx[4][4];
			tap_texels(texture, tap, C, tx);
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				col[cc] = cc < C ? bilinear_mix(tap, (double)tx[0][cc], (double)tx[1][cc], (double)tx[2][cc], (double)tx[3][cc]) * L : 0.0;
		}
		double err = 0; // (AA) the squared residual of the un-antialiased pixel, then blended by the edges in place of the colour
		if (AA)
		{
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				if (cc < C && inb)
				{
					const double d = col[cc] - (double)ob[cc];
					err += d * d;
				}
		}
		// ---- pass 2: edges far -> near, TB at a time (H.h:2839-2900)
		int n_edges = 0;
		uint32_t tm[NBATCH] = {}; // (fused adjoint) bit j of tm[b]: edge 16 b + j of the blending order is drawn over this pixel
		if (nedge > 0)
			n_edges = gather_sorted_edges(s_es[wave], w, p, tile, nedge, lane);
		static_assert(TB <= K_EDGE && TB <= EMAX, "a tile of at most one batch of edges has them all in its inline list: gather_sorted_edges cannot fail there");
		if (one_batch) // (at most TB <= K_EDGE edges: all of them in the tile's inline list, so n_edges is the tile's count: 1 .. TB)
			n_edges = n_edges < 0 ? 0 : n_edges > TB ? TB : n_edges;
		if (split && n_edges < 0)
		{ // (pairs of this tile lost to a pool overflow -- the call is repeated anyway: the first copy alone takes the un-staged path)
			if (part > 0)
				return;
			split = false;
		}
		const int SPLIT_PART = p.split_part;
		const int last_part = split ? (nedge + SPLIT_PART - 1) / SPLIT_PART - 1 : 0;
		const int part_end = (part + 1) * SPLIT_PART; // (split tile) the first edge of the blending order behind this part's
		double colp[CH] = {0, 0, 0, 0}, trp = 1; // (split tile) the colour after this part's batch, the transparency of everything drawn later
		if (n_edges > 0)
		{
			static_assert(EMAX == 128 && TB == 16, "layout of the saved masks: one 16-bit word per batch of 16 edges");
			uint32_t snap = 0; // 1 + index of this tile's per-batch snapshots
			if (sweep_slot)
			{ // the blending order of the tile's edges: the adjoint need not gather and sort them again
				for (int i = lane; i < n_edges; i += 64)
					((uint32_t *)(w.edge_sweep + (size_t)(sweep_slot - 1) * SWEEP_BYTES + SWEEP_ORDER))[i] = s_es[wave].sorted[i];
				if (n_edges > TB)
				{
					uint32_t at = 0;
					if (lane == 0)
						at = atomicAdd(&w.hdr->snap_count[w.hdr->cur], 1u);
					at = (uint32_t)uniform((int)at);
					snap = at < (uint32_t)SNAP_CAP ? at + 1 : 0u;
				}
				if (lane == 0)
					*(uint32_t *)(w.edge_sweep + (size_t)(sweep_slot - 1) * SWEEP_BYTES + SWEEP_SNAP) = snap;
			}
			const RecSlot *erec = S.rec;
			for (int first = 0; first < n_edges; first += TB)
			{
				const int nb = n_edges - first < TB ? n_edges - first : TB;
				const uint32_t ecov = stage_edge_batch(S, s_es[wave], w, P, first, nb, lane, x0, y0, W, inb);
				uint32_t drawn_batch = 0;
				for (int j = 0; j < nb; j++)
				{
					if (fuse_edges && split && first + j == part_end)
					{ // the colour after this part's edges: where its reverse sweep starts
#pragma unroll
						for (int cc = 0; cc < CH; cc++)
							colp[cc] = col[cc];
					}
					const bool c = (ecov >> j) & 1u;
					if (__ballot(c) == 0)
						continue;
					const EdgeRec &e = erec[j].edge();
					double Ze = plane_at(e.xZ, x, y);
					if (persp)
						Ze = 1 / Ze;
					if (c && Ze < st.zbest)
					{
						drawn_batch |= 1u << j;
						const double *ep = &S.planes[j * 12];
						const double Tr = plane_at(e.x2t, x, y);
						Tap etap;
						double eL = 0, eUV[2];
						if (e.kind == KIND_TEXTURED && TEX)
							textured_tap(ep, x, y, persp, Ze, p.tex_w, p.tex_h, C, etap, eL, eUV);
						if (AA)
						{ // H.h:2154-2193, 2441-2472: the edge paints its squared distance to the observation over the error buffer
							double Err = 0;
#pragma unroll
							for (int cc = 0; cc < CH; cc++)
								if (cc < C)
								{
									const double d = edge_channel<PixT, TEX>(e, ep, texture, etap, eL, cc, x, y, persp, Ze) - (double)ob[cc];
									Err += d * d;
								}
							err *= Tr;
							err += (1 - Tr) * Err;
						}
						else
						{
#pragma unroll
						for (int cc = 0; cc < CH; cc++)
							if (cc < C)
							{
								const double A = edge_channel<PixT, TEX>(e, ep, texture, etap, eL, cc, x, y, persp, Ze);
								col[cc] *= Tr;
								col[cc] += (1 - Tr) * A;
							}
						}
						if (fuse_edges && split && first + j >= part_end)
							trp *= Tr;
					}
				}
				if (fuse_edges)
				{
#pragma unroll
					for (int bb = 0; bb < NBATCH; bb++)
						tm[bb] = bb == first / TB ? drawn_batch : tm[bb];
				}
				if (sweep_slot) // bit j: edge first + j of the blending order is drawn over this pixel
					((uint16_t *)(w.edge_sweep + (size_t)(sweep_slot - 1) * SWEEP_BYTES + CH * 64 * sizeof(double)))[(first / TB) * 64 + lane] =
						(uint16_t)drawn_batch;
				if (snap && first + TB < n_edges)
				{ // the colour after this batch: where the reverse sweep of the previous (farther) batches starts
					double *shot = (double *)(w.edge_snap + (size_t)(snap - 1) * SNAP_BYTES) + (size_t)(first / TB) * CH * 64;
#pragma unroll
					for (int cc = 0; cc < CH; cc++)
						shot[cc * 64 + lane] = col[cc];
				}
			}
			if (sweep_slot)
			{ // with the masks, what the adjoint's forward sweep would recompute: the antialiased colour in double
				double *slot = (double *)(w.edge_sweep + (size_t)(sweep_slot - 1) * SWEEP_BYTES);
#pragma unroll
				for (int cc = 0; cc < CH; cc++)
					slot[cc * 64 + lane] = col[cc];
			}
		}
		else if (!one_batch && n_edges < 0)
		{ // more than EMAX edges in one tile: ordered search through list + pool, records straight from memory
			uint32_t edge_spill_n = w.hdr->edge_spill[w.hdr->cur];
			if (edge_spill_n > p.L.edge_pool_cap)
				edge_spill_n = p.L.edge_pool_cap;
			EdgeCursor cur = {0, 0};
			for (int r = 0; r < nedge; r++)
			{
				EdgeCursor f;
				const uint32_t slot = next_edge(w, tile, nedge, edge_spill_n, r == 0, cur, false, lane, f);
				cur = f;
				if (slot == 0xffffffffu)
					break;
				const EdgeRec &e = w.edge_rec[slot];
				if (edge_touches(e, px, py, W, persp, st.zbest, inb))
				{
					const double *ep = w.edge_planes + (size_t)slot * 3 * P;
					double Ze = plane_at(e.xZ, x, y);
					if (persp)
						Ze = 1 / Ze;
					const double Tr = plane_at(e.x2t, x, y);
					Tap etap;
					double eL = 0, eUV[2];
					if (e.kind == KIND_TEXTURED && TEX)
						textured_tap(ep, x, y, persp, Ze, p.tex_w, p.tex_h, C, etap, eL, eUV);
					double Err = 0;
#pragma unroll
					for (int cc = 0; cc < CH; cc++)
						if (cc < C)
						{
							const double A = edge_channel<PixT, TEX>(e, ep, texture, etap, eL, cc, x, y, persp, Ze);
							if (AA)
								Err += (A - (double)ob[cc]) * (A - (double)ob[cc]);
							else
							{
								col[cc] *= Tr;
								col[cc] += (1 - Tr) * A;
							}
						}
					if (AA)
					{
						err *= Tr;
						err += (1 - Tr) * Err;
					}
				}
			}
		}
		// ---- one write per pixel
		if (inb && (!split || part == last_part))
		{
			if (p.image)
			{
				PixT *out = (PixT *)p.image + vpix * C;
				// streaming (non-temporal) stores: the frame is written once and not re-read by this kernel, keep L2 for records
				if (C == 4)
				{
					typedef PixT V4 __attribute__((ext_vector_type(4)));
					const V4 v = {(PixT)col[0], (PixT)col[1], (PixT)col[2], (PixT)col[3]};
					__builtin_nontemporal_store(v, (V4 *)out);
				}
				else
				{
#pragma unroll
					for (int cc = 0; cc < CH; cc++)
						if (cc < C)
							__builtin_nontemporal_store((PixT)col[cc], out + cc);
				}
			}
			if (p.zbuf)
				__builtin_nontemporal_store((PixT)st.zbest, (PixT *)p.zbuf + vpix);
			if (AA && p.err)
				__builtin_nontemporal_store((PixT)err, (PixT *)p.err + vpix);
			// a fused forward back-propagates through a tile without edges right below: nobody reads its owner ids again
			// (nor those of a tile with edges when its adjoint is fused too -- except the pathological tile of more than EMAX edges,
			// whose adjoint runs on the un-staged code and reads them)
			if (!FUSED || (nedge > 0 && (!fuse_edges || n_edges < 0)))
				__builtin_nontemporal_store(pack_owner(st.kbest, st.kind), w.face_id + pix);
		}
		if (FUSED && p.loss_wave && (!split || part == last_part))
		{ // this tile's part of the loss sum (image - obs)^2, of the frame as stored (rounded to the pixel type), less what the tile would
		  // contribute as pure background: the caller's table accounts for every tile as background, empty or not
			double r2 = 0;
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				if (cc < C && inb)
				{
					const double d = fit_value<CLAMP>(p, (double)(PixT)col[cc]) - (double)ob[cc];
					r2 += d * d;
				}
			r2 = wave_sum(r2);
			if (lane == 0)
				atomic_add_f64(p.loss_wave + (size_t)view * LOSS_SLOTS + q % LOSS_SLOTS, r2 - p.loss_tile_bg[1 + (size_t)view * p.L.ntiles + tile]);
		}
		if (fuse_edges && nedge > 0)
		{ // ---- adjoint of a tile with silhouette edges, in the same wavefront: reverse sweep over its edges (near -> far), then pass 1.
		  // Nothing is saved for a later kernel (no sweep, no snapshots, no owner ids) and the latency of this long dependent chain
		  // hides among the thousands of short tiles of the same launch instead of being a kernel of its own (31 us per 8-view step).
			double g[CH], base[CH] = {0, 0, 0, 0};
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				g[cc] = (cc < C && inb) ? fit_residual<CLAMP>(p, (double)(PixT)col[cc], (double)ob[cc]) : 0.0;
			if (n_edges > 0)
			{
				bool have_base = false;
				auto pixel_base = [&]() { // the un-antialiased colour (only a replay over an edge of transparency ~0 needs it)
					if (st.kbest >= 0)
					{
						const double *planes = w.tri_planes + (size_t)st.kbest * 3 * P;
#pragma unroll
						for (int cc = 0; cc < CH; cc++)
							if (cc < C)
								base[cc] = (st.kind == KIND_TEXTURED && TEX) ? textured_channel(texture, tap, cc) * L : interp_channel(planes, cc, x, y, false, 0.0);
					}
					else if (inb)
					{
#pragma unroll
						for (int cc = 0; cc < CH; cc++)
							if (cc < C)
								base[cc] = background_channel<PixT>(p, view, pix, cc);
					}
				};
				const int nbatch = (n_edges + TB - 1) / TB;
				if (split && part < last_part)
				{ // where the reverse sweep stands when it reaches this part's batch
#pragma unroll
					for (int cc = 0; cc < CH; cc++)
						g[cc] *= trp, col[cc] = colp[cc];
				}
				lds_sync();
				const int pb = part * SPLIT_PART / TB; // the batch this part's edges lie in
				edge_reverse_sweep<PixT, TEX>(p, w, S, &s_es[wave], lane, x, y, n_edges, split ? pb : nbatch - 1, split ? pb : 0, !split || pb == nbatch - 1, tm,
											  col, g, base, have_base, pixel_base, split ? part * SPLIT_PART % TB : 0, split ? part * SPLIT_PART % TB + SPLIT_PART - 1 : TB - 1);
				lds_sync();
				if (!split || part == 0)
					owner_adjoint<PixT, TEX>(p, w, lane, x, y, st.kbest, st.kbest >= 0 ? st.kind : (int)KIND_NONE, g, tap, L, (double *)&S.rec[0],
											(uint32_t *)&S.cover[0][0]);
			}
			else if (!one_batch && n_edges < 0)
			{ // more than EMAX edges in one tile: the un-staged adjoint reads the frame and the owner ids this wavefront has just written
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
				lds_sync();
				bwd_tile_generic_impl<PixT, true, TEX>(p, view, tx, ty, lane, (volatile uint32_t *)s_es[wave].sorted);
			}
		}
		if (FUSED && nedge == 0 && __ballot(st.kbest >= 0) != 0)
		{ // same residual as raster_bwd_fast_kernel forms from the stored frame: the colour is rounded to the pixel type first
			double g[CH];
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				asm volatile("" : "+v"(ob[cc])); // the observation stays in the pixel type until here: converted to double right
												 // after its load, it was spilled (four doubles per lane) through the whole of pass 1
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				g[cc] = (cc < C && inb) ? fit_residual<CLAMP>(p, (double)(PixT)col[cc], (double)ob[cc]) : 0.0;
			lds_sync();
			// (round 5: owner_adjoint_slots for the unpaired tiles -- and for the tiles with edges above -- was built and measured: with both
			// adjoints in the walker the headline instance spills 123 registers instead of 98 and the step is 0.1140 - 0.1160 ms against
			// 0.1130 with the pairs alone, profiles/r05g_*)
			// (a tile without edges: the whole LDS of the wavefront -- staging area and edge order -- is idle: twice the window)
			owner_adjoint<PixT, TEX>(p, w, lane, x, y, st.kbest, st.kbest >= 0 ? st.kind : (int)KIND_NONE, g, tap, L, (double *)&S.rec[0],
									(uint32_t *)&S.cover[0][0], (int)((sizeof(WaveLds) + sizeof(EdgeSort)) / sizeof(double)));
		}
		}
		}
		};
		if (MODE == FWD_EDGE_ADJ && DR_ONE_BATCH_BODY && !(e_nedge & SPLIT_FLAG) && e_nedge <= (uint32_t)TB)
			tile_body(std::true_type{});
		else
			tile_body(std::false_type{});
		lds_sync(); // the next tile of this wavefront reuses the staging area
	}
	if (q == 0 && hop == 0 && wave_lane() == 0)
		close_epoch(p, w, FUSED);
#ifndef DR_DYN_HOPS
#define DR_DYN_HOPS 6 // other groups a persistent walker may move on to
#endif
	if (!dyn || hop >= DR_DYN_HOPS)
		break;
	// This group's entries are all handed out.  Groups finish at different times (views differ in work, XCDs in the long tiles they hold): look at
	// every group's counter against its number of entries -- lane = 8 * view + group, eight views per pass, ONE round trip -- and move on to one that
	// has entries left: one of the own XCD's if there is any (its L2 holds that part of the scene), picked by the walker's index so that the
	// walkers that run dry together do not all queue at one counter.
	int best = -1;
	for (int v0 = 0; v0 < p.n_views && best < 0; v0 += 8)
	{
		const int v = v0 + (lane0 >> 3), gg = lane0 & 7;
		int remain = 0;
		if (v < p.n_views)
		{
			const ViewPtrs wv = view_ptrs(p, v);
			const uint32_t issued = __hip_atomic_load(&wv.edge_tile_cnt[(EDGE_LISTS + 1 + gg) * CNT_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const uint32_t nw = wv.hdr->work_count[1];
			const int tail = (int)(nw & 63u) - gg * 8;
			remain = (int)((nw >> 6) * 8u) + (tail < 0 ? 0 : (tail > 8 ? 8 : tail)) - (int)issued;
		}
		const unsigned long long any = __ballot(remain > 1), same = any & (0x0101010101010101ull << grp);
		unsigned long long pick = same ? same : any;
		if (pick)
		{
			uint32_t k = ((b * 2654435761u) >> 16) % (uint32_t)__popcll(pick);
			for (; k > 0; k--)
				pick &= pick - 1ull;
			best = v0 * 8 + (int)__builtin_ctzll(pick);
		}
	}
	if (best < 0)
		break;
	view = best >> 3;
	grp = best & 7;
	}
}

#ifndef DR_FUSE_TEX_EDGES
#define DR_FUSE_TEX_EDGES 1 // (measurement builds: 0 = the tiles with silhouette edges of a TEXTURED fit step wait for raster_bwd_edge_kernel, as until round 4)
#endif
// The forward raster of a textured fit step of DR_TEX_TWO_KERNELS views or more as TWO kernels on two streams -- the head walkers (edge adjoint: 168
// registers, three waves per SIMD) on the library's side stream, everybody else (128 registers, four waves again) on the caller's, both behind the scan
// kernel, joined in front of finalize.  As one kernel the 95 % of the tiles that hold no edge run at the occupancy the edge adjoint dictates.
// configs[4]: 8 views 0.839 - 0.841 -> 0.800 - 0.810 ms; 4 views level (0.436 / 0.434); 2 views 0.235 -> 0.275, one view 0.166 -> 0.21 (the fork and the
// join cost more than the occupancy returns), hence the threshold; head walkers at two waves: worse everywhere (profiles/r05z6_ab_two_kernels.txt).
// 0: never.  Not while the stream is being captured (the single kernel then).
#ifndef DR_TEX_TWO_KERNELS
#define DR_TEX_TWO_KERNELS 8
#endif
#ifndef DR_FUSE_EDGES
#define DR_FUSE_EDGES 1 // (measurement builds: 0 = the tiles with silhouette edges of a fit step wait for raster_bwd_edge_kernel, as in round 2)
#endif
// CLAMP: residual of sum (clamp(image) - obs)^2 (KParams::clamp).  NC: the channel count at compile time (0: whatever the scene says).
// Every per-channel statement of the walkers is guarded by `cc < C`; with C known the guards and the code behind the false ones go:
// the 8-view benchmark step (C = 4) 0.160 -> 0.150 ms.  The host picks the instance (3 and 4 channels, the fit step's kernels).
// COMMON: strict_edge = true and a frame whose sides are multiples of the tile (every pixel of every tile is inside it), the usual
// case, at compile time as well: 0.144 -> 0.141 ms.
// TEXE: (FUSED && TEX) 1: the instance for KParams::fuse_edges -- its head walkers run the adjoint of the tiles with silhouette edges too;
// 2 / 3: the same grid as TWO kernels for two streams, the head walkers (2) and everybody else (3: four waves per SIMD again) -- KParams::block_base.
// VAR (round 6, forward-only instances): 1 = antialiase_error (the edges blend the error buffer), 2 = more than CH channels (fwd_manyc_tile).
template <class PixT, bool FUSED, bool TEX, bool CLAMP = false, int NC = 0, bool COMMON = false, int TEXE = 0, int VAR = 0>
__global__ __launch_bounds__(64, DR_FWD_WAVES) void raster_fwd_fast_kernel(KParams p)
{
	p.aligned = COMMON ? 1 : 0;
	if (COMMON)
		p.strict = 1;
	if (NC)
	{ // (what the host passed, now known to the compiler: the walkers read p.C and p.L.P)
		p.C = NC;
		p.L.P = NC < 3 ? 3 : NC;
	}
	if (FUSED)
		p.persp = 0; // (a fit step: fill_params refuses perspective_correct for anything with an adjoint, as the reference does, H.h:810)
	constexpr size_t LDS_BYTES = sizeof(WaveLds) + sizeof(EdgeSort);
	__shared__ __attribute__((aligned(16))) char s_mem[LDS_BYTES];
	WaveLds *const s_lds = (WaveLds *)s_mem;
	EdgeSort *const s_es = (EdgeSort *)(s_mem + sizeof(WaveLds));
	// Roles of the grid: the tile walkers and the workgroups that stream the background of this kernel's share of the empty tiles
	// (fill_share).  (tools/variants/finalize_in_forward.patch adds a third: the per-primitive adjoint algebra, finalize_kernel's work.)
	// In a fit step of an untextured scene the fill workgroups are DEALT among the walkers, eight (one per XCD) behind every 64:
	// dispatched behind the last walker they START when the last walker has a slot, and the kernel then ends a fill later (73 MB of
	// stores per 8-view step: same-box A/B 0.1279 / 0.1274 -> 0.1238 / 0.1232 ms, profiles/r04l).  (Round 3 measured "spread evenly:
	// nothing" -- with the heavy tiles still deciding when the kernel ends.)
	const uint32_t n_walk = (uint32_t)p.n_views * p.fwd_walkers, n_fill = p.fwd_n_fill; // (= n_views * fill_share(fill_mode, 2, nwords), from the host)
#ifndef DR_FILL_DEAL
#define DR_FILL_DEAL 1 // (measurement builds: 0 = the fill workgroups behind the walkers, as in round 3)
#endif
	const uint32_t dealt = (DR_FILL_DEAL && FUSED && !TEX) ? p.fwd_dealt : 0; // groups of 64 walkers + 8 fill workgroups (the host: fuse_edges && n_walk >= 8 n_fill ? n_fill / 8 : 0)
	uint32_t b = blockIdx.x + p.block_base; // (32-bit throughout: see fwd_tiles)
	int fi = -1;
	if (DR_DYN_WALKERS && p.dyn_groups > 0)
	{ // (persistent walkers: half as many workgroups for the same fill -- sixteen fill workgroups behind every 64 walkers)
		if (b < dealt * 80)
		{
			const uint32_t grp = b / 80, r = b - grp * 80;
			if (r < 64)
				b = grp * 64 + r;
			else
				fi = (int)(grp * 16 + (r - 64));
		}
		else
		{
			b -= dealt * 16;
			if (b >= n_walk)
				fi = (int)(dealt * 16 + (b - n_walk));
		}
	}
	else if (b < dealt * 72)
	{
		const uint32_t grp = b / 72, r = b - grp * 72;
		if (r < 64)
			b = grp * 64 + r;
		else
			fi = (int)(grp * 8 + (r - 64));
	}
	else
	{
		b -= dealt * 8;
		if (b >= n_walk)
			fi = (int)(dealt * 8 + (b - n_walk));
	}
	if (fi >= 0)
	{
		DR_WAVE_TRACE_SCOPE(2);
		DR_WAVE_TRACE_ROLE(1u);
		if ((uint32_t)fi < n_fill)
			fill_share_word(p, 2, (int)((uint32_t)fi % (uint32_t)p.n_views), (int)((uint32_t)fi / (uint32_t)p.n_views), wave_lane());
		return;
	}
	if constexpr (TEXE == 2)
		fwd_tiles<PixT, FUSED, TEX, FWD_EDGE_ADJ, CLAMP>(p, s_lds, s_es, b); // (this launch: the head walkers, nobody else)
	else if constexpr (TEXE == 3)
		fwd_tiles<PixT, FUSED, TEX, FWD_NO_EDGES, CLAMP>(p, s_lds, s_es, b); // (this launch: everybody else)
	else if (FUSED && DR_FUSE_EDGES && (!TEX || TEXE))
	{ // (p.fuse_edges is set: the host and the scan kernel follow the same rule -- fit step of an untextured scene)
		const int G = p.tile_blocks;
		const bool chunked = G % (8 * WORK_CHUNK) == 0;
		const int q = chunked ? (int)div_views(p, b >> 3) * 8 + (int)(b & 7) : 0;
		if (chunked && q >= (int)p.fwd_heads)
			fwd_tiles<PixT, FUSED, TEX, FWD_NO_EDGES, CLAMP, false, false, TEXE == 1>(p, s_lds, s_es, b); // the rest of the list: no tile with edges
		else
			fwd_tiles<PixT, FUSED, TEX, FWD_EDGE_ADJ, CLAMP>(p, s_lds, s_es, b); // the head (tiny frames: the whole list)
	}
	else
		fwd_tiles<PixT, FUSED, TEX, FWD_PLAIN, CLAMP, VAR == 1, VAR == 2>(p, s_lds, s_es, b);
}

} // namespace
