// deodr_amd/csrc/dr_kernels.hip -- HIP kernels (gfx950 / CDNA4, wave64) and the C ABI of libdeodr_hip.so.
//
// Kernels (n_views views per launch; DESIGN.md section 4 has the why of every choice below):
//
//   setup_bin_kernel        blocks of 256 threads on a 1-D grid, the edge blocks first: 256 edge slots compacted to the flagged
//                           ones, or one triangle per thread.  Cull, depth sum, stencil + attribute planes in double and in
//                           registers (dr_prims.h), edge records (+ the EdgeFin record finalize reads), the index checks of
//                           checkSceneValid, binning into 8 x 8 tiles (all slot requests of a 3 x 3 block of tiles in flight;
//                           large boxes by the whole wavefront), optional gradient clearing
//   tile_scan_kernel        one thread per tile: counters -> work list of the NON-EMPTY tiles (entries carry the first triangle
//                           ids; many-primitive tiles first), tile bitmap, three lists of edge tiles by edge count, sweep slots
//   raster_fwd_fast_kernel  1 wavefront / work-list entry, lane = pixel.  Pass 1 (z-buffered triangles staged 16 at a time
//                           through LDS, exact scanline spans, winner = min (Z, index)), shading of the winner, pass 2 (ordered
//                           edge overdraw) fused in registers, ONE write of image / z (/ owner) per pixel.  FUSED: also the
//                           adjoint of pass 1 for the sum-of-squares residual in tiles without edges (deodr_hip_render_scene_fit)
//   fill_kernel / fill_word background + depth = inf of the empty tiles, runs of tiles written as contiguous 16-byte pieces:
//                           a kernel on a forked stream (forward-only calls) or extra workgroups of the two kernels below (fit step)
//   raster_bwd_fast_kernel  (two-call path) adjoint of pass 1 in every non-empty tile without edges
//   raster_bwd_edge_kernel  persistent waves over the listed edge tiles: adjoint of pass 2 (un-blend in reverse order, moments
//                           by a transposing butterfly, one 15-lane atomic per edge and tile), then of pass 1
//   finalize_kernel         per primitive: moments -> plane adjoints -> 3x3-inverse adjoint -> vertex gradients
//   raster_fwd_kernel / raster_bwd_kernel   the same algorithm without LDS staging: nb_colors > 4, antialiase_error
//
// No MFMA anywhere: the path is gather / scatter + streaming writes.  The workspace is self-cleaning (tile counters are
// zeroed by the scan kernel, spill counters are double-buffered by the parity of the forward count, list counters are zeroed
// by set-up, the moment accumulators by finalize) so a call never needs a memset node.  Every global atomic of the path is
// executed at the memory side on this part (TCC_EA0_ATOMIC == TCC_ATOMIC): their number, not their addresses, is what counts.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/deodr_hip.h"
#include "dr_prims.h"

using namespace dr;

namespace
{

#ifndef DR_ABLATE
#define DR_ABLATE 0 // measurement builds only (tools/build_variants.sh): 4 no frame stores of non-empty tiles, 8 no fill waves'
					// stores, 128 no accumulator atomics of the owner adjoint, 256 no owner adjoint in the fused forward, 512 no span
					// arithmetic (every staged triangle covers its whole tile).  The product is always built with 0.
#endif
constexpr int TILE = 8;		// 8 x 8 pixels = one wavefront, lane = (y & 7) * 8 + (x & 7)
constexpr int K_TRI = 32;	// inline triangle slots per tile; more spill to the pool
constexpr int K_EDGE = 32;	// inline edge slots per tile (== TB: one staged batch)
constexpr int CH = 4;		// colour channels kept in registers at a time
constexpr int MAX_SORTED = 64; // edges of a tile whose blending order is cached in LDS
constexpr int CNT_STRIDE = 32; // uint32 between two append counters (one 128-byte line each)
constexpr int PRIO_EDGES = 8; // tiles with more edges than this are listed apart: the adjoint's edge kernel starts with them
// Tiles that receive more than FIRST_PRIMS triangles (or edges) are the long poles of the forward raster: the scan kernel
// puts them at the head of the work list, so that the 25-50 us waves start at time 0 instead of ending 30 us after every
// other wave of the kernel.
#ifndef DR_FIRST_PRIMS
#define DR_FIRST_PRIMS 8
#endif
constexpr int FIRST_PRIMS = DR_FIRST_PRIMS;
// Lists of the tiles that hold silhouette edges, by edge count (disjoint; written by tile_scan_kernel, walked by
// raster_bwd_edge_kernel): 0 = 1 .. PRIO_EDGES edges, 1 = PRIO_EDGES + 1 .. TB (one batch), 2 = more than one batch.
constexpr int EDGE_LISTS = 3;
// The forward sweep over a tile's edges (pass 2) leaves, per pixel, the antialiased colour in double and the mask of the
// edges drawn: the forward raster saves both for the first SAVE_SUB edge tiles of every sub-list, so that the adjoint's edge
// kernel starts with the reverse sweep instead of repeating the forward one (half of its time per tile).
constexpr int SWEEP_CAP = 4096; // saved sweeps per view
constexpr uint32_t SWEEP_SAVED = 0x80000000u; // flag in edge_saved[tile]
constexpr size_t SWEEP_ORDER = 64 * (CH * sizeof(double) + (128 / 16) * sizeof(uint16_t)); // offset of the saved blending order
constexpr size_t SWEEP_SNAP = SWEEP_ORDER + 128 * sizeof(uint32_t);   // offset of the word: 1 + index of the tile's snapshots, 0: none
constexpr size_t SWEEP_BYTES = SWEEP_SNAP + 64;						   // 3.6 KB per tile: cur[CH][64], masks[8][64], order[128], word
// A tile with more than one batch of edges is the long pole of the adjoint's edge kernel (a 50-edge tile: 30 us of dependent
// arithmetic).  For up to SNAP_CAP such tiles per view the forward also saves the colour after every batch, so that every
// batch of the reverse sweep can be given to a wavefront of its own (it starts from the colour before its batch, and from the
// gradient scaled by the transparencies of the nearer edges drawn over the pixel).
constexpr int ROW_GROUP = 2; // tile rows per strip dealt to an XCD by the raster kernels (xcd_strip_row)
constexpr int SNAP_CAP = 256;
constexpr int CHUNKS = 128 / 16; // batches of a tile = wavefronts that may share its reverse sweep
constexpr size_t SNAP_BYTES = (CHUNKS - 1) * CH * 64 * sizeof(double);

// Entry of the staged forward's work list (one per non-empty tile, written by tile_scan_kernel): everything a wavefront needs to
// start on the tile comes with ONE memory round trip -- the header with a scalar load, the first triangle ids with a vector
// load issued at the same time (a tile with more triangles reads its inline list / the spill pool as well).
constexpr int ENTRY_IDS = 12;
struct alignas(64) WorkEntry
{
	uint32_t tile, ntri, nedge, sweep_slot;
	uint32_t ids[ENTRY_IDS];
};
static_assert(sizeof(WorkEntry) == 64, "");

struct WsHeader // 64 bytes per view at the start of the view's workspace (also the status block the host may poll)
{
	// Spill counters are double-buffered by the parity of `epoch` (one forward = one epoch): the set-up kernel of a forward
	// counts into [cur] and clears [1 - cur] for the next forward, so no memset node and no last-block ticket is needed.
	uint32_t tri_spill[2];	// (tile, triangle) pairs pushed to the pool
	uint32_t edge_spill[2]; // (tile, edge) pairs pushed to the pool
	uint32_t epoch;			// number of forwards run on this workspace (advanced by one thread of the forward raster)
	uint32_t cur;			// parity used by the forward whose state the workspace holds (written by its set-up kernel)
	uint32_t needed_max;	// sticky: largest spill count ever seen (the host compares it with the pool capacity)
	uint32_t scene_errors;	// sticky: DEODR_HIP_ERR_* bits raised by the set-up kernel (checkSceneValid's index checks, H.h:2700-2712)
	uint32_t owners_partial; // 1: the last forward was the fused one (owner ids only written for the tiles that hold edges)
	uint32_t snap_count[2];	 // tiles whose forward sweep is also saved batch by batch (edge_snap), by forward parity
	// view 0 only: maximum / union of needed_max / scene_errors over the views, so that the host polls ONE 64-byte block
	uint32_t all_needed_max, all_scene_errors;
	uint32_t work_count[2]; // entries of the forward's work list: [0] many-primitive tiles (from the front), [1] the others (from the back)
	uint32_t pad[1];
};
static_assert(sizeof(WsHeader) == 64, "");
static_assert(offsetof(WsHeader, all_needed_max) == 4 * DEODR_HIP_STATUS_WORD_NEEDED_PAIRS &&
				  offsetof(WsHeader, all_scene_errors) == 4 * DEODR_HIP_STATUS_WORD_SCENE_ERRORS &&
				  (int)dr::SCENE_ERR_FACES == DEODR_HIP_ERR_FACES && (int)dr::SCENE_ERR_FACES_UV == DEODR_HIP_ERR_FACES_UV &&
				  (int)dr::SCENE_ERR_NO_TEXTURE == DEODR_HIP_ERR_NO_TEXTURE,
			  "status block layout published in include/deodr_hip.h");

struct Layout
{
	size_t hdr, tri_rec, tri_planes, tri_acc, edge_rec, edge_planes, edge_acc, tri_cnt, edge_cnt, edge_saved, tri_list, edge_list, tri_pool,
		edge_pool, face_id, tile_bits, tri_flag, work_list, edge_tile_cnt, edge_tiles, edge_slot, edge_sweep, edge_snap, view_bytes;
	uint32_t tri_pool_cap, edge_pool_cap;
	size_t edge_fin;
	int tiles_x, tiles_y, ntiles, nwords, P, sweep_cap;
};

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

Layout make_layout(int T, int H, int W, int C, size_t pool_pairs)
{
	Layout L;
	L.P = planes_per_prim(C);
	L.tiles_x = (W + TILE - 1) / TILE;
	L.tiles_y = (H + TILE - 1) / TILE;
	L.ntiles = L.tiles_x * L.tiles_y;
	size_t pool = pool_pairs ? pool_pairs : (size_t)4 * T + (size_t)4 * L.ntiles + 4096;
	if (pool > 0x7fffffffu)
		pool = 0x7fffffffu;
	L.tri_pool_cap = L.edge_pool_cap = (uint32_t)pool;
	size_t o = 0;
	auto take = [&](size_t bytes) {
		size_t at = o;
		o = align256(o + bytes);
		return at;
	};
	L.hdr = take(sizeof(WsHeader));
	L.tri_rec = take(sizeof(TriRec) * (size_t)T);
	L.tri_planes = take(sizeof(double) * 3 * L.P * (size_t)T);
	L.tri_acc = take(sizeof(double) * 3 * L.P * (size_t)T);
	L.edge_rec = take(sizeof(EdgeRec) * 3 * (size_t)T);
	L.edge_planes = take(sizeof(double) * 3 * L.P * 3 * (size_t)T);
	L.edge_acc = take(sizeof(double) * (3 * L.P + 3) * 3 * (size_t)T);
	L.tri_cnt = take(sizeof(uint32_t) * L.ntiles);
	L.edge_cnt = take(sizeof(uint32_t) * L.ntiles);
	L.edge_saved = take(sizeof(uint32_t) * L.ntiles);
	L.tri_list = take(sizeof(uint32_t) * K_TRI * (size_t)L.ntiles);
	L.edge_list = take(sizeof(uint32_t) * K_EDGE * (size_t)L.ntiles);
	L.tri_pool = take(sizeof(uint2) * (size_t)L.tri_pool_cap);
	L.edge_pool = take(sizeof(uint2) * (size_t)L.edge_pool_cap);
	L.face_id = take(sizeof(int32_t) * (size_t)H * W);
	// one bit per tile (the tile received a primitive) and the work list of the staged forward: one uint4 {tile, triangles,
	// edges, sweep slot} per non-empty tile, both written by tile_scan_kernel between set-up and forward raster
	L.nwords = (L.ntiles + 31) / 32;
	L.tile_bits = take(sizeof(uint32_t) * L.nwords);
	L.work_list = take(sizeof(WorkEntry) * (size_t)L.ntiles);
	// kind | front << 2 of every triangle of the last forward: what finalize_kernel needs to know about a triangle before it
	// touches anything else (one coalesced byte per thread instead of a 128-byte record line per triangle, two out of three
	// of which are culled)
	L.tri_flag = take((size_t)T);
	L.edge_tile_cnt = take(sizeof(uint32_t) * (EDGE_LISTS + 1) * CNT_STRIDE); // append counters of the lists + the sweep-slot counter
	L.edge_tiles = take(sizeof(uint32_t) * EDGE_LISTS * (size_t)L.ntiles);	   // [list][ntiles]
	L.edge_slot = take(sizeof(uint32_t) * L.ntiles); // 1 + index of the tile's slot in edge_sweep, 0: none
	L.sweep_cap = SWEEP_CAP < L.ntiles ? SWEEP_CAP : L.ntiles;
	L.edge_sweep = take(SWEEP_BYTES * (size_t)L.sweep_cap);
	// what finalize_kernel needs of a drawn silhouette edge besides its record: vertex ids, positions, attributes (written by the
	// set-up kernel, which has them in registers: the finalize thread of an edge then has ONE memory round trip before its arithmetic
	// instead of three -- indices, vertices, record)
	L.edge_fin = take(sizeof(EdgeFin) * 3 * (size_t)T);
	L.edge_snap = take(SNAP_BYTES * SNAP_CAP);
	L.view_bytes = o;
	return L;
}

struct KParams
{
	// scene
	const uint32_t *faces, *faces_uv;
	const uint8_t *textured, *shaded, *edgeflags;
	const void *depths, *ij, *shade, *colors, *uv;
	const void *texture, *bg_image, *bg_color;
	void *uv_b, *ij_b, *shade_b, *colors_b, *texture_b;
	int T, V, Vuv, H, W, C, tex_h, tex_w;
	int clockwise, culling, strict, persp, vtx_f64;
	double offset, sigma;
	// pixel buffers of this call
	void *image, *zbuf, *err;
	const void *image_b, *obs, *err_b, *image_in;
	int aa_err;
	int n_views;
	int heavy_share; // staged forward: one workgroup in heavy_share walks the many-primitive tiles (heavy_share_for)
	int tile_blocks; // staged forward: workgroups per view that walk the work list (multiple of 512, or tiny frames: <= ntiles)
	int row_group;	 // tile rows per strip dealt to an XCD by the raster kernels (xcd_strip_row); 0: one band per XCD
	int pix_f64;	 // pixel buffers are double (for the kernels that are not templates on the pixel type)
	// Background fill of a fit step (see fill_word): 0 = by fill_kernel on the side stream; otherwise by extra workgroups of the
	// adjoint's kernels -- bit 0: raster_bwd_edge_kernel takes part, bit 1: finalize_kernel does (both: even / odd bitmap words)
	int fill_mode;
	int clear_grads; // the set-up kernel zeroes the per-view gradient arrays (a fit step that wants fresh gradients: no separate fills)
	// workspace
	char *ws;
	Layout L;
};

struct ViewPtrs
{
	WsHeader *hdr;
	TriRec *tri_rec;
	double *tri_planes, *tri_acc;
	EdgeRec *edge_rec;
	double *edge_planes, *edge_acc;
	uint32_t *tri_cnt, *edge_cnt, *edge_saved, *tri_list, *edge_list;
	uint2 *tri_pool, *edge_pool;
	int32_t *face_id;
	uint32_t *tile_bits;
	uint8_t *tri_flag;
	uint32_t *edge_slot;
	char *edge_sweep, *edge_snap;
	WorkEntry *work_list;
	uint32_t *edge_tile_cnt, *edge_tiles; // tiles with silhouette edges: EDGE_LISTS (+ 1) counters, EDGE_LISTS lists of ntiles entries
	EdgeFin *edge_fin;
};

__device__ __forceinline__ ViewPtrs view_ptrs(const KParams &p, int view)
{
	char *b = p.ws + (size_t)view * p.L.view_bytes;
	ViewPtrs v;
	v.hdr = (WsHeader *)(b + p.L.hdr);
	v.tri_rec = (TriRec *)(b + p.L.tri_rec);
	v.tri_planes = (double *)(b + p.L.tri_planes);
	v.tri_acc = (double *)(b + p.L.tri_acc);
	v.edge_rec = (EdgeRec *)(b + p.L.edge_rec);
	v.edge_planes = (double *)(b + p.L.edge_planes);
	v.edge_acc = (double *)(b + p.L.edge_acc);
	v.tri_cnt = (uint32_t *)(b + p.L.tri_cnt);
	v.edge_cnt = (uint32_t *)(b + p.L.edge_cnt);
	v.edge_saved = (uint32_t *)(b + p.L.edge_saved);
	v.tri_list = (uint32_t *)(b + p.L.tri_list);
	v.edge_list = (uint32_t *)(b + p.L.edge_list);
	v.tri_pool = (uint2 *)(b + p.L.tri_pool);
	v.edge_pool = (uint2 *)(b + p.L.edge_pool);
	v.face_id = (int32_t *)(b + p.L.face_id);
	v.tile_bits = (uint32_t *)(b + p.L.tile_bits);
	v.tri_flag = (uint8_t *)(b + p.L.tri_flag);
	v.edge_tile_cnt = (uint32_t *)(b + p.L.edge_tile_cnt);
	v.edge_tiles = (uint32_t *)(b + p.L.edge_tiles);
	v.work_list = (WorkEntry *)(b + p.L.work_list);
	v.edge_slot = (uint32_t *)(b + p.L.edge_slot);
	v.edge_fin = (EdgeFin *)(b + p.L.edge_fin);
	v.edge_sweep = b + p.L.edge_sweep;
	v.edge_snap = b + p.L.edge_snap;
	return v;
}

__device__ __forceinline__ SceneView scene_view(const KParams &p, int view)
{
	const size_t es = p.vtx_f64 ? 8 : 4;
	SceneView s;
	s.faces = p.faces;
	s.faces_uv = p.faces_uv;
	s.textured = p.textured;
	s.shaded = p.shaded;
	s.edgeflags = p.edgeflags + (size_t)view * 3 * p.T;
	s.depths = (const char *)p.depths + (size_t)view * p.V * es;
	s.ij = (const char *)p.ij + (size_t)view * p.V * 2 * es;
	s.shade = (const char *)p.shade + (size_t)view * p.V * es;
	s.colors = (const char *)p.colors + (size_t)view * p.V * p.C * es;
	s.uv = p.uv;
	s.T = p.T;
	s.V = p.V;
	s.Vuv = p.Vuv;
	s.H = p.H;
	s.W = p.W;
	s.C = p.C;
	s.P = p.L.P;
	s.tex_h = p.tex_h;
	s.tex_w = p.tex_w;
	s.clockwise = p.clockwise;
	s.culling = p.culling;
	s.strict = p.strict;
	s.persp = p.persp;
	s.vtx_f64 = p.vtx_f64;
	s.has_texture = p.texture != nullptr;
	s.offset = p.offset;
	s.sigma = p.sigma;
	return s;
}

// owner buffer: triangle index in the low 30 bits, its PrimKind in the top 2 (3 = no owner), so that the adjoint does not
// have to gather the kind from the 128-byte record of every pixel's owner
__device__ __forceinline__ int32_t pack_owner(int k, int kind) { return k < 0 ? -1 : (int32_t)((uint32_t)k | ((uint32_t)kind << 30)); }
__device__ __forceinline__ void unpack_owner(int32_t raw, int &owner, int &kind)
{
	const uint32_t u = (uint32_t)raw;
	kind = (int)(u >> 30);
	owner = kind == 3 ? -1 : (int)(u & 0x3fffffffu);
	if (kind == 3)
		kind = KIND_NONE;
}

// ------------------------------------------------------------------------------------------------ wave primitives

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ void atomic_add_f64(double *p, double v) { unsafeAtomicAdd(p, v); }

// Cross-lane moves on the VALU (DPP), no LDS round trip.  CTRL: 0x110 + n = row_shr:n (lane i <- lane i - n inside its
// 16-lane row), 0x100 + n = row_shl:n (lane i <- lane i + n); lanes without a source read 0.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v)
{
	return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v)
{
	return __hiloint2double(dpp_i<CTRL>(__double2hiint(v)), dpp_i<CTRL>(__double2loint(v)));
}
// sum over the 64 lanes, returned to every lane (4 DPP steps inside each 16-lane row, then 4 readlanes)
__device__ __forceinline__ double wave_sum_dpp(double v)
{
	v += dpp_d<0x111>(v);
	v += dpp_d<0x112>(v);
	v += dpp_d<0x114>(v);
	v += dpp_d<0x118>(v);
	const int hi = __double2hiint(v), lo = __double2loint(v);
	double r = __hiloint2double(__builtin_amdgcn_readlane(hi, 15), __builtin_amdgcn_readlane(lo, 15));
	r += __hiloint2double(__builtin_amdgcn_readlane(hi, 31), __builtin_amdgcn_readlane(lo, 31));
	r += __hiloint2double(__builtin_amdgcn_readlane(hi, 47), __builtin_amdgcn_readlane(lo, 47));
	r += __hiloint2double(__builtin_amdgcn_readlane(hi, 63), __builtin_amdgcn_readlane(lo, 63));
	return r;
}
// NOTE: must be called with all 64 lanes enabled (a DPP move reads 0 from a disabled lane)
__device__ __forceinline__ double wave_sum(double v) { return wave_sum_dpp(v); }

// Sixteen wave sums for the price of about three: a transposing butterfly.  At step b the lanes whose bit b is clear keep the
// even member of every pair of values and receive it from a lane whose bit b is set, and vice versa, so the number of live
// values halves at each step while the number of lanes that share a value halves too (15 exchanges instead of 16 x 6).
// Lane l returns  sum over the wave of v[l & 15].  v is clobbered.  All 64 lanes must be enabled.
template <int N, int CTRL>
__device__ __forceinline__ void reduce_halve(double *v, bool bit)
{
#pragma unroll
	for (int j = 0; j < N / 2; j++)
	{
		const double keep = bit ? v[2 * j + 1] : v[2 * j];
		const double send = bit ? v[2 * j] : v[2 * j + 1];
		v[j] = keep + dpp_d<CTRL>(send);
	}
}
__device__ __forceinline__ double wave_sum16(double *v, int lane)
{
	reduce_halve<16, 0xB1>(v, lane & 1);	// quad_perm [1,0,3,2]: lane ^ 1
	reduce_halve<8, 0x4E>(v, lane & 2);		// quad_perm [2,3,0,1]: lane ^ 2
	reduce_halve<4, 0x124>(v, lane & 4);	// row_ror:4: a source whose bit 2 differs (each lane is a source exactly once)
	reduce_halve<2, 0x128>(v, lane & 8);	// row_ror:8: lane ^ 8
	// v[0]: the 16-lane row's sum of value (lane & 15); add the four rows (gfx950 row swaps: no LDS, no readlane)
	double r = v[0];
	{
		const auto h = __builtin_amdgcn_permlane16_swap(__double2hiint(r), __double2hiint(r), false, false);
		const auto l = __builtin_amdgcn_permlane16_swap(__double2loint(r), __double2loint(r), false, false);
		r = __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
	}
	{
		const auto h = __builtin_amdgcn_permlane32_swap(__double2hiint(r), __double2hiint(r), false, false);
		const auto l = __builtin_amdgcn_permlane32_swap(__double2loint(r), __double2loint(r), false, false);
		r = __hiloint2double(h[0], l[0]) + __hiloint2double(h[1], l[1]);
	}
	return r;
}

struct DeviceAdd // accumulate a vertex gradient (the reference's `+=` into scene.*_b)
{
	__device__ __forceinline__ void operator()(void *arr, size_t i, bool f64, double v) const
	{
		if (v == 0 || (DR_ABLATE & 1024))
			return;
		if (f64)
			unsafeAtomicAdd((double *)arr + i, v);
		else
			unsafeAtomicAdd((float *)arr + i, (float)v);
	}
};

// finalize_triangle's sinks (dr_prims.h).  AtomicSink: every contribution goes straight to the gradient arrays.
struct AtomicSink
{
	const SceneView &s;
	const GradView &g;
	uint32_t f[3], fuv[3];
	__device__ __forceinline__ void color(int i, int c, double v)
	{
		if ((DR_ABLATE & 131072) && c >= 2) // (measurement build: a third fewer atomic instructions per triangle)
			return;
		DeviceAdd()(g.colors_b, (size_t)f[i] * s.C + c, s.vtx_f64, v);
	}
	__device__ __forceinline__ void shade(int i, double v) { DeviceAdd()(g.shade_b, f[i], s.vtx_f64, v); }
	__device__ __forceinline__ void uv(int i, int c, double v) { DeviceAdd()(g.uv_b, 2 * (size_t)fuv[i] + c, s.vtx_f64, v); }
	__device__ __forceinline__ void ij(int i, int d, double v) { DeviceAdd()(g.ij_b, 2 * (size_t)f[i] + d, s.vtx_f64, v); }
};
// XCD-aware block order: the dispatcher sends block b to XCD b % 8; give every XCD one contiguous band of the
// screen so that neighbouring tiles (which share triangle records) share an L2.  Bijective for any block count.
__device__ __forceinline__ int xcd_band(int b, int n)
{
	int q = n >> 3, r = n & 7, xcd = b & 7, idx = b >> 3;
	return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// The bands are then dealt to the XCDs in strips of `group` tile rows (band-ordered row pr = band * rows_per_band + i becomes
// row (i / group) * 8 * group + band * group + i % group): one contiguous band per XCD leaves the XCDs that own the top and
// the bottom of the frame -- usually background -- idle while the others rasterize the object in the middle, and blocks are
// dispatched in order.  Needs tiles_y % (8 * group) == 0, otherwise the bands stay whole (any bijection is correct).
__device__ __forceinline__ int xcd_strip_row(int pr, int tiles_y, int group)
{
	if (group <= 0 || tiles_y % (8 * group) != 0)
		return pr;
	const int per_band = tiles_y / 8, band = pr / per_band, i = pr - band * per_band;
	return (i / group) * 8 * group + band * group + i % group;
}

// ----------------------------------------------------------------------------------------------------- set-up + bin

__device__ __forceinline__ void place_in_tile(uint32_t *list, int cap_inline, uint2 *pool, uint32_t pool_cap, uint32_t *spill, int tile,
											  uint32_t prim, uint32_t slot)
{
	if (slot < (uint32_t)cap_inline)
		list[(size_t)tile * cap_inline + slot] = prim;
	else
	{
		uint32_t o = atomicAdd(spill, 1u);
		if (o < pool_cap)
			pool[o] = make_uint2((uint32_t)tile, prim);
	}
}

// The same rejection for the 3 x 3 block of tiles whose first tile is (tx0, ty0): bit 3 dy + dx of the result is set when tile
// (tx0 + dx, ty0 + dy) is clearly outside one of the N half-planes.  The corner where a half-plane function is largest
// is the same in every tile, so a x and b y are formed once per column / row of tiles (the per-tile form above costs
// ~20 operations per half-plane and tile, and binning was half of the arithmetic of the set-up kernel).  The slack uses the
// largest scale of the block, i.e. it is at least as cautious as the per-tile test.
template <int N>
__device__ __forceinline__ uint32_t tiles3x3_outside_halfplanes(const double *eq, int tx0, int ty0)
{
	uint32_t out = 0;
	const double xmax = (tx0 + 2) * TILE + (TILE - 1), ymax = (ty0 + 2) * TILE + (TILE - 1);
#pragma unroll
	for (int k = 0; k < N; k++)
	{
		const double a = eq[3 * k], b = eq[3 * k + 1], c = eq[3 * k + 2];
		const double limit = -1e-9 * (fabs(a) * xmax + fabs(b) * ymax + fabs(c)) - 1e-12;
		const double cx = tx0 * TILE + (a > 0 ? TILE - 1 : 0), cy = ty0 * TILE + (b > 0 ? TILE - 1 : 0);
		double ax[3], by[3];
#pragma unroll
		for (int d = 0; d < 3; d++)
		{
			ax[d] = a * (cx + d * TILE);
			by[d] = b * (cy + d * TILE) + c;
		}
#pragma unroll
		for (int q = 0; q < 9; q++)
			out |= (ax[q % 3] + by[q / 3] < limit) ? (1u << q) : 0u;
	}
	return out;
}

__device__ __forceinline__ uint32_t push_tile(uint32_t *cnt, uint32_t *list, int cap_inline, uint2 *pool, uint32_t pool_cap, uint32_t *spill,
											  int tile, uint32_t prim)
{
	const uint32_t slot = atomicAdd(&cnt[tile], 1u);
	place_in_tile(list, cap_inline, pool, pool_cap, spill, tile, prim, slot);
	return slot; // 0: first primitive of the tile
}

// Conservative rejection for binning: a primitive covers a pixel only where every one of its half-plane functions
// E = a x + b y + c is >= 0 (or > 0); if some E is clearly negative on all four corner pixels of the tile, no pixel of the
// tile can be covered.  The slack keeps the test safe against the rounding of the exact span arithmetic used later.
template <int N>
__device__ __forceinline__ bool tile_outside_halfplanes(const double *eq, int tx, int ty)
{
	const double xa = tx * TILE, xb = tx * TILE + (TILE - 1), ya = ty * TILE, yb = ty * TILE + (TILE - 1);
#pragma unroll
	for (int k = 0; k < N; k++)
	{
		const double a = eq[3 * k], b = eq[3 * k + 1], c = eq[3 * k + 2];
		const double emax = a * (a > 0 ? xb : xa) + b * (b > 0 ? yb : ya) + c;
		const double scale = fabs(a) * xb + fabs(b) * yb + fabs(c);
		if (emax < -1e-9 * scale - 1e-12)
			return true;
	}
	return false;
}

// Work split of the per-primitive kernels (set-up, finalize).  Triangle blocks take PRIM_BLOCK triangles each.  The
// other blocks take PRIM_BLOCK edge slots (3 k + n) each, of which only the few per cent flagged as silhouette edges need
// work: the block compacts them through LDS so that they fill the lanes of its first wavefront(s) and the others retire at
// once (one thread per slot left ~2 busy lanes in almost every wavefront of the long edge path).
// (Workgroups of one wavefront -- 64 triangles, or a span of 256 edge slots compacted by each of four single-wave blocks -- were
// measured: every wave starts within 10 us instead of 23, and the kernels take 38 / 35 us instead of 35 / 33: they are bound by
// the memory-side atomics and the arithmetic of the long waves, not by wave slots.)
#ifndef DR_PRIM_BLOCK
#define DR_PRIM_BLOCK 256
#endif
constexpr int PRIM_BLOCK = DR_PRIM_BLOCK;
#ifndef DR_PRIM_WAVES
#define DR_PRIM_WAVES 3 // waves per SIMD the per-primitive kernels are compiled for (4: spills, same time)
#endif
constexpr int COOP_BLOCKS = 8; // 3 x 3-tile blocks of a bounding box one thread bins by itself

__host__ __device__ inline int prim_tri_blocks(int T) { return (T + PRIM_BLOCK - 1) / PRIM_BLOCK; }
__host__ __device__ inline int prim_blocks(int T) { return prim_tri_blocks(T) + (3 * T + PRIM_BLOCK - 1) / PRIM_BLOCK; }

// Grid of the per-primitive kernels: 1-D, n_views * prim_blocks(T) workgroups.  The edge-slot blocks of every view come first,
// then the triangle blocks (views fastest inside each class): the wavefront that works on flagged edges is the longest
// dependent chain of both kernels (13 - 20 us against 3 us for a triangle wavefront, tools/wave_trace.py), and dispatched after
// the triangle blocks it was the 15 us tail of the kernel.
#ifndef DR_EDGE_FIRST
#define DR_EDGE_FIRST 1
#endif
struct PrimWork
{
	int view, index; // index of the block inside its class
	bool tri;
	int view_block; // a block id in [0, prim_blocks(T)) inside the view (housekeeping loops)
};
__device__ __forceinline__ PrimWork prim_work(const KParams &p, bool edge_first = DR_EDGE_FIRST, int skip = 0)
{
	const int TBk = prim_tri_blocks(p.T), EB = prim_blocks(p.T) - TBk, nv = p.n_views;
	int b = (int)blockIdx.x - skip;
	PrimWork w;
	const int first = (edge_first ? EB : TBk) * nv;
	const bool in_first = b < first;
	if (!in_first)
		b -= first;
#ifndef DR_VIEW_MAJOR
#define DR_VIEW_MAJOR 0 // measurement builds: 1 = all blocks of a view, then the next view (instead of views fastest)
#endif
	const int per_view = in_first == edge_first ? EB : TBk; // blocks per view of this block's class
	w.view = DR_VIEW_MAJOR ? b / per_view : b % nv;
	w.index = DR_VIEW_MAJOR ? b % per_view : b / nv;
	w.tri = edge_first ? !in_first : in_first;
	w.view_block = w.tri ? w.index : TBk + w.index;
	return w;
}

// -> the slot this thread works on, or -1.  Called by every thread of an edge block.
__device__ __forceinline__ int compact_flagged_slots(const KParams &p, const uint8_t *edgeflags, int edge_block)
{
	__shared__ uint32_t s_slots[PRIM_BLOCK];
	__shared__ uint32_t s_count[PRIM_BLOCK / 64];
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const int slot = edge_block * PRIM_BLOCK + tid;
	const bool flagged = p.sigma > 0 && slot < 3 * p.T && edgeflags[slot] != 0;
	const unsigned long long m = __ballot(flagged);
	if (lane == 0)
		s_count[wave] = (uint32_t)__popcll(m);
	__syncthreads();
	uint32_t before = 0, total = 0;
#pragma unroll
	for (int i = 0; i < PRIM_BLOCK / 64; i++)
	{
		const uint32_t c = s_count[i];
		before += i < wave ? c : 0u;
		total += c;
	}
	if (flagged)
		s_slots[before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)slot;
	__syncthreads();
	return (uint32_t)tid < total ? (int)s_slots[tid] : -1;
}

#ifdef DR_WAVE_TRACE
// timeline of the per-primitive kernels: [wave slot] = (start, end) in 10 ns ticks of the constant 100 MHz counter
__device__ unsigned long long g_wave_trace[3][1 << 18][2]; // 0 set-up, 1 finalize, 2 forward raster
__device__ unsigned long long g_wave_phase[4][1 << 16][8];  // 0 set-up, 1 finalize: time stamps inside the wavefronts that work on edges
struct WaveTrace
{
	int which;
	unsigned long long t0;
	__device__ void phase(int i, int tri = 0) const
	{
		if ((threadIdx.x & 63) == 0 && which < 2)
		{
			const unsigned id = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
			if (id < (1u << 16))
			{
				if (i == 1)
					g_wave_phase[which + 2 * tri][id][0] = t0;
				g_wave_phase[which + 2 * tri][id][i] = __builtin_amdgcn_s_memrealtime();
			}
		}
	}
	__device__ WaveTrace(int w) : which(w), t0(__builtin_amdgcn_s_memrealtime()) { phase(0); }
	__device__ ~WaveTrace()
	{
		if ((threadIdx.x & 63) == 0)
		{
			const unsigned id = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
			if (id < (1u << 18))
			{
				g_wave_trace[which][id][0] = t0;
				g_wave_trace[which][id][1] = __builtin_amdgcn_s_memrealtime();
			}
		}
	}
};
#define DR_WAVE_TRACE_SCOPE(w) WaveTrace wave_trace_scope(w)
#define DR_WAVE_PHASE(i) wave_trace_scope.phase(i)
#define DR_WAVE_PHASE_T(i) wave_trace_scope.phase(i, 1)
#else
#define DR_WAVE_TRACE_SCOPE(w)
#define DR_WAVE_PHASE(i)
#define DR_WAVE_PHASE_T(i)
#endif

__global__ __launch_bounds__(PRIM_BLOCK, DR_PRIM_WAVES) void setup_bin_kernel(KParams p)
{
	DR_WAVE_TRACE_SCOPE(0);
	const PrimWork pw = prim_work(p);
	const int view = pw.view;
	const int item = pw.view_block * PRIM_BLOCK + threadIdx.x; // only an id for the housekeeping below
	const int n_items = prim_blocks(p.T) * PRIM_BLOCK;
	const bool tri_block = pw.tri;
	const SceneView s = scene_view(p, view);
	const ViewPtrs w = view_ptrs(p, view);
	const uint32_t cur = w.hdr->epoch & 1u; // stable during this kernel: only the forward raster advances the epoch
	if (item == 0)
	{
		w.hdr->cur = cur;
		w.hdr->tri_spill[1 - cur] = 0;
		w.hdr->edge_spill[1 - cur] = 0;
		w.hdr->snap_count[1 - cur] = 0;
		w.hdr->work_count[0] = w.hdr->work_count[1] = 0; // filled by tile_scan_kernel, read by the forward raster
	}
	if (item <= EDGE_LISTS) // appended to by tile_scan_kernel, the next kernel on the stream
		w.edge_tile_cnt[item * CNT_STRIDE] = 0;
	if (p.clear_grads && view == 0 && p.uv_b)
		for (int v = item; v < 2 * p.Vuv; v += n_items)
		{ // shared by the views: zeroed once
			if (p.vtx_f64)
				((double *)p.uv_b)[v] = 0;
			else
				((float *)p.uv_b)[v] = 0;
		}
	if (p.clear_grads)
		for (int v = item; v < p.V; v += n_items)
		{ // nothing accumulates into them before finalize_kernel, two kernels later
			const size_t at = (size_t)view * p.V + v;
			if (p.vtx_f64)
			{
				if (p.ij_b)
					((double2 *)p.ij_b)[at] = make_double2(0, 0);
				if (p.shade_b)
					((double *)p.shade_b)[at] = 0;
				if (p.colors_b)
					for (int c = 0; c < p.C; c++)
						((double *)p.colors_b)[at * p.C + c] = 0;
			}
			else
			{
				if (p.ij_b)
					((float2 *)p.ij_b)[at] = make_float2(0, 0);
				if (p.shade_b)
					((float *)p.shade_b)[at] = 0;
				if (p.colors_b)
					for (int c = 0; c < p.C; c++)
						((float *)p.colors_b)[at * p.C + c] = 0;
			}
		}
	// Records are built in registers and leave with one 128-byte store: the binning below reads the local copy (reading a
	// record back from HBM right after writing it costs a full memory round trip per field), a culled triangle only gets its two
	// flags written, an edge slot that is not a silhouette edge nothing at all.
	// A primitive whose bounding box needs more than COOP_BLOCKS blocks of 3 x 3 tiles is not binned by its own thread
	// (hundreds of dependent atomic round trips in one lane: 0.3 ms of set-up for a 1 000-triangle mesh filling a 1024^2
	// frame) but handed to the whole wavefront below: its half-planes and box are kept here.  Smaller ones stay with their
	// thread (all threads at once beat the wavefront working through its large primitives one after the other).
	const int lane = threadIdx.x & 63;
	// ---- the large primitives of this wavefront, one after the other, 64 tiles of the bounding box at a time
	auto bin_large = [&](bool big, const double *hp, int btx0, int bty0, int bntx, int bnty, int bprim) {
		unsigned long long todo = __ballot(big);
		while (todo)
		{
			const int src = __ffsll((long long)todo) - 1;
			todo &= todo - 1;
			double q[12];
#pragma unroll
			for (int i = 0; i < 12; i++)
				q[i] = __shfl(hp[i], src, 64);
			const int tx0 = __shfl(btx0, src, 64), ty0 = __shfl(bty0, src, 64), ntx = __shfl(bntx, src, 64), nty = __shfl(bnty, src, 64);
			const uint32_t prim = (uint32_t)__shfl(bprim, src, 64);
			for (int t = lane; t < ntx * nty; t += 64)
			{
				const int tx = tx0 + t % ntx, ty = ty0 + t / ntx, tile = ty * p.L.tiles_x + tx;
				if (tri_block)
				{
					if (!tile_outside_halfplanes<3>(q, tx, ty) || (!p.strict && tx == tx0 + ntx - 1))
						push_tile(w.tri_cnt, w.tri_list, K_TRI, w.tri_pool, p.L.tri_pool_cap, &w.hdr->tri_spill[cur], tile, prim);
				}
				else if (!tile_outside_halfplanes<4>(q, tx, ty))
					push_tile(w.edge_cnt, w.edge_list, K_EDGE, w.edge_pool, p.L.edge_pool_cap, &w.hdr->edge_spill[cur], tile, prim);
			}
		}
	};
	if (tri_block)
	{
		bool big = false;
		double hp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
		int btx0 = 0, bty0 = 0, bntx = 0, bnty = 0, bprim = 0;
		do
		{
			const int k = pw.index * PRIM_BLOCK + threadIdx.x;
			if (k >= p.T)
				break;
			TriInputs t;
			TriRec rec;
			TriRec &out = w.tri_rec[k];
			if (const uint32_t bad = load_triangle(s, k, t, true))
			{ // checkSceneValid (H.h:2700-2712): the triangle is dropped and the sticky error word tells the host
				atomicOr(&w.hdr->scene_errors, bad);
				w.tri_flag[k] = 0;
				break;
			}
			DR_WAVE_PHASE_T(1); // inputs (?)
			double x2b[9];
			const bool drawn = setup_tri_geometry(s, t, rec, x2b) && rec.kind != KIND_NONE;
			w.tri_flag[k] = (uint8_t)(rec.kind | (rec.front ? 4 : 0));
			if (!drawn)
				break; // culled (or textured without shading): its record is never read -- the raster kernels reach records
					   // through the tile lists, the finalize kernel looks at tri_flag first
			const int x0 = rec.x_min < 0 ? 0 : rec.x_min, x1 = rec.x_max > s.W - 1 ? s.W - 1 : rec.x_max;
			const int y0 = rec.y_begin[0] < 0 ? 0 : rec.y_begin[0], y1 = rec.y_end[1] > s.H - 1 ? s.H - 1 : rec.y_end[1];
			const bool on_screen = !(x0 > x1 || y0 > y1 || (DR_ABLATE & 2048));
			const double eq[9] = {rec.eq[0][0], rec.eq[0][1], rec.eq[0][2], rec.eq[1][0], rec.eq[1][1], rec.eq[1][2], rec.eq[2][0], rec.eq[2][1], rec.eq[2][2]};
			const int tx0 = x0 / TILE, ty0 = y0 / TILE, ntx = x1 / TILE - tx0 + 1, nty = y1 / TILE - ty0 + 1;
			const bool large = on_screen && ((ntx + 2) / 3) * ((nty + 2) / 3) > COOP_BLOCKS;
			// Non-strict fill rule: get_xrange's ceil_div clamps the left end of a row to x_max (H.h:895), so a row whose span lies
			// wholly between x_max and the rightmost vertex -- or beyond the right border of the frame -- still draws the pixel of
			// column x_max although that pixel is outside the left edge.  tri_half_span reproduces it; the half-plane test must
			// then not drop the tiles of that column.
			const int keep_dx = s.strict ? -1 : ntx - 1;
			// The slot requests of the first 3 x 3 block of tiles (for the usual small triangle: all of them) leave NOW, before the
			// attribute planes are formed and the record is stored: that arithmetic and those stores then overlap the round trip
			// of the requests (5.5 of the 12.6 us of a triangle wavefront, tools/wave_trace.py) instead of preceding it.
			uint32_t slot0[9];
			bool use0[9];
#pragma unroll
			for (int q = 0; q < 9; q++)
				use0[q] = false, slot0[q] = 0;
			if (on_screen && !large)
			{
				const uint32_t outside = tiles3x3_outside_halfplanes<3>(eq, tx0, ty0);
#pragma unroll
				for (int q = 0; q < 9; q++)
				{
					const int dx = q % 3, dy = q / 3;
					use0[q] = dx < ntx && dy < nty && (!((outside >> q) & 1u) || dx == keep_dx);
					if (use0[q])
						slot0[q] = atomicAdd(&w.tri_cnt[(ty0 + dy) * p.L.tiles_x + tx0 + dx], 1u);
				}
			}
			setup_tri_attributes(s, t, rec, x2b, w.tri_planes + (size_t)k * 3 * s.P);
			DR_WAVE_PHASE_T(2); // record computed
			rec.pad0[0] = rec.pad0[1] = 0;
			rec.pad1[0] = rec.pad1[1] = rec.pad1[2] = 0;
			if (!(DR_ABLATE & 4096))
				out = rec;
			if (!on_screen)
				break;
			if (large)
			{
				big = true;
#pragma unroll
				for (int i = 0; i < 9; i++)
					hp[i] = eq[i];
				btx0 = tx0, bty0 = ty0, bntx = ntx, bnty = nty, bprim = k;
				break;
			}
			DR_WAVE_PHASE_T(3); // record stored
#pragma unroll
			for (int q = 0; q < 9; q++)
				if (use0[q])
				{
					const int tile = (ty0 + q / 3) * p.L.tiles_x + tx0 + q % 3;
					place_in_tile(w.tri_list, K_TRI, w.tri_pool, p.L.tri_pool_cap, &w.hdr->tri_spill[cur], tile, (uint32_t)k, slot0[q]);
				}
			// the other 3 x 3 blocks of a wider box, the slot requests of a block all in flight together
			for (int by = 0; by < nty; by += 3)
				for (int bx = by == 0 ? 3 : 0; bx < ntx; bx += 3)
				{
					uint32_t slot[9];
					bool use[9];
					const uint32_t outside = tiles3x3_outside_halfplanes<3>(eq, tx0 + bx, ty0 + by);
#pragma unroll
					for (int q = 0; q < 9; q++)
					{
						const int dx = bx + q % 3, dy = by + q / 3;
						use[q] = dx < ntx && dy < nty && (!((outside >> q) & 1u) || dx == keep_dx);
						slot[q] = 0;
						if (use[q])
							slot[q] = atomicAdd(&w.tri_cnt[(ty0 + dy) * p.L.tiles_x + tx0 + dx], 1u);
					}
#pragma unroll
					for (int q = 0; q < 9; q++)
						if (use[q])
						{
							const int tile = (ty0 + by + q / 3) * p.L.tiles_x + tx0 + bx + q % 3;
							place_in_tile(w.tri_list, K_TRI, w.tri_pool, p.L.tri_pool_cap, &w.hdr->tri_spill[cur], tile, (uint32_t)k, slot[q]);
						}
				}
		} while (false);
		DR_WAVE_PHASE_T(4);
		bin_large(big, hp, btx0, bty0, bntx, bnty, bprim);
		return;
	}
	// nothing is written for the ~97 % of slots that are not silhouette edges: records are only reached through the
	// tile lists, and finalize_kernel works from the same flags
	const int slot = compact_flagged_slots(p, s.edgeflags, pw.index);
	DR_WAVE_PHASE(1); // flags compacted
	{
		bool big = false;
		double hp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
		int btx0 = 0, bty0 = 0, bntx = 0, bnty = 0, bprim = 0;
		do
		{
			if (slot < 0)
				break;
			const int k = slot / 3, n = slot - 3 * k;
			TriInputs t;
			EdgeRec e;
			EdgeRec &eout = w.edge_rec[slot];
			if (load_triangle(s, k, t, true))
			{ // invalid indices (reported by the triangle's own thread)
				eout.kind = KIND_NONE;
				break;
			}
			DR_WAVE_PHASE(2); // inputs arrived (?)
			// (the finalize inputs go straight to memory: kept in registers until the record is complete they cost the kernel a
			// wave per SIMD; those of an edge that turns out not to be drawn are never read)
			setup_edge_only(s, t, k, n, e, w.edge_planes + (size_t)slot * 3 * s.P, &w.edge_fin[slot]);
			DR_WAVE_PHASE(3); // record computed
			if (e.kind == KIND_NONE)
			{
				eout.kind = KIND_NONE;
				break;
			}
			for (int i = 0; i < 7; i++)
				e.pad0[i] = 0;
			eout = e;
			DR_WAVE_PHASE(4); // record stored
			if (e.x_begin > e.x_end || e.y_begin > e.y_end)
				break;
			const double band[12] = {e.x2b[0], e.x2b[1], e.x2b[2], e.x2b[3], e.x2b[4], e.x2b[5], e.x2t[0], e.x2t[1], e.x2t[2],
									 -e.x2t[0], -e.x2t[1], 1 - e.x2t[2]}; // the four half-planes of the band, H.h:1418-1435
			const int tx0 = e.x_begin / TILE, ty0 = e.y_begin / TILE, ntx = e.x_end / TILE - tx0 + 1, nty = e.y_end / TILE - ty0 + 1;
			if (((ntx + 2) / 3) * ((nty + 2) / 3) > COOP_BLOCKS)
			{
				big = true;
#pragma unroll
				for (int i = 0; i < 12; i++)
					hp[i] = band[i];
				btx0 = tx0, bty0 = ty0, bntx = ntx, bnty = nty, bprim = slot;
				break;
			}
			for (int by = 0; by < nty; by += 3)
				for (int bx = 0; bx < ntx; bx += 3)
				{
					uint32_t got[9];
					bool use[9];
					const uint32_t outside = tiles3x3_outside_halfplanes<4>(band, tx0 + bx, ty0 + by);
#pragma unroll
					for (int q = 0; q < 9; q++)
					{
						const int dx = bx + q % 3, dy = by + q / 3;
						use[q] = dx < ntx && dy < nty && !((outside >> q) & 1u);
						got[q] = 1;
						if (use[q])
							got[q] = atomicAdd(&w.edge_cnt[(ty0 + dy) * p.L.tiles_x + tx0 + dx], 1u);
					}
#pragma unroll
					for (int q = 0; q < 9; q++)
						if (use[q])
						{
							const int tile = (ty0 + by + q / 3) * p.L.tiles_x + tx0 + bx + q % 3;
							place_in_tile(w.edge_list, K_EDGE, w.edge_pool, p.L.edge_pool_cap, &w.hdr->edge_spill[cur], tile, (uint32_t)slot, got[q]);
						}
				}
		} while (false);
		DR_WAVE_PHASE(5); // own binning done
		bin_large(big, hp, btx0, bty0, bntx, bnty, bprim);
	}
}

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

// ---------------------------------------------------------------------------------- forward raster, LDS-staged fast path
//
// Same arithmetic as raster_fwd_kernel, restructured for latency: the tile's primitives are fetched with ONE batched
// load (ids -> 128-byte records + planes, 16 B per lane) into LDS instead of one dependent global round trip per
// primitive; the reference's scanline spans (two double divisions each, H.h:864-906) are computed once per
// (primitive, row) by lane = primitive_slot * 8 + row -- not once per pixel -- and exchanged as 8-bit column masks;
// the depth test and shading then read plane coefficients as LDS broadcasts.  Handles nb_colors <= 4 without
// antialiase_error; everything else runs on raster_fwd_kernel.

constexpr int TB = 16; // triangles (or edges) staged per batch: small, so that LDS never limits the number of resident waves

struct alignas(16) WaveLds
{
	TriRec rec[TB];			   // EdgeRec has the same size and is staged in the same place
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
template <class Rec>
__device__ __forceinline__ void stage_batch(WaveLds &S, const Rec *recs, const double *planes, int P, int nb, int lane)
{
	static_assert(TB == 16, "two record pieces and three plane doubles per lane");
	const int piece = lane & 7;
	const int np = 3 * P, total = nb * np; // np = 9 or 12
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

#ifdef DR_FWD_TRACE
#define DR_TRACE_ARGS , uint32_t *ftr, uint64_t ftr0
#define DR_TRACE_PASS , ftr, ftr0
#define DR_BTRACE(i)                                                                                                         \
	if (ftr[i] == 0)                                                                                                         \
	ftr[i] = (uint32_t)(__builtin_readcyclecounter() - ftr0)
#else
#define DR_TRACE_ARGS
#define DR_TRACE_PASS
#define DR_BTRACE(i)
#endif
template <bool TEX>
__device__ __forceinline__ void tri_batch(const KParams &p, WaveLds &S, int nb, int lane, int x0, int y0, bool inb, PixState &st DR_TRACE_ARGS)
{
	DR_BTRACE(8); // records staged (first batch)
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
			const TriRec &rec = S.rec[j];
			if (DR_ABLATE & 512)
				m = 0xffu;
			else if (rec.kind != KIND_NONE)
			{
				// A row lies in one half of the triangle (above or below its middle vertex), so one span (two divisions, not
				// four) per (triangle, row); only the non-strict fill rule puts the middle-vertex row in both halves.
				const int yy = y0 + r;
				const bool in0 = yy >= rec.y_begin[0] && yy <= rec.y_end[0], in1 = yy >= rec.y_begin[1] && yy <= rec.y_end[1];
				int xb, xe;
				tri_half_span(rec, in0 ? 0 : 1, yy, W, H, strict, xb, xe);
				m = column_mask(xb, xe, x0);
				if (DR_ABLATE & 16384)
				{ // measurement: the span arithmetic a second time (its cost = the difference in instruction counts)
					int yy2 = yy;
					asm volatile("" : "+v"(yy2));
					int xb2, xe2;
					tri_half_span(rec, in0 ? 0 : 1, yy2, W, H, strict, xb2, xe2);
					m &= column_mask(xb2, xe2, x0);
				}
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
	DR_BTRACE(9); // spans computed
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
		double Z = plane_at(S.rec[j].xZ, x, y);
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
	DR_BTRACE(10); // depth test done
	if (jbest >= 0)
	{ // per-lane reads of the winner's record and planes (LDS, a few distinct slots per tile)
		st.slot = jbest;
		const int kind = S.rec[jbest].kind;
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
	const EdgeRec *erec = (const EdgeRec *)S.rec;
#pragma unroll
	for (int q = 0; q < TB / 8; q++)
	{
		const int j = q * 8 + (lane >> 3), r = lane & 7;
		uint32_t m = 0;
		if (j < nb)
		{
			const EdgeRec &e = erec[j];
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
											  const Tap &tap, double L, double *tab, uint32_t *own);

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
constexpr int SCAN_BLOCK = 256, WORK_CHUNK = DR_WORK_CHUNK;
// One tile workgroup in `heavy_share` walks the list of the many-primitive tiles (the head of the grid: dispatched first).  One in
// eight, unless the head of all views together would then take more than ~40 % of the chip's wave slots (5 120 at five waves per
// SIMD): with every slot of the first dispatch round on a 25 - 50 us tile the short tiles -- whose arithmetic hides those tiles'
// round trips -- start late.  Measured on the 8-view benchmark step: 1/8 0.183 ms, 1/12 0.1775, 1/16 0.1767, 1/24 0.1784; on one
// 2048^2 view (2 048 head workgroups at 1/8) 1/16 costs 4 %; on one 1024^2 view 1/2 0.0775, 1/4 0.0772, 1/8 0.0809, 1/16 0.090 ms.
#ifndef DR_HEAVY_SHARE
#define DR_HEAVY_SHARE 0 // measurement builds: a fixed share
#endif
__host__ inline int heavy_share_for(int n_views, int tile_blocks)
{
	if (DR_HEAVY_SHARE)
		return DR_HEAVY_SHARE;
	// ~2 048 head workgroups over all views (40 % of the wave slots), the share a power of two between 1/4 and 1/16
	const long long want = ((long long)n_views * tile_blocks + 2047) / 2048;
	int share = 4;
	while (share < 16 && share < want)
		share *= 2;
	return share;
}

__global__ __launch_bounds__(SCAN_BLOCK) void tile_scan_kernel(KParams p)
{
	// classes compacted by this kernel: 0 many-primitive tiles (front of the work list), 1 the other non-empty tiles (back of it),
	// 2 .. 4 the three lists of edge tiles, 5 every edge tile (its rank is the tile's slot in edge_sweep)
	constexpr int NCLS = 3 + EDGE_LISTS;
	__shared__ uint32_t s_cnt[NCLS][SCAN_BLOCK / 64];
	__shared__ uint32_t s_base[NCLS];
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
		idb = ntri > 4 ? ids[1] : ida;
		idc = ntri > 8 ? ids[2] : ida;
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
	const bool heavy = work && p.tile_blocks % (8 * WORK_CHUNK) == 0 && (ntri > (uint32_t)FIRST_PRIMS || nedge > (uint32_t)FIRST_PRIMS);
	const int elist = nedge == 0 ? -1 : (nedge <= (uint32_t)PRIO_EDGES ? 0 : (nedge <= (uint32_t)TB ? 1 : 2));
	unsigned long long m[NCLS];
	m[0] = __ballot(heavy);
	m[1] = wm & ~m[0];
#pragma unroll
	for (int c = 0; c < EDGE_LISTS; c++)
		m[2 + c] = __ballot(elist == c);
	m[2 + EDGE_LISTS] = __ballot(nedge > 0);
	const unsigned long long below = (1ull << lane) - 1ull;
	if (lane < NCLS)
	{
		unsigned long long mine = 0;
#pragma unroll
		for (int c = 0; c < NCLS; c++)
			mine = lane == c ? m[c] : mine;
		s_cnt[lane][wave] = (uint32_t)__popcll(mine);
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
		sweep_slot = (nedge <= (uint32_t)EMAX && !p.persp) ? slot_word : 0u;
		static_assert(EDGE_LISTS == 3, "select below");
		w.edge_tiles[(size_t)elist * p.L.ntiles + position(2 + elist, elist == 0 ? m[2] : (elist == 1 ? m[3] : m[4]))] = (uint32_t)tile;
	}
	if (valid)
		w.edge_saved[tile] = nedge | (sweep_slot ? SWEEP_SAVED : 0u);
	if (work)
	{
		WorkEntry &e = heavy ? w.work_list[position(0, m[0])] : w.work_list[(uint32_t)p.L.ntiles - 1u - position(1, m[1])];
		uint4 *out = (uint4 *)&e;
		out[0] = make_uint4((uint32_t)tile, ntri, nedge, sweep_slot);
		out[1] = ida;
		out[2] = idb;
		out[3] = idc;
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
			int ph = C == 3 ? (piece * E) % 3 : ((piece * E) & (C - 1)); // channel of the piece's first element
#pragma unroll
			for (int j = 0; j < E; j++)
			{
				v[j] = (PixT)(ph == 0 ? bgc[0] : (ph == 1 ? bgc[1] : (ph == 2 ? bgc[2] : bgc[3])));
				ph = ph + 1 == C ? 0 : ph + 1;
			}
			return v;
		};
		if (n >= 64 && !bgi)
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

// A fit step has two latency-bound kernels after the forward raster (edge tiles, finalize) whose wave slots and store bandwidth
// are mostly idle: the background fill rides on them as extra workgroups instead of a kernel of its own on a forked stream --
// the fork / join event packets cost the caller's stream two bubbles of ~7 us per step (rocprofv3 kernel trace: scan -> forward,
// finalize -> next set-up).  Word wi of a view goes to the kernels that take part by parity.
// When both kernels take part, FILL_EDGE_NUM of every FILL_DEN consecutive words go to the edge-tile kernel, the others to finalize.
#ifndef DR_FILL_EDGE_NUM
#define DR_FILL_EDGE_NUM 1
#endif
#ifndef DR_FILL_DEN
#define DR_FILL_DEN 2
#endif
constexpr int FILL_EDGE_NUM = DR_FILL_EDGE_NUM, FILL_DEN = DR_FILL_DEN;
static_assert(FILL_EDGE_NUM > 0 && FILL_EDGE_NUM < FILL_DEN, "both kernels get some");
__host__ __device__ inline int fill_share(int fill_mode, int bit, int nwords)
{ // bitmap words per view the kernel `bit` (0 edge tiles, 1 finalize) fills
	if (!(fill_mode & (1 << bit)))
		return 0;
	if (fill_mode != 3)
		return nwords;
	const int full = nwords / FILL_DEN, rest = nwords - full * FILL_DEN; // whole groups + a partial one
	const int edge = full * FILL_EDGE_NUM + (rest < FILL_EDGE_NUM ? rest : FILL_EDGE_NUM);
	return bit == 0 ? edge : nwords - edge;
}
// workgroups (edge kernel: per view, along grid y, limited to 65535) that stream a share of n words: one word each up to a cap,
// beyond it (frames of more than ~4 M tiles) every workgroup takes several
__host__ __device__ inline int fill_share_blocks(int n) { return n < 32768 ? n : 32768; }
__device__ __forceinline__ void fill_share_word(const KParams &p, int bit, int view, int i, int lane)
{ // the i-th word of the share of kernel `bit`
	const int per = bit == 0 ? FILL_EDGE_NUM : FILL_DEN - FILL_EDGE_NUM; // words of a group that are this kernel's
	const int wi = p.fill_mode == 3 ? (i / per) * FILL_DEN + (bit == 0 ? 0 : FILL_EDGE_NUM) + i % per : i;
	if (wi >= p.L.nwords)
		return;
	if (p.pix_f64)
		fill_word<double>(p, view, wi, lane, 0);
	else
		fill_word<float>(p, view, wi, lane, 0);
}

// Adjoint of pass 1 for a tile whose triangles are ONE staged batch (S.ids[0 .. ntri), the usual case), untextured: the moments
//   M[owner][3 q + m] = sum over the owner's pixels of  g_q * {x, y, 1}[m]
// are a small dense contraction over the 64 pixels of the tile -- (one-hot owner matrix)^T (64 x 16) times the 64 x 12 matrix of
// per-pixel values -- and the forward raster is bound by vector-ALU issue while its matrix cores idle: sixteen
// v_mfma_f64_16x16x4_f64 (K = 4 pixels each) can replace the segmented scans, run tables and merge loops of owner_adjoint
// (about half of its vector instructions).  Operand layout (cdna_hip_programming.md, checked by tools/probes/mfma_f64_probe.hip): lane l
// feeds A[l & 15][l >> 4] and B[l >> 4][l & 15], and receives D[(l >> 4) + 4 r][l & 15] in register r.  The per-pixel values
// cross lanes through the (idle) staging area, 32 pixels at a time; the one-hot entries are exact, so only the order of the
// additions differs from the scan (both differ from the reference's row-by-row sums; tolerance of the parity tests 1e-8).
// MEASURED AND NOT USED (the product is built with DR_OWNER_MFMA = 0; tools/build_variants.sh can build the other): parity green
// (all 205 GPU tests), 100 fewer vector instructions per tile -- and the forward raster 14 us SLOWER (85 -> 99 us per 8-view
// launch): on MI355X the f64 matrix rate equals the f64 vector rate (78.6 TFLOP/s), a 16 x 16 x 4 f64 MFMA holds its SIMD for
// ~64 cycles, and 16 of them (of whose 16 k multiply-adds ~2.5 k are useful: 3 - 4 owners x 12 moments x 64 pixels) cost more
// issue time than the ~100 vector instructions they replace.
#ifndef DR_OWNER_MFMA
#define DR_OWNER_MFMA 0
#endif
typedef double mfma_f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void owner_adjoint_mfma(const KParams &p, const ViewPtrs &w, WaveLds &S, int lane, double x, double y, int slot, int ntri,
												   const double *g)
{
	static_assert(sizeof(S.rec) + sizeof(S.planes) >= 32 * 12 * sizeof(double) && sizeof(S.cover) >= 64 && TB == 16, "LDS reuse");
	const int C = p.C, nm = 3 * p.L.P;
	double *bm = (double *)&S.rec[0]; // [32 pixels][12]: g_q x, g_q y, g_q of planes q = 0 .. 3
	uint8_t *jb = &S.cover[0][0];	  // [64 pixels]: slot of the owner, 0xff: none
	const int col = lane & 15, kq = lane >> 4;
	jb[lane] = (uint8_t)(slot < 0 ? 0xff : slot);
	mfma_f64x4 acc = {0, 0, 0, 0};
#pragma unroll
	for (int h = 0; h < 2; h++)
	{
		lds_sync();
		if ((lane >> 5) == h)
		{
			double *row = bm + (lane & 31) * 12;
#pragma unroll
			for (int q = 0; q < CH; q++)
			{
				const double v = q < C ? g[q] : 0.0;
				row[3 * q] = v * x;
				row[3 * q + 1] = v * y;
				row[3 * q + 2] = v;
			}
		}
		lds_sync();
#pragma unroll
		for (int st = 0; st < 8; st++)
		{
			const int pl = 4 * st + kq; // pixel of this lane's A / B entries, inside the half
			const double a = jb[32 * h + pl] == (uint8_t)col ? 1.0 : 0.0;
			const double b = col < 12 ? bm[pl * 12 + col] : 0.0;
			acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
		}
	}
#pragma unroll
	for (int r = 0; r < 4; r++)
	{ // owner slot kq + 4 r, moment `col`: the 3P moments of an owner are contiguous (one atomic instruction per four owners)
		const int i = kq + 4 * r;
		const double v = acc[r];
#if !(DR_ABLATE & 128)
		if (i < ntri && col < nm && v != 0)
			atomic_add_f64(w.tri_acc + (size_t)S.ids[i] * nm + col, v);
#endif
	}
}

// Grid of the staged forward (1-D, one wavefront per workgroup).  Workgroup b: view (b / 8) % n_views,
// q = (b / 8 / n_views) * 8 + b % 8 in [0, p.tile_blocks); it walks the entries rank(q), rank(q) + tile_blocks, ... of the
// view's work list (usually one or two).  rank() deals the list to the XCDs in chunks of 64 consecutive entries (workgroup b
// runs on XCD b % 8; consecutive entries are neighbouring tiles, which share triangle records and should share an L2).

__host__ __device__ inline int fwd_tile_blocks(int ntiles)
{ // workgroups per view that walk the work list: a quarter of the tiles (about a third of a frame's tiles hold primitives)
	const int unit = 8 * WORK_CHUNK;
#ifndef DR_TILE_DIV
#define DR_TILE_DIV 4
#endif
	const int g = ((ntiles / DR_TILE_DIV + unit - 1) / unit) * unit;
	return g > 0 && g <= ntiles ? g : ntiles; // tiny frames: one workgroup per tile, plain order
}

// FUSED: the forward of a fit step.  The loss is L = sum (image - obs)^2, so dL/dimage is known the moment a pixel is
// resolved: tiles without silhouette edges back-propagate into their owners' accumulators right here (no second pass over the
// frame, no owner buffer round trip -- the owner ids of those tiles are not even written); tiles with edges are left to
// raster_bwd_edge_kernel.
// Waves per SIMD the staged forward is compiled for: without texture code it fits five (96 registers), with it four
// (tools/build_variants.sh builds the neighbours: -DDR_FWD_WAVES=n forces n for both).
#ifndef DR_FWD_WAVES
#define DR_FWD_WAVES (TEX ? 4 : 5)
#endif
template <class PixT, bool FUSED, bool TEX>
__global__ __launch_bounds__(64, DR_FWD_WAVES) void raster_fwd_fast_kernel(KParams p)
{
	DR_WAVE_TRACE_SCOPE(2);
	__shared__ WaveLds s_lds[1];
	__shared__ EdgeSort s_es[1];
#ifdef DR_FWD_TRACE
	// per-tile phase timing (tools/fwd_trace.py): eight counters over the first row of the tile in the z buffer
	uint32_t ftr[16] = {0x7fc0f00du, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define DR_FTRACE(i) ftr[i] = (uint32_t)(__builtin_readcyclecounter() - ftr0)
#else
#define DR_FTRACE(i)
#endif
	constexpr int wave = 0;
	const int lane0 = threadIdx.x & 63;
	const int G = p.tile_blocks;
	const long long b = blockIdx.x;
	int view, q;
	const bool chunked = G % (8 * WORK_CHUNK) == 0;
	if (chunked)
	{
		view = (int)((b >> 3) % p.n_views);
		q = (int)((b >> 3) / p.n_views) * 8 + (int)(b & 7);
	}
	else
	{
		view = (int)(b % p.n_views);
		q = (int)(b / p.n_views);
	}
	const ViewPtrs w = view_ptrs(p, view);
	const int W = p.W, H = p.H, C = p.C, P = p.L.P;
	const bool persp = p.persp;
	const PixT *texture = (const PixT *)p.texture;
	WaveLds &S = s_lds[wave];
	// The first G / p.heavy_share workgroups of a view walk the many-primitive tiles (front of the list), the others the rest (from
	// the back): the index of a workgroup's entry does not depend on the counts, so the counts, the entry header and the
	// entry's triangle ids are all requested at once.
	// (tiny frames -- G not a multiple of 512 -- have one class only: the scan kernel lists every tile as "other")
	const int Gh = chunked ? G / p.heavy_share : 0;
	const bool heavy_list = q < Gh;
	const int qq = heavy_list ? q : q - Gh, stride = heavy_list ? Gh : G - Gh;
	const bool chunk_here = chunked && stride % (8 * WORK_CHUNK) == 0;
	uint32_t rank = chunk_here ? (uint32_t)((((qq >> 3) / WORK_CHUNK) * 8 + (qq & 7)) * WORK_CHUNK + (qq >> 3) % WORK_CHUNK) : (uint32_t)qq;
	const uint32_t n_work = w.hdr->work_count[heavy_list ? 0 : 1];
	for (; rank < n_work; rank += (uint32_t)stride)
	{
#ifdef DR_FWD_TRACE
		const uint64_t ftr0 = __builtin_readcyclecounter();
		ftr[8] = ftr[9] = ftr[10] = ftr[11] = ftr[12] = 0;
#endif
		// (the lane index is made opaque per iteration: otherwise every lane-dependent address of the body is hoisted out of the
		// loop and kept -- spilled -- in registers across it: + 150 VGPRs for a loop that usually runs once or twice)
		int lane = lane0;
		asm volatile("" : "+v"(lane));
		const WorkEntry &entry = w.work_list[heavy_list ? rank : (uint32_t)p.L.ntiles - 1u - rank];
		const uint32_t ids12 = entry.ids[lane < ENTRY_IDS ? lane : 0];
		const int tile = uniform((int)entry.tile), ntri = uniform((int)entry.ntri), nedge = uniform((int)entry.nedge);
		const uint32_t sweep_slot = (uint32_t)uniform((int)entry.sweep_slot);
		const int tx = tile % p.L.tiles_x, ty = tile / p.L.tiles_x;
		const int x0 = tx * TILE, y0 = ty * TILE;
		const int px = x0 + (lane & 7), py = y0 + (lane >> 3);
		const bool inb = px < W && py < H;
		const size_t pix = (size_t)py * W + px;
		const size_t vpix = (size_t)view * H * W + pix;
		const double x = px, y = py;
		// more than ENTRY_IDS triangles: the rest of the inline list (one more round trip, one tile in ten)
		const uint32_t list_entry = ntri <= ENTRY_IDS ? ids12 : w.tri_list[(size_t)tile * K_TRI + (lane & (K_TRI - 1))];
		{
		{
		PixT ob[CH] = {0, 0, 0, 0};
		if (FUSED && ntri > 0 && nedge == 0 && inb)
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
#ifdef DR_FWD_TRACE
		ftr[1] = (uint32_t)ntri | ((uint32_t)nedge << 16);
#endif
		DR_FTRACE(2); // counters arrived
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
				tri_batch<TEX>(p, S, nb, lane, x0, y0, inb, st DR_TRACE_PASS);
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
							tri_batch<TEX>(p, S, TB, lane, x0, y0, inb, st DR_TRACE_PASS);
							fill = 0;
						}
					}
				}
				if (fill > 0)
				{
					lds_sync();
					stage_batch(S, w.tri_rec, w.tri_planes, P, fill, lane);
					lds_sync();
					tri_batch<TEX>(p, S, fill, lane, x0, y0, inb, st DR_TRACE_PASS);
				}
			}
		}
		DR_FTRACE(3); // pass 1 done
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
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				col[cc] = cc < C ? textured_channel(texture, tap, cc) * L : 0.0;
		}
		// ---- pass 2: edges far -> near, TB at a time (H.h:2839-2900)
		int n_edges = 0;
		if (nedge > 0)
			n_edges = gather_sorted_edges(s_es[wave], w, p, tile, nedge, lane);
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
			const EdgeRec *erec = (const EdgeRec *)S.rec;
			for (int first = 0; first < n_edges; first += TB)
			{
				const int nb = n_edges - first < TB ? n_edges - first : TB;
				const uint32_t ecov = stage_edge_batch(S, s_es[wave], w, P, first, nb, lane, x0, y0, W, inb);
				uint32_t drawn_batch = 0;
				for (int j = 0; j < nb; j++)
				{
					const bool c = (ecov >> j) & 1u;
					if (__ballot(c) == 0)
						continue;
					const EdgeRec &e = erec[j];
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
#pragma unroll
						for (int cc = 0; cc < CH; cc++)
							if (cc < C)
							{
								const double A = edge_channel<PixT, TEX>(e, ep, texture, etap, eL, cc, x, y, persp, Ze);
								col[cc] *= Tr;
								col[cc] += (1 - Tr) * A;
							}
					}
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
		else if (n_edges < 0)
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
#pragma unroll
					for (int cc = 0; cc < CH; cc++)
						if (cc < C)
						{
							const double A = edge_channel<PixT, TEX>(e, ep, texture, etap, eL, cc, x, y, persp, Ze);
							col[cc] *= Tr;
							col[cc] += (1 - Tr) * A;
						}
				}
			}
		}
		DR_FTRACE(4); // colour resolved, edges blended
		// ---- one write per pixel
		if (inb && !(DR_ABLATE & 4))
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
			// a fused forward back-propagates through a tile without edges right below: nobody reads its owner ids again
			if (!FUSED || nedge > 0)
				__builtin_nontemporal_store(pack_owner(st.kbest, st.kind), w.face_id + pix);
		}
		DR_FTRACE(5); // frame stores issued
		if (FUSED && nedge == 0 && __ballot(st.kbest >= 0) != 0)
		{ // same residual as raster_bwd_fast_kernel forms from the stored frame: the colour is rounded to the pixel type first
			double g[CH];
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				asm volatile("" : "+v"(ob[cc])); // the observation stays in the pixel type until here: converted to double right
												 // after its load, it was spilled (four doubles per lane) through the whole of pass 1
#pragma unroll
			for (int cc = 0; cc < CH; cc++)
				g[cc] = (cc < C && inb) ? 2 * ((double)(PixT)col[cc] - (double)ob[cc]) : 0.0;
			lds_sync();
			if (DR_ABLATE & 256)
			{
			}
			else if (!TEX && DR_OWNER_MFMA && ntri <= TB)
				owner_adjoint_mfma(p, w, S, lane, x, y, st.kbest >= 0 ? st.slot : -1, ntri, g);
			else
				owner_adjoint<PixT, TEX>(p, w, lane, x, y, st.kbest, st.kbest >= 0 ? st.kind : (int)KIND_NONE, g, tap, L, (double *)&S.rec[0],
									(uint32_t *)&S.cover[0][0]);
		}
#ifdef DR_FWD_TRACE
		DR_FTRACE(6); // adjoint of pass 1 issued
		if (lane < 16 && p.zbuf)
		{
			uint32_t v = 0;
			for (int i = 0; i < 16; i++)
				v = lane == i ? ftr[i] : v;
			((uint32_t *)p.zbuf)[(size_t)view * H * W + (size_t)(y0 + (lane >> 3)) * W + x0 + (lane & 7)] = v;
		}
#endif
		}
		}
		lds_sync(); // the next tile of this wavefront reuses the staging area
	}
	if (q == 0 && threadIdx.x == 0)
		close_epoch(p, w, FUSED);
}

// ------------------------------------------------------------------------------------------------ backward raster

// adds  sum over the wave of  v * [x, y, 1]  to acc[0..2]
__device__ __forceinline__ void add_moments(double *acc, double v, double x, double y, int lane)
{
	double mx = wave_sum(v * x), my = wave_sum(v * y), m1 = wave_sum(v);
	if (lane == 0)
	{
		if (mx != 0)
			atomic_add_f64(acc + 0, mx);
		if (my != 0)
			atomic_add_f64(acc + 1, my);
		if (m1 != 0)
			atomic_add_f64(acc + 2, m1);
	}
}

template <class PixT>
__device__ __forceinline__ void texture_scatter(PixT *texture_b, const Tap &tap, int c, const double wgt[4])
{
#pragma unroll
	for (int q = 0; q < 4; q++)
		if (wgt[q] != 0)
			unsafeAtomicAdd(texture_b + tap.idx[q] + c, (PixT)wgt[q]);
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
								texture_scatter(texture_b, etap, c, wgt);
						}
						else // H.h:2579-2588, with the row fold the reference forgot (defect D2) restored
							A_B = 2 * (interp_channel(ep, c, x, y, false, 0.0) - (double)obs[c]) * Err_B;
					}
					if (e.kind != KIND_TEXTURED || !TEX)
						add_moments(eacc + 3 * c, A_B, x, y, lane);
				}
				if (e.kind == KIND_TEXTURED && TEX)
				{
					add_moments(eacc + 0, (hit && !etap.out[0]) ? e_B[0] : 0.0, x, y, lane);
					add_moments(eacc + 3, (hit && !etap.out[1]) ? e_B[1] : 0.0, x, y, lane);
					add_moments(eacc + 6, L_B, x, y, lane);
				}
				add_moments(eacc + 3 * P, T_B, x, y, lane);
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
										 : 2 * ((double)((const PixT *)p.image_in)[vpix * C + c0 + j] - (double)((const PixT *)p.obs)[vpix * C + c0 + j]);
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
									texture_scatter(texture_b, etap, c, wgt);
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
							add_moments(eacc + 3 * c, A_B, x, y, lane);
					}
					if (e.kind == KIND_TEXTURED && TEX)
					{
						add_moments(eacc + 0, (hit && !etap.out[0]) ? e_B[0] : 0.0, x, y, lane);
						add_moments(eacc + 3, (hit && !etap.out[1]) ? e_B[1] : 0.0, x, y, lane);
						add_moments(eacc + 6, L_B, x, y, lane);
					}
					add_moments(eacc + 3 * P, T_B, x, y, lane);
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
						texture_scatter(texture_b, tap, c, wgt);
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
						add_moments(acc + 3 * (c0 + j), mine ? g[j] : 0.0, x, y, lane);
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
		add_moments(acc + 0, (mine && !tap.out[0]) ? own_e_B[0] : 0.0, x, y, lane);
		add_moments(acc + 3, (mine && !tap.out[1]) ? own_e_B[1] : 0.0, x, y, lane);
		add_moments(acc + 6, mine ? own_L_B : 0.0, x, y, lane);
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

// ---------------------------------------------------------------------------------- backward raster, LDS-staged fast path
//
// nb_colors <= 4, no antialiase_error, at most K_EDGE edges in the tile (other tiles call bwd_tile_generic).
// Differences from the generic tile: edges are staged / ranked / span-tested exactly as in raster_fwd_fast_kernel, and the
// segmented reductions "sum over the pixels of a primitive" are done with LDS atomics (ds_add_f64, one slot per distinct
// primitive of the tile) followed by ONE global atomic per (primitive, moment), issued by 64 lanes in parallel -- instead
// of a 64-lane butterfly per moment and primitive.

constexpr int NMOM = 12; // moments per owner slot: 3 per channel (or 9 for a textured owner)

struct alignas(16) BwdLds
{
	EdgeRec rec[TB];
	double planes[TB * 12];
	uint32_t ids[TB];
	uint8_t cover[TILE][TB];
	uint32_t order[TB];
};

__device__ __forceinline__ void lds_add(double *slot, double v)
{
	if (v != 0)
		unsafeAtomicAdd(slot, v);
}

constexpr int RUNS = 32; // run totals flushed per pass: 32 x 12 doubles fit in the (by then idle) record staging area of the wave
static_assert(RUNS * NMOM * sizeof(double) <= sizeof(WaveLds::rec) + sizeof(WaveLds::planes) && RUNS * 4 <= sizeof(WaveLds::cover), "LDS reuse");

// Adjoint of pass 1 for one tile: g = dL/d(colour written by pass 1) of this lane's pixel, owned by triangle `owner`.
// tab (RUNS * NMOM doubles) and own (RUNS words) are LDS scratch of this wave.  All 64 lanes must call it.
template <class PixT, bool TEX>
__device__ __forceinline__ void owner_adjoint(const KParams &p, const ViewPtrs &w, int lane, double x, double y, int owner, int kind, const double *g,
											  const Tap &tap, double L, double *tab, uint32_t *own)
{
	const int C = p.C, P = p.L.P;
	const PixT *texture = (const PixT *)p.texture;
	PixT *texture_b = (PixT *)p.texture_b;
	// per-pixel adjoint of the owner's (up to four) attribute planes; its moments  sum v * [x, y, 1]  over the owner's pixels
	// are what the per-triangle finalize needs
	double val[CH] = {0, 0, 0, 0};
	// Texture gradient.  The taps of the 64 pixels of a tile fall into a small window of texels (a magnified texture: a
	// dozen texels for 768 contributions), and atomics to one address serialise in the L2 at ~80 ns each: when the window
	// fits the LDS scratch, the contributions are summed there (ds_add_f64) and each touched texel leaves with ONE global
	// atomic -- "per-tile LDS partials before a single atomicAdd".
	const bool textured = kind == KIND_TEXTURED && TEX;
	int fu = 0, fv = 0, win_u0 = 0, win_v0 = 0, win_w = 0, win_h = 0;
	bool windowed = false;
	if (texture_b && __ballot(textured))
	{
		if (textured)
		{
			const int t0 = tap.idx[0] / C;
			fv = t0 / p.tex_w;
			fu = t0 - fv * p.tex_w;
		}
		int lo_u = textured ? fu : 0x7fffffff, lo_v = textured ? fv : 0x7fffffff, hi_u = textured ? fu : -1, hi_v = textured ? fv : -1;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1)
		{
			lo_u = min(lo_u, __shfl_xor(lo_u, d, 64));
			lo_v = min(lo_v, __shfl_xor(lo_v, d, 64));
			hi_u = max(hi_u, __shfl_xor(hi_u, d, 64));
			hi_v = max(hi_v, __shfl_xor(hi_v, d, 64));
		}
		win_u0 = lo_u, win_v0 = lo_v, win_w = hi_u - lo_u + 2, win_h = hi_v - lo_v + 2;
		windowed = win_w * win_h * C <= RUNS * NMOM;
		if (windowed)
		{
			lds_sync();
			for (int i = lane; i < win_w * win_h * C; i += 64)
				tab[i] = 0;
			lds_sync();
		}
	}
	if (textured)
	{ // H.h:1320-1353
		double L_B = 0, e_B[2] = {0, 0};
		const int wbase = ((fv - win_v0) * win_w + (fu - win_u0)) * C;
#pragma unroll
		for (int cc = 0; cc < CH; cc++)
			if (cc < C)
			{
				const double i00 = ldp(texture, tap.idx[0] + cc), i10 = ldp(texture, tap.idx[1] + cc);
				const double i01 = ldp(texture, tap.idx[2] + cc), i11 = ldp(texture, tap.idx[3] + cc);
				L_B += g[cc] * bilinear_mix(tap, i00, i10, i01, i11);
				double wgt[4];
				bilinear_mix_adjoint(tap, g[cc] * L, i00, i10, i01, i11, wgt, e_B);
				if (windowed)
				{
					lds_add(&tab[wbase + cc], wgt[0]);
					lds_add(&tab[wbase + C + cc], wgt[1]);
					lds_add(&tab[wbase + win_w * C + cc], wgt[2]);
					lds_add(&tab[wbase + win_w * C + C + cc], wgt[3]);
				}
				else if (texture_b)
					texture_scatter(texture_b, tap, cc, wgt);
			}
		val[0] = tap.out[0] ? 0.0 : e_B[0];
		val[1] = tap.out[1] ? 0.0 : e_B[1];
		val[2] = L_B;
	}
	if (windowed)
	{
		lds_sync();
		for (int i = lane; i < win_w * win_h * C; i += 64)
		{
			const double v = tab[i];
			if (v != 0)
			{
				const int texel = i / C, c = i - texel * C, jv = texel / win_w, ju = texel - jv * win_w;
				if (!(DR_ABLATE & 1048576)) // (measurement build: no texture-gradient atomics)
					unsafeAtomicAdd(texture_b + (size_t)C * ((win_u0 + ju) + (size_t)p.tex_w * (win_v0 + jv)) + c, (PixT)v);
			}
		}
		lds_sync(); // tab is reused for the run totals below
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
#if !(DR_ABLATE & 128)
			if (m < nm && acc != 0)
				atomic_add_f64(w.tri_acc + (size_t)o * nm + m, acc);
#endif
		}
		emask &= ~__ballot(sel);
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
#ifdef DR_TILE_TRACE
	uint32_t tr[8] = {0x7fc0beefu, (uint32_t)nedge, 0, 0, 0, 0, 0, 0};
	const uint64_t tr0 = __builtin_readcyclecounter();
#define DR_TRACE(i) tr[i] = (uint32_t)(__builtin_readcyclecounter() - tr0)
#else
#define DR_TRACE(i)
#endif
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
	DR_TRACE(2);
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
		if (p.image_b)
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
				g[cc] = (cc < C && inb) ? 2 * ((double)im[cc] - (double)ob[cc]) : 0.0;
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
				const EdgeRec &eq = S.rec[j];
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
		DR_TRACE(3);
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
		for (int b = b_hi; b >= b_lo; b--)
		{
			const int first = b * TB, nb = n_edges - first < TB ? n_edges - first : TB;
			if (b < nbatch - 1 || sweep_saved) // the records of pass A's last batch (if it ran) are still in LDS
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
				const EdgeRec &e = S.rec[r];
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
						for (int bb = 0; bb < EMAX / TB; bb++)
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

	DR_TRACE(4);
	if (EDGES && b_lo > 0)
		return; // the wavefront that ran batch 0 (the farthest edges) holds the gradient that reaches pass 1
	// ---- adjoint of pass 1: g now belongs to the triangle that owns the pixel
	owner_adjoint<PixT, TEX>(p, w, lane, x, y, owner, kind, g, tap, L, (double *)&S.rec[0], (uint32_t *)&S.cover[0][0]);
#ifdef DR_TILE_TRACE
	DR_TRACE(5);
	if (EDGES && lane < 8)
	{
		uint32_t v = 0;
		for (int i = 0; i < 8; i++)
			v = lane == i ? tr[i] : v;
		((uint32_t *)p.image_in)[((size_t)view * H * W + (size_t)y0 * W + x0) * C + lane] = v; // C == 4: the first two pixels of the tile
	}
#endif
#undef DR_TRACE
}

template <class PixT, bool TEX>
__global__ __launch_bounds__(64, 6) void raster_bwd_fast_kernel(KParams p)
{ // (two-call path) the forward's work list of the non-empty tiles, walked exactly as raster_fwd_fast_kernel walks it (same grid,
  // same entry of the list for the same workgroup); the tiles with silhouette edges are left to raster_bwd_edge_kernel.  One
  // wavefront per tile OF THE FRAME, which found out from the tile bitmap that two out of three had nothing to do, took ~51 us
  // per 8-view launch; this one ~43 (two-call step 0.254 -> 0.246 ms).
	__shared__ BwdLds s_lds;
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
		const WorkEntry &entry = w.work_list[heavy_list ? rank : (uint32_t)p.L.ntiles - 1u - rank];
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
template <class PixT, bool TEX>
__global__ __launch_bounds__(64, TEX ? 2 : DR_EDGE_OCC) void raster_bwd_edge_kernel(KParams p)
{ // persistent waves over the lists of tiles that hold silhouette edges (built by tile_scan_kernel).  Grid (views, waves):
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
	const uint32_t n_short = w.edge_tile_cnt[0], n_long = (DR_ABLATE & 65536) ? 0u : w.edge_tile_cnt[CNT_STRIDE],
				   n_multi = (DR_ABLATE & 32768) ? 0u : w.edge_tile_cnt[2 * CNT_STRIDE] * CHUNKS; // (measurement builds: without the multi-batch / the 9-16-edge tiles)
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

// ------------------------------------------------------------------------------------------------------- finalize

__global__ __launch_bounds__(PRIM_BLOCK, DR_PRIM_WAVES) void finalize_kernel(KParams p)
{ // same split as setup_bin_kernel: triangle blocks, then edge-slot blocks compacted to the flagged slots.
  // (Lists of the front-facing triangles / drawn edges compacted by the set-up kernel were tried: a quarter as many wavefronts,
  // all lanes busy -- and 32 -> 41 us: the kernel is a chain of dependent round trips, fewer wavefronts overlap fewer of them.)
	DR_WAVE_TRACE_SCOPE(1);
	const int fill_n = fill_share(p.fill_mode, 1, p.L.nwords), fill_blocks = (p.n_views * fill_n + PRIM_BLOCK / 64 - 1) / (PRIM_BLOCK / 64);
	const int fb = DR_FILL_FIRST ? (int)blockIdx.x : (int)blockIdx.x - p.n_views * prim_blocks(p.T); // index among the fill workgroups
	if (DR_FILL_FIRST ? fb < fill_blocks : fb >= 0)
	{ // workgroups that stream the background of this kernel's share of the empty tiles (fill_share)
		const int gw = fb * (PRIM_BLOCK / 64) + (int)(threadIdx.x >> 6);
		if (fill_n > 0 && gw < p.n_views * fill_n)
			fill_share_word(p, 1, gw / fill_n, gw % fill_n, threadIdx.x & 63);
		return;
	}
#ifndef DR_FIN_EDGE_FIRST
#define DR_FIN_EDGE_FIRST 1 // (triangle blocks first: finalize 37.5 -> 43.5 us)
#endif
	const PrimWork pw = prim_work(p, DR_FIN_EDGE_FIRST, DR_FILL_FIRST ? fill_blocks : 0);
	const int view = pw.view;
	const bool tri_block = pw.tri;
	const SceneView s = scene_view(p, view);
	const ViewPtrs w = view_ptrs(p, view);
	const size_t es = p.vtx_f64 ? 8 : 4;
	GradView g;
	g.ij_b = (char *)p.ij_b + (size_t)view * p.V * 2 * es;
	g.colors_b = (char *)p.colors_b + (size_t)view * p.V * p.C * es;
	g.shade_b = (char *)p.shade_b + (size_t)view * p.V * es;
	g.uv_b = p.uv_b;
	const int P = s.P;
	if (tri_block)
	{
		const int k = pw.index * PRIM_BLOCK + threadIdx.x;
		if (k >= p.T)
			return;
		// the vertex indices are requested together with the flag (one memory round trip, not two): using them in the branch
		// condition keeps the compiler from sinking the loads below it (an index never has its top bit set: V < 2^31)
		const uint32_t flag = w.tri_flag[k];
		const uint32_t f0 = p.faces[3 * (size_t)k], f1 = p.faces[3 * (size_t)k + 1], f2 = p.faces[3 * (size_t)k + 2];
		double *acc = w.tri_acc + (size_t)k * 3 * P;
		if (!(flag & 4u) || (flag & 3u) == KIND_NONE || (int32_t)(f0 | f1 | f2) < 0)
			return; // culled triangles own no accumulators
		DR_WAVE_PHASE_T(1); // flags + indices arrived
		AtomicSink sink = {s, g, {f0, f1, f2}, {p.faces_uv[3 * (size_t)k], p.faces_uv[3 * (size_t)k + 1], p.faces_uv[3 * (size_t)k + 2]}};
		if (P <= 4)
		{ // a register copy of the accumulators: all twelve loads in flight together (read through the pointer, each plane's
		  // loads would wait behind the atomics of the plane before: they might alias)
			double la[12];
#pragma unroll
			for (int i = 0; i < 12; i++)
				la[i] = i < 3 * P ? acc[i] : 0.0;
			finalize_triangle<true>(s, k, (int)(flag & 3u), la, sink);
		}
		else
			finalize_triangle<false>(s, k, (int)(flag & 3u), acc, sink);
		// (merging the adjoints of the triangles of a wavefront that share a vertex in an LDS table before they leave -- a third
		// fewer atomic requests at the memory side -- was measured: 34 -> 35 us)
		DR_WAVE_PHASE_T(2); // arithmetic done, atomics issued
		for (int i = 0; i < 3 * P; i++)
			acc[i] = 0; // self-cleaning accumulators
		DR_WAVE_PHASE_T(3);
		return;
	}
	const int slot = compact_flagged_slots(p, s.edgeflags, pw.index);
	DR_WAVE_PHASE(1); // flags compacted
	if (slot >= 0)
	{
		// Record, finalize inputs and accumulators of the slot are all requested at once: ONE memory round trip before the
		// arithmetic.  The set-up kernel of this forward wrote the record's kind for EVERY flagged slot (KIND_NONE for an edge of
		// a back-facing triangle), so nothing read here is stale.
		const EdgeRec &er = w.edge_rec[slot];
		const int kind = er.kind;
		double *acc = w.edge_acc + (size_t)slot * (3 * P + 3);
		if (P <= 4)
		{
			double x2b[6], la[12], lt[3];
#pragma unroll
			for (int i = 0; i < 6; i++)
				x2b[i] = er.x2b[i];
			EdgeFin fin = w.edge_fin[slot];
#pragma unroll
			for (int i = 0; i < 12; i++)
				la[i] = i < 3 * P ? acc[i] : 0.0;
#pragma unroll
			for (int i = 0; i < 3; i++)
				lt[i] = acc[3 * P + i];
			// (empty statement that "uses" one value of every group: the loads are issued -- and waited for together -- before
			// the branch instead of being sunk below it, where each group would cost a round trip of its own)
			asm volatile("" : "+v"(x2b[0]), "+v"(la[0]), "+v"(lt[0]), "+v"(fin.V[0][0]));
			DR_WAVE_PHASE(2); // inputs arrived
			if (kind == KIND_NONE)
				return;
			if (fin.has_att)
				finalize_edge_fin(s, g, kind, x2b, fin, la, lt, DeviceAdd());
			else
				finalize_edge(s, g, slot / 3, slot % 3, er, acc, DeviceAdd());
		}
		else
		{
			if (kind == KIND_NONE)
				return;
			finalize_edge(s, g, slot / 3, slot % 3, er, acc, DeviceAdd());
		}
		DR_WAVE_PHASE(3); // arithmetic done, atomics issued
		for (int i = 0; i < 3 * P + 3; i++)
			acc[i] = 0;
		DR_WAVE_PHASE(4);
	}
}

// ------------------------------------------------------------------------------------------------------ host side

thread_local char g_error[256] = "";

int fail(const char *msg)
{
	snprintf(g_error, sizeof g_error, "%s", msg);
	return 1;
}

int check_hip(hipError_t e, const char *what)
{
	if (e == hipSuccess)
		return 0;
	snprintf(g_error, sizeof g_error, "%s: %s", what, hipGetErrorString(e));
	return 1;
}

int fill_params(const DeodrHipScene *sc, double sigma, void *workspace, size_t workspace_bytes, KParams &p, bool backward)
{
	if (!sc)
		return fail("scene == NULL");
	// the checks of checkSceneValid (H.h:2664-2715) that do not need to read device memory
	if (!sc->faces || !sc->faces_uv || !sc->depths || !sc->uv || !sc->ij || !sc->shade || !sc->colors || !sc->edgeflags || !sc->textured ||
		!sc->shaded)
		return fail("scene array == NULL");
	if ((sc->background_image == nullptr) == (sc->background_color == nullptr))
		return fail("exactly one of scene.background_image / scene.background_color must be given");
	if (sc->nb_triangles < 0 || sc->nb_vertices <= 0 || sc->nb_uv <= 0 || sc->height <= 0 || sc->width <= 0 || sc->n_views <= 0)
		return fail("invalid scene dimensions");
	if (sc->nb_colors <= 0 || sc->nb_colors > DEODR_HIP_MAX_COLORS)
		return fail("nb_colors out of range");
	if (sc->nb_triangles >= (1 << 30))
		return fail("more than 2^30 triangles");
	if (sc->height > 32767 || sc->width > 32767)
		return fail("image larger than 32767 pixels (pixel coordinates are 16-bit, as in the reference)");
	if ((sc->vertex_dtype != DEODR_HIP_F32 && sc->vertex_dtype != DEODR_HIP_F64) || (sc->pixel_dtype != DEODR_HIP_F32 && sc->pixel_dtype != DEODR_HIP_F64))
		return fail("unknown dtype tag");
	if (sc->texture && (sc->texture_height < 2 || sc->texture_width < 2))
		return fail("texture must be at least 2 x 2");
	if (backward)
	{
		if (!sc->backface_culling)
			return fail("You have to use backface_culling true if you ant to compute gradients"); // H.h:2924
		if (sc->perspective_correct)
			return fail("backward gradient propagation not supported yet with perspective_correct=True"); // H.h:810
		if (!sc->uv_b || !sc->ij_b || !sc->shade_b || !sc->colors_b)
			return fail("scene gradient array == NULL");
		if (sc->texture && !sc->texture_b)
			return fail("scene.texture_b == NULL although scene.texture is given"); // H.h:2694
	}
	if (!workspace)
		return fail("workspace == NULL");
	memset(&p, 0, sizeof p);
	{ // the spill-pool capacity is implied by the workspace size: the largest pool_pairs whose layout fits
		const size_t per_view = workspace_bytes / (size_t)sc->n_views;
		if (make_layout(sc->nb_triangles, sc->height, sc->width, sc->nb_colors, 1).view_bytes > per_view)
			return fail("workspace too small (see deodr_hip_workspace_bytes)");
		size_t lo = 1, hi = 0x7fffffffu;
		while (lo < hi)
		{
			size_t mid = lo + (hi - lo + 1) / 2;
			if (make_layout(sc->nb_triangles, sc->height, sc->width, sc->nb_colors, mid).view_bytes <= per_view)
				lo = mid;
			else
				hi = mid - 1;
		}
		p.L = make_layout(sc->nb_triangles, sc->height, sc->width, sc->nb_colors, lo);
	}
	p.faces = sc->faces;
	p.faces_uv = sc->faces_uv;
	p.textured = sc->textured;
	p.shaded = sc->shaded;
	p.edgeflags = sc->edgeflags;
	p.depths = sc->depths;
	p.ij = sc->ij;
	p.shade = sc->shade;
	p.colors = sc->colors;
	p.uv = sc->uv;
	p.texture = sc->texture;
	p.bg_image = sc->background_image;
	p.bg_color = sc->background_color;
	p.uv_b = sc->uv_b;
	p.ij_b = sc->ij_b;
	p.shade_b = sc->shade_b;
	p.colors_b = sc->colors_b;
	p.texture_b = sc->texture_b;
	p.T = sc->nb_triangles;
	p.V = sc->nb_vertices;
	p.Vuv = sc->nb_uv;
	p.H = sc->height;
	p.W = sc->width;
	p.C = sc->nb_colors;
	p.tex_h = sc->texture_height;
	p.tex_w = sc->texture_width;
	p.clockwise = sc->clockwise != 0;
	p.culling = sc->backface_culling != 0;
	p.strict = sc->strict_edge != 0;
	p.persp = sc->perspective_correct != 0;
	p.vtx_f64 = sc->vertex_dtype == DEODR_HIP_F64;
	p.pix_f64 = sc->pixel_dtype == DEODR_HIP_F64;
	p.offset = sc->integer_pixel_centers ? 0.0 : 0.5;
	p.sigma = sigma;
	p.ws = (char *)workspace;
	p.row_group = ROW_GROUP;
	return 0;
}

// ---- optional per-kernel timing (bench.py's roofline leg): hipEvents recorded on the launch stream around each kernel
enum KernelId
{
	KID_SETUP = 0,
	KID_RASTER_FWD = 1,
	KID_RASTER_BWD = 2,
	KID_FINALIZE = 3,
	KID_COUNT = 4
};
struct ProfEvent
{
	hipEvent_t start, stop;
	int kid;
};
bool g_profile = false;	  // the launches of the current call are bracketed by events
int g_profile_every = 0;	  // deodr_hip_profile_enable(n): 0 off, n > 0: every n-th forward (and the adjoint that follows it)
unsigned g_profile_calls = 0; // forwards seen since profiling was enabled
bool g_force_generic = false; // deodr_hip_force_generic(1): run the un-staged kernels (the parity suite covers both families)
// Tuning constants (measured in round 1, profiles/README.md); deliberately NOT read from the environment: nothing outside the
// arguments of a call may change what the call launches.
#ifndef DR_EDGE_WAVES
#define DR_EDGE_WAVES 1024
#endif
constexpr int EDGE_WAVES = DR_EDGE_WAVES; // persistent waves per view of the adjoint's edge kernel

template <class PixT>
void launch_adjoint_raster(const KParams &p, bool fast, bool owner_tiles, dim3 grid4, dim3 edge_grid, hipStream_t st)
{
	if (!fast)
	{
		hipLaunchKernelGGL(raster_bwd_kernel<PixT>, grid4, dim3(256), 0, st, p);
		return;
	}
	// the kernels are compiled twice: a scene without texture (no KIND_TEXTURED primitive can exist: the set-up kernel drops
	// textured triangles of such a scene and raises DEODR_HIP_ERR_NO_TEXTURE) runs the instances without any texture code
	const bool tex = p.texture != nullptr;
	if (owner_tiles) // (after a fused forward the tiles without edges have already been back-propagated)
	{
		KParams q = p;
		q.tile_blocks = fwd_tile_blocks(p.L.ntiles); // the grid of the forward that built the work list
		q.heavy_share = heavy_share_for(p.n_views, q.tile_blocks);
		const dim3 grid((unsigned)p.n_views * (unsigned)q.tile_blocks);
		if (tex)
			hipLaunchKernelGGL((raster_bwd_fast_kernel<PixT, true>), grid, dim3(64), 0, st, q);
		else
			hipLaunchKernelGGL((raster_bwd_fast_kernel<PixT, false>), grid, dim3(64), 0, st, q);
	}
	// (running the two kernels side by side on a forked stream was measured: no gain, the edge kernel just stretches)
	if (p.sigma > 0)
	{
		if (tex)
			hipLaunchKernelGGL((raster_bwd_edge_kernel<PixT, true>), edge_grid, dim3(64), 0, st, p);
		else
			hipLaunchKernelGGL((raster_bwd_edge_kernel<PixT, false>), edge_grid, dim3(64), 0, st, p);
	}
}

std::vector<ProfEvent> g_prof_events;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_free;

struct ScopedKernelTimer
{
	hipStream_t stream;
	ProfEvent ev;
	bool on;
	ScopedKernelTimer(int kid, hipStream_t st) : stream(st), on(g_profile)
	{
		if (!on)
			return;
		if (g_prof_free.empty())
		{
			(void)hipEventCreate(&ev.start);
			(void)hipEventCreate(&ev.stop);
		}
		else
		{
			ev.start = g_prof_free.back().first;
			ev.stop = g_prof_free.back().second;
			g_prof_free.pop_back();
		}
		ev.kid = kid;
		(void)hipEventRecord(ev.start, stream);
	}
	~ScopedKernelTimer()
	{
		if (!on)
			return;
		(void)hipEventRecord(ev.stop, stream);
		g_prof_events.push_back(ev);
	}
};

// The side stream the background fill runs on (one per device, created at first use), with the two events that fork it from
// and join it back to the caller's stream.  Re-recording an event does not disturb a wait already enqueued on its previous
// record, so one pair serves every call; the mutex keeps the record / wait pairs of concurrent host threads together.  Under
// stream capture the fork / join become edges of the captured graph.
struct SideStream
{
	hipStream_t stream = nullptr;
	hipEvent_t fork = nullptr, join = nullptr;
};
std::mutex g_side_mutex;
std::vector<SideStream> g_side;

int side_stream(SideStream &out)
{
	int dev = 0;
	if (check_hip(hipGetDevice(&dev), "hipGetDevice"))
		return 1;
	if ((size_t)dev >= g_side.size())
		g_side.resize(dev + 1);
	SideStream &ss = g_side[dev];
	if (!ss.stream)
	{
		if (check_hip(hipStreamCreateWithFlags(&ss.stream, hipStreamNonBlocking), "side stream") ||
			check_hip(hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming), "side event") ||
			check_hip(hipEventCreateWithFlags(&ss.join, hipEventDisableTiming), "side event"))
			return 1;
	}
	out = ss;
	return 0;
}

// Staged forward: counters -> work list + tile bitmap (scan), then the raster on the caller's stream and, forked from it, the
// background fill on the side stream.  *join receives the event the caller's stream has to wait for before the call returns
// control to it (the fill overlaps whatever the call launches in between).
template <class PixT>
int launch_forward_staged(const KParams &p, bool fused, hipStream_t stream, hipEvent_t *join)
{
	KParams q = p;
	q.tile_blocks = fwd_tile_blocks(p.L.ntiles);
	q.heavy_share = heavy_share_for(p.n_views, q.tile_blocks);
	hipLaunchKernelGGL(tile_scan_kernel, dim3((p.L.ntiles + SCAN_BLOCK - 1) / SCAN_BLOCK, p.n_views), dim3(SCAN_BLOCK), 0, stream, q);
	if (p.fill_mode == 0)
	{
		std::lock_guard<std::mutex> lock(g_side_mutex);
		SideStream ss;
		if (side_stream(ss) || check_hip(hipEventRecord(ss.fork, stream), "fork") || check_hip(hipStreamWaitEvent(ss.stream, ss.fork, 0), "fork"))
			return 1;
		const unsigned words = (unsigned)(p.n_views * p.L.nwords);
		hipLaunchKernelGGL(fill_kernel<PixT>, dim3((words + FILL_WAVES - 1) / FILL_WAVES), dim3(64 * FILL_WAVES), 0, ss.stream, q, fused ? 0 : 1);
		if (check_hip(hipEventRecord(ss.join, ss.stream), "join"))
			return 1;
		*join = ss.join;
	}
	const dim3 grid((unsigned)p.n_views * (unsigned)q.tile_blocks);
	const bool tex = p.texture != nullptr; // (see launch_adjoint_raster)
	if (fused && tex)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, true>), grid, dim3(64), 0, stream, q);
	else if (fused)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, false>), grid, dim3(64), 0, stream, q);
	else if (tex)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, false, true>), grid, dim3(64), 0, stream, q);
	else
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, false, false>), grid, dim3(64), 0, stream, q);
	return 0;
}

// grid of the un-staged kernels: four tiles (wavefronts) per workgroup
dim3 generic_grid(const KParams &p, int n_views) { return dim3((unsigned)(((p.L.tiles_x + 3) / 4) * p.L.tiles_y), (unsigned)n_views); }

// fused: the forward also back-propagates L = sum (image - obs)^2 through the tiles that have no silhouette edge (staged
// kernels only; the caller checks).  *join: see launch_forward_staged (nullptr when nothing was forked).
int launch_forward(const DeodrHipScene *sc, KParams &p, hipStream_t stream, hipEvent_t *join, bool fused = false)
{
	const int n_views = sc->n_views;
	*join = nullptr;
	g_profile = g_profile_every > 0 && (g_profile_calls++ % (unsigned)g_profile_every) == 0;
	p.n_views = n_views;
	const bool fast = !p.aa_err && p.C <= CH && !g_force_generic;
	if (p.T > 0)
	{
		dim3 grid((unsigned)prim_blocks(p.T) * (unsigned)n_views);
		ScopedKernelTimer t(KID_SETUP, stream);
		hipLaunchKernelGGL(setup_bin_kernel, grid, dim3(PRIM_BLOCK), 0, stream, p);
	}
	{
		ScopedKernelTimer t(KID_RASTER_FWD, stream);
		const bool f64 = sc->pixel_dtype == DEODR_HIP_F64;
		if (fast)
		{
			if (f64 ? launch_forward_staged<double>(p, fused, stream, join) : launch_forward_staged<float>(p, fused, stream, join))
				return 1;
		}
		else if (f64)
			hipLaunchKernelGGL(raster_fwd_kernel<double>, generic_grid(p, n_views), dim3(256), 0, stream, p);
		else
			hipLaunchKernelGGL(raster_fwd_kernel<float>, generic_grid(p, n_views), dim3(256), 0, stream, p);
	}
	return check_hip(hipGetLastError(), "forward launch");
}

// the caller's stream waits for the side stream's work of this call
int join_side(hipStream_t stream, hipEvent_t join) { return join ? check_hip(hipStreamWaitEvent(stream, join, 0), "join") : 0; }

// adjoint raster and the per-primitive finalize; owner_tiles = false after a fused forward
int launch_adjoint(const DeodrHipScene *sc, KParams &p, hipStream_t st, bool owner_tiles)
{
	const bool fast = !p.aa_err && p.C <= CH && !g_force_generic;
	p.n_views = sc->n_views;
	// persistent waves of the edge kernel: enough to cover a silhouette-heavy single view, few enough that with many views
	// the waves that find their sub-list exhausted cost nothing
	const int edge_waves = p.L.ntiles < EDGE_WAVES ? p.L.ntiles : EDGE_WAVES;
	dim3 edge_grid(sc->n_views, edge_waves + (fast ? fill_share_blocks(fill_share(p.fill_mode, 0, p.L.nwords)) : 0));
	{
		ScopedKernelTimer t(KID_RASTER_BWD, st);
		if (sc->pixel_dtype == DEODR_HIP_F64)
			launch_adjoint_raster<double>(p, fast, owner_tiles, generic_grid(p, sc->n_views), edge_grid, st);
		else
			launch_adjoint_raster<float>(p, fast, owner_tiles, generic_grid(p, sc->n_views), edge_grid, st);
	}
	if (p.T > 0)
	{
		const int fill_words = fast ? sc->n_views * fill_share(p.fill_mode, 1, p.L.nwords) : 0;
		dim3 g2((unsigned)prim_blocks(p.T) * (unsigned)sc->n_views + (unsigned)((fill_words + PRIM_BLOCK / 64 - 1) / (PRIM_BLOCK / 64)));
		ScopedKernelTimer t(KID_FINALIZE, st);
		hipLaunchKernelGGL(finalize_kernel, g2, dim3(PRIM_BLOCK), 0, st, p);
	}
	return check_hip(hipGetLastError(), "backward launch");
}

// Workspaces whose last forward was the fused one: their owner buffer is incomplete (tiles without edges are not written), so a
// later deodr_hip_render_scene_b that claims to have the forward state must rebuild it.
std::mutex g_fused_mutex;
std::unordered_set<const void *> g_fused_ws;
void note_forward(const void *workspace, bool fused)
{
	std::lock_guard<std::mutex> lock(g_fused_mutex);
	if (fused)
		g_fused_ws.insert(workspace);
	else
		g_fused_ws.erase(workspace);
}
bool last_forward_was_fused(const void *workspace)
{
	std::lock_guard<std::mutex> lock(g_fused_mutex);
	return g_fused_ws.count(workspace) != 0;
}

} // namespace

extern "C" {

int deodr_hip_abi_version(void) { return DEODR_HIP_ABI_VERSION; }

int deodr_hip_force_generic(int on)
{
	g_force_generic = on != 0;
	return 0;
}

int deodr_hip_profile_enable(int every)
{
	g_profile_every = every > 0 ? every : 0;
	g_profile_calls = 0;
	g_profile = false;
	return 0;
}

int deodr_hip_profile_read(double ms_sum[4], unsigned long long launches[4])
{
	for (int i = 0; i < KID_COUNT; i++)
	{
		ms_sum[i] = 0;
		launches[i] = 0;
	}
	for (ProfEvent &e : g_prof_events)
	{
		if (check_hip(hipEventSynchronize(e.stop), "profile sync"))
			return 1;
		float ms = 0;
		if (check_hip(hipEventElapsedTime(&ms, e.start, e.stop), "profile elapsed"))
			return 1;
		ms_sum[e.kid] += ms;
		launches[e.kid] += 1;
		g_prof_free.push_back({e.start, e.stop});
	}
	g_prof_events.clear();
	return 0;
}

const char *deodr_hip_last_error(void) { return g_error; }

size_t deodr_hip_workspace_bytes(int nb_triangles, int height, int width, int nb_colors, int n_views, size_t pool_pairs)
{
	if (nb_triangles < 0 || height <= 0 || width <= 0 || nb_colors <= 0 || n_views <= 0)
		return 0;
	return make_layout(nb_triangles, height, width, nb_colors, pool_pairs).view_bytes * (size_t)n_views;
}

int deodr_hip_render_scene(const DeodrHipScene *sc, void *image, void *z_buffer, double sigma, int antialiase_error, const void *obs,
						   void *err_buffer, void *workspace, size_t workspace_bytes, void *stream)
{
	KParams p;
	if (fill_params(sc, sigma, workspace, workspace_bytes, p, false))
		return 1;
	if (antialiase_error && (!obs || !err_buffer))
		return fail("antialiase_error needs obs and err_buffer");
	p.image = image;
	p.zbuf = z_buffer;
	p.aa_err = antialiase_error != 0;
	p.obs = obs;
	p.err = err_buffer;
	note_forward(workspace, false);
	hipEvent_t join = nullptr;
	if (launch_forward(sc, p, (hipStream_t)stream, &join))
		return 1;
	return join_side((hipStream_t)stream, join);
}

int deodr_hip_render_scene_b(const DeodrHipScene *sc, const void *image, const void *z_buffer, const void *image_b, double sigma,
							 int antialiase_error, const void *obs, const void *err_buffer, const void *err_buffer_b, void *workspace,
							 size_t workspace_bytes, int have_forward_state, void *stream)
{
	(void)z_buffer;
	(void)err_buffer;
	KParams p;
	if (fill_params(sc, sigma, workspace, workspace_bytes, p, true))
		return 1;
	if (antialiase_error)
	{
		if (!obs || !err_buffer_b)
			return fail("antialiase_error needs obs and err_buffer_b");
	}
	else if (!image_b && !(image && obs))
		return fail("image_b == NULL (or, for the residual mode, image and obs)");
	hipStream_t st = (hipStream_t)stream;
	hipEvent_t join = nullptr;
	if (!have_forward_state || last_forward_was_fused(workspace))
	{ // stateless use (or a fused forward, which leaves no complete owner buffer): rebuild records, tile lists and the owner
	  // buffer (no image / z written)
		note_forward(workspace, false);
		KParams f = p;
		f.image = nullptr;
		f.zbuf = nullptr;
		f.aa_err = 0;
		if (launch_forward(sc, f, st, &join) || join_side(st, join)) // (the generic adjoint reads the owner ids of every tile)
			return 1;
	}
	p.image_b = image_b;
	p.image_in = image;
	p.obs = obs;
	p.err_b = err_buffer_b;
	p.aa_err = antialiase_error != 0;
	return launch_adjoint(sc, p, st, true);
}

int deodr_hip_render_scene_fit(const DeodrHipScene *sc, void *image, void *z_buffer, double sigma, const void *obs, int clear_gradients,
							   void *workspace, size_t workspace_bytes, void *stream)
{
	KParams p;
	if (fill_params(sc, sigma, workspace, workspace_bytes, p, true))
		return 1;
	if (!image || !obs)
		return fail("render_scene_fit needs image and obs");
	hipStream_t st = (hipStream_t)stream;
	p.image = image;
	p.zbuf = z_buffer;
	p.obs = obs;
	p.image_in = image;
	if (clear_gradients)
	{
		p.clear_grads = 1; // vertex arrays: zeroed by the set-up kernel; the texture gradient (large, if any) by a fill
		const size_t ps = sc->pixel_dtype == DEODR_HIP_F64 ? 8 : 4;
		if (p.texture_b && check_hip(hipMemsetAsync(p.texture_b, 0, (size_t)p.tex_h * p.tex_w * p.C * ps, st), "clear texture_b"))
			return 1;
	}
	const bool fused = p.C <= CH && !g_force_generic;
	// the background of the empty tiles rides on the adjoint's kernels (fill_share); without any of them: the side stream
#ifndef DR_FILL_MASK
#define DR_FILL_MASK 3 // measurement builds: 0 side stream, 1 edge kernel only, 2 finalize only
#endif
	p.fill_mode = fused ? (((sigma > 0 ? 1 : 0) | (p.T > 0 ? 2 : 0)) & DR_FILL_MASK) : 0;
	note_forward(workspace, fused);
	hipEvent_t join = nullptr;
	if (launch_forward(sc, p, st, &join, fused))
		return 1;
	if (launch_adjoint(sc, p, st, !fused))
		return 1;
	return join_side(st, join); // the background fill has been overlapping the adjoint
}

#ifdef DR_WAVE_TRACE
int deodr_hip_debug_wave_phase(void *dst, size_t bytes) // tools/wave_trace.py
{
	return check_hip(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wave_phase), bytes < sizeof(g_wave_phase) ? bytes : sizeof(g_wave_phase)), "wave phase");
}
int deodr_hip_debug_wave_trace(void *dst, size_t bytes) // tools/wave_trace.py
{
	return check_hip(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wave_trace), bytes < sizeof(g_wave_trace) ? bytes : sizeof(g_wave_trace)), "wave trace");
}
#endif

int deodr_hip_workspace_status(const DeodrHipScene *sc, void *workspace, size_t workspace_bytes, void *stream, int *overflowed,
							   unsigned long long *needed_pairs, int *scene_errors)
{
	KParams p;
	if (fill_params(sc, 1.0, workspace, workspace_bytes, p, false))
		return 1;
	if (check_hip(hipStreamSynchronize((hipStream_t)stream), "status sync"))
		return 1;
	unsigned long long worst = 0;
	unsigned errors = 0;
	for (int v = 0; v < sc->n_views; v++)
	{
		WsHeader h;
		if (check_hip(hipMemcpy(&h, (char *)workspace + (size_t)v * p.L.view_bytes + p.L.hdr, sizeof h, hipMemcpyDeviceToHost), "status copy"))
			return 1;
		if (h.needed_max > worst)
			worst = h.needed_max;
		errors |= h.scene_errors;
	}
	if (needed_pairs)
		*needed_pairs = worst;
	if (overflowed)
		*overflowed = worst > (p.L.tri_pool_cap < p.L.edge_pool_cap ? p.L.tri_pool_cap : p.L.edge_pool_cap);
	if (scene_errors)
		*scene_errors = (int)errors;
	return 0;
}

int deodr_hip_workspace_census(const DeodrHipScene *sc, void *workspace, size_t workspace_bytes, void *stream,
								unsigned long long *nonempty_tiles, unsigned long long *edge_tiles)
{ // measurement hook: how many tiles of the last forward held a primitive / a silhouette edge (all views)
	KParams p;
	if (fill_params(sc, 1.0, workspace, workspace_bytes, p, false))
		return 1;
	if (check_hip(hipStreamSynchronize((hipStream_t)stream), "census sync"))
		return 1;
	unsigned long long filled = 0, edged = 0;
	std::vector<uint32_t> bits(p.L.nwords), saved(p.L.ntiles);
	for (int v = 0; v < sc->n_views; v++)
	{
		const char *base = (const char *)workspace + (size_t)v * p.L.view_bytes;
		WsHeader h;
		if (check_hip(hipMemcpy(&h, base + p.L.hdr, sizeof h, hipMemcpyDeviceToHost), "census copy") ||
			check_hip(hipMemcpy(bits.data(), base + p.L.tile_bits, sizeof(uint32_t) * p.L.nwords, hipMemcpyDeviceToHost), "census copy") ||
			check_hip(hipMemcpy(saved.data(), base + p.L.edge_saved, sizeof(uint32_t) * p.L.ntiles, hipMemcpyDeviceToHost), "census copy"))
			return 1;
		for (int t = 0; t < p.L.ntiles; t++)
			if ((bits[t >> 5] >> (t & 31)) & 1u)
			{
				filled++;
				edged += (saved[t] & ~SWEEP_SAVED) != 0;
			}
	}
	if (nonempty_tiles)
		*nonempty_tiles = filled;
	if (edge_tiles)
		*edge_tiles = edged;
	return 0;
}

int deodr_hip_workspace_pool_pairs(const DeodrHipScene *sc, size_t workspace_bytes, unsigned long long *pool_pairs)
{ // capacity (in pairs) of the spill pools of a workspace of this size: what the polled `all_needed_max` is compared with
	if (!sc || sc->n_views <= 0)
		return fail("scene == NULL");
	char dummy;
	KParams p;
	if (fill_params(sc, 1.0, &dummy, workspace_bytes, p, false))
		return 1;
	if (pool_pairs)
		*pool_pairs = p.L.tri_pool_cap < p.L.edge_pool_cap ? p.L.tri_pool_cap : p.L.edge_pool_cap;
	return 0;
}

} // extern "C"
