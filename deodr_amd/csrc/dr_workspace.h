// deodr_amd/csrc/dr_workspace.h -- part of the single translation unit dr_kernels.hip (device code, gfx950 / wave64).
// Workspace layout, kernel parameters and the wave-level primitives shared by every kernel of libdeodr_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <stddef.h>
#include <stdint.h>

#include "../../include/deodr_hip.h"
#include "dr_prims.h"

using namespace dr;

namespace
{

constexpr int TILE = 8;		// 8 x 8 pixels = one wavefront, lane = (y & 7) * 8 + (x & 7)
#ifndef DR_K_LIST
#define DR_K_LIST 64 // 32 until round 4: a tile with more primitives than inline slots scans the view's spill pool (two more round trips for the
					 // long tiles, and every such walker reads the whole pool): 8 views 0.1211 -> 0.1175 ms, 1 view 0.0693 -> 0.0667 (profiles/r04t)
#endif
constexpr int K_TRI = DR_K_LIST;  // inline triangle slots per tile; more spill to the pool (at most 64: one per lane)
constexpr int K_EDGE = DR_K_LIST; // inline edge slots per tile
static_assert(K_TRI <= 64 && (K_TRI & (K_TRI - 1)) == 0, "a wavefront loads the inline list with one lane per slot");
constexpr int CH = 4;		// colour channels kept in registers at a time
constexpr int MAX_SORTED = 64; // edges of a tile whose blending order is cached in LDS
constexpr int CNT_STRIDE = 32; // uint32 between two append counters (one 128-byte line each)
constexpr int DYN_GROUPS = 8; // ticket counters per view of the forward raster's persistent walkers (one per XCD; behind the append counters)
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
constexpr int SCAN_TILES = 256;	 // tiles per workgroup of tile_scan_kernel
constexpr int SPLIT_BUDGET = 32; // extra work-list entries (copies of tiles of many edges) such a workgroup may add
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
				  (int)dr::SCENE_ERR_NO_TEXTURE == DEODR_HIP_ERR_NO_TEXTURE && (int)dr::SCENE_ERR_DET_RANGE == DEODR_HIP_ERR_DET_RANGE,
			  "status block layout published in include/deodr_hip.h");

// Step-done flag (KParams::done_flag): finalize_kernel's wavefronts count themselves on DONE_SUBS counters, a cache line apart -- on ONE word
// the 24 000 returning atomics of an 8-view step are executed one after the other at the memory side: 55 us
constexpr int DONE_SUBS = 256, DONE_STRIDE = 16;
constexpr int LOSS_SLOTS = 256; // partial sums of the loss per view (one per walker was 32 768 values for ONE workgroup to add up: 20 us)

struct Layout
{
	size_t hdr, tri_rec, tri_planes, tri_acc, edge_rec, edge_planes, edge_acc, tri_cnt, edge_cnt, edge_saved, tri_list, edge_list, tri_pool,
		edge_pool, face_id, tile_bits, tri_flag, work_list, edge_tile_cnt, edge_tiles, edge_slot, edge_sweep, edge_snap, view_bytes;
	uint32_t tri_pool_cap, edge_pool_cap;
	size_t edge_fin;
	size_t done_counts;
	int work_cap; // entries of work_list
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
	// one entry per non-empty tile (many-primitive tiles from the front, the others from the back) + the extra copies of the tiles of many
	// edges, at most SPLIT_BUDGET per block of the scan kernel (tile_scan_kernel)
	L.work_cap = L.ntiles + ((L.ntiles + SCAN_TILES - 1) / SCAN_TILES) * SPLIT_BUDGET;
	L.work_list = take(sizeof(WorkEntry) * (size_t)L.work_cap);
	// kind | front << 2 of every triangle of the last forward: what finalize_kernel needs to know about a triangle before it
	// touches anything else (one coalesced byte per thread instead of a 128-byte record line per triangle, two out of three
	// of which are culled)
	L.tri_flag = take((size_t)T);
	L.edge_tile_cnt = take(sizeof(uint32_t) * (EDGE_LISTS + 1 + DYN_GROUPS) * CNT_STRIDE); // append counters of the lists + the sweep-slot counter + the walkers' ticket counters
	L.edge_tiles = take(sizeof(uint32_t) * EDGE_LISTS * (size_t)L.ntiles);	   // [list][ntiles]
	L.edge_slot = take(sizeof(uint32_t) * L.ntiles); // 1 + index of the tile's slot in edge_sweep, 0: none
	L.sweep_cap = SWEEP_CAP < L.ntiles ? SWEEP_CAP : L.ntiles;
	L.edge_sweep = take(SWEEP_BYTES * (size_t)L.sweep_cap);
	// what finalize_kernel needs of a drawn silhouette edge besides its record: vertex ids, positions, attributes (written by the
	// set-up kernel, which has them in registers: the finalize thread of an edge then has ONE memory round trip before its arithmetic
	// instead of three -- indices, vertices, record)
	L.edge_fin = take(sizeof(EdgeFin) * 3 * (size_t)T);
	L.edge_snap = take(SNAP_BYTES * SNAP_CAP);
	L.done_counts = take(sizeof(uint32_t) * (DONE_SUBS + 1) * DONE_STRIDE); // (view 0's are used: the step-done flag, dr_finalize.h)
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
	int fuse_edges;	 // fit step: the forward raster also runs the adjoint of the tiles that hold silhouette edges (no edge-tile kernel)
	int clear_grads; // the set-up kernel zeroes the per-view gradient arrays (a fit step that wants fresh gradients: no separate fills)
	// set by the host per launch: the per-primitive kernels go through their per-workgroup LDS tables (finalize: vertex adjoints merged
	// before they leave; set-up: one slot request per tile and workgroup) -- worth it when the launch is large enough to be bound by the
	// rate of memory-side atomics rather than by the latency of one wavefront's chain (a single 20 k-triangle view is not: + 1.7 us)
	int prim_tables;
	// fit step of an untextured scene (round 4): the per-primitive adjoint algebra of finalize_kernel runs INSIDE the forward raster, by
	// extra workgroups at the end of its grid that wait, block of tiles by block of tiles, for the walkers (set by the host; the
	// set-up kernel then lists the edges it draws, the scan kernel counts the non-empty tiles of every block)
	// Deterministic accumulation (deodr_hip_set_deterministic; the un-staged kernels only): every gradient sum is an INTEGER sum of
	// contributions rounded to multiples of 2^-32 -- integer addition is associative, so the order in which the memory system executes
	// the atomics no longer shows in the result (the reference is bit-reproducible by construction: one thread, H.h:1029-1037).  The
	// moment accumulators of the workspace hold int64 then (zero is zero in both readings); the vertex / texture gradients are summed
	// in int64 shadow arrays (det_*: library-owned scratch) and added to the caller's arrays by one thread per element afterwards.
	int det;
	long long *det_ij, *det_colors, *det_shade, *det_uv, *det_texture;
	uint32_t *det_err; // sticky error word (view 0's WsHeader::scene_errors): SCENE_ERR_DET_RANGE when a contribution or a sum leaves +- 2^31
	// Measurement (deodr_hip_profile_stamps): the first thread of set-up / tile scan / finalize writes the 100 MHz realtime counter into
	// stamp[0 / 1 / 2] -- kernels of one stream run back to back, so the difference of two consecutive stamps IS the duration of the kernel(s)
	// between them, with no event packet between the launches (a hipEvent pair per kernel costs the step it measures ~ 36 us)
	unsigned long long *stamp;
	// Step-done flag (DeodrHipFitOptions::done_flag): the last wavefront of finalize_kernel to finish stores done_value there once every
	// wavefront's gradients are visible device-wide -- what a consumer on another stream waits for (deodr_hip_wait_flag) instead of an event
	uint32_t *done_flag;
	uint32_t done_value;
	int split_part; // edges per copy of a tile with several batches of silhouette edges (fused forward: dr_forward.h)
	// Launch constants of the staged forward raster, formed ONCE on the host (round 6): every one of its 27 000 one-wave workgroups used to derive them
	// on the scalar unit before its first load -- four integer divisions (by the number of views, by the head share, inside fill_share) among the ~420
	// instructions in front of a walker's first tile, on a scalar unit that twenty wavefronts of a CU share.
	int pair_tex;		  // tile_scan_kernel: textured scenes pair their edge-free tiles too (launches of fewer than DR_TEX_TWO_KERNELS views)
	int setup_sparse;	  // set-up kernel: a triangle every `setup_sparse` lanes (1, or 4 for small launches: dr_setup.h)
	uint32_t fwd_walkers; // walkers per view in the forward raster's grid: tile_blocks, or fwd_heads + the persistent walkers of the others' list (dyn_groups)
	int dyn_groups;		  // > 0: the walkers of the others' list are persistent and take their entries by ticket (DYN_GROUPS counters per view)
	uint32_t fwd_heads;	  // walkers per view on the head of the work list: tile_blocks / heavy_share (0: the list has one class)
	uint32_t fwd_n_fill;  // workgroups of this launch that stream the background of the forward's share of the empty tiles
	uint32_t fwd_dealt;	  // of those, groups of eight dealt among the walkers (behind every 64), see raster_fwd_fast_kernel
	uint32_t views_magic; // floor(2^32 / n_views) + 1: x / n_views = mulhi(x, magic) for x < 2^32 / n_views (every workgroup index >> 3 is)
	uint32_t block_base; // staged forward launched as two kernels (textured fit step): index of this launch's first workgroup in the one-kernel grid
	// loss of a fit step, sum (image - obs)^2 (deodr_hip_render_scene_fit_loss): loss_tile_bg[0] = the loss of a frame that is all
	// background, [1 + view * ntiles + tile] = that of one tile; loss_wave[view * LOSS_SLOTS + q % LOSS_SLOTS]: walker q of the forward
	// raster adds (loss of a tile - its background loss) for every tile it walks; one workgroup of finalize_kernel writes loss_out[0] =
	// [0] + sum of those (zeroed by the set-up kernel)
	const double *loss_tile_bg;
	double *loss_wave, *loss_out;
	// residual mode with a clamp: L = sum (clamp(image, clamp_lo, clamp_hi) - obs)^2, the data term of the depth fitter
	// (deodr/mesh_fitter.py:108-123); the gradient passes on the closed interval, as torch.clamp's does
	int clamp;
	double clamp_lo, clamp_hi;
	int aligned; // set INSIDE raster_fwd_fast_kernel from a template argument: width and height are multiples of the tile (no pixel of a tile is outside)
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

// value whose squared distance to the observation is the loss, and d loss / d value, of a rendered value v (already rounded to the
// pixel type) against the observation o
// CLAMP false: compiled without the clamp (the fused forward raster is at its register limits: the two limits kept in scalar
// registers cost it 37 more spilled registers and 5 us per 8-view step, so a clamped fit step runs its own instance of that kernel)
template <bool CLAMP>
__device__ __forceinline__ double fit_value(const KParams &p, double v)
{
	return (CLAMP && p.clamp) ? (v < p.clamp_lo ? p.clamp_lo : (v > p.clamp_hi ? p.clamp_hi : v)) : v;
}
template <bool CLAMP>
__device__ __forceinline__ double fit_residual(const KParams &p, double v, double o)
{
	if (CLAMP && p.clamp && (v < p.clamp_lo || v > p.clamp_hi))
		return 0.0;
	return 2 * (v - o);
}

// the same residual from float32 operands: 2 (v - o) rounded once -- what the double version gives when its result is rounded to float32
template <bool CLAMP>
__device__ __forceinline__ float fit_residual_f32(const KParams &p, float v, float o)
{
	if (CLAMP && p.clamp && ((double)v < p.clamp_lo || (double)v > p.clamp_hi))
		return 0.0f;
	return 2.0f * (v - o);
}

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

__device__ __forceinline__ void kernel_stamp(const KParams &p, int id)
{
	if (p.stamp && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
		p.stamp[id] = __builtin_amdgcn_s_memrealtime();
}

// ------------------------------------------------------------------------------------------------ wave primitives

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// The lane's index in its wavefront, from the execution mask (v_mbcnt) instead of threadIdx.x: in a one-wave workgroup the two are the same number,
// but threadIdx.x is a live-in of the kernel -- register v0 from the first instruction to the last use -- and the register allocator of the forward
// raster (96 registers, spilling) chose to send it to scratch memory at the kernel's first instruction and to reload it right in front of a walker's
// first load: a memory round trip in the prologue of every wavefront.  This one is two instructions wherever it is needed.  All lanes must be enabled.
__device__ __forceinline__ int wave_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ void atomic_add_f64(double *p, double v) { unsafeAtomicAdd(p, v); }

// fixed point of the deterministic mode: |sum| < 2^31, resolution 2^-32
constexpr double DET_SCALE = 4294967296.0, DET_INV_SCALE = 1.0 / 4294967296.0;
// `err`: a sticky scene-error word of the workspace (view 0's WsHeader::scene_errors).  A contribution beyond the range saturates in the
// conversion and a sum beyond it wraps: both raise SCENE_ERR_DET_RANGE instead of passing silently (the add returns the old value for that:
// the mode is a test mode, several times slower than the default path anyway).
// (ADVICE r5 proposed to look at the FINAL sum once per element instead -- order-independent, and the atomic need not return.  Not taken: a sum
// that wrapped is a valid int64 again, the final value cannot tell; what the returning add buys is that EVERY wrap is seen.  The price is the one
// the advice names: a sum that leaves the range and comes back -- harmless in two's complement -- raises the bit in the runs whose order of
// execution takes it out, and not in the others.  The GRADIENTS are bit-identical either way; the bit then says "too close to the range".)
__device__ __forceinline__ void det_add(void *slot, double v, uint32_t *err)
{
	const long long add = __double2ll_rn(v * DET_SCALE);
	const long long old = (long long)atomicAdd((unsigned long long *)slot, (unsigned long long)add);
	const long long sum = (long long)((unsigned long long)old + (unsigned long long)add);
	if (!(fabs(v) < 2147483648.0) || (((old ^ sum) & (add ^ sum)) < 0)) // (NaN, saturated conversion, or signed overflow of the sum)
		atomicOr(err, (uint32_t)SCENE_ERR_DET_RANGE);
}
__device__ __forceinline__ double det_value(const double *slot) { return (double)*(const long long *)slot * DET_INV_SCALE; }
// accumulate into a moment accumulator of the workspace
// (det: nullptr, or the error word of the deterministic mode)
__device__ __forceinline__ void acc_add(double *slot, double v, uint32_t *det)
{
	if (det)
		det_add(slot, v, det);
	else
		unsafeAtomicAdd(slot, v);
}

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

struct DetAdd // deterministic mode: the same contribution into the int64 shadow of the array (KParams::det_*, this view's part)
{
	const void *ij_b, *colors_b, *shade_b; // the view's arrays as the caller of the functor names them
	long long *ij, *colors, *shade, *uv;
	uint32_t *err;
	__device__ __forceinline__ void operator()(void *arr, size_t i, bool, double v) const
	{
		if (v == 0)
			return;
		det_add((arr == ij_b ? ij : (arr == colors_b ? colors : (arr == shade_b ? shade : uv))) + i, v, err);
	}
};

struct DeviceAdd // accumulate a vertex gradient (the reference's `+=` into scene.*_b)
{
	__device__ __forceinline__ void operator()(void *arr, size_t i, bool f64, double v) const
	{
		if (v == 0)
			return;
		if (f64)
			unsafeAtomicAdd((double *)arr + i, v);
		else
			unsafeAtomicAdd((float *)arr + i, (float)v);
	}
};

// finalize_triangle's sinks (dr_prims.h).  AtomicSink: every contribution goes straight to the gradient arrays (Add = DeviceAdd), or to
// their int64 shadows (DetAdd).
template <class Add = DeviceAdd>
struct AtomicSinkT
{
	const SceneView &s;
	const GradView &g;
	uint32_t f[3], fuv[3];
	Add add;
	__device__ __forceinline__ void color(int i, int c, double v) { add(g.colors_b, (size_t)f[i] * s.C + c, s.vtx_f64, v); }
	__device__ __forceinline__ void shade(int i, double v) { add(g.shade_b, f[i], s.vtx_f64, v); }
	__device__ __forceinline__ void uv(int i, int c, double v) { add(g.uv_b, 2 * (size_t)fuv[i] + c, s.vtx_f64, v); }
	__device__ __forceinline__ void ij(int i, int d, double v) { add(g.ij_b, 2 * (size_t)f[i] + d, s.vtx_f64, v); }
};
typedef AtomicSinkT<DeviceAdd> AtomicSink;
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

} // namespace
