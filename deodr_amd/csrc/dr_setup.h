// deodr_amd/csrc/dr_setup.h -- part of the single translation unit dr_kernels.hip (device code, gfx950 / wave64).
// setup_bin_kernel: per-primitive set-up (cull, stencils, planes, edge records), index checks, binning into 8 x 8 tiles.
#pragma once

#include "dr_workspace.h"

using namespace dr;

namespace
{

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

__host__ __device__ inline int prim_tri_blocks(int T) { return (T + PRIM_BLOCK - 1) / PRIM_BLOCK; }
// An edge-slot block looks at EDGE_SLOTS consecutive slots per thread (a few per cent of the slots are flagged as silhouette edges:
// with one slot per thread three quarters of both kernels' wavefronts did nothing but look at 64 flags)
#ifndef DR_EDGE_SLOTS
#define DR_EDGE_SLOTS 4
#endif
constexpr int EDGE_SLOTS = DR_EDGE_SLOTS, EDGE_BLOCK_SLOTS = PRIM_BLOCK * EDGE_SLOTS;
__host__ __device__ inline int prim_blocks(int T) { return prim_tri_blocks(T) + (3 * T + EDGE_BLOCK_SLOTS - 1) / EDGE_BLOCK_SLOTS; }

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
// tri_slots: triangle slots of a view (p.T, or p.T * KParams::setup_sparse in the set-up kernel of a small scene: see setup_bin_kernel)
__host__ __device__ inline int prim_edge_blocks(int T, int slots_per_thread) { return (3 * T + PRIM_BLOCK * slots_per_thread - 1) / (PRIM_BLOCK * slots_per_thread); }
// edge_slots: edge slots an edge-slot block looks at per thread (EDGE_SLOTS, or 1 in the set-up kernel of a small scene)
__device__ __forceinline__ PrimWork prim_work(const KParams &p, bool edge_first = DR_EDGE_FIRST, int skip = 0, int tri_slots = -1, int edge_slots = EDGE_SLOTS)
{
	const int TBk = prim_tri_blocks(tri_slots < 0 ? p.T : tri_slots), EB = prim_edge_blocks(p.T, edge_slots), nv = p.n_views;
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

// Compacts the flagged slots of an edge block into s_slots (LDS) and returns how many there are; the threads of the block then take
// them PRIM_BLOCK at a time: round r, thread t -> edge_round_slot(total, r) (or -1).  Called by every thread of an edge block.
__shared__ uint32_t s_edge_slots[EDGE_BLOCK_SLOTS];
__device__ __forceinline__ uint32_t compact_flagged_slots(const KParams &p, const uint8_t *edgeflags, int edge_block, int per_thread = EDGE_SLOTS)
{ // per_thread <= EDGE_SLOTS: slots per thread of THIS launch's edge blocks (prim_work's edge_slots)
	__shared__ uint32_t s_count[PRIM_BLOCK / 64];
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const int slot0 = (edge_block * PRIM_BLOCK + tid) * per_thread;
	uint32_t flags = 0; // bit i: slot0 + i is flagged
	if (p.sigma > 0)
#pragma unroll
		for (int i = 0; i < EDGE_SLOTS; i++)
			if (i < per_thread && slot0 + i < 3 * p.T && edgeflags[slot0 + i] != 0)
				flags |= 1u << i;
	const uint32_t mine = (uint32_t)__popc(flags);
	// exclusive prefix of `mine` over the lanes of the wavefront, then over the wavefronts
	uint32_t incl = mine;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1)
	{
		const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
		incl += lane >= d ? up : 0u;
	}
	if (lane == 63)
		s_count[wave] = incl;
	__syncthreads();
	uint32_t before = 0, total = 0;
#pragma unroll
	for (int i = 0; i < PRIM_BLOCK / 64; i++)
	{
		const uint32_t c = s_count[i];
		before += i < wave ? c : 0u;
		total += c;
	}
	uint32_t at = before + incl - mine;
#pragma unroll
	for (int i = 0; i < EDGE_SLOTS; i++)
		if ((flags >> i) & 1u)
			s_edge_slots[at++] = (uint32_t)(slot0 + i);
	__syncthreads();
	return total;
}
__device__ __forceinline__ int edge_round_slot(uint32_t total, int round)
{
	const uint32_t i = (uint32_t)round * PRIM_BLOCK + threadIdx.x;
	return i < total ? (int)s_edge_slots[i] : -1;
}

#ifdef DR_WAVE_TRACE
// timeline of the per-primitive kernels: [wave slot] = (start, end) in 10 ns ticks of the constant 100 MHz counter
__device__ unsigned long long g_wave_trace[3][1 << 18][2]; // 0 set-up, 1 finalize, 2 forward raster
__device__ unsigned long long g_wave_phase[4][1 << 16][8];  // 0 set-up, 1 finalize: time stamps inside the wavefronts that work on edges
__device__ uint32_t g_wave_hw[3][1 << 18][2]; // where the wavefront ran: HW_ID (wave slot, SIMD, CU, SE) | role << 31 (1: a fill workgroup), XCC_ID
struct WaveTrace
{
	int which;
	unsigned long long t0;
	uint32_t role = 0;
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
				uint32_t hw, xcc;
				asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
				asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
				g_wave_hw[which][id][0] = (hw & 0x7fffffffu) | role << 31;
				g_wave_hw[which][id][1] = xcc;
			}
		}
	}
};
#define DR_WAVE_TRACE_SCOPE(w) WaveTrace wave_trace_scope(w)
#define DR_WAVE_TRACE_ROLE(r) wave_trace_scope.role = (r)
#define DR_WAVE_PHASE(i) wave_trace_scope.phase(i)
#define DR_WAVE_PHASE_T(i) wave_trace_scope.phase(i, 1)
#else
#define DR_WAVE_TRACE_SCOPE(w)
#define DR_WAVE_TRACE_ROLE(r)
#define DR_WAVE_PHASE(i)
#define DR_WAVE_PHASE_T(i)
#endif

// VTX64: the dtype of the vertex arrays as a compile-time constant.  As a run-time flag every vertex value was loaded behind its own
// branch, and a float32 value converted -- i.e. waited for -- right behind its load: 21 memory round trips one after the other for
// the inputs of one triangle.
// NC: the channel count at compile time (0: whatever the scene says), as for raster_fwd_fast_kernel.
template <bool VTX64, int NC>
__global__ __launch_bounds__(PRIM_BLOCK, DR_PRIM_WAVES) void setup_bin_kernel(KParams p)
{
	p.vtx_f64 = VTX64 ? 1 : 0; // (what the host passed: now known to the compiler; scene_view() copies it)
	if (NC)
		p.C = NC, p.L.P = NC < 3 ? 3 : NC;
	kernel_stamp(p, 0);
	DR_WAVE_TRACE_SCOPE(0);
	// Small scenes (round 6): a triangle every `sparse` lanes.  A mesh of a thousand large triangles (the hand at 1024^2: 33 tiles per triangle) is 17
	// wavefronts on 256 CUs, each with 64 triangles' worth of 3 x 3-tile blocks to bin -- two or three rounds of dependent slot requests (bin_rest), a
	// soup of 200 triangles nine; the lanes in between take part in that dealing and have nothing else to do, so spread over four times the wavefronts
	// (on a chip that is empty anyway) a wavefront's share fits one round.  The host sets it for launches of at most DR_SPARSE_MAX triangles.
	// The same for the edge-slot blocks: ONE slot per thread instead of EDGE_SLOTS (a soup flags every edge: 600 slots were one workgroup working
	// through three rounds of 256 edges, each with its own rounds of slot requests).
	const int sparse = p.setup_sparse > 1 ? p.setup_sparse : 1, edge_slots = p.setup_sparse > 1 ? 1 : EDGE_SLOTS;
	const PrimWork pw = prim_work(p, DR_EDGE_FIRST, 0, p.T * sparse, edge_slots);
	const int view = pw.view;
	const int item = pw.view_block * PRIM_BLOCK + threadIdx.x; // only an id for the housekeeping below
	const int n_items = (prim_tri_blocks(p.T * sparse) + prim_edge_blocks(p.T, edge_slots)) * PRIM_BLOCK;
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
	if (item <= EDGE_LISTS + DYN_GROUPS) // appended to by tile_scan_kernel, the next kernel on the stream (+ the ticket counters of the forward raster's persistent walkers)
		w.edge_tile_cnt[item * CNT_STRIDE] = 0;
	if (p.loss_wave) // (one partial per tile walker of the forward raster, two kernels later)
		for (int v = item; v < LOSS_SLOTS; v += n_items)
			p.loss_wave[(size_t)view * LOSS_SLOTS + v] = 0;
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
	// A thread bins the first 3 x 3 block of tiles of its primitive itself (for the usual small triangle: all of it).  The other
	// blocks of all the primitives of the wavefront are dealt out evenly over its 64 lanes, nine slot requests per lane and round:
	// a lane working through a wide bounding box alone is a chain of dependent atomic round trips of ~2.5 us each -- the hand mesh
	// at 1024^2 (1 048 triangles of 33 tiles on average, 17 wavefronts on the whole chip) spent 68 us of set-up that way, up to 8
	// rounds per lane plus the wavefront walking its 9 largest boxes one after the other.
	const int lane = threadIdx.x & 63;
	auto bin_rest = [&](int extra, bool first_done, const double *hp, int btx0, int bty0, int bntx, int bnty, int bprim) {
		if (__ballot(extra > 0) == 0)
			return;
		int incl = extra; // inclusive scan over the lanes
#pragma unroll
		for (int d = 1; d < 64; d <<= 1)
		{
			const int up = __shfl_up(incl, d, 64);
			incl += lane >= d ? up : 0;
		}
		const int total = __shfl(incl, 63, 64);
		// BIN_ITEMS blocks per lane and round: their slot requests are all in flight before the first answer is used (a round is one
		// atomic round trip of ~2.5 us whatever the number of requests a lane has pending)
		constexpr int BIN_ITEMS = 2;
		uint32_t *cnt = tri_block ? w.tri_cnt : w.edge_cnt;
		for (int base = 0; base < total; base += 64 * BIN_ITEMS)
		{
			uint32_t got[BIN_ITEMS][9], use[BIN_ITEMS], prim[BIN_ITEMS];
			int tile0[BIN_ITEMS];
#pragma unroll
			for (int u = 0; u < BIN_ITEMS; u++)
			{
				const int item = base + u * 64 + lane;
				int owner = 0; // first lane whose inclusive count exceeds item (lanes past the end: any lane, nothing is written)
#pragma unroll
				for (int step = 32; step; step >>= 1)
					owner += __shfl(incl, owner + step - 1, 64) <= item ? step : 0;
				owner = owner > 63 ? 63 : owner;
				const int before = __shfl(incl - extra, owner, 64);
				double q[12];
#pragma unroll
				for (int i = 0; i < 12; i++)
					q[i] = (tri_block && i >= 9) ? 0.0 : __shfl(hp[i], owner, 64);
				const int tx0 = __shfl(btx0, owner, 64), ty0 = __shfl(bty0, owner, 64), ntx = __shfl(bntx, owner, 64), nty = __shfl(bnty, owner, 64);
				prim[u] = (uint32_t)__shfl(bprim, owner, 64);
				use[u] = 0;
				tile0[u] = 0;
				if (item < total)
				{
					const int nbx = (ntx + 2) / 3, blk = item - before + (first_done ? 1 : 0);
					const int bx = (blk % nbx) * 3, by = (blk / nbx) * 3;
					const int keep_dx = (tri_block && !p.strict) ? ntx - 1 : -1; // (non-strict fill rule: see the first block below)
					const uint32_t outside =
						tri_block ? tiles3x3_outside_halfplanes<3>(q, tx0 + bx, ty0 + by) : tiles3x3_outside_halfplanes<4>(q, tx0 + bx, ty0 + by);
					tile0[u] = (ty0 + by) * p.L.tiles_x + tx0 + bx;
#pragma unroll
					for (int c = 0; c < 9; c++)
					{
						const int dx = bx + c % 3, dy = by + c / 3;
						if (dx < ntx && dy < nty && (!((outside >> c) & 1u) || dx == keep_dx))
							use[u] |= 1u << c;
					}
				}
#pragma unroll
				for (int c = 0; c < 9; c++)
				{
					got[u][c] = 0;
					if ((use[u] >> c) & 1u)
						got[u][c] = atomicAdd(&cnt[tile0[u] + (c / 3) * p.L.tiles_x + c % 3], 1u);
				}
			}
#pragma unroll
			for (int u = 0; u < BIN_ITEMS; u++)
#pragma unroll
				for (int c = 0; c < 9; c++)
					if ((use[u] >> c) & 1u)
					{
						const int tile = tile0[u] + (c / 3) * p.L.tiles_x + c % 3;
						if (tri_block)
							place_in_tile(w.tri_list, K_TRI, w.tri_pool, p.L.tri_pool_cap, &w.hdr->tri_spill[cur], tile, prim[u], got[u][c]);
						else
							place_in_tile(w.edge_list, K_EDGE, w.edge_pool, p.L.edge_pool_cap, &w.hdr->edge_spill[cur], tile, prim[u], got[u][c]);
					}
		}
	};
	if (tri_block)
	{
		int extra = 0;
		double hp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
		int btx0 = 0, bty0 = 0, bntx = 0, bnty = 0, bprim = 0;
		do
		{
			const int slot = pw.index * PRIM_BLOCK + threadIdx.x, k = slot / sparse;
			if (k >= p.T || slot != k * sparse)
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
			const bool on_screen = !(x0 > x1 || y0 > y1);
			const double eq[9] = {rec.eq[0][0], rec.eq[0][1], rec.eq[0][2], rec.eq[1][0], rec.eq[1][1], rec.eq[1][2], rec.eq[2][0], rec.eq[2][1], rec.eq[2][2]};
			const int tx0 = x0 / TILE, ty0 = y0 / TILE, ntx = x1 / TILE - tx0 + 1, nty = y1 / TILE - ty0 + 1;
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
			if (on_screen)
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
			out = rec;
			if (!on_screen)
				break;
			DR_WAVE_PHASE_T(3); // record stored
#pragma unroll
			for (int q = 0; q < 9; q++)
				if (use0[q])
				{
					const int tile = (ty0 + q / 3) * p.L.tiles_x + tx0 + q % 3;
					place_in_tile(w.tri_list, K_TRI, w.tri_pool, p.L.tri_pool_cap, &w.hdr->tri_spill[cur], tile, (uint32_t)k, slot0[q]);
				}
			extra = ((ntx + 2) / 3) * ((nty + 2) / 3) - 1; // the other 3 x 3 blocks of a wider box: shared out below
#pragma unroll
			for (int i = 0; i < 9; i++)
				hp[i] = eq[i];
			btx0 = tx0, bty0 = ty0, bntx = ntx, bnty = nty, bprim = k;
		} while (false);
		DR_WAVE_PHASE_T(4);
		bin_rest(extra, true, hp, btx0, bty0, bntx, bnty, bprim);
		return;
	}
	// nothing is written for the ~97 % of slots that are not silhouette edges: records are only reached through the
	// tile lists, and finalize_kernel works from the same flags
	const uint32_t n_flagged = compact_flagged_slots(p, s.edgeflags, pw.index, edge_slots);
	DR_WAVE_PHASE(1); // flags compacted
	// A round per PRIM_BLOCK flagged slots: one, unless most edges of the block are flagged (a triangle soup).  The first round is
	// written out and the others loop over a second copy of the same code: as ONE loop the body kept its loop-invariant values in
	// registers across a loop that runs once (93 spilled registers, on the path of every edge block).
	auto edge_round = [&](int round) {
		if ((threadIdx.x >> 6) * 64 + round * PRIM_BLOCK >= (int)n_flagged)
			return; // (a wavefront without a slot in this round)
		const int slot = edge_round_slot(n_flagged, round);
		int extra = 0;
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
			extra = ((ntx + 2) / 3) * ((nty + 2) / 3); // all its 3 x 3 blocks: shared out below
#pragma unroll
			for (int i = 0; i < 12; i++)
				hp[i] = band[i];
			btx0 = tx0, bty0 = ty0, bntx = ntx, bnty = nty, bprim = slot;
		} while (false);
		DR_WAVE_PHASE(5); // record done
		bin_rest(extra, false, hp, btx0, bty0, bntx, bnty, bprim);
	};
	if (n_flagged > 0)
		edge_round(0);
	if (n_flagged > (uint32_t)PRIM_BLOCK)
		for (int round = 1; round * PRIM_BLOCK < (int)n_flagged; round++)
			edge_round(round);
}

} // namespace
