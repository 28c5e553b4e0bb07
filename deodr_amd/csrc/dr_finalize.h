// deodr_amd/csrc/dr_finalize.h -- part of the single translation unit dr_kernels.hip (device code, gfx950 / wave64).
// finalize_kernel: moments -> plane adjoints -> adjoint of the 3 x 3 inverse -> vertex gradients.
#pragma once

#include "dr_backward.h"

using namespace dr;

namespace
{

// ------------------------------------------------------------------------------------------------------- finalize

// Vertex adjoints of the triangles of one workgroup, merged in LDS before they leave (round 4).  A vertex receives contributions
// from the ~6 triangles around it, and neighbouring triangles usually sit in the same workgroup; written straight to the gradient
// arrays, every contribution is one memory-side atomic, and an atomic INSTRUCTION whose 64 lanes name 64 different vertices touches
// 64 cache lines (tools/probes/order_atomic_probe.hip: 160 000 triangles x 18 adds take 117 us that way, 14 us through a table like
// this one -- 38 us even when no two triangles share a vertex, because the table is flushed vertex-major: the lanes of one
// instruction add to consecutive addresses).  Open addressing on the vertex index; a corner that finds no slot after VT_PROBES
// steps falls back to the direct atomics.  Only KIND_INTERP triangles with nb_colors <= 4 (row: ij 2 + colours 4).
#ifndef DR_FIN_MERGE
#define DR_FIN_MERGE 1 // (measurement builds: 0 = every contribution straight to the gradient arrays, as in round 3)
#endif
constexpr int VT_SLOTS = 512, VT_ROW = 6, VT_PROBES = 8;
struct VertexTable
{
	uint32_t key[VT_SLOTS];
	double val[VT_SLOTS][VT_ROW];
};
__device__ __forceinline__ int vertex_slot(VertexTable &t, uint32_t v)
{
	uint32_t h = (v * 2654435761u) >> (32 - 9);
	static_assert(VT_SLOTS == 512, "hash width");
#pragma unroll 1
	for (int i = 0; i < VT_PROBES; i++, h = (h + 1) & (VT_SLOTS - 1))
	{
		const uint32_t old = atomicCAS(&t.key[h], 0xffffffffu, v);
		if (old == 0xffffffffu || old == v)
			return (int)h;
	}
	return -1;
}
struct MergeSink // finalize_triangle's sink: corner i of the triangle adds into row slot[i] of the table (or, without a slot, to memory)
{
	const SceneView &s;
	const GradView &g;
	VertexTable &t;
	uint32_t f[3];
	int slot[3];
	__device__ __forceinline__ void put(int i, int col, void *arr, size_t at, double v)
	{
		if (v == 0)
			return;
		if (slot[i] >= 0)
			unsafeAtomicAdd(&t.val[slot[i]][col], v);
		else
			DeviceAdd()(arr, at, s.vtx_f64, v);
	}
	__device__ __forceinline__ void color(int i, int c, double v) { put(i, 2 + c, g.colors_b, (size_t)f[i] * s.C + c, v); }
	__device__ __forceinline__ void ij(int i, int d, double v) { put(i, d, g.ij_b, 2 * (size_t)f[i] + d, v); }
	__device__ __forceinline__ void shade(int, double) {}	  // (KIND_INTERP triangles only)
	__device__ __forceinline__ void uv(int, int, double) {} // (KIND_INTERP triangles only)
};

// DET: the deterministic mode (KParams::det): accumulators are read as int64 fixed point, contributions go to the int64 shadow arrays,
// no vertex table (its LDS atomics are shared by four wavefronts: their order is not reproducible).
// TABLE: the instance with the per-workgroup vertex table (launches of >= DR_PRIM_TABLES_MIN triangles: KParams::prim_tables); the one
// without keeps round 3's registers and LDS -- a single 20 k-triangle view is a chain of round trips and lost 1.7 us to the table's mere presence.
template <bool VTX64, int NC, bool DET = false, bool TABLE = false> // (the dtype of the vertex arrays and the channel count at compile time: see setup_bin_kernel)
#ifndef DR_FIN_WAVES
#define DR_FIN_WAVES 4 // waves per SIMD finalize_kernel is compiled for (3: 144 registers with the vertex table, 21.4 -> 22.4 us)
#endif
__device__ __forceinline__ void finalize_body(KParams &p)
{ // same split as setup_bin_kernel: triangle blocks, then edge-slot blocks compacted to the flagged slots.
  // (Lists of the front-facing triangles / drawn edges compacted by the set-up kernel were tried: a quarter as many wavefronts,
  // all lanes busy -- and 32 -> 41 us: the kernel is a chain of dependent round trips, fewer wavefronts overlap fewer of them.)
	DR_WAVE_TRACE_SCOPE(1);
	p.vtx_f64 = VTX64 ? 1 : 0;
	if (NC)
		p.C = NC, p.L.P = NC < 3 ? 3 : NC;
	kernel_stamp(p, 2);
	const int loss_blocks = p.loss_out ? 1 : 0;
	if (loss_blocks && blockIdx.x == 0)
	{ // one extra workgroup, the FIRST of the grid (it overlaps the others): loss = background loss of the whole frame + the walkers'
	  // partials (complete: the forward raster is over; LOSS_SLOTS per view), then lanes (DPP tree) and wavefronts in order.
		__shared__ double s_loss[PRIM_BLOCK / 64];
		double s = 0;
		for (int j = threadIdx.x; j < p.n_views * LOSS_SLOTS; j += PRIM_BLOCK) // (one value per thread and view)
			s += p.loss_wave[j];
		s = wave_sum(s);
		if ((threadIdx.x & 63) == 0)
			s_loss[threadIdx.x >> 6] = s;
		__syncthreads();
		if (threadIdx.x == 0)
		{
			double sum = p.loss_tile_bg[0];
			for (int i = 0; i < PRIM_BLOCK / 64; i++)
				sum += s_loss[i];
			p.loss_out[0] = sum;
		}
		return;
	}
	const int fill_n = fill_share(p.fill_mode, 1, p.L.nwords), fill_blocks = (p.n_views * fill_n + PRIM_BLOCK / 64 - 1) / (PRIM_BLOCK / 64);
	const int bx = (int)blockIdx.x - loss_blocks;
	const int fb = DR_FILL_FIRST ? bx : bx - p.n_views * prim_blocks(p.T); // index among the fill workgroups
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
	const PrimWork pw = prim_work(p, DR_FIN_EDGE_FIRST, (DR_FILL_FIRST ? fill_blocks : 0) + loss_blocks);
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
	if constexpr (DET)
	{ // the deterministic mode: plain and slow (one round trip after the other) -- it exists for tests, not for speed
		const DetAdd dadd = {g.ij_b, g.colors_b, g.shade_b, p.det_ij + (size_t)view * p.V * 2, p.det_colors + (size_t)view * p.V * p.C,
							 p.det_shade + (size_t)view * p.V, p.det_uv};
		double la[3 * DEODR_HIP_MAX_COLORS + 3];
		if (tri_block)
		{
			const int k = pw.index * PRIM_BLOCK + threadIdx.x;
			if (k >= p.T)
				return;
			const uint32_t flag = w.tri_flag[k];
			const uint32_t f0 = p.faces[3 * (size_t)k], f1 = p.faces[3 * (size_t)k + 1], f2 = p.faces[3 * (size_t)k + 2];
			if (!(flag & 4u) || (flag & 3u) == KIND_NONE || (int32_t)(f0 | f1 | f2) < 0)
				return;
			double *acc = w.tri_acc + (size_t)k * 3 * P;
			for (int i = 0; i < 3 * P; i++)
				la[i] = det_value(acc + i);
			AtomicSinkT<DetAdd> sink = {s, g, {f0, f1, f2}, {p.faces_uv[3 * (size_t)k], p.faces_uv[3 * (size_t)k + 1], p.faces_uv[3 * (size_t)k + 2]}, dadd};
			finalize_triangle<false>(s, k, (int)(flag & 3u), la, sink);
			for (int i = 0; i < 3 * P; i++)
				acc[i] = 0;
			return;
		}
		const uint32_t n_flagged = compact_flagged_slots(p, s.edgeflags, pw.index);
		for (int round = 0; round * PRIM_BLOCK < (int)n_flagged; round++)
		{
			const int slot = edge_round_slot(n_flagged, round);
			if (slot < 0)
				continue;
			const EdgeRec &er = w.edge_rec[slot];
			if (er.kind == KIND_NONE)
				continue;
			double *acc = w.edge_acc + (size_t)slot * (3 * P + 3);
			for (int i = 0; i < 3 * P + 3; i++)
				la[i] = det_value(acc + i);
			finalize_edge(s, g, slot / 3, slot % 3, er, la, dadd);
			for (int i = 0; i < 3 * P + 3; i++)
				acc[i] = 0;
		}
		return;
	}
	if (tri_block)
	{
		const int k = pw.index * PRIM_BLOCK + threadIdx.x;
		// the vertex indices are requested together with the flag (one memory round trip, not two): using them in the branch
		// condition keeps the compiler from sinking the loads below it (an index never has its top bit set: V < 2^31)
		uint32_t flag = 0, f0 = 0, f1 = 0, f2 = 0;
		if (k < p.T)
		{
			flag = w.tri_flag[k];
			f0 = p.faces[3 * (size_t)k], f1 = p.faces[3 * (size_t)k + 1], f2 = p.faces[3 * (size_t)k + 2];
		}
		// culled triangles own no accumulators
		const bool live = k < p.T && (flag & 4u) && (flag & 3u) != KIND_NONE && (int32_t)(f0 | f1 | f2) >= 0;
		// (the table's 26 KB are only touched by a block that has a triangle for it: half the blocks of a closed mesh are all back-facing)
		__shared__ __attribute__((aligned(16))) char s_vt_storage[TABLE ? sizeof(VertexTable) : 16]; // (no table, no LDS for it)
		VertexTable &s_vt = *(VertexTable *)s_vt_storage;
		const bool merge = TABLE && DR_FIN_MERGE && p.prim_tables && P <= 4 && __syncthreads_or(live && (flag & 3u) == KIND_INTERP);
		if (merge)
		{
			for (int i = threadIdx.x; i < VT_SLOTS; i += PRIM_BLOCK)
				s_vt.key[i] = 0xffffffffu;
			for (int i = threadIdx.x; i < VT_SLOTS * VT_ROW; i += PRIM_BLOCK)
				(&s_vt.val[0][0])[i] = 0;
			__syncthreads();
		}
		if (live)
		{
			double *acc = w.tri_acc + (size_t)k * 3 * P;
			DR_WAVE_PHASE_T(1); // flags + indices arrived
			if (P <= 4)
			{ // a register copy of the accumulators: all twelve loads in flight together (read through the pointer, each plane's
			  // loads would wait behind the atomics of the plane before: they might alias)
				double la[12];
#pragma unroll
				for (int i = 0; i < 12; i++)
					la[i] = i < 3 * P ? acc[i] : 0.0;
				if (merge && (flag & 3u) == KIND_INTERP)
				{
					MergeSink sink = {s, g, s_vt, {f0, f1, f2}, {vertex_slot(s_vt, f0), vertex_slot(s_vt, f1), vertex_slot(s_vt, f2)}};
					finalize_triangle<true>(s, k, KIND_INTERP, la, sink);
				}
				else
				{
					AtomicSink sink = {s, g, {f0, f1, f2}, {p.faces_uv[3 * (size_t)k], p.faces_uv[3 * (size_t)k + 1], p.faces_uv[3 * (size_t)k + 2]}};
					finalize_triangle<true>(s, k, (int)(flag & 3u), la, sink);
				}
			}
			else
			{
				AtomicSink sink = {s, g, {f0, f1, f2}, {p.faces_uv[3 * (size_t)k], p.faces_uv[3 * (size_t)k + 1], p.faces_uv[3 * (size_t)k + 2]}};
				finalize_triangle<false>(s, k, (int)(flag & 3u), acc, sink);
			}
			DR_WAVE_PHASE_T(2); // arithmetic done, contributions issued
			for (int i = 0; i < 3 * P; i++)
				acc[i] = 0; // self-cleaning accumulators
			DR_WAVE_PHASE_T(3);
		}
		if (merge)
		{ // flush, vertex-major: eight lanes per row (six used), i.e. one instruction adds to the 16 + 32 contiguous bytes of eight vertices
			__syncthreads();
			for (int i = threadIdx.x; i < VT_SLOTS * 8; i += PRIM_BLOCK)
			{
				const int row = i >> 3, col = i & 7;
				const uint32_t v = s_vt.key[row];
				if (col < 2 + p.C && v != 0xffffffffu)
				{
					const double x = s_vt.val[row][col];
					if (col < 2)
						DeviceAdd()(g.ij_b, 2 * (size_t)v + col, p.vtx_f64, x);
					else
						DeviceAdd()(g.colors_b, (size_t)v * p.C + (col - 2), p.vtx_f64, x);
				}
			}
		}
		return;
	}
	const uint32_t n_flagged = compact_flagged_slots(p, s.edgeflags, pw.index);
	DR_WAVE_PHASE(1); // flags compacted
	for (int round = 0; round * PRIM_BLOCK < (int)n_flagged; round++)
	{ // (a round per PRIM_BLOCK flagged slots: see setup_bin_kernel)
		const int slot = edge_round_slot(n_flagged, round);
		if (slot < 0)
			continue;
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
				continue;
			if (fin.has_att)
				finalize_edge_fin(s, g, kind, x2b, fin, la, lt, DeviceAdd());
			else
				finalize_edge(s, g, slot / 3, slot % 3, er, acc, DeviceAdd());
		}
		else
		{
			if (kind == KIND_NONE)
				continue;
			finalize_edge(s, g, slot / 3, slot % 3, er, acc, DeviceAdd());
		}
		DR_WAVE_PHASE(3); // arithmetic done, atomics issued
		for (int i = 0; i < 3 * P + 3; i++)
			acc[i] = 0;
		DR_WAVE_PHASE(4);
	}
}

// The step-done flag: every wavefront counts itself when its gradient contributions have been executed; the last one stores the flag.
// The gradients leave this kernel as atomics, which are executed at the memory side: a wavefront whose vmcnt has
// reached zero has them in memory, for every XCD to see, and nothing else is promised by the flag -- a release fence per wavefront
// (= a write-back of its XCD's L2, 24 000 times per step) made the kernel 160 us longer.  Two levels of counters (DONE_SUBS of them a
// cache line apart, then one): the returning atomics on ONE word are executed one after the other -- 24 000 of them, 55 us.
// The counters are left zero: whoever completes one resets it.
__device__ __forceinline__ void step_done_signal(const KParams &p)
{
	__atomic_signal_fence(__ATOMIC_SEQ_CST);
	__builtin_amdgcn_s_waitcnt(0); // vmcnt(0) expcnt(0) lgkmcnt(0): this wavefront's atomics have been acknowledged
	__atomic_signal_fence(__ATOMIC_SEQ_CST);
	if ((threadIdx.x & 63) == 0)
	{
		uint32_t *counts = (uint32_t *)(p.ws + p.L.done_counts); // view 0's
		const uint32_t waves = gridDim.x * (PRIM_BLOCK / 64), id = blockIdx.x * (PRIM_BLOCK / 64) + (threadIdx.x >> 6), sub = id % DONE_SUBS;
		const uint32_t expected = (waves - sub + DONE_SUBS - 1) / DONE_SUBS; // wavefronts whose id is sub (mod DONE_SUBS)
		if (__hip_atomic_fetch_add(&counts[sub * DONE_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expected - 1u)
		{
			__hip_atomic_store(&counts[sub * DONE_STRIDE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const uint32_t subs = waves < (uint32_t)DONE_SUBS ? waves : (uint32_t)DONE_SUBS;
			if (__hip_atomic_fetch_add(&counts[DONE_SUBS * DONE_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == subs - 1u)
			{
				__hip_atomic_store(&counts[DONE_SUBS * DONE_STRIDE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(p.done_flag, p.done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
	}
}

template <bool VTX64, int NC, bool DET = false, bool TABLE = false>
__global__ __launch_bounds__(PRIM_BLOCK, DR_FIN_WAVES) void finalize_kernel(KParams p)
{
	finalize_body<VTX64, NC, DET, TABLE>(p);
	if (p.done_flag)
		step_done_signal(p);
}

// ---- the step-done flag (DeodrHipFitOptions::done_flag) and its consumer's side
__global__ void store_flag_kernel(uint32_t *flag, uint32_t value) { __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

// one lane polls (agent scope: past the L2 of its XCD) until *flag has reached `value` (serial-number arithmetic), for at most `ticks` of
// the 100 MHz counter; a wait that timed out leaves status[0] = 1 and every later wait on that status word returns at once
__global__ void wait_flag_kernel(const uint32_t *flag, uint32_t value, uint32_t *status, unsigned long long ticks)
{
	if (threadIdx.x != 0)
		return;
	if (status && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
		return;
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	for (;;)
	{
		// RELAXED: an acquire load at agent scope invalidates the L2 of the XCD this wavefront sits on at every poll -- the forward raster that
		// runs beside it lost 3 us per step to that (rocprofv3 trace, profiles/r04v_*); what is acquired here is acquired by the end of
		// this kernel and the start of the next one on the stream
		const uint32_t v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if ((int32_t)(v - value) >= 0)
			return;
		if (__builtin_amdgcn_s_memrealtime() - t0 > ticks)
		{
			if (status)
				__hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			return;
		}
		__builtin_amdgcn_s_sleep(16); // (~0.5 us)
	}
}



// Deterministic mode, last step: every element of a gradient array receives the integer sum of its shadow (one thread per element: a
// fixed order of two operands) and the shadow is cleared for the next call.
__global__ __launch_bounds__(256) void det_convert_kernel(long long *shadow, void *out, size_t n, int f64)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n)
		return;
	const long long v = shadow[i];
	if (v == 0)
		return;
	shadow[i] = 0;
	if (f64)
		((double *)out)[i] += (double)v * DET_INV_SCALE;
	else
		((float *)out)[i] += (float)((double)v * DET_INV_SCALE);
}

// ------------------------------------------------------------------------ finalize under the forward raster (round 4)
//
// In a fit step of an untextured scene the work of finalize_kernel is done by workgroups at the END OF THE GRID of
// raster_fwd_fast_kernel (one wavefront each: 64 triangles, or 64 entries at a time of the list of drawn edges).  finalize_kernel
// is a chain of dependent round trips that leaves the vector units idle (11 % busy), the forward raster is bound by vector issue
// and ends with a tail of long tiles during which most wave slots are empty (tools/wave_trace.py): run one under the other.
// A primitive may be finalized as soon as every tile it was binned to has been walked.  The walkers count finished tiles per
// BLOCK of BLK x BLK tiles (signal_tiles_done), the scan kernel has counted the non-empty tiles of every block, and a finalize
// lane waits for the blocks under the bounding box of its primitive -- not for its view: the one tile of 43 triangles and 37
// edges that a view waits 50 us for delays the handful of primitives around it, nothing else.
// Deadlock: a finalize workgroup only waits for walkers, walkers wait for nobody, and every walker has a lower index in the grid:
// on the XCD a workgroup is dispatched to, workgroups are dispatched in order, so a walker that has not been dispatched yet never
// finds its XCD occupied by finalize workgroups.  The wait is bounded all the same (a hung GPU box is worse than a wrong gradient
// that the status block reports): DEODR_HIP_ERR_INTERNAL.
struct FinTable // vertex table of ONE wavefront (64 triangles, ~70 distinct vertices of a mesh in strip order)
{
	static constexpr int SLOTS = 128, PROBES = 8;
	uint32_t key[SLOTS];
	double val[SLOTS][VT_ROW];
};
static_assert(sizeof(FinTable) == FIN_LDS_BYTES, "carved out of the walkers' staging area (dr_forward.h)");
struct FinMergeSink
{
	const SceneView &s;
	const GradView &g;
	FinTable &t;
	uint32_t f[3];
	int slot[3];
	__device__ __forceinline__ void put(int i, int col, void *arr, size_t at, double v)
	{
		if (v == 0)
			return;
		if (slot[i] >= 0)
			unsafeAtomicAdd(&t.val[slot[i]][col], v);
		else
			DeviceAdd()(arr, at, s.vtx_f64, v);
	}
	__device__ __forceinline__ void color(int i, int c, double v) { put(i, 2 + c, g.colors_b, (size_t)f[i] * s.C + c, v); }
	__device__ __forceinline__ void ij(int i, int d, double v) { put(i, d, g.ij_b, 2 * (size_t)f[i] + d, v); }
	__device__ __forceinline__ void shade(int, double) {}
	__device__ __forceinline__ void uv(int, int, double) {}
};
__device__ __forceinline__ int fin_vertex_slot(FinTable &t, uint32_t v)
{
	uint32_t h = (v * 2654435761u) >> (32 - 7);
	static_assert(FinTable::SLOTS == 128, "hash width");
#pragma unroll 1
	for (int i = 0; i < FinTable::PROBES; i++, h = (h + 1) & (FinTable::SLOTS - 1))
	{
		const uint32_t old = atomicCAS(&t.key[h], 0xffffffffu, v);
		if (old == 0xffffffffu || old == v)
			return (int)h;
	}
	return -1;
}

__device__ __forceinline__ uint32_t sync_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double sync_load(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Blocks of tiles under the pixel box [x0, x1] x [y0, y1] (a superset of the primitive's tiles: waiting for a neighbouring block too
// costs nothing), and whether all their walkers have signalled.
struct BlockBox
{
	int bx0, bx1, by0, by1;
};
__device__ __forceinline__ BlockBox block_box(const KParams &p, double x0, double x1, double y0, double y1)
{
	const double side = (double)(BLK * TILE);
	const double fx0 = floor(x0 / side), fx1 = floor(x1 / side), fy0 = floor(y0 / side), fy1 = floor(y1 / side);
	BlockBox b; // (clamped in double: a vertex may lie anywhere; NaN compares false and ends in an empty range)
	b.bx0 = fx0 > 0 ? (fx0 < p.L.blk_x ? (int)fx0 : p.L.blk_x) : 0;
	b.by0 = fy0 > 0 ? (fy0 < p.L.blk_y ? (int)fy0 : p.L.blk_y) : 0;
	b.bx1 = fx1 < p.L.blk_x - 1 ? (fx1 >= 0 ? (int)fx1 : -1) : p.L.blk_x - 1;
	b.by1 = fy1 < p.L.blk_y - 1 ? (fy1 >= 0 ? (int)fy1 : -1) : p.L.blk_y - 1;
	return b;
}
__device__ __forceinline__ bool box_ready(const KParams &p, const ViewPtrs &w, const BlockBox &bb)
{
	const uint32_t *expected = w.blk_sync, *done = w.blk_sync + p.L.nblk;
	bool all = true;
	for (int by = bb.by0; by <= bb.by1; by++)
		for (int bx = bb.bx0; bx <= bb.bx1; bx++)
		{
			const int b = by * p.L.blk_x + bx;
			all = all && sync_load(done + b) >= sync_load(expected + b);
		}
	return all;
}
// All 64 lanes: work() runs once in every lane that `need`s it, as soon as the walkers under that lane's box have signalled -- lane by
// lane, not wavefront by wavefront: 64 consecutive triangles of a mesh in strip order reach from the middle of the object to its
// limb, where the tiles with silhouette edges are the last to finish (tools/wave_trace.py: waited for as a wavefront, EVERY finalize
// wavefront of a view waited 44 us for its slowest lane).  work() must be lane-local (no cross-lane operation).
template <class Work>
__device__ __forceinline__ void run_when_ready(const KParams &p, const ViewPtrs &w, bool need, const BlockBox &bb, Work work)
{
	bool pending = need;
	uint32_t polls = 0;
	while (__ballot(pending) != 0)
	{
		bool go = pending && box_ready(p, w, bb);
		if (__ballot(go) == 0)
		{
			__builtin_amdgcn_s_sleep(32);
			if (++polls <= (1u << 22))
				continue;
			// (minutes: never reached unless the protocol is broken -- say so, then finish: a hung GPU box is worse than a wrong gradient)
			atomicOr(&w.hdr->scene_errors, (uint32_t)DEODR_HIP_ERR_INTERNAL);
			go = pending;
		}
		__atomic_signal_fence(__ATOMIC_SEQ_CST);
		if (go)
			work();
		pending = pending && !go;
	}
}

// One triangle per lane (k < 0: none): vertex colours interpolated linearly (the only kind a scene without texture draws; the set-up kernel
// drops the others with DEODR_HIP_ERR_NO_TEXTURE), at most four channels, written for FEW REGISTERS: this code shares the register budget of
// the tile walkers (96), where finalize_triangle -- every input in flight at once, the reference's cofactor sweep for the adjoint of the
// 3 x 3 inverse -- spilled 280.  Same algebra (H.h:841-858) with the channels streamed two at a time and  S_B = -T^T T_B T^T  for the
// inverse's adjoint (T = S^-1; agrees with the sweep to rounding).  All 64 lanes call it; contributions go to the wavefront's table.
__device__ __forceinline__ void fin_role_triangles(const KParams &p, const SceneView &s, const ViewPtrs &w, const GradView &g, FinTable &vt, int k)
{
	const int P = s.P;
	uint32_t flag = 0, f0 = 0, f1 = 0, f2 = 0;
	if (k >= 0)
	{
		flag = w.tri_flag[k];
		f0 = p.faces[3 * (size_t)k], f1 = p.faces[3 * (size_t)k + 1], f2 = p.faces[3 * (size_t)k + 2];
	}
	const bool live = k >= 0 && (flag & 4u) && (flag & 3u) == KIND_INTERP && (int32_t)(f0 | f1 | f2) >= 0 && P <= 4;
	if (__ballot(live) == 0)
		return;
	double xa = 0, xb = 0, ya = 0, yb = 0;
	double vx[3] = {0, 0, 0}, vy[3] = {0, 0, 0};
	if (live)
	{ // the pixel box of the triangle, two pixels wider than its vertices (fill rules, pixel-centre offset)
		vx[0] = ldv(s.ij, 2 * (size_t)f0, s.vtx_f64), vy[0] = ldv(s.ij, 2 * (size_t)f0 + 1, s.vtx_f64);
		vx[1] = ldv(s.ij, 2 * (size_t)f1, s.vtx_f64), vy[1] = ldv(s.ij, 2 * (size_t)f1 + 1, s.vtx_f64);
		vx[2] = ldv(s.ij, 2 * (size_t)f2, s.vtx_f64), vy[2] = ldv(s.ij, 2 * (size_t)f2 + 1, s.vtx_f64);
		xa = fmin(vx[0], fmin(vx[1], vx[2])) - 2, xb = fmax(vx[0], fmax(vx[1], vx[2])) + 2;
		ya = fmin(vy[0], fmin(vy[1], vy[2])) - 2, yb = fmax(vy[0], fmax(vy[1], vy[2])) + 2;
	}
	run_when_ready(p, w, live, block_box(p, xa, xb, ya, yb), [&]() {
		double *acc = w.tri_acc + (size_t)k * 3 * P;
		FinMergeSink sink = {s, g, vt, {f0, f1, f2}, {fin_vertex_slot(vt, f0), fin_vertex_slot(vt, f1), fin_vertex_slot(vt, f2)}};
		double T[9];
		{
			const double S[9] = {vx[0] - s.offset, vx[1] - s.offset, vx[2] - s.offset, vy[0] - s.offset, vy[1] - s.offset, vy[2] - s.offset, 1, 1, 1};
			inv3(S, T);
		}
		double TB[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
		const uint32_t fk[3] = {f0, f1, f2};
#pragma unroll 1
		for (int c0 = 0; c0 < p.C; c0 += 2)
		{
			double M[2][3], a[2][3];
#pragma unroll
			for (int h = 0; h < 2; h++)
#pragma unroll
				for (int j = 0; j < 3; j++)
				{
					const bool on = c0 + h < p.C;
					M[h][j] = on ? sync_load(acc + 3 * (c0 + h) + j) : 0.0;
					a[h][j] = on ? ldv(s.colors, (size_t)fk[j] * p.C + c0 + h, s.vtx_f64) : 0.0;
				}
#pragma unroll
			for (int h = 0; h < 2; h++)
#pragma unroll
				for (int kk = 0; kk < 3; kk++)
				{
					double a_B = 0;
#pragma unroll
					for (int j = 0; j < 3; j++)
					{
						a_B += M[h][j] * T[3 * kk + j];
						TB[3 * kk + j] += a[h][kk] * M[h][j];
					}
					if (c0 + h < p.C)
						sink.color(kk, c0 + h, a_B);
				}
		}
		double U[9]; // T_B T^T
#pragma unroll
		for (int kk = 0; kk < 3; kk++)
#pragma unroll
			for (int m = 0; m < 3; m++)
				U[3 * kk + m] = TB[3 * kk] * T[3 * m] + TB[3 * kk + 1] * T[3 * m + 1] + TB[3 * kk + 2] * T[3 * m + 2];
#pragma unroll
		for (int d = 0; d < 2; d++)
#pragma unroll
			for (int v = 0; v < 3; v++)
				sink.ij(v, d, -(T[d] * U[v] + T[3 + d] * U[3 + v] + T[6 + d] * U[6 + v]));
		for (int i = 0; i < 3 * P; i++)
			acc[i] = 0; // self-cleaning accumulators
	});
}

// One drawn silhouette edge per lane (slot < 0: none).  Vertex colours only, at most four channels, and again written for few registers:
// the record already holds the inverse frame (x2b, sigma x2t = the three rows of x2e, H.h:1407-1435), the channels are streamed two at a
// time, and the adjoint of the inverse is -x2e^T x2e_B x2e^T.  Contributions go straight to the gradient arrays (a few hundred edges per view).
__device__ __forceinline__ void fin_role_edges(const KParams &p, const SceneView &s, const ViewPtrs &w, const GradView &g, int slot)
{
	const int P = s.P;
	const bool mine = slot >= 0;
	const EdgeRec &er = w.edge_rec[mine ? slot : 0];
	const EdgeFin &fin = w.edge_fin[mine ? slot : 0];
	double *acc = w.edge_acc + (size_t)(mine ? slot : 0) * (3 * P + 3);
	double V[2][2] = {{0, 0}, {0, 0}};
	uint32_t vid[2] = {0, 0};
	bool live = false;
	double xa = 0, xb = 0, ya = 0, yb = 0;
	if (mine)
	{
		live = er.kind == KIND_INTERP && fin.has_att && P <= 4;
		V[0][0] = fin.V[0][0], V[0][1] = fin.V[0][1], V[1][0] = fin.V[1][0], V[1][1] = fin.V[1][1];
		vid[0] = fin.vid[0], vid[1] = fin.vid[1];
		const double m = p.sigma + 2 + p.offset;
		xa = fmin(V[0][0], V[1][0]) - m, xb = fmax(V[0][0], V[1][0]) + m;
		ya = fmin(V[0][1], V[1][1]) - m, yb = fmax(V[0][1], V[1][1]) + m;
	}
	run_when_ready(p, w, live, block_box(p, xa, xb, ya, yb), [&]() {
		double X[9], XB[9]; // x2e and its adjoint, row-major: rows 0, 1 = x2b, row 2 = sigma x2t
#pragma unroll
		for (int i = 0; i < 6; i++)
			X[i] = er.x2b[i], XB[i] = 0;
#pragma unroll
		for (int i = 0; i < 3; i++)
		{
			X[6 + i] = p.sigma * er.x2t[i];
			XB[6 + i] = sync_load(acc + 3 * P + i) * (1 / p.sigma); // x2t = x2e[2] / sigma (H.h:1430-1435)
		}
#pragma unroll 1
		for (int c0 = 0; c0 < p.C; c0 += 2)
		{
			double M[2][3], a[2][2];
#pragma unroll
			for (int h = 0; h < 2; h++)
			{
				const bool on = c0 + h < p.C;
#pragma unroll
				for (int j = 0; j < 3; j++)
					M[h][j] = on ? sync_load(acc + 3 * (c0 + h) + j) : 0.0;
				a[h][0] = on ? fin.att[0][c0 + h] : 0.0;
				a[h][1] = on ? fin.att[1][c0 + h] : 0.0;
			}
#pragma unroll
			for (int h = 0; h < 2; h++)
#pragma unroll
				for (int i = 0; i < 2; i++)
				{
					double a_B = 0;
#pragma unroll
					for (int j = 0; j < 3; j++)
					{
						a_B += M[h][j] * X[3 * i + j];
						XB[3 * i + j] += a[h][i] * M[h][j];
					}
					if (c0 + h < p.C)
						DeviceAdd()(g.colors_b, (size_t)vid[i] * p.C + c0 + h, p.vtx_f64, a_B);
				}
		}
		double U[9]; // x2e_B x2e^T
#pragma unroll
		for (int kk = 0; kk < 3; kk++)
#pragma unroll
			for (int m = 0; m < 3; m++)
				U[3 * kk + m] = XB[3 * kk] * X[3 * m] + XB[3 * kk + 1] * X[3 * m + 1] + XB[3 * kk + 2] * X[3 * m + 2];
		double EB[2][3]; // rows x, y of e2x_B = -x2e^T U: columns = vertex 0, vertex 1, normal
#pragma unroll
		for (int d = 0; d < 2; d++)
#pragma unroll
			for (int v = 0; v < 3; v++)
				EB[d][v] = -(X[d] * U[v] + X[3 + d] * U[3 + v] + X[6 + d] * U[6 + v]);
		// the normal n = nt / |nt| of the edge frame (edge_stencil_adjoint, H.h:1508-1538)
		double nt[2], inv_norm;
		edge_normal(V, p.clockwise != 0, nt, inv_norm);
		const double inv_norm_B = EB[0][2] * nt[0] + EB[1][2] * nt[1];
		const double nor_s_B = -inv_norm_B * (inv_norm * inv_norm) * 0.5 * inv_norm;
		const double ntx_B = EB[0][2] * inv_norm + 2 * nt[0] * nor_s_B, nty_B = EB[1][2] * inv_norm + 2 * nt[1] * nor_s_B;
		const double sgn = p.clockwise ? 1.0 : -1.0;
		DeviceAdd()(g.ij_b, 2 * (size_t)vid[0], p.vtx_f64, EB[0][0] - sgn * nty_B);
		DeviceAdd()(g.ij_b, 2 * (size_t)vid[0] + 1, p.vtx_f64, EB[1][0] + sgn * ntx_B);
		DeviceAdd()(g.ij_b, 2 * (size_t)vid[1], p.vtx_f64, EB[0][1] + sgn * nty_B);
		DeviceAdd()(g.ij_b, 2 * (size_t)vid[1] + 1, p.vtx_f64, EB[1][1] - sgn * ntx_B);
		for (int i = 0; i < 3 * P + 3; i++)
			acc[i] = 0;
	});
}

// Workgroup fi of the finalize workgroups of the forward raster: FIN_ROLES per view walk the view's list of WORK ITEMS (tile_scan_kernel:
// up to 64 triangles, or 64 drawn edges, of ONE block of tiles -- the set-up kernel files every primitive under the block of the first tile
// of its box --, or 64 entries of the overflow list), one more adds up the loss.
template <bool VTX64>
__device__ __forceinline__ void fin_in_fwd_role(const KParams &p0, char *lds, long long fi)
{
	KParams p = p0;
	p.vtx_f64 = VTX64 ? 1 : 0;
	DR_WAVE_TRACE_SCOPE(1); // (tools/wave_trace.py: these wavefronts take finalize_kernel's place in the trace)
	DR_WAVE_PHASE(1);
	const int lane = threadIdx.x & 63;
	const int roles = fin_roles_per_view(p.L.nblk);
	if (fi >= (long long)p.n_views * roles)
	{ // the one workgroup behind them all: the loss, once every walker of every view has added its partial sums
		if (!p.loss_out || fi > (long long)p.n_views * roles)
			return;
		uint32_t polls = 0;
		while (true)
		{
			bool ready = true;
			for (int v = lane; v < p.n_views; v += 64)
				ready = ready && sync_load(view_ptrs(p, v).blk_sync + 2 * p.L.nblk + SYNC_WALKERS) >= (uint32_t)p.tile_blocks;
			if (__ballot(!ready) == 0)
				break;
			__builtin_amdgcn_s_sleep(32);
			if (++polls > (1u << 22))
			{
				atomicOr(&view_ptrs(p, 0).hdr->scene_errors, (uint32_t)DEODR_HIP_ERR_INTERNAL);
				break;
			}
		}
		__atomic_signal_fence(__ATOMIC_SEQ_CST);
		double sum = 0;
		for (int j = lane; j < p.n_views * LOSS_SLOTS; j += 64)
			sum += sync_load(p.loss_wave + j);
		sum = wave_sum(sum);
		if (lane == 0)
			p.loss_out[0] = p.loss_tile_bg[0] + sum;
		return;
	}
	const int view = (int)(fi % p.n_views), role = (int)(fi / p.n_views);
	const SceneView s = scene_view(p, view);
	const ViewPtrs w = view_ptrs(p, view);
	const size_t es = p.vtx_f64 ? 8 : 4;
	GradView g;
	g.ij_b = (char *)p.ij_b + (size_t)view * p.V * 2 * es;
	g.colors_b = (char *)p.colors_b + (size_t)view * p.V * p.C * es;
	g.shade_b = (char *)p.shade_b + (size_t)view * p.V * es;
	g.uv_b = p.uv_b;
	FinTable &vt = *(FinTable *)lds;
	const uint32_t n_items = w.blk_sync[2 * p.L.nblk + SYNC_ITEMS]; // (written by the scan kernel, the launch before this one)
#pragma unroll 1
	for (uint32_t it = (uint32_t)role; it < n_items; it += (uint32_t)roles)
	{
		const uint2 item = w.fin_items[it];
		const int kind = (int)(item.x >> 28), n = (int)((item.x >> 20) & 0x3fu) + 1, blk = (int)(item.x & 0xfffffu);
		const bool has = lane < n;
		int tri = -1, edge = -1;
		if (kind == FIN_ITEM_TRI)
			tri = has ? (int)w.blk_lists[(size_t)blk * (BLK_TRI_CAP + BLK_EDGE_CAP) + item.y + lane] : -1;
		else if (kind == FIN_ITEM_EDGE)
			edge = has ? (int)w.blk_lists[(size_t)blk * (BLK_TRI_CAP + BLK_EDGE_CAP) + BLK_TRI_CAP + item.y + lane] : -1;
		else if (has)
		{ // overflow list: a triangle, or (top bit) an edge slot
			const uint32_t v = w.fin_overflow[item.y + lane];
			if (v & 0x80000000u)
				edge = (int)(v & 0x7fffffffu);
			else
				tri = (int)v;
		}
		if (__ballot(tri >= 0) != 0)
		{
			for (int i = lane; i < FinTable::SLOTS; i += 64)
				vt.key[i] = 0xffffffffu;
			for (int i = lane; i < FinTable::SLOTS * VT_ROW; i += 64)
				(&vt.val[0][0])[i] = 0;
			lds_sync();
			fin_role_triangles(p, s, w, g, vt, tri);
			DR_WAVE_PHASE(2); // every triangle of this item finalized into the table
			lds_sync();
			for (int i = lane; i < FinTable::SLOTS * 8; i += 64)
			{ // flush, vertex-major (see finalize_kernel)
				const int row = i >> 3, col = i & 7;
				const uint32_t v = vt.key[row];
				if (col < 2 + p.C && v != 0xffffffffu)
				{
					const double x = vt.val[row][col];
					if (col < 2)
						DeviceAdd()(g.ij_b, 2 * (size_t)v + col, p.vtx_f64, x);
					else
						DeviceAdd()(g.colors_b, (size_t)v * p.C + (col - 2), p.vtx_f64, x);
				}
			}
			lds_sync();
			DR_WAVE_PHASE(3);
		}
		if (__ballot(edge >= 0) != 0)
			fin_role_edges(p, s, w, g, edge);
	}
}

} // namespace
