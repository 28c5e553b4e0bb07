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
	// (small launches, KParams::setup_sparse: ONE edge slot per thread, as in the set-up kernel -- a soup's 600 flagged edges are three workgroups of one
	// round each instead of one workgroup of three rounds)
	const int edge_slots = p.setup_sparse > 1 ? 1 : EDGE_SLOTS, pblocks = prim_tri_blocks(p.T) + prim_edge_blocks(p.T, edge_slots);
	const int fb = DR_FILL_FIRST ? bx : bx - p.n_views * pblocks; // index among the fill workgroups
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
	const PrimWork pw = prim_work(p, DR_FIN_EDGE_FIRST, (DR_FILL_FIRST ? fill_blocks : 0) + loss_blocks, -1, edge_slots);
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
							 p.det_shade + (size_t)view * p.V, p.det_uv, p.det_err};
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
		const uint32_t n_flagged = compact_flagged_slots(p, s.edgeflags, pw.index, edge_slots);
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
	const uint32_t n_flagged = compact_flagged_slots(p, s.edgeflags, pw.index, edge_slots);
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



// Deterministic mode: DEODR_HIP_ERR_DET_RANGE speaks about ONE adjoint ("the gradients of that call are wrong"), so the bit is cleared when a
// deterministic adjoint starts (begin = 1: in view 0's sticky word, where det_add raises it, and in the polled word) and copied to the polled
// word when it ends (begin = 0) -- an asynchronous poller sees it in the same step, and a later, good call does not inherit it.
__global__ void det_status_kernel(WsHeader *view0, int begin)
{
	if (begin)
	{
		atomicAnd(&view0->scene_errors, ~(uint32_t)SCENE_ERR_DET_RANGE);
		atomicAnd(&view0->all_scene_errors, ~(uint32_t)SCENE_ERR_DET_RANGE);
	}
	else if (__hip_atomic_load(&view0->scene_errors, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (uint32_t)SCENE_ERR_DET_RANGE)
		atomicOr(&view0->all_scene_errors, (uint32_t)SCENE_ERR_DET_RANGE);
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

} // namespace
