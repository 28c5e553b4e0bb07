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

template <bool VTX64, int NC> // (the dtype of the vertex arrays and the channel count at compile time: see setup_bin_kernel)
#ifndef DR_FIN_WAVES
#define DR_FIN_WAVES 4 // waves per SIMD finalize_kernel is compiled for (3: 144 registers with the vertex table, 21.4 -> 22.4 us)
#endif
__global__ __launch_bounds__(PRIM_BLOCK, DR_FIN_WAVES) void finalize_kernel(KParams p)
{ // same split as setup_bin_kernel: triangle blocks, then edge-slot blocks compacted to the flagged slots.
  // (Lists of the front-facing triangles / drawn edges compacted by the set-up kernel were tried: a quarter as many wavefronts,
  // all lanes busy -- and 32 -> 41 us: the kernel is a chain of dependent round trips, fewer wavefronts overlap fewer of them.)
	DR_WAVE_TRACE_SCOPE(1);
	p.vtx_f64 = VTX64 ? 1 : 0;
	if (NC)
		p.C = NC, p.L.P = NC < 3 ? 3 : NC;
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
		__shared__ VertexTable s_vt;
		const bool merge = DR_FIN_MERGE && p.prim_tables && P <= 4 && __syncthreads_or(live && (flag & 3u) == KIND_INTERP);
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

} // namespace
