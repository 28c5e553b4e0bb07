// deodr_amd/csrc/dr_finalize.h -- part of the single translation unit dr_kernels.hip (device code, gfx950 / wave64).
// finalize_kernel: moments -> plane adjoints -> adjoint of the 3 x 3 inverse -> vertex gradients.
#pragma once

#include "dr_backward.h"

using namespace dr;

namespace
{

// ------------------------------------------------------------------------------------------------------- finalize

template <bool VTX64, int NC> // (the dtype of the vertex arrays and the channel count at compile time: see setup_bin_kernel)
__global__ __launch_bounds__(PRIM_BLOCK, DR_PRIM_WAVES) void finalize_kernel(KParams p)
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
