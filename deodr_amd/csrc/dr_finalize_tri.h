// deodr_amd/csrc/dr_finalize_tri.h -- part of the single translation unit dr_kernels.hip (device code, gfx950 / wave64).
// The per-triangle step of the adjoint (moments -> vertex gradients) as it is run from two places: finalize_kernel, and -- for
// the triangles whose accumulators are already complete when the forward raster ends -- extra workgroups of the edge-tile kernel.
#pragma once

#include "dr_backward_generic.h"

using namespace dr;

namespace
{

__device__ __forceinline__ GradView grad_view(const KParams &p, int view)
{
	const size_t es = p.vtx_f64 ? 8 : 4;
	GradView g;
	g.ij_b = (char *)p.ij_b + (size_t)view * p.V * 2 * es;
	g.colors_b = (char *)p.colors_b + (size_t)view * p.V * p.C * es;
	g.shade_b = (char *)p.shade_b + (size_t)view * p.V * es;
	g.uv_b = p.uv_b;
	return g;
}

// Triangle k (front-facing, of a drawable kind, valid indices f0..f2): moments -> plane adjoints -> adjoint of the 3 x 3 inverse ->
// atomic adds into the vertex gradients (H.h:838-858, 1138-1156); zeroes the accumulators it consumed.
__device__ __forceinline__ void finalize_tri_thread(const KParams &p, const SceneView &s, const ViewPtrs &w, const GradView &g, int k, int kind,
													uint32_t f0, uint32_t f1, uint32_t f2)
{
	const int P = s.P;
	double *acc = w.tri_acc + (size_t)k * 3 * P;
	AtomicSink sink = {s, g, {f0, f1, f2}, {p.faces_uv[3 * (size_t)k], p.faces_uv[3 * (size_t)k + 1], p.faces_uv[3 * (size_t)k + 2]}};
	if (P <= 4)
	{ // a register copy of the accumulators: all twelve loads in flight together (read through the pointer, each plane's
	  // loads would wait behind the atomics of the plane before: they might alias)
		double la[12];
#pragma unroll
		for (int i = 0; i < 12; i++)
			la[i] = i < 3 * P ? acc[i] : 0.0;
		finalize_triangle<true>(s, k, kind, la, sink);
	}
	else
		finalize_triangle<false>(s, k, kind, acc, sink);
	for (int i = 0; i < 3 * P; i++)
		acc[i] = 0; // self-cleaning accumulators
}

// Early finalize.  When the forward raster of a fit step (or the owner-tile kernel of the two-call path) ends, the accumulators of
// every triangle are complete EXCEPT those of the triangles listed in a tile that holds silhouette edges: only the edge-tile kernel
// adds to those.  That kernel is a few hundred persistent wavefronts walking long dependent chains -- most of the machine idles
// for its 30 us -- while finalize_kernel, the next launch, was bound by its number of wavefronts (12 000 at three per SIMD) and by
// the memory-side atomics of 7 000 triangles per view.  So the edge-tile kernel's launch carries one extra single-wave workgroup per
// 64 triangles: a triangle that is not stamped (tri_stamp != fwd_id, written by the forward raster for the triangles of edge tiles)
// is finalized right there; a stamped one is appended to late_list {k | kind << 30, f0, f1, f2} for finalize_kernel, which then only walks
// that list (~10 % of the front-facing triangles) and the silhouette edges.
__device__ __forceinline__ void finalize_early(const KParams &p, int view, int k0, int lane)
{
	const SceneView s = scene_view(p, view);
	const ViewPtrs w = view_ptrs(p, view);
	const GradView g = grad_view(p, view);
	const int k = k0 + lane;
	const uint32_t fwd_id = w.hdr->fwd_id;
	uint32_t flag = 0, f0 = 0, f1 = 0, f2 = 0, stamp = 0;
	if (k < p.T)
	{ // one round trip: flag, indices, stamp
		flag = w.tri_flag[k];
		f0 = p.faces[3 * (size_t)k], f1 = p.faces[3 * (size_t)k + 1], f2 = p.faces[3 * (size_t)k + 2];
		stamp = w.tri_stamp[k];
	}
	const bool takes_part = (flag & 4u) && (flag & 3u) != KIND_NONE && (int32_t)(f0 | f1 | f2) >= 0;
	const bool late = takes_part && stamp == fwd_id;
	const unsigned long long m = __ballot(late);
	if (m)
	{
		uint32_t base = 0;
		if (lane == 0)
			base = atomicAdd(&w.hdr->late_count, (uint32_t)__popcll(m));
		base = (uint32_t)uniform((int)base);
		if (late)
			w.late_list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = make_uint4((uint32_t)k | ((flag & 3u) << 30), f0, f1, f2); // (k < 2^30)
	}
	if (takes_part && !late)
		finalize_tri_thread(p, s, w, g, k, (int)(flag & 3u), f0, f1, f2);
}

} // namespace
