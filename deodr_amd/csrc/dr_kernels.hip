// deodr_amd/csrc/dr_kernels.hip -- the translation unit of libdeodr_hip.so: HIP kernels (gfx950 / CDNA4, wave64; in the
// headers listed after the kernels below) + the host side and the C ABI (this file).
//
// Kernels (n_views views per launch; DESIGN.md section 4 has the why of every choice below):
//
//   setup_bin_kernel        blocks of 256 threads on a 1-D grid, the edge blocks first: 256 edge slots compacted to the flagged
//                           ones, or one triangle per thread.  Cull, depth sum, stencil + attribute planes in double and in
//                           registers (dr_prims.h), edge records (+ the EdgeFin record finalize reads), the index checks of
//                           checkSceneValid, binning into 8 x 8 tiles (all slot requests of a 3 x 3 block of tiles in flight; a
//                           thread bins the first block of its primitive, the other blocks of all primitives of a wavefront
//                           are dealt out evenly over its lanes), optional gradient clearing
//   tile_scan_kernel        one thread per tile: counters -> work list of the NON-EMPTY tiles (entries carry the first triangle
//                           ids; many-primitive tiles first), tile bitmap, three lists of edge tiles by edge count, sweep slots
//   raster_fwd_fast_kernel  1 wavefront / work-list entry, lane = pixel.  Pass 1 (z-buffered triangles staged 16 at a time
//                           through LDS, exact scanline spans, winner = min (Z, index)), shading of the winner, pass 2 (ordered
//                           edge overdraw) fused in registers, ONE write of image / z (/ owner) per pixel.  FUSED
//                           (deodr_hip_render_scene_fit): also the adjoint for the sum-of-squares residual of EVERY tile (tiles with
//                           silhouette edges: reverse sweep + pass 1 by a second instance of the walker on the head of the work
//                           list; untextured scenes: pairs of adjacent edge-free tiles share a wavefront; textured scenes, round 5:
//                           instances of their own at three waves per SIMD, from 8 views per launch on as two kernels on two
//                           streams -- the head walkers / everybody else)
//   fill_kernel / fill_word background + depth = inf of the empty tiles, runs of tiles written as contiguous 16-byte pieces:
//                           a kernel on a forked stream (forward-only calls) or extra workgroups of the forward raster and of
//                           finalize, dealt 2 : 1 (fit step)
//   raster_bwd_fast_kernel  (two-call path) adjoint of pass 1 in every non-empty tile without edges
//   raster_bwd_edge_kernel  (two-call path) persistent waves over the listed edge tiles: adjoint of pass 2 (un-blend in reverse order, moments
//                           by a transposing butterfly, one 15-lane atomic per edge and tile), then of pass 1
//   finalize_kernel         per primitive: moments -> plane adjoints -> 3x3-inverse adjoint -> vertex gradients
//   raster_fwd_kernel / raster_bwd_kernel   the same algorithm without LDS staging: nb_colors > 4, antialiase_error
//   dr_fronthalf.h, dr_fititer.h   the O(V) kernels either side of the rasterizer in a fit iteration (pose + projection, shading,
//                           silhouette flags, their adjoints, rigid energy, data terms, momentum update): deterministic sums
//
// Where: dr_workspace.h (layout, KParams, wave primitives) . dr_setup.h (setup_bin_kernel) . dr_forward.h (tile_scan_kernel, fill,
// raster_fwd_fast_kernel) . dr_backward.h (raster_bwd_fast_kernel, raster_bwd_edge_kernel) . dr_finalize.h (finalize_kernel) .
// dr_forward_generic.h / dr_backward_generic.h (the un-staged family, edge ordering) . dr_math.h / dr_prims.h (per-primitive math) .
// dr_fronthalf.h / dr_fititer.h (fit iteration).
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
#include <type_traits>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/deodr_hip.h"
#include "dr_fititer.h" // <- dr_finalize.h <- dr_backward.h <- dr_backward_generic.h <- dr_forward.h <- dr_forward_generic.h <- dr_setup.h <- dr_workspace.h <- dr_prims.h

using namespace dr;

namespace
{

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
	// (see KParams::prim_tables; measured on the 20 k-triangle benchmark scene: 1 view loses 1.7 us to the tables, 8 views gain 6)
#ifndef DR_PRIM_TABLES_MIN
#define DR_PRIM_TABLES_MIN 40000
#endif
	p.prim_tables = (long long)sc->n_views * sc->nb_triangles >= DR_PRIM_TABLES_MIN;
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
unsigned long long *g_stamps = nullptr; // deodr_hip_profile_stamps: device buffer of g_stamp_rows x 4 words, one row per forward
int g_stamp_rows = 0;
unsigned g_stamp_calls = 0;
bool g_force_generic = false; // deodr_hip_force_generic(1): run the un-staged kernels (the parity suite covers both families)
bool g_det = false;			  // deodr_hip_set_deterministic(1): un-staged kernels + integer accumulation (KParams::det) for EVERY scene
inline bool det_mode(const DeodrHipScene *sc) { return g_det || sc->deterministic != 0; } // (DeodrHipScene::deterministic: for this scene)

// int64 shadows of the gradient arrays in the deterministic mode: one library-owned buffer per (device, stream), grown on demand (the only
// allocation the library ever makes, and only in this mode: a test mode), zero between calls (det_convert clears what it reads).  Per
// STREAM: calls on one stream are ordered, so they may share the shadows; two fit steps on two streams (or threads) of one device each
// get their own -- with one buffer per device their sums mixed.  Never freed (a handful of streams per process).
struct DetScratch
{
	int dev = -1;
	hipStream_t stream = nullptr;
	long long *ptr = nullptr;
	size_t words = 0;
};
std::mutex g_det_mutex;
std::vector<DetScratch> g_det_scratch;
int det_shadows(KParams &p, int n_views, hipStream_t st)
{
	const size_t n_ij = (size_t)n_views * p.V * 2, n_col = (size_t)n_views * p.V * p.C, n_sh = (size_t)n_views * p.V, n_uv = (size_t)p.Vuv * 2,
				 n_tex = p.texture_b ? (size_t)p.tex_h * p.tex_w * p.C : 0, need = n_ij + n_col + n_sh + n_uv + n_tex;
	int dev = 0;
	if (check_hip(hipGetDevice(&dev), "hipGetDevice"))
		return 1;
	std::lock_guard<std::mutex> lock(g_det_mutex);
	size_t at = 0;
	while (at < g_det_scratch.size() && !(g_det_scratch[at].dev == dev && g_det_scratch[at].stream == st))
		at++;
	if (at == g_det_scratch.size())
	{
		g_det_scratch.emplace_back();
		g_det_scratch[at].dev = dev, g_det_scratch[at].stream = st;
	}
	DetScratch &sc = g_det_scratch[at];
	if (sc.words < need)
	{
		hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
		(void)hipStreamIsCapturing(st, &cap);
		if (cap != hipStreamCaptureStatusNone)
			return fail("deterministic mode: the shadow buffer has to grow, which cannot happen under stream capture (run the step once before capturing)");
		if (check_hip(hipDeviceSynchronize(), "deterministic mode: synchronise"))
			return 1;
		if (sc.ptr)
			(void)hipFree(sc.ptr);
		sc.ptr = nullptr, sc.words = 0;
		if (check_hip(hipMalloc((void **)&sc.ptr, 8 * need), "deterministic mode: shadow buffer") || check_hip(hipMemset(sc.ptr, 0, 8 * need), "deterministic mode: clear"))
			return 1;
		sc.words = need;
	}
	p.det = 1;
	p.det_err = &((WsHeader *)(p.ws + p.L.hdr))->scene_errors; // (view 0: what deodr_hip_workspace_status reads; det_convert copies it to the polled word)
	hipLaunchKernelGGL(det_status_kernel, dim3(1), dim3(1), 0, st, (WsHeader *)(p.ws + p.L.hdr), 1); // (the range bit of the PREVIOUS deterministic adjoint goes)
	p.det_ij = sc.ptr;
	p.det_colors = p.det_ij + n_ij;
	p.det_shade = p.det_colors + n_col;
	p.det_uv = p.det_shade + n_sh;
	p.det_texture = n_tex ? p.det_uv + n_uv : nullptr;
	return 0;
}
void det_convert(const KParams &p, int n_views, hipStream_t st)
{
	const size_t n_ij = (size_t)n_views * p.V * 2, n_col = (size_t)n_views * p.V * p.C, n_sh = (size_t)n_views * p.V, n_uv = (size_t)p.Vuv * 2,
				 n_tex = p.det_texture ? (size_t)p.tex_h * p.tex_w * p.C : 0;
	auto run = [&](long long *shadow, void *out, size_t n, int f64) {
		if (n && out)
			hipLaunchKernelGGL(det_convert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, shadow, out, n, f64);
		else if (n && shadow) // (an array the caller did not ask for: whatever the kernels added to its shadow must not reach the next call)
			(void)hipMemsetAsync(shadow, 0, 8 * n, st);
	};
	run(p.det_ij, p.ij_b, n_ij, p.vtx_f64);
	run(p.det_colors, p.colors_b, n_col, p.vtx_f64);
	run(p.det_shade, p.shade_b, n_sh, p.vtx_f64);
	run(p.det_uv, p.uv_b, n_uv, p.vtx_f64);
	run(p.det_texture, p.texture_b, n_tex, p.pix_f64);
	hipLaunchKernelGGL(det_status_kernel, dim3(1), dim3(1), 0, st, (WsHeader *)(p.ws + p.L.hdr), 0);
}
// Tuning constants (measured in round 1, profiles/README.md); deliberately NOT read from the environment: nothing outside the
// arguments of a call may change what the call launches.
#ifndef DR_EDGE_WAVES
#define DR_EDGE_WAVES 1024
#endif
constexpr int EDGE_WAVES = DR_EDGE_WAVES; // persistent waves per view of the adjoint's edge kernel

// The channel count (and the "usual frame" flag) as compile-time constants of the raster kernels -- for float32 pixel buffers, the storage of the
// fit loops; with float64 buffers (the 1e-9 parity path, the NumPy drop-ins of the reference's entry points) every call takes the run-time-C
// instance: 28 raster instances fewer to compile (the library builds in ~3.5 minutes instead of ~4.7; their step is a few per cent longer).
template <class PixT, int NC>
constexpr int nc_for = sizeof(PixT) == 4 ? NC : 0;
template <class PixT>
constexpr bool common_for = sizeof(PixT) == 4;

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
		q.tile_blocks = fwd_tile_blocks(p.L.ntiles, p.n_views, false); // the grid of the forward that built the work list (never a fused one)
		q.heavy_share = heavy_share_for(p.n_views, q.tile_blocks, false);
		q.split_part = 16; // (no fit step here: nothing is split)
		const dim3 grid((unsigned)p.n_views * (unsigned)q.tile_blocks);
		// (instances for the channel counts that occur: RGB, RGB + depth -- see raster_fwd_fast_kernel)
#define DR_LAUNCH_NC(kernel, tex_, grid_, q_)                                                        \
	do                                                                                               \
	{                                                                                                \
		if (tex_ && (q_).C == 3)                                                                     \
			hipLaunchKernelGGL((kernel<PixT, true, nc_for<PixT, 3>>), grid_, dim3(64), 0, st, q_);                 \
		else if (tex_)                                                                               \
			hipLaunchKernelGGL((kernel<PixT, true, 0>), grid_, dim3(64), 0, st, q_);                 \
		else if ((q_).C == 4)                                                                        \
			hipLaunchKernelGGL((kernel<PixT, false, nc_for<PixT, 4>>), grid_, dim3(64), 0, st, q_);                \
		else if ((q_).C == 3)                                                                        \
			hipLaunchKernelGGL((kernel<PixT, false, nc_for<PixT, 3>>), grid_, dim3(64), 0, st, q_);                \
		else                                                                                         \
			hipLaunchKernelGGL((kernel<PixT, false, 0>), grid_, dim3(64), 0, st, q_);                \
	} while (0)
		DR_LAUNCH_NC(raster_bwd_fast_kernel, tex, grid, q);
	}
	// (running the two kernels side by side on a forked stream was measured: no gain, the edge kernel just stretches)
	if (p.sigma > 0 && p.aa_err)
	{ // antialiase_error: the sweep ran over the error buffer, and so does its adjoint (bwd_err_tile, dr_backward.h)
		const dim3 grid(edge_grid.x, edge_grid.y);
		if (tex)
			hipLaunchKernelGGL((raster_bwd_edge_err_kernel<PixT, true>), grid, dim3(64), 0, st, p);
		else
			hipLaunchKernelGGL((raster_bwd_edge_err_kernel<PixT, false>), grid, dim3(64), 0, st, p);
	}
	else if (p.sigma > 0 && !p.fuse_edges) // (a fit step back-propagates the tiles with edges inside its forward raster)
	{
		DR_LAUNCH_NC(raster_bwd_edge_kernel, tex, edge_grid, p);
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

// KParams::fwd_* / views_magic: what every workgroup of the staged forward (and of raster_bwd_fast_kernel, which walks the same list) would otherwise
// derive for itself (tile_blocks, heavy_share, fill_mode, n_views are set)
void forward_launch_constants(KParams &q)
{
	const bool chunked = q.tile_blocks % (8 * WORK_CHUNK) == 0;
	q.fwd_heads = chunked ? (uint32_t)(q.tile_blocks / q.heavy_share) : 0u;
	q.fwd_walkers = (uint32_t)q.tile_blocks;
	q.dyn_groups = 0;
#ifndef DR_DYN_MIN
#define DR_DYN_MIN 20000 // walkers (all views) from which the others' list of a fit step is walked by persistent walkers with tickets
#endif
#ifndef DR_DYN_TOTAL
#define DR_DYN_TOTAL 4608 // persistent walkers of all views together (the chip holds 5 120 wavefronts of the forward raster)
#endif
	if (DR_DYN_WALKERS && chunked && q.fuse_edges && q.texture == nullptr && (long long)q.n_views * q.tile_blocks >= DR_DYN_MIN)
	{
		const uint32_t others = ((uint32_t)DR_DYN_TOTAL / (uint32_t)q.n_views) & ~7u;
		if (others >= 8 && q.fwd_heads + others < (uint32_t)q.tile_blocks)
		{
			q.fwd_walkers = q.fwd_heads + others;
			q.dyn_groups = DYN_GROUPS;
		}
	}
	const uint32_t n_walk = (uint32_t)q.n_views * q.fwd_walkers;
	q.fwd_n_fill = (uint32_t)q.n_views * (uint32_t)fill_share(q.fill_mode, 2, q.L.nwords);
	if (q.dyn_groups)
		q.fwd_dealt = q.fwd_n_fill / 16 < n_walk / 64 ? q.fwd_n_fill / 16 : n_walk / 64; // sixteen fill workgroups behind every 64 walkers
	else
		q.fwd_dealt = (q.fuse_edges && n_walk >= 8 * q.fwd_n_fill) ? q.fwd_n_fill / 8 : 0u;
	q.views_magic = q.n_views == 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / (unsigned long long)q.n_views) + 1u; // (see div_views)
}

#ifndef DR_SPARSE_MAX
#define DR_SPARSE_MAX 16384 // triangles (all views) up to which the per-primitive kernels spread their work over more wavefronts (KParams::setup_sparse)
#endif
bool small_launch(int T, int n_views) { return (long long)T * n_views <= DR_SPARSE_MAX; }

// Staged forward: counters -> work list + tile bitmap (scan), then the raster on the caller's stream and, forked from it, the
// background fill on the side stream.  *join receives the event the caller's stream has to wait for before the call returns
// control to it (the fill overlaps whatever the call launches in between).
template <class PixT>
int launch_forward_staged(const KParams &p, bool fused, hipStream_t stream, hipEvent_t *join)
{
	KParams q = p;
	q.tile_blocks = fwd_tile_blocks(p.L.ntiles, p.n_views, fused && p.fuse_edges);
	q.heavy_share = heavy_share_for(p.n_views, q.tile_blocks, fused && p.fuse_edges);
	q.split_part = split_part_for(p.n_views, q.tile_blocks);
#ifndef DR_PAIR_TEX
#define DR_PAIR_TEX 1 // (measurement builds: 0 = textured scenes do not pair their tiles, as until round 6)
#endif
	q.pair_tex = DR_PAIR_TEX && p.texture != nullptr && (!DR_TEX_TWO_KERNELS || p.n_views < DR_TEX_TWO_KERNELS);
	forward_launch_constants(q);
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
	// (+ the workgroups that stream this kernel's share of the background of the empty tiles, one bitmap word each)
	const dim3 grid((unsigned)p.n_views * (unsigned)q.fwd_walkers + (unsigned)p.n_views * (unsigned)fill_share(p.fill_mode, 2, p.L.nwords));
	const bool tex = p.texture != nullptr; // (see launch_adjoint_raster)
	const bool common = p.strict && p.W % TILE == 0 && p.H % TILE == 0;
	hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
	if (DR_TEX_TWO_KERNELS && fused && tex && p.fuse_edges)
		(void)hipStreamIsCapturing(stream, &capturing);
	// (the split point must fall between two groups of eight workgroups -- a walker's list and XCD follow from its index in the one-kernel grid
	// (KParams::block_base) --: true for the shares heavy_share_for returns, checked here for measurement builds with another DR_HEAVY_SHARE)
	if (DR_TEX_TWO_KERNELS && fused && tex && p.fuse_edges && !p.clamp && p.n_views >= DR_TEX_TWO_KERNELS && q.tile_blocks % (8 * WORK_CHUNK) == 0 &&
		q.tile_blocks % q.heavy_share == 0 && (q.tile_blocks / q.heavy_share) % 8 == 0 && capturing == hipStreamCaptureStatusNone)
	{ // the head walkers (edge adjoint: many registers) on the side stream, everybody else (+ the fill workgroups) on the caller's, both behind the scan
		const unsigned head = (unsigned)p.n_views * (unsigned)(q.tile_blocks / q.heavy_share);
		std::lock_guard<std::mutex> lock(g_side_mutex);
		SideStream ss;
		if (side_stream(ss) || check_hip(hipEventRecord(ss.fork, stream), "fork") || check_hip(hipStreamWaitEvent(ss.stream, ss.fork, 0), "fork"))
			return 1;
		KParams rest = q;
		rest.block_base = head;
		if (p.C == 3)
		{
			hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, true, false, nc_for<PixT, 3>, false, 2>), dim3(head), dim3(64), 0, ss.stream, q);
			hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, true, false, nc_for<PixT, 3>, false, 3>), dim3(grid.x - head), dim3(64), 0, stream, rest);
		}
		else
		{
			hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, true, false, 0, false, 2>), dim3(head), dim3(64), 0, ss.stream, q);
			hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, true, false, 0, false, 3>), dim3(grid.x - head), dim3(64), 0, stream, rest);
		}
		if (check_hip(hipEventRecord(ss.join, ss.stream), "join") || check_hip(hipStreamWaitEvent(stream, ss.join, 0), "join"))
			return 1;
		return 0;
	}
	if (p.C > CH) // (more than CH channels: launch_forward sends only forward-only calls without edges and without texture here)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, false, false, false, 0, false, 0, 2>), grid, dim3(64), 0, stream, q);
	else if (p.aa_err && !fused && tex) // (antialiase_error: the edges blend the error buffer, the image stays un-antialiased)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, false, true, false, 0, false, 0, 1>), grid, dim3(64), 0, stream, q);
	else if (p.aa_err && !fused)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, false, false, false, 0, false, 0, 1>), grid, dim3(64), 0, stream, q);
	else if (fused && p.clamp && tex && p.fuse_edges) // (the clamped residual of the depth fitter: its own instances of the fused kernel)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, true, true, 0, false, 1>), grid, dim3(64), 0, stream, q);
	else if (fused && tex && p.fuse_edges && p.C == 3) // (textured fit step, sigma > 0: the instances with the edge adjoint)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, true, false, nc_for<PixT, 3>, false, 1>), grid, dim3(64), 0, stream, q);
	else if (fused && tex && p.fuse_edges)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, true, false, 0, false, 1>), grid, dim3(64), 0, stream, q);
	else if (fused && p.clamp && tex)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, true, true>), grid, dim3(64), 0, stream, q);
	else if (fused && p.clamp && p.C == 1) // (a depth image)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, false, true, nc_for<PixT, 1>>), grid, dim3(64), 0, stream, q);
	else if (fused && p.clamp)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, false, true>), grid, dim3(64), 0, stream, q);
	else if (fused && tex && p.C == 3) // (the channel counts that occur: RGB, RGB + depth; others take the run-time instance)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, true, false, nc_for<PixT, 3>>), grid, dim3(64), 0, stream, q);
	else if (fused && tex)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, true>), grid, dim3(64), 0, stream, q);
	else if (fused && p.C == 4 && common)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, false, false, nc_for<PixT, 4>, common_for<PixT>>), grid, dim3(64), 0, stream, q);
	else if (fused && p.C == 3 && common)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, false, false, nc_for<PixT, 3>, common_for<PixT>>), grid, dim3(64), 0, stream, q);
	else if (fused && p.C == 4)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, false, false, nc_for<PixT, 4>>), grid, dim3(64), 0, stream, q);
	else if (fused && p.C == 3)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, false, false, nc_for<PixT, 3>>), grid, dim3(64), 0, stream, q);
	else if (fused)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, true, false>), grid, dim3(64), 0, stream, q);
	else if (tex && p.C == 3)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, false, true, false, nc_for<PixT, 3>>), grid, dim3(64), 0, stream, q);
	else if (tex)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, false, true>), grid, dim3(64), 0, stream, q);
	else if (p.C == 4)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, false, false, false, nc_for<PixT, 4>>), grid, dim3(64), 0, stream, q);
	else if (p.C == 3)
		hipLaunchKernelGGL((raster_fwd_fast_kernel<PixT, false, false, false, nc_for<PixT, 3>>), grid, dim3(64), 0, stream, q);
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
	p.stamp = (g_stamps && g_stamp_calls < (unsigned)g_stamp_rows) ? g_stamps + 4 * (size_t)g_stamp_calls++ : nullptr;
	p.n_views = n_views;
	// (antialiase_error: staged since round 6, the AA instances of the forward raster; more than CH channels: staged when the frame has no silhouette
	// edge and no texture and nothing is fused -- Scene3D.render_deferred's frame, sigma = 0 -- through fwd_manyc_tile)
	const bool many_channels = p.C > CH && !(p.sigma > 0) && !p.aa_err && !p.texture && !fused;
	const bool fast = (p.C <= CH || many_channels) && !g_force_generic && !det_mode(sc);
	if (p.T > 0)
	{
		p.setup_sparse = small_launch(p.T, n_views) ? 4 : 1;
		dim3 grid((unsigned)(prim_tri_blocks(p.T * p.setup_sparse) + prim_edge_blocks(p.T, p.setup_sparse > 1 ? 1 : EDGE_SLOTS)) * (unsigned)n_views);
		ScopedKernelTimer t(KID_SETUP, stream);
		// (instances for the vertex dtype and for the channel counts that occur -- RGB, RGB + depth --; other counts: the run-time one)
#define DR_LAUNCH_PRIM(kernel, grid_, stream_)                                                                              \
	do                                                                                                                      \
	{                                                                                                                       \
		if (p.vtx_f64 && p.C == 4)                                                                                          \
			hipLaunchKernelGGL((kernel<true, 4>), grid_, dim3(PRIM_BLOCK), 0, stream_, p);                                  \
		else if (p.vtx_f64 && p.C == 3)                                                                                     \
			hipLaunchKernelGGL((kernel<true, 3>), grid_, dim3(PRIM_BLOCK), 0, stream_, p);                                  \
		else if (p.vtx_f64)                                                                                                 \
			hipLaunchKernelGGL((kernel<true, 0>), grid_, dim3(PRIM_BLOCK), 0, stream_, p);                                  \
		else if (p.C == 4)                                                                                                  \
			hipLaunchKernelGGL((kernel<false, 4>), grid_, dim3(PRIM_BLOCK), 0, stream_, p);                                 \
		else if (p.C == 3)                                                                                                  \
			hipLaunchKernelGGL((kernel<false, 3>), grid_, dim3(PRIM_BLOCK), 0, stream_, p);                                 \
		else                                                                                                                \
			hipLaunchKernelGGL((kernel<false, 0>), grid_, dim3(PRIM_BLOCK), 0, stream_, p);                                 \
	} while (0)
		DR_LAUNCH_PRIM(setup_bin_kernel, grid, stream);
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

template <class Kernel>
void launch_finalize(Kernel kernel, dim3 grid, hipStream_t st, const KParams &p)
{
	hipLaunchKernelGGL(kernel, grid, dim3(PRIM_BLOCK), 0, st, p);
}

// adjoint raster and the per-primitive finalize; owner_tiles = false after a fused forward
int launch_adjoint(const DeodrHipScene *sc, KParams &p, hipStream_t st, bool owner_tiles)
{
	// (antialiase_error, round 6: the tiles without silhouette edges through raster_bwd_fast_kernel -- their gradient is -2 (obs - image) err_buffer_b,
	// H.h:3054-3060 --, the tiles with edges through the un-staged tile code, called tile by tile from raster_bwd_edge_kernel's work lists)
	const bool fast = p.C <= CH && !g_force_generic && !det_mode(sc);
	p.n_views = sc->n_views;
	if (det_mode(sc) && det_shadows(p, sc->n_views, st))
		return 1;
	// persistent waves of the edge kernel: enough to cover a silhouette-heavy single view, few enough that with many views
	// the waves that find their sub-list exhausted cost nothing
	const int edge_waves = p.L.ntiles < EDGE_WAVES ? p.L.ntiles : EDGE_WAVES;
	dim3 edge_grid(sc->n_views, edge_waves + (fast ? fill_share_blocks(fill_share(p.fill_mode, 0, p.L.nwords)) : 0));
	if (!fast || owner_tiles || (p.sigma > 0 && !p.fuse_edges)) // (a fit step with fused edge tiles launches nothing here)
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
		p.setup_sparse = small_launch(p.T, sc->n_views) ? 4 : 1; // (finalize_kernel: one edge slot per thread then)
		dim3 g2((unsigned)(prim_tri_blocks(p.T) + prim_edge_blocks(p.T, p.setup_sparse > 1 ? 1 : EDGE_SLOTS)) * (unsigned)sc->n_views +
				(unsigned)((fill_words + PRIM_BLOCK / 64 - 1) / (PRIM_BLOCK / 64)) +
				(p.loss_out ? 1u : 0u)); // (+ the workgroup that adds up the loss)
		ScopedKernelTimer t(KID_FINALIZE, st);
		if (p.det && p.vtx_f64)
			launch_finalize(finalize_kernel<true, 0, true>, g2, st, p);
		else if (p.det)
			launch_finalize(finalize_kernel<false, 0, true>, g2, st, p);
		else if (p.prim_tables && p.C <= CH)
		{ // (the instances with the per-workgroup vertex table, for the channel counts that occur)
			if (p.vtx_f64 && p.C == 4)
				launch_finalize(finalize_kernel<true, 4, false, true>, g2, st, p);
			else if (p.vtx_f64 && p.C == 3)
				launch_finalize(finalize_kernel<true, 3, false, true>, g2, st, p);
			else if (p.vtx_f64)
				launch_finalize(finalize_kernel<true, 0, false, true>, g2, st, p);
			else if (p.C == 4)
				launch_finalize(finalize_kernel<false, 4, false, true>, g2, st, p);
			else if (p.C == 3)
				launch_finalize(finalize_kernel<false, 3, false, true>, g2, st, p);
			else
				launch_finalize(finalize_kernel<false, 0, false, true>, g2, st, p);
		}
		else
			DR_LAUNCH_PRIM(finalize_kernel, g2, st);
	}
	if (p.det)
		det_convert(p, sc->n_views, st);
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

// Streaming copy / fill / read with 16-byte non-temporal accesses, the access pattern of the rasterizer's own frame stores and
// background fill (deodr_hip_copy_probe: the copy ceiling of the box a roofline fraction may also be quoted against).
typedef uint32_t probe_u4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void copy_probe_kernel(probe_u4 *dst, const probe_u4 *src, size_t n16)
{
	const size_t stride = (size_t)gridDim.x * 256 * 4;
	probe_u4 acc = {0, 0, 0, 0};
	for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride)
	{ // four independent 16-byte pieces per thread and round, consecutive lanes on consecutive pieces
		probe_u4 v[4];
#pragma unroll
		for (int u = 0; u < 4; u++)
			if (MODE != 1 && i + u * 256 < n16)
				v[u] = __builtin_nontemporal_load(src + i + u * 256);
			else
				v[u] = probe_u4{1, 2, 3, 4};
#pragma unroll
		for (int u = 0; u < 4; u++)
			if (i + u * 256 < n16)
			{
				if (MODE == 2)
					acc ^= v[u];
				else
					__builtin_nontemporal_store(v[u], dst + i + u * 256);
			}
	}
	if (MODE == 2 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u)
		((uint32_t *)dst)[0] = acc.x; // (never true for real data: keeps the loads alive)
}

} // namespace

extern "C" {

int deodr_hip_abi_version(void) { return DEODR_HIP_ABI_VERSION; }

int deodr_hip_force_generic(int on)
{
	g_force_generic = on != 0;
	return 0;
}

int deodr_hip_set_deterministic(int on)
{
	g_det = on != 0;
	return 0;
}

int deodr_hip_profile_enable(int every)
{
	g_profile_every = every > 0 ? every : 0;
	g_profile_calls = 0;
	g_profile = false;
	return 0;
}

int deodr_hip_profile_stamps(void *device_buffer, int rows)
{
	g_stamps = (unsigned long long *)device_buffer;
	g_stamp_rows = device_buffer && rows > 0 ? rows : 0;
	g_stamp_calls = 0;
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

int deodr_hip_copy_probe(void *dst, const void *src, size_t bytes, int mode, int reps, void *stream)
{
	if (!dst || (mode != 1 && !src) || bytes < 16 || (bytes & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15) || mode < 0 || mode > 2 || reps <= 0)
		return fail("copy_probe: bad arguments");
	const size_t n16 = bytes / 16;
	size_t blocks = (n16 + 1023) / 1024;
	if (blocks > 256 * 32)
		blocks = 256 * 32; // 32 workgroups per CU, grid-stride beyond
	for (int r = 0; r < reps; r++)
	{
		if (mode == 0)
			hipLaunchKernelGGL(copy_probe_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (probe_u4 *)dst, (const probe_u4 *)src, n16);
		else if (mode == 1)
			hipLaunchKernelGGL(copy_probe_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (probe_u4 *)dst, (const probe_u4 *)src, n16);
		else
			hipLaunchKernelGGL(copy_probe_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (probe_u4 *)dst, (const probe_u4 *)src, n16);
	}
	return check_hip(hipGetLastError(), "copy_probe launch");
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

static int l2_loss_impl(const void *image, const void *obs, int pixel_dtype, size_t count, double *out, void *scratch, size_t scratch_bytes, void *stream,
					   int clamp, double clamp_lo, double clamp_hi);

static int render_scene_fit_impl(const DeodrHipScene *sc, void *image, void *z_buffer, double sigma, const void *obs, int clear_gradients,
								 const DeodrHipFitOptions *opt, void *workspace, size_t workspace_bytes, void *stream)
{
	const double *tile_loss = opt ? opt->tile_loss : nullptr;
	double *loss_out = opt ? opt->loss : nullptr, *loss_scratch = opt ? (double *)opt->loss_scratch : nullptr;
	if (opt && opt->clamp && !(opt->clamp_lo <= opt->clamp_hi))
		return fail("render_scene_fit: clamp_lo > clamp_hi");
	if (loss_out && (!tile_loss || !loss_scratch))
		return fail("render_scene_fit: the loss needs the background table (deodr_hip_background_loss) and its scratch");
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
	const bool fused = p.C <= CH && !g_force_generic && !det_mode(sc);
	const bool loss_in_kernels = loss_out && fused && p.T > 0; // (the tile walkers of the staged forward + finalize's last workgroup)
	if (loss_in_kernels)
		p.loss_tile_bg = tile_loss, p.loss_wave = loss_scratch, p.loss_out = loss_out;
	if (opt && opt->clamp)
		p.clamp = 1, p.clamp_lo = opt->clamp_lo, p.clamp_hi = opt->clamp_hi;
	// the background of the empty tiles rides on the adjoint's kernels (fill_share); without any of them: the side stream
#ifndef DR_FILL_MASK
#define DR_FILL_MASK 7 // measurement builds: 0 side stream, 1 edge kernel only, 2 finalize only, 4 forward raster only
#endif
	// The forward raster also back-propagates the tiles with silhouette edges (no edge-tile kernel, no saved sweeps) and streams a share
	// of the background.  Textured scenes too since round 5 (round 2 measured 0.264 -> 0.407 ms for one 2048^2 view of 100 k triangles:
	// 400 spilled registers on the edge path at four waves per SIMD; with the instances of their own at three waves, the many-edge tiles
	// split into parts and the quadrant windows of the texture gradient it is 0.195 -> 0.168 ms, 2 / 4 / 8 views 0.257 -> 0.243 /
	// 0.448 -> 0.451 / 0.890 -> 0.893: profiles/r05y_ab_fused_textured_edge_tiles.txt).  (sigma = 0: no edge anywhere, the lighter instances)
	p.fuse_edges = DR_FUSE_EDGES && fused && (!p.texture || (DR_FUSE_TEX_EDGES && sigma > 0));
	p.fill_mode = fused ? ((((sigma > 0 && !p.fuse_edges) ? 1 : 0) | (p.T > 0 ? 2 : 0) | ((p.T > 0 && p.fuse_edges) ? 4 : 0)) & DR_FILL_MASK) : 0;
	note_forward(workspace, fused);
	hipEvent_t join = nullptr;
	if (launch_forward(sc, p, st, &join, fused))
		return 1;
	// The step-done flag: stored by the last wavefront of finalize_kernel to finish when that kernel is the step's last (the usual fit
	// step), by a one-thread kernel behind everything otherwise.
	uint32_t *done_flag = opt ? opt->done_flag : nullptr;
	const bool fin_signals = done_flag && p.T > 0 && !det_mode(sc) && !join && !(loss_out && !loss_in_kernels);
	if (fin_signals)
		p.done_flag = done_flag, p.done_value = opt->done_value;
	if (launch_adjoint(sc, p, st, !fused))
		return 1;
	if (join_side(st, join)) // the background fill has been overlapping the adjoint
		return 1;
	if (loss_out && !loss_in_kernels)
	{ // un-staged kernels (more than 4 channels) or a scene without triangles: one pass over the finished frame
		if (check_hip(hipMemsetAsync(loss_scratch, 0, 64 + 8 * (size_t)L2_BLOCKS, st), "loss scratch"))
			return 1;
		if (l2_loss_impl(image, obs, sc->pixel_dtype, (size_t)sc->n_views * sc->height * sc->width * sc->nb_colors, loss_out, loss_scratch,
						 64 + 8 * (size_t)L2_BLOCKS, stream, p.clamp, p.clamp_lo, p.clamp_hi))
			return 1;
	}
	if (done_flag && !fin_signals)
	{
		hipLaunchKernelGGL(store_flag_kernel, dim3(1), dim3(1), 0, st, done_flag, opt->done_value);
		return check_hip(hipGetLastError(), "done flag launch");
	}
	return 0;
}

int deodr_hip_render_scene_fit(const DeodrHipScene *sc, void *image, void *z_buffer, double sigma, const void *obs, int clear_gradients,
							   void *workspace, size_t workspace_bytes, void *stream)
{
	return render_scene_fit_impl(sc, image, z_buffer, sigma, obs, clear_gradients, nullptr, workspace, workspace_bytes, stream);
}

int deodr_hip_render_scene_fit_ex(const DeodrHipScene *sc, void *image, void *z_buffer, double sigma, const void *obs, int clear_gradients,
								  const DeodrHipFitOptions *options, void *workspace, size_t workspace_bytes, void *stream)
{
	return render_scene_fit_impl(sc, image, z_buffer, sigma, obs, clear_gradients, options, workspace, workspace_bytes, stream);
}

static size_t loss_table_doubles(int height, int width, int n_views)
{ // [0] the whole frame, then one value per view and tile; never less than the partials of the one-pass fallback
	const size_t tiles = (size_t)((width + TILE - 1) / TILE) * ((height + TILE - 1) / TILE);
	size_t n = 1 + (size_t)n_views * tiles;
	const size_t floors[2] = {(size_t)L2_BLOCKS + 16, (size_t)n_views * LOSS_SLOTS};
	for (size_t f : floors)
		n = n > f ? n : f;
	return n;
}

size_t deodr_hip_fit_loss_bytes(int height, int width, int n_views)
{
	return height > 0 && width > 0 && n_views > 0 ? sizeof(double) * loss_table_doubles(height, width, n_views) : 0;
}

int deodr_hip_background_loss(const DeodrHipScene *sc, const void *obs, const DeodrHipFitOptions *options, double *tile_loss, void *workspace,
							  size_t workspace_bytes, void *stream)
{
	KParams p;
	if (fill_params(sc, 1.0, workspace, workspace_bytes, p, false))
		return 1;
	if (!obs || !tile_loss)
		return fail("background_loss needs obs and the table");
	if (options && options->clamp)
		p.clamp = 1, p.clamp_lo = options->clamp_lo, p.clamp_hi = options->clamp_hi;
	p.obs = obs;
	p.n_views = sc->n_views;
	hipStream_t st = (hipStream_t)stream;
	const dim3 grid((unsigned)p.L.ntiles, (unsigned)sc->n_views);
	if (sc->pixel_dtype == DEODR_HIP_F64)
		hipLaunchKernelGGL(background_loss_kernel<double>, grid, dim3(64), 0, st, p, tile_loss);
	else
		hipLaunchKernelGGL(background_loss_kernel<float>, grid, dim3(64), 0, st, p, tile_loss);
	hipLaunchKernelGGL(background_loss_total_kernel, dim3(1), dim3(FH_BLOCK), 0, st, tile_loss, (size_t)sc->n_views * p.L.ntiles);
	return check_hip(hipGetLastError(), "background_loss launch");
}

// ---- front half of a fit iteration (dr_fronthalf.h): plain double arrays on the device, asynchronous on `stream`

int deodr_hip_rigid_transform(const double *vertices, const double *quaternions, const double *translations, double *out, int V, int n, void *stream)
{
	if (!vertices || !quaternions || !translations || !out || V <= 0 || n <= 0)
		return fail("rigid_transform: bad arguments");
	hipLaunchKernelGGL(rigid_transform_kernel, dim3((V + FH_BLOCK - 1) / FH_BLOCK, n), dim3(FH_BLOCK), 0, (hipStream_t)stream, vertices, quaternions, translations, out, V, n);
	return check_hip(hipGetLastError(), "rigid_transform launch");
}

int deodr_hip_rigid_transform_b(const double *vertices, const double *quaternions, const double *out_b, double *vertices_b, double *pose_b, int V, int n,
								void *stream)
{ // pose_b [n,7] = (quaternion adjoint 4, translation adjoint 3) per view, overwritten
	if (!vertices || !quaternions || !out_b || !vertices_b || !pose_b || V <= 0 || n <= 0)
		return fail("rigid_transform_b: bad arguments");
	hipStream_t st = (hipStream_t)stream;
	if (check_hip(hipMemsetAsync(pose_b, 0, sizeof(double) * 7 * (size_t)n, st), "rigid_transform_b clear"))
		return 1;
	// (the kernel addresses the two adjoints separately: quaternions first [n,4], then translations [n,3])
	hipLaunchKernelGGL(rigid_transform_b_kernel, dim3((V + FH_BLOCK - 1) / FH_BLOCK), dim3(FH_BLOCK), 0, st, vertices, quaternions, out_b, vertices_b, pose_b,
					   pose_b + 4 * (size_t)n, V, n);
	return check_hip(hipGetLastError(), "rigid_transform_b launch");
}

int deodr_hip_project_points(const double *points, const double *extrinsic, const double *intrinsic, const double *distortion, double *ij, double *depths,
							 int V, int n, void *stream)
{
	if (!points || !extrinsic || !intrinsic || !ij || !depths || V <= 0 || n <= 0)
		return fail("project_points: bad arguments");
	hipLaunchKernelGGL(project_points_kernel, dim3((V + FH_BLOCK - 1) / FH_BLOCK, n), dim3(FH_BLOCK), 0, (hipStream_t)stream, points, extrinsic, intrinsic,
					   distortion, ij, depths, V, n);
	return check_hip(hipGetLastError(), "project_points launch");
}

int deodr_hip_project_points_b(const double *points, const double *extrinsic, const double *intrinsic, const double *distortion, const double *ij_b,
							   const double *depths_b, double *points_b, int V, int n, void *stream)
{
	if (!points || !extrinsic || !intrinsic || !ij_b || !points_b || V <= 0 || n <= 0)
		return fail("project_points_b: bad arguments");
	hipLaunchKernelGGL(project_points_b_kernel, dim3((V + FH_BLOCK - 1) / FH_BLOCK, n), dim3(FH_BLOCK), 0, (hipStream_t)stream, points, extrinsic, intrinsic,
					   distortion, ij_b, depths_b, points_b, V, n);
	return check_hip(hipGetLastError(), "project_points_b launch");
}

int deodr_hip_silhouette_flags(const double *ij, const uint32_t *faces, const uint32_t *edge_faces, uint8_t *flags, int T, int V, int n, int clockwise,
							   void *stream)
{
	if (!ij || !faces || !edge_faces || !flags || T <= 0 || V <= 0 || n <= 0)
		return fail("silhouette_flags: bad arguments");
	hipLaunchKernelGGL(silhouette_flags_kernel, dim3((T + FH_BLOCK - 1) / FH_BLOCK, n), dim3(FH_BLOCK), 0, (hipStream_t)stream, ij, faces, edge_faces, flags, T, V,
					   clockwise);
	return check_hip(hipGetLastError(), "silhouette_flags launch");
}

// scratch of the fit-iteration kernels: 16 counter words (zero between launches: allocate zero-filled once), then doubles
static size_t fh_blocks(long long count) { return (size_t)((count + FH_BLOCK - 1) / FH_BLOCK); }
static size_t fit_scratch_need_pose_b(int, int n) { return 64 + 8 * (size_t)POSE_B_BLOCKS * (size_t)(7 * n + 3); }
static size_t fit_scratch_need_shade_b(int V, int n) { return 64 + 8 * (3 * (size_t)n * V + 7 * fh_blocks((long long)n * V * GATHER_LANES)); }
static size_t fit_scratch_need_rigid(int V) { return 64 + 8 * fh_blocks((long long)V * GATHER_LANES); }
static size_t fit_scratch_need_l2(void) { return 64 + 8 * (size_t)L2_BLOCKS; }
static size_t fit_scratch_need_momentum(int most) { return 64 + 8 * (size_t)MOMENTUM_MAX * 3 * fh_blocks(most); }
enum FitCounter
{
	FC_POSE_B = 0,
	FC_SHADE_B = 1,
	FC_RIGID = 2,
	FC_L2 = 3,
	FC_MOMENTUM = 4 // ... + MOMENTUM_MAX
};

size_t deodr_hip_fit_scratch_bytes(int V, int n)
{
	if (V <= 0 || n <= 0)
		return 0;
	size_t need = fit_scratch_need_pose_b(V, n);
	const size_t others[4] = {fit_scratch_need_shade_b(V, n), fit_scratch_need_rigid(V), fit_scratch_need_momentum(3 * V > 7 * n ? 3 * V : 7 * n),
							  fit_scratch_need_l2()};
	for (size_t o : others)
		need = o > need ? o : need;
	return need;
}

int deodr_hip_momentum_update(int n_tensors, double *const *x, double *const *speed, const double *const *grad, const double *const *grad2,
							  const double *factor, const double *step_max, const int *count, const int *normalize_rows, double inertia, double damping,
							  const double *grad_scale, const double *const *grad_mean, double *const *mean_out, double *energy, const double *data_energy,
							  double data_weight, void *scratch, size_t scratch_bytes, void *stream)
{
	if (energy && !data_energy)
		return fail("momentum_update: energy[1] = data_weight * data_energy[0] + energy[0] needs data_energy");
	if (n_tensors <= 0 || n_tensors > MOMENTUM_MAX || !x || !speed || !grad || !factor || !step_max || !count)
		return fail("momentum_update: bad arguments (at most 8 tensors per call)");
	MomentumArgs a;
	memset(&a, 0, sizeof a);
	int most = 0;
	bool means = false;
	for (int k = 0; k < n_tensors; k++)
	{
		if (!x[k] || !speed[k] || !grad[k] || count[k] <= 0)
			return fail("momentum_update: NULL tensor");
		a.x[k] = x[k], a.speed[k] = speed[k], a.grad[k] = grad[k], a.grad2[k] = grad2 ? grad2[k] : nullptr;
		a.factor[k] = factor[k], a.step_max[k] = step_max[k], a.count[k] = count[k], a.normalize_rows[k] = normalize_rows ? normalize_rows[k] : 0;
		a.grad_scale[k] = grad_scale ? grad_scale[k] : 1.0;
		a.grad_mean[k] = grad_mean ? grad_mean[k] : nullptr;
		a.mean_out[k] = mean_out ? mean_out[k] : nullptr;
		if ((a.grad_mean[k] || a.mean_out[k]) && (count[k] % 3 || a.normalize_rows[k]))
			return fail("momentum_update: grad_mean / mean_out are for [count/3, 3] tensors without row normalisation");
		means = means || a.mean_out[k];
		most = count[k] > most ? count[k] : most;
	}
	if (means && (!scratch || scratch_bytes < fit_scratch_need_momentum(most)))
		return fail("momentum_update: mean_out needs the fit scratch (deodr_hip_fit_scratch_bytes)");
	a.n = n_tensors, a.inertia = inertia, a.damping = damping;
	a.energy = energy, a.data_energy = data_energy, a.data_weight = data_weight;
	a.counters = scratch ? (unsigned *)scratch + FC_MOMENTUM : nullptr;
	a.partials = scratch ? (double *)((char *)scratch + 64) : nullptr;
	hipLaunchKernelGGL(momentum_update_kernel, dim3((most + FH_BLOCK - 1) / FH_BLOCK, n_tensors), dim3(FH_BLOCK), 0, (hipStream_t)stream, a);
	return check_hip(hipGetLastError(), "momentum_update launch");
}

int deodr_hip_fit_pose_project(double *vertices, const double *vertices_mean, const double *quaternions, const double *translations, const double *extrinsic,
							   const double *intrinsic, const double *distortion, double *posed, double *ij, double *depths, double *depth_colors,
							   double depth_scale, int V, int n, void *stream)
{
	if (!vertices || !quaternions || !translations || !extrinsic || !intrinsic || !posed || !ij || !depths || V <= 0 || n <= 0)
		return fail("fit_pose_project: bad arguments");
	if (n == 1)
		hipLaunchKernelGGL(fit_pose_project_kernel<1>, dim3(fh_blocks(V)), dim3(FH_BLOCK), 0, (hipStream_t)stream, vertices, vertices_mean, quaternions, translations,
					   extrinsic, intrinsic, distortion, posed, ij, depths, depth_colors, depth_scale, V, n);
	else
		hipLaunchKernelGGL(fit_pose_project_kernel<GATHER_LANES>, dim3(fh_blocks((long long)V * GATHER_LANES)), dim3(FH_BLOCK), 0, (hipStream_t)stream, vertices, vertices_mean, quaternions, translations,
					   extrinsic, intrinsic, distortion, posed, ij, depths, depth_colors, depth_scale, V, n);
	return check_hip(hipGetLastError(), "fit_pose_project launch");
}

int deodr_hip_fit_pose_project_b(const double *vertices, const double *quaternions, const double *posed, const double *extrinsic, const double *intrinsic,
								 const double *distortion, const double *posed_b, const double *ij_b, const double *depths_b, double depths_b_scale,
								 double *vertices_b, double *out, void *scratch, size_t scratch_bytes, int V, int n, const double *colors_b, int nb_colors,
								 double *colors_sum, void *stream)
{
	if (!vertices || !quaternions || !posed || !extrinsic || !intrinsic || !ij_b || !vertices_b || !out || V <= 0 || n <= 0)
		return fail("fit_pose_project_b: bad arguments");
	if (n > FIT_MAX_VIEWS)
		return fail("fit_pose_project_b: at most 64 views per call");
	if (colors_sum && (!colors_b || nb_colors <= 0 || nb_colors > 4))
		return fail("fit_pose_project_b: colors_sum needs colors_b with 1 - 4 channels");
	if (!scratch || scratch_bytes < fit_scratch_need_pose_b(V, n))
		return fail("fit_pose_project_b: scratch too small (deodr_hip_fit_scratch_bytes)");
	const size_t pose_blocks = fh_blocks(n == 1 ? (long long)V : (long long)V * GATHER_LANES);
	const dim3 pose_grid((unsigned)(pose_blocks < (size_t)POSE_B_BLOCKS ? pose_blocks : (size_t)POSE_B_BLOCKS));
	if (n == 1)
		hipLaunchKernelGGL(fit_pose_project_b_kernel<1>, pose_grid, dim3(FH_BLOCK), 0, (hipStream_t)stream, vertices, quaternions, posed, extrinsic, intrinsic,
					   distortion, posed_b, ij_b, depths_b, depths_b_scale, vertices_b, out, (double *)((char *)scratch + 64), (unsigned *)scratch + FC_POSE_B, V, n, colors_b, nb_colors, colors_sum);
	else
		hipLaunchKernelGGL(fit_pose_project_b_kernel<GATHER_LANES>, pose_grid, dim3(FH_BLOCK), 0, (hipStream_t)stream, vertices, quaternions, posed, extrinsic, intrinsic,
					   distortion, posed_b, ij_b, depths_b, depths_b_scale, vertices_b, out, (double *)((char *)scratch + 64), (unsigned *)scratch + FC_POSE_B, V, n, colors_b, nb_colors, colors_sum);
	return check_hip(hipGetLastError(), "fit_pose_project_b launch");
}

int deodr_hip_views_gradient_sum(const double *posed, const double *extrinsic, const double *intrinsic, const double *distortion, const double *ij_b,
								 const double *depths_b, double depths_b_scale, double *vertices_b, int V, int n, const double *colors_b, int nb_colors,
								 double *colors_sum, void *stream)
{
	if (!posed || !extrinsic || !intrinsic || !ij_b || !vertices_b || V <= 0 || n <= 0)
		return fail("views_gradient_sum: bad arguments");
	if (colors_sum && (!colors_b || nb_colors <= 0 || nb_colors > 4))
		return fail("views_gradient_sum: colors_sum needs colors_b with 1 - 4 channels");
	if (n == 1)
		hipLaunchKernelGGL(views_gradient_sum_kernel<1>, dim3(fh_blocks(V)), dim3(FH_BLOCK), 0, (hipStream_t)stream, posed, extrinsic, intrinsic, distortion, ij_b,
						   depths_b, depths_b_scale, vertices_b, V, n, colors_b, nb_colors, colors_sum);
	else
		hipLaunchKernelGGL(views_gradient_sum_kernel<GATHER_LANES>, dim3(fh_blocks((long long)V * GATHER_LANES)), dim3(FH_BLOCK), 0, (hipStream_t)stream, posed,
						   extrinsic, intrinsic, distortion, ij_b, depths_b, depths_b_scale, vertices_b, V, n, colors_b, nb_colors, colors_sum);
	return check_hip(hipGetLastError(), "views_gradient_sum launch");
}

int deodr_hip_wait_flag(const uint32_t *flag, uint32_t value, uint32_t *status, double timeout_seconds, void *stream)
{
	if (!flag || !(timeout_seconds > 0))
		return fail("wait_flag: bad arguments");
	hipLaunchKernelGGL(wait_flag_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flag, value, status, (unsigned long long)(timeout_seconds * 1e8));
	return check_hip(hipGetLastError(), "wait_flag launch");
}

static int shade_args(ShadeArgs &a, const double *posed, const uint32_t *faces, const uint32_t *vf_offsets, const uint32_t *vf_corners, const double *light,
					  const double *ambient, const double *color, int C, int V, int n, int clockwise)
{
	if (!posed || !faces || !vf_offsets || !vf_corners || !light || !ambient || V <= 0 || n <= 0 || (color && (C < 1 || C > 3)))
		return fail("vertex_shade: bad arguments (one colour of 1 - 3 channels, or none)");
	a = ShadeArgs{posed, faces, vf_offsets, vf_corners, light, ambient, color, color ? C : 0, V, n, clockwise ? -1.0 : 1.0};
	return 0;
}

int deodr_hip_vertex_shade(const double *posed, const uint32_t *faces, const uint32_t *vf_offsets, const uint32_t *vf_corners, const double *light,
						   const double *ambient, const double *color, int C, double *luminosity, double *colors, int V, int n, int clockwise, void *stream)
{
	ShadeArgs a;
	if (shade_args(a, posed, faces, vf_offsets, vf_corners, light, ambient, color, C, V, n, clockwise))
		return 1;
	if ((!luminosity && !colors) || (colors && !color))
		return fail("vertex_shade: nothing to write, or colours asked for without a colour");
	hipLaunchKernelGGL(vertex_shade_kernel, dim3(fh_blocks((long long)V * GATHER_LANES), n), dim3(FH_BLOCK), 0, (hipStream_t)stream, a, luminosity, colors);
	return check_hip(hipGetLastError(), "vertex_shade launch");
}

int deodr_hip_vertex_shade_b(const double *posed, const uint32_t *faces, const uint32_t *vf_offsets, const uint32_t *vf_corners, const double *light,
							 const double *ambient, const double *color, int C, const double *luminosity_b, const double *colors_b, double *posed_b, double *out,
							 void *scratch, size_t scratch_bytes, int V, int n, int clockwise, void *stream)
{
	ShadeArgs a;
	if (shade_args(a, posed, faces, vf_offsets, vf_corners, light, ambient, color, C, V, n, clockwise))
		return 1;
	if ((!luminosity_b && !colors_b) || (colors_b && !color) || !posed_b || !out)
		return fail("vertex_shade_b: bad arguments");
	if (!scratch || scratch_bytes < fit_scratch_need_shade_b(V, n))
		return fail("vertex_shade_b: scratch too small (deodr_hip_fit_scratch_bytes)");
	hipStream_t st = (hipStream_t)stream;
	double *acc_b = (double *)((char *)scratch + 64), *partials = acc_b + 3 * (size_t)n * V;
	hipLaunchKernelGGL(vertex_shade_b1_kernel, dim3(fh_blocks((long long)n * V * GATHER_LANES)), dim3(FH_BLOCK), 0, st, a, luminosity_b, colors_b, acc_b, out, partials,
					   (unsigned *)scratch + FC_SHADE_B);
	hipLaunchKernelGGL(vertex_shade_b2_kernel, dim3(fh_blocks((long long)V * GATHER_LANES), n), dim3(FH_BLOCK), 0, st, a, (const double *)acc_b, posed_b);
	return check_hip(hipGetLastError(), "vertex_shade_b launch");
}

int deodr_hip_rigid_energy(const double *vertices, const double *vertices_ref, const uint32_t *m_offsets, const uint32_t *m_cols, const double *m_vals,
						   double cregu, double *gradient, double *energy, const double *data_energy, double data_weight, void *scratch, size_t scratch_bytes,
						   int V, void *stream)
{
	if (!vertices || !vertices_ref || !m_offsets || !m_cols || !m_vals || !gradient || !energy || V <= 0)
		return fail("rigid_energy: bad arguments");
	if (!scratch || scratch_bytes < fit_scratch_need_rigid(V))
		return fail("rigid_energy: scratch too small (deodr_hip_fit_scratch_bytes)");
	const RigidArgs a = {vertices, vertices_ref, m_offsets, m_cols, m_vals, cregu, gradient, energy, data_energy, data_weight, (double *)((char *)scratch + 64),
						 (unsigned *)scratch + FC_RIGID, V};
	hipLaunchKernelGGL(rigid_energy_kernel, dim3(fh_blocks((long long)V * GATHER_LANES)), dim3(FH_BLOCK), 0, (hipStream_t)stream, a);
	return check_hip(hipGetLastError(), "rigid_energy launch");
}

int deodr_hip_fit_front(const double *ij, const uint32_t *faces, const uint32_t *edge_faces, uint8_t *flags, int T, const double *posed,
						const uint32_t *vf_offsets, const uint32_t *vf_corners, const double *light, const double *ambient, const double *color, int C,
						double *luminosity, double *colors, const double *vertices, const double *vertices_ref, const uint32_t *m_offsets, const uint32_t *m_cols,
						const double *m_vals, double cregu, double *gradient, double *energy, void *scratch, size_t scratch_bytes, int V, int n, int clockwise,
						void *stream)
{
	if (!faces || V <= 0 || n <= 0)
		return fail("fit_front: bad arguments");
	FrontArgs a;
	memset(&a, 0, sizeof a);
	a.V = V, a.clockwise = clockwise, a.faces = faces;
	if (flags)
	{
		if (!ij || !edge_faces || T <= 0)
			return fail("fit_front: silhouette flags need ij, edge_faces and the number of faces");
		a.ij = ij, a.edge_faces = edge_faces, a.flags = flags, a.T = T, a.sil_x = (unsigned)fh_blocks(T);
	}
	a.shade.n = n;
	if (luminosity || colors)
	{
		if (shade_args(a.shade, posed, faces, vf_offsets, vf_corners, light, ambient, color, C, V, n, clockwise))
			return 1;
		if (colors && !color)
			return fail("fit_front: colours asked for without a colour");
		a.lum_out = luminosity, a.colors_out = colors, a.shade_x = (unsigned)fh_blocks((long long)V * GATHER_LANES);
	}
	if (gradient)
	{
		if (!vertices || !vertices_ref || !m_offsets || !m_cols || !m_vals || !energy)
			return fail("fit_front: the rigid energy needs vertices, reference, the CSR of L^T L and energy[2]");
		if (!scratch || scratch_bytes < fit_scratch_need_rigid(V))
			return fail("fit_front: scratch too small (deodr_hip_fit_scratch_bytes)");
		a.rigid = RigidArgs{vertices, vertices_ref, m_offsets, m_cols, m_vals, cregu, gradient, energy, nullptr, 0.0, (double *)((char *)scratch + 64),
							(unsigned *)scratch + FC_RIGID, V};
		a.rigid_blocks = (unsigned)fh_blocks((long long)V * GATHER_LANES);
	}
	const unsigned blocks = a.rigid_blocks + (a.shade_x + a.sil_x) * (unsigned)n;
	if (!blocks)
		return fail("fit_front: nothing to do");
	hipLaunchKernelGGL(fit_front_kernel, dim3(blocks), dim3(FH_BLOCK), 0, (hipStream_t)stream, a);
	return check_hip(hipGetLastError(), "fit_front launch");
}

static int l2_loss_impl(const void *image, const void *obs, int pixel_dtype, size_t count, double *out, void *scratch, size_t scratch_bytes, void *stream,
					   int clamp, double clamp_lo, double clamp_hi)
{
	if (!image || !obs || !out || count == 0 || (pixel_dtype != DEODR_HIP_F32 && pixel_dtype != DEODR_HIP_F64))
		return fail("l2_loss: bad arguments");
	if (!scratch || scratch_bytes < fit_scratch_need_l2())
		return fail("l2_loss: scratch too small (deodr_hip_fit_scratch_bytes)");
	if (((uintptr_t)image | (uintptr_t)obs) & 31)
		return fail("l2_loss: image and obs must be 32-byte aligned");
	const size_t want = (count + FH_BLOCK * 32 - 1) / (FH_BLOCK * 32);
	const dim3 grid((unsigned)(want < (size_t)L2_BLOCKS ? want : (size_t)L2_BLOCKS));
	double *partials = (double *)((char *)scratch + 64);
	unsigned *counter = (unsigned *)scratch + FC_L2;
	if (pixel_dtype == DEODR_HIP_F64)
		hipLaunchKernelGGL(l2_loss_kernel<double>, grid, dim3(FH_BLOCK), 0, (hipStream_t)stream, (const double *)image, (const double *)obs, count, out, partials,
						   counter, clamp, clamp_lo, clamp_hi);
	else
		hipLaunchKernelGGL(l2_loss_kernel<float>, grid, dim3(FH_BLOCK), 0, (hipStream_t)stream, (const float *)image, (const float *)obs, count, out, partials,
						   counter, clamp, clamp_lo, clamp_hi);
	return check_hip(hipGetLastError(), "l2_loss launch");
}

int deodr_hip_l2_loss(const void *image, const void *obs, int pixel_dtype, size_t count, double *out, void *scratch, size_t scratch_bytes, void *stream)
{
	return l2_loss_impl(image, obs, pixel_dtype, count, out, scratch, scratch_bytes, stream, 0, 0.0, 0.0);
}

int deodr_hip_depth_residual(const void *image, int pixel_dtype, const double *obs, double max_depth, size_t count, double *depth, double *diff, void *image_b,
							 double *loss, void *scratch, size_t scratch_bytes, void *stream)
{
	if (!image || !obs || !depth || !diff || !image_b || !loss || count == 0 || (pixel_dtype != DEODR_HIP_F32 && pixel_dtype != DEODR_HIP_F64))
		return fail("depth_residual: bad arguments");
	if (!scratch || scratch_bytes < fit_scratch_need_l2())
		return fail("depth_residual: scratch too small (deodr_hip_fit_scratch_bytes)");
	const size_t want = (count + FH_BLOCK * 4 - 1) / (FH_BLOCK * 4);
	const dim3 grid((unsigned)(want < (size_t)L2_BLOCKS ? want : (size_t)L2_BLOCKS));
	double *partials = (double *)((char *)scratch + 64);
	unsigned *counter = (unsigned *)scratch + FC_L2;
	if (pixel_dtype == DEODR_HIP_F64)
		hipLaunchKernelGGL(depth_residual_kernel<double>, grid, dim3(FH_BLOCK), 0, (hipStream_t)stream, (const double *)image, obs, max_depth, count, depth, diff,
						   (double *)image_b, loss, partials, counter);
	else
		hipLaunchKernelGGL(depth_residual_kernel<float>, grid, dim3(FH_BLOCK), 0, (hipStream_t)stream, (const float *)image, obs, max_depth, count, depth, diff,
						   (float *)image_b, loss, partials, counter);
	return check_hip(hipGetLastError(), "depth_residual launch");
}

#ifdef DR_WAVE_TRACE
int deodr_hip_debug_wave_phase(void *dst, size_t bytes) // tools/wave_trace.py
{
	return check_hip(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wave_phase), bytes < sizeof(g_wave_phase) ? bytes : sizeof(g_wave_phase)), "wave phase");
}
int deodr_hip_debug_wave_hw(void *dst, size_t bytes) // tools/wave_trace.py --slots
{
	return check_hip(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wave_hw), bytes < sizeof(g_wave_hw) ? bytes : sizeof(g_wave_hw)), "wave hw ids");
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
