// deodr_amd/csrc/dr_fititer.h -- part of the single translation unit dr_kernels.hip (device code, gfx950 / wave64).
// One iteration of the reference's fitters (deodr/mesh_fitter.py:108-190, 287-376, 529-632) WITHOUT an autograd graph: the chain
//
//   fit_pose_project_kernel   centre the vertices, pose them with every view's (renormalised) quaternion + translation, project them
//                             with every view's camera                                       -> posed, ij, depths
//   vertex_shade_kernel       vertex normals (normalised sum of the unit normals of the faces around a vertex,
//                             deodr/triangulated_mesh.py:113-151), luminosity max(0, -n.l) + ambient (dr.py:814-822), colour * luminosity
//   ... silhouette flags, the rasterizer's fit step (image, loss gradient w.r.t. ij and colours) ...
//   vertex_shade_b1/b2        adjoint of the shading: b1 per vertex (light / ambient / colour sums, adjoint of the accumulated normal),
//                             b2 gathers the adjoint of the posed vertices over the faces around each vertex
//   fit_pose_project_b        adjoint of projection and pose: vertices_b (summed over the views, fixed order), its column mean (the data
//                             gradient is projected on zero-mean displacements, mesh_fitter.py:140), pose_b of every view
//   rigid_energy_kernel       0.5 c d^T (L^T L) d and its gradient over a CSR of L^T L (deodr/laplacian_rigid_energy.py:15-41)
//   momentum_update_kernel    (dr_fronthalf.h) all parameters, in place
//
// All sums over vertices are DETERMINISTIC: per-workgroup partials in a fixed order, added up by the last workgroup to arrive
// (grid_sum below) -- the torch formulation of the same chain (deodr_amd/scene3d.py, mesh_fitter.py) is what these are tested against.
// The gathers use a vertex -> incident (face, corner) CSR built once per topology (MeshTopology), no atomics.
#pragma once

#include "dr_fronthalf.h"

namespace
{

__device__ __forceinline__ Vec3 load3(const double *p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ Vec3 sub3(const Vec3 &a, const Vec3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ void store3(double *p, const Vec3 &a) { p[0] = a.x, p[1] = a.y, p[2] = a.z; }

// q = (x, y, z, w): p + 2 (w (u x p) + u x (u x p))   (deodr/tools.py:8-22)
__device__ __forceinline__ Vec3 qrot_point(const Vec3 &u, double w, const Vec3 &p)
{
	const Vec3 a = cross3(u, p), bb = cross3(u, a);
	return {p.x + 2 * (w * a.x + bb.x), p.y + 2 * (w * a.y + bb.y), p.z + 2 * (w * a.z + bb.z)};
}

__device__ __forceinline__ void project_point(const CameraRow &c, const Vec3 &p, double &i, double &j, double &depth)
{
	const double cx = c.E[0] * p.x + c.E[1] * p.y + c.E[2] * p.z + c.E[3], cy = c.E[4] * p.x + c.E[5] * p.y + c.E[6] * p.z + c.E[7],
				 cz = c.E[8] * p.x + c.E[9] * p.y + c.E[10] * p.z + c.E[11];
	double x = cx / cz, y = cy / cz;
	if (c.distort)
	{
		const double k1 = c.d[0], k2 = c.d[1], p1 = c.d[2], p2 = c.d[3], k3 = c.d[4];
		const double x2 = x * x, y2 = y * y, r2 = x2 + y2, r4 = r2 * r2;
		const double radial = 1 + k1 * r2 + k2 * r4 + k3 * (r2 * r4);
		const double xd = x * radial + (2 * p1 * x * y + p2 * (r2 + 2 * x2)), yd = y * radial + (p1 * (r2 + 2 * y2) + 2 * p2 * x * y);
		x = xd, y = yd;
	}
	i = c.K[0] * x + c.K[1] * y + c.K[2];
	j = c.K[3] * x + c.K[4] * y + c.K[5];
	depth = cz;
}

__device__ __forceinline__ Vec3 project_point_b(const CameraRow &c, const Vec3 &p, double g0, double g1, double gd)
{ // (the same lines as project_points_b_kernel)
	const double cx = c.E[0] * p.x + c.E[1] * p.y + c.E[2] * p.z + c.E[3], cy = c.E[4] * p.x + c.E[5] * p.y + c.E[6] * p.z + c.E[7],
				 cz = c.E[8] * p.x + c.E[9] * p.y + c.E[10] * p.z + c.E[11];
	const double x = cx / cz, y = cy / cz;
	const double xd_b = c.K[0] * g0 + c.K[3] * g1, yd_b = c.K[1] * g0 + c.K[4] * g1;
	double x_b = xd_b, y_b = yd_b;
	if (c.distort)
	{
		const double k1 = c.d[0], k2 = c.d[1], p1 = c.d[2], p2 = c.d[3], k3 = c.d[4];
		const double r2 = x * x + y * y, r4 = r2 * r2;
		const double radial = 1 + k1 * r2 + k2 * r4 + k3 * (r2 * r4);
		const double radial_b = x * xd_b + y * yd_b;
		x_b = radial * xd_b + 2 * p1 * y * xd_b + 4 * p2 * x * xd_b + 2 * p2 * y * yd_b;
		y_b = radial * yd_b + 2 * p1 * x * xd_b + 4 * p1 * y * yd_b + 2 * p2 * x * yd_b;
		const double r2_b = p2 * xd_b + p1 * yd_b + radial_b * (k1 + 2 * k2 * r2 + 3 * k3 * r4);
		x_b += 2 * x * r2_b;
		y_b += 2 * y * r2_b;
	}
	const double cx_b = x_b / cz, cy_b = y_b / cz, cz_b = gd - (x * x_b + y * y_b) / cz;
	return {c.E[0] * cx_b + c.E[4] * cy_b + c.E[8] * cz_b, c.E[1] * cx_b + c.E[5] * cy_b + c.E[9] * cz_b, c.E[2] * cx_b + c.E[6] * cy_b + c.E[10] * cz_b};
}

constexpr int FIT_MAX_VIEWS = 64; // views (poses) of one fit_pose_project_b call

struct UnitQuaternion
{
	Vec3 u;
	double w, norm;
};
__device__ __forceinline__ UnitQuaternion load_unit_quaternion(const double *q, int b)
{
	const double x = q[4 * b], y = q[4 * b + 1], z = q[4 * b + 2], w = q[4 * b + 3];
	const double norm = sqrt(x * x + y * y + z * z + w * w);
	return {{x / norm, y / norm, z / norm}, w / norm, norm};
}

// The gathers of the shading and rigid-energy kernels give every list (the faces around a vertex, a row of L^T L) to GATHER_LANES adjacent lanes: one list entry is a
// chain of dependent loads (slot -> vertex ids -> positions, ~1 us each on an idle chip), and a list of 6 - 20 entries walked by one
// lane IS the kernel's duration (measured: 15 - 20 us per kernel, one lane per list).  Lane s takes the entries s, s + GATHER_LANES, ...
// in order; the lanes' sums meet in a butterfly -- an order fixed by the list alone.
constexpr int GATHER_LANES = 8;
__device__ __forceinline__ double lanes_sum(double v)
{ // all lanes of the wavefront call it; -> the sum over each group of GATHER_LANES adjacent lanes, in every lane of the group
	v += __shfl_xor(v, 1);
	v += __shfl_xor(v, 2);
	v += __shfl_xor(v, 4);
	return v;
}
__device__ __forceinline__ Vec3 lanes_sum3(const Vec3 &a) { return {lanes_sum(a.x), lanes_sum(a.y), lanes_sum(a.z)}; }

constexpr int POSE_B_BLOCKS = 64;		 // workgroups of fit_pose_project_b_kernel at most

// ---- forward.  The L (8; 1 for a single view) adjacent lanes of a vertex take the views b = sub, sub + L, ... (one thread per vertex walking
// the views one after the other made 8 views cost 8.4 us against 4.7 for one).  vertices [V,3] are centred IN PLACE when `mean` is
// given (the reference re-centres its vertices at the start of every step, mesh_fitter.py:131): the eight lanes of a vertex read it
// in ONE load instruction, lane 0 of them stores the centred value afterwards.  Quaternions are the raw parameters (normalised here).
// grid: V GATHER_LANES / FH_BLOCK
template <int L>
__global__ __launch_bounds__(FH_BLOCK) void fit_pose_project_kernel(double *vertices, const double *mean, const double *q, const double *t,
																	 const double *extrinsic, const double *intrinsic, const double *distortion, double *posed,
																	 double *ij, double *depths, double *depth_colors, double depth_scale, int V, int n)
{
	const int th = blockIdx.x * FH_BLOCK + threadIdx.x, v = th / L, sub = th % L;
	if (v >= V)
		return;
	Vec3 c = load3(vertices + 3 * v);
	if (mean)
	{
		c = sub3(c, load3(mean));
		if (sub == 0)
			store3(vertices + 3 * v, c);
	}
	for (int b = sub; b < n; b += L)
	{
		const UnitQuaternion uq = load_unit_quaternion(q, b);
		const Vec3 p = add3(qrot_point(uq.u, uq.w, c), load3(t + 3 * b));
		const size_t at = (size_t)b * V + v;
		store3(posed + 3 * at, p);
		const CameraRow cam = load_camera(extrinsic, intrinsic, distortion, b);
		project_point(cam, p, ij[2 * at], ij[2 * at + 1], depths[at]);
		if (depth_colors) // a depth image is rendered with the scaled depth of a vertex as its one-channel colour (dr.py:1001-1036)
			depth_colors[at] = depths[at] * depth_scale;
	}
}

// ---- adjoint.  posed_b (may be NULL): what the shading back-propagated to the posed vertices; ij_b, depths_b (may be NULL): the
// rasterizer's.  -> vertices_b [V,3] (sum over the views), out[0..3) = column mean of vertices_b, then pose_b: quaternion adjoints
// [n,4] (w.r.t. the RAW quaternions: through the normalisation) and translation adjoints [n,3].  partials: (7 n + 3) doubles per
// workgroup.
template <int L> // lanes per vertex: 8, or 1 for a single view
__global__ __launch_bounds__(FH_BLOCK) void fit_pose_project_b_kernel(const double *vertices, const double *q, const double *posed, const double *extrinsic,
																	   const double *intrinsic, const double *distortion, const double *posed_b, const double *ij_b,
																	   const double *depths_b, double depths_b_scale, double *vertices_b, double *out,
																	   double *partials, unsigned *counter, int V, int n, const double *colors_b, int C,
																	   double *colors_sum)
{
	__shared__ double s_wave[FH_BLOCK / 64][7 * FIT_MAX_VIEWS + 3];
	__shared__ double s_quat[FIT_MAX_VIEWS][4];
	__shared__ int s_last;
	// The L (8; 1 for a single view: seven idle lanes per vertex cost the one-view fitters 2 us) adjacent lanes of a vertex take the views
	// b = sub, sub + L, ...: one round trip for eight views (one thread per
	// vertex walking the views: 19 us for the 8 views of the hand, 25 - 35 us for those of a 10 000-vertex mesh).  Sums over the views
	// of a vertex: a butterfly over lane bits 0-2; sums over the vertices of a view: a butterfly over lane bits 3-5 (the eight vertices of
	// the wavefront), then the wavefronts through LDS, the workgroups through `partials` -- orders fixed by the launch geometry alone.
	// At most POSE_B_BLOCKS workgroups, each walking the vertices in strides: with one workgroup per 32 vertices the 313 of a
	// 10 000-vertex mesh cost a render loop that overlaps this kernel on a second stream 26 us per step instead of 15 (bench.py's
	// shared-gradient reduction), and the last workgroup's sum over the workgroups grows with their number.
	const int sub = threadIdx.x % L, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int K = 7 * n + 3;
	double *mine = partials + (size_t)blockIdx.x * K;
	for (int k = lane; k < K; k += 64)
		s_wave[wave][k] = 0; // (a wavefront's own line: written and read by its lanes only until the sums below are gathered)
	__syncthreads();
	auto vertices_sum = [](double x) { // over the 64 / L vertices of the wavefront, for this lane's view; in every lane
#pragma unroll
		for (int d = L; d < 64; d <<= 1)
			x += __shfl_xor(x, d);
		return x;
	};
	auto views_sum = [](double x) { // over the L lanes (views) of a vertex; in every lane of the vertex
#pragma unroll
		for (int d = 1; d < L; d <<= 1)
			x += __shfl_xor(x, d);
		return x;
	};
	constexpr int PER_BLOCK = FH_BLOCK / L;
	for (int base = blockIdx.x * PER_BLOCK; base < V; base += gridDim.x * PER_BLOCK) // (the same trips for every thread of the workgroup)
	{
		const int v = base + threadIdx.x / L;
		const bool on = v < V;
		const Vec3 c = on ? load3(vertices + 3 * v) : Vec3{0, 0, 0};
		Vec3 acc = {0, 0, 0};
		double col_sum[4] = {0, 0, 0, 0};
		for (int b0 = 0; b0 < n; b0 += L)
		{
			const int b = b0 + sub;
			const bool act = on && b < n;
			const int bq = b < n ? b : 0;
			Vec3 g = {0, 0, 0};
			if (act)
			{
				const size_t at = (size_t)b * V + v;
				if (colors_sum)
#pragma unroll
					for (int cc = 0; cc < 4; cc++)
						if (cc < C)
							col_sum[cc] += colors_b[at * C + cc];
				const CameraRow cam = load_camera(extrinsic, intrinsic, distortion, b);
				g = project_point_b(cam, load3(posed + 3 * at), ij_b[2 * at], ij_b[2 * at + 1], depths_b ? depths_b[at] * depths_b_scale : 0.0);
				if (posed_b)
					g = add3(g, load3(posed_b + 3 * at));
			}
			const UnitQuaternion uq = load_unit_quaternion(q, bq);
			// r = c + 2 w a + 2 bb, a = u x c, bb = u x a   (deodr/tools.py:25-35)
			const Vec3 &u = uq.u;
			const Vec3 a = cross3(u, c);
			const double w_b = 2 * dot3(g, a);
			const Vec3 bb_b = scale3(2, g);
			const Vec3 a_b = add3(scale3(2 * uq.w, g), cross3(bb_b, u));
			const Vec3 u_b = add3(cross3(a, bb_b), cross3(c, a_b));
			acc = add3(acc, add3(g, cross3(a_b, u))); // (g = 0 for a lane without a view: nothing added)
			const double sums[7] = {u_b.x, u_b.y, u_b.z, w_b, g.x, g.y, g.z};
#pragma unroll
			for (int i = 0; i < 7; i++)
			{
				const double t = vertices_sum(sums[i]); // (every lane takes part: lanes beyond V or n hold zeros)
				if (lane < L && b < n)
					s_wave[wave][7 * b + i] += t;
			}
		}
		acc = {views_sum(acc.x), views_sum(acc.y), views_sum(acc.z)};
		if (on && sub == 0)
			store3(vertices_b + 3 * v, acc);
		if (colors_sum) // per-vertex colours shared by the views (a multi-view fit of a coloured mesh): their adjoints summed over the views
#pragma unroll
			for (int cc = 0; cc < 4; cc++)
				if (cc < C)
				{
					const double t = views_sum(col_sum[cc]);
					if (on && sub == 0)
						colors_sum[(size_t)v * C + cc] = t;
				}
		const double sums[3] = {acc.x, acc.y, acc.z};
#pragma unroll
		for (int i = 0; i < 3; i++)
		{
			const double t = vertices_sum(sums[i]); // (the L lanes of a vertex hold the same sum: lane 0 = the wavefront's vertices)
			if (lane == 0)
				s_wave[wave][7 * n + i] += t;
		}
	}
	__syncthreads();
	for (int k = threadIdx.x; k < K; k += FH_BLOCK)
	{
		double s = 0;
		for (int w = 0; w < FH_BLOCK / 64; w++)
			s += s_wave[w][k];
		mine[k] = s;
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		__threadfence(); // release (one thread, after the barrier: the workgroup's partials are visible to the device before its ticket)
		s_last = atomicAdd(counter, 1u) == gridDim.x - 1;
	}
	__syncthreads();
	if (!s_last)
		return;
	__threadfence(); // acquire
	double *pose_b = out + 3; // [n,4] then [n,3]
	// lane k of every wavefront adds up output k (consecutive lanes read consecutive addresses) over the workgroups w, w + 4, ... of its
	// wavefront w, eight independent partial sums at a time (a lane adding 400 partials one after the other waits 400 times: 142 us for
	// a 100 000-vertex mesh); the four wavefront sums meet in LDS.  The order depends on the launch geometry only.
	__syncthreads(); // (s_wave is reused)
	constexpr int NW = FH_BLOCK / 64;
	for (int k = lane; k < K; k += 64)
	{
		double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		for (unsigned blk = wave; blk < gridDim.x; blk += NW * 8)
#pragma unroll
			for (int j = 0; j < 8; j++)
				if (blk + NW * j < gridDim.x)
					a[j] += partials[(size_t)(blk + NW * j) * K + k];
		s_wave[wave][k] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
	}
	__syncthreads();
	for (int k = threadIdx.x; k < K; k += FH_BLOCK)
	{
		double s = 0;
		for (int w = 0; w < NW; w++)
			s += s_wave[w][k];
		if (k >= 7 * n)
			out[k - 7 * n] = s / V;
		else if (k % 7 >= 4)
			pose_b[4 * n + 3 * (k / 7) + (k % 7 - 4)] = s;
		else
			s_quat[k / 7][k % 7] = s; // (the quaternion needs its four sums together: below)
	}
	if (threadIdx.x == 0)
		atomicExch(counter, 0u);
	__syncthreads();
	for (int b = threadIdx.x; b < n; b += FH_BLOCK)
	{ // q = raw / |raw|:  raw_b = (q_b - q (q . q_b)) / |raw|
		const UnitQuaternion uq = load_unit_quaternion(q, b);
		const double g[4] = {s_quat[b][0], s_quat[b][1], s_quat[b][2], s_quat[b][3]};
		const double qq[4] = {uq.u.x, uq.u.y, uq.u.z, uq.w};
		const double along = g[0] * qq[0] + g[1] * qq[1] + g[2] * qq[2] + g[3] * qq[3];
		for (int i = 0; i < 4; i++)
			pose_b[4 * b + i] = (g[i] - qq[i] * along) / uq.norm;
	}
}

// ---- what the views of a multi-view fit share (deodr/mesh_fitter.py:518-527: `vertices_b += ...` frame after frame): the adjoint of every
// view's camera projection applied to its ij_b (and depths_b), summed over the views -> vertices_b [V,3]; the colour adjoints summed over the
// views -> colors_sum [V,C].  No pose, no sum over the vertices: one round trip, as wide as the mesh (the L lanes of a vertex take its views) --
// this is the buffer a sharded fit all-reduces, and as a tail of fit_pose_project_b_kernel (at most 64 workgroups, built to idle beside
// the raster kernels) it took 30 us of the step it ran in.
template <int L>
__global__ __launch_bounds__(FH_BLOCK) void views_gradient_sum_kernel(const double *posed, const double *extrinsic, const double *intrinsic, const double *distortion,
																	   const double *ij_b, const double *depths_b, double depths_b_scale, double *vertices_b, int V,
																	   int n, const double *colors_b, int C, double *colors_sum)
{
	const int th = blockIdx.x * FH_BLOCK + threadIdx.x, v = th / L, sub = th % L;
	const bool on = v < V;
	Vec3 acc = {0, 0, 0};
	double col_sum[4] = {0, 0, 0, 0};
	for (int b = sub; b < n && on; b += L)
	{
		const size_t at = (size_t)b * V + v;
		if (colors_sum)
#pragma unroll
			for (int cc = 0; cc < 4; cc++)
				if (cc < C)
					col_sum[cc] += colors_b[at * C + cc];
		const CameraRow cam = load_camera(extrinsic, intrinsic, distortion, b);
		acc = add3(acc, project_point_b(cam, load3(posed + 3 * at), ij_b[2 * at], ij_b[2 * at + 1], depths_b ? depths_b[at] * depths_b_scale : 0.0));
	}
	auto views_sum = [](double x) { // over the L lanes of a vertex (every lane of the wavefront takes part)
#pragma unroll
		for (int d = 1; d < L; d <<= 1)
			x += __shfl_xor(x, d);
		return x;
	};
	acc = {views_sum(acc.x), views_sum(acc.y), views_sum(acc.z)};
	if (on && sub == 0)
		store3(vertices_b + 3 * v, acc);
	if (colors_sum)
#pragma unroll
		for (int cc = 0; cc < 4; cc++)
			if (cc < C)
			{
				const double t = views_sum(col_sum[cc]);
				if (on && sub == 0)
					colors_sum[(size_t)v * C + cc] = t;
			}
}

// ---- shading.  vf_offsets [V+1], vf_corners [3T]: for every vertex the (3 face + corner) slots it occupies (static per mesh)
struct FaceNormal
{
	Vec3 e1, e2, unit; // edges from corner 0, n / |n|
	double len;
	uint32_t i0, i1, i2;
};
__device__ __forceinline__ FaceNormal face_normal(const double *P, const uint32_t *faces, uint32_t f)
{
	FaceNormal r;
	r.i0 = faces[3 * f], r.i1 = faces[3 * f + 1], r.i2 = faces[3 * f + 2];
	const Vec3 p0 = load3(P + 3 * (size_t)r.i0);
	r.e1 = sub3(load3(P + 3 * (size_t)r.i1), p0), r.e2 = sub3(load3(P + 3 * (size_t)r.i2), p0);
	const Vec3 nn = cross3(r.e1, r.e2);
	r.len = sqrt(dot3(nn, nn));
	r.unit = {nn.x / r.len, nn.y / r.len, nn.z / r.len};
	return r;
}

struct ShadeArgs
{
	const double *posed;	   // [n,V,3]
	const uint32_t *faces;	   // [T,3]
	const uint32_t *vf_offsets; // [V+1]
	const uint32_t *vf_corners; // [3T]
	const double *light;	   // [3] directional light
	const double *ambient;	   // [1]
	const double *color;	   // [C] one colour for the whole mesh, or NULL (luminosity only)
	int C, V, n;
	double sign; // -1: clockwise faces
};

// sign * sum of the unit normals of the faces around vertex v (`on` false: an empty list); in every lane of the group
__device__ __forceinline__ Vec3 accumulated_normal(const ShadeArgs &a, const double *P, int v, int sub, bool on)
{
	Vec3 acc = {0, 0, 0};
	const uint32_t begin = on ? a.vf_offsets[v] : 0, end = on ? a.vf_offsets[v + 1] : 0;
	for (uint32_t k = begin + sub; k < end; k += GATHER_LANES)
		acc = add3(acc, face_normal(P, a.faces, a.vf_corners[k] / 3).unit);
	return scale3(a.sign, lanes_sum3(acc));
}

// grid: (V GATHER_LANES / FH_BLOCK, n)
__device__ __forceinline__ void vertex_shade_block(const ShadeArgs &a, double *lum_out, double *colors_out, int bx, int b)
{
	const int t = bx * FH_BLOCK + threadIdx.x, v = t / GATHER_LANES, sub = t % GATHER_LANES;
	const bool on = v < a.V;
	const Vec3 acc = accumulated_normal(a, a.posed + (size_t)b * a.V * 3, v, sub, on);
	if (!on || sub != 0)
		return;
	const double len = sqrt(dot3(acc, acc));
	const Vec3 N = {acc.x / len, acc.y / len, acc.z / len};
	const double d = -dot3(N, load3(a.light));
	const double lum = (d > 0 ? d : 0.0) + a.ambient[0];
	const size_t at = (size_t)b * a.V + v;
	if (lum_out)
		lum_out[at] = lum;
	if (colors_out)
		for (int c = 0; c < a.C; c++)
			colors_out[at * a.C + c] = a.color[c] * lum;
}
__global__ __launch_bounds__(FH_BLOCK) void vertex_shade_kernel(ShadeArgs a, double *lum_out, double *colors_out)
{
	vertex_shade_block(a, lum_out, colors_out, blockIdx.x, blockIdx.y);
}

// b1: per (view, vertex), a 1-D grid over n V GATHER_LANES threads.  -> acc_b [n,V,3] (adjoint of the accumulated, not yet normalised,
// normal), out[0..3) light_b, out[3] ambient_b, out[4..4+C) color_b  (C <= 3)
__global__ __launch_bounds__(FH_BLOCK) void vertex_shade_b1_kernel(ShadeArgs a, const double *lum_b_in, const double *colors_b, double *acc_b, double *out,
																	double *partials, unsigned *counter)
{
	const long long t = (long long)blockIdx.x * FH_BLOCK + threadIdx.x, at = t / GATHER_LANES;
	const int sub = (int)(t % GATHER_LANES);
	const bool on = at < (long long)a.n * a.V;
	const int b = on ? (int)(at / a.V) : 0, v = on ? (int)(at % a.V) : 0;
	const Vec3 acc = accumulated_normal(a, a.posed + (size_t)b * a.V * 3, v, sub, on);
	double sums[7] = {0, 0, 0, 0, 0, 0, 0};
	if (on && sub == 0)
	{
		const double len = sqrt(dot3(acc, acc));
		const Vec3 N = {acc.x / len, acc.y / len, acc.z / len}, L = load3(a.light);
		const double d = -dot3(N, L);
		const double lum = (d > 0 ? d : 0.0) + a.ambient[0];
		double lum_b = lum_b_in ? lum_b_in[at] : 0.0;
		if (colors_b)
			for (int c = 0; c < a.C; c++)
			{
				const double g = colors_b[at * a.C + c];
				lum_b += g * a.color[c];
				sums[4 + c] = g * lum;
			}
		const double d_b = d > 0 ? lum_b : 0.0;
		sums[0] = -N.x * d_b, sums[1] = -N.y * d_b, sums[2] = -N.z * d_b, sums[3] = lum_b;
		const Vec3 N_b = scale3(-d_b, L);
		const double along = dot3(N, N_b);
		store3(acc_b + 3 * at, {(N_b.x - N.x * along) / len, (N_b.y - N.y * along) / len, (N_b.z - N.z * along) / len});
	}
	double total[7];
	if (grid_sum<7>(sums, partials, counter, total) && threadIdx.x < 4 + a.C)
		out[threadIdx.x] = total[threadIdx.x];
}

// b2: posed_b[b][v] = sum over the faces around v of the adjoint of that face's corner (a gather: every element written once);
// grid: (V GATHER_LANES / FH_BLOCK, n)
__global__ __launch_bounds__(FH_BLOCK) void vertex_shade_b2_kernel(ShadeArgs a, const double *acc_b, double *posed_b)
{
	const int t = blockIdx.x * FH_BLOCK + threadIdx.x, v = t / GATHER_LANES, sub = t % GATHER_LANES, b = blockIdx.y;
	const bool on = v < a.V;
	const double *P = a.posed + (size_t)b * a.V * 3, *A = acc_b + (size_t)b * a.V * 3;
	Vec3 g = {0, 0, 0};
	const uint32_t begin = on ? a.vf_offsets[v] : 0, end = on ? a.vf_offsets[v + 1] : 0;
	for (uint32_t k = begin + sub; k < end; k += GATHER_LANES)
	{
		const uint32_t slot = a.vf_corners[k], f = slot / 3, corner = slot % 3;
		const FaceNormal fn = face_normal(P, a.faces, f);
		const Vec3 unit_b = scale3(a.sign, add3(add3(load3(A + 3 * (size_t)fn.i0), load3(A + 3 * (size_t)fn.i1)), load3(A + 3 * (size_t)fn.i2)));
		const double along = dot3(fn.unit, unit_b);
		const Vec3 n_b = {(unit_b.x - fn.unit.x * along) / fn.len, (unit_b.y - fn.unit.y * along) / fn.len, (unit_b.z - fn.unit.z * along) / fn.len};
		const Vec3 e1_b = cross3(fn.e2, n_b), e2_b = cross3(n_b, fn.e1); // n = e1 x e2
		g = add3(g, corner == 0 ? scale3(-1, add3(e1_b, e2_b)) : corner == 1 ? e1_b : e2_b);
	}
	g = lanes_sum3(g);
	if (on && sub == 0)
		store3(posed_b + ((size_t)b * a.V + v) * 3, g);
}

// ---- rigid energy over a CSR of M = L^T L: grad = c M (x - ref), energy[0] = 0.5 (x - ref) . grad; with a data term at hand,
// energy[1] = data_weight * data_energy[0] + energy[0] (what a fitter's step reports, mesh_fitter.py:147).  grid: V GATHER_LANES / FH_BLOCK
struct RigidArgs
{
	const double *x, *ref;
	const uint32_t *offsets, *cols;
	const double *vals;
	double cregu;
	double *grad, *energy;
	const double *data_energy;
	double data_weight;
	double *partials;
	unsigned *counter;
	int V;
};
__device__ __forceinline__ void rigid_energy_block(const RigidArgs &a, unsigned bx, unsigned nblocks)
{
	const int t = bx * FH_BLOCK + threadIdx.x, v = t / GATHER_LANES, sub = t % GATHER_LANES;
	const bool on = v < a.V;
	Vec3 g = {0, 0, 0};
	const uint32_t begin = on ? a.offsets[v] : 0, end = on ? a.offsets[v + 1] : 0;
	for (uint32_t k = begin + sub; k < end; k += GATHER_LANES)
	{
		const size_t j = a.cols[k];
		g = add3(g, scale3(a.vals[k], sub3(load3(a.x + 3 * j), load3(a.ref + 3 * j))));
	}
	g = scale3(a.cregu, lanes_sum3(g));
	double e[1] = {0};
	if (on && sub == 0)
	{
		store3(a.grad + 3 * (size_t)v, g);
		e[0] = 0.5 * dot3(sub3(load3(a.x + 3 * (size_t)v), load3(a.ref + 3 * (size_t)v)), g);
	}
	double total[1];
	if (grid_sum<1>(e, a.partials, a.counter, total, bx, nblocks) && threadIdx.x == 0)
	{
		a.energy[0] = total[0];
		if (a.data_energy)
			a.energy[1] = a.data_weight * a.data_energy[0] + total[0];
	}
}
__global__ __launch_bounds__(FH_BLOCK) void rigid_energy_kernel(RigidArgs a) { rigid_energy_block(a, blockIdx.x, gridDim.x); }

// ---- what lies between pose + projection and the rasterizer in a fit iteration -- silhouette flags (from ij), vertex colours (from the
// posed vertices) and the rigid energy with its gradient (from the vertices) -- does not depend on one another: ONE launch, the
// workgroups of the three kernels side by side (each kernel of a replayed iteration costs 3 - 5 us whatever it does).
// grid: rigid_blocks + shade_x n + sil_x n workgroups (a part with no workgroups is not wanted)
struct FrontArgs
{
	RigidArgs rigid;
	ShadeArgs shade;
	double *lum_out, *colors_out;
	const double *ij;
	const uint32_t *faces, *edge_faces;
	uint8_t *flags;
	int T, V, clockwise;
	unsigned rigid_blocks, shade_x, sil_x;
};
__global__ __launch_bounds__(FH_BLOCK) void fit_front_kernel(FrontArgs a)
{
	unsigned bx = blockIdx.x;
	if (bx < a.rigid_blocks)
	{ // (first: its last workgroup adds the partial energies up)
		rigid_energy_block(a.rigid, bx, a.rigid_blocks);
		return;
	}
	bx -= a.rigid_blocks;
	if (bx < a.shade_x * (unsigned)a.shade.n)
	{
		vertex_shade_block(a.shade, a.lum_out, a.colors_out, (int)(bx % a.shade_x), (int)(bx / a.shade_x));
		return;
	}
	bx -= a.shade_x * (unsigned)a.shade.n;
	silhouette_flags_block(a.ij, a.faces, a.edge_faces, a.flags, a.T, a.V, a.clockwise, (int)(bx % a.sil_x), (int)(bx / a.sil_x));
}

// ---- sum (image - obs)^2 over a frame batch in the pixel type PixT, accumulated in double: the data energy of the colour fitters
// (mesh_fitter.py:296-318); the rasterizer's fit step back-propagates exactly this residual.  One partial per workgroup, fixed order.
constexpr int L2_BLOCKS = 256; // (every workgroup ends with a ticket on ONE counter word: ~30 ns each, one after the other -- 2 048
								// workgroups measured 64 us for a 50 MB frame, 1 024 workgroups 32 us, 512 workgroups 27 us)
constexpr int L2_ROUND = 4;	   // chunks of 32 bytes per array a thread has in flight
template <class PixT>
__global__ __launch_bounds__(FH_BLOCK) void l2_loss_kernel(const PixT *image, const PixT *obs, size_t count, double *out, double *partials, unsigned *counter,
															int clamp, double clamp_lo, double clamp_hi)
{ // clamp: sum (clamp(image, clamp_lo, clamp_hi) - obs)^2
	auto value = [&](PixT v) {
		const double x = (double)v;
		return clamp ? (x < clamp_lo ? clamp_lo : (x > clamp_hi ? clamp_hi : x)) : x;
	};
	constexpr int W = 32 / sizeof(PixT);
	struct alignas(32) Chunk
	{
		PixT v[W];
	};
	double s[1] = {0};
	const size_t chunks = count / W, stride = (size_t)gridDim.x * FH_BLOCK;
	for (size_t i = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; i < chunks; i += L2_ROUND * stride)
	{
		Chunk a[L2_ROUND], b[L2_ROUND];
#pragma unroll
		for (int u = 0; u < L2_ROUND; u++)
		{
			const size_t at = i + u * stride < chunks ? i + u * stride : i;
			a[u] = ((const Chunk *)image)[at], b[u] = ((const Chunk *)obs)[at];
		}
#pragma unroll
		for (int u = 0; u < L2_ROUND; u++)
			if (i + u * stride < chunks)
#pragma unroll
				for (int j = 0; j < W; j++)
				{
					const double r = value(a[u].v[j]) - (double)b[u].v[j];
					s[0] += r * r;
				}
	}
	if (blockIdx.x == 0 && threadIdx.x < count - chunks * W)
	{
		const double r = value(image[chunks * W + threadIdx.x]) - (double)obs[chunks * W + threadIdx.x];
		s[0] += r * r;
	}
	double total[1];
	if (grid_sum<1>(s, partials, counter, total) && threadIdx.x == 0)
		out[0] = total[0];
}

// ---- the data term of the depth fitter (deodr/mesh_fitter.py:108-123): depth = clamp(image, 0, max_depth), diff = (depth - obs)^2,
// loss = sum diff, image_b = d loss / d image = 2 (depth - obs) where the clamp passes (0 <= image <= max_depth), else 0
template <class PixT>
__global__ __launch_bounds__(FH_BLOCK) void depth_residual_kernel(const PixT *image, const double *obs, double max_depth, size_t count, double *depth, double *diff,
																   PixT *image_b, double *loss, double *partials, unsigned *counter)
{
	double s[1] = {0};
	for (size_t i = (size_t)blockIdx.x * FH_BLOCK + threadIdx.x; i < count; i += (size_t)gridDim.x * FH_BLOCK)
	{
		const double v = (double)image[i];
		const double d = v < 0 ? 0.0 : v > max_depth ? max_depth : v;
		const double r = d - obs[i];
		depth[i] = d;
		diff[i] = r * r;
		image_b[i] = (PixT)((v >= 0 && v <= max_depth) ? 2 * r : 0.0);
		s[0] += r * r;
	}
	double total[1];
	if (grid_sum<1>(s, partials, counter, total) && threadIdx.x == 0)
		loss[0] = total[0];
}

// ---- the loss of a fit step without a pass over the frame (deodr_hip_render_scene_fit_loss): the caller's table holds, per tile and in
// total, the loss sum (background - obs)^2 of a frame that is all background -- computed ONCE per observation by the two kernels
// below --; the forward raster only adds, per non-empty tile, (loss of the tile as rendered - its background loss).
template <class PixT>
__global__ __launch_bounds__(64) void background_loss_kernel(KParams p, double *table)
{ // one wavefront per tile: table[1 + view * ntiles + tile]
	const int tile = blockIdx.x, view = blockIdx.y, lane = threadIdx.x;
	const int px = (tile % p.L.tiles_x) * TILE + (lane & 7), py = (tile / p.L.tiles_x) * TILE + (lane >> 3);
	double r2 = 0;
	if (px < p.W && py < p.H)
	{
		const size_t pix = (size_t)py * p.W + px;
		const PixT *o = (const PixT *)p.obs + ((size_t)view * p.H * p.W + pix) * p.C;
		for (int c = 0; c < p.C; c++)
		{
			const double d = fit_value<true>(p, (double)(PixT)background_channel<PixT>(p, view, pix, c)) - (double)o[c]; // (the frame holds the background rounded to PixT)
			r2 += d * d;
		}
	}
	r2 = wave_sum(r2);
	if (lane == 0)
		table[1 + (size_t)view * p.L.ntiles + tile] = r2;
}
__global__ __launch_bounds__(FH_BLOCK) void background_loss_total_kernel(double *table, size_t count)
{ // table[0] = sum of the others, by one workgroup in a fixed order
	__shared__ double s_wave[FH_BLOCK / 64];
	double s = 0;
	for (size_t i = threadIdx.x; i < count; i += FH_BLOCK)
		s += table[1 + i];
	s = wave_sum(s);
	if ((threadIdx.x & 63) == 0)
		s_wave[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		double total = 0;
		for (int w = 0; w < FH_BLOCK / 64; w++)
			total += s_wave[w];
		table[0] = total;
	}
}

} // namespace
