// deodr_amd/csrc/dr_fronthalf.h -- part of the single translation unit dr_kernels.hip (device code, gfx950 / wave64).
// The O(V) algebra either side of the rasterizer in a fit iteration, as a handful of kernels:
//
//   rigid_transform_kernel (+ _b)   centred vertices -> posed vertices of every view: qrot(q, v) + t
//                                   (deodr/tools.py:8-35 qrot / qrot_backward; deodr/mesh_fitter.py:139-151)
//   project_points_kernel (+ _b)    pinhole camera with OpenCV's distortion: posed vertices -> image coordinates + depths
//                                   (Camera.project_points / project_points_backward, deodr/differentiable_renderer.py:341-438)
//   silhouette_flags_kernel         edge flags of every view: exactly one of the (at most two) faces on an edge is front-facing in
//                                   the image (TriMeshAdjacencies.edge_on_silhouette, deodr/triangulated_mesh.py:153-166)
//   momentum_update_kernel          x += s, s = (1 - damping)(inertia s + (1 - inertia) clamp(-factor g)) for all parameters of a
//                                   fitter in one launch (deodr/mesh_fitter.py:153-190)
//
// Why kernels: as torch ops one fit iteration is ~240 launches of 2 - 8 us each whatever the mesh size (profiles/README.md, round 3);
// these four groups are ~150 of them.  The arithmetic is double like the reference's NumPy; nothing here is on the rasterizer's path.
#pragma once

#include "dr_finalize.h"

using namespace dr;

namespace
{

struct Vec3
{
	double x, y, z;
};
__device__ __forceinline__ Vec3 cross3(const Vec3 &a, const Vec3 &b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ Vec3 add3(const Vec3 &a, const Vec3 &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ Vec3 scale3(double s, const Vec3 &a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot3(const Vec3 &a, const Vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

constexpr int FH_BLOCK = 256;

// ---- grid-wide sums without atomics on the values.  Every thread of every workgroup of a launch whose blockIdx.x runs over the
// reduced range brings K values; `partials` holds K doubles per workgroup, `counter` one word that is zero between launches (the last
// workgroup to arrive resets it).  -> true in ALL threads of the last workgroup, whose total[] then holds the sums (in workgroup order).
// Deterministic grid-wide sums without atomics on the values.  Every thread of every workgroup of a launch (blockIdx.x over the reduced
// range) brings K values; `partials` holds K doubles per workgroup, `counter` one word that is zero between launches (the last
// workgroup to arrive resets it).  -> true in ALL threads of the last workgroup, whose total[] then holds the sums: the partial of
// workgroup i is added by thread i mod FH_BLOCK in the order i, i + FH_BLOCK, ...; then the lanes of a wavefront (DPP tree), then
// the wavefronts in order -- fixed by the launch geometry alone.
template <int K>
__device__ __forceinline__ bool grid_sum(const double (&v)[K], double *partials, unsigned *counter, double (&total)[K], unsigned block = blockIdx.x,
										 unsigned nblocks = gridDim.x)
{ // (block / nblocks: the position of this workgroup among the ones that take part, for kernels that are a part of a launch)
	__shared__ double s_wave[FH_BLOCK / 64][K];
	__shared__ int s_last;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int k = 0; k < K; k++)
	{
		const double s = wave_sum(v[k]);
		if (lane == 0)
			s_wave[wave][k] = s;
	}
	__syncthreads();
	if (threadIdx.x < K)
	{
		double s = 0;
		for (int w = 0; w < FH_BLOCK / 64; w++)
			s += s_wave[w][threadIdx.x];
		partials[(size_t)block * K + threadIdx.x] = s;
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		__threadfence(); // release (after the barrier: the workgroup's partials are visible to the device before its ticket)
		s_last = atomicAdd(counter, 1u) == nblocks - 1;
	}
	__syncthreads();
	if (!s_last)
		return false;
	__threadfence(); // acquire: the loads below see every workgroup's partial
	double mine[K];
#pragma unroll
	for (int k = 0; k < K; k++)
		mine[k] = 0;
	for (unsigned i = threadIdx.x; i < nblocks; i += FH_BLOCK)
#pragma unroll
		for (int k = 0; k < K; k++)
			mine[k] += partials[(size_t)i * K + k];
#pragma unroll
	for (int k = 0; k < K; k++)
	{
		const double s = wave_sum(mine[k]);
		if (lane == 0)
			s_wave[wave][k] = s; // (every thread is past its reads of s_wave: the barriers above)
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < K; k++)
	{
		total[k] = 0;
		for (int w = 0; w < FH_BLOCK / 64; w++)
			total[k] += s_wave[w][k];
	}
	if (threadIdx.x == 0)
		atomicExch(counter, 0u);
	return true;
}

// ---- rigid transform: out[b][v] = qrot(q[b], v[v]) + t[b]   (q = (x, y, z, w), unit)
__global__ __launch_bounds__(FH_BLOCK) void rigid_transform_kernel(const double *vc, const double *q, const double *t, double *out, int V, int n)
{
	const int v = blockIdx.x * FH_BLOCK + threadIdx.x, b = blockIdx.y;
	if (v >= V)
		return;
	const Vec3 p = {vc[3 * v], vc[3 * v + 1], vc[3 * v + 2]}, u = {q[4 * b], q[4 * b + 1], q[4 * b + 2]};
	const double w = q[4 * b + 3];
	const Vec3 a = cross3(u, p), bb = cross3(u, a);
	double *o = out + ((size_t)b * V + v) * 3;
	o[0] = p.x + 2 * (w * a.x + bb.x) + t[3 * b];
	o[1] = p.y + 2 * (w * a.y + bb.y) + t[3 * b + 1];
	o[2] = p.z + 2 * (w * a.z + bb.z) + t[3 * b + 2];
}

// adjoint: one thread per vertex walks the views (vc_b[v] = sum over views: no atomics, fixed order); the sums over the vertices --
// q_b [n,4], t_b [n,3] -- are reduced per wavefront and leave with one atomic per value, wavefront and view (zeroed by the caller)
__global__ __launch_bounds__(FH_BLOCK) void rigid_transform_b_kernel(const double *vc, const double *q, const double *out_b, double *vc_b, double *q_b,
																	  double *t_b, int V, int n)
{
	const int v = blockIdx.x * FH_BLOCK + threadIdx.x, lane = threadIdx.x & 63;
	const bool on = v < V;
	const Vec3 p = on ? Vec3{vc[3 * v], vc[3 * v + 1], vc[3 * v + 2]} : Vec3{0, 0, 0};
	Vec3 acc = {0, 0, 0};
	for (int b = 0; b < n; b++)
	{
		const Vec3 u = {q[4 * b], q[4 * b + 1], q[4 * b + 2]};
		const double w = q[4 * b + 3];
		Vec3 g = {0, 0, 0};
		if (on)
		{
			const double *gb = out_b + ((size_t)b * V + v) * 3;
			g = {gb[0], gb[1], gb[2]};
		}
		const Vec3 a = cross3(u, p);
		// r = p + 2 w a + 2 bb, a = u x p, bb = u x a
		const double w_b = 2 * dot3(g, a);
		const Vec3 bb_b = scale3(2, g);
		Vec3 a_b = add3(scale3(2 * w, g), cross3(bb_b, u)); // from 2 w a and from bb = u x a
		Vec3 u_b = cross3(a, bb_b);							   // from bb = u x a
		u_b = add3(u_b, cross3(p, a_b));					   // from a = u x p
		const Vec3 p_b = add3(g, cross3(a_b, u));
		acc = add3(acc, p_b);
		const double sums[7] = {u_b.x, u_b.y, u_b.z, w_b, g.x, g.y, g.z};
#pragma unroll
		for (int i = 0; i < 7; i++)
		{
			const double s = wave_sum(sums[i]); // (every lane takes part: lanes beyond V hold zeros)
			if (lane == 0 && s != 0)
				atomic_add_f64(i < 4 ? q_b + 4 * b + i : t_b + 3 * b + (i - 4), s);
		}
	}
	if (on)
	{
		vc_b[3 * v] = acc.x;
		vc_b[3 * v + 1] = acc.y;
		vc_b[3 * v + 2] = acc.z;
	}
}

// ---- projection (dr.py:341-395): pc = R p + T; (x, y) = pc.xy / pc.z; distortion (k1, k2, p1, p2, k3); ij = K[:2,:2] (xd, yd) + K[:2,2]
struct CameraRow
{
	double E[12], K[6], d[5];
	bool distort;
};
__device__ __forceinline__ CameraRow load_camera(const double *extrinsic, const double *intrinsic, const double *distortion, int b)
{
	CameraRow c;
#pragma unroll
	for (int i = 0; i < 12; i++)
		c.E[i] = extrinsic[12 * b + i];
#pragma unroll
	for (int i = 0; i < 6; i++)
		c.K[i] = intrinsic[9 * b + i]; // the first two rows of the 3 x 3 matrix
	c.distort = distortion != nullptr;
#pragma unroll
	for (int i = 0; i < 5; i++)
		c.d[i] = c.distort ? distortion[5 * b + i] : 0.0;
	return c;
}

__global__ __launch_bounds__(FH_BLOCK) void project_points_kernel(const double *points, const double *extrinsic, const double *intrinsic,
																   const double *distortion, double *ij, double *depths, int V, int n)
{
	const int v = blockIdx.x * FH_BLOCK + threadIdx.x, b = blockIdx.y;
	if (v >= V)
		return;
	const CameraRow c = load_camera(extrinsic, intrinsic, distortion, b);
	const double *pp = points + ((size_t)b * V + v) * 3;
	const double px = pp[0], py = pp[1], pz = pp[2];
	const double cx = c.E[0] * px + c.E[1] * py + c.E[2] * pz + c.E[3], cy = c.E[4] * px + c.E[5] * py + c.E[6] * pz + c.E[7],
				 cz = c.E[8] * px + c.E[9] * py + c.E[10] * pz + c.E[11];
	double x = cx / cz, y = cy / cz;
	if (c.distort)
	{
		const double k1 = c.d[0], k2 = c.d[1], p1 = c.d[2], p2 = c.d[3], k3 = c.d[4];
		const double x2 = x * x, y2 = y * y, r2 = x2 + y2, r4 = r2 * r2;
		const double radial = 1 + k1 * r2 + k2 * r4 + k3 * (r2 * r4);
		const double xd = x * radial + (2 * p1 * x * y + p2 * (r2 + 2 * x2)), yd = y * radial + (p1 * (r2 + 2 * y2) + 2 * p2 * x * y);
		x = xd, y = yd;
	}
	const size_t at = (size_t)b * V + v;
	ij[2 * at] = c.K[0] * x + c.K[1] * y + c.K[2];
	ij[2 * at + 1] = c.K[3] * x + c.K[4] * y + c.K[5];
	depths[at] = cz;
}

__global__ __launch_bounds__(FH_BLOCK) void project_points_b_kernel(const double *points, const double *extrinsic, const double *intrinsic,
																	 const double *distortion, const double *ij_b, const double *depths_b, double *points_b, int V,
																	 int n)
{ // adjoint of the above (Camera.project_points_backward, dr.py:397-438); depths_b may be NULL
	const int v = blockIdx.x * FH_BLOCK + threadIdx.x, b = blockIdx.y;
	if (v >= V)
		return;
	const CameraRow c = load_camera(extrinsic, intrinsic, distortion, b);
	const size_t at = (size_t)b * V + v;
	const double *pp = points + at * 3;
	const double px = pp[0], py = pp[1], pz = pp[2];
	const double cx = c.E[0] * px + c.E[1] * py + c.E[2] * pz + c.E[3], cy = c.E[4] * px + c.E[5] * py + c.E[6] * pz + c.E[7],
				 cz = c.E[8] * px + c.E[9] * py + c.E[10] * pz + c.E[11];
	const double x = cx / cz, y = cy / cz;
	const double g0 = ij_b[2 * at], g1 = ij_b[2 * at + 1];
	double xd_b = c.K[0] * g0 + c.K[3] * g1, yd_b = c.K[1] * g0 + c.K[4] * g1;
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
	const double cx_b = x_b / cz, cy_b = y_b / cz, cz_b = (depths_b ? depths_b[at] : 0.0) - (x * x_b + y * y_b) / cz;
	double *o = points_b + at * 3;
	o[0] = c.E[0] * cx_b + c.E[4] * cy_b + c.E[8] * cz_b;
	o[1] = c.E[1] * cx_b + c.E[5] * cy_b + c.E[9] * cz_b;
	o[2] = c.E[2] * cx_b + c.E[6] * cy_b + c.E[10] * cz_b;
}

// ---- silhouette flags: flag[b][f][e] = 1 when exactly one face on edge e of face f is front-facing in view b.
// edge_faces [T,3]: the face on the other side of edge (v_e, v_{e+1}) of face f, or 0xffffffff on a boundary (static per mesh).
__device__ __forceinline__ void silhouette_flags_block(const double *ij, const uint32_t *faces, const uint32_t *edge_faces, uint8_t *flags, int T, int V,
													   int clockwise, int bx, int b)
{
	const int f = bx * FH_BLOCK + threadIdx.x;
	if (f >= T)
		return;
	const double *p = ij + (size_t)b * V * 2;
	auto visible = [&](uint32_t face) {
		const uint32_t i0 = faces[3 * face], i1 = faces[3 * face + 1], i2 = faces[3 * face + 2];
		const double ux = p[2 * i1] - p[2 * i0], uy = p[2 * i1 + 1] - p[2 * i0 + 1], vx = p[2 * i2] - p[2 * i0], vy = p[2 * i2 + 1] - p[2 * i0 + 1];
		const double cr = ux * vy - uy * vx;
		return clockwise ? cr > 0 : cr < 0;
	};
	const bool mine = visible((uint32_t)f);
#pragma unroll
	for (int e = 0; e < 3; e++)
	{
		const uint32_t other = edge_faces[3 * f + e];
		const int count = (mine ? 1 : 0) + ((other != 0xffffffffu && visible(other)) ? 1 : 0);
		flags[((size_t)b * T + f) * 3 + e] = count == 1 ? 1 : 0;
	}
}
__global__ __launch_bounds__(FH_BLOCK) void silhouette_flags_kernel(const double *ij, const uint32_t *faces, const uint32_t *edge_faces, uint8_t *flags, int T,
																	 int V, int clockwise)
{
	silhouette_flags_block(ij, faces, edge_faces, flags, T, V, clockwise, blockIdx.x, blockIdx.y);
}

// ---- momentum update of up to MOMENTUM_MAX parameter tensors in one launch, IN PLACE
constexpr int MOMENTUM_MAX = 8;
struct MomentumArgs
{
	double *x[MOMENTUM_MAX], *speed[MOMENTUM_MAX];
	const double *grad[MOMENTUM_MAX], *grad2[MOMENTUM_MAX]; // the step follows -(grad_scale (grad - grad_mean) + grad2): grad2 = the rigid energy's
	const double *grad_mean[MOMENTUM_MAX];					// optional [3]: subtracted from every row of a [count/3, 3] gradient (zero-mean projection)
	double *mean_out[MOMENTUM_MAX];							// optional [3]: column mean of the updated [count/3, 3] tensor (next step's centring)
	double grad_scale[MOMENTUM_MAX], factor[MOMENTUM_MAX], step_max[MOMENTUM_MAX]; // step_max <= 0: no clamp
	int count[MOMENTUM_MAX], normalize_rows[MOMENTUM_MAX]; // normalize_rows = row length: rows of x renormalised afterwards (quaternions)
	int n;
	double inertia, damping;
	double *partials; // 3 doubles per workgroup and tensor (mean_out)
	double *energy;	  // optional [2]: energy[1] = data_weight * data_energy[0] + energy[0], what a fitter's step reports (mesh_fitter.py:147)
	const double *data_energy;
	double data_weight;
	unsigned *counters; // one word per tensor
};
__device__ __forceinline__ double momentum_step(const MomentumArgs &a, int k, int at, int column)
{
	double g = a.grad[k][at];
	if (a.grad_mean[k])
		g -= a.grad_mean[k][column];
	g = g * a.grad_scale[k] + (a.grad2[k] ? a.grad2[k][at] : 0.0);
	double step = -g * a.factor[k];
	if (a.step_max[k] > 0)
		step = fmin(fmax(step, -a.step_max[k]), a.step_max[k]);
	const double s = (1 - a.damping) * (a.speed[k][at] * a.inertia + (1 - a.inertia) * step);
	a.speed[k][at] = s;
	return s;
}
__global__ __launch_bounds__(FH_BLOCK) void momentum_update_kernel(MomentumArgs a)
{
	const int k = blockIdx.y;
	const int i = blockIdx.x * FH_BLOCK + threadIdx.x;
	const int rows = a.normalize_rows[k];
	if (a.energy && k == 0 && i == 0)
		a.energy[1] = a.data_weight * a.data_energy[0] + a.energy[0];
	if (rows > 0)
	{ // one thread per row (a handful of quaternions)
		const int nrow = a.count[k] / rows;
		if (i >= nrow)
			return;
		double norm2 = 0;
		for (int j = 0; j < rows; j++)
		{
			const int at = i * rows + j;
			const double xn = a.x[k][at] + momentum_step(a, k, at, 0);
			a.x[k][at] = xn;
			norm2 += xn * xn;
		}
		const double inv = 1 / sqrt(norm2);
		for (int j = 0; j < rows; j++)
			a.x[k][i * rows + j] *= inv;
		return;
	}
	double col[3] = {0, 0, 0};
	if (i < a.count[k])
	{
		const double xn = a.x[k][i] + momentum_step(a, k, i, i % 3);
		a.x[k][i] = xn;
		col[i % 3] = xn;
	}
	if (!a.mean_out[k])
		return; // (uniform per workgroup row)
	double total[3];
	if (grid_sum<3>(col, a.partials + (size_t)k * gridDim.x * 3, a.counters + k, total) && threadIdx.x < 3)
		a.mean_out[k][threadIdx.x] = total[threadIdx.x] / (a.count[k] / 3);
}

} // namespace
