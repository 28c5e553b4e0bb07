// tools/probes/order_atomic_probe.hip -- two questions asked of the hardware before round 4's restructuring:
//  (1) does hipExtAnyOrderLaunch let a kernel start while its predecessor ON THE SAME STREAM is still running (the header says
//      "not supported on GFX9xx"), and are the workgroups of consecutive packets of one queue dispatched in order?
//  (2) what is the rate of memory-side f64 atomic adds (finalize_kernel: 18 per front-facing triangle), as a function of how many
//      lanes of an instruction hit the same vertex, and what does merging them in LDS first buy?
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o tools/probes/order_atomic_probe tools/probes/order_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void spin_kernel(unsigned long long *t, unsigned long long ticks)
{
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	while (__builtin_amdgcn_s_memrealtime() - t0 < ticks)
		__builtin_amdgcn_s_sleep(8);
	if (threadIdx.x == 0)
	{
		t[2 * blockIdx.x] = t0;
		t[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
	}
}

// ---- atomics: T "triangles", each adds 6 contiguous doubles to 3 "vertices" of a V-vertex array
template <int MODE> // 0: straight global atomics, 1: LDS hash-merge per workgroup first
__global__ __launch_bounds__(256) void atomic_kernel(double *grad, const unsigned *faces, int T, int V)
{
	constexpr int SLOTS = 1024;
	__shared__ unsigned s_key[SLOTS];
	__shared__ double s_val[SLOTS][6];
	const int k = blockIdx.x * 256 + threadIdx.x;
	if (MODE == 1)
	{
		for (int i = threadIdx.x; i < SLOTS; i += 256)
		{
			s_key[i] = 0xffffffffu;
			for (int c = 0; c < 6; c++)
				s_val[i][c] = 0;
		}
		__syncthreads();
	}
	if (k < T)
	{
		unsigned f[3] = {faces[3 * k], faces[3 * k + 1], faces[3 * k + 2]};
		for (int i = 0; i < 3; i++)
		{
			double v[6];
			for (int c = 0; c < 6; c++)
				v[c] = 1.0 + c + 0.001 * (k & 7);
			if (MODE == 0)
			{
				for (int c = 0; c < 6; c++)
					unsafeAtomicAdd(grad + (size_t)f[i] * 6 + c, v[c]);
			}
			else
			{
				unsigned h = (f[i] * 2654435761u) >> 22; // 10 bits
				for (int probe = 0; probe < SLOTS; probe++, h = (h + 1) & (SLOTS - 1))
				{
					const unsigned old = atomicCAS(&s_key[h], 0xffffffffu, f[i]);
					if (old == 0xffffffffu || old == f[i])
						break;
				}
				for (int c = 0; c < 6; c++)
					unsafeAtomicAdd(&s_val[h][c], v[c]);
			}
		}
	}
	if (MODE == 1)
	{
		__syncthreads();
		for (int i = threadIdx.x; i < SLOTS * 6; i += 256)
		{
			const int s = i / 6, c = i - 6 * s;
			const unsigned key = s_key[s];
			if (key != 0xffffffffu)
				unsafeAtomicAdd(grad + (size_t)key * 6 + c, s_val[s][c]);
		}
	}
}

static double time_ms(hipStream_t st, std::function<void()> fn, int reps)
{
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	fn();
	CK(hipStreamSynchronize(st));
	CK(hipEventRecord(a, st));
	for (int i = 0; i < reps; i++)
		fn();
	CK(hipEventRecord(b, st));
	CK(hipEventSynchronize(b));
	float ms;
	CK(hipEventElapsedTime(&ms, a, b));
	return ms / reps;
}

int main()
{
	hipStream_t st;
	CK(hipStreamCreate(&st));
	{ // ---- (1) any-order launch
		const int NA = 8192, NB = 256;
		unsigned long long *ta, *tb;
		CK(hipMalloc(&ta, 16 * NA));
		CK(hipMalloc(&tb, 16 * NB));
		std::vector<unsigned long long> ha(2 * NA), hb(2 * NB);
		for (int flags = 0; flags < 2; flags++)
		{
			hipLaunchKernelGGL(spin_kernel, dim3(NA), dim3(64), 0, st, ta, 2000ull); // 20 us per workgroup (100 MHz counter)
			hipExtLaunchKernelGGL(spin_kernel, dim3(NB), dim3(64), 0, st, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, tb, 100ull);
			CK(hipStreamSynchronize(st));
			CK(hipMemcpy(ha.data(), ta, 16 * NA, hipMemcpyDeviceToHost));
			CK(hipMemcpy(hb.data(), tb, 16 * NB, hipMemcpyDeviceToHost));
			unsigned long long a0 = ~0ull, a_last_start = 0, a_end = 0, b0 = ~0ull;
			for (int i = 0; i < NA; i++)
				a0 = std::min(a0, ha[2 * i]), a_last_start = std::max(a_last_start, ha[2 * i]), a_end = std::max(a_end, ha[2 * i + 1]);
			for (int i = 0; i < NB; i++)
				b0 = std::min(b0, hb[2 * i]);
			printf("any-order flag %d: kernel A runs [0, %.1f] us, its last workgroup starts at %.1f us; kernel B's first workgroup starts at %.1f us -> %s\n", flags,
				   (a_end - a0) * 0.01, (a_last_start - a0) * 0.01, ((long long)b0 - (long long)a0) * 0.01,
				   b0 < a_end ? (b0 >= a_last_start ? "OVERLAPS, dispatched in order" : "OVERLAPS, NOT in dispatch order") : "serialised");
		}
	}
	{ // ---- (2) atomic rates.  faces of a UV sphere strip (consecutive triangles share vertices) and random faces
		const int V = 10002, T = 20000, VIEWS = 8;
		std::vector<unsigned> grid_faces(3 * T * VIEWS), rnd_faces(3 * T * VIEWS);
		const int nu = 100;
		for (int v = 0; v < VIEWS; v++)
			for (int k = 0; k < T; k++)
			{
				const int q = k / 2, r = q / nu, c = q % nu;
				const unsigned a = (r * (nu + 1) + c) % V, b = (a + 1) % V, cc = (a + nu + 1) % V, d = (cc + 1) % V;
				unsigned *f = &grid_faces[3 * ((size_t)v * T + k)];
				if (k & 1)
					f[0] = b, f[1] = d, f[2] = cc;
				else
					f[0] = a, f[1] = b, f[2] = cc;
				for (int i = 0; i < 3; i++)
					rnd_faces[3 * ((size_t)v * T + k) + i] = (unsigned)(rand() % V);
			}
		unsigned *d_faces;
		double *d_grad;
		CK(hipMalloc(&d_faces, 4 * grid_faces.size()));
		CK(hipMalloc(&d_grad, 8 * 6 * (size_t)V * VIEWS));
		CK(hipMemset(d_grad, 0, 8 * 6 * (size_t)V * VIEWS));
		for (int pattern = 0; pattern < 2; pattern++)
		{
			CK(hipMemcpy(d_faces, pattern ? rnd_faces.data() : grid_faces.data(), 4 * grid_faces.size(), hipMemcpyHostToDevice));
			for (int mode = 0; mode < 2; mode++)
			{
				const int total = T * VIEWS; // one launch = 8 views' worth, every view its own gradient array
				auto fn = [&]() {
					for (int v = 0; v < 1; v++)
					{
						if (mode == 0)
							hipLaunchKernelGGL(atomic_kernel<0>, dim3((total + 255) / 256), dim3(256), 0, st, d_grad, d_faces, total, V);
						else
							hipLaunchKernelGGL(atomic_kernel<1>, dim3((total + 255) / 256), dim3(256), 0, st, d_grad, d_faces, total, V);
					}
				};
				const double ms = time_ms(st, fn, 20);
				printf("atomics: %s faces, %s: %.1f us per launch of %d triangles x 18 f64 adds = %.1f G lane-ops/s\n", pattern ? "random" : "grid-ordered",
					   mode ? "LDS hash-merge per 256 triangles" : "straight global atomics", ms * 1e3, total, total * 18.0 / (ms * 1e-3) / 1e9);
			}
		}
	}
	return 0;
}
