// tools/probes/l2_atomic_probe.hip -- round 4: every global atomic of the rasterizer is executed at the memory side (TCC_EA0_ATOMIC ==
// TCC_ATOMIC) at ~12 G requests/s, and configs[4]'s kernels all run at that rate.  Are atomics of a narrower SCOPE executed in the
// XCD's own L2 instead, and how fast?  An L2-local read-modify-write is only correct when no other XCD touches the address, so the
// probe gives every XCD (s_getreg HW_REG_XCC_ID) a private copy of the array and adds the copies up at the end.
//   scope 0: agent (what unsafeAtomicAdd emits), 1: workgroup, 2: wavefront; layout 0: one shared array, 1: one copy per XCD
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o tools/probes/l2_atomic_probe tools/probes/l2_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id()
{
	unsigned v;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
	return v & 0xf;
}

template <class T, int SCOPE>
__device__ __forceinline__ void add(T *p, T v)
{
	if (SCOPE == 0)
		__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else if (SCOPE == 1)
		__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	else
		__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// N elements; every lane makes `adds` additions of 1 to pseudo-random elements (a texture-gradient-like scatter: runs of 4 neighbours)
template <class T, int SCOPE>
__global__ __launch_bounds__(256) void scatter_kernel(T *arr, size_t n, size_t copy_stride, int adds, unsigned *xcd_seen)
{
	const unsigned x = xcc_id();
	if (threadIdx.x == 0)
		atomicOr(xcd_seen, 1u << x);
	T *base = arr + (size_t)x * copy_stride;
	unsigned s = (blockIdx.x * 256u + threadIdx.x) / 4u * 2654435761u;
	for (int i = 0; i < adds; i++)
	{
		s = s * 1664525u + 1013904223u;
		const size_t at = ((size_t)(s >> 8) * 4u + (threadIdx.x & 3u)) % n;
		add<T, SCOPE>(base + at, (T)1);
	}
}

template <class T>
__global__ void sum_kernel(const T *arr, size_t n, size_t copy_stride, int copies, double *out)
{
	double acc = 0;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		for (int c = 0; c < copies; c++)
			acc += (double)arr[(size_t)c * copy_stride + i];
	atomicAdd(out, acc);
}

template <class T, int SCOPE>
void run(const char *tname, int layout, size_t n, hipStream_t st)
{
	const int copies = 8, blocks = 8192, adds = 64;
	T *arr;
	double *out;
	unsigned *seen;
	CK(hipMalloc(&arr, sizeof(T) * n * copies));
	CK(hipMalloc(&out, 8));
	CK(hipMalloc(&seen, 4));
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	float best = 1e9;
	double sum = 0;
	unsigned hseen = 0;
	for (int rep = 0; rep < 4; rep++)
	{
		CK(hipMemsetAsync(arr, 0, sizeof(T) * n * copies, st));
		CK(hipMemsetAsync(out, 0, 8, st));
		CK(hipMemsetAsync(seen, 0, 4, st));
		CK(hipEventRecord(a, st));
		hipLaunchKernelGGL((scatter_kernel<T, SCOPE>), dim3(blocks), dim3(256), 0, st, arr, n, layout ? n : 0, adds, seen);
		CK(hipEventRecord(b, st));
		hipLaunchKernelGGL(sum_kernel<T>, dim3(1024), dim3(256), 0, st, arr, n, n, layout ? copies : 1, out);
		CK(hipStreamSynchronize(st));
		float ms;
		CK(hipEventElapsedTime(&ms, a, b));
		best = ms < best ? ms : best;
		CK(hipMemcpy(&sum, out, 8, hipMemcpyDeviceToHost));
		CK(hipMemcpy(&hseen, seen, 4, hipMemcpyDeviceToHost));
	}
	const double expect = (double)blocks * 256 * adds;
	printf("%s scope %s, %s, %zu elements: %.1f us for %.0f M adds = %.1f G lane-ops/s; sum %s (%.0f of %.0f), XCDs seen 0x%x\n", tname,
		   SCOPE == 0 ? "agent" : (SCOPE == 1 ? "workgroup" : "wavefront"), layout ? "one copy per XCD" : "one shared array", n, best * 1e3, expect / 1e6,
		   expect / (best * 1e-3) / 1e9, sum == expect ? "EXACT" : "WRONG", sum, expect, hseen);
	CK(hipFree(arr));
	CK(hipFree(out));
	CK(hipFree(seen));
}

int main()
{
	hipStream_t st;
	CK(hipStreamCreate(&st));
	for (size_t n : {(size_t)1 << 16, (size_t)3 << 20})
	{ // 64 K elements (fits every L2) and 3 M elements (a 1024^2 x 3 texture gradient)
		run<float, 0>("f32", 0, n, st);
		run<float, 1>("f32", 0, n, st);
		run<float, 0>("f32", 1, n, st);
		run<float, 1>("f32", 1, n, st);
		run<float, 2>("f32", 1, n, st);
		run<double, 0>("f64", 0, n, st);
		run<double, 1>("f64", 0, n, st);
		run<double, 1>("f64", 1, n, st);
	}
	return 0;
}
