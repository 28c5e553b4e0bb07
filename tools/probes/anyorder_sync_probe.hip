// tools/probes/anyorder_sync_probe.hip -- round 4: can a consumer kernel launched with hipExtAnyOrderLaunch (no barrier bit in its AQL
// packet) run UNDER its producer on the same stream, ordered only by a device-side counter?
//  (1) producer leaves most of the chip free: when does the any-order consumer start?
//  (2) the mechanism finalize-under-forward would use: producer workgroups add to accumulators with returnless f64 atomics, wait for
//      vmcnt(0), then bump a per-view counter; consumer workgroups spin (bounded) on the counter, then read the accumulators with
//      agent-scope atomic loads and check the sums.  Repeated many times: any race shows up as a wrong sum.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o tools/probes/anyorder_sync_probe tools/probes/anyorder_sync_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

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

constexpr int NACC = 4096; // accumulators
__global__ __launch_bounds__(64) void producer(double *acc, unsigned *done, int adds, unsigned long long spin_ticks)
{ // every workgroup adds 1.0 to `adds` x 64 accumulators (scattered), some of them after a delay, then signals
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	if (blockIdx.x % 7 == 0)
		while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks)
			__builtin_amdgcn_s_sleep(4);
	for (int i = 0; i < adds; i++)
		unsafeAtomicAdd(acc + ((blockIdx.x * 131u + i * 64u + threadIdx.x * 17u) % NACC), 1.0);
	__builtin_amdgcn_s_waitcnt(0); // vmcnt(0) expcnt(0) lgkmcnt(0): the returnless atomics above have been performed
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
	if (threadIdx.x == 0)
		__hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(64) void consumer(double *acc, unsigned *done, unsigned expect, double *out, unsigned *status)
{
	unsigned polls = 0;
	while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect)
	{
		__builtin_amdgcn_s_sleep(16);
		if (++polls > 2000000u)
		{ // never hang the box: give up and say so
			if (threadIdx.x == 0)
				atomicOr(status, 1u);
			return;
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	if (threadIdx.x == 0 && polls > 0)
		atomicAdd(status + 1, 1u); // workgroups that actually had to wait
	const int i = blockIdx.x * 64 + threadIdx.x;
	if (i < NACC)
	{
		out[i] = __hip_atomic_load(acc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		acc[i] = 0; // self-cleaning, like finalize_kernel's accumulators (the next producer runs behind a barrier packet)
	}
}

int main()
{
	hipStream_t st;
	CK(hipStreamCreate(&st));
	{ // ---- (1)
		const int NA = 2048, NB = 256;
		unsigned long long *ta, *tb;
		CK(hipMalloc(&ta, 16 * NA));
		CK(hipMalloc(&tb, 16 * NB));
		std::vector<unsigned long long> ha(2 * NA), hb(2 * NB);
		for (int rep = 0; rep < 2; rep++)
			for (int flags = 0; flags < 2; flags++)
			{
				hipLaunchKernelGGL(spin_kernel, dim3(NA), dim3(64), 0, st, ta, 3000ull); // 30 us per workgroup (100 MHz counter)
				hipExtLaunchKernelGGL(spin_kernel, dim3(NB), dim3(64), 0, st, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, tb, 100ull);
				CK(hipStreamSynchronize(st));
				CK(hipMemcpy(ha.data(), ta, 16 * NA, hipMemcpyDeviceToHost));
				CK(hipMemcpy(hb.data(), tb, 16 * NB, hipMemcpyDeviceToHost));
				unsigned long long a0 = ~0ull, a_last = 0, a_end = 0, b0 = ~0ull, b_end = 0;
				for (int i = 0; i < NA; i++)
					a0 = std::min(a0, ha[2 * i]), a_last = std::max(a_last, ha[2 * i]), a_end = std::max(a_end, ha[2 * i + 1]);
				for (int i = 0; i < NB; i++)
					b0 = std::min(b0, hb[2 * i]), b_end = std::max(b_end, hb[2 * i + 1]);
				printf("producer fills a quarter of the chip, any-order flag %d: A runs [0, %.1f] us (last start %.1f); B runs [%.1f, %.1f] us\n", flags, (a_end - a0) * 0.01,
					   (a_last - a0) * 0.01, ((long long)b0 - (long long)a0) * 0.01, ((long long)b_end - (long long)a0) * 0.01);
			}
	}
	{ // ---- (2)
		double *acc, *out;
		unsigned *done, *status;
		CK(hipMalloc(&acc, 8 * NACC));
		CK(hipMalloc(&out, 8 * NACC));
		CK(hipMalloc(&done, 64));
		CK(hipMalloc(&status, 64));
		CK(hipMemset(acc, 0, 8 * NACC));
		CK(hipMemset(status, 0, 64));
		const int NP = 20000, ADDS = 8; // 20 000 producer workgroups (more than the chip holds at once) x 8 x 64 adds
		std::vector<double> h(NACC);
		long long bad = 0, waited_runs = 0;
		hipEvent_t e0, e1;
		CK(hipEventCreate(&e0));
		CK(hipEventCreate(&e1));
		for (int mode = 0; mode < 2; mode++)
		{
			const int ITER = 200;
			CK(hipEventRecord(e0, st));
			for (int it = 0; it < ITER; it++)
			{
				CK(hipMemsetAsync(done, 0, 4, st));
				hipLaunchKernelGGL(producer, dim3(NP), dim3(64), 0, st, acc, done, ADDS, 500ull);
				hipExtLaunchKernelGGL(consumer, dim3(NACC / 64), dim3(64), 0, st, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, acc, done, (unsigned)NP, out, status);
				if (it % 20 == 19)
				{
					CK(hipStreamSynchronize(st));
					CK(hipMemcpy(h.data(), out, 8 * NACC, hipMemcpyDeviceToHost));
					double sum = 0;
					for (double v : h)
						sum += v;
					if (sum != (double)NP * ADDS * 64)
						bad++, printf("  iteration %d: sum %.0f, expected %.0f\n", it, sum, (double)NP * ADDS * 64);
				}
			}
			CK(hipEventRecord(e1, st));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			unsigned hs[2];
			CK(hipMemcpy(hs, status, 8, hipMemcpyDeviceToHost));
			printf("producer -> consumer, any-order %d: %.1f us per pair; wrong sums %lld; consumer workgroups that had to wait: %u (timeouts: %u)\n", mode, ms * 1e3 / ITER, bad,
				   hs[1], hs[0]);
			CK(hipMemset(status, 0, 64));
		}
	}
	return 0;
}
