// round 6: (1) accuracy of v_rcp_f64 (what dr_math.h's quick_floor_quotient assumes: relative error <= 2^-20), (2) quick_floor_quotient against
// floor(a / b) / ceil(a / b) on random and on adversarial operands (integer quotients, quotients one ulp either side of an integer, tiny / huge
// divisors).  Tools only; not part of the library.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o rcp_probe rcp_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include "../../deodr_amd/csrc/dr_math.h"

__device__ uint64_t rng(uint64_t &s)
{
	s ^= s << 13, s ^= s >> 7, s ^= s << 17;
	return s;
}
__global__ void rcp_error(double *worst, int iters)
{
	uint64_t s = 0x9e3779b97f4a7c15ull * (blockIdx.x * 256 + threadIdx.x + 1);
	double w = 0;
	for (int i = 0; i < iters; i++)
	{
		const uint64_t u = rng(s);
		const int e = (int)(rng(s) % 600) - 300; // 2^-300 .. 2^300
		double b = ldexp(1.0 + (double)(u >> 11) * 0x1p-53, e);
		if (u & 1)
			b = -b;
		const double r = __builtin_amdgcn_rcp(b), err = fabs(__builtin_fma(-b, r, 1.0)); // |1 - b r|, exact to an ulp
		w = err > w ? err : w;
	}
	atomicMax((unsigned long long *)worst, (unsigned long long)__double_as_longlong(w)); // (positive doubles order as integers)
}
// kind 0: random a, b in a pixel-like range; 1: integer quotient k (a = k b exactly when representable); 2: a = nextafter(k b, +-inf);
// 3: |b| tiny / huge
__global__ void quotient_check(unsigned long long *counts, int iters, int kind)
{
	uint64_t s = 0xda942042e4dd58b5ull * (blockIdx.x * 256 + threadIdx.x + 1) + kind;
	unsigned long long quick = 0, bad = 0, tested = 0;
	for (int i = 0; i < iters; i++)
	{
		const double u1 = (double)(rng(s) >> 11) * 0x1p-53, u2 = (double)(rng(s) >> 11) * 0x1p-53;
		double a, b;
		if (kind == 0)
		{
			b = (u1 - 0.5) * 2000.0;
			a = (u2 - 0.5) * 4e6;
		}
		else
		{
			const int k = (int)(rng(s) % 4001) - 2000;
			b = (u1 - 0.5) * ((kind == 3) ? ((rng(s) & 1) ? 1e-300 : 1e300) : 64.0);
			if (rng(s) & 1)
				b = floor(b * 16) / 16; // (often a short mantissa: k b exact)
			a = (double)k * b;
			if (kind == 2)
				a = nextafter(a, (rng(s) & 1) ? INFINITY : -INFINITY);
		}
		if (!(fabs(b) * DR_SHRT_MAX > fabs(a) + fabs(b)))
			continue;
		tested++;
		double fl;
		if (dr::quick_floor_quotient(a, b, fl))
		{
			quick++;
			const double q = a / b;
			if (fl != floor(q) || fl + 1 != ceil(q))
				bad++;
		}
	}
	atomicAdd(&counts[0], tested);
	atomicAdd(&counts[1], quick);
	atomicAdd(&counts[2], bad);
}
int main()
{
	double *w;
	unsigned long long *c, h[3];
	hipMalloc(&w, 8);
	hipMalloc(&c, 24);
	hipMemset(w, 0, 8);
	rcp_error<<<1024, 256>>>(w, 4000);
	double hw;
	hipMemcpy(&hw, w, 8, hipMemcpyDeviceToHost);
	printf("v_rcp_f64: worst |1 - b rcp(b)| over 1.05e9 operands = %.3e = 2^%.1f\n", hw, log2(hw));
	const char *names[4] = {"random pixel-like", "integer quotients", "one ulp off an integer quotient", "tiny / huge divisors"};
	for (int kind = 0; kind < 4; kind++)
	{
		hipMemset(c, 0, 24);
		quotient_check<<<1024, 256>>>(c, 2000, kind);
		hipMemcpy(h, c, 24, hipMemcpyDeviceToHost);
		printf("%-34s tested %llu, quick path %llu (%.4f %%), wrong %llu\n", names[kind], h[0], h[1], 100.0 * h[1] / (h[0] ? h[0] : 1), h[2]);
	}
	return 0;
}
