// Memory round-trip latency probe: every wave does K dependent loads (the next address comes from the loaded value) over a
// full-cycle LCG permutation of `nodes` slots spaced `stride` bytes apart.  Reports shader cycles per dependent load for several
// footprints (L2 / MALL / HBM + TLB reach) and numbers of concurrent waves.
//   hipcc --offload-arch=gfx950 -O3 tools/lat_probe.hip -o tools/lat_probe.bin && ./tools/lat_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void fill(uint32_t *buf, uint32_t nodes, uint32_t stride_words)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nodes)
		buf[(size_t)i * stride_words] = (i * 1664525u + 1013904223u) & (nodes - 1); // full period for power-of-two `nodes`
}

__global__ void chase(const uint32_t *next, uint32_t nodes, uint32_t stride_words, int K, unsigned long long *cycles, uint32_t *sink, int lanes_words)
{
	const int lane = threadIdx.x & 63;
	uint32_t at = (blockIdx.x * 2654435761u) & (nodes - 1);
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (int k = 0; k < K; k++)
	{
		uint32_t v = next[(size_t)at * stride_words + (lanes_words ? lane % lanes_words : 0)];
		at = __builtin_amdgcn_readfirstlane(v);
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	if (lane == 0)
	{
		cycles[blockIdx.x] = t1 - t0;
		sink[blockIdx.x] = at;
	}
}

int main()
{
	const uint32_t stride_words = 64 + 1024; // 4352 B: consecutive nodes walk over the channels
	for (uint32_t nodes : {1u << 12, 1u << 15, 1u << 18, 1u << 20})
	{
		uint32_t *dev;
		const size_t bytes = (size_t)nodes * stride_words * 4;
		if (hipMalloc(&dev, bytes) != hipSuccess)
			return 1;
		hipLaunchKernelGGL(fill, dim3((nodes + 255) / 256), dim3(256), 0, 0, dev, nodes, stride_words);
		for (int waves : {64, 1024, 4096, 16384})
		{
			const int K = 32;
			unsigned long long *cyc;
			uint32_t *sink;
			(void)hipMalloc(&cyc, waves * 8);
			(void)hipMalloc(&sink, waves * 4);
			for (int rep = 0; rep < 2; rep++)
				hipLaunchKernelGGL(chase, dim3(waves), dim3(64), 0, 0, dev, nodes, stride_words, K, cyc, sink, 1);
			(void)hipDeviceSynchronize();
			std::vector<unsigned long long> h(waves);
			(void)hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
			double mean = 0;
			for (auto c : h)
				mean += (double)c;
			mean /= waves * (double)K;
			printf("footprint %7.1f MiB (%8u lines)  waves %6d : %8.0f cycles per dependent load\n", bytes / 1048576.0, nodes, waves, mean);
			(void)hipFree(cyc);
			(void)hipFree(sink);
		}
		(void)hipFree(dev);
	}
	return 0;
}
