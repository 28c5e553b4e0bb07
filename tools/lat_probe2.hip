// Latency of ONE tile-shaped load (64 lanes x float4 = 8 rows of 128 B, rows 16 KiB apart: an 8 x 8 tile of a 1024-wide RGBA
// float frame) issued by many waves at once over a large, cold buffer:
//   (1) cold line + cold TLB entry, (2) another tile in the same 2 MiB region (TLB warm, lines cold), (3) the same lines again.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const float4 *frames, uint32_t ntiles_total, unsigned long long *out, float *sink)
{
	const int lane = threadIdx.x & 63;
	const uint32_t t = (blockIdx.x * 2654435761u) % ntiles_total; // tile index over all frames: 128 x 128 tiles per frame
	const uint32_t frame = t >> 14, ty = (t >> 7) & 127, tx = t & 127;
	const size_t pix = ((size_t)frame << 20) + (size_t)(ty * 8 + (lane >> 3)) * 1024 + tx * 8 + (lane & 7);
	const size_t pix2 = pix ^ 8; // the neighbouring tile in x
	float acc = 0;
	unsigned long long c[4];
	c[0] = __builtin_readcyclecounter();
	float4 a = frames[pix];
	acc += a.x + a.y + a.z + a.w;
	if (acc == 12345.f) acc += 1;
	c[1] = __builtin_readcyclecounter();
	float4 b = frames[pix2];
	acc += b.x + b.y + b.z + b.w;
	if (acc == 12345.f) acc += 1;
	c[2] = __builtin_readcyclecounter();
	float4 d = frames[pix + (__float_as_uint(acc) == 0x12345u ? 1 : 0)];
	acc += d.x + d.y + d.z + d.w;
	if (acc == 12345.f) acc += 1;
	c[3] = __builtin_readcyclecounter();
	if (lane == 0)
	{
		out[blockIdx.x * 3 + 0] = c[1] - c[0];
		out[blockIdx.x * 3 + 1] = c[2] - c[1];
		out[blockIdx.x * 3 + 2] = c[3] - c[2];
		sink[blockIdx.x] = acc;
	}
}

int main()
{
	const int nframes = 128; // 2 GiB
	float4 *frames;
	if (hipMalloc(&frames, (size_t)nframes << 24) != hipSuccess)
		return 1;
	(void)hipMemset(frames, 0, (size_t)nframes << 24);
	for (int waves : {64, 1024, 4096, 16384, 65536})
	{
		unsigned long long *out;
		float *sink;
		(void)hipMalloc(&out, (size_t)waves * 24);
		(void)hipMalloc(&sink, (size_t)waves * 4);
		hipLaunchKernelGGL(probe, dim3(waves), dim3(64), 0, 0, frames, (uint32_t)nframes << 14, out, sink);
		(void)hipDeviceSynchronize();
		std::vector<unsigned long long> h((size_t)waves * 3);
		(void)hipMemcpy(h.data(), out, (size_t)waves * 24, hipMemcpyDeviceToHost);
		double m[3] = {0, 0, 0};
		for (int i = 0; i < waves; i++)
			for (int j = 0; j < 3; j++)
				m[j] += (double)h[(size_t)i * 3 + j] / waves;
		printf("waves %6d : cold %7.0f   neighbour tile (TLB warm) %7.0f   same lines again %7.0f  cycles\n", waves, m[0], m[1], m[2]);
		(void)hipFree(out);
		(void)hipFree(sink);
	}
	return 0;
}
