// probe of DPP lane-move semantics on gfx950 (tools only; not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL> __device__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__global__ void probe(int *out)
{
	int lane = threadIdx.x;
	out[lane] = dpp_i<0x111>(lane + 100);
	out[64 + lane] = dpp_i<0x101>(lane + 100);
	out[128 + lane] = dpp_i<0x118>(lane + 100);
	out[192 + lane] = dpp_i<0x114>(lane + 100);
}
int main()
{
	int *d, h[256];
	hipMalloc(&d, sizeof h);
	probe<<<1, 64>>>(d);
	hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
	const char *names[4] = {"row_shr:1", "row_shl:1", "row_shr:8", "row_shr:4"};
	for (int k = 0; k < 4; k++)
	{
		printf("%s:", names[k]);
		for (int i = 0; i < 20; i++)
			printf(" %d", h[64 * k + i]);
		printf("\n");
	}
	return 0;
}
