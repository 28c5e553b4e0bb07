// probe of the row-segmented DPP scan used by raster_bwd_fast_kernel (tools only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ double dpp_d(double v)
{
	return __hiloint2double(dpp_i<CTRL>(__double2hiint(v)), dpp_i<CTRL>(__double2loint(v)));
}
#define NMOM 12
__global__ void probe(const int *owner_in, const double *val, double *out, int *tail_out)
{
	int lane = threadIdx.x;
	double mom[NMOM];
	for (int i = 0; i < NMOM; i++)
		mom[i] = val[lane] * (i + 1);
	const int lx = lane & 7;
	const int oid = owner_in[lane];
	const int left_oid = dpp_i<0x111>(oid); // evaluated by ALL lanes: a DPP move under a divergent branch reads 0 from disabled lanes
	const bool head = (lx == 0) | (left_oid != oid);
	int f = head ? 1 : 0;
#define DR_SEG_STEP(CTRL)                                                                                                    \
	{                                                                                                                        \
		const int tf = dpp_i<CTRL>(f);                                                                                       \
		double t[NMOM];                                                                                                      \
		_Pragma("unroll") for (int i = 0; i < NMOM; i++) t[i] = dpp_d<CTRL>(mom[i]);                                         \
		_Pragma("unroll") for (int i = 0; i < NMOM; i++) mom[i] += f ? 0.0 : t[i];                                           \
		f = f ? f : tf;                                                                                                      \
	}
	DR_SEG_STEP(0x111)
	DR_SEG_STEP(0x112)
	DR_SEG_STEP(0x114)
	const int right_head = dpp_i<0x101>(head ? 1 : 0);
	const bool tail = (lx == 7) | (right_head != 0);
	tail_out[lane] = tail;
	for (int i = 0; i < NMOM; i++)
		out[lane * NMOM + i] = mom[i];
}
int main()
{
	int ho[64], ht[64];
	double hv[64], hout[64 * NMOM];
	srand(1);
	for (int i = 0; i < 64; i++)
	{
		ho[i] = (i & 7) == 0 ? rand() % 3 : (rand() % 3 ? ho[i - 1] : rand() % 3);
		hv[i] = 1 + i;
	}
	int *downer, *dt;
	double *dv, *dout;
	(void)hipMalloc(&downer, sizeof ho);
	(void)hipMalloc(&dt, sizeof ht);
	(void)hipMalloc(&dv, sizeof hv);
	(void)hipMalloc(&dout, sizeof hout);
	(void)hipMemcpy(downer, ho, sizeof ho, hipMemcpyHostToDevice);
	(void)hipMemcpy(dv, hv, sizeof hv, hipMemcpyHostToDevice);
	probe<<<1, 64>>>(downer, dv, dout, dt);
	(void)hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
	(void)hipMemcpy(ht, dt, sizeof ht, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int row = 0; row < 8; row++)
	{
		double run = 0;
		for (int x = 0; x < 8; x++)
		{
			int i = row * 8 + x;
			if (x == 0 || ho[i] != ho[i - 1])
				run = 0;
			run += hv[i];
			int tail = x == 7 || ho[i + 1] != ho[i];
			if (hout[i * NMOM] != run || hout[i * NMOM + 11] != run * 12 || ht[i] != tail)
			{
				bad++;
				if (bad < 12)
					printf("lane %d owner %d: got %g (x12 %g) tail %d, want %g tail %d\n", i, ho[i], hout[i * NMOM], hout[i * NMOM + 11], ht[i], run, tail);
			}
		}
	}
	printf("bad = %d\n", bad);
	for (int i = 0; i < 16; i++)
		printf("%d:%d:%g ", ho[i], ht[i], hout[i * NMOM]);
	printf("\n");
	return 0;
}
