// Layout check of v_mfma_f64_16x16x4_f64 (used by the forward raster's owner reduction): D = A (16x4) * B (4x16), asymmetric data.
// A: lane l holds A[l & 15][l >> 4]; B: lane l holds B[l >> 4][l & 15]; D: lane l, register r holds D[(l >> 4) + 4 r][l & 15].
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, double *D)
{
	const int l = threadIdx.x;
	double4_t acc = {0, 0, 0, 0};
	for (int s = 0; s < 4; s++) // K = 16 in four steps of 4
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 16 + 4 * s + (l >> 4)], B[(4 * s + (l >> 4)) * 16 + (l & 15)], acc, 0, 0, 0);
	for (int r = 0; r < 4; r++)
		D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
extern "C" double mfma_probe()
{
	double hA[256], hB[256], hD[256], *dA, *dB, *dD;
	for (int i = 0; i < 256; i++)
		hA[i] = (i * 7 % 13) - 5.5 + 0.01 * i, hB[i] = (i * 5 % 11) * 0.25 - 1 + 0.003 * i * i;
	hipMalloc(&dA, sizeof hA), hipMalloc(&dB, sizeof hB), hipMalloc(&dD, sizeof hD);
	hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice), hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
	hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
	double worst = 0;
	for (int i = 0; i < 16; i++)
		for (int j = 0; j < 16; j++)
		{
			double ref = 0;
			for (int kk = 0; kk < 16; kk++)
				ref += hA[i * 16 + kk] * hB[kk * 16 + j];
			double e = fabs(ref - hD[i * 16 + j]);
			if (e > worst)
				worst = e;
		}
	return worst;
}
