// mfma4_probe.hip — operand layout of v_mfma_i32_4x4x4_16B_i8 on gfx950 (16 blocks of 4x4x4):
// which lane supplies which row of A / column of B, and where D[i][c] lands.
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma4_probe.hip -o build_ablate/mfma4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const int *a, const int *b, int *d)
{
	const int l = threadIdx.x;
	v4i c = {0, 0, 0, 0};
	v4i r = __builtin_amdgcn_mfma_i32_4x4x4i8(a[l], b[l], c, 0, 0, 0);
	d[4 * l + 0] = r.x; d[4 * l + 1] = r.y; d[4 * l + 2] = r.z; d[4 * l + 3] = r.w;
}
__global__ void k_time(int *out, int iters)
{
	const int l = threadIdx.x;
	v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
	int a = l * 0x01010101, b = l * 0x01020304 + 7;
	long long t0 = clock64();
	for (int i = 0; i < iters; i++) {
		c0 = __builtin_amdgcn_mfma_i32_4x4x4i8(a, b, c0, 0, 0, 0);
		c1 = __builtin_amdgcn_mfma_i32_4x4x4i8(a, b + 1, c1, 0, 0, 0);
		c2 = __builtin_amdgcn_mfma_i32_4x4x4i8(a, b + 2, c2, 0, 0, 0);
		c3 = __builtin_amdgcn_mfma_i32_4x4x4i8(a, b + 3, c3, 0, 0, 0);
	}
	long long t1 = clock64();
	out[l] = c0.x + c1.y + c2.z + c3.w;
	if (l == 0) { out[64] = (int)(t1 - t0); }
}
int main()
{
	int ha[64], hb[64], hd[256];
	srand(5);
	for (int i = 0; i < 64; i++) { ha[i] = rand() * 77 + rand(); hb[i] = rand() * 91 + rand(); }
	int *a, *b, *d;
	hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024 + 64);
	hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
	hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
	auto by = [](int v, int k) { return (int)(signed char)((unsigned)v >> (8 * k)); };
	// hypothesis: block = l / 4; A row i from lane 4*blk + i (bytes = k); B column c from lane 4*blk + c;
	// lane (blk, c) holds D[i][c] in register i
	int bad = 0;
	for (int l = 0; l < 64; l++) {
		const int blk = l / 4, c = l % 4;
		for (int i = 0; i < 4; i++) {
			int want = 0;
			for (int kk = 0; kk < 4; kk++) want += by(ha[4 * blk + i], kk) * by(hb[4 * blk + c], kk);
			if (want != hd[4 * l + i]) bad++;
		}
	}
	printf("hypothesis D[reg i] of lane (blk,c) = sum_k A[lane (blk,i)][k] * B[lane (blk,c)][k]: %s (%d mismatches)\n", bad ? "NO" : "YES", bad);
	if (bad) {
		// alternative: lane (blk, r) holds row r: D[r][j] in register j
		int bad2 = 0;
		for (int l = 0; l < 64; l++) {
			const int blk = l / 4, r = l % 4;
			for (int j = 0; j < 4; j++) {
				int want = 0;
				for (int kk = 0; kk < 4; kk++) want += by(ha[4 * blk + r], kk) * by(hb[4 * blk + j], kk);
				if (want != hd[4 * l + j]) bad2++;
			}
		}
		printf("alternative (lane holds a row of D): %s (%d)\n", bad2 ? "NO" : "YES", bad2);
	}
	int *o; hipMalloc(&o, 1024);
	const int iters = 10000;
	hipLaunchKernelGGL(k_time, dim3(1), dim3(64), 0, 0, o, iters);
	int ho[65]; hipMemcpy(ho, o, 260, hipMemcpyDeviceToHost);
	printf("4 independent chains, %d iterations: %.2f clock64 ticks per MFMA\n", iters, (double)ho[64] / (4.0 * iters));
	return 0;
}
