// mfma_probe.hip — operand/result lane maps of v_mfma_i32_16x16x64_i8 on gfx950,
// determined with exact integer data (one-hot A and B).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const int* a, const int* b, int* d)
{
	int l = threadIdx.x;
	v4i A = {a[4*l], a[4*l+1], a[4*l+2], a[4*l+3]};
	v4i B = {b[4*l], b[4*l+1], b[4*l+2], b[4*l+3]};
	v4i C = {0,0,0,0};
	v4i D = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, C, 0, 0, 0);
	d[4*l] = D.x; d[4*l+1] = D.y; d[4*l+2] = D.z; d[4*l+3] = D.w;
}
int main()
{
	// hypothesis: A[row = l&15][k = 16*(l>>4) + byte], B[k = 16*(l>>4) + byte][col = l&15],
	//             D[row = 4*(l>>4) + reg][col = l&15]
	std::vector<signed char> A(16*64), B(64*16);
	for (int r = 0; r < 16; r++) for (int kk = 0; kk < 64; kk++) A[r*64+kk] = (signed char)((r*7 + kk*3) % 11 - 5);
	for (int kk = 0; kk < 64; kk++) for (int c = 0; c < 16; c++) B[kk*16+c] = (signed char)((kk*5 + c*13) % 17 - 8);
	std::vector<int> ha(256), hb(256), hd(256);
	for (int l = 0; l < 64; l++) for (int w = 0; w < 4; w++) {
		unsigned va = 0, vb = 0;
		for (int by = 0; by < 4; by++) {
			int kk = 16*(l>>4) + 4*w + by;
			va |= (unsigned)(unsigned char)A[(l&15)*64 + kk] << (8*by);
			vb |= (unsigned)(unsigned char)B[kk*16 + (l&15)] << (8*by);
		}
		ha[4*l+w] = (int)va; hb[4*l+w] = (int)vb;
	}
	int *da, *db, *dd;
	hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 1024);
	hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
	hipMemcpy(hd.data(), dd, 1024, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int l = 0; l < 64; l++) for (int reg = 0; reg < 4; reg++) {
		int row = 4*(l>>4) + reg, col = l & 15, ref = 0;
		for (int kk = 0; kk < 64; kk++) ref += (int)A[row*64+kk] * (int)B[kk*16+col];
		if (ref != hd[4*l+reg]) { if (bad < 5) printf("mismatch l=%d reg=%d got %d want %d\n", l, reg, hd[4*l+reg], ref); bad++; }
	}
	printf("layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "CONFIRMED", bad);
	return 0;
}
