#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int kIters = 500;
__global__ void __launch_bounds__(64) probe(uint32_t *out, int s1, int c1, int s2, int c2, int wr, int x1)
{
	__shared__ __attribute__((aligned(128))) uint32_t lds[8192];
	const int lane = threadIdx.x;
	for (int k = lane; k < 8192; k += 64) lds[k] = k * 2654435761u;
	__builtin_amdgcn_wave_barrier();
	int u = 2 * lane + (lane >> s1) * c1 + (lane >> s2) * c2;
	u ^= x1 ? ((lane >> x1) & 1) : 0;
	uint32_t acc = 0;
	uint4 *p = reinterpret_cast<uint4 *>(lds) + u;
	if (wr) for (int it = 0; it < kIters; it++) { p[0] = make_uint4(acc, it, 1, lane); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }
	else for (int it = 0; it < kIters; it++) { uint4 v = p[0]; acc += v.x ^ v.y ^ v.z ^ v.w; __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); }
	if (acc == 0x12345u) out[blockIdx.x * 64 + lane] = acc;
}
int main()
{
	uint32_t *o; CK(hipMalloc(&o, 1 << 20));
	int id = 0;
	for (int s1 = 1; s1 <= 5; s1++) for (int c1 = 0; c1 <= 3; c1++) for (int s2 = s1 + 1; s2 <= 6; s2++) for (int c2 = 0; c2 <= 3; c2++) for (int x1 = 0; x1 <= 0; x1++) {
		if (c1 == 0 && s1 > 1) continue;
		if (c2 == 0 && s2 > s1 + 1) continue;
		for (int wr = 0; wr < 2; wr++) {
			hipLaunchKernelGGL(probe, dim3(1024), dim3(64), 0, 0, o, s1, c1, s2, c2, wr, x1);
			printf("D %d s1=%d c1=%d s2=%d c2=%d x1=%d wr=%d\n", ++id, s1, c1, s2, c2, x1, wr);
		}
	}
	CK(hipDeviceSynchronize());
	return 0;
}
