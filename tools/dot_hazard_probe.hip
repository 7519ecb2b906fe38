// dot_hazard_probe.hip — does gfx950 interlock a VALU read of a DOT result that follows it at once?  (LLVM's hazard
// recogniser keeps "a VALU that is not the same dot opcode" three wait states behind the dot whose result it reads, and
// does not look into inline assembly: the kernels' one-instruction dot wrappers rely on the answer.)
// One wave alone on a SIMD issues back to back; the register is pre-set to a sentinel; 0, 1, 2, 3 s_nop states between.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int NOPS>
__global__ void k(const int *a, const int *b, int *out, int iters)
{
	const int l = threadIdx.x + blockIdx.x * blockDim.x;
	int x = a[l], y = b[l], bad = 0;
	for (int i = 0; i < iters; i++) {
		int d = 0x5a5a5a5a, r;
		if (NOPS == 0) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\tv_add_u32 %1, %0, %0" : "+v"(d), "=v"(r) : "v"(x), "v"(y));
		if (NOPS == 1) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\ts_nop 0\n\tv_add_u32 %1, %0, %0" : "+v"(d), "=v"(r) : "v"(x), "v"(y));
		if (NOPS == 2) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\ts_nop 1\n\tv_add_u32 %1, %0, %0" : "+v"(d), "=v"(r) : "v"(x), "v"(y));
		if (NOPS == 3) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\ts_nop 2\n\tv_add_u32 %1, %0, %0" : "+v"(d), "=v"(r) : "v"(x), "v"(y));
		const int lo = (short)(x & 0xffff) * (short)(y & 0xffff) + (short)(x >> 16) * (short)(y >> 16);
		if (r != 2 * lo) bad++;
		x = x * 1664525 + 1013904223; y = y * 22695477 + 1;
	}
	out[l] = bad;
}
template <int NOPS>
__global__ void k4(const int *a, const int *b, int *out, int iters)
{
	const int l = threadIdx.x + blockIdx.x * blockDim.x;
	int x = a[l], y = b[l], bad = 0;
	for (int i = 0; i < iters; i++) {
		int d = 0x5a5a5a5a, r;
		if (NOPS == 0) asm volatile("v_dot4_i32_i8 %0, %2, %3, 0\n\tv_add_u32 %1, %0, %0" : "+v"(d), "=v"(r) : "v"(x), "v"(y));
		if (NOPS == 1) asm volatile("v_dot4_i32_i8 %0, %2, %3, 0\n\ts_nop 0\n\tv_add_u32 %1, %0, %0" : "+v"(d), "=v"(r) : "v"(x), "v"(y));
		if (NOPS == 3) asm volatile("v_dot4_i32_i8 %0, %2, %3, 0\n\ts_nop 2\n\tv_add_u32 %1, %0, %0" : "+v"(d), "=v"(r) : "v"(x), "v"(y));
		int lo = 0;
		for (int q = 0; q < 4; q++) lo += (int)(signed char)(x >> (8 * q)) * (int)(signed char)(y >> (8 * q));
		if (r != 2 * lo) bad++;
		x = x * 1664525 + 1013904223; y = y * 22695477 + 1;
	}
	out[l] = bad;
}
int main()
{
	const int n = 64 * 1024;
	int *a, *b, *o;
	hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&o, n * 4);
	int *h = new int[n];
	for (int i = 0; i < n; i++) h[i] = i * 2654435761u;
	hipMemcpy(a, h, n * 4, hipMemcpyHostToDevice);
	for (int i = 0; i < n; i++) h[i] = i * 40503u + 12345;
	hipMemcpy(b, h, n * 4, hipMemcpyHostToDevice);
	auto report = [&](const char *what) {
		hipDeviceSynchronize();
		hipMemcpy(h, o, n * 4, hipMemcpyDeviceToHost);
		long long bad = 0;
		for (int i = 0; i < n; i++) bad += h[i];
		printf("%-54s wrong results: %lld\n", what, bad);
	};
	// one wave per workgroup, few workgroups: waves alone on their SIMDs; then the device full
	for (int grid : {4, 1024}) {
		printf("-- %d workgroups of one wave, 20000 iterations each\n", grid);
		hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, a, b, o, 20000); report("v_dot2_i32_i16 -> v_add_u32 at once");
		hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, a, b, o, 20000); report("v_dot2_i32_i16, s_nop 0, v_add_u32");
		hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, a, b, o, 20000); report("v_dot2_i32_i16, s_nop 1, v_add_u32");
		hipLaunchKernelGGL(k<3>, dim3(grid), dim3(64), 0, 0, a, b, o, 20000); report("v_dot2_i32_i16, s_nop 2, v_add_u32");
		hipLaunchKernelGGL(k4<0>, dim3(grid), dim3(64), 0, 0, a, b, o, 20000); report("v_dot4_i32_i8 -> v_add_u32 at once");
		hipLaunchKernelGGL(k4<1>, dim3(grid), dim3(64), 0, 0, a, b, o, 20000); report("v_dot4_i32_i8, s_nop 0, v_add_u32");
		hipLaunchKernelGGL(k4<3>, dim3(grid), dim3(64), 0, 0, a, b, o, 20000); report("v_dot4_i32_i8, s_nop 2, v_add_u32");
	}
	return 0;
}
