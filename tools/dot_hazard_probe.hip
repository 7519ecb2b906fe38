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

// Round 6: other consumers of a v_dot2_i32_i16 result than v_add_u32 - the ones tools/check_dot_hazard.py found
// one and two wait states behind the bare dot2 in round 5's kernels (v_cvt_f64_i32 straight into atan2_q14, v_sub / v_max
// into fast_atan2), a DPP move, a compare, v_mad_i32_i16, an SDWA move - and the write-after-write case (another opcode
// overwriting the dot's destination one state later, as v_cvt_f64_i32 v[34:35], v34 did).
// KIND: 0 cvt_f64_i32, 1 v_cmp + cndmask, 2 v_mov_dpp, 3 v_mad_i32_i16, 4 v_mov_sdwa (low word), 5 overwrite by v_mov then read
template <int KIND, int NOPS>
__global__ void kc(const int *a, const int *b, int *out, int iters)
{
	const int l = threadIdx.x + blockIdx.x * blockDim.x;
	int x = a[l], y = b[l], bad = 0;
	for (int i = 0; i < iters; i++) {
		const int lo = (short)(x & 0xffff) * (short)(y & 0xffff) + (short)(x >> 16) * (short)(y >> 16);
		int d = 0x5a5a5a5a, r = 0;
		double f = 0.0;
#define NOPSTR(n) (n == 0 ? "" : n == 1 ? "s_nop 0\n\t" : n == 2 ? "s_nop 1\n\t" : "s_nop 2\n\t")
		if (KIND == 0) {
			if (NOPS == 0) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\tv_cvt_f64_i32 %1, %0" : "+v"(d), "=v"(f) : "v"(x), "v"(y));
			if (NOPS == 1) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\ts_nop 0\n\tv_cvt_f64_i32 %1, %0" : "+v"(d), "=v"(f) : "v"(x), "v"(y));
			if (NOPS == 3) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\ts_nop 2\n\tv_cvt_f64_i32 %1, %0" : "+v"(d), "=v"(f) : "v"(x), "v"(y));
			if (f != (double)lo) bad++;
		} else if (KIND == 1) {
			if (NOPS == 0) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\tv_cmp_lt_i32 vcc, %0, %4\n\tv_cndmask_b32 %1, 0, 1, vcc" : "+v"(d), "=v"(r) : "v"(x), "v"(y), "v"(0) : "vcc");
			if (NOPS == 1) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\ts_nop 0\n\tv_cmp_lt_i32 vcc, %0, %4\n\tv_cndmask_b32 %1, 0, 1, vcc" : "+v"(d), "=v"(r) : "v"(x), "v"(y), "v"(0) : "vcc");
			if (NOPS == 3) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\ts_nop 2\n\tv_cmp_lt_i32 vcc, %0, %4\n\tv_cndmask_b32 %1, 0, 1, vcc" : "+v"(d), "=v"(r) : "v"(x), "v"(y), "v"(0) : "vcc");
			if (r != (lo < 0 ? 1 : 0)) bad++;
		} else if (KIND == 2) {
			// row_shr:0 is not encodable; quad_perm:[0,1,2,3] is the identity
			if (NOPS == 0) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\tv_mov_b32_dpp %1, %0 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf" : "+v"(d), "=v"(r) : "v"(x), "v"(y));
			if (NOPS == 1) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\ts_nop 0\n\tv_mov_b32_dpp %1, %0 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf" : "+v"(d), "=v"(r) : "v"(x), "v"(y));
			if (NOPS == 3) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\ts_nop 2\n\tv_mov_b32_dpp %1, %0 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf" : "+v"(d), "=v"(r) : "v"(x), "v"(y));
			if (r != lo) bad++;
		} else if (KIND == 3) {
			if (NOPS == 0) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\tv_mad_i32_i16 %1, %0, %4, %0" : "+v"(d), "=v"(r) : "v"(x), "v"(y), "v"(3));
			if (NOPS == 1) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\ts_nop 0\n\tv_mad_i32_i16 %1, %0, %4, %0" : "+v"(d), "=v"(r) : "v"(x), "v"(y), "v"(3));
			if (NOPS == 3) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\ts_nop 2\n\tv_mad_i32_i16 %1, %0, %4, %0" : "+v"(d), "=v"(r) : "v"(x), "v"(y), "v"(3));
			if (r != (int)(short)(lo & 0xffff) * 3 + lo) bad++;
		} else if (KIND == 4) {
			if (NOPS == 0) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\tv_mov_b32_sdwa %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "+v"(d), "=v"(r) : "v"(x), "v"(y));
			if (NOPS == 1) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\ts_nop 0\n\tv_mov_b32_sdwa %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "+v"(d), "=v"(r) : "v"(x), "v"(y));
			if (NOPS == 3) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\ts_nop 2\n\tv_mov_b32_sdwa %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "+v"(d), "=v"(r) : "v"(x), "v"(y));
			if (r != (lo & 0xffff)) bad++;
		} else {
			// write after write: the later v_mov must win
			if (NOPS == 0) asm volatile("v_dot2_i32_i16 %0, %1, %2, 0\n\tv_mov_b32 %0, %3\n\ts_nop 4" : "+v"(d) : "v"(x), "v"(y), "v"(i));
			if (NOPS == 1) asm volatile("v_dot2_i32_i16 %0, %1, %2, 0\n\ts_nop 0\n\tv_mov_b32 %0, %3\n\ts_nop 4" : "+v"(d) : "v"(x), "v"(y), "v"(i));
			if (NOPS == 3) asm volatile("v_dot2_i32_i16 %0, %1, %2, 0\n\ts_nop 2\n\tv_mov_b32 %0, %3\n\ts_nop 4" : "+v"(d) : "v"(x), "v"(y), "v"(i));
			if (d != i) bad++;
		}
		x = x * 1664525 + 1013904223; y = y * 22695477 + 1;
	}
	out[l] = bad;
}
// the same with a 16x16x64 int8 MFMA of another wave's making in flight on the SIMD: four waves per SIMD, half of them
// only issue MFMAs (what the product's pass 0 does beside the discriminator of its neighbours)
template <int NOPS>
__global__ void kc_mfma(const int *a, const int *b, int *out, int iters)
{
	typedef int v4i __attribute__((ext_vector_type(4)));
	const int l = threadIdx.x + blockIdx.x * blockDim.x;
	if ((blockIdx.x & 1) == 0) {
		v4i acc = {0, 0, 0, 0}, A = {a[l], b[l], a[l] ^ 7, b[l] ^ 9};
		for (int i = 0; i < iters; i++) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, A, acc, 0, 0, 0);
		out[l] = (acc.x | acc.y | acc.z | acc.w) == 0x12345 ? 1 : 0;
		return;
	}
	int x = a[l], y = b[l], bad = 0;
	for (int i = 0; i < iters; i++) {
		const int lo = (short)(x & 0xffff) * (short)(y & 0xffff) + (short)(x >> 16) * (short)(y >> 16);
		int d = 0x5a5a5a5a;
		double f = 0.0;
		if (NOPS == 0) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\tv_cvt_f64_i32 %1, %0" : "+v"(d), "=v"(f) : "v"(x), "v"(y));
		if (NOPS == 1) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\ts_nop 0\n\tv_cvt_f64_i32 %1, %0" : "+v"(d), "=v"(f) : "v"(x), "v"(y));
		if (NOPS == 3) asm volatile("v_dot2_i32_i16 %0, %2, %3, 0\n\ts_nop 2\n\tv_cvt_f64_i32 %1, %0" : "+v"(d), "=v"(f) : "v"(x), "v"(y));
		if (f != (double)lo) bad++;
		x = x * 1664525 + 1013904223; y = y * 22695477 + 1;
	}
	out[l] = bad;
}
int main()
{
	const int n = 64 * 8192;
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
	static const char *names[] = {"v_cvt_f64_i32", "v_cmp_lt_i32 + v_cndmask", "v_mov_b32_dpp", "v_mad_i32_i16", "v_mov_b32_sdwa", "overwritten by v_mov_b32"};
	for (int grid : {4, 4096}) {
		printf("-- round 6: v_dot2_i32_i16 and other consumers; %d workgroups of one wave, 20000 iterations each\n", grid);
		char what[96];
#define RUN3(K) \
		hipLaunchKernelGGL((kc<K, 0>), dim3(grid), dim3(64), 0, 0, a, b, o, 20000); snprintf(what, sizeof(what), "v_dot2 -> %s at once", names[K]); report(what); \
		hipLaunchKernelGGL((kc<K, 1>), dim3(grid), dim3(64), 0, 0, a, b, o, 20000); snprintf(what, sizeof(what), "v_dot2, s_nop 0, %s", names[K]); report(what); \
		hipLaunchKernelGGL((kc<K, 3>), dim3(grid), dim3(64), 0, 0, a, b, o, 20000); snprintf(what, sizeof(what), "v_dot2, s_nop 2, %s", names[K]); report(what);
		RUN3(0) RUN3(1) RUN3(2) RUN3(3) RUN3(4) RUN3(5)
	}
	printf("-- v_dot2 -> v_cvt_f64_i32 with every second wave of the SIMD issuing int8 MFMAs (8192 waves, 20000 iterations)\n");
	hipLaunchKernelGGL(kc_mfma<0>, dim3(8192), dim3(64), 0, 0, a, b, o, 20000); report("beside MFMAs: v_dot2 -> v_cvt_f64_i32 at once");
	hipLaunchKernelGGL(kc_mfma<1>, dim3(8192), dim3(64), 0, 0, a, b, o, 20000); report("beside MFMAs: v_dot2, s_nop 0, v_cvt_f64_i32");
	hipLaunchKernelGGL(kc_mfma<3>, dim3(8192), dim3(64), 0, 0, a, b, o, 20000); report("beside MFMAs: v_dot2, s_nop 2, v_cvt_f64_i32");
	return 0;
}
