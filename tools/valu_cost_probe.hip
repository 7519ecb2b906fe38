// valu_cost_probe.hip — what one wave-instruction of the kinds the front ends are made of costs a gfx950 SIMD to issue.
// Per opcode: a loop of 8 x 16 independent instructions (eight accumulator chains), W waves per SIMD on every SIMD of
// the device, every wave stamping s_memtime before and after.  Reported: shader cycles per instruction as ONE wave sees
// them (W = 1: the issue cost of a lone stream) and per SIMD with W = 2 / 4 waves sharing it (wave cycles / W).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_cost_probe tools/valu_cost_probe.hip && /tmp/valu_cost_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

// OPS: eight independent chains a0..a7 (32-bit) or d0..d3 (64-bit pairs)
#define K32(name, body)                                                                                             \
	__global__ void __launch_bounds__(256) name(uint64_t *stamps, int iters, int seed)                              \
	{                                                                                                               \
		int a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
		int b = seed | 1, c = seed * 31 + 7;                                                                        \
		const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                           \
		for (int i = 0; i < iters; i++) {                                                                           \
			REP16(asm volatile(body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) \
		}                                                                                                           \
		const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                           \
		if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345678) stamps[0] = 1;                                   \
		if ((threadIdx.x & 63) == 0) stamps[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                     \
	}

#define E8(op) op(%0) op(%1) op(%2) op(%3) op(%4) op(%5) op(%6) op(%7)

#define OP_ADD(r) "v_add_u32 " #r ", " #r ", %8\n\t"
K32(k_add_u32, E8(OP_ADD))
#define OP_DOT4(r) "v_dot4_i32_i8 " #r ", %8, %9, " #r "\n\t"
K32(k_dot4, E8(OP_DOT4))
#define OP_DOT2(r) "v_dot2_i32_i16 " #r ", %8, %9, " #r "\n\t"
K32(k_dot2, E8(OP_DOT2))
#define OP_PERM(r) "v_perm_b32 " #r ", " #r ", %8, %9\n\t"
K32(k_perm, E8(OP_PERM))
#define OP_PKADD(r) "v_pk_add_u16 " #r ", " #r ", %8\n\t"
K32(k_pk_add_u16, E8(OP_PKADD))
#define OP_PKMUL(r) "v_pk_mul_lo_u16 " #r ", " #r ", %8\n\t"
K32(k_pk_mul_lo_u16, E8(OP_PKMUL))
#define OP_PKMAD(r) "v_pk_mad_i16 " #r ", " #r ", %8, %9\n\t"
K32(k_pk_mad_i16, E8(OP_PKMAD))
#define OP_MADI16(r) "v_mad_i32_i16 " #r ", " #r ", %8, %9\n\t"
K32(k_mad_i32_i16, E8(OP_MADI16))
#define OP_MADI24(r) "v_mad_i32_i24 " #r ", " #r ", %8, %9\n\t"
K32(k_mad_i32_i24, E8(OP_MADI24))
#define OP_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", %8\n\t"
K32(k_mul_lo_u32, E8(OP_MULLO))
#define OP_MULHI(r) "v_mul_hi_u32 " #r ", " #r ", %8\n\t"
K32(k_mul_hi_u32, E8(OP_MULHI))
#define OP_LSHLADD(r) "v_lshl_add_u32 " #r ", " #r ", 2, %8\n\t"
K32(k_lshl_add_u32, E8(OP_LSHLADD))
#define OP_ALIGNBIT(r) "v_alignbit_b32 " #r ", " #r ", %8, 16\n\t"
K32(k_alignbit, E8(OP_ALIGNBIT))
#define OP_BFE(r) "v_bfe_u32 " #r ", " #r ", 5, 3\n\t"
K32(k_bfe_u32, E8(OP_BFE))
#define OP_XOR(r) "v_xor_b32 " #r ", " #r ", %8\n\t"
K32(k_xor, E8(OP_XOR))
#define OP_MAX(r) "v_max_i32 " #r ", " #r ", %8\n\t"
K32(k_max_i32, E8(OP_MAX))
#define OP_ADD3(r) "v_add3_u32 " #r ", " #r ", %8, %9\n\t"
K32(k_add3_u32, E8(OP_ADD3))
#define OP_DPP(r) "v_add_u32_dpp " #r ", " #r ", " #r " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
K32(k_add_dpp_row_shr, E8(OP_DPP))
#define OP_DPPW(r) "v_mov_b32_dpp " #r ", " #r " wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
K32(k_mov_dpp_wave_shr, E8(OP_DPPW))
#define OP_MOV(r) "v_mov_b32 " #r ", %8\n\t"
K32(k_mov, E8(OP_MOV))
#define OP_CMPCND(r) "v_cmp_gt_i32 vcc, " #r ", %8\n\tv_cndmask_b32 " #r ", " #r ", %9, vcc\n\t"
K32(k_cmp_cndmask_pair, E8(OP_CMPCND))
#define OP_RCPF(r) "v_rcp_f32 " #r ", " #r "\n\t"
K32(k_rcp_f32, E8(OP_RCPF))
#define OP_CVTF(r) "v_cvt_f32_u32 " #r ", " #r "\n\t"
K32(k_cvt_f32_u32, E8(OP_CVTF))
#define OP_FMAF(r) "v_fma_f32 " #r ", " #r ", %8, %9\n\t"
K32(k_fma_f32, E8(OP_FMAF))
#define OP_SNOP(r) "s_nop 0\n\t"
K32(k_s_nop0, E8(OP_SNOP))
#define OP_MFMA(r) ""
// 64-bit: four chains in register pairs
#define K64(name, body)                                                                                             \
	__global__ void __launch_bounds__(256) name(uint64_t *stamps, int iters, int seed)                              \
	{                                                                                                               \
		double d0 = 1.0 + seed + threadIdx.x, d1 = d0 * 1.5, d2 = d0 * 2.5, d3 = d0 * 3.5, d4 = d0 * 4.5, d5 = d0 * 5.5, d6 = d0 * 6.5, d7 = d0 * 7.5; \
		double b = 1.0000001, c = 1e-9;                                                                             \
		const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                           \
		for (int i = 0; i < iters; i++) {                                                                           \
			REP16(asm volatile(body : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(b), "v"(c));) \
		}                                                                                                           \
		const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                           \
		if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == 0.12345678) stamps[0] = 1;                                     \
		if ((threadIdx.x & 63) == 0) stamps[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                     \
	}
#define OP_FMA64(r) "v_fma_f64 " #r ", " #r ", %8, %9\n\t"
K64(k_fma_f64, E8(OP_FMA64))
#define OP_MUL64(r) "v_mul_f64 " #r ", " #r ", %8\n\t"
K64(k_mul_f64, E8(OP_MUL64))
#define OP_ADD64(r) "v_add_f64 " #r ", " #r ", %9\n\t"
K64(k_add_f64, E8(OP_ADD64))
#define OP_RCP64(r) "v_rcp_f64 " #r ", " #r "\n\t"
K64(k_rcp_f64, E8(OP_RCP64))
#define OP_TRUNC64(r) "v_trunc_f64 " #r ", " #r "\n\t"
K64(k_trunc_f64, E8(OP_TRUNC64))
#define OP_CMP64(r) "v_cmp_gt_f64 vcc, " #r ", %8\n\t"
K64(k_cmp_f64, E8(OP_CMP64))
// conversions between the files: 32-bit source in %8 / 64-bit destination, and back
#define OP_CVT64I(r) "v_cvt_f64_i32 " #r ", %10\n\t"
#define OP_CVTI64(r) "v_cvt_i32_f64 %10, " #r "\n\t"
#define K64X(name, body)                                                                                            \
	__global__ void __launch_bounds__(256) name(uint64_t *stamps, int iters, int seed)                              \
	{                                                                                                               \
		double d0 = 1.0 + seed + threadIdx.x, d1 = d0 * 1.5, d2 = d0 * 2.5, d3 = d0 * 3.5, d4 = d0 * 4.5, d5 = d0 * 5.5, d6 = d0 * 6.5, d7 = d0 * 7.5; \
		double b = 1.0000001, c = 1e-9;                                                                             \
		int x = seed + threadIdx.x;                                                                                 \
		const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                           \
		for (int i = 0; i < iters; i++) {                                                                           \
			REP16(asm volatile(body : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(b), "v"(c), "v"(x));) \
		}                                                                                                           \
		const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                           \
		if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == 0.12345678) stamps[0] = 1;                                     \
		if ((threadIdx.x & 63) == 0) stamps[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                     \
	}
K64X(k_cvt_f64_i32, E8(OP_CVT64I))

typedef void (*kern_t)(uint64_t *, int, int);
struct Entry { const char *name; kern_t k; int per_rep; };

int main()
{
	hipDeviceProp_t pr;
	hipGetDeviceProperties(&pr, 0);
	const int cus = pr.multiProcessorCount;
	const Entry tab[] = {
		{"v_add_u32", k_add_u32, 8}, {"v_xor_b32", k_xor, 8}, {"v_mov_b32", k_mov, 8}, {"v_max_i32", k_max_i32, 8}, {"v_add3_u32", k_add3_u32, 8},
		{"v_lshl_add_u32", k_lshl_add_u32, 8}, {"v_bfe_u32", k_bfe_u32, 8}, {"v_alignbit_b32", k_alignbit, 8}, {"v_perm_b32", k_perm, 8},
		{"v_dot4_i32_i8", k_dot4, 8}, {"v_dot2_i32_i16", k_dot2, 8}, {"v_pk_add_u16", k_pk_add_u16, 8}, {"v_pk_mul_lo_u16", k_pk_mul_lo_u16, 8},
		{"v_pk_mad_i16", k_pk_mad_i16, 8}, {"v_mad_i32_i16", k_mad_i32_i16, 8}, {"v_mad_i32_i24", k_mad_i32_i24, 8},
		{"v_mul_lo_u32", k_mul_lo_u32, 8}, {"v_mul_hi_u32", k_mul_hi_u32, 8},
		{"v_add_u32_dpp row_shr", k_add_dpp_row_shr, 8}, {"v_mov_b32_dpp wave_shr", k_mov_dpp_wave_shr, 8},
		{"v_cmp + v_cndmask (pair)", k_cmp_cndmask_pair, 16}, {"s_nop 0", k_s_nop0, 8},
		{"v_fma_f32", k_fma_f32, 8}, {"v_rcp_f32", k_rcp_f32, 8}, {"v_cvt_f32_u32", k_cvt_f32_u32, 8},
		{"v_fma_f64", k_fma_f64, 8}, {"v_mul_f64", k_mul_f64, 8}, {"v_add_f64", k_add_f64, 8}, {"v_rcp_f64", k_rcp_f64, 8},
		{"v_trunc_f64", k_trunc_f64, 8}, {"v_cmp_gt_f64", k_cmp_f64, 8}, {"v_cvt_f64_i32", k_cvt_f64_i32, 8},
	};
	uint64_t *d;
	const int maxblocks = cus * 4;
	hipMalloc(&d, (size_t)(1 + maxblocks * 4) * 8);
	std::vector<uint64_t> h(1 + maxblocks * 4);
	const int iters = 400;
	printf("%d CUs; shader cycles per wave-instruction: as one wave sees it (W = 1), and per SIMD (wave cycles / W) with W waves on every SIMD\n", cus);
	printf("%-28s %8s %8s %8s\n", "instruction", "W=1", "W=2", "W=4");
	for (const Entry &e : tab) {
		double res[3];
		int wi = 0;
		for (int W : {1, 2, 4}) {
			const int blocks = cus * W;  // 256 threads = one wave per SIMD of a CU; W workgroups per CU
			for (int rep = 0; rep < 2; rep++) {
				hipMemset(d, 0, (size_t)(1 + maxblocks * 4) * 8);
				hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d, iters, 12345);
				hipDeviceSynchronize();
			}
			hipMemcpy(h.data(), d, (size_t)(1 + blocks * 4) * 8, hipMemcpyDeviceToHost);
			std::vector<uint64_t> v(h.begin() + 1, h.begin() + 1 + blocks * 4);
			std::sort(v.begin(), v.end());
			const double med = (double)v[v.size() / 2];
			res[wi++] = med / ((double)iters * 16 * e.per_rep) / W;
		}
		printf("%-28s %8.2f %8.2f %8.2f\n", e.name, res[0], res[1], res[2]);
	}
	hipFree(d);
	return 0;
}
