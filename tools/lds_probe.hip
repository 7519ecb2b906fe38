// lds_probe.hip — bank-conflict census of the LDS access patterns of the fused kernel's MFMA
// pass-0 engine, one kernel per pattern, to be run under
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace ...
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o build_ablate/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int kIters = 2000;
constexpr int kDwords = 2400;

#define PROBE(name, body)                                                              \
	__global__ void __launch_bounds__(64) name(uint32_t *out)                          \
	{                                                                                  \
		__shared__ __attribute__((aligned(128))) uint32_t lds[kDwords];                \
		const int lane = threadIdx.x;                                                  \
		for (int k = lane; k < kDwords; k += 64) lds[k] = k * 2654435761u;             \
		__builtin_amdgcn_wave_barrier();                                               \
		uint32_t acc = 0;                                                              \
		char *const b = reinterpret_cast<char *>(lds);                                 \
		const int n = lane & 15, q = lane >> 4;                                        \
		(void)n; (void)q; (void)b;                                                     \
		for (int it = 0; it < kIters; it++) { body }                                   \
		if (acc == 0x12345u) out[blockIdx.x * 64 + lane] = acc;                        \
	}

// 1. staging: chunk 64k + lane, ds_write_b128
PROBE(p_stage_write, {
	uint4 *c = reinterpret_cast<uint4 *>(lds + 32);
_Pragma("unroll")
	for (int k = 0; k < 8; k++) c[64 * k + lane] = make_uint4(acc, it, k, lane);
	__builtin_amdgcn_wave_barrier();
	acc += lds[32 + lane];
})
// 2. operand reads: chunk 32s + 2n + q - 1, ds_read_b128
PROBE(p_operand_read, {
	const uint4 *rd = reinterpret_cast<const uint4 *>(lds + 32) + (2 * n + q - 1);
_Pragma("unroll")
	for (int k = 0; k < 16; k++) { uint4 v = rd[32 * k]; acc += v.x ^ v.y ^ v.z ^ v.w; }
	__builtin_amdgcn_wave_barrier();
})
// 3. output stores: swizzled ds_write_b64 (two per instruction, 64*8 B apart)
PROBE(p_y_write, {
	const int yk = ((2 * n + (q >> 1)) & 7) ^ (n >> 3);
	const uint32_t wr0 = 4u * (uint32_t)(32 + 32 * (n >> 2) + ((2 * q) & 3) + 4 * yk);
_Pragma("unroll")
	for (int k = 0; k < 16; k++)
		reinterpret_cast<uint2 *>(b + (wr0 ^ (32u * (uint32_t)(k & 3))))[64 * k] = make_uint2(acc + k, it);
	__builtin_amdgcn_wave_barrier();
	acc += lds[32 + lane];
})
// 3b. the same stores without the swizzle (natural order), for reference
PROBE(p_y_write_plain, {
	uint2 *wr = reinterpret_cast<uint2 *>(lds + 32 + 8 * n + 2 * q);
_Pragma("unroll")
	for (int k = 0; k < 16; k++) wr[64 * k] = make_uint2(acc + k, it);
	__builtin_amdgcn_wave_barrier();
	acc += lds[32 + lane];
})
// 4. read-back: lane's 128-byte run, slots XOR-swizzled, ds_read_b128
PROBE(p_readback, {
	const uint32_t yl0 = 4u * 32u + 128u * lane + 16u * ((lane >> 1) & 7);
_Pragma("unroll")
	for (int k = 0; k < 8; k++) { uint4 v = *reinterpret_cast<const uint4 *>(b + (yl0 ^ (16u * k))); acc += v.x ^ v.y ^ v.z ^ v.w; }
	__builtin_amdgcn_wave_barrier();
})
// 4b. read-back without the swizzle
PROBE(p_readback_plain, {
	const uint4 *yl = reinterpret_cast<const uint4 *>(lds + 32) + 8 * lane;
_Pragma("unroll")
	for (int k = 0; k < 8; k++) { uint4 v = yl[k]; acc += v.x ^ v.y ^ v.z ^ v.w; }
	__builtin_amdgcn_wave_barrier();
})
// 5. hand-off: five dwords to slot lane+1, five from slot lane
PROBE(p_handoff5, {
	uint32_t *tr = lds + 96;
_Pragma("unroll")
	for (int k = 0; k < 5; k++) tr[(lane + 1) * 5 + k] = acc + k;
	__builtin_amdgcn_wave_barrier();
_Pragma("unroll")
	for (int k = 0; k < 5; k++) acc += tr[lane * 5 + k];
	__builtin_amdgcn_wave_barrier();
})
// 6. ring exchange C=4 (P=5's Y3): 4 writes at lane*4+k, 5 reads at lane*4-5+k
PROBE(p_ring4, {
	uint32_t *ring = lds + 2100;
_Pragma("unroll")
	for (int k = 0; k < 4; k++) ring[16 + lane * 4 + k] = acc + k;
	__builtin_amdgcn_wave_barrier();
_Pragma("unroll")
	for (int k = 0; k < 5; k++) acc += ring[16 + lane * 4 - 5 + k];
	__builtin_amdgcn_wave_barrier();
})

int main()
{
	uint32_t *o; CK(hipMalloc(&o, 1 << 20));
	const int waves = 4096;
#define RUN(k) hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, o); CK(hipDeviceSynchronize());
	RUN(p_stage_write) RUN(p_operand_read) RUN(p_y_write) RUN(p_y_write_plain) RUN(p_readback) RUN(p_readback_plain)
	RUN(p_handoff5) RUN(p_ring4)
	printf("done: %d waves x %d iterations per pattern\n", waves, kIters);
	return 0;
}
