// debug_poison.h - RTLFM_POISON=1: nothing the library reads may depend on what fresh memory happens to hold.
// hipMalloc returns zero pages in a fresh process and a workgroup's LDS usually still holds what the kernel before
// left there - in the test suite, the same kernel's own data: code that reads a buffer or an LDS word before writing
// it passes every test and fails in the field (round 4 found an unfilled dummy tile that way, only when four test
// processes shared the GPU).  With RTLFM_POISON=1 in the environment every device allocation of the library is
// filled with 0xA5 before use and every run / scan entry point first launches a kernel that leaves 0xA5 in all of
// every CU's LDS (tests/test_poison_gpu.py runs the parity suites that way).  Off (the default) it costs one branch
// per allocation / launch.  Include AFTER <hip/hip_runtime.h> and before any code that allocates.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>

namespace rtl_debug {

inline bool poison_on()
{
	static const bool on = [] { const char *e = getenv("RTLFM_POISON"); return e && *e && *e != '0'; }();
	return on;
}

inline hipError_t poison_malloc(void **p, size_t n)
{
	const hipError_t e = (hipMalloc)(p, n);  // (the parentheses keep the macro below out of this call)
	if (e == hipSuccess && n && poison_on()) {
		if ((hipMemset)(*p, 0xA5, n) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return hipErrorUnknown;
	}
	return e;
}

// one workgroup per wave slot's share of the LDS: 16 waves x 10 KiB cover a CU's 160 KiB
static __global__ void __launch_bounds__(64) k_poison_lds()
{
	extern __shared__ uint32_t poison_sm[];
	for (int i = threadIdx.x; i < 10 * 1024 / 4; i += 64) poison_sm[i] = 0xA5A5A5A5u;
	__syncthreads();
	// keep the workgroups resident together for a moment so that they spread over every CU instead of reusing one slot
	if (poison_sm[(threadIdx.x * 37) % (10 * 1024 / 4)] != 0xA5A5A5A5u) __builtin_trap();
	__builtin_amdgcn_s_sleep(127);
}
inline void poison_lds(hipStream_t q)
{
	if (!poison_on()) return;
	hipLaunchKernelGGL(k_poison_lds, dim3(256 * 16), dim3(64), 10 * 1024, q);
}

}  // namespace rtl_debug

#define hipMalloc(p, n) ::rtl_debug::poison_malloc((void **)(p), (n))
