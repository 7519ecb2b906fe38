// bank_probe.hip — which address bits decide whether a read stream and a write stream get in each
// other's way?  (tools/mode_probe*.py: the same front-end launch takes 0.806 or 0.863 ms depending on
// WHICH allocations hold its input and its output - a stable property of the pair.)
//
// One physically contiguous arena (hipDeviceMallocContiguous), the bandwidth-probe skeleton of
// rtlsdr_amd/csrc/bw_probe_kernel.h (8192 waves, 8 KiB tiles, 512 B written per tile), the 4 GiB input
// at arena + a and the 256 MiB output at arena + b: GB/s as a function of (a, b).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Irtlsdr_amd/csrc -Iinclude tools/bank_probe.hip -o tools/bank_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/rtlfm_hip.h"
#include "bw_probe_kernel.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static float run(const uint8_t *in, uint8_t *out, uint32_t *sink, size_t seg, int waves, int reps, bool rd_only)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	auto go = [&] {
		if (rd_only) hipLaunchKernelGGL((rtlfm::bwprobe::k_stream<0>), dim3(waves), dim3(64), rtlfm::bwprobe::kLdsBytes, 0, in, seg, sink, out);
		else hipLaunchKernelGGL((rtlfm::bwprobe::k_stream<8>), dim3(waves), dim3(64), rtlfm::bwprobe::kLdsBytes, 0, in, seg, sink, out);
	};
	go(); go();
	CK(hipEventRecord(a));
	for (int i = 0; i < reps; i++) go();
	CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
	float ms; CK(hipEventElapsedTime(&ms, a, b));
	hipEventDestroy(a); hipEventDestroy(b);
	return ms / reps;
}

int main(int argc, char **argv)
{
	const size_t GiB = (size_t)1 << 30;
	const size_t arena_bytes = 26 * GiB, in_bytes = 4 * GiB;
	setvbuf(stdout, nullptr, _IONBF, 0);
	uint8_t *arena = nullptr;
	hipError_t e = hipExtMallocWithFlags((void **)&arena, arena_bytes, hipDeviceMallocContiguous);
	printf("contiguous arena: %s\n", e == hipSuccess ? "yes" : "no (plain hipMalloc)");
	if (e != hipSuccess) CK(hipMalloc((void **)&arena, arena_bytes));
	CK(hipMemset(arena, 0x5a, arena_bytes)); CK(hipDeviceSynchronize());
	uint32_t *sink; CK(hipMalloc((void **)&sink, 8192 * 256));
	const int waves = 8192;
	const size_t seg = in_bytes / waves;
	printf("arena @%p\n", (void *)arena);
	printf("read only: %.4f ms\n", run(arena, arena + 8 * GiB, sink, seg, waves, 10, true));
	// (1) input at 0, output at 8 GiB + delta
	std::vector<size_t> deltas = {0};
	for (int k = 8; k <= 33; k++) deltas.push_back((size_t)1 << k);
	for (size_t a : {(size_t)0, (size_t)4 * GiB, (size_t)1 << 21, (size_t)1 << 28}) {
		printf("input at +0x%zx, output at 8 GiB + delta:\n", a);
		for (size_t d : deltas) {
			if (8 * GiB + a + d + 300 * ((size_t)1 << 20) > arena_bytes) continue;
			const float ms = run(arena + a, arena + 8 * GiB + a + d, sink, seg, waves, 10, false);
			printf("  delta 2^%-2d %12zu: %.4f ms  %6.0f GB/s\n", d ? (int)__builtin_ctzll(d) : -1, d, ms, (double)in_bytes * (1 + 1.0 / 16) / ms / 1e6);
		}
	}
	// (2) both move together
	printf("input at +x, output at 8 GiB + x:\n");
	for (int k = 12; k <= 32; k += 2) {
		const size_t x = (size_t)1 << k;
		const float ms = run(arena + x, arena + 8 * GiB + x, sink, seg, waves, 10, false);
		printf("  x 2^%-2d: %.4f ms  %6.0f GB/s\n", k, ms, (double)in_bytes * (1 + 1.0 / 16) / ms / 1e6);
	}
	return 0;
}
