// bank_probe2.hip — the read+write rate of the bandwidth-probe skeleton for ONE 4 GiB input allocation
// against a long sequence of separate 512 MiB output allocations (all kept alive): which outputs are in
// the input's "class" (slow pairs, tools/mode_probe3.py), and how does that follow the allocation order?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Irtlsdr_amd/csrc -Iinclude tools/bank_probe2.hip -o tools/bank_probe2.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/rtlfm_hip.h"
#include "bw_probe_kernel.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static float run(const uint8_t *in, uint8_t *out, uint32_t *sink, size_t seg, int waves, int reps)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	auto go = [&] { hipLaunchKernelGGL((rtlfm::bwprobe::k_stream<8>), dim3(waves), dim3(64), rtlfm::bwprobe::kLdsBytes, 0, in, seg, sink, out); };
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
	setvbuf(stdout, nullptr, _IONBF, 0);
	const size_t GiB = (size_t)1 << 30, in_bytes = 4 * GiB, piece = GiB / 2;
	const int nout = argc > 1 ? atoi(argv[1]) : 96;
	uint32_t *sink; CK(hipMalloc((void **)&sink, 8192 * 256));
	const int waves = 8192;
	const size_t seg = in_bytes / waves;
	std::vector<uint8_t *> ins, outs;
	for (int i = 0; i < 3; i++) { uint8_t *p; CK(hipMalloc((void **)&p, in_bytes)); CK(hipMemset(p, 0x5a, in_bytes)); ins.push_back(p); }
	for (int i = 0; i < nout; i++) { uint8_t *p; CK(hipMalloc((void **)&p, piece)); outs.push_back(p); }
	CK(hipDeviceSynchronize());
	for (size_t k = 0; k < ins.size(); k++) printf("in%zu @%p\n", k, (void *)ins[k]);
	printf("out#  address          ms with in0 / in1 / in2   (S = slow pair)\n");
	for (int i = 0; i < nout; i++) {
		float t[3];
		for (int k = 0; k < 3; k++) t[k] = run(ins[k], outs[i], sink, seg, waves, 6);
		printf("%3d  %p  %.3f %.3f %.3f   %c%c%c\n", i, (void *)outs[i], t[0], t[1], t[2], t[0] > 0.76 ? 'S' : '.', t[1] > 0.76 ? 'S' : '.', t[2] > 0.76 ? 'S' : '.');
	}
	// inputs as outputs of each other (first 512 MiB of the other input)
	for (int a = 0; a < 3; a++)
		for (int b = 0; b < 3; b++)
			printf("in%d -> writes into in%d: %.3f ms\n", a, b, run(ins[a], ins[b] + (a == b ? 2 * GiB : 0), sink, seg / 2, waves, 6));
	return 0;
}
