// store_width_probe.hip — what a wave-instruction of stores costs by width: the same 92 MB written as 2 bytes per lane (64 lanes,
// 128 contiguous bytes per instruction: what k_deemph_spec_arb's resampling loop does), 4 bytes per lane on every second lane,
// 16 bytes per lane on every eighth lane (the same 128 bytes per instruction), and 16 bytes per lane on all lanes (1 KiB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void __launch_bounds__(256) k(uint8_t *out, size_t per_wave, int iters)
{
	const int lane = threadIdx.x & 63;
	const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	uint8_t *p = out + wave * per_wave;
	uint32_t v = lane * 2654435761u;
	if (MODE == 3) {
		for (int i = 0; i < iters / 8; i++) { *reinterpret_cast<uint4 *>(p + (size_t)i * 1024 + lane * 16) = make_uint4(v, v + 1, v + 2, v + 3); v += 7; }
		return;
	}
	for (int i = 0; i < iters; i++) {
		uint8_t *q = p + (size_t)i * 128;
		if (MODE == 0) *reinterpret_cast<uint16_t *>(q + lane * 2) = (uint16_t)v;
		if (MODE == 1) { if (!(lane & 1)) *reinterpret_cast<uint32_t *>(q + lane * 2) = v; }
		if (MODE == 2) { if (!(lane & 7)) *reinterpret_cast<uint4 *>(q + lane * 2) = make_uint4(v, v + 1, v + 2, v + 3); }
		v += 7;
	}
}
template <class F> static float timeit(F f)
{
	for (int i = 0; i < 3; i++) f();
	hipDeviceSynchronize();
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipEventRecord(a, 0);
	for (int i = 0; i < 20; i++) f();
	hipEventRecord(b, 0); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	return ms / 20;
}
int main()
{
	const int waves = 8192, iters = 88;             // 8192 x 88 x 128 B = 92 MB, as config 3's tail writes per step
	const size_t per_wave = (size_t)iters * 128;
	uint8_t *d; hipMalloc(&d, (size_t)waves * per_wave);
	const char *names[] = {"2 B per lane, 64 lanes (global_store_short)", "4 B per lane, 32 lanes", "16 B per lane, 8 lanes", "16 B per lane, 64 lanes (1 KiB per instruction)"};
	float t[4];
	t[0] = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(waves / 4), dim3(256), 0, 0, d, per_wave, iters); });
	t[1] = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(waves / 4), dim3(256), 0, 0, d, per_wave, iters); });
	t[2] = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(waves / 4), dim3(256), 0, 0, d, per_wave, iters); });
	t[3] = timeit([&] { hipLaunchKernelGGL(k<3>, dim3(waves / 4), dim3(256), 0, 0, d, per_wave, iters); });
	for (int m = 0; m < 4; m++) printf("%-52s %8.1f us for 92 MB  (%5.0f GB/s)\n", names[m], t[m] * 1e3, (double)waves * per_wave / t[m] / 1e6);
	return 0;
}
