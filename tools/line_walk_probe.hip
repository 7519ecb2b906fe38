// line_walk_probe.hip — what -M wbfm's tail pays for walking a row per lane (k_deemph_spec_lpr: 2720 samples = 5440 bytes per
// lane, 16 bytes per load, eight loads per 128-byte line, so a wave-instruction touches 64 lines) against the same bytes fetched
// 1 KiB per instruction - eight lanes share a row's 128 bytes - and handed to their lanes through LDS (a transpose per 64
// samples).  WORK dependent multiply-adds per sample stand for the filter and the resampler (13 in the kernel).  Same grid as the
// kernel's busy part at the benchmark's shape: 2048 waves (two per SIMD), 717 MB.
//   hipcc --offload-arch=gfx950 -O3 -o line_walk_probe tools/line_walk_probe.hip && ./line_walk_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int kStageStride = 9;  // uint4 per row in LDS: 144 bytes (the lanes' 16-byte reads fall on different banks)

template <int WORK>
__device__ __forceinline__ void walk(const uint4 (&g)[8], uint32_t &acc)
{
#pragma unroll
	for (int j = 0; j < 8; j++) {
		const uint32_t w[4] = {g[j].x, g[j].y, g[j].z, g[j].w};
#pragma unroll
		for (int i = 0; i < 4; i++) {
			uint32_t x = w[i] & 0xffffu, y = w[i] >> 16;
#pragma unroll
			for (int k = 0; k < WORK; k++) acc = acc * 3u + x;
#pragma unroll
			for (int k = 0; k < WORK; k++) acc = acc * 3u + y;
			if (WORK == 0) acc ^= w[i];
		}
	}
}

template <int MODE, int WORK>
__global__ void __launch_bounds__(256) k(const uint4 *__restrict__ in, uint32_t *__restrict__ out, int lines, size_t row_u4)
{
	__shared__ uint4 stage[4][64 * kStageStride];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const size_t wave = (size_t)blockIdx.x * 4 + wv;
	uint32_t acc = lane;
	if (MODE == 0) {
		const uint4 *p = in + (wave * 64 + lane) * row_u4;
		uint4 cur[8], nxt[8];
#pragma unroll
		for (int j = 0; j < 8; j++) cur[j] = p[j];
		for (int l = 0; l < lines; l++) {
			const uint4 *np = p + (size_t)(l + 1 < lines ? l + 1 : l) * 8;
#pragma unroll
			for (int j = 0; j < 8; j++) nxt[j] = np[j];
			walk<WORK>(cur, acc);
#pragma unroll
			for (int j = 0; j < 8; j++) cur[j] = nxt[j];
		}
	} else {
		// instruction j fetches the current line of rows 8 j .. 8 j + 7: lane l the piece (l & 7) of row 8 j + (l >> 3)
		const uint4 *q = in + (wave * 64 + (lane >> 3)) * row_u4 + (lane & 7);
		uint4 *st = stage[wv];
		uint4 v[8], cur[8];
#pragma unroll
		for (int j = 0; j < 8; j++) v[j] = q[(size_t)8 * j * row_u4];
		for (int l = 0; l < lines; l++) {
#pragma unroll
			for (int j = 0; j < 8; j++) st[(8 * j + (lane >> 3)) * kStageStride + (lane & 7)] = v[j];
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
			for (int j = 0; j < 8; j++) cur[j] = st[lane * kStageStride + j];
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			const size_t ln = (size_t)(l + 1 < lines ? l + 1 : l) * 8;
#pragma unroll
			for (int j = 0; j < 8; j++) v[j] = q[(size_t)8 * j * row_u4 + ln];
			walk<WORK>(cur, acc);
		}
	}
	out[wave * 64 + lane] = acc;
}

template <class F> static float timeit(F f)
{
	for (int i = 0; i < 3; i++) f();
	hipDeviceSynchronize();
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipEventRecord(a, 0);
	for (int i = 0; i < 10; i++) f();
	hipEventRecord(b, 0); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	return ms / 10;
}

template <int WORK> static void run(const uint4 *d, uint32_t *o, int waves, int lines, size_t row_u4, uint32_t *h0, uint32_t *h1)
{
	const double bytes = (double)waves * 64 * lines * 128;
	const float a = timeit([&] { hipLaunchKernelGGL((k<0, WORK>), dim3(waves / 4), dim3(256), 0, 0, d, o, lines, row_u4); });
	hipMemcpy(h0, o, (size_t)waves * 64 * 4, hipMemcpyDeviceToHost);
	const float b = timeit([&] { hipLaunchKernelGGL((k<1, WORK>), dim3(waves / 4), dim3(256), 0, 0, d, o, lines, row_u4); });
	hipMemcpy(h1, o, (size_t)waves * 64 * 4, hipMemcpyDeviceToHost);
	size_t bad = 0;
	for (size_t i = 0; i < (size_t)waves * 64; i++) bad += h0[i] != h1[i];
	printf("%2d multiply-adds per sample: a row per lane %7.1f us (%5.0f GB/s)   through LDS %7.1f us (%5.0f GB/s)   %s\n", WORK, a * 1e3,
	       bytes / a / 1e6, b * 1e3, bytes / b / 1e6, bad ? "RESULTS DIFFER" : "same sums");
}

int main()
{
	const int waves = 2048, lines = 42;             // 2048 x 64 lanes x 42 lines x 128 B = 705 MB
	const size_t row_u4 = (size_t)lines * 8 + 4;    // rows 5440 bytes apart, as chunks of 2720 samples are (16-byte aligned, not line aligned)
	const size_t n_u4 = (size_t)waves * 64 * row_u4 + 64;
	uint4 *d; hipMalloc(&d, n_u4 * 16);
	hipMemset(d, 0x5a, n_u4 * 16);
	{
		// something that is not constant: the index in every word
		uint32_t *h = (uint32_t *)malloc(n_u4 * 16);
		for (size_t i = 0; i < n_u4 * 4; i++) h[i] = (uint32_t)(i * 2654435761u);
		hipMemcpy(d, h, n_u4 * 16, hipMemcpyHostToDevice);
		free(h);
	}
	uint32_t *o; hipMalloc(&o, (size_t)waves * 64 * 4);
	uint32_t *h0 = (uint32_t *)malloc((size_t)waves * 64 * 4), *h1 = (uint32_t *)malloc((size_t)waves * 64 * 4);
	run<0>(d, o, waves, lines, row_u4, h0, h1);
	run<4>(d, o, waves, lines, row_u4, h0, h1);
	run<13>(d, o, waves, lines, row_u4, h0, h1);
	return 0;
}
