// bw_probe.hip — read-bandwidth ceilings of candidate access patterns for the
// fused kernel's mapping (one wave walks a contiguous segment tile by tile).
// Build: hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o /tmp/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// pattern 0: lane-contiguous 128 B per lane per tile (8 x dwordx4), prefetch 1 tile
// pattern 1: coalesced (lane*16 + k*1024)
template <int PATTERN, int DEPTH>
__global__ void __launch_bounds__(64) k_read(const uint8_t* __restrict__ base, size_t seg_bytes, uint32_t* out)
{
	extern __shared__ uint32_t dyn_lds[];  // only to limit occupancy (size given at launch)
	const int lane = threadIdx.x;
	if (seg_bytes == 1) dyn_lds[lane] = lane;  // never true: keeps the allocation
	const uint8_t* p = base + (size_t)blockIdx.x * seg_bytes;
	const int tiles = (int)(seg_bytes / 8192);
	uint4 acc = make_uint4(0, 0, 0, 0);
	uint4 buf[DEPTH][8];
	auto issue = [&](int t, uint4 (&dst)[8]) {
		const uint8_t* q = p + (size_t)t * 8192;
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const uint4* a = PATTERN == 0 ? reinterpret_cast<const uint4*>(q + lane * 128 + k * 16)
			                              : reinterpret_cast<const uint4*>(q + k * 1024 + lane * 16);
			dst[k] = *a;
		}
	};
#pragma unroll
	for (int d = 0; d < DEPTH - 1; d++) if (d < tiles) issue(d, buf[d]);
	for (int t = 0; t < tiles; t += DEPTH) {
#pragma unroll
		for (int d = 0; d < DEPTH; d++) {
			int tt = t + d;
			if (tt + DEPTH - 1 < tiles) issue(tt + DEPTH - 1, buf[(d + DEPTH - 1) % DEPTH]);
			if (tt < tiles) {
#pragma unroll
				for (int k = 0; k < 8; k++) { acc.x ^= buf[d][k].x; acc.y += buf[d][k].y; acc.z ^= buf[d][k].z; acc.w += buf[d][k].w; }
			}
		}
	}
	uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
	if (r == 0x12345678u) out[blockIdx.x * 64 + lane] = r;
}

// pattern 2: 256-thread blocks, each wave its own segment (4 waves / WG)
template <int PATTERN>
__global__ void __launch_bounds__(256) k_read256(const uint8_t* __restrict__ base, size_t seg_bytes, uint32_t* out)
{
	const int lane = threadIdx.x & 63;
	const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
	const uint8_t* p = base + (size_t)wave * seg_bytes;
	const int tiles = (int)(seg_bytes / 8192);
	uint4 acc = make_uint4(0, 0, 0, 0);
	uint4 cur[8], nxt[8];
	auto issue = [&](int t, uint4 (&dst)[8]) {
		const uint8_t* q = p + (size_t)t * 8192;
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const uint4* a = PATTERN == 0 ? reinterpret_cast<const uint4*>(q + lane * 128 + k * 16)
			                              : reinterpret_cast<const uint4*>(q + k * 1024 + lane * 16);
			dst[k] = *a;
		}
	};
	issue(0, cur);
	for (int t = 0; t < tiles; t++) {
		if (t + 1 < tiles) issue(t + 1, nxt);
#pragma unroll
		for (int k = 0; k < 8; k++) { acc.x ^= cur[k].x; acc.y += cur[k].y; acc.z ^= cur[k].z; acc.w += cur[k].w; }
#pragma unroll
		for (int k = 0; k < 8; k++) cur[k] = nxt[k];
	}
	uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
	if (r == 0x12345678u) out[wave * 64 + lane] = r;
}

template <typename F> float timeit(F f, int reps = 10)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	f(); CK(hipDeviceSynchronize());
	CK(hipEventRecord(a));
	for (int i = 0; i < reps; i++) f();
	CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
	float ms; CK(hipEventElapsedTime(&ms, a, b));
	return ms / reps;
}

__global__ void k_fill(uint32_t* d, size_t n, int mode)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) {
		uint32_t x = (uint32_t)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
		d[i] = mode == 0 ? 0x01010101u : mode == 1 ? x : (0x7f7f7f7fu + (x & 0x03030303u));  // 2: IQ-like small noise around 127
	}
}

int main(int argc, char** argv)
{
	size_t total = (size_t)4 << 30;
	uint8_t* d; uint32_t* o;
	CK(hipMalloc(&d, total)); CK(hipMalloc(&o, 1 << 24));
	const char* names[3] = {"constant 0x01", "random bytes", "127 +- small noise"};
	for (int mode = 0; mode < 3; mode++) {
		hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, (uint32_t*)d, total / 4, mode);
		CK(hipDeviceSynchronize());
		for (int lds_kb : {0, 5, 10, 13}) {
			for (int waves : {4096, 8192, 16384}) {
				size_t seg = total / waves;
				float t0 = timeit([&] { hipLaunchKernelGGL((k_read<0, 2>), dim3(waves), dim3(64), lds_kb * 1024, 0, d, seg, o); }, 20);
				float t1 = timeit([&] { hipLaunchKernelGGL((k_read<1, 2>), dim3(waves), dim3(64), lds_kb * 1024, 0, d, seg, o); }, 20);
				float t3 = timeit([&] { hipLaunchKernelGGL((k_read<1, 3>), dim3(waves), dim3(64), lds_kb * 1024, 0, d, seg, o); }, 20);
				printf("%-20s lds %2d KiB/wave waves %6d | lane-contig d2 %7.1f GB/s  coalesced d2 %7.1f  coalesced d3 %7.1f\n",
				       names[mode], lds_kb, waves, total / t0 / 1e6, total / t1 / 1e6, total / t3 / 1e6);
			}
		}
	}
	return 0;
}
