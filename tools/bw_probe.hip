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

// pattern 3: the fused kernel's skeleton around the coalesced stream: a per-tile 4-byte-per-lane
// store (FEAT & 1), an LDS-zeroing prologue (FEAT & 2), one extra warm-up tile and a repeated
// last tile per segment (FEAT & 4)
template <int FEAT>
__global__ void __launch_bounds__(64) k_skel(const uint8_t* __restrict__ base, size_t seg_bytes, uint32_t* out, uint32_t* pcm)
{
	extern __shared__ uint32_t dyn_lds[];
	const int lane = threadIdx.x;
	if (FEAT & 2) { for (int k = lane; k < 2400; k += 64) dyn_lds[k] = 0; }
	const uint8_t* p = base + (size_t)blockIdx.x * seg_bytes;
	const int tiles = (int)(seg_bytes / 8192);
	const int t0 = (FEAT & 4) && blockIdx.x ? -1 : 0;
	uint4 acc = make_uint4(0, 0, 0, 0);
	uint4 cur[8];
	auto issue = [&](int t) {
		const uint8_t* q = p + (ptrdiff_t)t * 8192;
#pragma unroll
		for (int k = 0; k < 8; k++) cur[k] = *reinterpret_cast<const uint4*>(q + k * 1024 + lane * 16);
	};
	issue(t0);
	uint32_t held = 0; uint32_t* held_dst = nullptr;
	for (int t = t0; t < tiles; t++) {
		uint4 x[8];
#pragma unroll
		for (int k = 0; k < 8; k++) x[k] = cur[k];
		if (FEAT & 1) { if (held_dst) *held_dst = held; }
		if (FEAT & 4) issue(t + 1 < tiles ? t + 1 : t); else if (t + 1 < tiles) issue(t + 1);
#pragma unroll
		for (int k = 0; k < 8; k++) { acc.x ^= x[k].x; acc.y += x[k].y; acc.z ^= x[k].z; acc.w += x[k].w; }
		if (FEAT & 1) { held = acc.x ^ acc.y; held_dst = t >= 0 ? pcm + ((size_t)blockIdx.x * tiles + t) * 64 + lane : nullptr; }
	}
	if (FEAT & 1) { if (held_dst) *held_dst = held; }
	uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
	if (r == 0x12345678u) out[blockIdx.x * 64 + lane] = r + ((FEAT & 2) ? dyn_lds[lane] : 0);
}

// store variants, same bytes written: MODE 0 = 4 B/lane every tile, 1 = 16 B/lane every 4th tile,
// 2 = 4 B/lane nontemporal, 3 = 16 B/lane every 4th tile nontemporal, 4 = 16 B/lane x4 every 16th tile
template <int MODE, int LOADNT = 0>
__global__ void __launch_bounds__(64) k_store(const uint8_t* __restrict__ base, size_t seg_bytes, uint32_t* out, uint32_t* pcm)
{
	extern __shared__ uint32_t dyn_lds[];
	const int lane = threadIdx.x;
	if (seg_bytes == 1) dyn_lds[lane] = lane;
	const uint8_t* p = base + (size_t)blockIdx.x * seg_bytes;
	const int tiles = (int)(seg_bytes / 8192);
	uint4 acc = make_uint4(0, 0, 0, 0);
	uint4 cur[8];
	auto issue = [&](int t) {
		const uint8_t* q = p + (ptrdiff_t)t * 8192;
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const uint4* a = reinterpret_cast<const uint4*>(q + k * 1024 + lane * 16);
			if (LOADNT == 0) cur[k] = *a;
			if (LOADNT == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(cur[k]) : "v"(a) : "memory");
			if (LOADNT == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(cur[k]) : "v"(a) : "memory");
			if (LOADNT == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(cur[k]) : "v"(a) : "memory");
		}
		if (LOADNT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // asm loads are not tracked by the compiler (no prefetch here)
	};
	issue(0);
	uint32_t* wbase = pcm + (size_t)(MODE == 6 ? (blockIdx.x & 63) : blockIdx.x) * tiles * (MODE >= 5 ? 128 : 64);
	uint4 hold[4] = {};
	for (int t = 0; t < tiles; t++) {
		uint4 x[8];
#pragma unroll
		for (int k = 0; k < 8; k++) x[k] = cur[k];
		// stores go out before the reload, like the fused kernel's deferred PCM store
		if (MODE == 0 && t > 0) wbase[(t - 1) * 64 + lane] = acc.x;
		if ((MODE == 5 || MODE == 6) && t > 0) reinterpret_cast<uint2*>(wbase)[(size_t)(t - 1) * 64 + lane] = make_uint2(acc.x, acc.y);
		if (MODE >= 7 && MODE <= 9 && t > 0) {
			uint2* q = reinterpret_cast<uint2*>(wbase) + (size_t)(t - 1) * 64 + lane;
			uint2 v = make_uint2(acc.x, acc.y);
			if (MODE == 7) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(q), "v"(v) : "memory");
			if (MODE == 8) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(q), "v"(v) : "memory");
			if (MODE == 9) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" :: "v"(q), "v"(v) : "memory");
		}
		if (MODE == 2 && t > 0) __builtin_nontemporal_store(acc.x, wbase + (t - 1) * 64 + lane);
		if (MODE == 1 && t > 0 && (t & 3) == 0) reinterpret_cast<uint4*>(wbase + (t - 4) * 64)[lane] = acc;
		if (MODE == 3 && t > 0 && (t & 3) == 0) {
			uint32_t* q = wbase + (t - 4) * 64 + lane * 4;
			__builtin_nontemporal_store(acc.x, q); __builtin_nontemporal_store(acc.y, q + 1);
			__builtin_nontemporal_store(acc.z, q + 2); __builtin_nontemporal_store(acc.w, q + 3);
		}
		if (MODE == 4) {
			if ((t & 3) == 0 && t > 0) hold[((t >> 2) - 1) & 3] = acc;
			if (t > 0 && (t & 15) == 0) {
#pragma unroll
				for (int k = 0; k < 4; k++) reinterpret_cast<uint4*>(wbase + (t - 16 + 4 * k) * 64)[lane] = hold[k];
			}
		}
		issue(t + 1 < tiles ? t + 1 : t);
#pragma unroll
		for (int k = 0; k < 8; k++) { acc.x ^= x[k].x; acc.y += x[k].y; acc.z ^= x[k].z; acc.w += x[k].w; }
	}
	uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
	if (r == 0x12345678u) out[blockIdx.x * 64 + lane] = r;
}

// lane-contiguous pattern (lane l reads its own 128 bytes as eight 16-byte loads) with the PCM
// store, and a non-temporal hint on: NT 0 none, 1 all eight, 2 only the first touch of a line,
// 3 only the last touch; NT 4 = coalesced pattern, all nt (reference)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <int NT>
__global__ void __launch_bounds__(64) k_lc(const uint8_t* __restrict__ base, size_t seg_bytes, uint32_t* out, uint32_t* pcm)
{
	extern __shared__ uint32_t dyn_lds[];
	const int lane = threadIdx.x;
	if (seg_bytes == 1) dyn_lds[lane] = lane;
	const uint8_t* p = base + (size_t)blockIdx.x * seg_bytes;
	const int tiles = (int)(seg_bytes / 8192);
	u32x4_t acc = {0, 0, 0, 0};
	u32x4_t cur[8];
	auto issue = [&](int t) {
		const uint8_t* q = p + (ptrdiff_t)t * 8192;
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const u32x4_t* a = reinterpret_cast<const u32x4_t*>(NT == 4 ? q + k * 1024 + lane * 16 : q + lane * 128 + k * 16);
			const bool nt = NT == 1 || NT == 4 || (NT == 2 && k == 0) || (NT == 3 && k == 7);
			cur[k] = nt ? __builtin_nontemporal_load(a) : *a;
		}
	};
	issue(0);
	uint2* wbase = reinterpret_cast<uint2*>(pcm) + (size_t)blockIdx.x * tiles * 64;
	for (int t = 0; t < tiles; t++) {
		u32x4_t x[8];
#pragma unroll
		for (int k = 0; k < 8; k++) x[k] = cur[k];
		if (t > 0) wbase[(size_t)(t - 1) * 64 + lane] = make_uint2(acc.x, acc.y);
		issue(t + 1 < tiles ? t + 1 : t);
#pragma unroll
		for (int k = 0; k < 8; k++) acc ^= x[k];
	}
	uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
	if (r == 0x12345678u) out[blockIdx.x * 64 + lane] = r;
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
	uint8_t* d; uint32_t* o; uint32_t* pcm;
	CK(hipMalloc(&d, total)); CK(hipMalloc(&o, 1 << 24)); CK(hipMalloc(&pcm, total / 16));
	hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, (uint32_t*)d, total / 4, 1);
	CK(hipDeviceSynchronize());
	const int waves = 8192, lds = 9560;
	const size_t seg = total / waves;
	for (int rep = 0; rep < 3; rep++) {
		float t[12];
		t[0] = timeit([&] { hipLaunchKernelGGL((k_lc<0>), dim3(waves), dim3(64), lds, 0, d, seg, o, pcm); }, 20);
		t[1] = timeit([&] { hipLaunchKernelGGL((k_lc<1>), dim3(waves), dim3(64), lds, 0, d, seg, o, pcm); }, 20);
		t[2] = timeit([&] { hipLaunchKernelGGL((k_lc<2>), dim3(waves), dim3(64), lds, 0, d, seg, o, pcm); }, 20);
		t[3] = timeit([&] { hipLaunchKernelGGL((k_lc<3>), dim3(waves), dim3(64), lds, 0, d, seg, o, pcm); }, 20);
		t[4] = timeit([&] { hipLaunchKernelGGL((k_lc<4>), dim3(waves), dim3(64), lds, 0, d, seg, o, pcm); }, 20);
		printf("8B stores + lane-contiguous loads: plain %.4f ms | all nt %.4f | first touch nt %.4f | last touch nt %.4f || coalesced, all nt %.4f\n",
		       t[0], t[1], t[2], t[3], t[4]);
	}
	return 0;
}
