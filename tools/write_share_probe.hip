// write_share_probe.hip — what the front ends' memory pattern can reach, by write share and by the size of a write visit.
// The front ends' skeleton: 8192 waves, each walks its own 512 KiB segment of a 4 GiB input in 8 KiB tiles (coalesced
// non-temporal 16-byte loads, the next tile in flight) and appends `wb` bytes per tile to a row of its own - as the
// kernels do it, `wb` bytes after every tile (T = 1), or T x wb contiguous bytes after every T-th tile.  No arithmetic
// to speak of: the time is the memory system's.  The output is placed with the library's own search (apart) and, for
// comparison, inside the input's allocation (same class).
//   hipcc --offload-arch=gfx950 -O3 -w -Iinclude tools/write_share_probe.hip -Lrtlsdr_amd/csrc -lrtlfm_hip -Wl,-rpath,$PWD/rtlsdr_amd/csrc -o /tmp/wsp && /tmp/wsp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "rtlfm_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void put16(u32x4 *q, u32x4 v)
{
	if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(q), "v"(v) : "memory");
	else *q = v;
}

// T: tiles per write visit.  The bytes wait in LDS (as the PCM does in the kernels) and leave as 16-byte pieces, lane after lane.
template <int T, bool NT = false>
__global__ void __launch_bounds__(64) k_walk(const uint8_t *__restrict__ in, size_t seg_bytes, uint8_t *__restrict__ out, size_t row_bytes, int wb, uint32_t *sink)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];  // T * wb bytes (+ what limits the occupancy to 4 waves per SIMD)
	const int lane = threadIdx.x;
	const uint8_t *p = in + (size_t)blockIdx.x * seg_bytes;
	uint8_t *row = out + (size_t)blockIdx.x * row_bytes;
	const int tiles = (int)(seg_bytes / 8192);
	u32x4 cur[8];
	auto issue = [&](int t) {
		const uint8_t *q = p + (size_t)t * 8192;
#pragma unroll
		for (int k = 0; k < 8; k++) cur[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(q + k * 1024 + lane * 16));
	};
	issue(0);
	uint32_t acc = 0;
	const int pieces = wb / 16;  // per tile
	for (int t = 0; t < tiles; t++) {
		u32x4 x[8];
#pragma unroll
		for (int k = 0; k < 8; k++) x[k] = cur[k];
		issue(t + 1 < tiles ? t + 1 : t);
#pragma unroll
		for (int k = 0; k < 8; k++) acc += x[k].x ^ x[k].y ^ x[k].z ^ x[k].w;
		// the tile's "PCM" into LDS
		const int slot = t % T;
		for (int j = lane; j < pieces; j += 64) reinterpret_cast<u32x4 *>(lds)[slot * pieces + j] = u32x4{acc, acc + 1, acc + 2, acc + 3};
		__builtin_amdgcn_wave_barrier();
		if (slot == T - 1) {
			const int first_tile = t - (T - 1);
			u32x4 *dst = reinterpret_cast<u32x4 *>(row + (size_t)first_tile * wb);
			for (int j = lane; j < T * pieces; j += 64) put16<NT>(dst + j, reinterpret_cast<const u32x4 *>(lds)[j]);
			__builtin_amdgcn_wave_barrier();
		}
	}
	if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

// T = 1 with the store instruction's cache policy: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt, 5 sc0
template <int MODE>
__global__ void __launch_bounds__(64) k_walk_policy(const uint8_t *__restrict__ in, size_t seg_bytes, uint8_t *__restrict__ out, size_t row_bytes, int wb, uint32_t *sink)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	const int lane = threadIdx.x;
	const uint8_t *p = in + (size_t)blockIdx.x * seg_bytes;
	uint8_t *row = out + (size_t)blockIdx.x * row_bytes;
	const int tiles = (int)(seg_bytes / 8192);
	u32x4 cur[8];
	auto issue = [&](int t) {
		const uint8_t *q = p + (size_t)t * 8192;
#pragma unroll
		for (int k = 0; k < 8; k++) cur[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(q + k * 1024 + lane * 16));
	};
	issue(0);
	uint32_t acc = 0;
	const int pieces = wb / 16;
	for (int t = 0; t < tiles; t++) {
		u32x4 x[8];
#pragma unroll
		for (int k = 0; k < 8; k++) x[k] = cur[k];
		issue(t + 1 < tiles ? t + 1 : t);
#pragma unroll
		for (int k = 0; k < 8; k++) acc += x[k].x ^ x[k].y ^ x[k].z ^ x[k].w;
		u32x4 *dst = reinterpret_cast<u32x4 *>(row + (size_t)t * wb);
		for (int j = lane; j < pieces; j += 64) {
			const u32x4 v = u32x4{acc, acc + 1, acc + 2, acc + 3};
			u32x4 *q = dst + j;
			if (MODE == 0) *q = v;
			if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(q), "v"(v) : "memory");
			if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(q), "v"(v) : "memory");
			if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(q), "v"(v) : "memory");
			if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(q), "v"(v) : "memory");
			if (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(q), "v"(v) : "memory");
		}
	}
	if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

// The same bytes, but a visit only writes whole 128-byte lines: what is left of the last line waits (in LDS) for the next tile.
template <bool NT>
__global__ void __launch_bounds__(64) k_walk_lines(const uint8_t *__restrict__ in, size_t seg_bytes, uint8_t *__restrict__ out, size_t row_bytes, int wb, uint32_t *sink, int line)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	const int lane = threadIdx.x;
	const uint8_t *p = in + (size_t)blockIdx.x * seg_bytes;
	uint8_t *row = out + (size_t)blockIdx.x * row_bytes;
	const int tiles = (int)(seg_bytes / 8192);
	u32x4 cur[8];
	auto issue = [&](int t) {
		const uint8_t *q = p + (size_t)t * 8192;
#pragma unroll
		for (int k = 0; k < 8; k++) cur[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(q + k * 1024 + lane * 16));
	};
	issue(0);
	uint32_t acc = 0;
	const int pieces = wb / 16;
	int done = 0;  // bytes of the row already written (a multiple of `line`)
	for (int t = 0; t < tiles; t++) {
		u32x4 x[8];
#pragma unroll
		for (int k = 0; k < 8; k++) x[k] = cur[k];
		issue(t + 1 < tiles ? t + 1 : t);
#pragma unroll
		for (int k = 0; k < 8; k++) acc += x[k].x ^ x[k].y ^ x[k].z ^ x[k].w;
		// LDS mirrors the row from `done` on: the carried piece sits at its start
		const int have = t * wb - done;  // carried bytes, < line
		for (int j = lane; j < pieces; j += 64) reinterpret_cast<u32x4 *>(lds)[have / 16 + j] = u32x4{acc, acc + 1, acc + 2, acc + 3};
		__builtin_amdgcn_wave_barrier();
		const int end = (t + 1) * wb;
		const int upto = t + 1 == tiles ? end : end / line * line;
		const int n16 = (upto - done) / 16;
		u32x4 *dst = reinterpret_cast<u32x4 *>(row + done);
		for (int j = lane; j < n16; j += 64) put16<NT>(dst + j, reinterpret_cast<const u32x4 *>(lds)[j]);
		__builtin_amdgcn_wave_barrier();
		// the rest moves to the front
		const int rest16 = (end - upto) / 16;
		u32x4 keep = u32x4{0, 0, 0, 0};
		if (lane < rest16) keep = reinterpret_cast<const u32x4 *>(lds)[n16 + lane];
		__builtin_amdgcn_wave_barrier();
		if (lane < rest16) reinterpret_cast<u32x4 *>(lds)[lane] = keep;
		__builtin_amdgcn_wave_barrier();
		done = upto;
	}
	if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

template <class F>
static float timeit(F f, int reps)
{
	for (int i = 0; i < 5; i++) f();
	CK(hipDeviceSynchronize());
	hipEvent_t a, b;
	CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	CK(hipEventRecord(a, 0));
	for (int i = 0; i < reps; i++) f();
	CK(hipEventRecord(b, 0));
	CK(hipEventSynchronize(b));
	float ms;
	CK(hipEventElapsedTime(&ms, a, b));
	return ms / reps;
}

__global__ void k_fill(uint32_t *d, size_t n)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	for (; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = (uint32_t)i * 2654435761u;
}

template <int T>
static float run(const uint8_t *d, size_t seg, uint8_t *o, size_t row, int wb, uint32_t *sink, int waves)
{
	const size_t lds = 9560 > (size_t)T * wb ? 9560 : (size_t)T * wb;  // four waves per SIMD, as the front ends have
	return timeit([&] { hipLaunchKernelGGL((k_walk<T>), dim3(waves), dim3(64), lds, 0, d, seg, o, row, wb, sink); }, 30);
}

template <int MODE>
static float run_policy(const uint8_t *d, size_t seg, uint8_t *o, size_t row, int wb, uint32_t *sink, int waves)
{
	return timeit([&] { hipLaunchKernelGGL((k_walk_policy<MODE>), dim3(waves), dim3(64), 9560, 0, d, seg, o, row, wb, sink); }, 30);
}

static float run_lines(const uint8_t *d, size_t seg, uint8_t *o, size_t row, int wb, uint32_t *sink, int waves, int line)
{
	return timeit([&] { hipLaunchKernelGGL(k_walk_lines<false>, dim3(waves), dim3(64), 9560, 0, d, seg, o, row, wb, sink, line); }, 30);
}

int main()
{
	const size_t total = (size_t)4 << 30;
	const int waves = 8192;
	const size_t seg = total / waves;
	const int tiles = (int)(seg / 8192);
	// the input with room for a same-class output behind it
	const size_t out_max = (size_t)waves * tiles * 1376;  // rows of up to 1360 bytes per tile
	uint8_t *d;
	uint32_t *sink;
	CK(hipMalloc(&d, total + out_max));
	CK(hipMalloc(&sink, waves * 4));
	hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, (uint32_t *)d, total / 4);
	CK(hipDeviceSynchronize());
	void *far = nullptr;
	int apart = 0;
	double sms = 0;
	size_t walked = 0;
	if (rtlfm_gpu_malloc_apart_ex(0, out_max, d, total, (size_t)96 << 30, &far, &apart, &sms, &walked) != 0) { printf("placement failed\n"); return 1; }
	printf("output placed apart: %d (%.0f ms, %zu MiB held)\n", apart, sms, walked >> 20);
	printf("%-34s %9s %9s %9s %9s %9s | %9s %9s %9s  (ms per 4 GiB; GB/s of read + written bytes at T = 1 and at the best)\n", "write bytes per 8 KiB tile", "T=1", "T=2", "T=4", "T=8", "T=16", "lines64", "lines128", "lines256");
	const struct { int wb; const char *what; } rows[] = {
		{96, "96 B (boxcar /84)"}, {256, "256 B (/32: five passes)"}, {512, "512 B (/16: the headline)"}, {640, "640 B"}, {768, "768 B"}, {816, "816 B (/10: config 0)"},
		{832, "832 B"}, {896, "896 B"}, {1024, "1024 B (/8)"}, {1280, "1280 B"}, {1360, "1360 B (/6: -M wbfm)"},
	};
	for (int where = 0; where < 2; where++) {
		uint8_t *o = where == 0 ? (uint8_t *)far : d + total;
		printf("-- output %s\n", where == 0 ? "apart (rtlfm_gpu_malloc_apart)" : "in the input's allocation (same class)");
		for (auto &r : rows) {
			const size_t row = (size_t)tiles * r.wb;
			float t[5];
			t[0] = run<1>(d, seg, o, row, r.wb, sink, waves);
			t[1] = run<2>(d, seg, o, row, r.wb, sink, waves);
			t[2] = run<4>(d, seg, o, row, r.wb, sink, waves);
			t[3] = run<8>(d, seg, o, row, r.wb, sink, waves);
			t[4] = run<16>(d, seg, o, row, r.wb, sink, waves);
			float u[3];
			u[0] = run_lines(d, seg, o, row, r.wb, sink, waves, 64);
			u[1] = run_lines(d, seg, o, row, r.wb, sink, waves, 128);
			u[2] = run_lines(d, seg, o, row, r.wb, sink, waves, 256);
			float best = t[0];
			for (float v : t) best = v < best ? v : best;
			for (float v : u) best = v < best ? v : best;
			const double bytes = (double)total + (double)waves * tiles * r.wb;
			printf("%-34s %9.4f %9.4f %9.4f %9.4f %9.4f | %9.4f %9.4f %9.4f  %6.0f -> %6.0f\n", r.what, t[0], t[1], t[2], t[3], t[4], u[0], u[1], u[2], bytes / t[0] / 1e6, bytes / best / 1e6);
		}
	}
	printf("-- the store's cache policy (T = 1, output apart): plain | nt | sc1 | sc0 sc1 | sc0 sc1 nt | sc0\n");
	for (auto &r : rows) {
		if (r.wb != 256 && r.wb != 512 && r.wb != 816 && r.wb != 1360) continue;
		const size_t row = (size_t)tiles * r.wb;
		uint8_t *o = (uint8_t *)far;
		printf("%-34s %9.4f %9.4f %9.4f %9.4f %9.4f %9.4f\n", r.what, run_policy<0>(d, seg, o, row, r.wb, sink, waves), run_policy<1>(d, seg, o, row, r.wb, sink, waves),
		       run_policy<2>(d, seg, o, row, r.wb, sink, waves), run_policy<3>(d, seg, o, row, r.wb, sink, waves), run_policy<4>(d, seg, o, row, r.wb, sink, waves),
		       run_policy<5>(d, seg, o, row, r.wb, sink, waves));
	}
	printf("-- nt stores, output apart: T = 1 | 4 | 16 | whole 64-byte pieces | whole 128-byte lines | whole 256 bytes\n");
	for (auto &r : rows) {
		if (r.wb != 512 && r.wb != 816 && r.wb != 1024 && r.wb != 1360) continue;
		const size_t row = (size_t)tiles * r.wb;
		uint8_t *o = (uint8_t *)far;
		const int wb = r.wb;
		const float a1 = timeit([&] { hipLaunchKernelGGL((k_walk<1, true>), dim3(waves), dim3(64), 9560, 0, d, seg, o, row, wb, sink); }, 30);
		const float a4 = timeit([&] { hipLaunchKernelGGL((k_walk<4, true>), dim3(waves), dim3(64), 9560, 0, d, seg, o, row, wb, sink); }, 30);
		const float a16 = timeit([&] { hipLaunchKernelGGL((k_walk<16, true>), dim3(waves), dim3(64), (size_t)16 * wb > 9560 ? (size_t)16 * wb : 9560, 0, d, seg, o, row, wb, sink); }, 30);
		const float l64 = timeit([&] { hipLaunchKernelGGL(k_walk_lines<true>, dim3(waves), dim3(64), 9560, 0, d, seg, o, row, wb, sink, 64); }, 30);
		const float l128 = timeit([&] { hipLaunchKernelGGL(k_walk_lines<true>, dim3(waves), dim3(64), 9560, 0, d, seg, o, row, wb, sink, 128); }, 30);
		const float l256 = timeit([&] { hipLaunchKernelGGL(k_walk_lines<true>, dim3(waves), dim3(64), 9560, 0, d, seg, o, row, wb, sink, 256); }, 30);
		printf("%-34s %9.4f %9.4f %9.4f %9.4f %9.4f %9.4f\n", r.what, a1, a4, a16, l64, l128, l256);
	}
	// does the output of a launch stay in the 256 MiB Infinity Cache?  the same walk with the rows of TWO launches alternating
	printf("-- two output buffers in turn (what a double-buffered consumer sees), T = 1, plain stores, output apart\n");
	for (auto &r : rows) {
		if (r.wb != 256 && r.wb != 512 && r.wb != 816) continue;
		const size_t row = (size_t)tiles * r.wb;
		uint8_t *o = (uint8_t *)far;
		uint8_t *o2 = o + (size_t)waves * row;
		if ((size_t)2 * waves * row > out_max) { printf("%-34s (no room for two)\n", r.what); continue; }
		int flip = 0;
		const int wb = r.wb;
		const float t2 = timeit([&] { hipLaunchKernelGGL((k_walk<1>), dim3(waves), dim3(64), 9560, 0, d, seg, (flip ^= 1) ? o : o2, row, wb, sink); }, 30);
		const float t3 = timeit([&] { hipLaunchKernelGGL((k_walk<1, true>), dim3(waves), dim3(64), 9560, 0, d, seg, (flip ^= 1) ? o : o2, row, wb, sink); }, 30);
		const float t4 = timeit([&] { hipLaunchKernelGGL((k_walk_policy<2>), dim3(waves), dim3(64), 9560, 0, d, seg, (flip ^= 1) ? o : o2, row, wb, sink); }, 30);
		printf("%-34s plain %9.4f   nt %9.4f   sc1 %9.4f\n", r.what, t2, t3, t4);
	}
	rtlfm_gpu_free(far);
	return 0;
}
