// rtlfm_place.hip — where a launch's write stream lives relative to its read stream, and the
// small device-memory helpers of the C ABI: rtlfm_gpu_malloc_apart / _ex, rtlfm_gpu_place_pair, rtlfm_gpu_placement_probe,
// rtlfm_gpu_bw_probe, rtlfm_gpu_malloc / _free / _copy, rtlfm_gpu_device_numa_node.  A translation unit of its own since
// round 6 (it needs nothing of a handle): rtlfm_hip.hip keeps the handle, the planners and the run paths.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#include "../../include/rtlfm_hip.h"
#include "debug_poison.h"
#include "bw_probe_kernel.h"

using namespace rtlfm;

#define HIP_TRY(expr)                                                              \
	do {                                                                           \
		hipError_t e_ = (expr);                                                    \
		if (e_ != hipSuccess) {                                                    \
			fprintf(stderr, "rtlfm_hip: %s -> %s (%s:%d)\n", #expr,                \
			        hipGetErrorString(e_), __FILE__, __LINE__);                    \
			return e_ == hipErrorOutOfMemory ? -ENOMEM : -EIO;                     \
		}                                                                          \
	} while (0)

// ---------------------------------------------- placement: read stream vs write stream ----
//
// MI355X's 288 GB of HBM3E behave as four quarters of 72 GB for this purpose: a kernel that streams
// reads from one quarter and writes (even 1/16 of the bytes) into the SAME quarter moves 5.6 TB/s,
// the same kernel writing into another quarter 6.5 TB/s (read only: 6.9) - tools/bank_probe2.hip walks
// the allocator across a boundary: 0.813 -> 0.697 ms for 4 GiB in + 256 MiB out, at the 72 GB mark, for
// any offsets inside an allocation (tools/bank_probe.hip) and stable over time (tools/mode_probe*.py).
// The driver hands out physical memory in order, so two buffers allocated one after the other share a
// quarter unless a boundary happens to fall between them: that was the "box-to-box" spread of rounds 1
// and 2 (0.78-0.80 vs 0.85-0.87 ms for the headline launch).  Nothing in HIP names the quarter, so it
// is found by measurement: the bandwidth-probe skeleton (bw_probe_kernel.h), read only and read + write.

namespace {

struct ProbeRig {
	uint32_t *sink = nullptr;
	hipEvent_t a = nullptr, b = nullptr;
	int init()
	{
		if (hipMalloc(&sink, (size_t)8192 * 256) != hipSuccess) return -ENOMEM;
		if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -EIO;
		return 0;
	}
	~ProbeRig()
	{
		if (a) hipEventDestroy(a);
		if (b) hipEventDestroy(b);
		if (sink) hipFree(sink);
	}
	// ms per launch: `region` bytes of `in` streamed by 8192 waves, W bytes stored per lane and tile into `out`
	// (region / 8192 * 64 * W / 8192 ... = region * W / 128 bytes in all)
	int run(const uint8_t *in, size_t region, uint8_t *out, int W, int reps, float *ms, int warm = 3, bool median = false)
	{
		const int waves = 8192;
		const size_t seg = (region / waves) & ~(size_t)8191;
		if (seg < 8192) return -EINVAL;
		auto go = [&]() {
			switch (W) {
			case 0: hipLaunchKernelGGL((bwprobe::k_stream<0>), dim3(waves), dim3(64), bwprobe::kLdsBytes, 0, in, seg, sink, out); break;
			case 2: hipLaunchKernelGGL((bwprobe::k_stream<2>), dim3(waves), dim3(64), bwprobe::kLdsBytes, 0, in, seg, sink, out); break;
			case 4: hipLaunchKernelGGL((bwprobe::k_stream<4>), dim3(waves), dim3(64), bwprobe::kLdsBytes, 0, in, seg, sink, out); break;
			case 8: hipLaunchKernelGGL((bwprobe::k_stream<8>), dim3(waves), dim3(64), bwprobe::kLdsBytes, 0, in, seg, sink, out); break;
			default: hipLaunchKernelGGL((bwprobe::k_stream<16>), dim3(waves), dim3(64), bwprobe::kLdsBytes, 0, in, seg, sink, out); break;
			}
		};
		for (int i = 0; i < warm; i++) go();  // clocks, TLBs
		if (median) {
			// a decision rests on this number (rtlfm_gpu_malloc_apart_ex): every launch timed by itself, the median taken -
			// one launch that met another tenant's burst or a clock step does not decide
			float t[9];
			if (reps > 9) reps = 9;
			for (int i = 0; i < reps; i++) {
				if (hipEventRecord(a, 0) != hipSuccess) return -EIO;
				go();
				if (hipEventRecord(b, 0) != hipSuccess || hipEventSynchronize(b) != hipSuccess) return -EIO;
				if (hipEventElapsedTime(&t[i], a, b) != hipSuccess) return -EIO;
			}
			std::sort(t, t + reps);
			*ms = t[reps / 2];
			return hipGetLastError() == hipSuccess ? 0 : -EIO;
		}
		if (hipEventRecord(a, 0) != hipSuccess) return -EIO;
		for (int i = 0; i < reps; i++) go();
		if (hipEventRecord(b, 0) != hipSuccess || hipEventSynchronize(b) != hipSuccess) return -EIO;
		if (hipEventElapsedTime(ms, a, b) != hipSuccess) return -EIO;
		*ms /= (float)reps;
		return hipGetLastError() == hipSuccess ? 0 : -EIO;
	}
};

// how much of `in` a placement test streams when the write stream has out_bytes to land in
size_t probe_region(size_t in_bytes, size_t out_bytes)
{
	size_t r = in_bytes;
	if (r > 16 * out_bytes) r = 16 * out_bytes;   // W = 8: one byte written per 16 read
	if (r > ((size_t)2 << 30)) r = (size_t)2 << 30;
	return r & ~(((size_t)8192 * 8192) - 1);      // whole tiles for 8192 waves
}

constexpr float kApartRatio = 1.21f;  // read+write over read-only time: 1.12 apart, 1.31 in the same class
// What a search may hold in temporary allocations unless the caller says otherwise.  Rounds 3-4 walked up to 150 GiB (a
// run of one class can be 64 GB long, profiles/r04_placement_classes.txt) and a first search of a session was seen to take
// 4.7 s and find nothing; round 5 bounds the search instead of the memory it walks: a handful of candidates of
// DIFFERENT SIZES - the driver's allocator serves sizes from different places (268 MiB blocks out of one class of holes,
// 1 GiB ones B A A B B A B ..., 4 GiB ones C C C C C C A A ...: profiles/r04_placement_sizes.txt) -, at most 16 GiB held,
// one short probe each.
constexpr size_t kApartBudgetDefault = (size_t)16 << 30;
// GiB of the candidates after the request's own size, in order; a step that found the placement on this device before
// is tried first the next time (g_recipe)
constexpr int kApartSchedule[] = {1, 2, 1, 4, 1, 2, 4};
constexpr int kApartSteps = (int)(sizeof(kApartSchedule) / sizeof(kApartSchedule[0])) + 1;  // + the request's own size
std::atomic<int> g_recipe[64];  // per device: 1 + the step that won last, 0 = nothing known

}  // namespace

// Device memory for a WRITE stream that is to run next to the read stream of `other` (the output of
// rtlfm_gpu_run_device next to its input): `bytes` in another class of the HBM than `other` (see above).
// Candidates are allocated and timed against `other` with the bandwidth probe (read only once, then read + write per
// candidate: one warm-up launch and the MEDIAN of three timed ones); those that share its class are kept until the search
// ends, so that the allocator moves on, and everything but the winner is freed.  *apart = 1 when a place away from
// `other` was found - the block returned may then be LARGER than `bytes` (the candidate itself: 1, 2 or 4 GiB; the
// handle's "placement_held_mb" says what its placed blocks really hold) -, 0 when the buffers are too small for it to
// matter (< 256 MiB streamed), the search ran out of its budget, or the probe failed: then the memory returned is a plain
// allocation of exactly `bytes` and nothing of the search is kept.  `other` is only read.
// budget_bytes: the most the search may hold at any time, winner included (0 = no search at all).
extern "C" int rtlfm_gpu_malloc_apart_ex(int device, size_t bytes, const void *other, size_t other_bytes, size_t budget_bytes,
                                         void **out, int *apart, double *search_ms, size_t *walked_bytes)
{
	if (!out || !bytes) return -EINVAL;
	if (apart) *apart = 0;
	if (search_ms) *search_ms = 0;
	if (walked_bytes) *walked_bytes = 0;
	const auto t_begin = std::chrono::steady_clock::now();
	struct Clock {  // whatever way the search ends: how long it took, how much it held at its peak
		std::chrono::steady_clock::time_point t0; double *ms; size_t *wb; size_t peak = 0;
		~Clock() {
			if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
			if (wb) *wb = peak;
		}
	} clk{t_begin, search_ms, walked_bytes};
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -ENODEV;
	HIP_TRY(hipSetDevice(device));
	const size_t region = other ? probe_region(other_bytes, bytes) : 0;
	if (region < ((size_t)256 << 20) || budget_bytes == 0) {  // too small to matter, or the search is switched off
		void *p = nullptr;
		HIP_TRY(hipMalloc(&p, bytes));
		*out = p;
		return 0;
	}
	ProbeRig rig;
	int r = rig.init();
	if (r < 0) return r;
	HIP_TRY(hipDeviceSynchronize());
	float rd = 0;
	if ((r = rig.run((const uint8_t *)other, region, nullptr, 0, 3, &rd, 2, true)) < 0) return r;
	size_t free_b = 0, total_b = 0;
	HIP_TRY(hipMemGetInfo(&free_b, &total_b));
	// never more than half of what is free: the device may have other tenants, whose next allocation must not fail
	// because of a search
	size_t budget = budget_bytes;
	if (budget > free_b / 2) budget = free_b / 2;
	std::vector<void *> cand;       // candidates that did not pass, with their times
	std::vector<float> cand_rw;
	void *win = nullptr;
	size_t held = 0;
	// the order of the steps: the one that found the placement on this device last time first
	int order[kApartSteps];
	{
		const int known = device < 64 ? g_recipe[device].load(std::memory_order_relaxed) - 1 : -1;
		int n = 0;
		if (known >= 0 && known < kApartSteps) order[n++] = known;
		for (int k = 0; k < kApartSteps; k++)
			if (k != known) order[n++] = k;
	}
	int won_step = -1;
	for (int t = 0; t < kApartSteps; t++) {
		const int step = order[t];
		size_t cb = bytes;
		if (step > 0) {
			const size_t g = (size_t)kApartSchedule[step - 1] << 30;
			if (g <= bytes) continue;  // the request itself is as large: that size has been tried
			cb = g;
		}
		if (held + cb > budget) continue;
		void *p = nullptr;
		if (hipMalloc(&p, cb) != hipSuccess) { (void)hipGetLastError(); continue; }
		held += cb;
		if (held > clk.peak) clk.peak = held;
		float rw = 0;
		if (rig.run((const uint8_t *)other, region, (uint8_t *)p, 8, 3, &rw, 1, true) < 0) { cand.push_back(p); cand_rw.push_back(1e30f); break; }
		if (rw < kApartRatio * rd) { win = p; won_step = step; break; }
		cand.push_back(p); cand_rw.push_back(rw);
	}
	// A caller that allows more than the schedule can hold (a measurement harness in a process that already holds tens of
	// GB: bench.py's later legs; rtlfm_gpu_bw_probe) gets the walk of rounds 3-4 behind it: 4 GiB candidates, kept, until one
	// is apart or the budget is used up - a run of one class can be 64 GB long.
	while (!win) {
		const size_t cb = bytes > ((size_t)4 << 30) ? bytes : ((size_t)4 << 30);
		if (held + cb > budget) break;
		void *p = nullptr;
		if (hipMalloc(&p, cb) != hipSuccess) { (void)hipGetLastError(); break; }
		held += cb;
		if (held > clk.peak) clk.peak = held;
		float rw = 0;
		if (rig.run((const uint8_t *)other, region, (uint8_t *)p, 8, 3, &rw, 1, true) < 0) { cand.push_back(p); cand_rw.push_back(1e30f); break; }
		if (rw < kApartRatio * rd) { win = p; break; }
		cand.push_back(p); cand_rw.push_back(rw);
	}
	// No candidate under the threshold (every candidate the budget allowed shares the input's class): nothing was found, so
	// nothing of the search is kept - until round 5 the fastest candidate was returned, possibly a 4 GiB block for a request
	// of a few hundred MiB, pinned for the handle's life and buying nothing.  Every candidate is freed and the
	// caller gets a plain allocation of exactly `bytes`, *apart = 0.
	if (win) {
		if (apart) *apart = 1;
		if (device < 64) g_recipe[device].store(won_step + 1, std::memory_order_relaxed);
	}
	for (void *c : cand)
		if (c) hipFree(c);
	if (!win) HIP_TRY(hipMalloc(&win, bytes));
	*out = win;
	return 0;
}

extern "C" int rtlfm_gpu_malloc_apart(int device, size_t bytes, const void *other, size_t other_bytes, void **out, int *apart)
{
	return rtlfm_gpu_malloc_apart_ex(device, bytes, other, other_bytes, kApartBudgetDefault, out, apart, nullptr, nullptr);
}

// The caller owns both sides: choose the pair (include/rtlfm_hip.h).  The ring's own retry (ingest_build) in exported form.
extern "C" int rtlfm_gpu_place_pair(int device, size_t in_bytes, size_t out_bytes, size_t budget_bytes, int max_tries,
                                    void **in, void **out, int *apart, int *tries, double *search_ms, size_t *walked_bytes)
{
	if (!in || !out || !in_bytes || !out_bytes || max_tries < 1 || max_tries > 8) return -EINVAL;
	if (apart) *apart = 0;
	if (tries) *tries = 0;
	if (search_ms) *search_ms = 0;
	if (walked_bytes) *walked_bytes = 0;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -ENODEV;
	HIP_TRY(hipSetDevice(device));
	std::vector<void *> parked;  // inputs that found no partner: held until the end, so that the next one lies elsewhere
	size_t parked_bytes = 0;
	void *cur_in = nullptr, *cur_out = nullptr;
	int rc = 0, found = 0, n = 0;
	double ms_all = 0;
	size_t peak = 0;
	for (; n < max_tries; ) {
		if (hipMalloc(&cur_in, in_bytes) != hipSuccess) { (void)hipGetLastError(); cur_in = nullptr; rc = parked.empty() ? -ENOMEM : 0; break; }
		double ms = 0; size_t walked = 0; int ap = 0;
		rc = rtlfm_gpu_malloc_apart_ex(device, out_bytes, cur_in, in_bytes, budget_bytes, &cur_out, &ap, &ms, &walked);
		n++;
		ms_all += ms;
		if (walked + parked_bytes + in_bytes > peak) peak = walked + parked_bytes + in_bytes;
		if (rc < 0) break;
		if (ap || walked == 0 || n == max_tries) { found = ap; break; }  // walked == 0: no search was made (too small, no budget)
		(void)hipFree(cur_out); cur_out = nullptr;
		parked.push_back(cur_in); parked_bytes += in_bytes;
		cur_in = nullptr;
	}
	if (rc == 0 && !cur_in && !parked.empty()) {
		// no memory left for another input: the last parked one is as good as any
		cur_in = parked.back(); parked.pop_back();
		if (hipMalloc(&cur_out, out_bytes) != hipSuccess) { (void)hipGetLastError(); rc = -ENOMEM; }
	}
	for (void *q : parked) (void)hipFree(q);
	if (rc < 0) {
		if (cur_in) (void)hipFree(cur_in);
		if (cur_out) (void)hipFree(cur_out);
		return rc;
	}
	*in = cur_in; *out = cur_out;
	if (apart) *apart = found;
	if (tries) *tries = n;
	if (search_ms) *search_ms = ms_all;
	if (walked_bytes) *walked_bytes = peak;
	return 0;
}

extern "C" int rtlfm_gpu_copy(int device, void *dst, const void *src, size_t bytes)
{
	if (!dst || !src) return -EINVAL;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -ENODEV;
	HIP_TRY(hipSetDevice(device));
	HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
	HIP_TRY(hipStreamSynchronize(nullptr));  // (a device-to-device hipMemcpy is ordered on the null stream but may return before it has run)
	return 0;
}

// Are two existing buffers a quarter apart?  1 = yes, 0 = no / too small to tell; `in` is read,
// the first in_bytes / 16 bytes of `out` are OVERWRITTEN.
extern "C" int rtlfm_gpu_placement_probe(int device, const void *in, size_t in_bytes, void *out, size_t out_bytes,
                                         double *read_ms, double *rw_ms)
{
	if (!in || !out) return -EINVAL;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -ENODEV;
	HIP_TRY(hipSetDevice(device));
	const size_t region = probe_region(in_bytes, out_bytes);
	if (region < ((size_t)256 << 20)) return 0;
	ProbeRig rig;
	int r = rig.init();
	if (r < 0) return r;
	HIP_TRY(hipDeviceSynchronize());
	float rd = 0, rw = 0;
	if ((r = rig.run((const uint8_t *)in, region, nullptr, 0, 6, &rd)) < 0) return r;
	if ((r = rig.run((const uint8_t *)in, region, (uint8_t *)out, 8, 6, &rw)) < 0) return r;
	if (read_ms) *read_ms = rd;
	if (rw_ms) *rw_ms = rw;
	return rw < kApartRatio * rd ? 1 : 0;
}

// The box's own streaming ceilings, measured with the front end's access pattern and none of its
// arithmetic (bw_probe_kernel.h): read only; read + write with the written bytes a quarter of the HBM
// away from the read ones (what rtlfm_gpu_malloc_apart arranges); read + write inside one allocation.
extern "C" int rtlfm_gpu_bw_probe(int device, size_t bytes, int write_div, int reps, double *read_gbs, double *rw_gbs,
                                  double *rw_colocated_gbs, double *write_fraction)
{
	if (bytes < ((size_t)256 << 20) || reps < 1 || write_div < 1) return -EINVAL;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -ENODEV;
	HIP_TRY(hipSetDevice(device));
	const size_t total = bytes & ~(((size_t)8192 * 8192) - 1);
	// bytes stored per lane and tile: the power of two nearest to 128 / write_div
	int W = 2;
	for (int w : {2, 4, 8, 16})
		if (fabs(128.0 / write_div - w) < fabs(128.0 / write_div - W)) W = w;
	const size_t wbytes = total / 128 * W + 4096;
	uint8_t *d_in = nullptr, *d_near = nullptr;
	void *d_far = nullptr;
	ProbeRig rig;
	int rc = rig.init();
	if (rc < 0) return rc;
	do {
		// input and the co-located output in ONE allocation: the same quarter by construction
		if (hipMalloc(&d_in, total + wbytes) != hipSuccess) { rc = -ENOMEM; break; }
		d_near = d_in + total;
		if (hipMemset(d_in, 0x5a, total) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) { rc = -EIO; break; }
		int apart = 0;
		// (a measurement, not a service path: the walk may be long)
		if ((rc = rtlfm_gpu_malloc_apart_ex(device, wbytes, d_in, total, (size_t)150 << 30, &d_far, &apart, nullptr, nullptr)) < 0) break;
		float ms[3] = {0, 0, 0};
		if ((rc = rig.run(d_in, total, nullptr, 0, reps, &ms[0])) < 0) break;
		if ((rc = rig.run(d_in, total, (uint8_t *)d_far, W, reps, &ms[1])) < 0) break;
		if ((rc = rig.run(d_in, total, d_near, W, reps, &ms[2])) < 0) break;
		const double moved = (double)total * (1.0 + W / 128.0);
		if (read_gbs) *read_gbs = (double)total / (ms[0] * 1e-3) / 1e9;
		if (rw_gbs) *rw_gbs = moved / (ms[1] * 1e-3) / 1e9;
		if (rw_colocated_gbs) *rw_colocated_gbs = moved / (ms[2] * 1e-3) / 1e9;
		if (write_fraction) *write_fraction = W / 128.0;
		rc = apart;  // 1: the "apart" figure really is a quarter away
	} while (0);
	if (d_in) hipFree(d_in);
	if (d_far) hipFree(d_far);
	return rc;
}

// NUMA node of the host the device hangs on (/sys/bus/pci/devices/<bdf>/numa_node), or -1: where the
// threads that feed the device's staging ring should run (host/ingest_bench.cpp pins them there)
extern "C" int rtlfm_gpu_device_numa_node(int device)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -1;
	char bdf[64] = {0};
	if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) return -1;
	for (char *c = bdf; *c; c++) *c = (char)tolower(*c);
	char path[256];
	snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
	FILE *f = fopen(path, "r");
	int node = -1;
	if (f) {
		if (fscanf(f, "%d", &node) != 1) node = -1;
		fclose(f);
	}
	return node;
}

// Plain device memory (hipMalloc) through the library, for callers that have no HIP runtime of their own at hand.
extern "C" int rtlfm_gpu_malloc(int device, size_t bytes, void **out)
{
	if (!out || !bytes) return -EINVAL;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -ENODEV;
	HIP_TRY(hipSetDevice(device));
	void *p = nullptr;
	HIP_TRY(hipMalloc(&p, bytes));
	*out = p;
	return 0;
}

extern "C" int rtlfm_gpu_free(void *p)
{
	if (!p) return 0;
	HIP_TRY(hipFree(p));
	return 0;
}

// the raw stamps of the last stamped launch: per wave {shader clock at start, at end, 100 MHz
// real-time counter at start, at end}
