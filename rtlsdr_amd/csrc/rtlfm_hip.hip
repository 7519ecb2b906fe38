// rtlfm_hip.hip — the C ABI of include/rtlfm_hip.h on HIP / gfx950.
//
// Host side: handle, HBM buffers, stage sequencing that reproduces the order
// of full_demod() (reference src/rtl_fm.c:1179-1272) behind the
// rtlsdr_read_async callback boundary (include/rtl-sdr.h:472-492).
// There is deliberately no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>

#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "../../include/rtlfm_hip.h"
#include "debug_poison.h"
#include "stream_pool.h"
#include "staged_kernels.h"
#include "fused_kernel.h"
#include "boxcar_kernel.h"

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

struct rtlfm_gpu {
	rtlfm_cfg cfg;
	int nstreams = 0;
	int device = 0;
	hipStream_t own_stream = nullptr;
	hipStream_t stream = nullptr;
	hipEvent_t ev_wait = nullptr, ev_release = nullptr;  // rtlfm_gpu_wait_for / _release_to
	int path = 0, last_path = 0;

	// geometry
	size_t xstride = 0;   // dwords per stream in a work buffer
	size_t rstride = 0;   // int16 per stream in a result buffer
	int cap_blocks = 0;

	// device memory
	uint32_t *bufA = nullptr, *bufB = nullptr;
	// demodulated samples on their way through the audio tail: [step parity][ping / pong], tstride
	// int16 per stream.  Two sets, because the tail of step k runs on its own stream while the front
	// end of step k + 1 already fills the other set.
	int16_t *res[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
	bool res_one_block = false;  // res[0][0] and res[1][0] are ONE placed allocation (res[1][0] points into it)
	size_t tstride = 0;
	int32_t *d_cnt[2] = {nullptr, nullptr}, *d_cnt2 = nullptr;  // d_cnt: per step parity
	DeemphChunk *d_deemph_tab = nullptr;  // time-parallel deemph (k_deemph_scan_*): [nstreams][deemph_chunks]
	uint32_t *d_deemph_inc = nullptr;
	LprChunk *d_lpr_chunks = nullptr;     // low_pass_real folded into the replay pass: [nstreams][deemph_chunks]
	SlimPlan *d_slim_plan = nullptr;      // k_lpr_slim_plan -> k_deemph_lpr_slim: [nstreams * chunks + 64]
	size_t slim_plan_cap = 0;             // entries
	int slim_magic_for = 0; uint32_t slim_magic = 0; size_t slim_magic_upto = 0;  // g / chunks as one multiply-high, checked up to slim_magic_upto
	int32_t *d_arb_i = nullptr;           // k_deemph_spec_arb: (i, frac) of every output of a buffer, [arb_len2]
	double *d_arb_frac = nullptr;
	ArbTab *d_arb_tab = nullptr;          // k_deemph_arb_span: the same as one 16-byte entry per output
	int arb_len2 = 0, arb_len1 = 0;
	int deemph_chunks = 0;   // capacity of d_deemph_tab / d_deemph_inc, chunks per stream
	int lpr_chunks_cap = 0;  // ... of d_lpr_chunks
	uint32_t *deepA = nullptr, *deepB = nullptr;  // /64 IQ work buffers of the 7..10-pass path
	size_t deep_stride = 0;
	// Carried state, three copies in rotation: step k reads st[cur] and writes st[(cur + 1) % 3].
	// The tail of step k (own stream) still reads st[cur] / writes the tail's fields of the next copy
	// while the front end of step k + 1 reads that copy and writes the third one.
	state_t *st[3] = {nullptr, nullptr, nullptr};
	int st_cur = 0;
	unsigned step = 0;                      // parity selects res[] / d_cnt[] / the events
	hipStream_t tail_stream = nullptr;
	int tail_priority = 0;
	int tail_prio_value = 0;                // the HIP priority tail_stream was taken from the pool with
	hipEvent_t ev_front[2] = {nullptr, nullptr}, ev_tail[2] = {nullptr, nullptr};
	bool tail_pending[2] = {false, false};  // ev_tail[p] has been recorded and not yet waited for
	bool tail_overlap = true;
	bool tail_serial_asked = false;         // option tail_serial: never overlap, whatever the stream
	int32_t *d_lut = nullptr;
	int32_t *d_mute = nullptr;        // [nstreams*cap_blocks]
	int32_t *d_levels = nullptr;      // [nstreams*cap_blocks] rms() per buffer of the last run
	int last_nblocks = 0;
	long long *d_sums = nullptr;      // [nstreams*cap_blocks*2]  dc_block_raw (front end's stream)
	long long *d_adc_sums = nullptr;  // [nstreams*cap_blocks]    dc_block_audio (the tail's stream)
	uint32_t *d_sq_sums = nullptr;    // [nstreams*cap_blocks*2]  rms()'s sums taken by the boxcar front end (SQ kernels)
	int2 *d_rdc_avg = nullptr;        // [nstreams*cap_blocks]
	int32_t *d_adc_avg = nullptr;
	struct Ingest *ing = nullptr;     // the callback side (push / run / fetch), allocated on first use
	bool no_deemph_scan = false;      // stream-range views (ragged runs) keep to the sequential filter
	// placement of the write streams (rtlfm_gpu_malloc_apart_ex): what the searches found and what they cost
	struct Placement {
		int budget_gb = 16;             // option apart_budget_gb: most a search may hold in candidates; 0 = no search
		int force_retry = 0;            // option ring_force_retry (tests): the ring's first search counts as failed
		int ring_tries = 0;             // searches the ring's placement took (2: the device inputs were moved once)
		int ring_apart = -1;            // -1: not allocated yet; 0 / 1: the ring's result buffers (both halves) are a quarter away from d_in
		int res_apart = -1;             // the same for the audio tail's work buffers against the first run's input
		int deep_apart = -1;            // ... and for what a front end's emit mode writes (deepA)
		double search_ms = 0;           // wall time of all searches of this handle
		size_t walked_peak = 0;         // most a search held in temporary allocations (bytes)
	} place;
	fused::Workspace fws;
	// A/B switches (rtlfm_gpu_set_option); none of them changes a result
	struct Options {
		int deemph_sequential = 0, deemph_four_pass = 0, lpr_separate = 0, lpr_scalar_stores = 0, tail_sync = 0;
		int lpr_chunk = 5440;  // samples per lane of the one-pass deemph + low_pass_real kernel (round 5: 2720 -> 5440 with the outputs leaving through LDS)
		int deep_rest = 1;     // 0: the passes beyond six, generic_fir and the demodulator as a launch each (round 4) instead of k_deep_rest
		int adc_separate = 0;  // 1: dc_block_audio_filter as three kernels (sums, smoothing, subtraction) instead of two
		int squelch_fused = 1; // 0: the squelch / -L behind the boxcar through the emit mode and k_squelch_rms / _hits / _zero / k_fm_demod (round 4) also where the front end can take rms()'s sums itself
		int lpr_ring = 1;      // 0: the one-pass deemph + low_pass_real kernel's outputs leave in 16-byte groups from registers (round 3) instead of 64-byte pieces from LDS
		int arb_span = 0;      // 1: k_deemph_arb_span instead of k_deemph_spec_arb for config 3's tail (18 % fewer instructions, the same time: LAB.md)
		int arb_chunk = 32;    // samples per lane of k_deemph_arb_span: 32 or 64
		int arb_serial = -1;   // deemph + arbitrary_upsample behind the front end on ITS stream: -1 = from 2048 streams on, 0 never, 1 always
		int lpr_threads = 256; // lanes per workgroup of k_deemph_spec_lpr: 64, 128, 192 or 256
		int arb_waves = 0;     // waves per stream of k_deemph_spec_arb: 0 = about 16384 waves in all, else 1 .. 8
		int lpr_slim = 0;      // 1: -M wbfm's tail as k_lpr_slim_plan + k_deemph_lpr_slim - 32 registers, no LDS, one-wave workgroups: a fifth wave beside the next step's four front-end waves per SIMD instead of in place of one (round 6: built, bit-exact, and no faster - LAB.md I.22); 0: k_deemph_spec_lpr
		int lpr_slim_prio = 3;      // s_setprio of that kernel's waves (0 .. 3)
		int lpr_slim_chunk = 6120;  // samples per lane of that kernel: 16 chunks per stream at the wbfm shape = 1024 waves, one per SIMD
		int verify_inject = 0; // tests: the shadow execution's first sample of stream 0 is overwritten before the comparison (it must be noticed)
		int verify_twice = 0;  // debug: every run_device runs twice - into a shadow output, then into the caller's - and the two are compared on the device
	} opt;
	// verify_twice (round 6): shadow rows / lengths / state, and what the comparisons found so far
	int16_t *vt_out = nullptr;
	size_t vt_out_cap = 0;        // int16 elements
	int32_t *vt_len = nullptr, *vt_len2 = nullptr;
	state_t *vt_state = nullptr;
	unsigned long long *vt_cnt = nullptr;  // [4]: differing PCM dwords, differing lengths, differing state dwords, first differing (stream << 32 | index) + 1
	long vt_mismatches = 0, vt_runs = 0;

	// timing of the decimating front end
	bool timing = false;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free;
};

// ------------------------------------------------------------ host planner ----

extern "C" void rtlfm_cfg_default(rtlfm_cfg *c)
{
	// demod_init(), src/rtl_fm.c:1608-1640
	memset(c, 0, sizeof(*c));
	c->mode = RTLFM_MODE_FM;
	c->downsample = 1;
	c->post_downsample = 1;
	c->rate_out = 24000;
	c->rate_out2 = -1;
	c->adc_block_const = 9;
	c->rdc_block_const = 9;
	c->output_scale = 1;
	c->block_len = 16384;  // dongle_init(), src/rtl_fm.c:1605
	c->max_blocks = 1;
}

extern "C" int rtlfm_optimal_settings(rtlfm_cfg *cfg, uint32_t freq, int32_t rate_in,
                                      int32_t min_capture_rate, int use_fifth_order, int edge,
                                      uint32_t *capture_freq, uint32_t *capture_rate)
{
	// optimal_settings(), src/rtl_fm.c:1407-1445
	if (!cfg || rate_in <= 0) return -EINVAL;
	cfg->downsample = min_capture_rate / rate_in + 1;
	cfg->downsample_passes = 0;
	if (use_fifth_order) {
		cfg->downsample_passes = (int)log2((double)cfg->downsample) + 1;
		cfg->downsample = 1 << cfg->downsample_passes;
	}
	uint32_t rate = (uint32_t)cfg->downsample * (uint32_t)rate_in;
	uint32_t f = freq;
	if (!cfg->offset_tuning) f = freq - rate / 4;
	f += (uint32_t)(edge * rate_in / 2);
	cfg->output_scale = (1 << 15) / (128 * cfg->downsample);
	if (cfg->output_scale < 1) cfg->output_scale = 1;
	if (cfg->mode == RTLFM_MODE_FM) cfg->output_scale = 1;
	if (capture_freq) *capture_freq = f;
	if (capture_rate) *capture_rate = rate;
	return 0;
}

extern "C" int32_t rtlfm_deemph_a(int32_t rate_out, int32_t tc_us)
{
	// src/rtl_fm.c:1929-1931
	double tc = (double)tc_us * 1e-6;
	return (int32_t)round(1.0 / (1.0 - exp(-1.0 / (rate_out * tc))));
}

// the first fifth_order pass that is handed a length which is not a multiple of four elements (src/rtl_fm.c:1188-1191),
// or downsample_passes if there is none; buffers are multiples of 512 bytes, so this is pass 8 or 9
static int first_irregular_pass(const rtlfm_cfg &c)
{
	for (int p = 0; p < c.downsample_passes; p++)
		if ((c.block_len >> p) % 4) return p;
	return c.downsample_passes;
}

// per-block decimated complex samples, or -1 when the boxcar phase makes it vary
static int dec_per_block(const rtlfm_cfg *c)
{
	int n0 = (int)(c->block_len / 2);
	if (c->downsample_passes > 0) return n0 >> c->downsample_passes;
	if (c->downsample <= 1) return n0;
	return n0 % c->downsample == 0 ? n0 / c->downsample : -1;
}

extern "C" int rtlfm_result_len(const rtlfm_cfg *c)
{
	if (!c) return -EINVAL;
	int n = dec_per_block(c);
	if (n < 0) return -1;
	if (c->mode == RTLFM_MODE_RAW) return c->downsample_passes > 0 ? (int)(c->block_len >> c->downsample_passes) : 2 * n;
	if (c->post_downsample > 1) n /= c->post_downsample;
	if (c->rate_out2 > 0) {
		if (c->resampler == RTLFM_RESAMPLE_ARBITRARY)
			return (int)((long long)n * c->rate_out2 / c->rate_out);
		return -1;
	}
	return n;
}

extern "C" int rtlfm_result_cap(const rtlfm_cfg *c)
{
	if (!c) return -EINVAL;
	int n0 = (int)(c->block_len / 2);
	int n = c->downsample_passes > 0 ? n0 >> c->downsample_passes
	                                  : (c->downsample > 1 ? n0 / c->downsample + 1 : n0);
	if (c->mode == RTLFM_MODE_RAW) return 2 * n + 1;
	if (c->rate_out2 > 0 && c->resampler == RTLFM_RESAMPLE_ARBITRARY && c->rate_out > 0) {
		long long up = (long long)n * c->rate_out2 / c->rate_out + 2;
		if (up > n) n = (int)up;
	}
	return n + 2;
}

static int validate_cfg(const rtlfm_cfg *c)
{
	if (c->mode < RTLFM_MODE_FM || c->mode > RTLFM_MODE_RAW) return -EINVAL;
	if (c->block_len < 512 || c->block_len > RTLFM_MAX_BLOCK_LEN || c->block_len % 512) return -EINVAL;
	if (c->downsample_passes < 0 || c->downsample_passes > RTLFM_MAX_PASSES) return -EINVAL;
	// (a buffer the passes do not divide - nine or ten passes, 512 n bytes with n odd / n % 4 != 0 - is something the
	// reference runs, src/rtl_fm.c:1188-1191: the last passes then see lengths that are not multiples of four elements;
	// k_fifth_irregular, staged_kernels.h)
	if (c->downsample_passes == 0 && c->downsample < 1) return -EINVAL;
	// a buffer shorter than the boxcar leaves low_pass() with lp_len == 0 and fm_demod() then reads
	// lowpassed[-2] (src/rtl_fm.c:955-956): outside the reference's domain
	if (c->downsample_passes == 0 && (uint32_t)c->downsample > c->block_len / 2) return -EDOM;
	if (c->comp_fir_size != 0 && c->comp_fir_size != 9) return -EINVAL;
	if (c->custom_atan < RTLFM_ATAN_STD || c->custom_atan > RTLFM_ATAN_LUT) return -EINVAL;
	if (c->max_blocks < 1) return -EINVAL;
	if (c->post_downsample < 1 || c->post_downsample > 16) return -EINVAL;
	if (c->deemph && c->deemph_a < 1) return -EINVAL;
	if (c->mode != RTLFM_MODE_RAW) {
		int per = dec_per_block(c);
		// low_pass_simple: "length must be multiple of step" (src/rtl_fm.c:740); otherwise the
		// reference sums stale samples past result_len.  Behind a boxcar that does not divide the
		// buffer the per-buffer count alternates, so no step > 1 can divide it every time.
		if (c->post_downsample > 1 && (per < 0 || per % c->post_downsample)) return -EDOM;
		if (c->rate_out2 > 0) {
			if (c->rate_out <= 0) return -EINVAL;
			if (c->resampler == RTLFM_RESAMPLE_LOW_PASS_REAL) {
				// the reference divides by zero here (src/rtl_fm.c:769)
				if (c->rate_out / c->rate_out2 == 0) return -EDOM;
			} else if (c->resampler == RTLFM_RESAMPLE_ARBITRARY) {
				// per < 0: the per-buffer count is n or n + 1 (boxcar not dividing the buffer)
				int n = per < 0 ? (int)(c->block_len / 2) / c->downsample : per / c->post_downsample;
				long long len2 = (long long)n * c->rate_out2 / c->rate_out;
				if (len2 < 1 || n < 2) return -EINVAL;
			} else {
				return -EINVAL;
			}
		}
	}
	return 0;
}

extern "C" int rtlfm_cfg_validate(const rtlfm_cfg *cfg) { return cfg ? validate_cfg(cfg) : -EINVAL; }

// ----------------------------------------------------------------- handle ----

static void init_states_host(std::vector<state_t> &v)
{
	for (auto &s : v) {
		memset(&s, 0, sizeof(s));
		s.squelch_hits = 11;  // demod_init(), src/rtl_fm.c:1615
	}
}

static int create_body(rtlfm_gpu *h);
static int options_from_env(rtlfm_gpu *h);
static long placement_held_mb(rtlfm_gpu *h);
static void ingest_destroy(rtlfm_gpu *h);
static void ingest_reset(rtlfm_gpu *h);

// everything the handle has launched: the front end's stream, then the audio tail's
static hipError_t sync_all(rtlfm_gpu *h)
{
	hipError_t e = hipStreamSynchronize(h->stream);
	if (e != hipSuccess) return e;
	return h->tail_stream ? hipStreamSynchronize(h->tail_stream) : hipSuccess;
}

extern "C" int rtlfm_gpu_create(const rtlfm_cfg *cfg, int nstreams, int device, rtlfm_gpu **out)
{
	if (!cfg || !out || nstreams < 1) return -EINVAL;
	int v = validate_cfg(cfg);
	if (v < 0) return v;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
		fprintf(stderr, "rtlfm_hip: no usable HIP device (count=%d, asked %d); there is no CPU fallback\n",
		        ndev, device);
		return -ENODEV;
	}
	HIP_TRY(hipSetDevice(device));
	rtlfm_gpu *h = new rtlfm_gpu();
	h->cfg = *cfg;
	h->nstreams = nstreams;
	h->device = device;
	// every failure below releases what was allocated so far (a service retrying on -ENOMEM must
	// not leak device memory); *out is written on success only
	int r = create_body(h);
	if (r < 0) {
		rtlfm_gpu_destroy(h);
		return r;
	}
	*out = h;
	return 0;
}

static int create_body(rtlfm_gpu *h)
{
	const rtlfm_cfg *cfg = &h->cfg;
	const int nstreams = h->nstreams;
	h->cap_blocks = cfg->max_blocks;
	const size_t L = cfg->block_len;
	h->xstride = (size_t)h->cap_blocks * (L / 2) + 16;
	size_t rs = (size_t)h->cap_blocks * L + 64;  // raw mode: 2 int16 per complex sample
	if (cfg->rate_out2 > 0 && cfg->resampler == RTLFM_RESAMPLE_ARBITRARY && cfg->rate_out2 > cfg->rate_out)
		rs = (size_t)((double)rs * cfg->rate_out2 / cfg->rate_out) + 64;
	h->rstride = (rs + 7) & ~(size_t)7;
	HIP_TRY(rtl_pool::stream_get(h->device, 0, &h->own_stream));  // (from the process-wide pool: stream_pool.h)
	h->stream = h->own_stream;
	const size_t S = (size_t)nstreams;
	for (int k = 0; k < 3; k++) HIP_TRY(hipMalloc(&h->st[k], S * sizeof(state_t)));
	for (int k = 0; k < 2; k++) {
		HIP_TRY(hipMalloc(&h->d_cnt[k], S * sizeof(int32_t)));
		HIP_TRY(hipEventCreateWithFlags(&h->ev_front[k], hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&h->ev_tail[k], hipEventDisableTiming));
	}
	// (default priority: the highest and the lowest one were both tried - round 3: highest 1 % slower; round 4,
	// three alternations on one box: c3 0.883 / 0.881 / 0.883 ms, wbfm 1.381 / 1.381 / 1.379 for default / lowest /
	// highest - the dispatcher does not seem to look at it between two compute queues)
	// Round 5: the LOWEST priority, and not for the dispatcher's sake: HIP maps the streams of one priority onto four
	// hardware queues in turn, and in a process that has created one or two other streams the tail's stream and the front
	// end's came to share one - the tail then runs behind the next front end instead of beside it (-M wbfm 1.31 -> 1.47 ms;
	// tools/queue_share_probe.py, profiles/r05_queue_share.txt).  A stream of another priority takes its queue from another pool.
	{
		int lo = 0, hi = 0;
		HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
		HIP_TRY(rtl_pool::stream_get(h->device, lo, &h->tail_stream));
		h->tail_priority = -1;
		h->tail_prio_value = lo;
	}
	h->tail_overlap = true;
	{
		// what mode_demod() can leave per stream and run: the decimated count (+1 per buffer behind a
		// boxcar that does not divide it); every later stage but the last one only shrinks it
		const size_t n0 = L / 2;
		const size_t per = cfg->downsample_passes > 0 ? n0 >> cfg->downsample_passes
		                                              : (cfg->downsample > 1 ? n0 / cfg->downsample + 1 : n0);
		h->tstride = ((size_t)h->cap_blocks * per + 64 + 63) & ~(size_t)63;
	}
	HIP_TRY(hipMalloc(&h->d_cnt2, S * sizeof(int32_t)));
	HIP_TRY(hipMalloc(&h->d_mute, S * h->cap_blocks * sizeof(int32_t)));
	HIP_TRY(hipMalloc(&h->d_levels, S * h->cap_blocks * sizeof(int32_t)));
	HIP_TRY(hipMalloc(&h->d_sums, S * h->cap_blocks * 2 * sizeof(long long)));
	HIP_TRY(hipMalloc(&h->d_adc_sums, S * h->cap_blocks * sizeof(long long)));
	HIP_TRY(hipMalloc(&h->d_rdc_avg, (S * h->cap_blocks + 32) * sizeof(int2)));  // (+ slack: k_boxcar_scan reads kRdcMany entries from a tile's first buffer on)
	HIP_TRY(hipMalloc(&h->d_sq_sums, S * h->cap_blocks * 2 * sizeof(uint32_t)));
	HIP_TRY(hipMalloc(&h->d_adc_avg, S * h->cap_blocks * sizeof(int32_t)));
	if (cfg->custom_atan == RTLFM_ATAN_LUT) {
		// atan_lut_init(), src/rtl_fm.c:881-892 — built with the host libm, as
		// the reference builds it
		std::vector<int32_t> lut(131072);
		for (int i = 0; i < 131072; i++)
			lut[i] = (int32_t)(atan((double)i / 256.0) / 3.14159 * 16384.0);
		HIP_TRY(hipMalloc(&h->d_lut, lut.size() * sizeof(int32_t)));
		HIP_TRY(hipMemcpy(h->d_lut, lut.data(), lut.size() * sizeof(int32_t), hipMemcpyHostToDevice));
	}
	int r = options_from_env(h);
	if (r < 0) return r;
	return rtlfm_gpu_reset(h);
}

static int ensure_work_buffers(rtlfm_gpu *h)
{
	const size_t S = (size_t)h->nstreams;
	if (!h->bufA) {
		HIP_TRY(hipMalloc(&h->bufA, S * h->xstride * sizeof(uint32_t)));
		HIP_TRY(hipMalloc(&h->bufB, S * h->xstride * sizeof(uint32_t)));
	}
	return 0;
}
// the decimated IQ a front end in emit mode leaves for the staged kernels: /2^level behind fifth_order passes
// (run_fused_emit), 1 / D (+ 1 per buffer) behind the boxcar (run_boxfused_emit)
extern "C" int rtlfm_gpu_malloc_apart_ex(int device, size_t bytes, const void *other, size_t other_bytes, size_t budget_bytes,
                                         void **out, int *apart, double *search_ms, size_t *walked_bytes);

// d_iq / iq_bytes: the input the first run streams from - what the front end's emit mode writes (deepA) goes a quarter of
// the HBM away from it, as the result buffers do (ensure_res_buffers; option "deep_apart" says whether it worked)
static int ensure_deep_buffers(rtlfm_gpu *h, const uint8_t *d_iq = nullptr, size_t iq_bytes = 0)
{
	if (h->deepA) return 0;
	const int N0 = (int)(h->cfg.block_len / 2);
	const bool fifth = fused::supported_emit(h->cfg);
	if (!fifth && !boxfused::supported_emit(h->cfg)) return 0;
	if (fifth) {
		const int level = h->cfg.downsample_passes < fused::kMaxP ? h->cfg.downsample_passes : fused::kMaxP;
		h->deep_stride = (size_t)h->cap_blocks * (N0 >> level);
	} else {
		h->deep_stride = (((size_t)h->cap_blocks * N0) / h->cfg.downsample + 1 + 16 + 3) & ~(size_t)3;  // rows start on 16-byte lines
	}
	const size_t bytes = (size_t)h->nstreams * h->deep_stride * sizeof(uint32_t);
	// (no search in the middle of an asynchronous call on a stream the caller owns, nor for buffers too small to matter)
	const bool search = d_iq && h->stream == h->own_stream && h->place.budget_gb > 0 && bytes >= ((size_t)32 << 20);
	if (search) {
		void *q = nullptr; int apart = 0; double ms = 0; size_t walked = 0;
		int r = rtlfm_gpu_malloc_apart_ex(h->device, bytes, d_iq, iq_bytes, (size_t)h->place.budget_gb << 30, &q, &apart, &ms, &walked);
		if (r < 0) return r;
		h->deepA = static_cast<uint32_t *>(q);
		h->place.search_ms += ms;
		if (walked > h->place.walked_peak) h->place.walked_peak = walked;
		h->place.deep_apart = apart;
	} else {
		HIP_TRY(hipMalloc(&h->deepA, bytes));
		h->place.deep_apart = 0;
	}
	if (fifth) HIP_TRY(hipMalloc(&h->deepB, bytes));
	return 0;
}
// d_iq / iq_bytes: the input the first run streams from - the buffers the front end writes the demodulated
// samples into ([parity][0]) go a quarter of the HBM away from it where they are large enough to matter
static int ensure_res_buffers(rtlfm_gpu *h, const uint8_t *d_iq = nullptr, size_t iq_bytes = 0)
{
	const size_t S = (size_t)h->nstreams;
	if (!h->res[0][0]) {
		// The search times probe launches on the null stream and synchronises the device: not in the middle
		// of an asynchronous call on a stream the caller owns - there the buffers are plain allocations, and
		// res_apart says 0 (the caller can place its OWN output with rtlfm_gpu_malloc_apart: INTEGRATION.md).
		const bool search = d_iq && h->stream == h->own_stream && h->place.budget_gb > 0;
		const size_t one = (S * h->tstride * sizeof(int16_t) + 255) & ~(size_t)255;
		int apart = 0;
		if (search) {
			// the two buffers the front end writes (one per step parity) as ONE placed block: one search, and a
			// candidate that had to be larger than the request (rtlfm_gpu_malloc_apart_ex) is shared by both
			void *q = nullptr; double ms = 0; size_t walked = 0;
			int r = rtlfm_gpu_malloc_apart_ex(h->device, 2 * one, d_iq, iq_bytes, (size_t)h->place.budget_gb << 30, &q, &apart, &ms, &walked);
			if (r < 0) return r;
			h->place.search_ms += ms;
			if (walked > h->place.walked_peak) h->place.walked_peak = walked;
			h->res[0][0] = (int16_t *)q;
			h->res[1][0] = (int16_t *)((char *)q + one);
			h->res_one_block = true;
		} else {
			for (int p = 0; p < 2; p++) HIP_TRY(hipMalloc(&h->res[p][0], one));
		}
		for (int p = 0; p < 2; p++) HIP_TRY(hipMalloc(&h->res[p][1], one));
		h->place.res_apart = apart;
	}
	return 0;
}
// bytes of the caller's input a run really covers: (S - 1) strides + the last stream's buffers (the stride may be padded)
static inline size_t iq_extent(const rtlfm_gpu *h, size_t stream_stride, int nblocks)
{
	return (size_t)(h->nstreams - 1) * stream_stride + (size_t)nblocks * h->cfg.block_len;
}

extern "C" int rtlfm_gpu_destroy(rtlfm_gpu *h)
{
	if (!h) return -EINVAL;
	hipSetDevice(h->device);
	if (h->stream) hipStreamSynchronize(h->stream);
	if (h->tail_stream) hipStreamSynchronize(h->tail_stream);
	for (hipEvent_t e : {h->ev_wait, h->ev_release, h->ev_front[0], h->ev_front[1], h->ev_tail[0], h->ev_tail[1]})
		if (e) hipEventDestroy(e);
	for (auto &p : h->ev_pending) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
	for (auto &p : h->ev_free) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
	void *ptrs[] = {h->d_arb_i, h->d_arb_frac, h->d_arb_tab, h->d_deemph_tab, h->d_deemph_inc, h->d_lpr_chunks, h->deepA, h->deepB, h->bufA, h->bufB, h->res[0][0], h->res[0][1], h->res_one_block ? nullptr : (void *)h->res[1][0], h->res[1][1],
	                h->d_cnt[0], h->d_cnt[1], h->d_cnt2,
	                h->st[0], h->st[1], h->st[2], h->d_lut, h->d_mute, h->d_levels, h->d_sq_sums, h->d_sums, h->d_adc_sums, h->d_rdc_avg, h->d_adc_avg,
	                h->vt_out, h->vt_len, h->vt_len2, h->vt_state, h->vt_cnt, h->d_slim_plan};
	for (void *p : ptrs)
		if (p) hipFree(p);
	h->fws.release();
	ingest_destroy(h);
	// The streams go back to the process-wide pool, never to hipStreamDestroy (stream_pool.h: the runtime releases a freed
	// object of a destroyed stream once more, later, in whoever's memory it has become - LAB.md I.21).
	// RTLFM_DESTROY_STREAMS=1 (tools/host_uaf_probe.py --mode destroy) restores round 5's behaviour for the comparison.
	static const bool destroy_streams = [] { const char *e = getenv("RTLFM_DESTROY_STREAMS"); return e && *e == '1'; }();
	if (destroy_streams) {
		if (h->own_stream) hipStreamDestroy(h->own_stream);
		if (h->tail_stream) hipStreamDestroy(h->tail_stream);
	} else {
		rtl_pool::stream_put(h->device, 0, h->own_stream);
		rtl_pool::stream_put(h->device, h->tail_prio_value, h->tail_stream);
	}
	delete h;
	return 0;
}

extern "C" int rtlfm_gpu_reset(rtlfm_gpu *h)
{
	if (!h) return -EINVAL;
	HIP_TRY(hipSetDevice(h->device));
	std::vector<state_t> init((size_t)h->nstreams);
	init_states_host(init);
	HIP_TRY(sync_all(h));
	HIP_TRY(hipMemcpy(h->st[h->st_cur], init.data(), init.size() * sizeof(state_t), hipMemcpyHostToDevice));
	ingest_reset(h);
	return 0;
}

extern "C" int rtlfm_gpu_state_get(rtlfm_gpu *h, int stream, rtlfm_stream_state *st)
{
	if (!h || !st || stream < 0 || stream >= h->nstreams) return -EINVAL;
	HIP_TRY(hipSetDevice(h->device));
	HIP_TRY(sync_all(h));
	HIP_TRY(hipMemcpy(st, h->st[h->st_cur] + stream, sizeof(*st), hipMemcpyDeviceToHost));
	return 0;
}

extern "C" int rtlfm_gpu_state_set(rtlfm_gpu *h, int stream, const rtlfm_stream_state *st)
{
	if (!h || !st || stream < 0 || stream >= h->nstreams) return -EINVAL;
	HIP_TRY(hipSetDevice(h->device));
	HIP_TRY(sync_all(h));
	HIP_TRY(hipMemcpy(h->st[h->st_cur] + stream, st, sizeof(*st), hipMemcpyHostToDevice));
	return 0;
}

extern "C" int rtlfm_gpu_sync(rtlfm_gpu *h)
{
	if (!h) return -EINVAL;
	HIP_TRY(hipSetDevice(h->device));
	HIP_TRY(sync_all(h));
	return 0;
}

extern "C" int rtlfm_gpu_set_stream(rtlfm_gpu *h, void *s)
{
	if (!h) return -EINVAL;
	HIP_TRY(sync_all(h));
	h->stream = s ? (hipStream_t)s : h->own_stream;
	// On a caller-owned stream everything the handle launches is ordered on THAT stream: the audio
	// tail does not move to the handle's second stream, so a consumer enqueued on the caller's stream
	// behind rtlfm_gpu_run_device() sees the finished d_out / d_out_len without any further call.
	h->tail_overlap = h->stream == h->own_stream && !h->tail_serial_asked;
	return 0;
}

// Cross-stream ordering without a host synchronisation.  NULL names the legacy default stream
// (what torch's default stream is), which hipEventRecord / hipStreamWaitEvent accept as such.
extern "C" int rtlfm_gpu_wait_for(rtlfm_gpu *h, void *producer_stream)
{
	if (!h) return -EINVAL;
	if ((hipStream_t)producer_stream == h->stream) return 0;
	HIP_TRY(hipSetDevice(h->device));
	if (!h->ev_wait) HIP_TRY(hipEventCreateWithFlags(&h->ev_wait, hipEventDisableTiming));
	HIP_TRY(hipEventRecord(h->ev_wait, (hipStream_t)producer_stream));
	HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_wait, 0));
	return 0;
}

extern "C" int rtlfm_gpu_release_to(rtlfm_gpu *h, void *consumer_stream)
{
	if (!h) return -EINVAL;
	if ((hipStream_t)consumer_stream == h->stream) return 0;
	HIP_TRY(hipSetDevice(h->device));
	if (!h->ev_release) HIP_TRY(hipEventCreateWithFlags(&h->ev_release, hipEventDisableTiming));
	HIP_TRY(hipEventRecord(h->ev_release, h->stream));
	HIP_TRY(hipStreamWaitEvent((hipStream_t)consumer_stream, h->ev_release, 0));
	for (int p = 0; p < 2; p++)  // and for the audio tails still running on their own stream
		if (h->tail_pending[p]) HIP_TRY(hipStreamWaitEvent((hipStream_t)consumer_stream, h->ev_tail[p], 0));
	return 0;
}

extern "C" int rtlfm_gpu_set_path(rtlfm_gpu *h, int path)
{
	if (!h || path < 0 || path > 4) return -EINVAL;
	h->path = path > 2 ? 2 : path;
	if (path == 2) h->fws.pass0_engine = -1;  // the compiled default
	if (path > 2) h->fws.pass0_engine = path - 3;
	return 0;
}
extern "C" int rtlfm_gpu_last_path(rtlfm_gpu *h) { return h ? h->last_path : -EINVAL; }

// The one place the library's tunables and A/B switches live (no environment look-ups on the
// launch path).  RTLFM_OPTIONS="name=value,..." is applied once, by rtlfm_gpu_create.
static int *option_slot(rtlfm_gpu *h, const char *name)
{
	struct { const char *n; int *p; } tab[] = {
		{"fused_waves", &h->fws.target_waves}, {"fused_waves_tail", &h->fws.target_waves_tail}, {"fused_min_tiles", &h->fws.min_tiles},
		{"fused_tiles_per_seg", &h->fws.tiles_per_seg}, {"fused_debug", &h->fws.debug}, {"fused_gss", &h->fws.gss_x10},
		{"pass0_engine", &h->fws.pass0_engine},
		{"deemph_sequential", &h->opt.deemph_sequential}, {"deemph_four_pass", &h->opt.deemph_four_pass},
		{"lpr_separate", &h->opt.lpr_separate}, {"lpr_scalar_stores", &h->opt.lpr_scalar_stores}, {"lpr_chunk", &h->opt.lpr_chunk},
		{"tail_sync", &h->opt.tail_sync}, {"apart_budget_gb", &h->place.budget_gb}, {"ring_force_retry", &h->place.force_retry},
		{"arb_span", &h->opt.arb_span}, {"arb_chunk", &h->opt.arb_chunk}, {"arb_waves", &h->opt.arb_waves}, {"lpr_threads", &h->opt.lpr_threads}, {"arb_serial", &h->opt.arb_serial}, {"lpr_ring", &h->opt.lpr_ring}, {"squelch_fused", &h->opt.squelch_fused}, {"adc_separate", &h->opt.adc_separate}, {"deep_rest", &h->opt.deep_rest}, {"box_store", &h->fws.box_store}, {"fused_store", &h->fws.fused_store},
		{"verify_twice", &h->opt.verify_twice}, {"verify_inject", &h->opt.verify_inject}, {"lpr_slim", &h->opt.lpr_slim}, {"lpr_slim_chunk", &h->opt.lpr_slim_chunk}, {"lpr_slim_prio", &h->opt.lpr_slim_prio},
	};
	for (auto &t : tab)
		if (!strcmp(t.n, name)) return t.p;
	return nullptr;
}

extern "C" int rtlfm_gpu_set_option(rtlfm_gpu *h, const char *name, long value)
{
	if (!h || !name) return -EINVAL;
	if (!strcmp(name, "tail_serial")) {
		// the audio tail on the front end's stream instead of its own
		HIP_TRY(sync_all(h));
		h->tail_overlap = value == 0 && h->stream == h->own_stream;
		h->tail_serial_asked = value != 0;
		return 0;
	}
	if (!strcmp(name, "tail_priority")) {
		// the tail's stream re-created with a priority of its own: -1 lowest, 0 default, 1 highest (a stream of another
		// priority never shares a hardware queue with the front end's: LAB.md I.8)
		if (value < -1 || value > 1) return -EINVAL;
		HIP_TRY(sync_all(h));
		int lo = 0, hi = 0;
		HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));  // lo = the numerically greatest = lowest priority
		hipStream_t q = nullptr;
		const int pv = value < 0 ? lo : value > 0 ? hi : 0;
		HIP_TRY(rtl_pool::stream_get(h->device, pv, &q));
		rtl_pool::stream_put(h->device, h->tail_prio_value, h->tail_stream);
		h->tail_stream = q;
		h->tail_priority = (int)value;
		h->tail_prio_value = pv;
		return 0;
	}
	int *slot = option_slot(h, name);
	if (!slot) return -ENOENT;
	// every range check first: a refused value must leave the handle as it was
	if ((!strcmp(name, "fused_waves") || !strcmp(name, "fused_waves_tail")) && value < 1) return -EINVAL;
	if (!strcmp(name, "pass0_engine") && (value < -1 || value > 1)) return -EINVAL;
	if ((!strcmp(name, "fused_min_tiles") || !strcmp(name, "fused_tiles_per_seg")) && value < 0) return -EINVAL;
	if (!strcmp(name, "apart_budget_gb") && (value < 0 || value > 256)) return -EINVAL;
	if ((!strcmp(name, "box_store") || !strcmp(name, "fused_store")) && (value < -1 || value > 1)) return -EINVAL;
	// the chunk tables of the one-pass deemph + low_pass_real kernel are sized from it: keep it in a sane range
	if ((!strcmp(name, "lpr_chunk") || !strcmp(name, "lpr_slim_chunk")) && (value < 256 || value > (1 << 20))) return -EINVAL;
	if (!strcmp(name, "arb_chunk") && value != 32 && value != 64) return -EINVAL;
	if (!strcmp(name, "arb_waves") && (value < 0 || value > kSpecArbMaxWaves)) return -EINVAL;
	if (!strcmp(name, "lpr_threads") && (value < 64 || value > kSpecLprThreads || value % 64)) return -EINVAL;
	if (!strcmp(name, "arb_serial")) {
		if (value < -1 || value > 1) return -EINVAL;
		HIP_TRY(sync_all(h));  // the tail may change streams: nothing of either in flight
	}
	// ... then what an accepted value implies
	if (!strcmp(name, "fused_waves")) h->fws.target_waves_tail = (int)value;  // one number for both unless fused_waves_tail follows
	if (!strcmp(name, "fused_waves") || !strcmp(name, "fused_waves_tail")) h->fws.target_waves_tail_fifth = 0;  // an explicit number rules
	if (!strcmp(name, "fused_waves") || !strcmp(name, "fused_min_tiles")) h->fws.plan_by_caller = true;       // ... and so do these
	*slot = (int)value;
	return 0;
}

extern "C" int rtlfm_gpu_get_option(rtlfm_gpu *h, const char *name, long *value)
{
	if (!h || !name || !value) return -EINVAL;
	if (!strcmp(name, "tail_serial")) { *value = h->tail_overlap ? 0 : 1; return 0; }
	if (!strcmp(name, "tail_priority")) { *value = h->tail_priority; return 0; }
	// read-only: where the write streams' buffers ended up (rtlfm_gpu_malloc_apart_ex) and what finding out cost
	if (!strcmp(name, "ring_apart")) { *value = h->place.ring_apart; return 0; }
	if (!strcmp(name, "ring_tries")) { *value = h->place.ring_tries; return 0; }
	if (!strcmp(name, "poison")) { *value = rtl_debug::poison_on() ? 1 : 0; return 0; }  // RTLFM_POISON=1 (debug_poison.h)
	if (!strcmp(name, "res_apart")) { *value = h->place.res_apart; return 0; }
	if (!strcmp(name, "deep_apart")) { *value = h->place.deep_apart; return 0; }
	if (!strcmp(name, "placement_ms")) { *value = (long)(h->place.search_ms + 0.5); return 0; }
	if (!strcmp(name, "placement_walked_mb")) { *value = (long)(h->place.walked_peak >> 20); return 0; }
	if (!strcmp(name, "placement_held_mb")) { *value = placement_held_mb(h); return 0; }
	if (!strcmp(name, "verify_mismatches")) { *value = h->vt_mismatches; return 0; }  // verify_twice: runs whose two executions differed
	if (!strcmp(name, "verify_runs")) { *value = h->vt_runs; return 0; }
	// debugging (tests/test_soak_gpu.py): the host addresses of the runtime objects the handle owns - a hipStream_t / hipEvent_t
	// IS the address of the runtime's object, which the runtime frees when the handle is destroyed
	if (!strcmp(name, "dbg_own_stream")) { *value = (long)(uintptr_t)h->own_stream; return 0; }
	if (!strcmp(name, "dbg_tail_stream")) { *value = (long)(uintptr_t)h->tail_stream; return 0; }
	if (!strncmp(name, "dbg_event", 9) && name[9] >= '0' && name[9] <= '5' && !name[10]) {
		const hipEvent_t ev[6] = {h->ev_wait, h->ev_release, h->ev_front[0], h->ev_front[1], h->ev_tail[0], h->ev_tail[1]};
		*value = (long)(uintptr_t)ev[name[9] - '0'];
		return 0;
	}
	if (!strcmp(name, "dbg_handle")) { *value = (long)(uintptr_t)h; return 0; }
	if (!strcmp(name, "dbg_handle_bytes")) { *value = (long)sizeof(*h); return 0; }
	int *slot = option_slot(h, name);
	if (!slot) return -ENOENT;
	*value = *slot;
	return 0;
}

static int options_from_env(rtlfm_gpu *h)
{
	const char *e = getenv("RTLFM_OPTIONS");
	if (!e || !*e) return 0;
	std::string all(e);
	size_t at = 0;
	while (at < all.size()) {
		size_t end = all.find(',', at);
		if (end == std::string::npos) end = all.size();
		const std::string kv = all.substr(at, end - at);
		const size_t eq = kv.find('=');
		if (eq != std::string::npos) {
			int r = rtlfm_gpu_set_option(h, kv.substr(0, eq).c_str(), atol(kv.c_str() + eq + 1));
			if (r < 0) { fprintf(stderr, "rtlfm_hip: RTLFM_OPTIONS: '%s' not accepted (%d)\n", kv.c_str(), r); return r; }
		}
		at = end + 1;
	}
	return 0;
}

extern "C" int rtlfm_gpu_timing_enable(rtlfm_gpu *h, int on)
{
	if (!h) return -EINVAL;
	h->timing = on != 0;
	return 0;
}

static int report_clock_stamps(rtlfm_gpu *h);

extern "C" int rtlfm_gpu_timing_read(rtlfm_gpu *h, double *front_ms, int *launches)
{
	if (!h) return -EINVAL;
	HIP_TRY(hipSetDevice(h->device));
	HIP_TRY(sync_all(h));
	if (h->fws.stamps && (h->fws.debug & 16)) {
		int r2 = report_clock_stamps(h);
		if (r2 < 0) return r2;
	}
	double total = 0;
	int n = 0;
	for (auto &p : h->ev_pending) {
		float ms = 0;
		HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
		total += ms;
		n++;
		h->ev_free.push_back(p);
	}
	h->ev_pending.clear();
	if (front_ms) *front_ms = total;
	if (launches) *launches = n;
	return 0;
}

static int timing_begin(rtlfm_gpu *h, std::pair<hipEvent_t, hipEvent_t> &ev)
{
	if (!h->timing) return 0;
	if (!h->ev_free.empty()) {
		ev = h->ev_free.back();
		h->ev_free.pop_back();
	} else {
		HIP_TRY(hipEventCreate(&ev.first));
		HIP_TRY(hipEventCreate(&ev.second));
	}
	HIP_TRY(hipEventRecord(ev.first, h->stream));
	return 0;
}
static int timing_end(rtlfm_gpu *h, std::pair<hipEvent_t, hipEvent_t> &ev)
{
	if (!h->timing) return 0;
	HIP_TRY(hipEventRecord(ev.second, h->stream));
	h->ev_pending.push_back(ev);
	return 0;
}

// ------------------------------------------------------------ the chain ----

static inline int grid_for(size_t work, int block = 256, int cap = 256 * 16)
{
	size_t g = (work + block - 1) / block;
	if (g < 1) g = 1;
	if (g > (size_t)cap) g = cap;
	return (int)g;
}

// Audio tail shared by both paths: everything full_demod() does after
// mode_demod() (src/rtl_fm.c:1260-1271).  `src` holds per-stream demodulated
// samples (uniform count T, or per-stream d_cnt when `varcnt`); the final
// result is left in dst with dst_stride and its per-stream length in
// d_out_len (may be NULL).
struct TailPlan {
	bool post, deemph, adc, lpr, arb;
	bool deemph_spec;  // deemph_filter alone on a long run: the one-pass kernel, which works OUT of place (k_deemph_spec)
	int oop() const { return (post ? 1 : 0) + (deemph_spec ? 1 : 0) + ((lpr || arb) ? 1 : 0); }
	bool any() const { return post || deemph || adc || lpr || arb; }
};
// the time-parallel forms of deemph_filter apply: a long run, a divisor whose contracted interval fits a wave
static bool deemph_scans(const rtlfm_gpu *h, int T)
{
	const rtlfm_cfg &c = h->cfg;
	return T >= 2048 && c.deemph_a >= 2 && 2 * c.deemph_a + 2 <= kDeemphGap && !h->no_deemph_scan && !h->opt.deemph_sequential;
}
// nblocks: the buffers of the run (the plan depends on how long a stream's run is)
static TailPlan plan_tail(const rtlfm_gpu *h, int nblocks)
{
	const rtlfm_cfg &c = h->cfg;
	TailPlan t{};
	if (c.mode == RTLFM_MODE_RAW) return t;
	t.post = c.post_downsample > 1;
	t.deemph = c.deemph != 0;
	t.adc = c.dc_block_audio != 0;
	t.lpr = c.rate_out2 > 0 && c.resampler == RTLFM_RESAMPLE_LOW_PASS_REAL;
	t.arb = c.rate_out2 > 0 && c.resampler == RTLFM_RESAMPLE_ARBITRARY;
	if (t.deemph && !t.lpr && !t.arb && !h->opt.deemph_four_pass) {
		// the demodulated samples of a stream's run, as run_tail() will be told (an upper bound behind a boxcar that does not
		// divide the buffer)
		const long long n0 = c.block_len / 2;
		long long T = c.downsample_passes > 0 ? (long long)nblocks * (n0 >> c.downsample_passes)
		                                      : ((long long)nblocks * n0) / c.downsample + ((n0 % c.downsample) ? 1 : 0);
		if (t.post) T /= c.post_downsample;
		t.deemph_spec = T < (1ll << 30) && deemph_scans(h, (int)T);
	}
	return t;
}

// Where mode_demod() should write so that the chain ends in `final`.
static void tail_route(rtlfm_gpu *h, const TailPlan &tp, int16_t *final_dst, size_t final_stride,
                       int16_t **demod_dst, size_t *demod_stride)
{
	if (tp.oop() == 0) {
		*demod_dst = final_dst;
		*demod_stride = final_stride;
	} else {
		*demod_dst = h->res[h->step & 1][0];
		*demod_stride = h->tstride;
	}
}

// Config 3's own tail (deemph + arbitrary_upsample: 16384 short waves that fill the GPU by themselves, 63 us alone) behind a
// front end of thousands of streams is better off IN LINE, on the front end's stream: beside the next front end it takes
// 1.5 % more of the step than behind this one, and a front end with nothing beside it runs best in few long segments
// (the planner's no-tail target: another 3 %, LAB.md I.31).  A rule of the configuration and the stream count only - a
// handle's tail never changes streams between runs.
static bool arb_tail_in_line(const rtlfm_gpu *h, const TailPlan &tp)
{
	const rtlfm_cfg &c = h->cfg;
	return tp.deemph && tp.arb && !tp.post && !tp.adc && !tp.lpr && c.rate_out2 > c.rate_out &&
	       (h->opt.arb_serial > 0 || (h->opt.arb_serial < 0 && h->nstreams >= 2048));
}

// Does this tail run as ONE kernel (run_tail: k_deemph_spec_arb = 1, k_deemph_spec_lpr = 2), 0 if not?
static int one_pass_tail(const rtlfm_gpu *h, const TailPlan &tp, const int16_t *cur, size_t cur_stride, int T, bool varcnt,
                         int nblocks, int Nblk, int D)
{
	const rtlfm_cfg &c = h->cfg;
	if (!tp.deemph || tp.post || tp.adc || h->opt.deemph_four_pass) return 0;
	const bool scan = deemph_scans(h, T);
	if (!scan) return 0;
	if (tp.lpr) return h->opt.lpr_separate ? 0 : 2;
	if (!tp.arb) return 0;
	const int Ws = ((16 * c.deemph_a + 64 + 63) / 64) * 64;
	const int arb_l2 = (int)((long long)Nblk * c.rate_out2 / c.rate_out);
	const bool ok = D == 1 && !varcnt && Nblk >= 2 && arb_l2 > Nblk && (long long)(Nblk + 1) * arb_l2 < (1ll << 31) && Ws <= 256 &&
	                (uintptr_t)cur % 16 == 0 && cur_stride % 8 == 0 && T == nblocks * Nblk;
	return ok ? 1 : 0;
}

// Buffer extents: buffer b of a stream owns the decimated samples [dec_block_begin(b),
// dec_block_begin(b + 1)) of the run — Nblk input samples per buffer through a boxcar D with the
// carried prev_index; (Nblk, 1) describes a uniform count of Nblk per buffer.
static int run_tail(rtlfm_gpu *h, const TailPlan &tp, int16_t *cur, size_t cur_stride, int T, bool varcnt,
                    int nblocks, int Nblk, int D, int16_t *final_dst, size_t final_stride, int32_t *d_out_len)
{
	const rtlfm_cfg &c = h->cfg;
	const int S = h->nstreams;
	const int par = (int)(h->step & 1);
	const state_t *sin = h->st[h->st_cur];
	state_t *sout = h->st[(h->st_cur + 1) % 3];
	int remaining_oop = tp.oop();
	int32_t *cnt = varcnt ? h->d_cnt[par] : nullptr;
	// The tail runs behind the front end on a stream of its own, so that the front end of the next
	// step (a different res[] set, d_cnt[] and state copy, see struct rtlfm_gpu) overlaps it: at
	// config 3 the tail is a quarter of the step but 3 % of the bytes.  rtlfm_gpu_run_device() makes
	// the step after next wait for it before it reuses this parity's buffers.
	hipStream_t q = h->stream;
	// ... but config 3's own tail in line (arb_tail_in_line above)
	const bool own_stream = tp.any() && h->tail_overlap && !arb_tail_in_line(h, tp);
	if (own_stream) {
		HIP_TRY(hipEventRecord(h->ev_front[par], h->stream));
		HIP_TRY(hipStreamWaitEvent(h->tail_stream, h->ev_front[par], 0));
		q = h->tail_stream;
		// The front end copied the whole record sin -> sout while the previous step's tail may still
		// have been writing the tail's fields of sin.  The tail stream is in order, so here those are
		// final: carry them over before any stage of this tail runs (a stage that leaves a field
		// alone - a skipped stream, an early return - then leaves the right value behind).
		// The one-kernel tails write every field a tail of their configuration owns, for every stream: no copy.
		if (!one_pass_tail(h, tp, cur, cur_stride, T, varcnt, nblocks, Nblk, D))
			k_tail_state_copy<<<grid_for(S, 64), 64, 0, q>>>(sin, sout, S);
	}
	struct Done {  // whatever path leaves: mark the end of this step's tail
		rtlfm_gpu *h; int par; bool on;
		~Done() { if (on && hipEventRecord(h->ev_tail[par], h->tail_stream) == hipSuccess) h->tail_pending[par] = true; }
	} done{h, par, own_stream};
	auto next_dst = [&](int16_t **d, size_t *ds) {
		remaining_oop--;
		if (remaining_oop == 0) { *d = final_dst; *ds = final_stride; }
		else { *d = (cur == h->res[par][0]) ? h->res[par][1] : h->res[par][0]; *ds = h->tstride; }
	};
	if (tp.post) {
		// low_pass_simple: "length must be multiple of step" (src/rtl_fm.c:740) — validate_cfg only
		// lets uniform per-buffer counts through
		int16_t *d; size_t ds;
		next_dst(&d, &ds);
		int Tout = T / c.post_downsample;
		k_post_downsample<<<grid_for((size_t)S * Tout), 256, 0, q>>>(cur, cur_stride, d, ds, Tout, S,
		                                                           c.post_downsample);
		cur = d; cur_stride = ds; T = Tout;
		Nblk = T / nblocks; D = 1;
	}
	bool fuse_lpr = false;
	int16_t *lpr_dst = nullptr; size_t lpr_ds = 0;
	if (tp.deemph && c.deemph_a == 1) {
		// the filter is the identity on the samples and keeps the last one (staged_kernels.h)
		k_deemph_identity<<<grid_for(S, 64), 64, 0, q>>>(cur, cur_stride, T, cnt, S, sout);
	} else if (tp.deemph) {
		DeemphStep st;
		st.a = (uint32_t)c.deemph_a; st.half = (uint32_t)(c.deemph_a / 2);
		const bool pow2 = c.deemph_a >= 1 && c.deemph_a <= 32768 && (c.deemph_a & (c.deemph_a - 1)) == 0;
		const bool magic = !pow2 && c.deemph_a >= 2 && c.deemph_a <= 32768;
		st.magic = 0;
		if (pow2) { while ((1u << st.magic) < st.a) st.magic++; }
		else if (magic) st.magic = (uint32_t)((0x100000000ull + st.a - 1) / st.a);
		const int M = pow2 ? 2 : magic ? 1 : 0;
		const unsigned grid = (unsigned)((S + 63) / 64);
		// few, long streams: parallel over time (staged_kernels.h, k_deemph_scan_*); the interval
		// of candidate states must fit a wave (2a + 2 <= kDeemphGap)
		const bool scan = deemph_scans(h, T);
		const bool dbg_sync = h->opt.tail_sync != 0;
#define RTLFM_DBG_SYNC(what) do { if (dbg_sync) { hipError_t e_ = hipStreamSynchronize(q); fprintf(stderr, "rtlfm_hip[tail]: %s done (%s)\n", what, hipGetErrorString(e_)); } } while (0)
		if (tp.deemph_spec) {
			// deemph_filter with no resampler behind it, a long run: ONE pass, out of place (staged_kernels.h, k_deemph_spec)
			int16_t *dd2; size_t dds2;
			next_dst(&dd2, &dds2);
			const int Ws = ((16 * c.deemph_a + 64 + 63) / 64) * 64;
			const bool congruent = (((uintptr_t)cur ^ (uintptr_t)dd2) & 15) == 0 && ((cur_stride * 2) & 15) == ((dds2 * 2) & 15);
			if (scan && congruent) {
				int Lc = h->opt.lpr_chunk;
				{
					long long fit = (long long)S * T / 65536;
					const long long lo = 6ll * Ws > 512 ? 6ll * Ws : 512;
					if (fit < lo) fit = lo;
					if (fit < Lc) Lc = (int)fit;
				}
				Lc = (Lc + 63) & ~63;  // whole 128-byte rounds
				const int mcd = T / Lc + 2;
				const int spw = mcd >= kSpecLprThreads ? 1 : kSpecLprThreads / mcd;
				const unsigned gsp = (unsigned)((S + spw - 1) / spw);
#define RTLFM_SPEC_ONLY(MM) k_deemph_spec<MM><<<gsp, kSpecLprThreads, 0, q>>>(cur, cur_stride, dd2, dds2, T, cnt, S, st, mcd, Lc, Ws, sin, sout)
				if (M == 2) RTLFM_SPEC_ONLY(2); else if (M == 1) RTLFM_SPEC_ONLY(1); else RTLFM_SPEC_ONLY(0);
#undef RTLFM_SPEC_ONLY
				RTLFM_DBG_SYNC("one pass (deemph)");
			} else {
				// (the plan promised an out-of-place stage and the run turned out otherwise - rows that do not share their
				// alignment, a run that is shorter than planned: the sequential filter in place, then a copy)
				if (M == 2) k_deemph<2><<<grid, 64, 0, q>>>(cur, cur_stride, T, cnt, S, st, sin, sout);
				else if (M == 1) k_deemph<1><<<grid, 64, 0, q>>>(cur, cur_stride, T, cnt, S, st, sin, sout);
				else k_deemph<0><<<grid, 64, 0, q>>>(cur, cur_stride, T, cnt, S, st, sin, sout);
				HIP_TRY(hipMemcpy2DAsync(dd2, dds2 * sizeof(int16_t), cur, cur_stride * sizeof(int16_t), (size_t)T * sizeof(int16_t), S,
				                         hipMemcpyDeviceToDevice, q));
			}
			cur = dd2; cur_stride = dds2;
		} else if (scan) {
			// deemph_filter followed directly by low_pass_real (-M wbfm): the filtered samples go straight
			// into the resampler's accumulator instead of back to memory (staged_kernels.h)
			fuse_lpr = tp.lpr && !tp.adc && !h->opt.lpr_separate;
			// One pass where the resampler follows directly: every chunk finds its incoming state from the
			// W samples before it (staged_kernels.h, k_deemph_spec_lpr); a stream it cannot settle that way
			// (silence) is redone by the kernel's own last lane for that stream.
			// ... and where arbitrary_resample follows directly, on uniform buffers that it upsamples
			// (config 3): one pass from the demodulated samples to the resampled output (k_deemph_spec_arb)
			const int one = one_pass_tail(h, tp, cur, cur_stride, T, varcnt, nblocks, Nblk, D);
			const bool spec = one == 2, spec_arb = one == 1;
			const int Ws = ((16 * c.deemph_a + 64 + 63) / 64) * 64;
			const int arb_l2 = (int)((long long)Nblk * c.rate_out2 / c.rate_out);
			const size_t arb_lds = (size_t)(Ws / kArbChunk + 64) * kArbStride * sizeof(int16_t);
			int32_t *cnt_dst = d_out_len ? d_out_len : h->d_cnt2;
			if (spec_arb) {
				if (h->arb_len2 != arb_l2 || h->arb_len1 != Nblk) {
					// (i, frac) of every output of a buffer, as arbitrary_upsample's loop (src/rtl_fm.c:1114-1135)
					// has them when it writes buf2[j]: walked once here, shared by every stream and buffer
					// (ti: the i, a gap, then the padded LDS offset of sample i - 1 for k_deemph_spec_arb's resampler)
					std::vector<int32_t> ti(2 * (size_t)arb_l2 + kArbTabGap);
					std::vector<double> tf((size_t)arb_l2);
					std::vector<ArbTab> tt((size_t)arb_l2);
					int i = 1, tick = 0;
					for (int j = 0; j < arb_l2; j++) {
						ti[j] = i;
						tf[j] = (double)tick / (double)arb_l2;
						tt[j] = ArbTab{tf[j], i, 0};
						ti[(size_t)arb_l2 + kArbTabGap + j] = 2 * i + 16 * (i >> 5) - 2;
						tick += Nblk;
						if (tick > arb_l2) { tick -= arb_l2; i++; }
						if (i >= Nblk) { i = Nblk - 1; tick = arb_l2; }
					}
					HIP_TRY(hipStreamSynchronize(q));  // the tables of another geometry may still be in use
					// the keys say "no tables" until every allocation and copy has succeeded (a failure half way must not
					// leave the old geometry's keys beside freed or short tables)
					h->arb_len2 = 0; h->arb_len1 = 0;
					for (void **pp : {(void **)&h->d_arb_i, (void **)&h->d_arb_frac, (void **)&h->d_arb_tab})
						if (*pp) { HIP_TRY(hipFree(*pp)); *pp = nullptr; }
					HIP_TRY(hipMalloc(&h->d_arb_i, ti.size() * sizeof(int32_t)));
					HIP_TRY(hipMalloc(&h->d_arb_frac, tf.size() * sizeof(double)));
					HIP_TRY(hipMalloc(&h->d_arb_tab, tt.size() * sizeof(ArbTab)));
					HIP_TRY(hipMemcpy(h->d_arb_i, ti.data(), ti.size() * sizeof(int32_t), hipMemcpyHostToDevice));
					HIP_TRY(hipMemcpy(h->d_arb_frac, tf.data(), tf.size() * sizeof(double), hipMemcpyHostToDevice));
					HIP_TRY(hipMemcpy(h->d_arb_tab, tt.data(), tt.size() * sizeof(ArbTab), hipMemcpyHostToDevice));
					h->arb_len2 = arb_l2; h->arb_len1 = Nblk;
				}
				int16_t *arb_dst = nullptr; size_t arb_ds = 0;
				next_dst(&arb_dst, &arb_ds);
				if (arb_dst != final_dst) return -EFAULT;  // routing bug
				if (h->opt.arb_span) {
					// round 5's form: the span linear in LDS, 32 or 64 samples per lane (staged_kernels.h, k_deemph_arb_span)
					const int Cc = h->opt.arb_chunk == 64 ? 64 : 32;
					const int sp_n = (T + 64 * Cc - 1) / (64 * Cc);
					const size_t lds_w = (((size_t)(Ws + 64 * Cc) * sizeof(int16_t) + 16) + 15) & ~(size_t)15;
					int wv = 8192 / S;
					if (wv > sp_n) wv = sp_n;
					if (wv > kSpecArbMaxWaves) wv = kSpecArbMaxWaves;
					if (wv < 1) wv = 1;
					const int M3 = c.deemph_a == 2 ? 3 : M;
#define RTLFM_ARB_SPAN(MM, CC) k_deemph_arb_span<MM, CC><<<(unsigned)S, 64 * wv, lds_w * wv, q>>>(cur, cur_stride, T, S, st, Ws, sp_n, Nblk, arb_l2, nblocks, \
				h->d_arb_tab, arb_dst, arb_ds, sin, sout, lds_w, cnt_dst)
					if (Cc == 64) { if (M3 == 3) RTLFM_ARB_SPAN(3, 64); else if (M3 == 2) RTLFM_ARB_SPAN(2, 64); else if (M3 == 1) RTLFM_ARB_SPAN(1, 64); else RTLFM_ARB_SPAN(0, 64); }
					else { if (M3 == 3) RTLFM_ARB_SPAN(3, 32); else if (M3 == 2) RTLFM_ARB_SPAN(2, 32); else if (M3 == 1) RTLFM_ARB_SPAN(1, 32); else RTLFM_ARB_SPAN(0, 32); }
#undef RTLFM_ARB_SPAN
					RTLFM_DBG_SYNC("one pass (arb span)");
					return 0;
				}
				const int arb_spans = (T + 64 * kArbChunk - 1) / (64 * kArbChunk);
				// a workgroup per stream (nothing crosses workgroups), its spans dealt to up to eight waves: about 16384
				// waves in all - two and a half rounds of what the GPU holds, so that the last round is a small part of
				// the launch (8192: config 3's tail alone 70 us instead of 65, LAB.md I.31)
				int wpw = h->opt.arb_waves > 0 ? h->opt.arb_waves : 16384 / S;
				if (wpw > arb_spans) wpw = arb_spans;
				if (wpw > kSpecArbMaxWaves) wpw = kSpecArbMaxWaves;
				if (wpw < 1) wpw = 1;
#define RTLFM_SPEC_ARB(MM) k_deemph_spec_arb<MM><<<(unsigned)S, 64 * wpw, arb_lds * wpw, q>>>(cur, cur_stride, T, S, st, Ws, arb_spans, Nblk, arb_l2, nblocks, \
				h->d_arb_i, h->d_arb_frac, arb_dst, arb_ds, sin, sout, arb_lds, cnt_dst)
				// a == 2 (rtl_fm -s 24k -E deemp): the filter step is (x + avg + [x > avg]) >> 1
				if (c.deemph_a == 2) RTLFM_SPEC_ARB(3); else if (M == 2) RTLFM_SPEC_ARB(2); else if (M == 1) RTLFM_SPEC_ARB(1); else RTLFM_SPEC_ARB(0);
#undef RTLFM_SPEC_ARB
				RTLFM_DBG_SYNC("one pass (arb)");
				return 0;  // the one operation of this tail: filter, resampler, state and counts
			}
			if (spec) {
				// chunk length of the one-pass form: about 2720 samples (2040: +1 %, 1360: +2 % on the wbfm step),
				// and a multiple of the resampler's period fast / gcd(fast, slow) where that is short - then all
				// chunks of a stream start at the same phase, the lanes of a wave emit at the same samples and
				// the emission branch is taken by whole waves instead of by a few lanes every sample
				// `lpr_chunk` samples per lane where the run fills the GPU that way (-M wbfm at 1024 streams: 64 K lanes); with
				// fewer samples in all, shorter chunks - about 64 K lanes again, never so short that the settling (W samples
				// walked twice per chunk) outweighs the chunk (256 streams of rtl_fm -s 48k -r 24k -E deemp: 294 waves on 1024
				// SIMDs took 1.45 ms for 102 M samples)
				const bool slim = h->opt.lpr_slim != 0;
				int Lwant = slim ? h->opt.lpr_slim_chunk : h->opt.lpr_chunk;  // 256 .. 2^20 (rtlfm_gpu_set_option)
				{
					long long fit = (long long)S * T / 65536;
					const long long lo = 6ll * Ws > 512 ? 6ll * Ws : 512;
					if (fit < lo) fit = lo;
					if (fit < Lwant) Lwant = (int)fit;
				}
				int Ls = Lwant;
				int mcsp;
				{
					long long gg = c.rate_out, bb = c.rate_out2;
					while (bb) { const long long t = gg % bb; gg = bb; bb = t; }
					long long per = c.rate_out / gg;
					while (per % 8) per *= 2;
					if ((long long)T >= 48ll * Lwant) {
						// long runs: a stream's chunks fill whole waves.  (T / 2720 + 2 = 130 chunks of the wbfm shape were two
						// full waves and a third with ONE busy lane that issued every instruction of the walk again:
						// SQ_INSTS_VALU 23.8 lane-operations per sample against 18.4 with packed lanes.)
						const int n64 = (int)(((long long)T + 32ll * Lwant) / (64ll * Lwant));
						const int lanes = 64 * (n64 < 1 ? 1 : n64);
						long long L0 = ((long long)T + lanes - 1) / lanes;
						L0 = per <= L0 ? per * ((L0 + per - 1) / per) : ((L0 + 7) & ~7ll);
						Ls = (int)L0;
						mcsp = lanes;
					} else {
						if (per <= Lwant) Ls = (int)(per * ((Lwant + per / 2) / per));
						mcsp = T / Ls + 2;
					}
				}
				if (slim) {
					// Round 6: the tail as a FIFTH wave per SIMD beside the next step's front end (staged_kernels.h, k_deemph_lpr_slim):
					// the plan - every lane's stretch, the run's totals - by a small kernel in front, then one-wave workgroups of
					// 32 registers and no LDS.  g / chunks is a multiply-high whose magic number is checked over the whole grid once.
					const size_t lanes_all = (size_t)S * mcsp;
					if (lanes_all + 64 < (1ull << 31)) {
						if (h->slim_magic_for != mcsp || h->slim_magic_upto < lanes_all + 64) {
							const uint32_t M = (uint32_t)(0x100000000ull / (uint32_t)mcsp) + 1u;
							bool ok = true;
							for (uint64_t g = 0; g < lanes_all + 64 && ok; g++) ok = (uint32_t)((g * M) >> 32) == (uint32_t)(g / (uint32_t)mcsp);
							h->slim_magic_for = ok ? mcsp : -1; h->slim_magic = M; h->slim_magic_upto = lanes_all + 64;
						}
						if (h->slim_magic_for == mcsp) {
							if (lanes_all + 64 > h->slim_plan_cap) {
								if (h->d_slim_plan) { HIP_TRY(hipStreamSynchronize(q)); HIP_TRY(hipFree(h->d_slim_plan)); }
								h->d_slim_plan = nullptr; h->slim_plan_cap = 0;
								HIP_TRY(hipMalloc(&h->d_slim_plan, (lanes_all + 64) * sizeof(SlimPlan)));
								h->slim_plan_cap = lanes_all + 64;
							}
							next_dst(&lpr_dst, &lpr_ds);
							if (lpr_dst != final_dst) return -EFAULT;  // routing bug
							const int vec16 = (uintptr_t)lpr_dst % 16 == 0 && lpr_ds % 8 == 0 && !h->opt.lpr_scalar_stores;
							const unsigned gw = (unsigned)((lanes_all + 63) / 64);
							k_lpr_slim_plan<<<gw, 64, 0, q>>>(cur, cur_stride, T, cnt, S, c.deemph_a, mcsp, Ls, lpr_dst, lpr_ds, c.rate_out, c.rate_out2,
							                                 sin, sout, cnt_dst, h->d_slim_plan);
#define RTLFM_LPR_SLIM(MM) k_deemph_lpr_slim<MM><<<gw, 64, 0, q>>>(cur, cur_stride, S, st, mcsp, h->slim_magic, Ws, lpr_dst, lpr_ds, c.rate_out, c.rate_out2, \
							sin, sout, vec16, h->d_slim_plan, h->opt.lpr_slim_prio)
							if (M == 2) RTLFM_LPR_SLIM(2); else if (M == 1) RTLFM_LPR_SLIM(1); else RTLFM_LPR_SLIM(0);
#undef RTLFM_LPR_SLIM
							RTLFM_DBG_SYNC("one pass (lpr, slim)");
							return 0;
						}
					}
				}
				if ((size_t)mcsp > (size_t)h->lpr_chunks_cap) {
					if (h->d_lpr_chunks) { HIP_TRY(hipStreamSynchronize(q)); HIP_TRY(hipFree(h->d_lpr_chunks)); }
					h->d_lpr_chunks = nullptr; h->lpr_chunks_cap = 0;
					HIP_TRY(hipMalloc(&h->d_lpr_chunks, (size_t)S * mcsp * sizeof(LprChunk)));
					h->lpr_chunks_cap = mcsp;
				}
				next_dst(&lpr_dst, &lpr_ds);
				if (lpr_dst != final_dst) return -EFAULT;  // routing bug
				// outputs leave in 64-byte pieces through LDS (2; any row alignment), else in 16-byte groups where the rows allow
				// it (1), else one by one (staged_kernels.h, LprSink)
				int lpr_vec = (uintptr_t)lpr_dst % 16 == 0 && lpr_ds % 8 == 0 && !h->opt.lpr_scalar_stores;
				if (h->opt.lpr_ring && !h->opt.lpr_scalar_stores) lpr_vec = 2;
				const int nthr = h->opt.lpr_threads;
				const size_t lpr_lds = lpr_vec == 2 ? (size_t)nthr * kLprRingStride * sizeof(int16_t) : 0;
				// a workgroup owns whole streams: 256 / chunks of them, or one with a loop over its chunks
				const int spw = mcsp >= nthr ? 1 : nthr / mcsp;
				const unsigned gsp = (unsigned)((S + spw - 1) / spw);
#define RTLFM_SPEC_LPR(MM) k_deemph_spec_lpr<MM><<<gsp, nthr, lpr_lds, q>>>(cur, cur_stride, T, cnt, S, st, mcsp, Ls, Ws, lpr_dst, lpr_ds, \
				c.rate_out, c.rate_out2, sin, sout, h->d_lpr_chunks, lpr_vec, cnt_dst)
				if (M == 2) RTLFM_SPEC_LPR(2); else if (M == 1) RTLFM_SPEC_LPR(1); else RTLFM_SPEC_LPR(0);
#undef RTLFM_SPEC_LPR
				RTLFM_DBG_SYNC("one pass (lpr)");
				return 0;
			}
			// the four passes, for every stream
			// chunk length: each of the passes A1 and C is one chunk long in time; a chunk must be
			// long enough for the interval to contract (~100 samples) and, normally, to merge
			const int L = T >= 8192 ? 1024 : 512;
			const int mc = T / L + 2;
			if ((size_t)mc > (size_t)h->deemph_chunks) {
				if (h->d_deemph_tab) { HIP_TRY(hipStreamSynchronize(q)); HIP_TRY(hipFree(h->d_deemph_tab)); HIP_TRY(hipFree(h->d_deemph_inc)); }
				h->d_deemph_tab = nullptr; h->d_deemph_inc = nullptr; h->deemph_chunks = 0;
				HIP_TRY(hipMalloc(&h->d_deemph_tab, (size_t)S * mc * sizeof(DeemphChunk)));
				HIP_TRY(hipMalloc(&h->d_deemph_inc, (size_t)S * mc * sizeof(uint32_t)));
				h->deemph_chunks = mc;
			}
			if (fuse_lpr && (size_t)mc > (size_t)h->lpr_chunks_cap) {
				if (h->d_lpr_chunks) { HIP_TRY(hipStreamSynchronize(q)); HIP_TRY(hipFree(h->d_lpr_chunks)); }
				h->d_lpr_chunks = nullptr; h->lpr_chunks_cap = 0;
				HIP_TRY(hipMalloc(&h->d_lpr_chunks, (size_t)S * mc * sizeof(LprChunk)));
				h->lpr_chunks_cap = mc;
			}
			const int mcs = mc;
			if (fuse_lpr) next_dst(&lpr_dst, &lpr_ds);
			const int lpr_vec = fuse_lpr && (uintptr_t)lpr_dst % 16 == 0 && lpr_ds % 8 == 0 && !h->opt.lpr_scalar_stores;
			int lpc = 8;  // lanes per chunk in pass A2: the contracted interval (<= 2a + 2 states) must fit
			while (lpc < 2 * c.deemph_a + 3) lpc *= 2;
			const size_t per_wave = 64 / lpc;
			const unsigned ga = (unsigned)(((size_t)S * mcs + per_wave - 1) / per_wave), gc = (unsigned)(((size_t)S * mcs + 63) / 64);
#define RTLFM_DEEMPH_SCAN(MM)                                                                                       \
	do {                                                                                                            \
		k_deemph_scan_a1<MM><<<gc, 64, 0, q>>>(cur, cur_stride, T, cnt, S, st, mcs, L, lpc, h->d_deemph_tab);        \
		RTLFM_DBG_SYNC("a1");                                                                                          \
		k_deemph_scan_a2<MM><<<ga, 64, 0, q>>>(cur, cur_stride, T, cnt, S, st, mcs, L, lpc, h->d_deemph_tab);        \
		RTLFM_DBG_SYNC("a2");                                                                                          \
		k_deemph_scan_b<MM><<<gc, 64, 0, q>>>(cur, cur_stride, T, cnt, S, st, mcs, L, h->d_deemph_tab,               \
		                                       h->d_deemph_inc, sin, sout);                                         \
		RTLFM_DBG_SYNC("b");                                                                                           \
		if (fuse_lpr) {                                                                                              \
			k_deemph_scan_c_lpr<MM><<<gc, 64, 0, q>>>(cur, cur_stride, T, cnt, S, st, mcs, L, h->d_deemph_inc, lpr_dst, \
			                                           lpr_ds, c.rate_out, c.rate_out2, sin, sout, h->d_lpr_chunks, lpr_vec); \
			k_lpr_fixup<<<gc, 64, 0, q>>>(cur, cur_stride, T, cnt, S, mcs, L, h->d_lpr_chunks, lpr_dst, lpr_ds,        \
			                              c.rate_out, c.rate_out2, sin, sout, h->d_cnt2);                            \
		} else {                                                                                                     \
			k_deemph_scan_c<MM><<<gc, 64, 0, q>>>(cur, cur_stride, T, cnt, S, st, mcs, L, h->d_deemph_inc, sout);      \
		}                                                                                                            \
		RTLFM_DBG_SYNC("c");                                                                                           \
	} while (0)
			if (M == 2) RTLFM_DEEMPH_SCAN(2);
			else if (M == 1) RTLFM_DEEMPH_SCAN(1);
			else RTLFM_DEEMPH_SCAN(0);
#undef RTLFM_DEEMPH_SCAN
		} else if (M == 2) k_deemph<2><<<grid, 64, 0, q>>>(cur, cur_stride, T, cnt, S, st, sin, sout);
		else if (M == 1) k_deemph<1><<<grid, 64, 0, q>>>(cur, cur_stride, T, cnt, S, st, sin, sout);
		else k_deemph<0><<<grid, 64, 0, q>>>(cur, cur_stride, T, cnt, S, st, sin, sout);
	}
	if (tp.adc) {
		// dc_block_audio_filter (src/rtl_fm.c:1028-1041): sums per buffer, then the smoothing recurrence and the subtraction
		// in one launch (staged_kernels.h, k_adc_smooth_apply); option adc_separate: round 4's three kernels
		k_adc_sums<<<S * nblocks, 256, 0, q>>>(cur, cur_stride, Nblk, D, nblocks, sin, h->d_adc_sums);
		if (!h->opt.adc_separate) {
			k_adc_smooth_apply<<<S * nblocks, 256, (size_t)(nblocks + 1) * sizeof(int32_t), q>>>(cur, cur_stride, Nblk, D, nblocks, c.adc_block_const,
			                                                                                  h->d_adc_sums, sin, sout);
		} else {
			k_adc_smooth<<<grid_for(S, 64), 64, 0, q>>>(h->d_adc_sums, Nblk, D, nblocks, S, c.adc_block_const, sin,
			                                           sout, h->d_adc_avg);
			k_adc_apply<<<grid_for((size_t)S * T), 256, 0, q>>>(cur, cur_stride, Nblk, D, nblocks, S, T, sin,
			                                                  h->d_adc_avg);
		}
	}
	if (tp.lpr && fuse_lpr) {
		cur = lpr_dst; cur_stride = lpr_ds;
		if (d_out_len)
			HIP_TRY(hipMemcpyAsync(d_out_len, h->d_cnt2, S * sizeof(int32_t), hipMemcpyDeviceToDevice, q));
		return 0;
	}
	if (tp.lpr) {
		int16_t *d; size_t ds;
		next_dst(&d, &ds);
		int maxout = (int)(((long long)c.rate_out - 1 + (long long)T * c.rate_out2) / c.rate_out);
		k_low_pass_real<<<grid_for((size_t)S * (maxout + 1)), 256, 0, q>>>(
		    cur, cur_stride, d, ds, T, cnt, S, c.rate_out, c.rate_out2, sin, sout, h->d_cnt2);
		cur = d; cur_stride = ds;
		if (d_out_len)
			HIP_TRY(hipMemcpyAsync(d_out_len, h->d_cnt2, S * sizeof(int32_t), hipMemcpyDeviceToDevice, q));
		return 0;
	}
	if (tp.arb) {
		// arbitrary_resample per buffer on whatever result_len it has (src/rtl_fm.c:1168-1177, 1270)
		int16_t *d; size_t ds;
		next_dst(&d, &ds);
		const int nlo = Nblk / D, nhi = (Nblk % D) ? nlo + 1 : nlo;
		const int l2lo = (int)((long long)nlo * c.rate_out2 / c.rate_out);
		const int l2hi = (int)((long long)nhi * c.rate_out2 / c.rate_out);
		const bool any_up = nlo < l2lo || nhi < l2hi, any_down = !(nlo < l2lo) || !(nhi < l2hi);
		const ArbPlan ap{Nblk, D, nlo, l2lo, (int)((long long)(nlo + 1) * c.rate_out2 / c.rate_out)};
		if (any_up)
			k_arb_upsample<<<grid_for((size_t)S * nblocks * l2hi), 256, 0, q>>>(
			    cur, cur_stride, d, ds, ap, l2hi, nblocks, S, sin, h->d_cnt2);
		if (any_down)
			k_arb_downsample<<<grid_for((size_t)S * nblocks, 64), 64, 0, q>>>(
			    cur, cur_stride, d, ds, ap, nblocks, S, sin, h->d_cnt2);
		cur = d; cur_stride = ds;
		if (cur != final_dst) return -EFAULT;  // routing bug
		if (d_out_len)
			HIP_TRY(hipMemcpyAsync(d_out_len, h->d_cnt2, S * sizeof(int32_t), hipMemcpyDeviceToDevice, q));
		return 0;
	}
	if (cur != final_dst) return -EFAULT;  // routing bug
	if (d_out_len) {
		if (varcnt)
			HIP_TRY(hipMemcpyAsync(d_out_len, h->d_cnt[par], S * sizeof(int32_t), hipMemcpyDeviceToDevice, q));
		else
			k_fill_cnt<<<grid_for(S, 64), 64, 0, q>>>(d_out_len, S, T);
	}
	return 0;
}

// Passes first .. passes - 1 where `first` is handed a length that is not a multiple of four elements, and everything
// up to mode_demod() behind them (staged_kernels.h, k_fifth_irregular); then the ordinary audio tail.  `cur` holds the
// level in front of pass `first` (n_in samples per buffer, buffers back to back).
static int run_tail(rtlfm_gpu *h, const TailPlan &tp, int16_t *cur, size_t cur_stride, int T, bool varcnt,
                    int nblocks, int Nblk, int D, int16_t *final_dst, size_t final_stride, int32_t *d_out_len);
static int run_irregular_rest(rtlfm_gpu *h, const uint32_t *cur, size_t xstride, int first, int nblocks, int16_t *d_out,
                              size_t out_stride, int32_t *d_out_len, const uint8_t *d_iq, size_t iq_bytes)
{
	const rtlfm_cfg &c = h->cfg;
	const int S = h->nstreams;
	hipStream_t q = h->stream;
	TailPlan tp = plan_tail(h, nblocks);
	if (tp.any()) {
		int r = ensure_res_buffers(h, d_iq, iq_bytes);
		if (r < 0) return r;
	}
	int16_t *dd; size_t dds;
	tail_route(h, tp, d_out, out_stride, &dd, &dds);
	IrregularParams ip{};
	ip.X = cur; ip.xstride = xstride; ip.n_in = (int)((c.block_len / 2) >> first); ip.nblocks = nblocks; ip.nstreams = S;
	ip.first = first; ip.passes = c.downsample_passes; ip.lp_len = (int)(c.block_len >> c.downsample_passes);
	ip.fir = c.comp_fir_size == 9 ? 1 : 0; ip.mode = c.mode; ip.variant = c.custom_atan; ip.output_scale = c.output_scale;
	ip.squelch_level = c.squelch_level; ip.report_levels = c.report_levels; ip.omit_dc_fix = c.dc_block_raw;
	ip.lut = h->d_lut; ip.R = dd; ip.rstride = dds;
	ip.levels = (c.squelch_level || c.report_levels) ? h->d_levels : nullptr;
	ip.sin = h->st[h->st_cur]; ip.sout = h->st[(h->st_cur + 1) % 3];
	if (2 * ip.n_in + 8 > kIrregularMaxElems) return -ENOTSUP;
	k_fifth_irregular<<<(unsigned)S, 64, 0, q>>>(ip);  // one wave per stream
	if (c.mode == RTLFM_MODE_RAW) {
		if (d_out_len) k_fill_cnt<<<grid_for(S, 64), 64, 0, q>>>(d_out_len, S, nblocks * ip.lp_len);
		return 0;
	}
	const int Nblk = ip.lp_len / 2, T = nblocks * Nblk;
	if (T == 0) {  // every buffer is down to one element or none: fm_demod() has nothing to pair
		if (d_out_len) k_fill_cnt<<<grid_for(S, 64), 64, 0, q>>>(d_out_len, S, 0);
		return 0;
	}
	return run_tail(h, tp, dd, dds, T, false, nblocks, Nblk, 1, d_out, out_stride, d_out_len);
}

static int run_staged(rtlfm_gpu *h, const uint8_t *d_iq, size_t stream_stride, int nblocks, int16_t *d_out,
                      size_t out_stride, int32_t *d_out_len)
{
	const rtlfm_cfg &c = h->cfg;
	const int S = h->nstreams;
	const uint32_t L = c.block_len;
	const int N0 = (int)(L / 2);
	hipStream_t q = h->stream;
	int r = ensure_work_buffers(h);
	if (r < 0) return r;
	const state_t *sin = h->st[h->st_cur];
	state_t *sout = h->st[(h->st_cur + 1) % 3];
	std::pair<hipEvent_t, hipEvent_t> ev;

	r = timing_begin(h, ev);
	if (r < 0) return r;
	// --- rtlsdr_callback: convert, raw DC block, rotate (src/rtl_fm.c:1326-1338)
	const int2 *rdc = nullptr;
	if (c.dc_block_raw) {
		k_rdc_sums<<<S * nblocks, 256, 0, q>>>(d_iq, stream_stride, L, nblocks, h->d_sums);
		k_rdc_smooth<<<(unsigned)S, 64, 0, q>>>(h->d_sums, L, nblocks, S, c.rdc_block_const, sin, sout,
		                                           h->d_rdc_avg);
		rdc = h->d_rdc_avg;
	}
	k_convert<<<grid_for((size_t)S * nblocks * (L / 16)), 256, 0, q>>>(d_iq, stream_stride, L, nblocks, S,
	                                                                  h->bufA, h->xstride,
	                                                                  c.offset_tuning ? 0 : 1, rdc);
	uint32_t *cur = h->bufA, *oth = h->bufB;
	// --- decimation (src/rtl_fm.c:1187-1202)
	int T;          // decimated complex samples per stream (upper bound if varcnt)
	int Nblk, D;    // block geometry for "first output of a block" tests
	bool varcnt = false;
	if (c.downsample_passes > 0) {
		const int first_irr = first_irregular_pass(c);
		for (int p = 0; p < first_irr; p++) {
			int N = N0 >> p;
			k_fifth<<<grid_for((size_t)S * nblocks * (N / 2)), 256, 0, q>>>(cur, oth, h->xstride, N, nblocks, S,
			                                                              p, sin, sout);
			std::swap(cur, oth);
		}
		if (first_irr < c.downsample_passes) {
			r = timing_end(h, ev);
			if (r < 0) return r;
			return run_irregular_rest(h, cur, h->xstride, first_irr, nblocks, d_out, out_stride, d_out_len, d_iq,
			                          iq_extent(h, stream_stride, nblocks));
		}
		Nblk = N0 >> c.downsample_passes;
		D = 1;
		T = nblocks * Nblk;
		if (c.comp_fir_size == 9) {
			k_fir9<<<grid_for((size_t)S * T), 256, 0, q>>>(cur, oth, h->xstride, T, S, c.downsample_passes,
			                                              sin, sout);
			std::swap(cur, oth);
		}
	} else {
		Nblk = N0;
		D = c.downsample;
		int Tin = nblocks * N0;
		int maxout = Tin / D + 1;
		k_boxcar<<<grid_for((size_t)S * (maxout + 1)), 256, 0, q>>>(cur, oth, h->xstride, Tin, S, D, sin, sout,
		                                                          h->d_cnt[h->step & 1]);
		std::swap(cur, oth);
		varcnt = (N0 % D) != 0;
		T = varcnt ? maxout : Tin / D;
	}
	r = timing_end(h, ev);
	if (r < 0) return r;
	// --- power squelch (src/rtl_fm.c:1204-1215)
	if (c.squelch_level || c.report_levels)
		k_squelch_rms<<<S * nblocks, 256, 0, q>>>(cur, h->xstride, Nblk, D, nblocks, sin, c.squelch_level,
		                                         c.dc_block_raw, h->d_mute, h->d_levels);
	if (c.squelch_level) {
		k_squelch_hits<<<grid_for(S, 64), 64, 0, q>>>(h->d_mute, nblocks, S, sin, sout);
		k_squelch_zero<<<S * nblocks, 256, 0, q>>>(cur, h->xstride, Nblk, D, nblocks, S, T, sin,
		                                                     h->d_mute);
	}
	// --- mode_demod (src/rtl_fm.c:1256-1259)
	TailPlan tp = plan_tail(h, nblocks);
	if (tp.any()) {
		r = ensure_res_buffers(h, d_iq, iq_extent(h, stream_stride, nblocks));
		if (r < 0) return r;
	}
	int16_t *dd; size_t dds;
	tail_route(h, tp, d_out, out_stride, &dd, &dds);
	const int32_t *cnt = varcnt ? h->d_cnt[h->step & 1] : nullptr;
	if (c.mode == RTLFM_MODE_FM)
		k_fm_demod<<<S * nblocks, 256, 0, q>>>(cur, h->xstride, dd, dds, T, S, Nblk, D, nblocks,
		                                         c.custom_atan, h->d_lut, cnt, sin, sout);
	else
		k_simple_demod<<<grid_for((size_t)S * T), 256, 0, q>>>(cur, h->xstride, dd, dds, T, S, c.mode,
		                                                     c.output_scale, cnt);
	if (c.mode == RTLFM_MODE_RAW) {
		if (d_out_len) {
			if (varcnt) {
				HIP_TRY(hipMemcpyAsync(d_out_len, h->d_cnt[h->step & 1], S * sizeof(int32_t), hipMemcpyDeviceToDevice, q));
				k_scale_cnt<<<grid_for(S, 64), 64, 0, q>>>(d_out_len, S, 2, 1);
			} else {
				k_fill_cnt<<<grid_for(S, 64), 64, 0, q>>>(d_out_len, S, 2 * T);
			}
		}
		return 0;
	}
	return run_tail(h, tp, dd, dds, T, varcnt, nblocks, Nblk, D, d_out, out_stride, d_out_len);
}


// shader clock held during the last fused fifth_order launch: every wave stamps s_memtime (shader
// clock) and s_memrealtime (100 MHz) at its first and last instruction
static int read_clock_stamps(rtlfm_gpu *h, double *mhz, double *span_ms, int *waves)
{
	HIP_TRY(sync_all(h));
	if (!h->fws.stamps || h->fws.stamp_last <= 0) return -ENODATA;
	std::vector<unsigned long long> st((size_t)h->fws.stamp_last * 4);
	HIP_TRY(hipMemcpy(st.data(), h->fws.stamp_half(0), st.size() * 8, hipMemcpyDeviceToHost));
	double sum = 0; int n = 0; unsigned long long t0 = ~0ull, t1 = 0;
	for (int w = 0; w < h->fws.stamp_last; w++) {
		double dc = (double)(st[w * 4 + 1] - st[w * 4]), dr = (double)(st[w * 4 + 3] - st[w * 4 + 2]);
		if (dr > 0) { sum += dc / dr * 100.0; n++; }
		if (st[w * 4 + 2] < t0) t0 = st[w * 4 + 2];
		if (st[w * 4 + 3] > t1) t1 = st[w * 4 + 3];
	}
	if (!n) return -ENODATA;
	if (mhz) *mhz = sum / n;
	if (span_ms) *span_ms = (double)(t1 - t0) / 1e5;
	if (waves) *waves = n;
	return 0;
}
static int report_clock_stamps(rtlfm_gpu *h)
{
	double mhz = 0, span = 0; int n = 0;
	int r = read_clock_stamps(h, &mhz, &span, &n);
	if (r < 0) return r == -ENODATA ? 0 : r;
	fprintf(stderr, "rtlfm_hip[debug]: mean shader clock %.0f MHz over %d waves, kernel span %.3f ms\n", mhz, n, span);
	return 0;
}

extern "C" int rtlfm_gpu_clock_probe(rtlfm_gpu *h, int on)
{
	if (!h) return -EINVAL;
	h->fws.want_stamps = on != 0;
	return 0;
}
extern "C" int rtlfm_gpu_clock_read(rtlfm_gpu *h, double *shader_mhz, double *span_ms)
{
	if (!h) return -EINVAL;
	HIP_TRY(hipSetDevice(h->device));
	return read_clock_stamps(h, shader_mhz, span_ms, nullptr);
}

// dc_block_raw_filter needs every buffer's mean before its first sample: one more pass over the input (the
// per-buffer sums), the smoothing recurrence over a stream's buffers; the averages then ride on the front end
// (fused_kernel.h: the MFMA accumulators' start values; boxcar_kernel.h: a correction of the prefix sums)
static const int2 *rdc_prepass(rtlfm_gpu *h, const uint8_t *d_iq, size_t stream_stride, int nblocks)
{
	const rtlfm_cfg &c = h->cfg;
	if (!c.dc_block_raw) return nullptr;
	const int S = h->nstreams;
	if (c.block_len < 8192) {  // buffers shorter than a tile: a wave per buffer (staged_kernels.h)
		const size_t total = (size_t)S * nblocks;
		k_rdc_sums_small<<<(unsigned)((total + 3) / 4), 256, 0, h->stream>>>(d_iq, stream_stride, c.block_len, nblocks, total, h->d_sums);
	} else
		k_rdc_sums_wide<<<(unsigned)((size_t)S * nblocks), 256, 0, h->stream>>>(d_iq, stream_stride, c.block_len, nblocks, h->d_sums);
	k_rdc_smooth<<<(unsigned)S, 64, 0, h->stream>>>(h->d_sums, c.block_len, nblocks, S, c.rdc_block_const, h->st[h->st_cur],
	                                                  h->st[(h->st_cur + 1) % 3], h->d_rdc_avg);
	return h->d_rdc_avg;
}

static int run_fused(rtlfm_gpu *h, const uint8_t *d_iq, size_t stream_stride, int nblocks, int16_t *d_out,
                     size_t out_stride, int32_t *d_out_len)
{
	const rtlfm_cfg &c = h->cfg;
	const int S = h->nstreams;
	hipStream_t q = h->stream;
	const state_t *sin = h->st[h->st_cur];
	state_t *sout = h->st[(h->st_cur + 1) % 3];
	TailPlan tp = plan_tail(h, nblocks);
	if (tp.any()) {
		int r = ensure_res_buffers(h, d_iq, iq_extent(h, stream_stride, nblocks));
		if (r < 0) return r;
	}
	int16_t *dd; size_t dds;
	tail_route(h, tp, d_out, out_stride, &dd, &dds);
	std::pair<hipEvent_t, hipEvent_t> ev;
	int r = timing_begin(h, ev);
	if (r < 0) return r;
	const int2 *rdc = rdc_prepass(h, d_iq, stream_stride, nblocks);
	const bool sq = c.squelch_level || c.report_levels;  // the front end takes rms()'s sums, k_squelch_apply decides (fused_kernel.h)
	// (a tail in line leaves the next front end alone: the plan of a front end that nothing runs beside)
	h->fws.tail_follows = (tp.any() && !arb_tail_in_line(h, tp)) || sq;
	if (sq) HIP_TRY(hipMemsetAsync(h->d_sq_sums, 0, (size_t)S * nblocks * 2 * sizeof(uint32_t), q));
	r = fused::launch(h->fws, c, S, d_iq, stream_stride, nblocks, dd, dds, sin, sout, h->d_lut, q, nullptr, 0, rdc,
	                  sq ? h->d_sq_sums : nullptr);
	if (r < 0) return r;
	r = timing_end(h, ev);
	if (r < 0) return r;
	if (sq) {
		const int Nb = (int)((c.block_len / 2) >> c.downsample_passes);
		k_squelch_apply<<<(unsigned)S, 256, (size_t)nblocks * sizeof(int32_t), q>>>(
		    h->d_sq_sums, dd, dds, Nb, 1, nblocks, S, c.squelch_level, c.dc_block_raw, c.mode == RTLFM_MODE_FM ? 1 : 0, nullptr,
		    nblocks * Nb, sin, sout, h->d_levels);
	}
	// bit 2 with bit 8: report the stamps of every launch (synchronises, so the GPU idles between
	// launches and clocks up); bit 2 alone with bit 16: only when timing_read() asks, i.e. the last
	// launch of an uninterrupted sequence
	if (h->fws.stamps && (h->fws.debug & 2) && !(h->fws.debug & 16)) {
		int r2 = report_clock_stamps(h);
		if (r2 < 0) return r2;
	}
	const int Nblk = (int)((c.block_len / 2) >> c.downsample_passes);
	return run_tail(h, tp, dd, dds, nblocks * Nblk, false, nblocks, Nblk, 1, d_out, out_stride, d_out_len);
}

// The fused front end in emit mode + staged kernels for what it does not do itself: with up to
// six passes it emits the decimated, FIR-compensated IQ (what full_demod() hands on after
// src/rtl_fm.c:1202); with 7..10 passes the /64 IQ, and the staged kernels run the remaining
// passes and the FIR on 1/64 of the data.  Then, as in full_demod(): squelch (:1204-1215),
// mode_demod incl. -M raw (:1256-1259), audio tail.  The work buffer has the staged path's
// layout for that level (a stream's buffers back to back).
static int run_fused_emit(rtlfm_gpu *h, const uint8_t *d_iq, size_t stream_stride, int nblocks, int16_t *d_out,
                          size_t out_stride, int32_t *d_out_len)
{
	const rtlfm_cfg &c = h->cfg;
	const int S = h->nstreams;
	const int N0 = (int)(c.block_len / 2);
	const int level = c.downsample_passes < fused::kMaxP ? c.downsample_passes : fused::kMaxP;
	hipStream_t q = h->stream;
	const state_t *sin = h->st[h->st_cur];
	state_t *sout = h->st[(h->st_cur + 1) % 3];
	// -M raw behind up to six passes, no squelch: the emitted IQ is the output, straight into the caller's rows where they
	// take 16-byte stores (as run_boxfused_emit)
	const bool raw_direct = c.mode == RTLFM_MODE_RAW && !c.squelch_level && !c.report_levels && c.downsample_passes <= fused::kMaxP &&
	                        !((uintptr_t)d_out & 15) && !(out_stride & 7);
	if (!raw_direct) {
		int r0 = ensure_deep_buffers(h, d_iq, iq_extent(h, stream_stride, nblocks));
		if (r0 < 0) return r0;
	}
	TailPlan tp = plan_tail(h, nblocks);
	if (tp.any() && !raw_direct) {  // (-M raw has no audio tail: full_demod returns behind raw_demod, src/rtl_fm.c:1257-1259)
		int r = ensure_res_buffers(h, d_iq, iq_extent(h, stream_stride, nblocks));
		if (r < 0) return r;
	}
	std::pair<hipEvent_t, hipEvent_t> ev;
	int r = timing_begin(h, ev);
	if (r < 0) return r;
	h->fws.tail_follows = !raw_direct;  // kernels follow on this stream and, with a tail, on the tail's (as run_boxfused_emit)
	const int2 *rdc = rdc_prepass(h, d_iq, stream_stride, nblocks);  // -E rdc in front of -M raw, the squelch, 7-10 passes
	r = fused::launch(h->fws, c, S, d_iq, stream_stride, nblocks, nullptr, 0, sin, sout, h->d_lut, q,
	                  raw_direct ? reinterpret_cast<uint32_t *>(d_out) : h->deepA, raw_direct ? out_stride / 2 : h->deep_stride, rdc);
	if (r < 0) return r;
	r = timing_end(h, ev);
	if (r < 0) return r;
	if (raw_direct) {
		if (d_out_len) k_fill_cnt<<<grid_for(S, 64), 64, 0, q>>>(d_out_len, S, 2 * nblocks * (N0 >> c.downsample_passes));
		return 0;
	}
	uint32_t *cur = h->deepA, *oth = h->deepB;
	const int first_irr = first_irregular_pass(c);
	// seven and more passes on buffers they divide: the passes beyond six, generic_fir and (without a squelch) the
	// demodulator in ONE launch, a workgroup per (stream, buffer) in LDS (staged_kernels.h, k_deep_rest)
	if (c.downsample_passes > level && first_irr >= c.downsample_passes && h->opt.deep_rest) {
		int kk[6];
		deep_rest_tails(c.downsample_passes, c.comp_fir_size == 9, kk);
		const int n6 = N0 >> level, nF = N0 >> c.downsample_passes;
		if (n6 >= 2 * kk[0] && nF - kk[c.downsample_passes - level] >= 3) {
			const bool demod_here = !c.squelch_level && !c.report_levels && c.mode != RTLFM_MODE_RAW;
			DeepRestParams dp{};
			dp.X = cur; dp.xstride = h->deep_stride; dp.n6 = n6; dp.nblocks = nblocks; dp.nstreams = S; dp.passes = c.downsample_passes;
			dp.fir = c.comp_fir_size == 9 ? 1 : 0; dp.kt = kk[0];
			dp.mode = c.mode; dp.variant = c.custom_atan; dp.output_scale = c.output_scale; dp.lut = h->d_lut;
			dp.sin = sin; dp.sout = sout;
			const int T2 = nblocks * nF;
			int16_t *dd2; size_t dds2;
			tail_route(h, tp, d_out, out_stride, &dd2, &dds2);
			if (demod_here) { dp.R = dd2; dp.rstride = dds2; }
			else { dp.Y = oth; dp.ystride = h->deep_stride; }
			const size_t lds_b = (size_t)2 * (kk[0] + n6 + 8) * sizeof(uint32_t);
			k_deep_rest<<<(unsigned)((size_t)S * nblocks), 256, lds_b, q>>>(dp);
			if (demod_here) return run_tail(h, tp, dd2, dds2, T2, false, nblocks, nF, 1, d_out, out_stride, d_out_len);
			std::swap(cur, oth);
			// the squelch, -L, -M raw on the final level, as below
			const int Nblk2 = nF;
			if (c.squelch_level || c.report_levels)
				k_squelch_rms<<<S * nblocks, 256, 0, q>>>(cur, h->deep_stride, Nblk2, 1, nblocks, sin, c.squelch_level,
				                                         c.dc_block_raw, h->d_mute, h->d_levels);
			if (c.squelch_level) {
				k_squelch_hits<<<grid_for(S, 64), 64, 0, q>>>(h->d_mute, nblocks, S, sin, sout);
				k_squelch_zero<<<S * nblocks, 256, 0, q>>>(cur, h->deep_stride, Nblk2, 1, nblocks, S, T2, sin, h->d_mute);
			}
			if (c.mode == RTLFM_MODE_FM)
				k_fm_demod<<<S * nblocks, 256, 0, q>>>(cur, h->deep_stride, dd2, dds2, T2, S, Nblk2, 1, nblocks, c.custom_atan,
				                                         h->d_lut, nullptr, sin, sout);
			else
				k_simple_demod<<<grid_for((size_t)S * T2), 256, 0, q>>>(cur, h->deep_stride, dd2, dds2, T2, S, c.mode,
				                                                      c.output_scale, nullptr);
			if (c.mode == RTLFM_MODE_RAW) {
				if (d_out_len) k_fill_cnt<<<grid_for(S, 64), 64, 0, q>>>(d_out_len, S, 2 * T2);
				return 0;
			}
			return run_tail(h, tp, dd2, dds2, T2, false, nblocks, Nblk2, 1, d_out, out_stride, d_out_len);
		}
	}
	for (int p = level; p < c.downsample_passes && p < first_irr; p++) {
		const int N = N0 >> p;
		k_fifth<<<grid_for((size_t)S * nblocks * (N / 2)), 256, 0, q>>>(cur, oth, h->deep_stride, N, nblocks, S, p,
		                                                              sin, sout);
		std::swap(cur, oth);
	}
	if (first_irr < c.downsample_passes)  // nine or ten passes on buffers they do not divide: the reference's own loops from here
		return run_irregular_rest(h, cur, h->deep_stride, first_irr, nblocks, d_out, out_stride, d_out_len, d_iq,
		                          iq_extent(h, stream_stride, nblocks));
	const int Nblk = N0 >> c.downsample_passes;
	const int T = nblocks * Nblk;
	if (c.downsample_passes > level && c.comp_fir_size == 9) {
		k_fir9<<<grid_for((size_t)S * T), 256, 0, q>>>(cur, oth, h->deep_stride, T, S, c.downsample_passes, sin, sout);
		std::swap(cur, oth);
	}
	if (c.squelch_level || c.report_levels)
		k_squelch_rms<<<S * nblocks, 256, 0, q>>>(cur, h->deep_stride, Nblk, 1, nblocks, sin, c.squelch_level,
		                                         c.dc_block_raw, h->d_mute, h->d_levels);
	if (c.squelch_level) {
		k_squelch_hits<<<grid_for(S, 64), 64, 0, q>>>(h->d_mute, nblocks, S, sin, sout);
		k_squelch_zero<<<S * nblocks, 256, 0, q>>>(cur, h->deep_stride, Nblk, 1, nblocks, S, T, sin,
		                                                     h->d_mute);
	}
	int16_t *dd; size_t dds;
	tail_route(h, tp, d_out, out_stride, &dd, &dds);
	if (c.mode == RTLFM_MODE_FM)
		k_fm_demod<<<S * nblocks, 256, 0, q>>>(cur, h->deep_stride, dd, dds, T, S, Nblk, 1, nblocks, c.custom_atan,
		                                         h->d_lut, nullptr, sin, sout);
	else
		k_simple_demod<<<grid_for((size_t)S * T), 256, 0, q>>>(cur, h->deep_stride, dd, dds, T, S, c.mode,
		                                                     c.output_scale, nullptr);
	if (c.mode == RTLFM_MODE_RAW) {
		if (d_out_len) k_fill_cnt<<<grid_for(S, 64), 64, 0, q>>>(d_out_len, S, 2 * T);
		return 0;
	}
	return run_tail(h, tp, dd, dds, T, false, nblocks, Nblk, 1, d_out, out_stride, d_out_len);
}

// The boxcar (low_pass) front end in one launch; output counts may differ per buffer and
// per stream (d_cnt), which the order-insensitive tail stages accept (run_tail, varcnt).
static int run_boxfused(rtlfm_gpu *h, const uint8_t *d_iq, size_t stream_stride, int nblocks, int16_t *d_out,
                        size_t out_stride, int32_t *d_out_len)
{
	const rtlfm_cfg &c = h->cfg;
	const int S = h->nstreams;
	hipStream_t q = h->stream;
	const state_t *sin = h->st[h->st_cur];
	state_t *sout = h->st[(h->st_cur + 1) % 3];
	TailPlan tp = plan_tail(h, nblocks);
	if (tp.any()) {
		int r = ensure_res_buffers(h, d_iq, iq_extent(h, stream_stride, nblocks));
		if (r < 0) return r;
	}
	int16_t *dd; size_t dds;
	tail_route(h, tp, d_out, out_stride, &dd, &dds);
	std::pair<hipEvent_t, hipEvent_t> ev;
	int r = timing_begin(h, ev);
	if (r < 0) return r;
	const bool sq = c.squelch_level || c.report_levels;  // the front end takes rms()'s sums, k_squelch_apply decides (boxcar_kernel.h, SQ)
	h->fws.tail_follows = tp.any() || sq;
	const int2 *rdc = rdc_prepass(h, d_iq, stream_stride, nblocks);
	if (sq) HIP_TRY(hipMemsetAsync(h->d_sq_sums, 0, (size_t)S * nblocks * 2 * sizeof(uint32_t), q));
	r = boxfused::launch(h->fws, c, S, d_iq, stream_stride, nblocks, dd, dds, h->d_cnt[h->step & 1], sin, sout, q, nullptr, 0, rdc,
	                     sq ? h->d_sq_sums : nullptr);
	if (r < 0) return r;
	r = timing_end(h, ev);
	if (r < 0) return r;
	const int N0 = (int)(c.block_len / 2), D = c.downsample;
	const int Tin = nblocks * N0;
	const bool varcnt = (N0 % D) != 0;
	const int T = varcnt ? Tin / D + 1 : Tin / D;
	if (sq)
		k_squelch_apply<<<(unsigned)S, 256, (size_t)nblocks * sizeof(int32_t), q>>>(
		    h->d_sq_sums, dd, dds, N0, D, nblocks, S, c.squelch_level, c.dc_block_raw, c.mode == RTLFM_MODE_FM ? 1 : 0,
		    varcnt ? h->d_cnt[h->step & 1] : nullptr, T, sin, sout, h->d_levels);
	return run_tail(h, tp, dd, dds, T, varcnt, nblocks, N0, D, d_out, out_stride, d_out_len);
}

// The boxcar front end in emit mode + staged kernels on 1 / D of the data, as full_demod() goes on behind
// low_pass(): power squelch (src/rtl_fm.c:1204-1215), -L levels (:1217-1237), mode_demod incl. -M raw
// (:1006-1009, 1256-1259), audio tail.  The everyday scanner line - rtl_fm -M fm -s 12k -l 50 - is this.
static int run_boxfused_emit(rtlfm_gpu *h, const uint8_t *d_iq, size_t stream_stride, int nblocks, int16_t *d_out,
                             size_t out_stride, int32_t *d_out_len)
{
	const rtlfm_cfg &c = h->cfg;
	const int S = h->nstreams;
	hipStream_t q = h->stream;
	const state_t *sin = h->st[h->st_cur];
	state_t *sout = h->st[(h->st_cur + 1) % 3];
	// -M raw without the squelch: what the launch emits IS the output (raw_demod() copies lowpassed[], src/rtl_fm.c:1002-1009)
	// - it goes straight into the caller's rows (round 5; a copy kernel over 429 M samples cost more than the decimator)
	const bool raw_direct = c.mode == RTLFM_MODE_RAW && !c.squelch_level && !c.report_levels;
	int r = raw_direct ? 0 : ensure_deep_buffers(h, d_iq, iq_extent(h, stream_stride, nblocks));
	if (r < 0) return r;
	TailPlan tp = plan_tail(h, nblocks);
	if (tp.any()) {
		r = ensure_res_buffers(h, d_iq, iq_extent(h, stream_stride, nblocks));
		if (r < 0) return r;
	}
	std::pair<hipEvent_t, hipEvent_t> ev;
	r = timing_begin(h, ev);
	if (r < 0) return r;
	h->fws.tail_follows = !raw_direct;  // kernels follow on this stream and, with a tail, on the tail's
	int32_t *dcnt = h->d_cnt[h->step & 1];
	const int2 *rdc = rdc_prepass(h, d_iq, stream_stride, nblocks);
	uint32_t *emit_to = raw_direct ? reinterpret_cast<uint32_t *>(d_out) : h->deepA;
	const size_t emit_stride = raw_direct ? out_stride / 2 : h->deep_stride;
	r = boxfused::launch(h->fws, c, S, d_iq, stream_stride, nblocks, nullptr, 0, dcnt, sin, sout, q, emit_to, emit_stride, rdc);
	if (r < 0) return r;
	r = timing_end(h, ev);
	if (r < 0) return r;
	const int N0 = (int)(c.block_len / 2), D = c.downsample;
	const int Tin = nblocks * N0;
	const bool varcnt = (N0 % D) != 0;
	const int T = varcnt ? Tin / D + 1 : Tin / D;
	if (raw_direct) {
		if (d_out_len) {
			if (varcnt) {
				HIP_TRY(hipMemcpyAsync(d_out_len, dcnt, S * sizeof(int32_t), hipMemcpyDeviceToDevice, q));
				k_scale_cnt<<<grid_for(S, 64), 64, 0, q>>>(d_out_len, S, 2, 1);
			} else {
				k_fill_cnt<<<grid_for(S, 64), 64, 0, q>>>(d_out_len, S, 2 * T);
			}
		}
		return 0;
	}
	uint32_t *cur = h->deepA;
	if (c.squelch_level || c.report_levels)
		k_squelch_rms<<<S * nblocks, 256, 0, q>>>(cur, h->deep_stride, N0, D, nblocks, sin, c.squelch_level, c.dc_block_raw,
		                                         h->d_mute, h->d_levels);
	if (c.squelch_level) {
		k_squelch_hits<<<grid_for(S, 64), 64, 0, q>>>(h->d_mute, nblocks, S, sin, sout);
		k_squelch_zero<<<S * nblocks, 256, 0, q>>>(cur, h->deep_stride, N0, D, nblocks, S, T, sin, h->d_mute);
	}
	int16_t *dd; size_t dds;
	tail_route(h, tp, d_out, out_stride, &dd, &dds);
	const int32_t *cnt = varcnt ? dcnt : nullptr;
	if (c.mode == RTLFM_MODE_FM)
		k_fm_demod<<<S * nblocks, 256, 0, q>>>(cur, h->deep_stride, dd, dds, T, S, N0, D, nblocks, c.custom_atan, h->d_lut, cnt,
		                                         sin, sout);
	else
		k_simple_demod<<<grid_for((size_t)S * T), 256, 0, q>>>(cur, h->deep_stride, dd, dds, T, S, c.mode, c.output_scale, cnt);
	if (c.mode == RTLFM_MODE_RAW) {
		if (d_out_len) {
			if (varcnt) {
				HIP_TRY(hipMemcpyAsync(d_out_len, dcnt, S * sizeof(int32_t), hipMemcpyDeviceToDevice, q));
				k_scale_cnt<<<grid_for(S, 64), 64, 0, q>>>(d_out_len, S, 2, 1);
			} else {
				k_fill_cnt<<<grid_for(S, 64), 64, 0, q>>>(d_out_len, S, 2 * T);
			}
		}
		return 0;
	}
	return run_tail(h, tp, dd, dds, T, varcnt, nblocks, N0, D, d_out, out_stride, d_out_len);
}

// one execution of a run: everything rtlfm_gpu_run_device does except moving on to the next state copy / step
static int run_device_once(rtlfm_gpu *h, const uint8_t *d_iq, size_t stream_stride, int nblocks,
                           int16_t *d_out, size_t out_stride, int32_t *d_out_len)
{
	rtl_debug::poison_lds(h->stream);  // RTLFM_POISON=1 only
	const size_t S = (size_t)h->nstreams;
	// the squelch / -L with rms()'s sums taken by the front end itself (round 5) comes before the emit mode
	const bool fuse_sq = h->opt.squelch_fused && fused::supported_sq(h->cfg);
	bool can_fuse = fused::supported(h->cfg, nblocks) || fuse_sq;
	// the raw DC block rides on the MFMA pass 0, and only that engine has the partial-tile kernels (-W n)
	const bool mfma_only = h->cfg.dc_block_raw || fused::needs_partial_tiles(h->cfg);
	if (mfma_only && fused::effective_engine(h->fws) != 1) can_fuse = false;
	if (can_fuse && plan_tail(h, nblocks).oop() == 0 && (((uintptr_t)d_out & 15) || (out_stride & 7)))
		can_fuse = false;  // the fused kernel stores 16-byte vectors straight into d_out
	const bool can_box = boxfused::supported(h->cfg) || (h->opt.squelch_fused && boxfused::supported_sq(h->cfg));
	const bool can_box_emit = boxfused::supported_emit(h->cfg);
	const bool can_deep = fused::supported_emit(h->cfg) && !(mfma_only && fused::effective_engine(h->fws) != 1);
	int r;
	{
		// this step reuses the res[] / d_cnt[] set and the state copy the step before last handed to
		// its audio tail
		const int par = (int)(h->step & 1);
		if (h->tail_pending[par]) {
			HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_tail[par], 0));
			h->tail_pending[par] = false;
		}
	}
	if (h->path == 2 && !can_fuse && !can_box && !can_box_emit && !can_deep) return -ENOTSUP;
	// state is double-buffered: kernels read st[cur], write st[cur^1].  The staged kernels each
	// update their own fields, so the record is copied first; the fused kernels copy it themselves.
	if (!(h->path != 1 && (can_fuse || can_box || can_box_emit || can_deep)))
		HIP_TRY(hipMemcpyAsync(h->st[(h->st_cur + 1) % 3], h->st[h->st_cur], S * sizeof(state_t),
		                       hipMemcpyDeviceToDevice, h->stream));
	if (h->path != 1 && can_box) {
		r = run_boxfused(h, d_iq, stream_stride, nblocks, d_out, out_stride, d_out_len);
		h->last_path = 2;
	} else if (h->path != 1 && can_box_emit) {
		r = run_boxfused_emit(h, d_iq, stream_stride, nblocks, d_out, out_stride, d_out_len);
		h->last_path = 2;
	} else if (h->path != 1 && can_deep && !(can_fuse && fuse_sq)) {
		r = run_fused_emit(h, d_iq, stream_stride, nblocks, d_out, out_stride, d_out_len);
		h->last_path = 2;
	} else if (h->path != 1 && can_fuse) {
		r = run_fused(h, d_iq, stream_stride, nblocks, d_out, out_stride, d_out_len);
		h->last_path = 2;
	} else {
		r = run_staged(h, d_iq, stream_stride, nblocks, d_out, out_stride, d_out_len);
		h->last_path = 1;
	}
	if (r < 0) return r;
	HIP_TRY(hipGetLastError());
	return 0;
}

// verify_twice: rows of two executions of one run, dword by dword up to each stream's length; lengths; state records
__global__ void k_verify_rows(const int16_t *a, const int16_t *b, size_t stride, const int32_t *la, const int32_t *lb, int S,
                              unsigned long long *cnt)
{
	const int s = blockIdx.x;
	if (s >= S) return;
	const int n = la[s];
	if (threadIdx.x == 0 && n != lb[s]) atomicAdd(cnt + 1, 1ull);
	const int16_t *ra = a + (size_t)s * stride, *rb = b + (size_t)s * stride;
	unsigned long long bad = 0, first = 0;
	for (int i = threadIdx.x; i < n; i += blockDim.x)
		if (ra[i] != rb[i]) { bad++; if (!first) first = ((unsigned long long)s << 32 | (unsigned)i) + 1; }
	if (bad) { atomicAdd(cnt, bad); atomicMin(cnt + 3, first); }
}
__global__ void k_verify_words(const uint32_t *a, const uint32_t *b, size_t n, unsigned long long *cnt)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && a[i] != b[i]) atomicAdd(cnt + 2, 1ull);
}

// The debug option verify_twice (round 6, VERDICT r5 task 1b): the run is executed twice from the same carried state -
// first into shadow rows, then into the caller's - with the device idle in between, and rows, lengths and the state
// records the two executions left are compared on the device.  A difference is a TRANSIENT fault of the device code
// (the kernels are deterministic functions of their input: tools/determinism_stress.py); a result that differs from the
// oracle while the two executions agree is a deterministic fault, or one outside the launch (the upload, the download).
static int run_device_verified(rtlfm_gpu *h, const uint8_t *d_iq, size_t stream_stride, int nblocks,
                               int16_t *d_out, size_t out_stride, int32_t *d_out_len)
{
	const size_t S = (size_t)h->nstreams;
	const size_t need = S * out_stride;
	if (h->vt_out_cap < need) {
		if (h->vt_out) hipFree(h->vt_out);
		h->vt_out = nullptr; h->vt_out_cap = 0;
		HIP_TRY(hipMalloc(&h->vt_out, need * sizeof(int16_t)));
		h->vt_out_cap = need;
	}
	if (!h->vt_len) {
		HIP_TRY(hipMalloc(&h->vt_len, S * sizeof(int32_t)));
		HIP_TRY(hipMalloc(&h->vt_len2, S * sizeof(int32_t)));
		HIP_TRY(hipMalloc(&h->vt_state, S * sizeof(state_t)));
		HIP_TRY(hipMalloc(&h->vt_cnt, 4 * sizeof(unsigned long long)));
	}
	state_t *sout = h->st[(h->st_cur + 1) % 3];
	int r = run_device_once(h, d_iq, stream_stride, nblocks, h->vt_out, out_stride, h->vt_len);
	if (r < 0) return r;
	HIP_TRY(sync_all(h));
	// (on the handle's own stream and waited for: a device-to-device hipMemcpy is only ordered on the NULL stream and returns
	// before it has run - the second execution, on a non-blocking stream, then overwrote the record while it was being copied,
	// and the comparison reported state words that had never differed: one red run of the suite in round 6)
	HIP_TRY(hipMemcpyAsync(h->vt_state, sout, S * sizeof(state_t), hipMemcpyDeviceToDevice, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	if (h->opt.verify_inject) {
		// (tests: a difference the comparison must report - the shadow's first sample with every bit turned over)
		int16_t v = 0;
		HIP_TRY(hipMemcpy(&v, h->vt_out, sizeof(v), hipMemcpyDeviceToHost));
		v = (int16_t)~v;
		HIP_TRY(hipMemcpy(h->vt_out, &v, sizeof(v), hipMemcpyHostToDevice));
	}
	int32_t *len2 = d_out_len ? d_out_len : h->vt_len2;
	r = run_device_once(h, d_iq, stream_stride, nblocks, d_out, out_stride, len2);
	if (r < 0) return r;
	HIP_TRY(sync_all(h));
	const unsigned long long init[4] = {0, 0, 0, ~0ull};
	HIP_TRY(hipMemcpy(h->vt_cnt, init, sizeof(init), hipMemcpyHostToDevice));
	k_verify_rows<<<(unsigned)S, 256, 0, h->stream>>>(h->vt_out, d_out, out_stride, h->vt_len, len2, (int)S, h->vt_cnt);
	const size_t nw = S * sizeof(state_t) / 4;
	k_verify_words<<<(unsigned)((nw + 255) / 256), 256, 0, h->stream>>>(reinterpret_cast<const uint32_t *>(h->vt_state),
	                                                                  reinterpret_cast<const uint32_t *>(sout), nw, h->vt_cnt);
	unsigned long long got[4];
	HIP_TRY(hipMemcpyAsync(got, h->vt_cnt, sizeof(got), hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	h->vt_runs++;
	if (got[0] || got[1] || got[2]) {
		h->vt_mismatches++;
		fprintf(stderr, "rtlfm_hip: verify_twice: two executions of one run differ: %llu PCM samples, %llu lengths, %llu state words; "
		        "first at stream %llu sample %llu (run %ld of this handle, input %p, rows %p / shadow %p)\n",
		        got[0], got[1], got[2], got[0] ? (got[3] - 1) >> 32 : 0ull, got[0] ? (got[3] - 1) & 0xffffffffull : 0ull,
		        h->vt_runs, (const void *)d_iq, (void *)d_out, (void *)h->vt_out);
	}
	return 0;
}

extern "C" int rtlfm_gpu_run_device(rtlfm_gpu *h, const uint8_t *d_iq, size_t stream_stride, int nblocks,
                                    int16_t *d_out, size_t out_stride, int32_t *d_out_len)
{
	if (!h || !d_iq || !d_out || nblocks < 1) return -EINVAL;
	if (nblocks > h->cap_blocks) return -E2BIG;
	if (((uintptr_t)d_iq & 15) || (stream_stride & 15) || ((uintptr_t)d_out & 3) || (out_stride & 1)) return -EINVAL;
	if (stream_stride < (size_t)nblocks * h->cfg.block_len) return -EINVAL;
	HIP_TRY(hipSetDevice(h->device));
	const int r = h->opt.verify_twice ? run_device_verified(h, d_iq, stream_stride, nblocks, d_out, out_stride, d_out_len)
	                                  : run_device_once(h, d_iq, stream_stride, nblocks, d_out, out_stride, d_out_len);
	if (r < 0) return r;
	h->st_cur = (h->st_cur + 1) % 3;
	h->step++;
	h->last_nblocks = nblocks;
	return 0;
}

extern "C" int rtlfm_gpu_levels(rtlfm_gpu *h, int stream, int32_t *rms, int cap, int *n)
{
	if (!h || !rms || !n || stream < 0 || stream >= h->nstreams) return -EINVAL;
	if (!h->cfg.squelch_level && !h->cfg.report_levels) return -ENODATA;
	*n = h->last_nblocks;
	if (h->last_nblocks > cap) return -ENOBUFS;
	HIP_TRY(hipSetDevice(h->device));
	HIP_TRY(sync_all(h));
	// the kernels index the buffers of a run [stream][nblocks]
	if (h->last_nblocks > 0)
		HIP_TRY(hipMemcpy(rms, h->d_levels + (size_t)stream * h->last_nblocks, (size_t)h->last_nblocks * sizeof(int32_t),
		                  hipMemcpyDeviceToHost));
	return 0;
}

// --------------------------------------------- callback-side: push / fetch ----
//
// What sits behind rtlsdr_read_async's callback (include/rtl-sdr.h:472).  librtlsdr owns the
// callback's buffer and resubmits it to USB the moment the callback returns
// (src/librtlsdr.c:2705-2707), so push() copies — into a pinned staging ring of two halves:
//
//   callbacks --memcpy--> h_stage[fill]      (shared lock: any number of streams at once, a slot
//                                             per (stream, buffer), per-stream atomic counters)
//   rtlfm_gpu_run(): flips `fill` (exclusive lock held only for the count check and the flip),
//                    then h_stage[f] --async H2D, copy stream--> d_in[f] --kernels--> d_result[f]
//
// so callbacks never wait for a transfer or a kernel: while run k's H2D and kernels are in
// flight the callbacks fill the other half, and run k + 1's H2D overlaps run k's kernels.  A half
// is handed back to the callbacks when its H2D has completed (event), d_in[f] is overwritten when
// the kernels that read it are done (event), results are double-buffered likewise: those of a run
// stay valid until the second run after it.
//
// len is the transfer's actual_length and may be short (src/rtl_fm.c:1326-1341 uses len
// throughout).  Full buffers take the batched path; a run that holds a short buffer goes buffer
// by buffer, each contiguous range of streams with the same length through a view of the handle
// configured for that length (the chain's carried state makes this exact, and rare).
struct Ingest {
	uint8_t *h_stage[2] = {nullptr, nullptr};   // pinned, [stream][cap_blocks][block_len]
	uint8_t *d_in[2] = {nullptr, nullptr};
	std::vector<uint32_t> h_len[2];             // bytes in each slot
	std::unique_ptr<std::atomic<int>[]> pushed[2];
	std::unique_ptr<std::atomic<int>[]> open_slot;  // [stream] 1 + half while a slot is out between acquire and commit
	int fill = 0;                               // guarded by mu (shared: read, exclusive: flip)
	std::shared_mutex mu;
	hipStream_t copy_stream = nullptr;
	hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_run[2] = {nullptr, nullptr};
	bool h2d_pending[2] = {false, false}, run_pending[2] = {false, false};
	int16_t *d_result[2] = {nullptr, nullptr};  // ostride int16 per stream
	bool result_one_block = false;              // d_result[1] points into d_result[0]'s allocation
	int32_t *d_result_len[2] = {nullptr, nullptr};
	size_t ostride = 0;
	int pending_f = -1, pending_nb = 0;         // rtlfm_gpu_run_begin() has flipped the halves, _end() has not run yet
	bool pending_ragged = false;
	int last = -1;                              // half of the last run
	int prev = -1;                              // ... and of the run before it (rtlfm_gpu_fetch_all_prev)
	int tail_par[2] = {0, 0};                   // per half: step parity of its run, and whether that run left an audio tail
	bool tail_any[2] = {false, false};          // on the tail stream (ev_tail[parity])
	int16_t *h_result = nullptr;                // pinned mirror of the last run's results (per-stream fetch)
	int32_t *h_result_len = nullptr;
	bool mirror_valid = false;
	int16_t *d_tmp = nullptr;                   // ragged runs: one buffer's results before they are appended
	int32_t *d_tmp_len = nullptr;
};

static std::mutex g_ingest_alloc_mu;

// what the handle's placed blocks really hold (a winning candidate of a placement search may be larger than the request:
// rtlfm_gpu_malloc_apart_ex): the ring's result block(s), the audio tail's work buffers, the emit mode's buffer
static size_t alloc_size(const void *p)
{
	hipDeviceptr_t base = nullptr;
	size_t size = 0;
	if (!p || hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return size;
}
static long placement_held_mb(rtlfm_gpu *h)
{
	size_t b = alloc_size(h->res[0][0]) + (h->res_one_block ? 0 : alloc_size(h->res[1][0])) + alloc_size(h->deepA);
	if (Ingest *in = __atomic_load_n(&h->ing, __ATOMIC_ACQUIRE))
		b += alloc_size(in->d_result[0]) + (in->result_one_block ? 0 : alloc_size(in->d_result[1]));
	return (long)(b >> 20);
}

static void ingest_free(Ingest *in)
{
	if (!in) return;
	if (in->copy_stream) { int dev_ = 0; (void)hipGetDevice(&dev_); rtl_pool::stream_put(dev_, 0, in->copy_stream); }
	for (int k = 0; k < 2; k++) {
		if (in->h_stage[k]) hipHostFree(in->h_stage[k]);
		for (void *p : {(void *)in->d_in[k], k == 1 && in->result_one_block ? nullptr : (void *)in->d_result[k], (void *)in->d_result_len[k]})
			if (p) hipFree(p);
		for (hipEvent_t e : {in->ev_h2d[k], in->ev_run[k]})
			if (e) hipEventDestroy(e);
	}
	if (in->h_result) hipHostFree(in->h_result);
	if (in->h_result_len) hipHostFree(in->h_result_len);
	if (in->d_tmp) hipFree(in->d_tmp);
	if (in->d_tmp_len) hipFree(in->d_tmp_len);
	delete in;
}

static int ingest_build(rtlfm_gpu *h, Ingest *in)
{
	const size_t S = (size_t)h->nstreams;
	const size_t bytes = S * h->cap_blocks * h->cfg.block_len;
	HIP_TRY(hipSetDevice(h->device));
	in->ostride = ((size_t)rtlfm_result_cap(&h->cfg) * h->cap_blocks + 16 + 63) & ~(size_t)63;
	HIP_TRY(rtl_pool::stream_get(h->device, 0, &in->copy_stream));
	for (int k = 0; k < 2; k++) {
		HIP_TRY(hipHostMalloc(&in->h_stage[k], bytes, hipHostMallocDefault));
		HIP_TRY(hipMalloc(&in->d_in[k], bytes));
		HIP_TRY(hipMalloc(&in->d_result_len[k], S * sizeof(int32_t)));
		HIP_TRY(hipEventCreateWithFlags(&in->ev_h2d[k], hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&in->ev_run[k], hipEventDisableTiming));
		in->h_len[k].assign(S * h->cap_blocks, 0);
		in->pushed[k].reset(new std::atomic<int>[S]);
		for (size_t s = 0; s < S; s++) in->pushed[k][s].store(0);
	}
	{
		// the results away from the input they are demodulated from (rtlfm_gpu_malloc_apart_ex; plain memory when the
		// ring is too small for it to matter).  Both halves' results as ONE block placed against the first half's input;
		// the second half's input was allocated right behind the first and is almost always of the same class - one probe
		// says whether it is, and only if not the second half gets a block and a search of its own.
		const size_t one = (S * in->ostride * sizeof(int16_t) + 255) & ~(size_t)255;
		const size_t budget = (size_t)h->place.budget_gb << 30;
		void *p = nullptr;
		int apart = 0; double ms = 0; size_t walked = 0;
		int r = rtlfm_gpu_malloc_apart_ex(h->device, 2 * one, in->d_in[0], bytes, budget, &p, &apart, &ms, &walked);
		if (r < 0) return r;
		h->place.search_ms += ms;
		if (walked > h->place.walked_peak) h->place.walked_peak = walked;
		h->place.ring_tries = 1;
		if ((!apart && walked > 0) || h->place.force_retry) {  // (force_retry: option "ring_force_retry", tests)
			// A search was made and every candidate within its bound shared the input's class (round 5 met boxes where one class
			// runs on for more than 16 GB).  Here the handle owns the OTHER side as well: the ring's device inputs move - new ones
			// are allocated while the old ones are still held, so that they come from somewhere else - and the search runs once
			// more against them.  One retry; whatever it finds is what the ring gets.
			void *n0 = nullptr, *n1 = nullptr;
			if (hipMalloc(&n0, bytes) == hipSuccess && hipMalloc(&n1, bytes) == hipSuccess) {
				(void)hipFree(p);
				p = nullptr;
				int apart2 = 0;
				r = rtlfm_gpu_malloc_apart_ex(h->device, 2 * one, n0, bytes, budget, &p, &apart2, &ms, &walked);
				if (r < 0) { (void)hipFree(n0); (void)hipFree(n1); return r; }
				h->place.search_ms += ms;
				if (walked > h->place.walked_peak) h->place.walked_peak = walked;
				h->place.ring_tries = 2;
				(void)hipFree(in->d_in[0]); (void)hipFree(in->d_in[1]);
				in->d_in[0] = (uint8_t *)n0; in->d_in[1] = (uint8_t *)n1;
				apart = apart2;
			} else {
				if (n0) (void)hipFree(n0);
				(void)hipGetLastError();
			}
		}
		in->d_result[0] = (int16_t *)p;
		in->d_result[1] = (int16_t *)((char *)p + one);
		in->result_one_block = true;
		h->place.ring_apart = apart;
		if (apart) {
			double rd = 0, rw = 0;
			const auto t0 = std::chrono::steady_clock::now();
			const int both = rtlfm_gpu_placement_probe(h->device, in->d_in[1], bytes, in->d_result[1], one, &rd, &rw);
			h->place.search_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
			if (both < 0) return both;
			if (!both) {
				void *p2 = nullptr; int apart2 = 0;
				r = rtlfm_gpu_malloc_apart_ex(h->device, one, in->d_in[1], bytes, budget, &p2, &apart2, &ms, &walked);
				if (r < 0) return r;
				h->place.search_ms += ms;
				if (walked > h->place.walked_peak) h->place.walked_peak = walked;
				in->d_result[1] = (int16_t *)p2;
				in->result_one_block = false;
				h->place.ring_apart = apart2;
			}
		}
	}
	in->open_slot.reset(new std::atomic<int>[S]);
	for (size_t s = 0; s < S; s++) in->open_slot[s].store(0);
	return 0;
}

// the ring is built by whoever comes first (callbacks of many streams may arrive at once) and
// published only when it is complete
static int ingest_ensure(rtlfm_gpu *h)
{
	if (__atomic_load_n(&h->ing, __ATOMIC_ACQUIRE)) return 0;
	std::lock_guard<std::mutex> g(g_ingest_alloc_mu);
	if (h->ing) return 0;
	Ingest *in = new Ingest();
	int r = ingest_build(h, in);
	if (r < 0) {
		ingest_free(in);
		return r;
	}
	__atomic_store_n(&h->ing, in, __ATOMIC_RELEASE);
	return 0;
}

static void ingest_destroy(rtlfm_gpu *h)
{
	ingest_free(h->ing);
	h->ing = nullptr;
}

static void ingest_reset(rtlfm_gpu *h)
{
	Ingest *in = h->ing;
	if (!in) return;
	std::unique_lock<std::shared_mutex> g(in->mu);
	for (int k = 0; k < 2; k++)
		for (int s = 0; s < h->nstreams; s++) in->pushed[k][s].store(0);
	for (int s = 0; s < h->nstreams; s++) in->open_slot[s].store(0);
	in->last = -1;
	in->prev = -1;
	in->pending_f = -1;
	in->mirror_valid = false;
}

extern "C" int rtlfm_gpu_push(rtlfm_gpu *h, int stream, const uint8_t *iq, uint32_t len)
{
	if (!h || !iq || stream < 0 || stream >= h->nstreams) return -EINVAL;
	// actual_length of a bulk transfer: whole 512-byte USB packets, at most the buffer
	if (len == 0 || len > h->cfg.block_len || len % 512) return -EINVAL;
	if (len != h->cfg.block_len) {
		// a short buffer is demodulated as a buffer of that length: refuse here, before anything is
		// queued, what the chain cannot take (e.g. a length the fifth_order passes do not divide)
		rtlfm_cfg c = h->cfg;
		c.block_len = len;
		c.max_blocks = 1;
		const int v = validate_cfg(&c);
		if (v < 0) return v;
	}
	int r = ingest_ensure(h);
	if (r < 0) return r;
	Ingest *in = h->ing;
	// shared: pushes of different streams run side by side; rtlfm_gpu_run() takes the lock
	// exclusively only to flip the halves, so it sees every buffer whole or not at all
	std::shared_lock<std::shared_mutex> g(in->mu);
	// a slot of this stream is out between acquire and commit: a push would land in that very slot
	if (in->open_slot[stream].load(std::memory_order_acquire)) return -EBUSY;
	const int f = in->fill;
	const int slot = in->pushed[f][stream].fetch_add(1, std::memory_order_acq_rel);
	if (slot >= h->cap_blocks) {
		in->pushed[f][stream].fetch_sub(1, std::memory_order_acq_rel);
		return -ENOSPC;
	}
	const size_t at = (size_t)stream * h->cap_blocks + slot;
	memcpy(in->h_stage[f] + at * h->cfg.block_len, iq, len);
	in->h_len[f][at] = len;
	return 0;
}

// Zero-copy ingest: the producer fills the pinned ring slot itself.  The reference has this at the
// USB side (use_zerocopy, src/librtlsdr.c:2744-2810: the kernel's transfer buffers are mapped into
// the process and handed to the callback as they are); here the device layer - a file reader, an
// rtl_tcp receiver, anything that can write to a pointer - asks for the next slot of its stream,
// writes the samples there, and commits the length.  No memcpy between the producer and the H2D copy.
extern "C" int rtlfm_gpu_acquire(rtlfm_gpu *h, int stream, uint8_t **buf, uint32_t *cap)
{
	if (!h || !buf || stream < 0 || stream >= h->nstreams) return -EINVAL;
	int r = ingest_ensure(h);
	if (r < 0) return r;
	Ingest *in = h->ing;
	std::shared_lock<std::shared_mutex> g(in->mu);
	const int f = in->fill;
	if (in->open_slot[stream].load(std::memory_order_acquire)) return -EBUSY;  // one open slot per stream
	const int slot = in->pushed[f][stream].load(std::memory_order_acquire);
	if (slot >= h->cap_blocks) return -ENOSPC;
	in->open_slot[stream].store(1 + f, std::memory_order_release);
	*buf = in->h_stage[f] + ((size_t)stream * h->cap_blocks + slot) * h->cfg.block_len;
	if (cap) *cap = h->cfg.block_len;
	return 0;
}

extern "C" int rtlfm_gpu_commit(rtlfm_gpu *h, int stream, uint32_t len)
{
	if (!h || stream < 0 || stream >= h->nstreams) return -EINVAL;
	Ingest *in = h->ing;
	if (!in) return -EINVAL;
	std::shared_lock<std::shared_mutex> g(in->mu);
	const int o = in->open_slot[stream].load(std::memory_order_acquire);
	if (!o) return -EINVAL;
	const int f = o - 1;  // rtlfm_gpu_run() does not flip the halves while a slot is open: f is still the filling half
	if (len == 0) {       // nothing arrived: give the slot back
		in->open_slot[stream].store(0, std::memory_order_release);
		return 0;
	}
	if (len > h->cfg.block_len || len % 512) return -EINVAL;
	if (len != h->cfg.block_len) {
		rtlfm_cfg c = h->cfg;
		c.block_len = len;
		c.max_blocks = 1;
		const int v = validate_cfg(&c);
		if (v < 0) return v;
	}
	const int slot = in->pushed[f][stream].load(std::memory_order_acquire);
	in->h_len[f][(size_t)stream * h->cap_blocks + slot] = len;
	in->pushed[f][stream].store(slot + 1, std::memory_order_release);
	in->open_slot[stream].store(0, std::memory_order_release);
	return 0;
}

// results of one buffer (tmp, tmp_len per stream) appended behind what the run has so far
__global__ void k_append_results(int16_t *dst, size_t dstride, int32_t *dst_len, const int16_t *src, size_t sstride,
                                 const int32_t *src_len, int s0, int ns, int cap)
{
	const int s = s0 + (int)blockIdx.x;
	if (s >= s0 + ns) return;
	const int n = src_len[s], at = dst_len[s];
	for (int k = threadIdx.x; k < n && at + k < cap; k += blockDim.x) dst[s * dstride + at + k] = src[s * sstride + k];
	__syncthreads();
	if (threadIdx.x == 0) dst_len[s] = at + n;
}

// A view of the handle on streams [s0, s0 + ns) with another buffer length: same device buffers,
// per-stream pointers shifted, everything on the handle's stream (no tail overlap, no timing).
static rtlfm_gpu make_view(rtlfm_gpu *h, int s0, int ns, uint32_t block_len)
{
	rtlfm_gpu v = *h;
	v.nstreams = ns;
	v.cfg.block_len = block_len;
	v.ing = nullptr;
	v.tail_overlap = false;
	v.timing = false;
	v.no_deemph_scan = true;
	v.opt.verify_twice = 0;  // views must not allocate (the shadow rows): ragged runs are not verified
	v.ev_pending.clear(); v.ev_free.clear();
	const size_t cb = (size_t)h->cap_blocks;
	for (int k = 0; k < 3; k++) v.st[k] = h->st[k] + s0;
	for (int k = 0; k < 2; k++) {
		v.d_cnt[k] = h->d_cnt[k] + s0;
		for (int j = 0; j < 2; j++) v.res[k][j] = h->res[k][j] ? h->res[k][j] + (size_t)s0 * h->tstride : nullptr;
	}
	v.d_cnt2 = h->d_cnt2 + s0;
	v.d_mute = h->d_mute + s0 * cb;
	v.d_levels = h->d_levels + s0 * cb;
	v.d_sums = h->d_sums + s0 * cb * 2;
	v.d_sq_sums = h->d_sq_sums + s0 * cb * 2;
	v.d_adc_sums = h->d_adc_sums + s0 * cb;
	v.d_rdc_avg = h->d_rdc_avg + s0 * cb;
	v.d_adc_avg = h->d_adc_avg + s0 * cb;
	if (h->bufA) { v.bufA = h->bufA + (size_t)s0 * h->xstride; v.bufB = h->bufB + (size_t)s0 * h->xstride; }
	if (h->deepA) v.deepA = h->deepA + (size_t)s0 * h->deep_stride;
	if (h->deepB) v.deepB = h->deepB + (size_t)s0 * h->deep_stride;
	return v;
}

// a run that holds short buffers: buffer by buffer, ranges of streams with equal length
static int run_ragged(rtlfm_gpu *h, Ingest *in, int f, int nb)
{
	const int S = h->nstreams;
	const size_t stride = (size_t)h->cap_blocks * h->cfg.block_len;
	int r;
	// views must not allocate: everything any path may need exists before the first one is made
	if ((r = ensure_work_buffers(h)) < 0 || (r = ensure_res_buffers(h)) < 0 || (r = ensure_deep_buffers(h)) < 0) return r;
	if (fused::ensure_dummy_tile(h->fws)) return -ENOMEM;
	if (!in->d_tmp) {
		HIP_TRY(hipMalloc(&in->d_tmp, (size_t)S * in->ostride * sizeof(int16_t)));
		HIP_TRY(hipMalloc(&in->d_tmp_len, (size_t)S * sizeof(int32_t)));
	}
	// The views run their audio tail on the handle's main stream: the tails of the two previous
	// (batched) steps may still be running on the tail stream, writing the state copy and the res[]
	// sets the views are about to read.  Order the main stream behind both, on the handle itself.
	for (int p = 0; p < 2; p++)
		if (h->tail_pending[p]) {
			HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_tail[p], 0));
			h->tail_pending[p] = false;
		}
	HIP_TRY(hipMemsetAsync(in->d_result_len[f], 0, (size_t)S * sizeof(int32_t), h->stream));
	for (int j = 0; j < nb; j++) {
		// every range of this buffer reads st[cur] and writes the next copy; the rotation advances once
		const int cur = h->st_cur;
		const unsigned step = h->step;
		for (int s0 = 0; s0 < S;) {
			const uint32_t len = in->h_len[f][(size_t)s0 * h->cap_blocks + j];
			int s1 = s0 + 1;
			while (s1 < S && in->h_len[f][(size_t)s1 * h->cap_blocks + j] == len) s1++;
			rtlfm_cfg c = h->cfg;
			c.block_len = len;
			c.max_blocks = 1;
			if ((r = validate_cfg(&c)) < 0) return r;
			rtlfm_gpu v = make_view(h, s0, s1 - s0, len);
			v.st_cur = cur;
			v.step = step;
			r = rtlfm_gpu_run_device(&v, in->d_in[f] + (size_t)s0 * stride + (size_t)j * h->cfg.block_len, stride, 1,
			                         in->d_tmp + (size_t)s0 * in->ostride, in->ostride, in->d_tmp_len + s0);
			h->fws = v.fws;  // lazily created tap tables / probe buffers belong to the handle
			h->last_path = v.last_path;
			if (r < 0) return r;
			s0 = s1;
		}
		h->st_cur = (cur + 1) % 3;
		h->step = step + 1;
		k_append_results<<<S, 256, 0, h->stream>>>(in->d_result[f], in->ostride, in->d_result_len[f], in->d_tmp, in->ostride,
		                                          in->d_tmp_len, 0, S, (int)in->ostride);
	}
	HIP_TRY(hipGetLastError());
	return 0;
}

// rtlfm_gpu_run() in two steps, for a caller that gates its producers around the flip only (host/rtl_fm_hip.cpp): _begin
// takes what the ring's filling half holds - every stream the same number of buffers, no slot open - and flips the
// halves, which is all that must exclude the producers; _end queues the H2D copy and the kernels.  Between the two the
// producers already fill the other half.  *taken (may be NULL) = buffers per stream the run took.
extern "C" int rtlfm_gpu_run_begin(rtlfm_gpu *h, int *taken)
{
	if (!h) return -EINVAL;
	int r = ingest_ensure(h);
	if (r < 0) return r;
	Ingest *in = h->ing;
	if (in->pending_f >= 0) return -EBUSY;  // a begun run has not been ended
	HIP_TRY(hipSetDevice(h->device));
	const int S = h->nstreams;
	const uint32_t L = h->cfg.block_len;
	int f, nb;
	bool ragged = false;
	{
		// the half the callbacks will fill next must have left for the GPU (its H2D done); waited
		// for before the lock, so that callbacks are never held up by a transfer
		const int other_guess = in->fill ^ 1;
		if (in->h2d_pending[other_guess]) {
			HIP_TRY(hipEventSynchronize(in->ev_h2d[other_guess]));
			in->h2d_pending[other_guess] = false;
		}
		std::unique_lock<std::shared_mutex> g(in->mu);
		f = in->fill;
		nb = in->pushed[f][0].load();
		for (int s = 1; s < S; s++)
			if (in->pushed[f][s].load() != nb) return -EAGAIN;
		if (nb == 0) return -EAGAIN;
		for (int s = 0; s < S; s++)
			if (in->open_slot[s].load(std::memory_order_acquire)) return -EAGAIN;  // a producer is still writing its slot
		for (int s = 0; s < S && !ragged; s++)
			for (int j = 0; j < nb; j++)
				if (in->h_len[f][(size_t)s * h->cap_blocks + j] != L) { ragged = true; break; }
		for (int s = 0; s < S; s++) in->pushed[f ^ 1][s].store(0);
		in->fill = f ^ 1;
	}
	in->pending_f = f; in->pending_nb = nb; in->pending_ragged = ragged;
	if (taken) *taken = nb;
	return 0;
}

extern "C" int rtlfm_gpu_run_end(rtlfm_gpu *h)
{
	if (!h || !h->ing) return -EINVAL;
	Ingest *in = h->ing;
	if (in->pending_f < 0) return -EINVAL;  // nothing begun
	HIP_TRY(hipSetDevice(h->device));
	const int S = h->nstreams;
	const uint32_t L = h->cfg.block_len;
	const size_t stride = (size_t)h->cap_blocks * L;
	const int f = in->pending_f, nb = in->pending_nb;
	const bool ragged = in->pending_ragged;
	in->pending_f = -1;  // whatever happens below, the buffers of this run are gone from the ring
	int r;
	// d_in[f] is free once the kernels of the run that read it are done
	if (in->run_pending[f]) {
		HIP_TRY(hipStreamWaitEvent(in->copy_stream, in->ev_run[f], 0));
		in->run_pending[f] = false;
	}
	if (nb == h->cap_blocks)
		HIP_TRY(hipMemcpyAsync(in->d_in[f], in->h_stage[f], stride * S, hipMemcpyHostToDevice, in->copy_stream));
	else
		HIP_TRY(hipMemcpy2DAsync(in->d_in[f], stride, in->h_stage[f], stride, (size_t)nb * L, S, hipMemcpyHostToDevice,
		                         in->copy_stream));
	HIP_TRY(hipEventRecord(in->ev_h2d[f], in->copy_stream));
	in->h2d_pending[f] = true;
	HIP_TRY(hipStreamWaitEvent(h->stream, in->ev_h2d[f], 0));
	if (ragged) r = run_ragged(h, in, f, nb);
	else r = rtlfm_gpu_run_device(h, in->d_in[f], stride, nb, in->d_result[f], in->ostride, in->d_result_len[f]);
	if (r < 0) return r;
	HIP_TRY(hipEventRecord(in->ev_run[f], h->stream));
	in->run_pending[f] = true;
	in->tail_par[f] = (int)((h->step - 1) & 1);
	in->tail_any[f] = !ragged && h->tail_pending[in->tail_par[f]];
	in->prev = in->last;
	in->last = f;
	in->mirror_valid = false;
	return 0;
}

extern "C" int rtlfm_gpu_run(rtlfm_gpu *h)
{
	int r = rtlfm_gpu_run_begin(h, nullptr);
	if (r < 0) return r;
	return rtlfm_gpu_run_end(h);
}

// Results of the run BEFORE the last one (they stay valid until the second run after theirs).  Waits for
// that run only - its front end's event and, if it left one, its audio tail's - not for the run that has
// been started since: a server that calls run(k + 1) as soon as the buffers of k + 1 are in and only then
// collects run k keeps the H2D copies of consecutive runs back to back on the link (host/ingest_bench.cpp
// --depth 2), where run / fetch / run leaves it idle for a kernel and a D2H copy per run.
extern "C" int rtlfm_gpu_fetch_all_prev(rtlfm_gpu *h, int16_t *out, size_t out_stride, int32_t *lens)
{
	if (!h || !out || !lens) return -EINVAL;
	Ingest *in = h->ing;
	if (!in || in->prev < 0 || in->prev == in->last) return -EAGAIN;
	HIP_TRY(hipSetDevice(h->device));
	const int f = in->prev;
	HIP_TRY(hipEventSynchronize(in->ev_run[f]));
	if (in->tail_any[f]) HIP_TRY(hipEventSynchronize(h->ev_tail[in->tail_par[f]]));
	const int S = h->nstreams;
	HIP_TRY(hipMemcpy(lens, in->d_result_len[f], (size_t)S * sizeof(int32_t), hipMemcpyDeviceToHost));
	int mx = 0;
	for (int s = 0; s < S; s++) mx = lens[s] > mx ? lens[s] : mx;
	if ((size_t)mx > out_stride) return -ENOBUFS;
	if (mx > 0)
		HIP_TRY(hipMemcpy2D(out, out_stride * sizeof(int16_t), in->d_result[f], in->ostride * sizeof(int16_t),
		                    (size_t)mx * sizeof(int16_t), S, hipMemcpyDeviceToHost));
	return 0;
}

extern "C" int rtlfm_gpu_fetch_all(rtlfm_gpu *h, int16_t *out, size_t out_stride, int32_t *lens)
{
	if (!h || !out || !lens) return -EINVAL;
	Ingest *in = h->ing;
	if (!in || in->last < 0) return -EAGAIN;
	HIP_TRY(hipSetDevice(h->device));
	HIP_TRY(sync_all(h));
	const int f = in->last;
	const int S = h->nstreams;
	HIP_TRY(hipMemcpy(lens, in->d_result_len[f], (size_t)S * sizeof(int32_t), hipMemcpyDeviceToHost));
	int mx = 0;
	for (int s = 0; s < S; s++) mx = lens[s] > mx ? lens[s] : mx;
	if ((size_t)mx > out_stride) return -ENOBUFS;
	if (mx > 0)
		HIP_TRY(hipMemcpy2D(out, out_stride * sizeof(int16_t), in->d_result[f], in->ostride * sizeof(int16_t),
		                    (size_t)mx * sizeof(int16_t), S, hipMemcpyDeviceToHost));
	return 0;
}

extern "C" int rtlfm_gpu_fetch(rtlfm_gpu *h, int stream, int16_t *out, int cap, int *n)
{
	if (!h || !out || !n || stream < 0 || stream >= h->nstreams) return -EINVAL;
	Ingest *in = h->ing;
	if (!in || in->last < 0) return -EAGAIN;
	HIP_TRY(hipSetDevice(h->device));
	if (!in->mirror_valid) {
		// one transfer for all streams, then every per-stream fetch is a host copy
		if (!in->h_result) {
			HIP_TRY(hipHostMalloc(&in->h_result, (size_t)h->nstreams * in->ostride * sizeof(int16_t), hipHostMallocDefault));
			HIP_TRY(hipHostMalloc(&in->h_result_len, (size_t)h->nstreams * sizeof(int32_t), hipHostMallocDefault));
		}
		int r = rtlfm_gpu_fetch_all(h, in->h_result, in->ostride, in->h_result_len);
		if (r < 0) return r;
		in->mirror_valid = true;
	}
	const int32_t len = in->h_result_len[stream];
	*n = len;
	if (len > cap) return -ENOBUFS;
	if (len > 0) memcpy(out, in->h_result + (size_t)stream * in->ostride, (size_t)len * sizeof(int16_t));
	return 0;
}

__global__ void k_selftest_atan2(const int32_t *yx, int n, int32_t *a, int32_t *b)
{
	RTLFM_GRID_STRIDE(i, n) {
		int y = yx[2 * i], x = yx[2 * i + 1];
		if (a) a[i] = atan2_q14(y, x, AtanNodesConst());
		if (b) b[i] = atan2_q14_libm(y, x);
	}
}
__global__ void k_selftest_fast_atan2(const int32_t *yx, int n, int32_t *a)
{
	RTLFM_GRID_STRIDE(i, n) a[i] = fast_atan2_q14(yx[2 * i], yx[2 * i + 1]);
}

__global__ void k_selftest_const_div(const int32_t *nd, int n, int32_t *q)
{
	RTLFM_GRID_STRIDE(i, n) {
		const ConstDiv cd(nd[2 * i + 1]);
		q[i] = cd(nd[2 * i]);
	}
}

extern "C" int rtlfm_gpu_selftest_const_div(int device, const int32_t *nd, int n, int32_t *q)
{
	if (!nd || !q || n < 1) return -EINVAL;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -ENODEV;
	HIP_TRY(hipSetDevice(device));
	int32_t *d_in = nullptr, *d_q = nullptr;
	HIP_TRY(hipMalloc(&d_in, (size_t)n * 8));
	HIP_TRY(hipMalloc(&d_q, (size_t)n * 4));
	HIP_TRY(hipMemcpy(d_in, nd, (size_t)n * 8, hipMemcpyHostToDevice));
	k_selftest_const_div<<<grid_for((size_t)n), 256>>>(d_in, n, d_q);
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipMemcpy(q, d_q, (size_t)n * 4, hipMemcpyDeviceToHost));
	hipFree(d_in); hipFree(d_q);
	return 0;
}

extern "C" int rtlfm_gpu_selftest_fast_atan2(int device, const int32_t *yx, int n, int32_t *q14)
{
	if (!yx || !q14 || n < 1) return -EINVAL;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -ENODEV;
	HIP_TRY(hipSetDevice(device));
	int32_t *d_in = nullptr, *d_a = nullptr;
	HIP_TRY(hipMalloc(&d_in, (size_t)n * 8));
	HIP_TRY(hipMalloc(&d_a, (size_t)n * 4));
	HIP_TRY(hipMemcpy(d_in, yx, (size_t)n * 8, hipMemcpyHostToDevice));
	k_selftest_fast_atan2<<<grid_for((size_t)n), 256>>>(d_in, n, d_a);
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipMemcpy(q14, d_a, (size_t)n * 4, hipMemcpyDeviceToHost));
	hipFree(d_in); hipFree(d_a);
	return 0;
}

extern "C" int rtlfm_gpu_selftest_atan2(int device, const int32_t *yx, int n, int32_t *q14, int32_t *q14_libm)
{
	if (!yx || n < 1) return -EINVAL;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -ENODEV;
	HIP_TRY(hipSetDevice(device));
	int32_t *d_in = nullptr, *d_a = nullptr, *d_b = nullptr;
	HIP_TRY(hipMalloc(&d_in, (size_t)n * 8));
	HIP_TRY(hipMalloc(&d_a, (size_t)n * 4));
	HIP_TRY(hipMalloc(&d_b, (size_t)n * 4));
	HIP_TRY(hipMemcpy(d_in, yx, (size_t)n * 8, hipMemcpyHostToDevice));
	k_selftest_atan2<<<grid_for((size_t)n), 256>>>(d_in, n, q14 ? d_a : nullptr, q14_libm ? d_b : nullptr);
	HIP_TRY(hipDeviceSynchronize());
	if (q14) HIP_TRY(hipMemcpy(q14, d_a, (size_t)n * 4, hipMemcpyDeviceToHost));
	if (q14_libm) HIP_TRY(hipMemcpy(q14_libm, d_b, (size_t)n * 4, hipMemcpyDeviceToHost));
	hipFree(d_in); hipFree(d_a); hipFree(d_b);
	return 0;
}

// rotate_90 (u8): per 8 bytes [a0 b0 a1 b1 a2 b2 a3 b3] -> [a0 b0 ~b1 a1 ~a2 ~b2 b3 ~a3]
// (~x = 255 - x).  Pure byte shuffling: one v_perm and one v_xor per dword, 16 bytes per lane.
__global__ void __launch_bounds__(256) k_rotate_90_u8(uint4 *buf, size_t n16)
{
	RTLFM_GRID_STRIDE(i, n16) {
		uint4 v = buf[i];
		v.x = __builtin_amdgcn_perm(0, v.x, 0x02030100u) ^ 0x00ff0000u;
		v.y = __builtin_amdgcn_perm(0, v.y, 0x02030100u) ^ 0xff00ffffu;
		v.z = __builtin_amdgcn_perm(0, v.z, 0x02030100u) ^ 0x00ff0000u;
		v.w = __builtin_amdgcn_perm(0, v.w, 0x02030100u) ^ 0xff00ffffu;
		buf[i] = v;
	}
}

__global__ void k_rotate_90_u8_tail(uint2 *p)  // an odd group of eight bytes at the end
{
	uint2 v = *p;
	v.x = __builtin_amdgcn_perm(0, v.x, 0x02030100u) ^ 0x00ff0000u;
	v.y = __builtin_amdgcn_perm(0, v.y, 0x02030100u) ^ 0xff00ffffu;
	*p = v;
}

extern "C" int rtlfm_gpu_rotate_90_u8(int device, void *d_buf, size_t len, void *hip_stream)
{
	if (!d_buf || (len & 7) || ((uintptr_t)d_buf & 15)) return -EINVAL;
	if (len == 0) return 0;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return -ENODEV;
	HIP_TRY(hipSetDevice(device));
	hipStream_t q = (hipStream_t)hip_stream;
	const size_t n16 = len / 16;
	if (n16) k_rotate_90_u8<<<grid_for(n16), 256, 0, q>>>((uint4 *)d_buf, n16);
	if (len & 8) k_rotate_90_u8_tail<<<1, 1, 0, q>>>((uint2 *)((uint8_t *)d_buf + len - 8));
	HIP_TRY(hipGetLastError());
	return 0;
}

extern "C" int rtlfm_gpu_clock_stamps(rtlfm_gpu *h, uint64_t *out, int cap_waves, int *waves)
{
	if (!h || !waves) return -EINVAL;
	HIP_TRY(hipSetDevice(h->device));
	HIP_TRY(sync_all(h));
	if (!h->fws.stamps || h->fws.stamp_last <= 0) return -ENODATA;
	*waves = h->fws.stamp_last;
	if (!out) return 0;
	if (cap_waves < h->fws.stamp_last) return -ENOBUFS;
	// fused_debug bit 64: the launch BEFORE the last one (the two alternate between the halves of the buffer), so that
	// a tool can see the gap between two launches that followed each other (tools/launch_gap.py)
	HIP_TRY(hipMemcpy(out, h->fws.stamp_half((h->fws.debug & 64) ? 1 : 0), (size_t)h->fws.stamp_last * 32, hipMemcpyDeviceToHost));
	return 0;
}

// How a front-end launch would be cut into waves (fused::plan_segments), without a GPU: for the planner's unit test.
// starts[0 .. *segs] = tile boundaries of a stream's segments (cap entries at least *segs + 1, else -ENOBUFS).
extern "C" int rtlfm_plan_segments(int nstreams, int total_tiles, int fifth_order, int tail_follows, int target_waves, int min_tiles,
                                   int tiles_per_seg, int gss_x10, int *segs, int *starts, int cap)
{
	if (nstreams < 1 || total_tiles < 1 || !segs || !starts) return -EINVAL;
	fused::Workspace ws;
	ws.tail_follows = tail_follows != 0;
	if (target_waves > 0) { ws.target_waves = target_waves; ws.target_waves_tail = target_waves; ws.target_waves_tail_fifth = 0; ws.plan_by_caller = true; }
	if (min_tiles > 0) { ws.min_tiles = min_tiles; ws.plan_by_caller = true; }
	ws.tiles_per_seg = tiles_per_seg;
	ws.gss_x10 = gss_x10;
	const fused::SegPlan sp = fused::plan_segments(ws, nstreams, total_tiles, fifth_order != 0);
	*segs = sp.segs;
	if (cap < sp.segs + 1) return -ENOBUFS;
	for (int j = 0; j <= sp.segs; j++) {
		int at = sp.nlist > 0 ? sp.start[j] : j * sp.tiles_per_seg;
		starts[j] = at < total_tiles ? at : total_tiles;
	}
	return 0;
}

extern "C" const char *rtlfm_gpu_strerror(int err)
{
	switch (err) {
	case 0: return "ok";
	case -EINVAL: return "invalid argument or configuration";
	case -ENODEV: return "no usable HIP device (there is no CPU fallback)";
	case -ENOMEM: return "out of device memory";
	case -ENOSPC: return "max_blocks already queued for this stream";
	case -EAGAIN: return "streams have unequal / zero queued blocks, or a producer still holds an acquired slot";
	case -EBUSY: return "the stream's previous slot is still open (rtlfm_gpu_acquire without rtlfm_gpu_commit)";
	case -ENOTSUP: return "configuration not supported on this path";
	case -EDOM: return "outside the reference's own domain: a boxcar longer than the buffer (fm_demod reads lowpassed[-2]), low_pass_simple on a "
	                   "count its step does not divide (src/rtl_fm.c:740), or rate_out2 > rate_out with low_pass_real (division by zero, :769)";
	case -E2BIG: return "more blocks than cfg.max_blocks";
	case -ENOBUFS: return "output buffer too small";
	case -EIO: return "HIP runtime error";
	case -ENODATA: return "no clock stamps recorded (rtlfm_gpu_clock_probe, fused fifth_order path only)";
	default: return strerror(-err);
	}
}

extern "C" int rtlfm_gpu_version(void) { return (0 << 16) | (1 << 8) | 0; }
