// rtlpower_hip.hip — the C ABI of include/rtlpower_hip.h on HIP / gfx950.
// Host side of rtl_power's scanner() (reference src/rtl_power.c:642-720).
// No CPU fallback.
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtlpower_hip.h"
#include "debug_poison.h"
#include "stream_pool.h"
#include "power_kernels.h"

using namespace rtlpower;

#define HIP_TRY(expr)                                                                         \
	do {                                                                                      \
		hipError_t e_ = (expr);                                                               \
		if (e_ != hipSuccess) {                                                               \
			fprintf(stderr, "rtlpower_hip: %s -> %s (%s:%d)\n", #expr, hipGetErrorString(e_), \
			        __FILE__, __LINE__);                                                      \
			return e_ == hipErrorOutOfMemory ? -ENOMEM : -EIO;                                \
		}                                                                                     \
	} while (0)

struct rtlpower_gpu {
	rtlpower_cfg cfg;
	int nstreams = 0, device = 0;
	hipStream_t own_stream = nullptr, stream = nullptr;
	hipEvent_t ev_wait = nullptr, ev_release = nullptr;
	int N = 1, len_dec = 0, chunks = 0, dec_elems = 0;
	bool decimates = false;
	int32_t *d_window = nullptr;
	uint16_t *d_window16 = nullptr;  // the low halves (k_power_scan_big multiplies in 16 bits)
	uint32_t *d_tw = nullptr;
	long long *d_avg = nullptr;
	int32_t *d_samples = nullptr;
	int16_t *d_decA = nullptr, *d_decB = nullptr;
	size_t dec_cap_reads = 0;
	uint8_t *d_one = nullptr;  // landing zone of rtlpower_gpu_scan()
	bool timing = false;
	int groups = 0;  // option "groups": workgroups per stream of the FFT kernel (0 = automatic)
	bool staged = false;                   // bin_e 15 .. 21 or more points per read than a workgroup's LDS holds: transform in HBM
	uint32_t *d_work = nullptr; int2 *d_ave = nullptr;
	// one undecimated frame per read, bin_e > 14: the window comb by comb, the batch's bytes likewise, the averages' partial sums
	uint16_t *d_window16T = nullptr; uint8_t *d_tbuf = nullptr; int2 *d_part = nullptr; bool attr_comb = false;
	int scan_frames = 1;  // option "scan_frames": 0 = k_power_scan also where k_power_scan_frames applies (A/B, tests)
	bool attr_frames = false;
	// decimated scans (a range below 1 MHz, src/rtl_power.c:466-480) in two launches: k_power_downsample_iq (-F) or
	// k_power_boxcar into packed int16 pairs, k_power_scan_frames<13, true> on them
	int dec_fast = 1;     // option "dec_fast": 0 = the general kernels (one launch per pass, k_power_scan) also where these apply
	uint32_t *d_dec32 = nullptr; size_t dec32_cap = 0;  // dwords
	bool attr_frames16 = false;
	int last_kernel = 0;  // option "last_kernel" (read-only): which transform kernel the last scan took (RTLPOWER_KERNEL_*)
	int staged_fast = 1;  // option "staged_fast": 0 = the general kernels also where the fast ones apply (A/B, tests)
	// Fine bins (one frame per read beyond 16384 points), round 5: the batches of a scan as a two-stream pipeline.  The
	// transform in LDS (k_power_scan_big<14, true>: issue-bound, 71 registers, one workgroup per CU) runs on the handle's
	// stream; what only moves bytes - the comb gather of the NEXT batch, the passes over HBM and the accumulation of the
	// one BEFORE - runs beside it on `aux` (their waves fit next to the transform's on every CU).  Two sets of work buffers.
	int staged_pipe = 0;  // option "staged_pipe": 0 (default) = one batch after the other on one stream, 1 = the pipeline from 2^18 bins on, 2 = wherever it applies
	int staged_batch = 0; // option "staged_batch": reads per batch, 0 = by the work buffer's size (tests: several batches of a small scan)
	struct Pipe {
		hipStream_t aux = nullptr;
		int aux_prio = 0;  // the HIP priority aux was taken from the stream pool with
		hipEvent_t start = nullptr, prep[2] = {nullptr, nullptr}, scanned[2] = {nullptr, nullptr}, done = nullptr;
	} pipe;
	bool work_two = false;  // d_work / d_ave / d_tbuf / d_part hold two batches
	size_t work_reads = 0;                 // reads the work buffer holds per stream
	bool attr_lds = false;
	bool want_stamps = false;              // rtlpower_gpu_clock_probe
	unsigned long long *d_stamps = nullptr;
	int stamp_cap = 0, stamp_last = 0;     // workgroups the buffer holds / of the last stamped launch
	bool attr_set = false, attr_big = false;  // the kernels' dynamic-LDS limits are raised on this handle's device
	std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending, ev_free;
};

// the window functions, src/rtl_power.c:329-408
static double window_fn(int window, int i, int length)
{
	const double n1 = (double)(length - 1);
	switch (window) {
	case RTLPOWER_WIN_HAMMING:
		return 25.0 / 46.0 - 21.0 / 46.0 * cos(2 * i * M_PI / n1);
	case RTLPOWER_WIN_BLACKMAN:
		return 7938.0 / 18608.0 - 9240.0 / 18608.0 * cos(2 * i * M_PI / n1) +
		       1430.0 / 18608.0 * cos(4 * i * M_PI / n1);
	case RTLPOWER_WIN_BLACKMAN_HARRIS:
		return 0.35875 - 0.48829 * cos(2 * i * M_PI / n1) + 0.14128 * cos(4 * i * M_PI / n1) -
		       0.01168 * cos(6 * i * M_PI / n1);
	case RTLPOWER_WIN_HANN_POISSON:
		return 0.5 * (1 - cos(2 * M_PI * i / n1)) * pow(M_E, (-2.0 * (double)abs((int)(n1 - 1 - 2 * i))) / n1);
	case RTLPOWER_WIN_YOUSSEF: {
		double w = 0.35875 - 0.48829 * cos(2 * i * M_PI / n1) + 0.14128 * cos(4 * i * M_PI / n1) -
		           0.01168 * cos(6 * i * M_PI / n1);
		return w * pow(M_E, (-0.0025 * (double)abs((int)(n1 - 1 - 2 * i))) / n1);
	}
	case RTLPOWER_WIN_BARTLETT: {
		double l = (double)length, w = (i - n1 / 2) / (l / 2);
		if (w < 0) w = -w;
		return 1 - w;
	}
	default:
		return 1.0;  // rectangle, and kaiser (a stub in the reference, :392-396)
	}
}

extern "C" int rtlpower_window_coefs(int window, int length, int32_t *out)
{
	if (!out || length < 1 || window < 0 || window > RTLPOWER_WIN_BARTLETT) return -EINVAL;
	for (int i = 0; i < length; i++)
		out[i] = (int32_t)(256 * window_fn(window, i, length));  // src/rtl_power.c:985-988
	return 0;
}

// ---- planner and CSV emitter (host only) ------------------------------------------

extern "C" int rtlpower_frequency_range(int32_t lower, int32_t upper, int32_t max_size, double crop, int boxcar,
                                        rtlpower_plan *out)
{
	// frequency_range(), src/rtl_power.c:438-540; MAXIMUM_RATE / MINIMUM_RATE :78-79, MAX_TUNES :111
	const int kMaxRate = 2800000, kMinRate = 1000000, kMaxTunes = 3000, kDefaultBuf = 16384;
	if (!out) return -EINVAL;
	int tune_count = 0, bw_seen = 0, bw_used = 0, downsample = 1, passes = 0, bin_e = 0;
	double bin_size = 0;
	for (int i = 1; i < 1500; i++) {  // evenly sized hops, as close to the maximum rate as possible
		bw_seen = (upper - lower) / i;
		bw_used = (int)((double)bw_seen / (1.0 - crop));
		if (bw_used > kMaxRate) continue;
		tune_count = i;
		break;
	}
	if (bw_used < kMinRate) {  // small bandwidth: one hop, decimated
		tune_count = 1;
		downsample = kMaxRate / bw_used;
		bw_used = bw_used * downsample;
	}
	if (!boxcar && downsample > 1) {
		passes = (int)log2((double)downsample);
		downsample = 1 << passes;
		bw_used = (int)((double)(bw_seen * downsample) / (1.0 - crop));
	}
	for (int i = 1; i <= 21; i++) {  // power-of-two bins no wider than asked
		bin_e = i;
		bin_size = (double)bw_used / (double)((1 << i) * downsample);
		if (bin_size <= (double)max_size) break;
	}
	if (max_size >= kMinRate) {  // giant bins: rms_power per hop
		bw_seen = max_size;
		bw_used = max_size;
		tune_count = (upper - lower) / bw_seen;
		bin_e = 0;
		crop = 0;
	}
	if (tune_count > kMaxTunes) return -E2BIG;
	int buf_len = 2 * (1 << bin_e) * downsample;
	if (buf_len < kDefaultBuf) buf_len = kDefaultBuf;
	out->lower = lower; out->upper = upper; out->max_size = max_size;
	out->tune_count = tune_count; out->bw_seen = bw_seen; out->rate = bw_used; out->bin_e = bin_e;
	out->downsample = downsample; out->downsample_passes = passes; out->buf_len = buf_len;
	out->crop = crop; out->bin_size = bin_size;
	return 0;
}

extern "C" int32_t rtlpower_tune_freq(const rtlpower_plan *p, int i)
{
	return p->lower + i * p->bw_seen + p->bw_seen / 2;  // src/rtl_power.c:509
}

extern "C" void rtlpower_plan_cfg(const rtlpower_plan *p, int window, int boxcar, int comp_fir_size, int peak_hold,
                                  rtlpower_cfg *c)
{
	memset(c, 0, sizeof(*c));
	c->bin_e = p->bin_e; c->window = window; c->downsample = p->downsample;
	c->downsample_passes = p->downsample_passes; c->boxcar = boxcar; c->comp_fir_size = comp_fir_size;
	c->peak_hold = peak_hold; c->buf_len = (uint32_t)p->buf_len;
}

extern "C" int rtlpower_csv_dbm(const rtlpower_plan *p, int tune, int64_t *avg, int32_t samples, char *out, size_t cap)
{
	// csv_dbm(), src/rtl_power.c:722-765
	if (!p || !avg || !out) return -EINVAL;
	const int len = 1 << p->bin_e, ds = p->downsample;
	const int freq = rtlpower_tune_freq(p, tune);
	if (p->bin_e > 0) {
		avg[0] = avg[1];  // "nuke DC component"
		for (int i = 0; i < len / 2; i++) {  // the FFT is translated by 180 degrees
			int64_t t = avg[i]; avg[i] = avg[i + len / 2]; avg[i + len / 2] = t;
		}
	}
	std::string sres;
	char tmp[96];
	const int bin_count = (int)((double)len * (1.0 - p->crop));
	const int bw2 = (int)(((double)p->rate * (double)bin_count) / (len * 2 * ds));
	snprintf(tmp, sizeof(tmp), "%i, %i, %.2f, %i, ", freq - bw2, freq + bw2, (double)p->rate / (double)(len * ds), samples);
	sres += tmp;
	const int i1 = 0 + (int)((double)len * p->crop * 0.5);
	const int i2 = (len - 1) - (int)((double)len * p->crop * 0.5);
	double dbm;
	for (int i = i1; i <= i2; i++) {
		dbm = (double)avg[i];
		dbm /= (double)p->rate;
		dbm /= (double)samples;
		dbm = 10 * log10(dbm);
		snprintf(tmp, sizeof(tmp), "%.2f, ", dbm);
		sres += tmp;
	}
	dbm = (double)avg[i2] / ((double)p->rate * (double)samples);
	if (p->bin_e == 0) dbm = ((double)avg[0] / ((double)p->rate * (double)samples));
	dbm = 10 * log10(dbm);
	snprintf(tmp, sizeof(tmp), "%.2f\n", dbm);
	sres += tmp;
	if (sres.size() + 1 > cap) return -ENOBUFS;
	memcpy(out, sres.c_str(), sres.size() + 1);
	return (int)sres.size();
}

static int validate(const rtlpower_cfg *c)
{
	if (c->bin_e < 0 || c->bin_e > 21) return -EINVAL;  // frequency_range() plans 2^1 .. 2^21 bins (src/rtl_power.c:483-486)
	if (c->window < 0 || c->window > RTLPOWER_WIN_BARTLETT) return -EINVAL;
	if (c->buf_len < 16 || c->buf_len % 4 || c->buf_len > (1u << 26)) return -EINVAL;
	if (c->downsample < 1) return -EINVAL;
	if (c->comp_fir_size != 0 && c->comp_fir_size != 9) return -EINVAL;
	if (!c->boxcar && c->downsample_passes) {
		if (c->downsample_passes < 1 || c->downsample_passes > 10) return -EINVAL;
		if (c->downsample != (1 << c->downsample_passes)) return -EINVAL;
		if (c->buf_len % (4u << c->downsample_passes)) return -EINVAL;
		if ((c->buf_len >> c->downsample_passes) < 48) return -EINVAL;  // ease-in reads x[0..8], fir 9 more
	}
	// scanner() transforms frames at offsets 0, 2N, ... < buf_len/ds (src/rtl_power.c:695).  A
	// trailing frame may run past the decimated data; inside buf_len it then sees this read's
	// own leftovers (reproduced here), but past buf_len it would see the previous read's FFT
	// output still lying in the reference's static fft_buf.  The reference's planner never
	// produces that (buf_len = 2N * downsample, at least 16384: src/rtl_power.c:501-504); it is rejected here.
	{
		const long long two_n = 2ll << c->bin_e;
		const long long per = (long long)c->buf_len / c->downsample;
		const long long frames = (per + two_n - 1) / two_n;
		if (frames * two_n > (long long)c->buf_len) return -EINVAL;
		// With fifth_order passes a partial trailing frame would cover the intermediate passes'
		// leftovers; powers of two never produce one through frequency_range(), only the boxcar
		// with an odd ratio does, and that case is reproduced.
		if (!c->boxcar && c->downsample_passes && per % two_n) return -EINVAL;
	}
	return 0;
}

extern "C" int rtlpower_cfg_validate(const rtlpower_cfg *cfg) { return cfg ? validate(cfg) : -EINVAL; }

static int power_create_body(rtlpower_gpu *h);

extern "C" int rtlpower_gpu_create(const rtlpower_cfg *cfg, int nstreams, int device, rtlpower_gpu **out)
{
	if (!cfg || !out || nstreams < 1) return -EINVAL;
	int v = validate(cfg);
	if (v < 0) return v;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
		fprintf(stderr, "rtlpower_hip: no usable HIP device (count=%d); there is no CPU fallback\n", ndev);
		return -ENODEV;
	}
	HIP_TRY(hipSetDevice(device));
	rtlpower_gpu *h = new rtlpower_gpu();
	h->cfg = *cfg;
	h->nstreams = nstreams;
	h->device = device;
	// every failure releases what was allocated so far; *out is written on success only
	int r = power_create_body(h);
	if (r < 0) {
		rtlpower_gpu_destroy(h);
		return r;
	}
	*out = h;
	return 0;
}

static int power_create_body(rtlpower_gpu *h)
{
	const rtlpower_cfg *cfg = &h->cfg;
	const int nstreams = h->nstreams;
	h->N = 1 << cfg->bin_e;
	const int ds = cfg->downsample;
	h->decimates = (cfg->boxcar && ds > 1) || (!cfg->boxcar && cfg->downsample_passes > 0);
	h->len_dec = (int)cfg->buf_len / ds;  // what remove_dc() and the chunk loop are given (:692-696)
	if (cfg->bin_e > 0) {
		h->chunks = (h->len_dec + 2 * h->N - 1) / (2 * h->N);
		// what one workgroup's LDS cannot hold goes through the work buffer in HBM (power_kernels.h, k_power_fft_*)
		h->staged = cfg->bin_e > 14 || (long long)h->chunks * h->N > kMaxPoints;
		if (cfg->boxcar && ds > 1) h->dec_elems = 2 * (((int)cfg->buf_len / 2 + ds - 1) / ds);
		else if (h->decimates) h->dec_elems = (int)cfg->buf_len >> cfg->downsample_passes;
	}
	HIP_TRY(rtl_pool::stream_get(h->device, 0, &h->own_stream));  // (from the process-wide pool: stream_pool.h)
	h->stream = h->own_stream;
	const size_t S = (size_t)nstreams;
	HIP_TRY(hipMalloc(&h->d_avg, S * h->N * sizeof(long long)));
	HIP_TRY(hipMalloc(&h->d_samples, S * sizeof(int32_t)));
	if (cfg->bin_e > 0) {
		std::vector<int32_t> w((size_t)h->N);
		rtlpower_window_coefs(cfg->window, h->N, w.data());
		HIP_TRY(hipMalloc(&h->d_window, w.size() * sizeof(int32_t)));
		HIP_TRY(hipMemcpy(h->d_window, w.data(), w.size() * sizeof(int32_t), hipMemcpyHostToDevice));
		std::vector<uint16_t> w16(w.size());
		for (size_t i = 0; i < w.size(); i++) w16[i] = (uint16_t)(uint32_t)w[i];
		HIP_TRY(hipMalloc(&h->d_window16, w16.size() * sizeof(uint16_t)));
		HIP_TRY(hipMemcpy(h->d_window16, w16.data(), w16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
		if (cfg->bin_e > 14) {
			// comb b of the frame = the points k << c | b: k_power_scan_big<14, true> reads its coefficients as it reads a read's
			const int cc = cfg->bin_e - 14;
			std::vector<uint16_t> wt(w16.size());
			for (size_t j = 0; j < w16.size(); j++) wt[((j & (((size_t)1 << cc) - 1)) << 14) | (j >> cc)] = w16[j];
			HIP_TRY(hipMalloc(&h->d_window16T, wt.size() * sizeof(uint16_t)));
			HIP_TRY(hipMemcpy(h->d_window16T, wt.data(), wt.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
		}
		// sine_table(), src/rtl_power.c:247-261, regrouped per FFT stage: stage s uses
		// wr = Sinewave[j + N/4] >> 1, wi = -Sinewave[j] >> 1 at j = m << (log2N - 1 - s)
		// for m < 2^s (:303-308); entry (1 << s) - 1 + m holds them DOUBLED and packed, (2 wr, 2 wi) - the
		// butterfly takes FIX_MPY as the high half of a 16 x 16 product (power_kernels.h); wr, wi lie in
		// -16384 .. 16383, so the doubled values still fit int16.
		std::vector<int16_t> sine((size_t)h->N * 3 / 4 + 1);
		for (int i = 0; i < h->N * 3 / 4; i++)
			sine[i] = (int16_t)(int)round(32767 * sin((double)i * 2.0 * M_PI / h->N));
		std::vector<uint32_t> tw((size_t)h->N, 0u);
		for (int st = 0; st < cfg->bin_e; st++) {
			const int k = cfg->bin_e - 1 - st;
			for (int m = 0; m < (1 << st); m++) {
				const int j = m << k;
				int16_t wr = sine[j + h->N / 4], wi = (int16_t)(-sine[j]);
				wr >>= 1; wi >>= 1;
				tw[(size_t)(1 << st) - 1 + m] = (uint32_t)(uint16_t)(wr * 2) | ((uint32_t)(uint16_t)(wi * 2) << 16);
			}
		}
		HIP_TRY(hipMalloc(&h->d_tw, tw.size() * sizeof(uint32_t)));
		HIP_TRY(hipMemcpy(h->d_tw, tw.data(), tw.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	}
	return rtlpower_gpu_clear(h);
}

extern "C" int rtlpower_gpu_destroy(rtlpower_gpu *h)
{
	if (!h) return -EINVAL;
	(void)hipSetDevice(h->device);
	if (h->stream) (void)hipStreamSynchronize(h->stream);
	for (hipEvent_t e : {h->ev_wait, h->ev_release})
		if (e) (void)hipEventDestroy(e);
	for (auto &p : h->ev_pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
	for (auto &p : h->ev_free) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
	void *ptrs[] = {h->d_window, h->d_window16, h->d_tw, h->d_avg, h->d_samples, h->d_decA, h->d_decB, h->d_one, h->d_stamps, h->d_work, h->d_ave, h->d_window16T, h->d_tbuf, h->d_part, h->d_dec32};
	for (void *p : ptrs)
		if (p) (void)hipFree(p);
	if (h->pipe.aux) (void)hipStreamSynchronize(h->pipe.aux);
	for (hipEvent_t e : {h->pipe.start, h->pipe.prep[0], h->pipe.prep[1], h->pipe.scanned[0], h->pipe.scanned[1], h->pipe.done})
		if (e) (void)hipEventDestroy(e);
	// back to the pool, never destroyed (stream_pool.h)
	rtl_pool::stream_put(h->device, h->pipe.aux_prio, h->pipe.aux);
	rtl_pool::stream_put(h->device, 0, h->own_stream);
	delete h;
	return 0;
}

extern "C" int rtlpower_gpu_clear(rtlpower_gpu *h)
{
	if (!h) return -EINVAL;
	HIP_TRY(hipSetDevice(h->device));
	HIP_TRY(hipMemsetAsync(h->d_avg, 0, (size_t)h->nstreams * h->N * sizeof(long long), h->stream));
	HIP_TRY(hipMemsetAsync(h->d_samples, 0, (size_t)h->nstreams * sizeof(int32_t), h->stream));
	return 0;
}

extern "C" int rtlpower_gpu_sync(rtlpower_gpu *h)
{
	if (!h) return -EINVAL;
	HIP_TRY(hipSetDevice(h->device));
	HIP_TRY(hipStreamSynchronize(h->stream));
	return 0;
}

extern "C" int rtlpower_gpu_set_stream(rtlpower_gpu *h, void *s)
{
	if (!h) return -EINVAL;
	HIP_TRY(hipStreamSynchronize(h->stream));
	h->stream = s ? (hipStream_t)s : h->own_stream;
	return 0;
}

// cross-stream ordering without a host synchronisation, as rtlfm_gpu_wait_for / _release_to
extern "C" int rtlpower_gpu_wait_for(rtlpower_gpu *h, void *producer_stream)
{
	if (!h) return -EINVAL;
	if ((hipStream_t)producer_stream == h->stream) return 0;
	HIP_TRY(hipSetDevice(h->device));
	if (!h->ev_wait) HIP_TRY(hipEventCreateWithFlags(&h->ev_wait, hipEventDisableTiming));
	HIP_TRY(hipEventRecord(h->ev_wait, (hipStream_t)producer_stream));
	HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_wait, 0));
	return 0;
}

extern "C" int rtlpower_gpu_release_to(rtlpower_gpu *h, void *consumer_stream)
{
	if (!h) return -EINVAL;
	if ((hipStream_t)consumer_stream == h->stream) return 0;
	HIP_TRY(hipSetDevice(h->device));
	if (!h->ev_release) HIP_TRY(hipEventCreateWithFlags(&h->ev_release, hipEventDisableTiming));
	HIP_TRY(hipEventRecord(h->ev_release, h->stream));
	HIP_TRY(hipStreamWaitEvent((hipStream_t)consumer_stream, h->ev_release, 0));
	return 0;
}

extern "C" int rtlpower_gpu_set_option(rtlpower_gpu *h, const char *name, long value)
{
	if (!h || !name) return -EINVAL;
	if (!strcmp(name, "groups")) {
		if (value < 0) return -EINVAL;
		h->groups = (int)value;
		return 0;
	}
	if (!strcmp(name, "staged_fast")) {
		h->staged_fast = value != 0;
		return 0;
	}
	if (!strcmp(name, "dec_fast")) {
		h->dec_fast = value != 0;
		return 0;
	}
	if (!strcmp(name, "staged_pipe")) {
		if (value < 0 || value > 2) return -EINVAL;
		h->staged_pipe = (int)value;
		return 0;
	}
	if (!strcmp(name, "staged_batch")) {
		if (value < 0 || value > (1 << 20)) return -EINVAL;
		h->staged_batch = (int)value;
		return 0;
	}
	if (!strcmp(name, "scan_frames")) {
		h->scan_frames = value != 0;
		return 0;
	}
	return -ENOENT;
}

extern "C" int rtlpower_gpu_get_option(rtlpower_gpu *h, const char *name, long *value)
{
	if (!h || !name || !value) return -EINVAL;
	if (!strcmp(name, "groups")) { *value = h->groups; return 0; }
	if (!strcmp(name, "staged_fast")) { *value = h->staged_fast; return 0; }
	if (!strcmp(name, "dec_fast")) { *value = h->dec_fast; return 0; }
	if (!strcmp(name, "staged_pipe")) { *value = h->staged_pipe; return 0; }
	if (!strcmp(name, "staged_batch")) { *value = h->staged_batch; return 0; }
	if (!strcmp(name, "scan_frames")) { *value = h->scan_frames; return 0; }
	if (!strcmp(name, "last_kernel")) { *value = h->last_kernel; return 0; }
	return -ENOENT;
}

// The shader clock the FFT kernel's workgroups actually ran at (k_power_scan_big stamps s_memtime and the
// 100 MHz s_memrealtime at its first and last instruction): what the VALU-issue ceiling of a launch is priced at.
extern "C" int rtlpower_gpu_clock_probe(rtlpower_gpu *h, int on)
{
	if (!h) return -EINVAL;
	h->want_stamps = on != 0;
	if (!on) h->stamp_last = 0;
	return 0;
}

extern "C" int rtlpower_gpu_clock_read(rtlpower_gpu *h, double *shader_mhz, double *span_ms)
{
	if (!h) return -EINVAL;
	HIP_TRY(hipSetDevice(h->device));
	HIP_TRY(hipStreamSynchronize(h->stream));
	if (!h->d_stamps || h->stamp_last <= 0) return -ENODATA;
	std::vector<unsigned long long> st((size_t)h->stamp_last * 4);
	HIP_TRY(hipMemcpy(st.data(), h->d_stamps, st.size() * 8, hipMemcpyDeviceToHost));
	double sum = 0; int n = 0; unsigned long long t0 = ~0ull, t1 = 0;
	for (int w = 0; w < h->stamp_last; w++) {
		const double dc = (double)(st[w * 4 + 1] - st[w * 4]), dr = (double)(st[w * 4 + 3] - st[w * 4 + 2]);
		if (st[w * 4 + 3] == 0 || dr <= 0) continue;  // a workgroup that had no reads
		sum += dc / dr * 100.0; n++;
		if (st[w * 4 + 2] < t0) t0 = st[w * 4 + 2];
		if (st[w * 4 + 3] > t1) t1 = st[w * 4 + 3];
	}
	if (!n) return -ENODATA;
	if (shader_mhz) *shader_mhz = sum / n;
	if (span_ms) *span_ms = (double)(t1 - t0) / 1e5;
	return 0;
}

extern "C" int rtlpower_gpu_timing_enable(rtlpower_gpu *h, int on)
{
	if (!h) return -EINVAL;
	h->timing = on != 0;
	return 0;
}

extern "C" int rtlpower_gpu_timing_read(rtlpower_gpu *h, double *ms, int *launches)
{
	if (!h) return -EINVAL;
	HIP_TRY(hipSetDevice(h->device));
	HIP_TRY(hipStreamSynchronize(h->stream));
	double total = 0;
	int n = 0;
	for (auto &p : h->ev_pending) {
		float t = 0;
		HIP_TRY(hipEventElapsedTime(&t, p.first, p.second));
		total += t; n++;
		h->ev_free.push_back(p);
	}
	h->ev_pending.clear();
	if (ms) *ms = total;
	if (launches) *launches = n;
	return 0;
}

static inline int grid_for(size_t work, int block = 256, int cap = 256 * 16)
{
	size_t g = (work + block - 1) / block;
	if (g < 1) g = 1;
	if (g > (size_t)cap) g = cap;
	return (int)g;
}

extern "C" int rtlpower_gpu_scan_device(rtlpower_gpu *h, const uint8_t *d_iq, size_t stream_stride, int nreads)
{
	if (!h || !d_iq || nreads < 1) return -EINVAL;
	if (stream_stride < (size_t)nreads * h->cfg.buf_len) return -EINVAL;
	HIP_TRY(hipSetDevice(h->device));
	const rtlpower_cfg &c = h->cfg;
	const int S = h->nstreams;
	hipStream_t q = h->stream;
	rtl_debug::poison_lds(q);  // RTLFM_POISON=1 only
	if (c.bin_e == 0) {
		k_power_rms<<<S, 256, 0, q>>>(d_iq, stream_stride, nreads, (int)c.buf_len, c.peak_hold, h->d_avg, h->d_samples);
		HIP_TRY(hipGetLastError());
		return 0;
	}
	// ---- a decimated scan whose reads are whole frames of at least 512 points: two launches (power_kernels.h) ----
	{
		const int Pr = h->len_dec / 2;  // points of a decimated read
		const bool pow2 = Pr >= 512 && Pr <= 8192 && (Pr & (Pr - 1)) == 0 && h->len_dec == 2 * Pr;
		const bool aligned = !(stream_stride & 15) && !((uintptr_t)d_iq & 15) && !(c.buf_len & 15);
		const bool by_fifth = !c.boxcar && c.downsample_passes >= 1 && c.downsample_passes <= kDecMaxPasses &&
		                      (int)((c.buf_len / 2) >> c.downsample_passes) == Pr;
		const bool by_boxcar = c.boxcar && c.downsample > 1 && (int)(c.buf_len / 2) == Pr * c.downsample;
		if (h->dec_fast && h->decimates && !h->staged && pow2 && aligned && c.bin_e >= 3 && h->N <= Pr && (by_fifth || by_boxcar)) {
			const size_t need = (size_t)S * nreads * Pr;
			if (h->dec32_cap < need) {
				HIP_TRY(hipStreamSynchronize(q));
				if (h->d_dec32) (void)hipFree(h->d_dec32);
				h->d_dec32 = nullptr; h->dec32_cap = 0;
				HIP_TRY(hipMalloc(&h->d_dec32, need * sizeof(uint32_t)));
				h->dec32_cap = need;
			}
			std::pair<hipEvent_t, hipEvent_t> ev;
			if (h->timing) {
				if (!h->ev_free.empty()) { ev = h->ev_free.back(); h->ev_free.pop_back(); }
				else { HIP_TRY(hipEventCreate(&ev.first)); HIP_TRY(hipEventCreate(&ev.second)); }
				HIP_TRY(hipEventRecord(ev.first, q));
			}
			if (by_fifth) {
				DecimateParams dp{};
				dp.iq8 = d_iq; dp.stride8 = stream_stride; dp.nreads = nreads; dp.buf_len = (int)c.buf_len;
				dp.passes = c.downsample_passes; dp.fir = c.comp_fir_size;
				dp.out = h->d_dec32; dp.out_stream_stride = (size_t)nreads * Pr; dp.out_per_read = Pr;
				dp.tile_out = kDecTileIn >> c.downsample_passes;
				dp.tiles_per_read = (Pr + dp.tile_out - 1) / dp.tile_out;
				dp.total_tiles = (size_t)S * nreads * dp.tiles_per_read;
				const unsigned g = (unsigned)(dp.total_tiles < 256 * 20 ? dp.total_tiles : 256 * 20);
				switch (c.downsample_passes) {
				case 1: k_power_downsample_iq<1><<<g, 256, 0, q>>>(dp); break;
				case 2: k_power_downsample_iq<2><<<g, 256, 0, q>>>(dp); break;
				case 3: k_power_downsample_iq<3><<<g, 256, 0, q>>>(dp); break;
				case 4: k_power_downsample_iq<4><<<g, 256, 0, q>>>(dp); break;
				case 5: k_power_downsample_iq<5><<<g, 256, 0, q>>>(dp); break;
				default: k_power_downsample_iq<6><<<g, 256, 0, q>>>(dp); break;
				}
			} else {
				k_power_boxcar<<<grid_for((size_t)S * nreads * Pr), 256, 0, q>>>(
				    d_iq, stream_stride, nreads, (int)c.buf_len, c.downsample, S, reinterpret_cast<int16_t *>(h->d_dec32),
				    (size_t)nreads * 2 * Pr, (size_t)2 * Pr, Pr);
			}
			ScanParams p{};
			p.dec32 = h->d_dec32; p.dec32_stream_stride = (size_t)nreads * Pr;
			for (p.dec_e = 9; (1 << p.dec_e) < Pr; p.dec_e++) {}
			p.nreads = nreads; p.bin_e = c.bin_e; p.ds = c.downsample; p.peak_hold = c.peak_hold;
			p.window16 = h->d_window16; p.tw = h->d_tw; p.avg = h->d_avg; p.samples = h->d_samples;
			constexpr int M = 8192;
			const int units = (nreads + (M / Pr) - 1) / (M / Pr);
			int groups = (512 + S - 1) / S;
			if (h->groups > 0) groups = h->groups;
			if (groups > units) groups = units;
			if (groups < 1) groups = 1;
			p.groups = groups;
			if (!h->attr_frames16) {
				HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_power_scan_frames<13, true>),
				                            hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
				h->attr_frames16 = true;
			}
			const size_t lds16 = ((size_t)skewed_size(M) + (size_t)h->N) * 4;
			hipLaunchKernelGGL((k_power_scan_frames<13, true>), dim3((unsigned)S * (unsigned)groups), dim3(kThreads), lds16, q, p);
			h->last_kernel = RTLPOWER_KERNEL_DECIMATED;
			HIP_TRY(hipGetLastError());
			if (h->timing) {
				HIP_TRY(hipEventRecord(ev.second, q));
				h->ev_pending.push_back(ev);
			}
			return 0;
		}
	}
	const int16_t *dec = nullptr;
	size_t dss = 0, drs = 0;
	if (h->decimates) {
		drs = (size_t)c.buf_len;  // elements reserved per read
		dss = drs * nreads;
		if (h->dec_cap_reads < (size_t)nreads) {
			if (h->d_decA) { (void)hipFree(h->d_decA); (void)hipFree(h->d_decB); h->d_decA = h->d_decB = nullptr; }
			HIP_TRY(hipMalloc(&h->d_decA, (size_t)S * dss * sizeof(int16_t)));
			HIP_TRY(hipMalloc(&h->d_decB, (size_t)S * dss * sizeof(int16_t)));
			h->dec_cap_reads = nreads;
		}
		int16_t *cur = h->d_decA, *oth = h->d_decB;
		if (c.boxcar) {
			const int out_cplx = h->dec_elems / 2;
			k_power_boxcar<<<grid_for((size_t)S * nreads * out_cplx), 256, 0, q>>>(
			    d_iq, stream_stride, nreads, (int)c.buf_len, c.downsample, S, cur, dss, drs, out_cplx);
		} else {
			for (int j = 0; j < c.downsample_passes; j++) {
				const int length = (int)c.buf_len >> j;
				k_power_fifth<<<grid_for((size_t)S * nreads * (length / 4)), 256, 0, q>>>(
				    j == 0 ? d_iq : nullptr, stream_stride, j == 0 ? nullptr : cur, dss, drs, nreads, (int)c.buf_len,
				    length, S, j == 0 ? cur : oth, dss, drs);
				if (j > 0) std::swap(cur, oth);
			}
			if (c.comp_fir_size == 9) {
				const int cplx = ((int)c.buf_len >> c.downsample_passes) / 2;
				k_power_fir9<<<grid_for((size_t)S * nreads * cplx), 256, 0, q>>>(cur, oth, dss, drs, nreads, cplx, S,
				                                                               c.downsample_passes);
				std::swap(cur, oth);
			}
		}
		dec = cur;
	}
	if (h->staged) {
		// bin_e 15 .. 21 (or frames beyond one workgroup's LDS): the same stages over a work buffer in HBM,
		// a batch of reads at a time (at most ~1 GiB of work buffer)
		const size_t M = (size_t)h->chunks * h->N;
		const bool fast_shape = c.bin_e > 14 && !dec && h->chunks == 1;
		// the pipeline holds two batches: half the points each, and at least four batches where each still fills the GPU
		// (measured, profiles/r05_power_big_pipe.txt: 2^19 .. 2^21 bins - two passes over HBM per batch - 4-5 % faster in two
		// sessions and 6 % slower in a third; 2^15 .. 2^17 1-3 % SLOWER: the kernels do run at the same time, but beside the
		// transform the accumulating pass takes three times as long as alone and becomes the longer chain; a scan of one batch
		// only pays for the events.  Not the default.)
		const bool two = fast_shape && h->staged_fast && (h->staged_pipe == 2 || (h->staged_pipe == 1 && c.bin_e >= 18));
		size_t batch = ((size_t)1 << (two ? 27 : 28)) / ((size_t)S * M);
		if (batch < 1) batch = 1;
		if (batch > (size_t)nreads) batch = (size_t)nreads;
		if (two && ((size_t)nreads + batch - 1) / batch < 4) {
			const size_t want = nreads < 4 ? (size_t)nreads : 4, b2 = ((size_t)nreads + want - 1) / want;
			if (((size_t)S * b2) << (c.bin_e - 14) >= 2048) batch = b2;  // (eight combs per CU and batch: 512 made a one-stream scan of 2^21 bins 1.6 x slower)
		}
		if (h->staged_batch > 0 && (size_t)h->staged_batch < batch) batch = (size_t)h->staged_batch;
		if (h->work_reads < batch || (two && !h->work_two)) {
			HIP_TRY(hipStreamSynchronize(q));
			if (h->pipe.aux) HIP_TRY(hipStreamSynchronize(h->pipe.aux));
			if (h->d_work) { (void)hipFree(h->d_work); (void)hipFree(h->d_ave); h->d_work = nullptr; h->d_ave = nullptr; }
			if (h->d_tbuf) { (void)hipFree(h->d_tbuf); (void)hipFree(h->d_part); h->d_tbuf = nullptr; h->d_part = nullptr; }
			h->work_reads = 0; h->work_two = false;
			const size_t sets = two ? 2 : 1;
			HIP_TRY(hipMalloc(&h->d_work, sets * S * batch * M * sizeof(uint32_t)));
			HIP_TRY(hipMalloc(&h->d_ave, sets * S * batch * sizeof(int2)));
			if (fast_shape) {
				HIP_TRY(hipMalloc(&h->d_tbuf, sets * S * batch * M * 2));
				HIP_TRY(hipMalloc(&h->d_part, sets * S * batch * (M / 8192) * sizeof(int2)));
			}
			h->work_reads = batch; h->work_two = two;
		}
		// rtl_power's own fine-bin shape - an undecimated read is exactly one frame - takes the kernels written for it
		const bool fast = h->staged_fast && h->d_tbuf && c.bin_e > 14 && !dec && h->chunks == 1 && h->len_dec == 2 * h->N &&
		                  c.buf_len == (uint32_t)(2 * h->N) && !(stream_stride & 15) && !((uintptr_t)d_iq & 15);
		if (fast && !h->attr_comb) {
			HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_power_scan_big<14, true>),
			                            hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
			h->attr_comb = true;
		}
		const int eb = c.bin_e < 14 ? c.bin_e : 14;
		const size_t lds = ((size_t)skewed_size(1 << eb) + ((size_t)1 << eb)) * 4;
		if (!h->attr_lds) {
			HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_power_fft_lds),
			                            hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
			h->attr_lds = true;
		}
		std::pair<hipEvent_t, hipEvent_t> ev;
		if (h->timing) {
			if (!h->ev_free.empty()) { ev = h->ev_free.back(); h->ev_free.pop_back(); }
			else { HIP_TRY(hipEventCreate(&ev.first)); HIP_TRY(hipEventCreate(&ev.second)); }
			HIP_TRY(hipEventRecord(ev.first, q));
		}
		const bool piped = fast && two && h->work_two && (h->staged_pipe == 2 || (size_t)nreads > batch);
		if (piped && !h->pipe.aux) {
			int lo = 0, hi = 0;
			HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
			HIP_TRY(rtl_pool::stream_get(h->device, lo, &h->pipe.aux));  // (a priority of its own: a hardware queue of its own)
			h->pipe.aux_prio = lo;
			for (hipEvent_t *e : {&h->pipe.start, &h->pipe.prep[0], &h->pipe.prep[1], &h->pipe.scanned[0], &h->pipe.scanned[1], &h->pipe.done})
				HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
		}
		hipStream_t qa = piped ? h->pipe.aux : q;  // where the byte-moving kernels go
		// the sets of work buffers (piped: batch k uses set k & 1; the handle's allocation holds work_reads reads per set)
		const size_t set_work = (size_t)S * h->work_reads * M, set_ave = (size_t)S * h->work_reads;
		const size_t set_tbuf = set_work * 2, set_part = (size_t)S * h->work_reads * (M / 8192);
		auto comb_of = [&](int r0, int nb, int set) {  // the comb gather + remove_dc's averages of one batch, on qa
			const size_t frames = (size_t)S * nb, tiles = frames << (c.bin_e - 13);
			k_power_comb_bytes<<<grid_for(tiles, 1, 256 * 32), 256, 0, qa>>>(d_iq + (size_t)r0 * c.buf_len, stream_stride, nb, (int)c.buf_len, c.bin_e, tiles,
			                                                                   h->d_tbuf + set * set_tbuf, h->d_part + set * set_part);
			k_power_dc_fin<<<grid_for(frames, 64), 64, 0, qa>>>(h->d_part + set * set_part, 1 << (c.bin_e - 13), h->len_dec, frames, h->d_ave + set * set_ave);
		};
		if (piped && nreads > 0) {
			// aux starts behind what the caller has queued on q (the producer of d_iq, an earlier scan's accumulation)
			HIP_TRY(hipEventRecord(h->pipe.start, q));
			HIP_TRY(hipStreamWaitEvent(qa, h->pipe.start, 0));
			comb_of(0, nreads < (int)batch ? nreads : (int)batch, 0);
			HIP_TRY(hipEventRecord(h->pipe.prep[0], qa));
		}
		int kb = 0;
		for (int r0 = 0; r0 < nreads; r0 += (int)batch, kb++) {
			const int nb = nreads - r0 < (int)batch ? nreads - r0 : (int)batch;
			const int set = piped ? (kb & 1) : 0;
			StagedParams sp{};
			sp.iq8 = dec ? nullptr : d_iq + (size_t)r0 * c.buf_len; sp.stride8 = stream_stride;
			sp.dec = dec ? dec + (size_t)r0 * drs : nullptr; sp.dec_stream_stride = dss; sp.dec_read_stride = drs; sp.dec_elems = h->dec_elems;
			sp.nreads = nb; sp.buf_len = (int)c.buf_len; sp.len_dec = h->len_dec; sp.bin_e = c.bin_e; sp.chunks = h->chunks;
			sp.ds = c.downsample; sp.peak_hold = c.peak_hold; sp.window = h->d_window; sp.tw = h->d_tw;
			sp.ave = h->d_ave + set * set_ave; sp.work = h->d_work + set * set_work; sp.avg = h->d_avg; sp.samples = h->d_samples; sp.nstreams = S;
			const size_t frames = (size_t)S * nb * h->chunks;
			if (fast) {
				if (!piped) comb_of(r0, nb, 0);
				else HIP_TRY(hipStreamWaitEvent(q, h->pipe.prep[set], 0));  // this batch's combs and averages are there (and, aux being
				                                                            // in order, the batch before the last has left this set)
				ScanParams cp{};
				cp.iq8 = h->d_tbuf + set * set_tbuf; cp.window16 = h->d_window16T; cp.tw = h->d_tw; cp.ave = sp.ave; cp.work = sp.work;
				cp.comb_c = c.bin_e - 14; cp.comb_blocks = frames << cp.comb_c;
				const size_t lds_big = ((size_t)skewed_size(16384) + 16384) * 4;
				const unsigned wgs = (unsigned)(cp.comb_blocks < 256 ? cp.comb_blocks : 256);
				hipLaunchKernelGGL((k_power_scan_big<14, true>), dim3(wgs), dim3(kThreads), lds_big, q, cp);
				if (piped) {
					HIP_TRY(hipEventRecord(h->pipe.scanned[set], q));
					// beside this batch's transform: the next batch's combs into the other set (its last user, batch kb - 1's
					// passes over HBM, is already queued on aux in front of them) ...
					if (r0 + (int)batch < nreads) {
						const int r1 = r0 + (int)batch, nb1 = nreads - r1 < (int)batch ? nreads - r1 : (int)batch;
						comb_of(r1, nb1, set ^ 1);
						HIP_TRY(hipEventRecord(h->pipe.prep[set ^ 1], qa));
					}
					// ... then, once the transform is through, this batch's stages beyond 13 and its accumulation
					HIP_TRY(hipStreamWaitEvent(qa, h->pipe.scanned[set], 0));
				}
			} else {
				k_power_dc<<<(unsigned)((size_t)S * nb), 256, 0, q>>>(sp);
				if (c.bin_e >= 12) k_power_place_tiled<<<grid_for(frames << (c.bin_e - 12), 1, 256 * 64), 256, 0, q>>>(sp);
				else k_power_place<<<grid_for((size_t)S * nb * M, 256, 256 * 64), 256, 0, q>>>(sp);
				const size_t nblk = frames << (c.bin_e - eb);
				hipLaunchKernelGGL(k_power_fft_lds, dim3((unsigned)(nblk < 4096 ? nblk : 4096)), dim3(kThreads), lds, q, h->d_work, h->d_tw, eb, nblk);
			}
			// stages 14 .. bin_e - 1 over HBM, three per pass; the last pass accumulates (k_power_fft_gl_acc)
			for (int st = 14; st < c.bin_e;) {
				const int R = c.bin_e - st >= 3 ? 3 : c.bin_e - st;
				if (st + R == c.bin_e) {
					const int g = grid_for((size_t)S << (c.bin_e - R), 256, 256 * 64);
					if (R == 3) k_power_fft_gl_acc<3><<<g, 256, 0, qa>>>(sp, st);
					else if (R == 2) k_power_fft_gl_acc<2><<<g, 256, 0, qa>>>(sp, st);
					else k_power_fft_gl_acc<1><<<g, 256, 0, qa>>>(sp, st);
				} else {
					const int g = grid_for(frames << (c.bin_e - R), 256, 256 * 64);
					k_power_fft_gl<3><<<g, 256, 0, qa>>>(sp.work, h->d_tw, c.bin_e, st, frames);
				}
				st += R;
			}
			if (c.bin_e <= 14) k_power_accum<<<grid_for((size_t)S * h->N, 256, 256 * 64), 256, 0, q>>>(sp);
			h->last_kernel = fast ? RTLPOWER_KERNEL_STAGED_FAST : RTLPOWER_KERNEL_STAGED;
		}
		if (piped) {  // the scan is complete on q when aux is through
			HIP_TRY(hipEventRecord(h->pipe.done, qa));
			HIP_TRY(hipStreamWaitEvent(q, h->pipe.done, 0));
		}
		HIP_TRY(hipGetLastError());
		if (h->timing) {
			HIP_TRY(hipEventRecord(ev.second, q));
			h->ev_pending.push_back(ev);
		}
		return 0;
	}
	ScanParams p{};
	p.iq8 = dec ? nullptr : d_iq; p.stride8 = stream_stride;
	p.dec = dec; p.dec_stream_stride = dss; p.dec_read_stride = drs; p.dec_elems = h->dec_elems;
	p.nreads = nreads; p.buf_len = (int)c.buf_len; p.len_dec = h->len_dec;
	p.bin_e = c.bin_e; p.chunks = h->chunks; p.ds = c.downsample; p.peak_hold = c.peak_hold;
	p.window = h->d_window; p.window16 = h->d_window16; p.tw = h->d_tw; p.avg = h->d_avg; p.samples = h->d_samples;
	const size_t lds = ((size_t)skewed_size(h->chunks * h->N) + (size_t)h->N) * 4;
	// per handle, not per process: the attribute belongs to the device the handle lives on, and one process
	// may hold handles on several
	if (!h->attr_set) {
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_power_scan),
		                            hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
		h->attr_set = true;
	}
	std::pair<hipEvent_t, hipEvent_t> ev;
	if (h->timing) {
		if (!h->ev_free.empty()) { ev = h->ev_free.back(); h->ev_free.pop_back(); }
		else { HIP_TRY(hipEventCreate(&ev.first)); HIP_TRY(hipEventCreate(&ev.second)); }
		HIP_TRY(hipEventRecord(ev.first, q));
	}
	const bool big = !dec && h->chunks == 1 && h->len_dec == 2 * h->N && (c.bin_e == 13 || c.bin_e == 14) &&
	                 !(stream_stride & 15) && !((uintptr_t)d_iq & 15) && !(c.buf_len & 15);
	if (big && !h->attr_big) {
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_power_scan_big<13>),
		                            hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_power_scan_big<14>),
		                            hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
		h->attr_big = true;
	}
	// enough workgroups for the 256 CUs (a large-FFT workgroup fills a CU's LDS, the small ones share):
	// a stream's reads are split when there are few streams (power_kernels.h)
	{
		const int want = 512;
		int groups = (want + S - 1) / S;
		if (groups > nreads) groups = nreads;
		if (groups < 1) groups = 1;
		if (h->groups > 0) groups = h->groups < nreads ? h->groups : nreads;
		p.groups = groups;
	}
	const unsigned grid = (unsigned)S * (unsigned)p.groups;
	if (big && h->want_stamps) {
		if (h->stamp_cap < (int)grid) {
			if (h->d_stamps) (void)hipFree(h->d_stamps);
			h->d_stamps = nullptr; h->stamp_cap = 0;
			HIP_TRY(hipMalloc(&h->d_stamps, (size_t)grid * 32));
			h->stamp_cap = (int)grid;
		}
		HIP_TRY(hipMemsetAsync(h->d_stamps, 0, (size_t)grid * 32, q));
		p.stamps = h->d_stamps;
		h->stamp_last = (int)grid;
	}
	// several frames per read (rtl_power's everyday shape: every scan below 8192 bins reads 16384 bytes): raw input,
	// reads of exactly 8192 or 16384 points
	const int M = h->chunks * h->N;
	const bool frames = h->scan_frames && !dec && h->chunks > 1 && c.bin_e >= 3 && (M == 8192 || M == 16384) && h->len_dec == 2 * M &&
	                    c.buf_len == (uint32_t)(2 * M) && !(stream_stride & 15) && !((uintptr_t)d_iq & 15);
	if (frames && !h->attr_frames) {
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_power_scan_frames<13>),
		                            hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
		HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_power_scan_frames<14>),
		                            hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
		h->attr_frames = true;
	}
	if (big && c.bin_e == 14) hipLaunchKernelGGL(k_power_scan_big<14>, dim3(grid), dim3(kThreads), lds, q, p);
	else if (big) hipLaunchKernelGGL(k_power_scan_big<13>, dim3(grid), dim3(kThreads), lds, q, p);
	else if (frames && M == 8192) hipLaunchKernelGGL(k_power_scan_frames<13>, dim3(grid), dim3(kThreads), lds, q, p);
	else if (frames) hipLaunchKernelGGL(k_power_scan_frames<14>, dim3(grid), dim3(kThreads), lds, q, p);
	else hipLaunchKernelGGL(k_power_scan, dim3(grid), dim3(kThreads), lds, q, p);
	h->last_kernel = big ? RTLPOWER_KERNEL_BIG : frames ? RTLPOWER_KERNEL_FRAMES : RTLPOWER_KERNEL_GENERAL;
	HIP_TRY(hipGetLastError());
	if (h->timing) {
		HIP_TRY(hipEventRecord(ev.second, q));
		h->ev_pending.push_back(ev);
	}
	return 0;
}

extern "C" int rtlpower_gpu_scan(rtlpower_gpu *h, int stream, const uint8_t *buf, uint32_t len)
{
	// one tuning_state, one rtlsdr_read_sync() buffer (src/rtl_power.c:657): run it
	// through a one-stream view of the handle
	if (!h || !buf || stream < 0 || stream >= h->nstreams || len != h->cfg.buf_len) return -EINVAL;
	HIP_TRY(hipSetDevice(h->device));
	if (!h->d_one) HIP_TRY(hipMalloc(&h->d_one, h->cfg.buf_len));
	HIP_TRY(hipMemcpyAsync(h->d_one, buf, len, hipMemcpyHostToDevice, h->stream));
	rtlpower_gpu view = *h;  // shallow: same device buffers, shifted to this stream
	view.nstreams = 1;
	view.d_avg = h->d_avg + (size_t)stream * h->N;
	view.d_samples = h->d_samples + stream;
	view.d_decA = view.d_decB = nullptr; view.dec_cap_reads = 0;
	view.d_work = nullptr; view.d_ave = nullptr; view.work_reads = 0; view.work_two = false;
	view.d_tbuf = nullptr; view.d_part = nullptr;
	view.d_dec32 = nullptr; view.dec32_cap = 0;
	view.want_stamps = false;  // (the stamp buffer would be the view's: one leak per scan, and nothing for rtlpower_gpu_clock_read to find)
	view.ev_pending.clear(); view.ev_free.clear(); view.timing = false;
	int r = rtlpower_gpu_scan_device(&view, h->d_one, h->cfg.buf_len, 1);
	// what the view learnt about this device stays learnt (the kernels' dynamic-LDS limits are raised once per handle)
	h->attr_set = view.attr_set; h->attr_big = view.attr_big; h->attr_lds = view.attr_lds; h->attr_comb = view.attr_comb;
	h->attr_frames = view.attr_frames; h->attr_frames16 = view.attr_frames16;
	h->last_kernel = view.last_kernel;
	h->pipe = view.pipe;  // (created by whoever needed it first)
	const hipError_t e = hipStreamSynchronize(h->stream);
	if (view.d_decA) { (void)hipFree(view.d_decA); (void)hipFree(view.d_decB); }  // also when the sync failed
	if (view.d_work) { (void)hipFree(view.d_work); (void)hipFree(view.d_ave); }
	if (view.d_tbuf) { (void)hipFree(view.d_tbuf); (void)hipFree(view.d_part); }
	if (view.d_dec32) (void)hipFree(view.d_dec32);
	if (e != hipSuccess) return -EIO;
	return r;
}

extern "C" int rtlpower_gpu_fetch(rtlpower_gpu *h, int stream, int64_t *avg, int32_t *samples)
{
	if (!h || stream < 0 || stream >= h->nstreams) return -EINVAL;
	HIP_TRY(hipSetDevice(h->device));
	HIP_TRY(hipStreamSynchronize(h->stream));
	if (avg) HIP_TRY(hipMemcpy(avg, h->d_avg + (size_t)stream * h->N, (size_t)h->N * sizeof(long long), hipMemcpyDeviceToHost));
	if (samples) HIP_TRY(hipMemcpy(samples, h->d_samples + stream, sizeof(int32_t), hipMemcpyDeviceToHost));
	return 0;
}
