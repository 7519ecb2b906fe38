/*
 * rtlpower_oracle.c — CPU restatement of rtl_power's scanner() DSP.
 * TEST INFRASTRUCTURE ONLY — never linked into or called by the HIP product.
 * Written from the behaviour of /root/reference/src/rtl_power.c; every function
 * cites the lines it follows.  See rtlpower_oracle.h for how it is pinned.
 */
#include "rtlpower_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_E
#define M_E 2.7182818284590452354
#endif

static const int cic9p[11][10] = {
	/* cic_9_tables, src/rtl_power.c:220-232 (same numbers as src/rtl_fm.c:355-367) */
	{0},
	{9, -156, -97, 2798, -15489, 61019, -15489, 2798, -97, -156},
	{9, -128, -568, 5593, -24125, 74126, -24125, 5593, -568, -128},
	{9, -129, -639, 6187, -26281, 77511, -26281, 6187, -639, -129},
	{9, -122, -612, 6082, -26353, 77818, -26353, 6082, -612, -122},
	{9, -120, -602, 6015, -26269, 77757, -26269, 6015, -602, -120},
	{9, -120, -582, 5951, -26128, 77542, -26128, 5951, -582, -120},
	{9, -119, -580, 5931, -26094, 77505, -26094, 5931, -580, -119},
	{9, -119, -578, 5921, -26077, 77484, -26077, 5921, -578, -119},
	{9, -119, -577, 5917, -26067, 77473, -26067, 5917, -577, -119},
	{9, -199, -362, 5303, -25505, 77489, -25505, 5303, -362, -199},
};

int16_t *orcp_sine_table(int log2n)
{
	/* src/rtl_power.c:247-261: three quarters of a period, round(32767 sin) */
	int n = 1 << log2n;
	int16_t *t = (int16_t *)malloc(sizeof(int16_t) * (size_t)(n * 3 / 4 + 1));
	for (int i = 0; i < n * 3 / 4; i++)
		t[i] = (int16_t)(int)round(32767 * sin((double)i * 2.0 * M_PI / n));
	return t;
}

static inline int16_t fix_mpy(int16_t a, int16_t b)
{
	/* FIX_MPY, src/rtl_power.c:263-269: Q15 product rounded to nearest through
	 * the bit below the result */
	int c = ((int)a * (int)b) >> 14;
	return (int16_t)((c >> 1) + (c & 1));
}

int orcp_fix_fft(int16_t *iq, int m, const int16_t *sinewave, int log2_n_wave)
{
	/* src/rtl_power.c:271-327: radix-2 decimation in time, every stage halves
	 * (shift is always 1), twiddles halved too, int16 wrap on every store */
	const int n = 1 << m, n_wave = 1 << log2_n_wave;
	if (n > n_wave)
		return -1;
	/* bit-reversal reordering (:282-297) — a counter that adds in reversed order */
	int rev = 0;
	for (int idx = 1; idx <= n - 1; idx++) {
		int bit = n;
		do {
			bit >>= 1;
		} while (rev + bit > n - 1);
		rev = (rev & (bit - 1)) + bit;
		if (rev <= idx)
			continue;
		int16_t tr = iq[2 * idx], ti = iq[2 * idx + 1];
		iq[2 * idx] = iq[2 * rev]; iq[2 * idx + 1] = iq[2 * rev + 1];
		iq[2 * rev] = tr; iq[2 * rev + 1] = ti;
	}
	int k = log2_n_wave - 1;
	for (int half = 1; half < n; half <<= 1, k--) {
		const int step = half << 1;
		for (int q = 0; q < half; q++) {
			const int j = q << k;
			int16_t wr = sinewave[j + n_wave / 4];
			int16_t wi = (int16_t)(-sinewave[j]);
			wr >>= 1; wi >>= 1;
			for (int i = q; i < n; i += step) {
				const int p = i + half;
				int16_t tr = (int16_t)(fix_mpy(wr, iq[2 * p]) - fix_mpy(wi, iq[2 * p + 1]));
				int16_t ti = (int16_t)(fix_mpy(wr, iq[2 * p + 1]) + fix_mpy(wi, iq[2 * p]));
				int16_t qr = (int16_t)(iq[2 * i] >> 1), qi = (int16_t)(iq[2 * i + 1] >> 1);
				iq[2 * p] = (int16_t)(qr - tr);
				iq[2 * p + 1] = (int16_t)(qi - ti);
				iq[2 * i] = (int16_t)(qr + tr);
				iq[2 * i + 1] = (int16_t)(qi + ti);
			}
		}
	}
	return 0;
}

double orcp_window(int window, int i, int length)
{
	/* src/rtl_power.c:329-408 */
	const double n1 = (double)(length - 1);
	switch (window) {
	case RTLPOWER_WIN_HAMMING:
		return 25.0 / 46.0 - 21.0 / 46.0 * cos(2 * i * M_PI / n1);
	case RTLPOWER_WIN_BLACKMAN:
		return 7938.0 / 18608.0 - 9240.0 / 18608.0 * cos(2 * i * M_PI / n1) + 1430.0 / 18608.0 * cos(4 * i * M_PI / n1);
	case RTLPOWER_WIN_BLACKMAN_HARRIS:
		return 0.35875 - 0.48829 * cos(2 * i * M_PI / n1) + 0.14128 * cos(4 * i * M_PI / n1) - 0.01168 * cos(6 * i * M_PI / n1);
	case RTLPOWER_WIN_HANN_POISSON:
		return 0.5 * (1 - cos(2 * M_PI * i / n1)) * pow(M_E, (-2.0 * (double)abs((int)(n1 - 1 - 2 * i))) / n1);
	case RTLPOWER_WIN_YOUSSEF: {
		double w = 0.35875 - 0.48829 * cos(2 * i * M_PI / n1) + 0.14128 * cos(4 * i * M_PI / n1) - 0.01168 * cos(6 * i * M_PI / n1);
		return w * pow(M_E, (-0.0025 * (double)abs((int)(n1 - 1 - 2 * i))) / n1);
	}
	case RTLPOWER_WIN_BARTLETT: {
		double l = (double)length, w = (i - n1 / 2) / (l / 2);
		if (w < 0) w = -w;
		return 1 - w;
	}
	case RTLPOWER_WIN_RECTANGLE:
	case RTLPOWER_WIN_KAISER:
	default:
		return 1.0;
	}
}

void orcp_window_coefs(int window, int length, int32_t *out)
{
	/* src/rtl_power.c:985-988 */
	for (int i = 0; i < length; i++)
		out[i] = (int32_t)(256 * orcp_window(window, i, length));
}

void orcp_fifth_order(int16_t *data, int length)
{
	/*
	 * src/rtl_power.c:554-579, the stateless variant.  With x[k] = data[2k]:
	 * three "ease-in" outputs from x[0..5], and because the loop then starts at
	 * i = 12 re-reading data[10], outputs 3 and 4 see x[5] twice:
	 *   y3 <- (x2, x3, x4, x5, x5, x6)   y4 <- (x4, x5, x5, x6, x7, x8)
	 * from y5 on it is the plain window x[2m-5 .. 2m].
	 */
	int x[6];
	for (int k = 0; k < 6; k++) x[k] = data[2 * k];
	data[0] = (int16_t)(((x[0] + x[1]) * 10 + (x[2] + x[3]) * 5 + x[3] + x[5]) >> 4);
	data[2] = (int16_t)(((x[1] + x[2]) * 10 + (x[0] + x[3]) * 5 + x[4] + x[5]) >> 4);
	data[4] = (int16_t)((x[0] + (x[1] + x[4]) * 5 + (x[2] + x[3]) * 10 + x[5]) >> 4);
	int w[6] = {x[0], x[1], x[2], x[3], x[4], x[5]};
	for (int m = 3; 4 * m < length; m++) {
		w[0] = w[2]; w[1] = w[3]; w[2] = w[4]; w[3] = w[5];
		w[4] = data[4 * m - 2];
		w[5] = data[4 * m];
		data[2 * m] = (int16_t)((w[0] + (w[1] + w[4]) * 5 + (w[2] + w[3]) * 10 + w[5]) >> 4);
	}
}

void orcp_generic_fir(int16_t *data, int length, int passes)
{
	/* src/rtl_power.c:598-626: the first nine samples pass unfiltered and seed the
	 * history; afterwards the 9-tap sum over the nine samples before the current */
	const int *t = cic9p[passes < 0 || passes > 10 ? 0 : passes];
	int h[9];
	for (int k = 0; k < 9; k++) h[k] = data[2 * k];
	for (int d = 18; d < length; d += 2) {
		int incoming = data[d];
		uint32_t acc = 0;
		acc += (uint32_t)(h[0] + h[8]) * (uint32_t)t[1];
		acc += (uint32_t)(h[1] + h[7]) * (uint32_t)t[2];
		acc += (uint32_t)(h[2] + h[6]) * (uint32_t)t[3];
		acc += (uint32_t)(h[3] + h[5]) * (uint32_t)t[4];
		acc += (uint32_t)h[4] * (uint32_t)t[5];
		data[d] = (int16_t)((int32_t)acc >> 15);
		memmove(h, h + 1, 8 * sizeof(int));
		h[8] = incoming;
	}
}

void orcp_remove_dc(int16_t *data, int length)
{
	/* src/rtl_power.c:581-596: the sum runs over every other element but is
	 * divided by `length`, so only half of the DC is removed */
	long sum = 0;
	for (int i = 0; i < length; i += 2) sum += data[i];
	int16_t ave = (int16_t)(sum / (long)length);
	if (ave == 0) return;
	for (int i = 0; i < length; i += 2) data[i] = (int16_t)(data[i] - ave);
}

void orcp_rms_power(const uint8_t *buf, int buf_len, int peak_hold, int64_t *avg0, int32_t *samples)
{
	/* src/rtl_power.c:410-436 */
	long p = 0, t = 0;
	for (int i = 0; i < buf_len; i++) {
		int s = (int)buf[i] - 127;
		t += s;
		p += (long)(s * s);
	}
	double dc = (double)t / (double)buf_len;
	double err = t * 2 * dc - dc * dc * buf_len;
	p -= (long)round(err);
	if (!peak_hold) *avg0 += p;
	else if (p > *avg0) *avg0 = p;
	*samples += 1;
}

int orcp_scan(const rtlpower_cfg *cfg, const uint8_t *buf8, int64_t *avg, int32_t *samples)
{
	/* scanner(), src/rtl_power.c:657-718, for one tuning_state and one read */
	const int buf_len = (int)cfg->buf_len;
	const int bin_e = cfg->bin_e, bin_len = 1 << bin_e;
	if (bin_len == 1) {
		orcp_rms_power(buf8, buf_len, cfg->peak_hold, &avg[0], samples);
		return 0;
	}
	static __thread int16_t *fft_buf;
	static __thread int fft_cap;
	static __thread int16_t *sine;
	static __thread int sine_e = -1;
	static __thread int32_t *coefs;
	static __thread int coefs_key = -1;
	if (fft_cap < buf_len) {
		free(fft_buf);
		fft_buf = (int16_t *)malloc(sizeof(int16_t) * (size_t)buf_len + 64);
		fft_cap = buf_len;
	}
	if (sine_e != bin_e) { free(sine); sine = orcp_sine_table(bin_e); sine_e = bin_e; }
	if (coefs_key != bin_e * 16 + cfg->window) {
		free(coefs);
		coefs = (int32_t *)malloc(sizeof(int32_t) * (size_t)bin_len);
		orcp_window_coefs(cfg->window, bin_len, coefs);
		coefs_key = bin_e * 16 + cfg->window;
	}
	for (int j = 0; j < buf_len; j++) fft_buf[j] = (int16_t)((int16_t)buf8[j] - 127);  /* :666-668 */
	const int ds = cfg->downsample, ds_p = cfg->downsample_passes;
	if (cfg->boxcar && ds > 1) {
		/* :671-681: sums of ds consecutive samples compacted to the front, int16 wrap */
		int j2 = 0;
		for (int j = 2; j < buf_len; ) {
			fft_buf[j2] = (int16_t)(fft_buf[j2] + fft_buf[j]);
			fft_buf[j2 + 1] = (int16_t)(fft_buf[j2 + 1] + fft_buf[j + 1]);
			fft_buf[j] = 0; fft_buf[j + 1] = 0;
			j += 2;
			if (j % (ds * 2) == 0) j2 += 2;
		}
	} else if (ds_p) {
		/* :682-691 */
		int j;
		for (j = 0; j < ds_p; j++) {
			orcp_fifth_order(fft_buf, buf_len >> j);       /* downsample_iq, :628-634 */
			orcp_fifth_order(fft_buf + 1, (buf_len >> j) - 1);
		}
		if (cfg->comp_fir_size == 9 && ds_p <= 10) {
			orcp_generic_fir(fft_buf, buf_len >> j, ds_p);
			orcp_generic_fir(fft_buf + 1, (buf_len >> j) - 1, ds_p);
		}
	}
	orcp_remove_dc(fft_buf, buf_len / ds);           /* :692-693 */
	orcp_remove_dc(fft_buf + 1, (buf_len / ds) - 1);
	for (int offset = 0; offset < buf_len / ds; offset += 2 * bin_len) {
		for (int j = 0; j < bin_len; j++) {          /* :697-706, int16 wrap */
			fft_buf[offset + 2 * j] = (int16_t)((int32_t)fft_buf[offset + 2 * j] * coefs[j]);
			fft_buf[offset + 2 * j + 1] = (int16_t)((int32_t)fft_buf[offset + 2 * j + 1] * coefs[j]);
		}
		orcp_fix_fft(fft_buf + offset, bin_e, sine, bin_e);
		for (int j = 0; j < bin_len; j++) {          /* real_conj :636-640, :708-716 */
			long re = fft_buf[offset + 2 * j], im = fft_buf[offset + 2 * j + 1];
			int64_t p = re * re + im * im;
			if (!cfg->peak_hold) avg[j] += p;
			else if (p > avg[j]) avg[j] = p;
		}
		*samples += ds;                               /* :717 */
	}
	return 0;
}

struct pjob {
	const rtlpower_cfg *cfg;
	const uint8_t *iq;
	size_t stride;
	int nreads, s0, s1;
	int64_t *avg;
	int32_t *samples;
};

static void *pworker(void *arg)
{
	struct pjob *j = (struct pjob *)arg;
	const int bins = 1 << j->cfg->bin_e;
	for (int s = j->s0; s < j->s1; s++)
		for (int r = 0; r < j->nreads; r++)
			orcp_scan(j->cfg, j->iq + (size_t)s * j->stride + (size_t)r * j->cfg->buf_len,
			          j->avg + (size_t)s * bins, j->samples + s);
	return NULL;
}

int orcp_scan_batch(const rtlpower_cfg *cfg, int nstreams, const uint8_t *iq, size_t stream_stride,
                    int nreads, int64_t *avg, int32_t *samples, int nthreads)
{
	if (nthreads < 1) nthreads = 1;
	if (nthreads > nstreams) nthreads = nstreams;
	struct pjob *jobs = (struct pjob *)calloc((size_t)nthreads, sizeof(*jobs));
	pthread_t *tid = (pthread_t *)calloc((size_t)nthreads, sizeof(*tid));
	int per = (nstreams + nthreads - 1) / nthreads;
	for (int t = 0; t < nthreads; t++) {
		jobs[t] = (struct pjob){cfg, iq, stream_stride, nreads, t * per,
		                        (t + 1) * per < nstreams ? (t + 1) * per : nstreams, avg, samples};
		if (jobs[t].s0 > jobs[t].s1) jobs[t].s0 = jobs[t].s1;
		if (nthreads == 1) pworker(&jobs[t]);
		else pthread_create(&tid[t], NULL, pworker, &jobs[t]);
	}
	for (int t = 0; t < nthreads && nthreads > 1; t++) pthread_join(tid[t], NULL);
	free(jobs); free(tid);
	return 0;
}
