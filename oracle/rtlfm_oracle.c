/*
 * rtlfm_oracle.c — CPU restatement of rtl_fm's demod chain (see header).
 *
 * TEST INFRASTRUCTURE ONLY — never linked into or called by the HIP product.
 *
 * Written from the behaviour of /root/reference/src/rtl_fm.c; every function
 * names the lines it follows.  Where the reference relies on signed overflow
 * or shifts of negative values (undefined in ISO C but two's-complement wrap
 * in the gcc build used as the pin), this file spells the wrap out through
 * uint32_t so that -O2, -O3 and sanitizer builds agree.
 */
#include "rtlfm_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static inline int32_t wrap_mul(int32_t a, int32_t b)
{
	return (int32_t)((uint32_t)a * (uint32_t)b);
}
static inline int32_t wrap_add(int32_t a, int32_t b)
{
	return (int32_t)((uint32_t)a + (uint32_t)b);
}
static inline int32_t wrap_sub(int32_t a, int32_t b)
{
	return (int32_t)((uint32_t)a - (uint32_t)b);
}

void orc_state_init(rtlfm_stream_state *st)
{
	/* demod_init(): every carried scalar is 0 except squelch_hits = 11
	 * (src/rtl_fm.c:1615); the history arrays live in BSS. */
	memset(st, 0, sizeof(*st));
	st->squelch_hits = 11;
}

/* ------------------------------------------------------------------------ */

void orc_u8_to_i16(const uint8_t *in, int16_t *out, int len)
{
	/* src/rtl_fm.c:1326-1328: offset is 127, not 127.5 or 128 */
	for (int k = 0; k < len; k++)
		out[k] = (int16_t)((int)in[k] - 127);
}

void orc_rotate16_neg90(int16_t *buf, int len)
{
	/* src/rtl_fm.c:424-434: sample n is multiplied by (-j)^n, the phase
	 * restarting at every call; the loop advances 4 complex samples. */
	for (int base = 0; base < len; base += 8) {
		int16_t re, im;
		/* n%4 == 1: (a + jb)(-j) = b - ja */
		re = buf[base + 2]; im = buf[base + 3];
		buf[base + 2] = im;
		buf[base + 3] = (int16_t)(-re);
		/* n%4 == 2: -(a + jb) */
		buf[base + 4] = (int16_t)(-buf[base + 4]);
		buf[base + 5] = (int16_t)(-buf[base + 5]);
		/* n%4 == 3: (a + jb)(+j) = -b + ja */
		re = buf[base + 6]; im = buf[base + 7];
		buf[base + 6] = (int16_t)(-im);
		buf[base + 7] = re;
	}
}

void orc_rotate_90_u8(uint8_t *buf, int len)
{
	/* src/rtl_fm.c:437-447 with NEG_U8(x) = 255 - x (:379) */
	for (int base = 0; base < len; base += 8) {
		uint8_t re, im;
		re = buf[base + 2]; im = buf[base + 3];     /* * +j */
		buf[base + 2] = (uint8_t)(255 - im);
		buf[base + 3] = re;
		buf[base + 4] = (uint8_t)(255 - buf[base + 4]); /* * -1 */
		buf[base + 5] = (uint8_t)(255 - buf[base + 5]);
		re = buf[base + 6]; im = buf[base + 7];     /* * -j */
		buf[base + 6] = im;
		buf[base + 7] = (uint8_t)(255 - re);
	}
}

void orc_dc_block_raw(int16_t *buf, int len, int k, int32_t *avg_i, int32_t *avg_q)
{
	/* src/rtl_fm.c:1043-1065 */
	int64_t acc_i = 0, acc_q = 0;
	int pairs = len / 2;
	for (int n = 0; n < pairs; n++) {
		acc_i += buf[2 * n];
		acc_q += buf[2 * n + 1];
	}
	int mean_i = (int)(acc_i / pairs);
	int mean_q = (int)(acc_q / pairs);
	mean_i = (mean_i + *avg_i * k) / (k + 1);
	mean_q = (mean_q + *avg_q * k) / (k + 1);
	for (int n = 0; n < pairs; n++) {
		buf[2 * n] = (int16_t)(buf[2 * n] - mean_i);
		buf[2 * n + 1] = (int16_t)(buf[2 * n + 1] - mean_q);
	}
	*avg_i = mean_i;
	*avg_q = mean_q;
}

int orc_low_pass(int16_t *lp, int lp_len, int downsample, int32_t *now_r,
                 int32_t *now_j, int32_t *prev_index)
{
	/* src/rtl_fm.c:461-481: boxcar SUM (no divide) of `downsample` complex
	 * samples; the partial sum and its fill count cross block boundaries. */
	int produced = 0;
	for (int n = 0; 2 * n < lp_len; n++) {
		*now_r = wrap_add(*now_r, lp[2 * n]);
		*now_j = wrap_add(*now_j, lp[2 * n + 1]);
		*prev_index += 1;
		if (*prev_index >= downsample) {
			lp[2 * produced] = (int16_t)*now_r;
			lp[2 * produced + 1] = (int16_t)*now_j;
			produced++;
			*prev_index = 0;
			*now_r = 0;
			*now_j = 0;
		}
	}
	return 2 * produced;
}

void orc_fifth_order(int16_t *data, int length, int16_t hist[6])
{
	/*
	 * src/rtl_fm.c:777-806.  With x[k] = data[2k] and x[-1..-5] = hist[5..1]:
	 *   y[m] = (x[2m-5] + 5 x[2m-4] + 10 x[2m-3] + 10 x[2m-2] + 5 x[2m-1] + x[2m]) >> 4
	 * written to data[2m].  The six-sample window of the LAST output is what
	 * is archived (:800-805), i.e. x[2M-7 .. 2M-2]; the final input sample
	 * x[2M-1] is never kept, so the first three outputs of the next call see
	 * a history that is one sample older than a continuous filter would use.
	 */
	/* the first output is produced unconditionally (:784-787), even when the
	 * caller passes length <= 0 (block_len < 2^(passes+1), a degenerate set-up) */
	int outputs = length > 0 ? (length + 3) / 4 : 1;
	int16_t win[6];
	win[0] = hist[1]; win[1] = hist[2]; win[2] = hist[3];
	win[3] = hist[4]; win[4] = hist[5]; win[5] = data[0];
	for (int m = 0; m < outputs; m++) {
		if (m > 0) {
			win[0] = win[2]; win[1] = win[3];
			win[2] = win[4]; win[3] = win[5];
			win[4] = data[4 * m - 2];
			win[5] = data[4 * m];
		}
		int acc = win[0] + win[5] + 5 * (win[1] + win[4]) + 10 * (win[2] + win[3]);
		data[2 * m] = (int16_t)(acc >> 4);
	}
	memcpy(hist, win, sizeof(win));
}

static const int cic9[11][10] = {
	/* cic_9_tables, src/rtl_fm.c:355-367: {length, 9 taps} scaled by 2^15 */
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

const int *orc_cic9_row(int passes)
{
	if (passes < 0 || passes > 10)
		return cic9[0];
	return cic9[passes];
}

void orc_generic_fir(int16_t *data, int length, int passes, int16_t hist[9])
{
	/*
	 * src/rtl_fm.c:808-831: the output at position n is the symmetric 9-tap
	 * sum over the nine samples BEFORE n (the current sample only enters the
	 * history afterwards), >> 15, stored as int16.  Only taps 1..5 of the row
	 * are read; symmetry supplies the rest.
	 */
	const int *t = orc_cic9_row(passes);
	for (int d = 0; d < length; d += 2) {
		int16_t incoming = data[d];
		int32_t acc = 0;
		acc = wrap_add(acc, wrap_mul(hist[0] + hist[8], t[1]));
		acc = wrap_add(acc, wrap_mul(hist[1] + hist[7], t[2]));
		acc = wrap_add(acc, wrap_mul(hist[2] + hist[6], t[3]));
		acc = wrap_add(acc, wrap_mul(hist[3] + hist[5], t[4]));
		acc = wrap_add(acc, wrap_mul(hist[4], t[5]));
		data[d] = (int16_t)(acc >> 15);
		memmove(hist, hist + 1, 8 * sizeof(int16_t));
		hist[8] = incoming;
	}
}

/* ---- discriminators ------------------------------------------------------ */

static inline void conj_product(int ar, int aj, int br, int bj, int32_t *cr, int32_t *cj)
{
	/* multiply(ar, aj, br, -bj), src/rtl_fm.c:836-840 */
	*cr = wrap_sub(wrap_mul(ar, br), wrap_mul(aj, -bj));
	*cj = wrap_add(wrap_mul(aj, br), wrap_mul(ar, -bj));
}

int orc_polar_discriminant(int ar, int aj, int br, int bj)
{
	/* src/rtl_fm.c:842-849: note the literal 3.14159 and the truncation */
	int32_t cr, cj;
	conj_product(ar, aj, br, bj, &cr, &cj);
	double angle = atan2((double)cj, (double)cr);
	return (int)(angle / 3.14159 * (1 << 14));
}

static int fast_atan2_restated(int32_t y, int32_t x)
{
	/* src/rtl_fm.c:851-872 (pi = 1<<14).  The products 4096*(x -/+ |y|) wrap
	 * in 32 bits once |x -/+ |y|| exceeds 2^19. */
	if (x == 0 && y == 0)
		return 0;
	int32_t ay = y < 0 ? wrap_sub(0, y) : y;
	int32_t angle;
	if (x >= 0) {
		int32_t den = wrap_add(x, ay);
		angle = 4096 - (den ? wrap_mul(4096, wrap_sub(x, ay)) / den : 0);
	} else {
		int32_t den = wrap_sub(ay, x);
		angle = 12288 - (den ? wrap_mul(4096, wrap_add(x, ay)) / den : 0);
	}
	return y < 0 ? -angle : angle;
}

int orc_fast_atan2(int y, int x) { return fast_atan2_restated(y, x); }

int orc_polar_disc_fast(int ar, int aj, int br, int bj)
{
	/* src/rtl_fm.c:874-879 */
	int32_t cr, cj;
	conj_product(ar, aj, br, bj, &cr, &cj);
	return fast_atan2_restated(cj, cr);
}

#define ORC_LUT_SIZE 131072 /* atan_lut_size, src/rtl_fm.c:105 */
#define ORC_LUT_COEF 8      /* atan_lut_coef, src/rtl_fm.c:106 */
static int32_t *g_lut;
static pthread_once_t g_lut_once = PTHREAD_ONCE_INIT;

static void build_lut(void)
{
	/* atan_lut_init(), src/rtl_fm.c:881-892 */
	g_lut = (int32_t *)malloc(sizeof(int32_t) * ORC_LUT_SIZE);
	for (int i = 0; i < ORC_LUT_SIZE; i++)
		g_lut[i] = (int32_t)(atan((double)i / (1 << ORC_LUT_COEF)) / 3.14159 * (1 << 14));
}

const int32_t *orc_atan_lut(void)
{
	pthread_once(&g_lut_once, build_lut);
	return g_lut;
}

int orc_polar_disc_lut(int ar, int aj, int br, int bj)
{
	/* src/rtl_fm.c:894-930, including the x == 0 fall-through into the
	 * final else (tiny angles with neither product zero return pi or 0). */
	const int32_t *lut = orc_atan_lut();
	int32_t cr, cj;
	conj_product(ar, aj, br, bj, &cr, &cj);
	if (cr == 0 || cj == 0) {
		if (cr == 0 && cj == 0) return 0;
		if (cr == 0) return cj > 0 ? (1 << 13) : -(1 << 13);
		return cr > 0 ? 0 : (1 << 14);
	}
	int32_t scaled = (int32_t)((uint32_t)cj << ORC_LUT_COEF);
	int32_t x;
	if (scaled == INT32_MIN && cr == -1)
		x = INT32_MIN; /* idiv would trap; unreachable for int16 inputs */
	else
		x = scaled / cr;
	int64_t mag = x < 0 ? -(int64_t)x : (int64_t)x;
	if (mag >= ORC_LUT_SIZE)
		return cj > 0 ? (1 << 13) : -(1 << 13);
	if (x > 0)
		return cj > 0 ? lut[x] : lut[x] - (1 << 14);
	return cj > 0 ? (1 << 14) - lut[-x] : -lut[-x];
}

int orc_fm_demod(const int16_t *lp, int lp_len, int16_t *result, int custom_atan,
                 int32_t *pre_r, int32_t *pre_j)
{
	/*
	 * src/rtl_fm.c:932-959.  Output 0 pairs the block's first sample with
	 * the carried (pre_r, pre_j) and ALWAYS uses polar_discriminant, whatever
	 * -A selected; outputs k >= 1 pair sample k with k-1 using the selected
	 * variant.
	 */
	result[0] = (int16_t)orc_polar_discriminant(lp[0], lp[1], *pre_r, *pre_j);
	for (int k = 1; 2 * k < lp_len - 1; k++) {
		int v = 0;
		const int16_t *cur = lp + 2 * k, *prv = lp + 2 * k - 2;
		if (custom_atan == RTLFM_ATAN_STD)
			v = orc_polar_discriminant(cur[0], cur[1], prv[0], prv[1]);
		else if (custom_atan == RTLFM_ATAN_FAST)
			v = orc_polar_disc_fast(cur[0], cur[1], prv[0], prv[1]);
		else if (custom_atan == RTLFM_ATAN_LUT)
			v = orc_polar_disc_lut(cur[0], cur[1], prv[0], prv[1]);
		result[k] = (int16_t)v;
	}
	*pre_r = lp[lp_len - 2];
	*pre_j = lp[lp_len - 1];
	return lp_len / 2;
}

int orc_am_demod(const int16_t *lp, int lp_len, int16_t *result, int output_scale)
{
	/* src/rtl_fm.c:961-976: (int16_t)sqrt(I^2+Q^2) * output_scale */
	for (int k = 0; 2 * k < lp_len; k++) {
		int p = lp[2 * k] * lp[2 * k] + lp[2 * k + 1] * lp[2 * k + 1];
		result[k] = (int16_t)wrap_mul((int16_t)sqrt((double)p), output_scale);
	}
	return lp_len / 2;
}

int orc_usb_demod(const int16_t *lp, int lp_len, int16_t *result, int output_scale)
{
	/* src/rtl_fm.c:978-988 */
	for (int k = 0; 2 * k < lp_len; k++) {
		int p = lp[2 * k] + lp[2 * k + 1];
		result[k] = (int16_t)wrap_mul((int16_t)p, output_scale);
	}
	return lp_len / 2;
}

int orc_lsb_demod(const int16_t *lp, int lp_len, int16_t *result, int output_scale)
{
	/* src/rtl_fm.c:990-1000 */
	for (int k = 0; 2 * k < lp_len; k++) {
		int p = lp[2 * k] - lp[2 * k + 1];
		result[k] = (int16_t)wrap_mul((int16_t)p, output_scale);
	}
	return lp_len / 2;
}

int orc_raw_demod(const int16_t *lp, int lp_len, int16_t *result)
{
	/* src/rtl_fm.c:1002-1009 */
	memcpy(result, lp, sizeof(int16_t) * (size_t)lp_len);
	return lp_len;
}

int orc_low_pass_simple(int16_t *sig, int len, int step)
{
	/* src/rtl_fm.c:739-753: sums of `step` audio samples, no divide.  The
	 * reference also duplicates one sample past the end (:751); that write is
	 * outside the returned length and is not part of the result. */
	int out = 0;
	for (int base = 0; base < len; base += step, out++) {
		int acc = 0;
		for (int k = 0; k < step; k++)
			acc += sig[base + k];
		sig[out] = (int16_t)acc;
	}
	return len / step;
}

void orc_deemph(int16_t *result, int len, int a, int32_t *avg)
{
	/* src/rtl_fm.c:1011-1026: avg += round-half-away((x - avg) / a) using C
	 * truncating division; `avg` is function-static there (one stream). */
	int32_t v = *avg;
	for (int k = 0; k < len; k++) {
		int d = result[k] - v;
		v += d > 0 ? (d + a / 2) / a : (d - a / 2) / a;
		result[k] = (int16_t)v;
	}
	*avg = v;
}

void orc_dc_block_audio(int16_t *result, int len, int k, int32_t *dc_avg)
{
	/* src/rtl_fm.c:1028-1041 */
	if (len <= 0)
		return;
	int64_t acc = 0;
	for (int n = 0; n < len; n++)
		acc += result[n];
	int mean = (int)(acc / len);
	mean = (mean + *dc_avg * k) / (k + 1);
	for (int n = 0; n < len; n++)
		result[n] = (int16_t)(result[n] - mean);
	*dc_avg = mean;
}

int orc_low_pass_real(int16_t *result, int len, int fast, int slow,
                      int32_t *now_lpr, int32_t *prev_lpr_index)
{
	/* src/rtl_fm.c:755-775: accumulate; every time the phase accumulator
	 * (+= slow per input) reaches `fast`, emit sum / (fast/slow) — the
	 * integer quotient of the rates.  The reference divides by zero when
	 * slow > fast; that is reported as -1 here. */
	if (slow <= 0 || fast / slow == 0)
		return -1;
	int div = fast / slow;
	int out = 0;
	for (int n = 0; n < len; n++) {
		*now_lpr = wrap_add(*now_lpr, result[n]);
		*prev_lpr_index += slow;
		if (*prev_lpr_index >= fast) {
			result[out++] = (int16_t)(*now_lpr / div);
			*prev_lpr_index -= fast;
			*now_lpr = 0;
		}
	}
	return out;
}

void orc_arbitrary_upsample(const int16_t *b1, int16_t *b2, int len1, int len2)
{
	/* src/rtl_fm.c:1114-1135: linear interpolation in double, truncated */
	int src = 1, tick = 0;
	for (int j = 0; j < len2; j++) {
		double frac = (double)tick / (double)len2;
		b2[j] = (int16_t)(b1[src - 1] * (1 - frac) + b1[src] * frac);
		tick += len1;
		if (tick > len2) {
			tick -= len2;
			src++;
		}
		if (src >= len1) {
			src = len1 - 1;
			tick = len2;
		}
	}
}

void orc_arbitrary_downsample(const int16_t *b1, int16_t *b2, int len1, int len2)
{
	/* src/rtl_fm.c:1137-1166: fractional boxcar; starts at b1[1]; the int16
	 * accumulator wraps; b2[len2] is written (one past the end) exactly as the
	 * reference does, so b2 needs len2+1 elements. */
	int src = 1, j = 0, tick = 0;
	double carry = 0;
	b2[0] = 0;
	while (j < len2) {
		double frac = 1.0;
		if (tick + len2 > len1)
			frac = (double)(len1 - tick) / (double)len2;
		b2[j] = (int16_t)(b2[j] + (int16_t)((double)b1[src] * frac + carry));
		carry = (double)b1[src] * (1.0 - frac);
		tick += len2;
		src++;
		if (tick > len1) {
			j++;
			b2[j] = 0;
			tick -= len1;
		}
		if (src >= len1) {
			src = len1 - 1;
			tick = len1;
		}
	}
	for (j = 0; j < len2; j++)
		b2[j] = (int16_t)(b2[j] * len2 / len1);
}

int orc_rms(const int16_t *samples, int len, int step, int omit_dc_fix)
{
	/* src/rtl_fm.c:1083-1112: uint32 sum of squares, int32 sum, the DC
	 * correction in double */
	uint32_t p = 0;
	int32_t t = 0;
	while (len > step * 32768)
		++step;
	for (int i = 0; i < len; i += step) {
		int32_t s = samples[i];
		t = wrap_add(t, s);
		p += (uint32_t)(s * s);
	}
	if (omit_dc_fix) {
		int num = len / step;
		return (int)sqrt((double)p / num);
	}
	double dc = (double)wrap_mul(t, step) / (double)len;
	double err = t * 2 * dc - dc * dc * len;
	return (int)sqrt((p - err) / len);
}

/* ---- planner -------------------------------------------------------------- */

void orc_optimal_settings(rtlfm_cfg *cfg, uint32_t freq, int rate_in,
                          int min_capture_rate, int use_fifth_order, int edge,
                          uint32_t *capture_freq, uint32_t *capture_rate)
{
	/* src/rtl_fm.c:1407-1445 */
	cfg->downsample = min_capture_rate / rate_in + 1;
	cfg->downsample_passes = 0;
	if (use_fifth_order) {
		cfg->downsample_passes = (int)log2(cfg->downsample) + 1;
		cfg->downsample = 1 << cfg->downsample_passes;
	}
	uint32_t rate = (uint32_t)cfg->downsample * (uint32_t)rate_in;
	uint32_t f = freq;
	if (!cfg->offset_tuning)
		f = freq - rate / 4;
	f += (uint32_t)(edge * rate_in / 2);
	cfg->output_scale = (1 << 15) / (128 * cfg->downsample);
	if (cfg->output_scale < 1)
		cfg->output_scale = 1;
	if (cfg->mode == RTLFM_MODE_FM)
		cfg->output_scale = 1;
	if (capture_freq) *capture_freq = f;
	if (capture_rate) *capture_rate = rate;
}

int orc_deemph_a(int rate_out, int tc_us)
{
	/* src/rtl_fm.c:1929-1931 */
	double tc = (double)tc_us * 1e-6;
	return (int)round(1.0 / (1.0 - exp(-1.0 / (rate_out * tc))));
}

/* ---- whole chain ------------------------------------------------------------ */

int orc_result_cap(const rtlfm_cfg *cfg)
{
	int lp = (int)cfg->block_len;
	if (cfg->downsample_passes > 0)
		lp >>= cfg->downsample_passes;
	else if (cfg->downsample > 1)
		lp = 2 * (lp / 2 / cfg->downsample + 1);
	int n = cfg->mode == RTLFM_MODE_RAW ? lp : lp / 2;
	if (cfg->mode != RTLFM_MODE_RAW && cfg->rate_out2 > 0 &&
	    cfg->resampler == RTLFM_RESAMPLE_ARBITRARY && cfg->rate_out > 0) {
		int64_t up = (int64_t)n * cfg->rate_out2 / cfg->rate_out + 2;
		if (up > n) n = (int)up;
	}
	return n + 2;
}

int orc_block(const rtlfm_cfg *cfg, rtlfm_stream_state *st, const uint8_t *iq,
              uint32_t len, int16_t *out)
{
	static __thread int16_t *lowpassed, *result;
	if (!lowpassed) {
		lowpassed = (int16_t *)malloc(sizeof(int16_t) * (RTLFM_MAX_BLOCK_LEN + 16));
		result = (int16_t *)malloc(sizeof(int16_t) * (2 * RTLFM_MAX_BLOCK_LEN + 16));
	}
	if (len > RTLFM_MAX_BLOCK_LEN || len < 8 || (len & 7))
		return -1;

	/* --- rtlsdr_callback, src/rtl_fm.c:1326-1342 --- */
	orc_u8_to_i16(iq, lowpassed, (int)len);
	if (cfg->dc_block_raw)
		orc_dc_block_raw(lowpassed, (int)len, cfg->rdc_block_const, &st->dc_avgI, &st->dc_avgQ);
	if (!cfg->offset_tuning)
		orc_rotate16_neg90(lowpassed, (int)len);
	int lp_len = (int)len;

	/* --- full_demod, src/rtl_fm.c:1179-1272 --- */
	int passes = cfg->downsample_passes;
	if (passes) {
		for (int p = 0; p < passes; p++) {
			orc_fifth_order(lowpassed, lp_len >> p, st->lp_i_hist[p]);
			orc_fifth_order(lowpassed + 1, (lp_len >> p) - 1, st->lp_q_hist[p]);
		}
		lp_len >>= passes;
		if (cfg->comp_fir_size == 9 && passes <= 10) {
			orc_generic_fir(lowpassed, lp_len, passes, st->droop_i_hist);
			orc_generic_fir(lowpassed + 1, lp_len - 1, passes, st->droop_q_hist);
		}
	} else {
		/* a buffer shorter than the boxcar can leave lp_len == 0; fm_demod() then reads
		 * lowpassed[-2] (src/rtl_fm.c:955-956): outside the reference's domain */
		if (cfg->downsample > lp_len / 2) return -4;
		lp_len = orc_low_pass(lowpassed, lp_len, cfg->downsample, &st->now_r,
		                      &st->now_j, &st->prev_index);
	}
	if (cfg->squelch_level) {
		/* src/rtl_fm.c:1204-1215 */
		int sr = orc_rms(lowpassed, lp_len, 1, cfg->dc_block_raw);
		if (sr >= 0) {
			if (sr < cfg->squelch_level) {
				st->squelch_hits++;
				memset(lowpassed, 0, sizeof(int16_t) * (size_t)lp_len);
			} else {
				st->squelch_hits = 0;
			}
		}
	}
	int n;
	switch (cfg->mode) {
	case RTLFM_MODE_FM:
		if (lp_len < 2) return 0; /* boxcar produced nothing this block */
		n = orc_fm_demod(lowpassed, lp_len, result, cfg->custom_atan, &st->pre_r, &st->pre_j);
		break;
	case RTLFM_MODE_AM: n = orc_am_demod(lowpassed, lp_len, result, cfg->output_scale); break;
	case RTLFM_MODE_USB: n = orc_usb_demod(lowpassed, lp_len, result, cfg->output_scale); break;
	case RTLFM_MODE_LSB: n = orc_lsb_demod(lowpassed, lp_len, result, cfg->output_scale); break;
	case RTLFM_MODE_RAW:
		n = orc_raw_demod(lowpassed, lp_len, result);
		memcpy(out, result, sizeof(int16_t) * (size_t)n);
		return n; /* src/rtl_fm.c:1257-1259 */
	default: return -1;
	}
	if (cfg->post_downsample > 1) {
		/* "no wrap around, length must be multiple of step" (src/rtl_fm.c:740): otherwise the
		 * reference's last sum reads stale samples past result_len — outside its domain */
		if (n % cfg->post_downsample) return -3;
		n = orc_low_pass_simple(result, n, cfg->post_downsample);
	}
	if (cfg->deemph)
		orc_deemph(result, n, cfg->deemph_a, &st->deemph_avg);
	if (cfg->dc_block_audio)
		orc_dc_block_audio(result, n, cfg->adc_block_const, &st->dc_avg);
	if (cfg->rate_out2 > 0) {
		if (cfg->resampler == RTLFM_RESAMPLE_ARBITRARY) {
			/* the commented-out call, src/rtl_fm.c:1270 */
			int len2 = (int)((int64_t)n * cfg->rate_out2 / cfg->rate_out);
			if (n < len2) {
				orc_arbitrary_upsample(result, out, n, len2);
			} else {
				static __thread int16_t *tmp;
				if (!tmp) tmp = (int16_t *)malloc(sizeof(int16_t) * (RTLFM_MAX_BLOCK_LEN + 16));
				orc_arbitrary_downsample(result, tmp, n, len2);
				memcpy(out, tmp, sizeof(int16_t) * (size_t)len2);
			}
			return len2;
		}
		n = orc_low_pass_real(result, n, cfg->rate_out, cfg->rate_out2,
		                      &st->now_lpr, &st->prev_lpr_index);
		if (n < 0) return -2;
	}
	memcpy(out, result, sizeof(int16_t) * (size_t)n);
	return n;
}

struct batch_job {
	const rtlfm_cfg *cfg;
	rtlfm_stream_state *st;
	const uint8_t *iq;
	size_t stream_stride;
	int nblocks;
	int16_t *out;
	size_t out_stride;
	int32_t *out_len;
	int s0, s1;
	int status;
};

static void *batch_worker(void *arg)
{
	struct batch_job *job = (struct batch_job *)arg;
	int cap = orc_result_cap(job->cfg);
	int16_t *scratch = (int16_t *)malloc(sizeof(int16_t) * (size_t)cap);
	for (int s = job->s0; s < job->s1; s++) {
		int total = 0;
		for (int b = 0; b < job->nblocks; b++) {
			const uint8_t *src = job->iq + (size_t)s * job->stream_stride +
			                     (size_t)b * job->cfg->block_len;
			int n = orc_block(job->cfg, &job->st[s], src, job->cfg->block_len, scratch);
			if (n < 0) { job->status = n; free(scratch); return NULL; }
			if (job->out)
				memcpy(job->out + (size_t)s * job->out_stride + total, scratch,
				       sizeof(int16_t) * (size_t)n);
			total += n;
		}
		if (job->out_len) job->out_len[s] = total;
	}
	free(scratch);
	return NULL;
}

int orc_run_batch(const rtlfm_cfg *cfg, rtlfm_stream_state *st, int nstreams,
                  const uint8_t *iq, size_t stream_stride, int nblocks,
                  int16_t *out, size_t out_stride, int32_t *out_len,
                  int nthreads)
{
	if (nthreads < 1) nthreads = 1;
	if (nthreads > nstreams) nthreads = nstreams;
	struct batch_job *jobs = (struct batch_job *)calloc((size_t)nthreads, sizeof(*jobs));
	pthread_t *tid = (pthread_t *)calloc((size_t)nthreads, sizeof(*tid));
	int per = (nstreams + nthreads - 1) / nthreads;
	for (int t = 0; t < nthreads; t++) {
		struct batch_job *j = &jobs[t];
		j->cfg = cfg; j->st = st; j->iq = iq; j->stream_stride = stream_stride;
		j->nblocks = nblocks; j->out = out; j->out_stride = out_stride;
		j->out_len = out_len;
		j->s0 = t * per;
		j->s1 = (t + 1) * per < nstreams ? (t + 1) * per : nstreams;
		if (j->s0 > j->s1) j->s0 = j->s1;
		if (nthreads == 1)
			batch_worker(j);
		else
			pthread_create(&tid[t], NULL, batch_worker, j);
	}
	int status = 0;
	for (int t = 0; t < nthreads; t++) {
		if (nthreads > 1) pthread_join(tid[t], NULL);
		if (jobs[t].status < 0 && status == 0) status = jobs[t].status;
	}
	free(jobs);
	free(tid);
	return status;
}
