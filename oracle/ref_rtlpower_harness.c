/*
 * ref_rtlpower_harness.c — builds the REAL rtl_power DSP into oracle/_ref/.
 * TEST INFRASTRUCTURE ONLY.  Contains no reference text: it textually includes
 * the reference translation unit from where it lies (REF_RTL_POWER_C).
 *
 * scanner() (src/rtl_power.c:642-720) cannot be called: its first statements go
 * into librtlsdr (rtlsdr_get_center_freq / rtlsdr_read_sync), which this image
 * cannot build and for which no stand-ins are written.  ref_power_scan() below
 * therefore walks the same sequence calling the REFERENCE'S OWN functions
 * (rms_power, downsample_iq/fifth_order, generic_fir, remove_dc, fix_fft,
 * real_conj, sine_table, the window functions) and restates only the inline
 * glue between them, each piece marked with the lines it stands for.
 */
#include <stdint.h>
#include <string.h>

#include "../include/rtlpower_hip.h"

#define main rtl_power_reference_main
#include REF_RTL_POWER_C
#undef main

static int ref_bin_e = -1, ref_buf_len;

int ref_power_setup(const rtlpower_cfg *c)
{
	double (*wf)(int, int) = rectangle;
	switch (c->window) {
	case RTLPOWER_WIN_HAMMING: wf = hamming; break;
	case RTLPOWER_WIN_BLACKMAN: wf = blackman; break;
	case RTLPOWER_WIN_BLACKMAN_HARRIS: wf = blackman_harris; break;
	case RTLPOWER_WIN_HANN_POISSON: wf = hann_poisson; break;
	case RTLPOWER_WIN_YOUSSEF: wf = youssef; break;
	case RTLPOWER_WIN_KAISER: wf = kaiser; break;
	case RTLPOWER_WIN_BARTLETT: wf = bartlett; break;
	default: wf = rectangle; break;
	}
	struct tuning_state *ts = &tunes[0];
	int bins = 1 << c->bin_e;
	tune_count = 1;
	free(ts->avg); free(ts->buf8);
	memset(ts, 0, sizeof(*ts));
	ts->bin_e = c->bin_e;
	ts->downsample = c->downsample;
	ts->downsample_passes = c->downsample_passes;
	ts->buf_len = (int)c->buf_len;
	ts->avg = (long *)calloc((size_t)bins, sizeof(long));
	ts->buf8 = (uint8_t *)malloc(c->buf_len);
	boxcar = c->boxcar;
	comp_fir_size = c->comp_fir_size;
	peak_hold = c->peak_hold;
	/* main(), src/rtl_power.c:979-988 */
	free(Sinewave); free(power_table); free(fft_buf); free(window_coefs);
	sine_table(ts->bin_e);
	fft_buf = (int16_t *)malloc(c->buf_len * sizeof(int16_t) + 64);
	window_coefs = (int *)malloc((size_t)bins * sizeof(int));
	for (int i = 0; i < bins; i++)
		window_coefs[i] = (int)(256 * wf(i, bins));
	ref_bin_e = c->bin_e;
	ref_buf_len = (int)c->buf_len;
	return 0;
}

/* One read of tunes[0], following scanner() :657-718 */
int ref_power_scan(const uint8_t *buf8)
{
	struct tuning_state *ts = &tunes[0];
	int j, j2, offset, ds, ds_p;
	int bin_e = ts->bin_e, bin_len = 1 << bin_e, buf_len = ts->buf_len;
	int32_t w;
	memcpy(ts->buf8, buf8, (size_t)buf_len);        /* stands for rtlsdr_read_sync, :657 */
	if (bin_len == 1) {
		rms_power(ts);                              /* reference function */
		return 0;
	}
	for (j = 0; j < buf_len; j++)                   /* glue :666-668 */
		fft_buf[j] = (int16_t)ts->buf8[j] - 127;
	ds = ts->downsample;
	ds_p = ts->downsample_passes;
	if (boxcar && ds > 1) {                         /* glue :671-681 */
		j = 2, j2 = 0;
		while (j < buf_len) {
			fft_buf[j2] += fft_buf[j];
			fft_buf[j2 + 1] += fft_buf[j + 1];
			fft_buf[j] = 0;
			fft_buf[j + 1] = 0;
			j += 2;
			if (j % (ds * 2) == 0) j2 += 2;
		}
	} else if (ds_p) {
		for (j = 0; j < ds_p; j++)
			downsample_iq(fft_buf, buf_len >> j);   /* reference function */
		if (comp_fir_size == 9 && ds_p <= CIC_TABLE_MAX) {
			generic_fir(fft_buf, buf_len >> j, cic_9_tables[ds_p]);        /* reference function */
			generic_fir(fft_buf + 1, (buf_len >> j) - 1, cic_9_tables[ds_p]);
		}
	}
	remove_dc(fft_buf, buf_len / ds);               /* reference function */
	remove_dc(fft_buf + 1, (buf_len / ds) - 1);
	for (offset = 0; offset < (buf_len / ds); offset += (2 * bin_len)) {
		for (j = 0; j < bin_len; j++) {             /* glue :697-706 */
			w = (int32_t)fft_buf[offset + j * 2];
			w *= (int32_t)(window_coefs[j]);
			fft_buf[offset + j * 2] = (int16_t)w;
			w = (int32_t)fft_buf[offset + j * 2 + 1];
			w *= (int32_t)(window_coefs[j]);
			fft_buf[offset + j * 2 + 1] = (int16_t)w;
		}
		fix_fft(fft_buf + offset, bin_e);           /* reference function */
		for (j = 0; j < bin_len; j++) {             /* glue :708-716 over real_conj() */
			long p = real_conj(fft_buf[offset + j * 2], fft_buf[offset + j * 2 + 1]);
			if (!peak_hold) ts->avg[j] += p;
			else ts->avg[j] = MAX(p, ts->avg[j]);
		}
		ts->samples += ds;                          /* :717 */
	}
	return 0;
}

int ref_power_get(int64_t *avg, int32_t *samples)
{
	int bins = 1 << tunes[0].bin_e;
	for (int i = 0; i < bins; i++) avg[i] = (int64_t)tunes[0].avg[i];
	*samples = tunes[0].samples;
	return bins;
}

void ref_power_clear(void)
{
	int bins = 1 << tunes[0].bin_e;
	memset(tunes[0].avg, 0, sizeof(long) * (size_t)bins);
	tunes[0].samples = 0;
}

const int *ref_window_coefs(void) { return window_coefs; }
const int16_t *ref_sinewave(void) { return Sinewave; }

/* frequency_range() (src/rtl_power.c:438-540) through the reference's own code:
 * out = {tune_count, bw_seen (from freq spacing), rate, bin_e, downsample, passes, buf_len, freq0, freq1} */
int ref_frequency_range(const char *arg, double crop_in, int boxcar_flag, int32_t out[10], double *crop_out)
{
	char tmp[256];
	strncpy(tmp, arg, sizeof(tmp) - 1); tmp[sizeof(tmp) - 1] = 0;
	for (int i = 0; i < tune_count && i < MAX_TUNES; i++) { free(tunes[i].avg); free(tunes[i].buf8); }
	memset(tunes, 0, sizeof(tunes));
	tune_count = 0;
	boxcar = boxcar_flag;
	frequency_range(tmp, crop_in);
	out[0] = tune_count; out[1] = tune_count > 1 ? tunes[1].freq - tunes[0].freq : 0;
	out[2] = tunes[0].rate; out[3] = tunes[0].bin_e; out[4] = tunes[0].downsample;
	out[5] = tunes[0].downsample_passes; out[6] = tunes[0].buf_len; out[7] = tunes[0].freq;
	out[8] = tune_count > 1 ? tunes[tune_count - 1].freq : tunes[0].freq;
	*crop_out = tunes[0].crop;
	return 0;
}

/* csv_dbm() (src/rtl_power.c:722-765) on tunes[0] as frequency_range left it, with the given
 * accumulators; the line is captured from the reference's FILE* `file`. */
int ref_csv_dbm(int tune, const int64_t *avg, int32_t samples, char *out, size_t cap)
{
	struct tuning_state *ts = &tunes[tune];
	int bins = 1 << ts->bin_e;
	char *mem = NULL; size_t sz = 0;
	for (int i = 0; i < bins; i++) ts->avg[i] = (long)avg[i];
	ts->samples = samples;
	file = open_memstream(&mem, &sz);
	csv_dbm(ts);
	fclose(file);
	file = NULL;
	if (sz + 1 > cap) { free(mem); return -1; }
	memcpy(out, mem, sz + 1);
	free(mem);
	return (int)sz;
}
