/*
 * ref_rtlpower_harness.c — builds the REAL rtl_power DSP into oracle/_ref/.
 * TEST INFRASTRUCTURE ONLY.  Contains no reference text: it textually includes
 * the reference translation unit from where it lies (REF_RTL_POWER_C) and drives
 * its own functions — scanner() (src/rtl_power.c:642-720) included.
 *
 * scanner() goes into librtlsdr first (rtlsdr_get_center_freq / rtlsdr_read_sync),
 * which this image cannot build; no stand-ins are written for it here.  The object
 * keeps those references undefined (oracle/ref_loader.c opens it RTLD_LAZY) and the
 * tests load the PRODUCT's file-backed device layer, librtlsdr_file.so (the §8b row
 * of SURVEY.md, built from rtlsdr_amd/csrc/host/rtlsdr_file.c), into the process
 * first: scanner() then reads its buffers from a file through the same 26-symbol
 * API the tools link against.
 */
#include <stdlib.h>
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

/* nreads calls of the reference's own scanner() (src/rtl_power.c:642-720).  scanner() begins
 * with rtlsdr_get_center_freq() / rtlsdr_read_sync(): they bind (lazily) to whatever device layer
 * the process has loaded — in the tests the product's file-backed librtlsdr_file.so
 * (SURVEY.md §8b), opened RTLD_GLOBAL before this object, reading the bytes from iq_path.  Nothing
 * of scanner() is restated here. */
int ref_power_scan_file(const char *iq_path, int nreads)
{
	struct tuning_state *ts = &tunes[0];
	setenv("RTLSDR_FILE", iq_path, 1);
	if (rtlsdr_open(&dev, 0) < 0 || !dev)
		return -1;
	/* the device already sits on the hop's frequency: retune() (:542-552) would drop 4096 bytes */
	rtlsdr_set_center_freq(dev, (uint32_t)ts->freq);
	for (int r = 0; r < nreads; r++)
		scanner();
	rtlsdr_close(dev);
	dev = NULL;
	return 0;
}

/* The same for the CPU baseline's worker threads (oracle/ref_loader.c: ref_power_bench_mt, one
 * private copy of this object per thread): RTLSDR_FILE / RTLSDR_FILE_LOOP are set ONCE by the caller
 * before the threads start (setenv is not thread-safe), every thread opens its own device on that
 * file and calls scanner() `nscans` times. */
int ref_power_scan_env(int nscans)
{
	struct tuning_state *ts = &tunes[0];
	if (rtlsdr_open(&dev, 0) < 0 || !dev)
		return -1;
	rtlsdr_set_center_freq(dev, (uint32_t)ts->freq);
	for (int r = 0; r < nscans; r++)
		scanner();
	rtlsdr_close(dev);
	dev = NULL;
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
