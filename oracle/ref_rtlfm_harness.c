/*
 * ref_rtlfm_harness.c — builds the REAL reference hot path into oracle/_ref/.
 *
 * TEST INFRASTRUCTURE ONLY (generator and checker).  This file contains no
 * reference text: it textually includes the reference translation unit from
 * where it lies (REF_RTL_FM_C, normally /root/reference/src/rtl_fm.c) so that
 * its static functions (rtlsdr_callback, optimal_settings) are reachable, and
 * adds thin single-threaded drivers around them.
 *
 * No stand-ins are written for librtlsdr / libusb: the object is linked as a
 * shared library whose rtlsdr_* function references stay unresolved, and it
 * is opened with RTLD_LAZY by oracle/ref_loader.c.  The DSP path never calls
 * them, so they are never bound.
 */
#include <stdint.h>
#include <string.h>

#include "../include/rtlfm_hip.h"

#define main rtl_fm_reference_main
#include REF_RTL_FM_C
#undef main

static unsigned char ref_inbuf[MAXIMUM_BUF_LENGTH];

/* Fresh demod_state/dongle_state as main() sets them up (src/rtl_fm.c:1715-1719).
 * deemph_filter's function-static avg cannot be reached from here: reload the
 * library (ref_loader's close/open) to reset it. */
void ref_reset(void)
{
	static int first = 1;
	if (!first) {
		demod_cleanup(&demod);
		output_cleanup(&output);
		controller_cleanup(&controller);
	}
	first = 0;
	memset(&demod, 0, sizeof(demod));
	memset(&dongle, 0, sizeof(dongle));
	memset(&output, 0, sizeof(output));
	memset(&controller, 0, sizeof(controller));
	memset(&cmd, 0, sizeof(cmd));
	dongle_init(&dongle);
	demod_init(&demod);
	output_init(&output);
	controller_init(&controller);
	cmd_init(&cmd);
	do_exit = 0;
}

static int ref_resampler;

/* Drive the demod_state fields a command line would set. */
int ref_configure(const rtlfm_cfg *cfg)
{
	switch (cfg->mode) {
	case RTLFM_MODE_FM: demod.mode_demod = &fm_demod; break;
	case RTLFM_MODE_AM: demod.mode_demod = &am_demod; break;
	case RTLFM_MODE_USB: demod.mode_demod = &usb_demod; break;
	case RTLFM_MODE_LSB: demod.mode_demod = &lsb_demod; break;
	case RTLFM_MODE_RAW: demod.mode_demod = &raw_demod; break;
	default: return -1;
	}
	demod.downsample = cfg->downsample;
	demod.downsample_passes = cfg->downsample_passes;
	demod.comp_fir_size = cfg->comp_fir_size;
	demod.custom_atan = cfg->custom_atan;
	if (cfg->custom_atan == RTLFM_ATAN_LUT && !atan_lut)
		atan_lut_init();
	demod.post_downsample = cfg->post_downsample;
	demod.deemph = cfg->deemph;
	demod.deemph_a = cfg->deemph_a;
	demod.rate_out = cfg->rate_out;
	demod.rate_out2 = cfg->rate_out2;
	ref_resampler = cfg->resampler;
	demod.dc_block_audio = cfg->dc_block_audio;
	demod.adc_block_const = cfg->adc_block_const;
	demod.dc_block_raw = cfg->dc_block_raw;
	demod.rdc_block_const = cfg->rdc_block_const;
	dongle.offset_tuning = cfg->offset_tuning;
	demod.output_scale = cfg->output_scale;
	demod.squelch_level = cfg->squelch_level;
	dongle.buf_len = cfg->block_len;
	return 0;
}

/* One async-callback buffer through the reference's own rtlsdr_callback()
 * and full_demod(); returns result_len and copies demod.result. */
int ref_block(const uint8_t *iq, uint32_t len, int16_t *out)
{
	int arbitrary = demod.rate_out2 > 0 && ref_resampler == RTLFM_RESAMPLE_ARBITRARY &&
	                demod.mode_demod != &raw_demod;
	int saved = demod.rate_out2;
	if (len > MAXIMUM_BUF_LENGTH)
		return -1;
	memcpy(ref_inbuf, iq, len);
	rtlsdr_callback(ref_inbuf, len, &dongle);
	if (arbitrary)
		demod.rate_out2 = -1;
	full_demod(&demod);
	demod.rate_out2 = saved;
	if (arbitrary) {
		/* the call the reference keeps commented out at src/rtl_fm.c:1270,
		 * made out of place */
		static int16_t tmp[MAXIMUM_BUF_LENGTH + 8];
		int len2 = (int)((long long)demod.result_len * demod.rate_out2 / demod.rate_out);
		arbitrary_resample(demod.result, tmp, demod.result_len, len2);
		memcpy(out, tmp, 2 * (size_t)len2);
		return len2;
	}
	memcpy(out, demod.result, 2 * (size_t)demod.result_len);
	return demod.result_len;
}

/* nblocks consecutive buffers of one stream; returns total output samples.
 * out may be NULL (timing runs). */
int ref_run_stream(const uint8_t *iq, uint32_t block_len, int nblocks, int16_t *out)
{
	static int16_t scratch[2 * MAXIMUM_BUF_LENGTH + 8];
	int total = 0;
	for (int b = 0; b < nblocks; b++) {
		int n = ref_block(iq + (size_t)b * block_len, block_len, scratch);
		if (n < 0)
			return n;
		if (out)
			memcpy(out + total, scratch, 2 * (size_t)n);
		total += n;
	}
	return total;
}

/* Carried state as the public struct (deemph avg is not reachable: left 0). */
void ref_state_get(rtlfm_stream_state *st)
{
	memset(st, 0, sizeof(*st));
	memcpy(st->lp_i_hist, demod.lp_i_hist, sizeof(st->lp_i_hist));
	memcpy(st->lp_q_hist, demod.lp_q_hist, sizeof(st->lp_q_hist));
	memcpy(st->droop_i_hist, demod.droop_i_hist, sizeof(st->droop_i_hist));
	memcpy(st->droop_q_hist, demod.droop_q_hist, sizeof(st->droop_q_hist));
	st->now_r = demod.now_r; st->now_j = demod.now_j; st->prev_index = demod.prev_index;
	st->pre_r = demod.pre_r; st->pre_j = demod.pre_j;
	st->now_lpr = demod.now_lpr; st->prev_lpr_index = demod.prev_lpr_index;
	st->dc_avg = demod.dc_avg; st->dc_avgI = demod.dc_avgI; st->dc_avgQ = demod.dc_avgQ;
	st->squelch_hits = demod.squelch_hits;
}

/* The reference's static optimal_settings() (src/rtl_fm.c:1407-1445). */
void ref_optimal_settings(uint32_t freq, int rate_in, int min_capture_rate,
                          int use_fifth_order, int edge, int mode, int offset_tuning,
                          int32_t out[6])
{
	ref_reset();
	MinCaptureRate = min_capture_rate;
	demod.rate_in = rate_in;
	demod.downsample_passes = use_fifth_order;
	controller.edge = edge;
	dongle.offset_tuning = offset_tuning;
	switch (mode) {
	case RTLFM_MODE_AM: demod.mode_demod = &am_demod; break;
	case RTLFM_MODE_USB: demod.mode_demod = &usb_demod; break;
	case RTLFM_MODE_LSB: demod.mode_demod = &lsb_demod; break;
	case RTLFM_MODE_RAW: demod.mode_demod = &raw_demod; break;
	default: demod.mode_demod = &fm_demod; break;
	}
	optimal_settings(freq, (uint32_t)rate_in);
	out[0] = demod.downsample;
	out[1] = demod.downsample_passes;
	out[2] = demod.output_scale;
	out[3] = (int32_t)dongle.freq;
	out[4] = (int32_t)dongle.rate;
	out[5] = 0;
}

const int *ref_atan_lut(void)
{
	if (!atan_lut)
		atan_lut_init();
	return atan_lut;
}

/* the reference's u8 rotate_90 (src/rtl_fm.c:437-447, dead code there), in place */
void ref_rotate_90_u8(unsigned char *buf, uint32_t len)
{
	rotate_90(buf, len);
}
