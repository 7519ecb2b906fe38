/*
 * rtlpower_hip.h — C ABI of the MI355X (gfx950) form of rtl_power's FFT loop.
 *
 * Replaces the body of scanner() (reference src/rtl_power.c:642-720) from the
 * u8 -> int16 conversion after rtlsdr_read_sync() (:657, :666-668) to the
 * accumulation into tuning_state.avg[] / .samples (:708-717):
 *
 *   rms_power()            src/rtl_power.c:410-436   (bin_e == 0)
 *   u8 -> int16 (-127)     src/rtl_power.c:666-668
 *   boxcar | downsample_iq src/rtl_power.c:671-681 | 628-634 (fifth_order :554-579)
 *   generic_fir            src/rtl_power.c:598-626
 *   remove_dc              src/rtl_power.c:581-596
 *   window multiply        src/rtl_power.c:697-706 (window_coefs :983-988, windows :329-408)
 *   fix_fft                src/rtl_power.c:271-327 (sine_table :247-261, FIX_MPY :263-269)
 *   integrate / peak hold  src/rtl_power.c:708-716
 *
 * One "stream" here is one tuning_state (one hop of the frequency plan, or
 * one dongle); a "read" is one rtlsdr_read_sync() buffer of buf_len bytes.
 * Same conventions as rtlfm_hip.h: int return, 0 or negative errno.
 */
#ifndef RTLPOWER_HIP_H
#define RTLPOWER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* -w option, src/rtl_power.c:843-862 */
enum rtlpower_window {
	RTLPOWER_WIN_RECTANGLE = 0,
	RTLPOWER_WIN_HAMMING = 1,
	RTLPOWER_WIN_BLACKMAN = 2,
	RTLPOWER_WIN_BLACKMAN_HARRIS = 3,
	RTLPOWER_WIN_HANN_POISSON = 4,
	RTLPOWER_WIN_YOUSSEF = 5,
	RTLPOWER_WIN_KAISER = 6,   /* a 1.0 stub in the reference, :392-396 */
	RTLPOWER_WIN_BARTLETT = 7
};

/* The tuning_state / global fields scanner() reads (src/rtl_power.c:86-120). */
typedef struct rtlpower_cfg {
	int32_t bin_e;              /* log2 of the FFT length; 0 selects rms_power() */
	int32_t window;             /* enum rtlpower_window */
	int32_t downsample;         /* ts->downsample */
	int32_t downsample_passes;  /* ts->downsample_passes (used when boxcar == 0) */
	int32_t boxcar;             /* global boxcar, default 1 (:118) */
	int32_t comp_fir_size;      /* 0 or 9 (:119) */
	int32_t peak_hold;          /* :120 */
	uint32_t buf_len;           /* bytes per read; the last FFT frame of a read must end inside it
	                             * (frequency_range() uses 2 * 2^bin_e * downsample, at least 16384) */
} rtlpower_cfg;

/*
 * The hop plan frequency_range() builds into tunes[] (src/rtl_power.c:438-540):
 * tune i is centred on lower + i*bw_seen + bw_seen/2 and sampled at `rate`.
 */
typedef struct rtlpower_plan {
	int32_t lower, upper, max_size;  /* the three fields of -f lower:upper:bin_size (after atofs) */
	int32_t tune_count;
	int32_t bw_seen;                 /* bandwidth kept per hop */
	int32_t rate;                    /* bw_used: the dongle rate of every hop */
	int32_t bin_e;
	int32_t downsample, downsample_passes;
	int32_t buf_len;
	double crop;                     /* may be forced to 0 (giant bins) */
	double bin_size;                 /* reported "FFT bin size" */
} rtlpower_plan;

typedef struct rtlpower_gpu rtlpower_gpu;

/* frequency_range() (src/rtl_power.c:438-540); `boxcar` is the global of :118 (0 after -F).
 * Returns 0, or -E2BIG when the plan needs more than MAX_TUNES (3000) hops. */
int rtlpower_frequency_range(int32_t lower, int32_t upper, int32_t max_size, double crop, int boxcar,
                             rtlpower_plan *out);
/* centre frequency of hop i (src/rtl_power.c:509) */
int32_t rtlpower_tune_freq(const rtlpower_plan *plan, int i);
/* the rtlpower_cfg one hop of the plan scans with */
void rtlpower_plan_cfg(const rtlpower_plan *plan, int window, int boxcar, int comp_fir_size, int peak_hold,
                       rtlpower_cfg *cfg);
/*
 * csv_dbm() (src/rtl_power.c:722-765) for hop `tune`: patches the DC bin, swaps the
 * spectrum halves IN avg[], then formats "Hz low, Hz high, Hz step, samples, dB, dB, ...\n"
 * (without the date/time prefix main() prints, :997-999) into out.  Returns the string
 * length, or -ENOBUFS.  The caller clears the accumulators afterwards (rtlpower_gpu_clear).
 */
int rtlpower_csv_dbm(const rtlpower_plan *plan, int tune, int64_t *avg, int32_t samples, char *out, size_t cap);

/* window_coefs[i] = (int)(256 * window_fn(i, length)) (src/rtl_power.c:985-988); host only */
int rtlpower_window_coefs(int window, int length, int32_t *out);

/* What rtlpower_gpu_create() accepts, without a GPU: 0, or the error it would return (-EINVAL).  bin_e runs to 21,
 * what frequency_range() can plan (src/rtl_power.c:483-486): up to 2^14 points per read the transform lives in a
 * workgroup's LDS, beyond that (bin_e 15 ... 21, or more frames per read) the same stages run over a work buffer in HBM.
 * Outside scanner()'s own domain - a trailing FFT frame that reaches past the read,
 * where the reference transforms whatever its static fft_buf still holds (src/rtl_power.c:695) - is refused. */
int rtlpower_cfg_validate(const rtlpower_cfg *cfg);
int rtlpower_gpu_create(const rtlpower_cfg *cfg, int nstreams, int device, rtlpower_gpu **out);
int rtlpower_gpu_destroy(rtlpower_gpu *h);

/*
 * scanner()'s work for `nreads` consecutive reads of every stream, input
 * resident in device memory: stream s, read r at d_iq + s*stream_stride +
 * r*buf_len.  Accumulates into the handle's avg[] / samples (asynchronous).
 */
int rtlpower_gpu_scan_device(rtlpower_gpu *h, const uint8_t *d_iq, size_t stream_stride, int nreads);

/* The same for one host buffer of one stream (what rtlsdr_read_sync() filled). */
int rtlpower_gpu_scan(rtlpower_gpu *h, int stream, const uint8_t *buf, uint32_t len);

/* tuning_state.avg[0 .. 2^bin_e) and .samples of one stream (what csv_dbm() reads, :722-765). */
int rtlpower_gpu_fetch(rtlpower_gpu *h, int stream, int64_t *avg, int32_t *samples);
/* csv_dbm() zeroes avg[] and samples after reporting (:761-764). */
int rtlpower_gpu_clear(rtlpower_gpu *h);
int rtlpower_gpu_sync(rtlpower_gpu *h);
int rtlpower_gpu_set_stream(rtlpower_gpu *h, void *hip_stream);
/* Cross-stream ordering as rtlfm_gpu_wait_for / rtlfm_gpu_release_to (include/rtlfm_hip.h). */
int rtlpower_gpu_wait_for(rtlpower_gpu *h, void *producer_stream);
int rtlpower_gpu_release_to(rtlpower_gpu *h, void *consumer_stream);
/* Tunables by name, as rtlfm_gpu_set_option: "groups" = workgroups per stream of the FFT kernel
 * (0 = automatic: enough to fill the 256 CUs); "staged_fast" = 0: transforms beyond 16384 bins take the general
 * kernels also where the ones written for rtl_power's own shape (an undecimated read = one frame) apply - A/B and
 * tests, the results are the same integers; "scan_frames" = 0 likewise for reads that hold several frames (the general
 * in-LDS kernel instead of k_power_scan_frames); "dec_fast" = 0 likewise for decimated scans (src/rtl_power.c:466-480, :671-691:
 * one launch per fifth_order pass and the general in-LDS kernel instead of k_power_downsample_iq + k_power_scan_frames<13, true>).
 * "staged_pipe": the batches of a fine-bin scan (one frame per read beyond 16384 points) as a two-stream pipeline - the in-LDS
 * transform on the handle's stream, the comb gather of the next batch and the passes over HBM of the one before beside it on
 * a stream of the handle's own; the scan is complete on the handle's stream either way.  0 (default): never - one batch after the
 * other; 1: from 2^18 bins on and for scans of several batches; 2: wherever it applies.  (Measured: the kernels overlap, the
 * scan gains 4-5 % at 2^19 .. 2^21 bins in two sessions of three and loses 1-3 % at 2^15 .. 2^17.)  "staged_batch" = n > 0: at most n reads per batch (tests).
 * -ENOENT for an unknown name.  get_option reads them back, and "last_kernel" (read-only): which transform the last
 * scan took, one of RTLPOWER_KERNEL_*. */
int rtlpower_gpu_set_option(rtlpower_gpu *h, const char *name, long value);
int rtlpower_gpu_get_option(rtlpower_gpu *h, const char *name, long *value);
enum rtlpower_kernel {
	RTLPOWER_KERNEL_NONE = 0,
	RTLPOWER_KERNEL_GENERAL = 1,      /* k_power_scan: any shape one workgroup's LDS holds */
	RTLPOWER_KERNEL_BIG = 2,          /* k_power_scan_big: one undecimated frame of 8192 / 16384 points per read (BASELINE configs[3]) */
	RTLPOWER_KERNEL_FRAMES = 3,       /* k_power_scan_frames: several undecimated frames per read */
	RTLPOWER_KERNEL_DECIMATED = 4,    /* k_power_downsample_iq | k_power_boxcar + k_power_scan_frames<13, true> */
	RTLPOWER_KERNEL_STAGED = 5,       /* transforms through HBM, the general kernels */
	RTLPOWER_KERNEL_STAGED_FAST = 6   /* ... rtl_power's own fine-bin shape */
};
/* HIP-event timing of the FFT kernel, as rtlfm_gpu_timing_*. */
int rtlpower_gpu_timing_enable(rtlpower_gpu *h, int on);
int rtlpower_gpu_timing_read(rtlpower_gpu *h, double *ms, int *launches);
/*
 * In-kernel clock probe of the large-FFT kernel (bin_e 13 / 14, one frame per read, raw input), as
 * rtlfm_gpu_clock_probe: with probe(1) every workgroup of a launch records the shader clock counter and the
 * 100 MHz real-time counter at its first and last instruction; read() synchronises and returns the mean shader
 * clock (MHz) of the LAST launch and the span from the first workgroup's start to the last one's end (ms), or
 * -ENODATA.  This path is bound by integer VALU issue, which scales with that clock.
 */
int rtlpower_gpu_clock_probe(rtlpower_gpu *h, int on);
int rtlpower_gpu_clock_read(rtlpower_gpu *h, double *shader_mhz, double *span_ms);

#ifdef __cplusplus
}
#endif
#endif
