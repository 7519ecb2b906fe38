/*
 * rtlfm_hip.h — C ABI of the MI355X (gfx950) demodulation layer.
 *
 * This is the drop-in boundary for rtl_fm's per-buffer hot path.  It sits
 * exactly where the reference's async callback hands over its uint8 IQ buffer:
 *
 *   reference boundary                       replaced by
 *   ---------------------------------------  -------------------------------
 *   rtlsdr_read_async_cb_t                   rtlfm_gpu_push()
 *     (include/rtl-sdr.h:472)
 *   rtlsdr_callback body from the u8->i16    rtlfm_gpu_push() + rtlfm_gpu_run()
 *     convert on (src/rtl_fm.c:1326-1343)
 *   full_demod(&demod)                       rtlfm_gpu_run() / _run_device()
 *     (src/rtl_fm.c:1179-1272)
 *   memcpy into output_state + fwrite        rtlfm_gpu_fetch()
 *     (src/rtl_fm.c:1382-1388, 1400)
 *   struct demod_state fields that persist   rtlfm_stream_state via
 *     across blocks (src/rtl_fm.c:172-208)     rtlfm_gpu_state_get/_set()
 *   optimal_settings() + deemph_a            rtlfm_optimal_settings(),
 *     (src/rtl_fm.c:1407-1445, 1929-1931)      rtlfm_deemph_a()
 *
 * Conventions follow include/rtl-sdr.h: every entry point returns int,
 * 0 on success and a negative value on error (-errno style).  All pointers
 * are plain C pointers; there are no C++ or torch types in any signature.
 *
 * One handle batches `nstreams` independent IQ streams that share one
 * configuration (one rtl_fm command line) but each own their filter state.
 * Blocks of one stream are processed in order and never dropped; this is a
 * deliberate, documented difference from rtl_fm's lossy thread hand-off
 * (src/rtl_fm.c:1339-1343, 1368-1371).
 */
#ifndef RTLFM_HIP_H
#define RTLFM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTLFM_MAX_PASSES 10          /* lp_i_hist[10][6], src/rtl_fm.c:178 */
#define RTLFM_MAX_BLOCK_LEN 262144u  /* MAXIMUM_BUF_LENGTH, src/rtl_fm.c:88-90 */

/* demod_state.mode_demod (src/rtl_fm.c:200), selected by -M (src/rtl_fm.c:1820-1841) */
enum rtlfm_mode {
	RTLFM_MODE_FM = 0,   /* fm_demod  src/rtl_fm.c:932  */
	RTLFM_MODE_AM = 1,   /* am_demod  src/rtl_fm.c:961  */
	RTLFM_MODE_USB = 2,  /* usb_demod src/rtl_fm.c:978  */
	RTLFM_MODE_LSB = 3,  /* lsb_demod src/rtl_fm.c:990  */
	RTLFM_MODE_RAW = 4   /* raw_demod src/rtl_fm.c:1002 */
};

/* demod_state.custom_atan, selected by -A (src/rtl_fm.c:1811-1819) */
enum rtlfm_atan {
	RTLFM_ATAN_STD = 0,  /* polar_discriminant src/rtl_fm.c:842 */
	RTLFM_ATAN_FAST = 1, /* polar_disc_fast    src/rtl_fm.c:874 */
	RTLFM_ATAN_LUT = 2   /* polar_disc_lut     src/rtl_fm.c:894 */
};

/* What runs when rate_out2 > 0 (src/rtl_fm.c:1268-1271). */
enum rtlfm_resampler {
	RTLFM_RESAMPLE_LOW_PASS_REAL = 0, /* live path, src/rtl_fm.c:755-775 */
	RTLFM_RESAMPLE_ARBITRARY = 1      /* arbitrary_resample, src/rtl_fm.c:1114-1177
	                                     (call site commented out at :1270) */
};

/*
 * Per-handle configuration: the demod_state / dongle_state fields that are
 * constant while samples flow (src/rtl_fm.c:172-208).  Field names are the
 * reference's.
 */
typedef struct rtlfm_cfg {
	int32_t mode;               /* enum rtlfm_mode */
	int32_t downsample;         /* boxcar factor of low_pass() when passes == 0 */
	int32_t downsample_passes;  /* fifth_order() iterations, 0..10 */
	int32_t comp_fir_size;      /* 0 or 9 (generic_fir with cic_9_tables) */
	int32_t custom_atan;        /* enum rtlfm_atan */
	int32_t post_downsample;    /* low_pass_simple() step, 1 = off */
	int32_t deemph;             /* deemph_filter() on/off */
	int32_t deemph_a;           /* its divisor, see rtlfm_deemph_a() */
	int32_t rate_out;           /* "fast" of low_pass_real() */
	int32_t rate_out2;          /* <= 0: no resampler */
	int32_t resampler;          /* enum rtlfm_resampler */
	int32_t dc_block_audio;     /* dc_block_audio_filter() on/off */
	int32_t adc_block_const;    /* default 9 */
	int32_t dc_block_raw;       /* dc_block_raw_filter() on/off */
	int32_t rdc_block_const;    /* default 9 */
	int32_t offset_tuning;      /* 1: skip rotate16_neg90 (src/rtl_fm.c:1336) */
	int32_t output_scale;       /* am/usb/lsb gain from optimal_settings() */
	int32_t squelch_level;      /* 0 = off (src/rtl_fm.c:1204) */
	uint32_t block_len;         /* bytes per callback buffer = lp_len, multiple of 512 */
	int32_t max_blocks;         /* most blocks per stream one run may take */
	int32_t report_levels;      /* 1: keep the per-buffer rms() of the decimated IQ (what -L prints,
	                               src/rtl_fm.c:1217-1237) for rtlfm_gpu_levels() also when the squelch is off */
} rtlfm_cfg;

/*
 * Everything a stream carries from one block to the next
 * (src/rtl_fm.c:178-199 plus deemph_filter's function-static avg, :1013).
 */
typedef struct rtlfm_stream_state {
	int16_t lp_i_hist[RTLFM_MAX_PASSES][6];
	int16_t lp_q_hist[RTLFM_MAX_PASSES][6];
	int16_t droop_i_hist[9];
	int16_t droop_q_hist[9];
	int16_t pad_[2];
	int32_t now_r, now_j, prev_index;   /* low_pass() */
	int32_t pre_r, pre_j;               /* fm_demod() */
	int32_t now_lpr, prev_lpr_index;    /* low_pass_real() */
	int32_t deemph_avg;                 /* static int avg in deemph_filter() */
	int32_t dc_avg;                     /* dc_block_audio_filter() */
	int32_t dc_avgI, dc_avgQ;           /* dc_block_raw_filter() */
	int32_t squelch_hits;               /* src/rtl_fm.c:1208-1213 */
} rtlfm_stream_state;

typedef struct rtlfm_gpu rtlfm_gpu;

/* ---- host-side planner helpers (no GPU needed) -------------------------- */

/* rtlfm_cfg with demod_init()'s defaults (src/rtl_fm.c:1608-1640). */
void rtlfm_cfg_default(rtlfm_cfg *cfg);

/*
 * optimal_settings() (src/rtl_fm.c:1407-1445).  `use_fifth_order` is the
 * truthy downsample_passes placeholder set by -F (src/rtl_fm.c:1807-1810).
 * Writes downsample, downsample_passes, output_scale into cfg; returns the
 * capture rate through *capture_rate and the tuned frequency through
 * *capture_freq (either may be NULL).
 */
int rtlfm_optimal_settings(rtlfm_cfg *cfg, uint32_t freq, int32_t rate_in,
                           int32_t min_capture_rate, int use_fifth_order,
                           int edge, uint32_t *capture_freq,
                           uint32_t *capture_rate);

/* deemph_a = round(1/(1-exp(-1/(rate_out*tc)))) (src/rtl_fm.c:1929-1931). */
int32_t rtlfm_deemph_a(int32_t rate_out, int32_t time_constant_us);

/* Output samples one block yields, or -1 if it depends on carried state. */
int rtlfm_result_len(const rtlfm_cfg *cfg);
/* Upper bound of output samples per block (always defined). */
int rtlfm_result_cap(const rtlfm_cfg *cfg);

/* What rtlfm_gpu_create() accepts, without a GPU: 0, or the error it would return (-EINVAL, or -EDOM
 * where the reference itself leaves its domain: see rtlfm_gpu_strerror). */
int rtlfm_cfg_validate(const rtlfm_cfg *cfg);

/* ---- the GPU layer ------------------------------------------------------ */

/* Allocates per-stream state (demod_init() values) and work buffers on
 * HIP device `device`.  Fails with -ENODEV if there is no usable GPU: there
 * is no CPU fallback. */
int rtlfm_gpu_create(const rtlfm_cfg *cfg, int nstreams, int device,
                     rtlfm_gpu **out);
int rtlfm_gpu_destroy(rtlfm_gpu *h);

/*
 * The rtlsdr_read_async callback body (rtlsdr_callback, src/rtl_fm.c:1274-1344): append one buffer
 * of interleaved u8 I,Q for `stream`.  `iq` is owned by the caller and may be reused as soon as
 * this returns (the driver resubmits it, src/librtlsdr.c:2705-2707): the bytes are copied into a
 * pinned staging ring.  len is the transfer's actual_length (include/rtl-sdr.h:472): at most
 * cfg.block_len, a multiple of 512; short buffers are demodulated as the reference would (every
 * stage sees that buffer's own length).
 * Callable from the thread that runs rtlsdr_read_async, concurrently for different streams and
 * concurrently with rtlfm_gpu_run(): the ring has two halves, and a callback never waits for a
 * transfer or a kernel.  -ENOSPC when max_blocks buffers are already queued for the stream, -EBUSY while
 * the stream has a slot open between rtlfm_gpu_acquire() and rtlfm_gpu_commit() (one producer per stream).
 */
int rtlfm_gpu_push(rtlfm_gpu *h, int stream, const uint8_t *iq, uint32_t len);

/*
 * The same boundary without the copy: the reference's zero-copy mode hands the callback the kernel's own
 * transfer buffers (use_zerocopy, src/librtlsdr.c:2744-2810); here the producer - a file reader, an rtl_tcp
 * receiver, any device layer that can write to a pointer - writes the samples of `stream`'s next buffer
 * straight into the pinned staging ring:
 *   rtlfm_gpu_acquire   *buf = where to write, *cap = cfg.block_len bytes of room; -ENOSPC when max_blocks
 *                       buffers are queued, -EBUSY while this stream's previous slot is still open
 *   rtlfm_gpu_commit    the first len bytes are a buffer (whole 512-byte packets, at most block_len;
 *                       len == 0 gives the slot back unused)
 * One open slot per stream; different streams concurrently, also while a run is in flight.
 * rtlfm_gpu_run() returns -EAGAIN while a slot is open (it sees every buffer whole or not at all).
 * rtlfm_gpu_push() stays for buffers that belong to someone else (librtlsdr's callback argument).
 */
int rtlfm_gpu_acquire(rtlfm_gpu *h, int stream, uint8_t **buf, uint32_t *cap);
int rtlfm_gpu_commit(rtlfm_gpu *h, int stream, uint32_t len);

/*
 * full_demod() for every queued buffer of every stream (all streams must have the same number
 * queued, else -EAGAIN and nothing changes).  Asynchronous: hands the filled half of the ring to
 * the GPU (async H2D on a copy stream, kernels behind it) and returns; callbacks go on filling the
 * other half, the next run's transfer overlaps this run's kernels.  One caller at a time.
 */
int rtlfm_gpu_run(rtlfm_gpu *h);
/*
 * The same in two steps.  _begin takes what the ring's filling half holds and flips the halves (-EAGAIN as above;
 * *taken, may be NULL = buffers per stream taken) - the only part of a run during which the producers must not
 * acquire or push; _end queues the transfer and the kernels (-EINVAL without a begun run; _begin says -EBUSY while one
 * is pending).  A caller that gates its producers itself (one that counts queued buffers: host/rtl_fm_hip.cpp) holds
 * its gate around _begin only, so that a callback waits for a flip, never for a transfer, a first run's allocations or
 * a kernel launch.  rtlfm_gpu_run() is _begin followed by _end.
 * A FAILED _end (a HIP error, -ENOTSUP) loses the buffers _begin took: the producers have been released, the carried state
 * has not moved on, so the stream now has a gap - what the filters carry no longer matches the input position.  The
 * handle stays usable, but the caller must start the streams afresh (rtlfm_gpu_reset, or rtlfm_gpu_state_set per stream)
 * before the next run if continuity matters; results of earlier runs stay fetchable.
 */
int rtlfm_gpu_run_begin(rtlfm_gpu *h, int *taken);
int rtlfm_gpu_run_end(rtlfm_gpu *h);

/*
 * The same on input already resident in device memory.
 *   d_iq        device pointer; stream s, block b starts at
 *               d_iq + s*stream_stride + b*block_len
 *   d_out       device pointer; stream s writes its concatenated block
 *               results at d_out + s*out_stride (int16 elements)
 *   d_out_len   device pointer to nstreams int32 (total samples per stream),
 *               may be NULL
 * Asynchronous on the handle's stream.
 * Any d_out / out_stride gives the same samples.  Rows that start on 128-byte lines (d_out 128-byte
 * aligned, out_stride a multiple of 64) let the front end's tile stores cover whole lines (1 % of the
 * launch); rows that are at least 16-byte aligned (out_stride a multiple of 8) let the resampler of
 * the -M wbfm tail store eight outputs at a time (17 % of that step).
 */
int rtlfm_gpu_run_device(rtlfm_gpu *h, const uint8_t *d_iq,
                         size_t stream_stride, int nblocks, int16_t *d_out,
                         size_t out_stride, int32_t *d_out_len);

/* Copy out what the last rtlfm_gpu_run() produced for `stream` (what the
 * reference fwrite()s, src/rtl_fm.c:1400).  Blocks until the run is done; the first fetch after a
 * run brings every stream's result over in one transfer, the others are host copies.  Results of a
 * run stay valid until the second run after it. */
int rtlfm_gpu_fetch(rtlfm_gpu *h, int stream, int16_t *out, int cap, int *n);
/* The same for all streams at once: stream s gets lens[s] samples at out + s * out_stride (int16
 * elements; rtlfm_result_cap() * max_blocks is always enough).  One device-to-host transfer. */
int rtlfm_gpu_fetch_all(rtlfm_gpu *h, int16_t *out, size_t out_stride, int32_t *lens);
/* The same for the run BEFORE the last one (-EAGAIN until there have been two).  Waits for that run only,
 * not for the one started since: run(k + 1) as soon as its buffers are in, then fetch_all_prev() for run k,
 * keeps consecutive runs' H2D copies back to back on the link. */
int rtlfm_gpu_fetch_all_prev(rtlfm_gpu *h, int16_t *out, size_t out_stride, int32_t *lens);

/*
 * rms() of the decimated IQ (`sr` in full_demod(), src/rtl_fm.c:1204-1237) of every buffer of the
 * last run for `stream`: what the squelch compared with squelch_level and what -L prints.  Needs
 * cfg.squelch_level or cfg.report_levels; a buffer whose rms() came out negative (the wrapped sum
 * of squares, see rms()) reads INT32_MIN as in the reference.  *n = buffers in the last run.
 */
int rtlfm_gpu_levels(rtlfm_gpu *h, int stream, int32_t *rms, int cap, int *n);

int rtlfm_gpu_state_get(rtlfm_gpu *h, int stream, rtlfm_stream_state *st);
int rtlfm_gpu_state_set(rtlfm_gpu *h, int stream, const rtlfm_stream_state *st);
/* demod_init() values for every stream. */
int rtlfm_gpu_reset(rtlfm_gpu *h);

int rtlfm_gpu_sync(rtlfm_gpu *h);
/* Launch on a caller-owned hipStream_t (NULL = the handle's own stream).  On a caller-owned stream
 * EVERYTHING the handle launches is ordered on that stream, the audio tail (deemph, DC block,
 * resamplers) included: work the caller enqueues on it behind rtlfm_gpu_run_device() sees the
 * finished d_out / d_out_len.  (On its own stream the handle runs the audio tail of a step on a
 * second internal stream, overlapped with the next step's front end; rtlfm_gpu_sync /
 * _release_to / _state_get / _fetch* wait for both.) */
int rtlfm_gpu_set_stream(rtlfm_gpu *h, void *hip_stream);
/*
 * Ordering against another HIP stream without a host synchronisation.  The library launches on
 * its own non-blocking stream, which is NOT ordered with the caller's streams by itself:
 *   rtlfm_gpu_wait_for(h, p)    work launched on the handle from now on starts after everything
 *                               already enqueued on p (the producer of d_iq, the allocator of d_out);
 *   rtlfm_gpu_release_to(h, c)  work enqueued on c from now on starts after everything the handle
 *                               has launched so far (the consumer of d_out / d_out_len).
 * NULL names the legacy default stream.  A caller that uses neither must rtlfm_gpu_sync().
 * (The handle's own streams come from a process-wide pool and go back to it when the handle is destroyed; the library
 * never calls hipStreamDestroy.  Not an economy: on the runtime this was written against, a stream that had been ordered
 * against another stream this way and was then destroyed had a freed object of its own released once more, later, in
 * whoever's heap block it had become - csrc/stream_pool.h, LAB.md I.21.  A caller that creates and destroys streams of
 * its own around these calls at a high rate may want to pool them too.)
 */
int rtlfm_gpu_wait_for(rtlfm_gpu *h, void *producer_stream);
int rtlfm_gpu_release_to(rtlfm_gpu *h, void *consumer_stream);

/*
 * Tunables and A/B switches of one handle, by name (none of them changes a result; the parity
 * suite runs under several).  The library reads no environment variable while samples flow:
 * RTLFM_OPTIONS="name=value,name=value" is applied once, inside rtlfm_gpu_create().
 *   fused_waves          waves a front-end launch aims for (default 8192 = twice the GPU's wave slots);
 *                        a stream's run is cut into that many segments / nstreams.  Sets fused_waves_tail too.
 *   fused_waves_tail     the same for configurations with an audio tail behind the front end (deemph, DC block,
 *                        resamplers; default 20480 behind the boxcar, 12288 behind fifth_order passes: shorter
 *                        segments let the tail's waves in sooner)
 *   fused_min_tiles      shortest segment in 8 KiB tiles (default 8: each segment but a stream's first
 *                        re-runs one warm-up tile); not applied while the launch cannot fill the GPU
 *   fused_tiles_per_seg  > 0: exactly this segment length (tests)
 *   pass0_engine         -1 compiled default, 0 v_dot4, 1 int8 MFMA (as rtlfm_gpu_set_path 3 / 4)
 *   tail_serial          1: audio tail on the front end's stream, no overlap with the next step
 *   tail_priority        -1 (default) / 0 / 1: the tail's own stream at the lowest / default / highest HIP priority.  Not a
 *                        scheduling hint: streams of one priority share four hardware queues, and a tail that shares the
 *                        front end's queue runs behind the next step instead of beside it; another priority = another pool
 *   deemph_sequential    1: deemph_filter one lane per stream, never parallel over time
 *   deemph_four_pass     1: no one-pass (speculative) deemph kernels
 *   lpr_separate         1: low_pass_real as a kernel of its own behind deemph_filter
 *   lpr_scalar_stores    1: the resampler's outputs one by one
 *   lpr_chunk            most samples per lane of the one-pass deemph + low_pass_real kernel (default 5440; 256 ... 2^20,
 *                        anything else -EINVAL); shorter runs get shorter chunks so that about 64 K lanes work
 *   lpr_threads          lanes per workgroup of that kernel: 64, 128, 192 or 256 (default; anything else -EINVAL).  A
 *                        workgroup owns whole streams (lpr_threads / chunks of them, or one with a loop over its chunks)
 *   lpr_slim             1: deemph_filter + low_pass_real behind a front end (-M wbfm) as k_lpr_slim_plan + k_deemph_lpr_slim:
 *                        one-wave workgroups of 32 registers and no LDS, which run as a FIFTH wave per SIMD beside the next
 *                        step's four front-end waves instead of in the place of one.  Bit-exact and no faster (the /6 front
 *                        end has no issue cycles to spare: the step stays front end + tail, LAB.md I.22): kept for A/B;
 *                        0 (default): k_deemph_spec_lpr (round 5)
 *   lpr_slim_chunk       samples per lane of that kernel (default 6120: 16 chunks per stream and one wave per SIMD at the
 *                        wbfm shape; 256 ... 2^20);  lpr_slim_prio: its waves' s_setprio, 0 ... 3 (default 3)
 *   lpr_ring             1 (default): that kernel's outputs leave through LDS in aligned 64-byte pieces; 0: 16 bytes per lane
 *   squelch_fused        1 (default): rms()'s sums per buffer inside the front end + k_squelch_apply; 0: emit mode + k_squelch_*
 *   adc_separate         1: dc_block_audio as sums / smooth / apply kernels (round 4) instead of sums + k_adc_smooth_apply
 *   box_store            how k_boxcar_scan's outputs leave: -1 (default) by the launch's output size - beyond 192 MiB (what the
 *                        256 MiB Infinity Cache cannot keep anyway) as whole 128-byte lines with non-temporal stores, the
 *                        rest of a tile's last line waiting in LDS for the next tile; below, plain stores -, 0 / 1 = always
 *                        plain / always lines
 *   fused_store          the fifth_order front end's PCM stores with 1-3 passes (4 / 2 / 1 KiB of PCM per 8 KiB tile): -1 (default)
 *                        non-temporal with three passes (whole lines per instruction) when the launch writes more than
 *                        192 MiB, 0 / 1 = never / always (with one or two passes non-temporal stores are slower: tests only)
 *   deep_rest            1 (default): passes 6 ... 9, generic_fir and the demodulator behind k_fused<6>'s emit mode in ONE
 *                        kernel per step (k_deep_rest); 0: one staged kernel per stage
 *   apart_budget_gb      most device memory (GiB, default 16, never more than half of what is free) a placement
 *                        search may hold in candidate allocations; 0 = no search, plain allocations
 *   arb_span             1: config 3's one-kernel tail as k_deemph_arb_span (the span linear in LDS, 16-byte table entries,
 *                        a four-instruction filter step for a == 2) instead of k_deemph_spec_arb: 18 % fewer instructions,
 *                        the same time - kept for A/B
 *   arb_chunk            samples per lane of k_deemph_arb_span: 32 (default) or 64, anything else -EINVAL
 *   arb_serial           where deemph_filter + arbitrary_upsample (config 3's tail) runs: 1 = on the front end's stream, behind
 *                        it; 0 = on the tail's own stream beside the next front end, as every other tail; -1 (default) = in
 *                        line from 2048 streams on (1.5 % of config 3's step).  Setting it waits for the handle's work
 *   arb_waves            waves per stream of k_deemph_spec_arb (each takes every arb_waves-th span of 2048 samples): 0
 *                        (default) = as many as make about 16384 waves of all streams, else 1 .. 8; outside -EINVAL
 *   verify_twice         debugging: 1 = every rtlfm_gpu_run_device() executes its run TWICE from the same carried state - into
 *                        shadow rows first, then into the caller's, the device idle in between - and compares rows, lengths and
 *                        the state records on the device; a difference is reported on stderr and counted.  Separates a transient
 *                        fault of the device code from a deterministic one (tests/test_soak_gpu.py).  Twice the time and a
 *                        second set of output rows; ragged runs (short callback buffers) are not verified
 * Read-only (rtlfm_gpu_get_option):
 *   verify_runs          runs executed under verify_twice so far, and
 *   verify_mismatches    ... how many of them differed between their two executions
 *   ring_apart           1 / 0: the result buffers behind rtlfm_gpu_push() / _run() are / are not a quarter of the HBM
 *                        away from the ring's device input; -1 before the ring exists (it is built by the first push)
 *   ring_tries           searches the ring's placement took: 2 = the first found every candidate in the input's class and the
 *                        ring's own device inputs were moved once (the handle owns both sides there)
 *   res_apart            the same for the audio tail's work buffers against the first run's input (-1: none yet;
 *                        0 also on a caller-owned stream (rtlfm_gpu_set_stream), where no search is made - the search
 *                        times launches on the null stream and synchronises the device; a caller on its own stream
 *                        places its OWN output with rtlfm_gpu_malloc_apart before the first run: INTEGRATION.md)
 *   deep_apart           the same for the buffer a front end's emit mode writes (-M raw, the squelch, -L, 7-10 passes);
 *                        -1: this configuration has none / not allocated yet
 *   placement_ms         wall time the placement searches of this handle took, in all
 *   placement_walked_mb  most a search held in temporary allocations (MiB)
 *   placement_held_mb    what the handle's placed blocks hold NOW (MiB): a search that finds a place returns the candidate
 *                        itself, which may be a 1, 2 or 4 GiB block for a smaller request (the buffers that belong together
 *                        share one); a search that finds nothing keeps nothing and the block is exactly the request
 *   poison               1 when RTLFM_POISON=1 was in the environment at the library's first allocation: every device
 *                        allocation of both libraries is filled with 0xA5 and every run / scan first leaves 0xA5 in all of
 *                        every CU's LDS, so that nothing read before it is written goes unnoticed (debug_poison.h;
 *                        tests/test_poison_gpu.py runs the parity suites that way)
 *   ring_force_retry     1: the ring's first placement search counts as failed (tests: the path that moves the device inputs)
 *   tail_sync            1: synchronise and report after every tail kernel (debugging)
 *   fused_debug          clock-stamp experiments (2 / 18 / 4, see fused_kernel.h)
 * Returns -ENOENT for an unknown name, -EINVAL for a value out of range.
 */
int rtlfm_gpu_set_option(rtlfm_gpu *h, const char *name, long value);
int rtlfm_gpu_get_option(rtlfm_gpu *h, const char *name, long *value);

/* 0 = automatic, 1 = staged reference kernels, 2 = fused streaming kernel
 * (fails with -ENOTSUP at run time when the configuration has no fused
 * form); 3 / 4 = fused with the first decimation pass forced onto v_dot4 /
 * onto the int8 MFMA pipe (2 takes the compiled default).  For tests and A/B measurement. */
int rtlfm_gpu_set_path(rtlfm_gpu *h, int path);
/* Which path the last run took (1 or 2). */
int rtlfm_gpu_last_path(rtlfm_gpu *h);

/*
 * HIP-event timing of the decimating front-end kernel (the dominant kernel)
 * on the stream it is launched on.  enable(1) starts recording one event pair
 * per run; read() synchronises and returns the summed milliseconds and the
 * number of launches since the last read, then clears them.
 */
int rtlfm_gpu_timing_enable(rtlfm_gpu *h, int on);
int rtlfm_gpu_timing_read(rtlfm_gpu *h, double *front_ms, int *launches);

/*
 * In-kernel clock probe of the fused fifth_order front end: with probe(1) every wave of a launch
 * records the shader clock counter and the 100 MHz real-time counter at its first and last
 * instruction; read() synchronises and returns the mean shader clock (MHz) the waves of the LAST
 * launch ran at and the span from the first wave's start to the last wave's end (ms), or -ENODATA.
 * (The package runs at its power cap on this path: the clock, not the instruction count, is what
 * moves between boxes and over a run.)
 */
int rtlfm_gpu_clock_probe(rtlfm_gpu *h, int on);
int rtlfm_gpu_clock_read(rtlfm_gpu *h, double *shader_mhz, double *span_ms);
/* The raw stamps of that launch: out[4 w .. 4 w + 3] = wave w's shader clock at its first / last
 * instruction and the 100 MHz counter at its first / last instruction.  *waves = waves of the launch
 * (out may be NULL to ask); -ENOBUFS when cap_waves is smaller. */
int rtlfm_gpu_clock_stamps(rtlfm_gpu *h, uint64_t *out, int cap_waves, int *waves);

/*
 * The box's own HBM streaming ceilings (SURVEY.md §8d asks for a measured ceiling next to the
 * nominal 8 TB/s): the front-end kernels' skeleton - one wave per contiguous segment, 8 KiB tiles,
 * non-temporal coalesced 16-byte loads with the next tile in flight, the same LDS footprint - and
 * none of their arithmetic, over `bytes` (>= 256 MiB) of device memory, `reps` launches each:
 *   *read_gbs          read only
 *   *rw_gbs            the same with one byte stored per ~write_div bytes read (16 = the /16 chain's PCM),
 *                      the written bytes a quarter of the HBM away from the read ones (rtlfm_gpu_malloc_apart);
 *                      (bytes read + bytes written) / time
 *   *rw_colocated_gbs  the same with input and output inside one allocation (the same quarter)
 *   *write_fraction    the share actually stored (2, 4, 8 or 16 bytes per lane and tile: 1/64 ... 1/8)
 * Allocates and frees its own buffers; no handle needed.  Returns 1 when the "apart" placement was found,
 * 0 when it was not (then *rw_gbs is another co-located figure), negative on error.
 */
int rtlfm_gpu_bw_probe(int device, size_t bytes, int write_div, int reps, double *read_gbs, double *rw_gbs,
                       double *rw_colocated_gbs, double *write_fraction);

/*
 * Diagnostic: evaluates the kernels' atan2 -> Q14 routine (the arithmetic of
 * polar_discriminant, src/rtl_fm.c:842-849) and the device math library's
 * atan2 chain on n host pairs yx[2k] = y, yx[2k+1] = x.  Either output may be
 * NULL.  Used by the parity tests to pin one against the other.
 */
int rtlfm_gpu_selftest_atan2(int device, const int32_t *yx, int n, int32_t *q14, int32_t *q14_libm);
/* The kernels' fast_atan2 (src/rtl_fm.c:851-872, incl. its 32-bit wrap-around) on n host pairs. */
int rtlfm_gpu_selftest_fast_atan2(int device, const int32_t *yx, int n, int32_t *q14);
/* q[i] = nd[2i] / nd[2i+1] (C's truncating division, divisor >= 1) as the resampler's walk
 * divides by fast / slow (src/rtl_fm.c:769): the magic-number form of staged_kernels.h */
int rtlfm_gpu_selftest_const_div(int device, const int32_t *nd, int n, int32_t *q);

/*
 * rotate_90 on raw u8 IQ (src/rtl_fm.c:437-447, NEG_U8(x) = 255 - x, :375-392): sample n
 * times (+j)^n, in place on device memory, len bytes (a multiple of 8, 16-byte aligned
 * buffer), on hip_stream (NULL = the default stream).  The reference keeps this function
 * but never calls it (its live path rotates the int16 copy by -90 degrees, which the
 * decimating kernels fold into their taps); it is offered as a standalone operator.
 */
int rtlfm_gpu_rotate_90_u8(int device, void *d_buf, size_t len, void *hip_stream);

/*
 * Where the output lives relative to the input matters on MI355X: every allocation belongs to one of (as far as the
 * probes have seen) three classes of its HBM, and a kernel that reads from one class and writes into the SAME class -
 * the PCM is 1/16 of the bytes at /16 - streams 5.6 TB/s where it streams 6.5 TB/s with the writes in another class
 * (read only: 6.9).  Buffers allocated one after the other normally share a class.
 *
 * rtlfm_gpu_malloc_apart: `bytes` of device memory for a write stream that runs beside the read stream
 * of `other` (other_bytes long; only read).  A bounded search: at most eight candidates - the request's own size, then
 * 1, 2, 1, 4, 1, 2, 4 GiB (the driver's allocator serves different sizes from different places) -, each timed against
 * `other` with one short bandwidth probe, never more than 16 GiB (rtlfm_gpu_malloc_apart_ex: budget_bytes) or half of
 * the free device memory held at once, typically 5-30 ms; the step that succeeded on a device is tried first the next
 * time in this process.  A winning candidate larger than the request is kept whole (the library asks for the buffers of
 * one handle that belong together as ONE block; "placement_held_mb" says what a handle's placed blocks hold).  *apart (may
 * be NULL) = 1 when found; otherwise - buffers too small to matter (< 256 MiB streamed), no budget, every candidate in
 * `other`'s class, probe failure - every candidate is freed and a plain allocation of exactly `bytes` is returned with
 * *apart = 0: the result is always usable, nothing of a failed search is kept, and the caller can see which it got.
 * A decision rests on the MEDIAN of three separately timed probe launches per candidate.
 * The library's own result buffers behind rtlfm_gpu_push() / _run() are placed this way.
 * rtlfm_gpu_placement_probe: 1 if existing buffers `in` / `out` are in different classes, 0 if not (or too
 * small to tell); OVERWRITES the first in_bytes / 16 bytes of `out`.
 * rtlfm_gpu_malloc / _free: plain device memory through the library.
 */
int rtlfm_gpu_malloc(int device, size_t bytes, void **out);
int rtlfm_gpu_malloc_apart(int device, size_t bytes, const void *other, size_t other_bytes, void **out, int *apart);
/* The same with the search's cost in the open: budget_bytes = most the search may hold in candidates (0: no search, plain
 * memory; rtlfm_gpu_malloc_apart uses 16 GiB; never more than half of the free device memory is taken),
 * *search_ms = wall time of the call, *walked_bytes = most it held at once (either may be NULL). */
int rtlfm_gpu_malloc_apart_ex(int device, size_t bytes, const void *other, size_t other_bytes, size_t budget_bytes,
                              void **out, int *apart, double *search_ms, size_t *walked_bytes);
int rtlfm_gpu_placement_probe(int device, const void *in, size_t in_bytes, void *out, size_t out_bytes,
                              double *read_ms, double *rw_ms);
/*
 * The caller owns BOTH sides (a service that allocates its own device input and output for rtlfm_gpu_run_device, on its
 * own stream or not): one call that chooses the PAIR.  in_bytes of input memory and out_bytes of output memory in
 * different classes of the HBM: an input is allocated, a bounded search (as rtlfm_gpu_malloc_apart_ex, budget_bytes)
 * looks for its partner; if every candidate shares the input's class the input itself MOVES - a new one is allocated
 * while the old one is still held, so that it comes from somewhere else - and the search runs again, up to max_tries
 * searches (1 ... 8; the library's ring does the same with its own device inputs).  Whatever was only held to push the
 * allocator on is freed before the call returns.  *in is uninitialised device memory for the caller to fill; both
 * pointers are released with rtlfm_gpu_free.  *apart = 1 when a pair was found (0: plain allocations, usable all the
 * same), *tries = searches made, *search_ms / *walked_bytes = wall time and most memory held at once (any may be NULL).
 * Runs probe launches on the null stream and synchronises the device: call it at set-up, not while samples flow.
 */
int rtlfm_gpu_place_pair(int device, size_t in_bytes, size_t out_bytes, size_t budget_bytes, int max_tries,
                         void **in, void **out, int *apart, int *tries, double *search_ms, size_t *walked_bytes);
/* bytes from one device buffer to another on `device` (synchronous), for callers without a HIP runtime at hand. */
int rtlfm_gpu_copy(int device, void *dst, const void *src, size_t bytes);
int rtlfm_gpu_free(void *p);
/* NUMA node of the host the device hangs on (sysfs), or -1 if unknown: where the threads that fill the
 * device's staging ring should run. */
int rtlfm_gpu_device_numa_node(int device);

/*
 * The front-end planner by itself (host only, no GPU): how a launch over `nstreams` streams of `total_tiles` 8 KiB tiles each
 * is cut into segments (one wave per stream and segment).  fifth_order / tail_follows as the run would set them;
 * target_waves, min_tiles, tiles_per_seg, gss_x10: the options of the same names (0 = the library's defaults).
 * *segs = segments per stream, starts[0 .. *segs] their tile boundaries (-ENOBUFS when cap < *segs + 1).
 */
int rtlfm_plan_segments(int nstreams, int total_tiles, int fifth_order, int tail_follows, int target_waves, int min_tiles,
                        int tiles_per_seg, int gss_x10, int *segs, int *starts, int cap);

const char *rtlfm_gpu_strerror(int err);
/* (major<<16)|(minor<<8)|patch */
int rtlfm_gpu_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RTLFM_HIP_H */
