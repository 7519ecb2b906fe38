/*
 * rtlfm_oracle.h — CPU restatement of rtl_fm's demod chain.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped HIP path may include,
 * link or call this: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker.
 *
 * Parity pin: the reference has no tests or golden vectors of its own
 * (SURVEY.md §4), so this restatement is pinned against the reference
 * itself, compiled in place from /root/reference/src/rtl_fm.c into
 * oracle/_ref/ (see oracle/Makefile, oracle/ref_rtlfm_harness.c), and against
 * the fixtures under tests/golden/ that build generated
 * (tests/golden/gen_golden.py).
 *
 * Each function cites the reference lines it restates.  The shared POD
 * structs (configuration, carried per-stream state) come from the public
 * header so that oracle and product are driven by identical inputs.
 */
#ifndef RTLFM_ORACLE_H
#define RTLFM_ORACLE_H

#include <stdint.h>
#include "../include/rtlfm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* demod_init() values, src/rtl_fm.c:1608-1640 (hist arrays are BSS zeros) */
void orc_state_init(rtlfm_stream_state *st);

/* ---- single stages ------------------------------------------------------ */
/* src/rtl_fm.c:1326-1328 */
void orc_u8_to_i16(const uint8_t *in, int16_t *out, int len);
/* src/rtl_fm.c:424-434 */
void orc_rotate16_neg90(int16_t *buf, int len);
/* src/rtl_fm.c:437-447 (dead code in the reference; named by the north star) */
void orc_rotate_90_u8(uint8_t *buf, int len);
/* src/rtl_fm.c:1043-1065 */
void orc_dc_block_raw(int16_t *buf, int len, int k, int32_t *avg_i, int32_t *avg_q);
/* src/rtl_fm.c:461-481; returns new lp_len */
int orc_low_pass(int16_t *lp, int lp_len, int downsample, int32_t *now_r,
                 int32_t *now_j, int32_t *prev_index);
/* src/rtl_fm.c:777-806; data is one component of interleaved IQ */
void orc_fifth_order(int16_t *data, int length, int16_t hist[6]);
/* src/rtl_fm.c:808-831 with table row src/rtl_fm.c:355-367 */
void orc_generic_fir(int16_t *data, int length, int passes, int16_t hist[9]);
const int *orc_cic9_row(int passes);
/* src/rtl_fm.c:842-849, 851-879, 881-930 */
int orc_polar_discriminant(int ar, int aj, int br, int bj);
int orc_polar_disc_fast(int ar, int aj, int br, int bj);
int orc_fast_atan2(int y, int x); /* fast_atan2 alone (src/rtl_fm.c:851-872) */
int orc_polar_disc_lut(int ar, int aj, int br, int bj);
const int32_t *orc_atan_lut(void); /* 131072 entries, built on first use */
/* src/rtl_fm.c:932-959; returns result_len */
int orc_fm_demod(const int16_t *lp, int lp_len, int16_t *result, int custom_atan,
                 int32_t *pre_r, int32_t *pre_j);
/* src/rtl_fm.c:961-1009 */
int orc_am_demod(const int16_t *lp, int lp_len, int16_t *result, int output_scale);
int orc_usb_demod(const int16_t *lp, int lp_len, int16_t *result, int output_scale);
int orc_lsb_demod(const int16_t *lp, int lp_len, int16_t *result, int output_scale);
int orc_raw_demod(const int16_t *lp, int lp_len, int16_t *result);
/* src/rtl_fm.c:739-753 */
int orc_low_pass_simple(int16_t *sig, int len, int step);
/* src/rtl_fm.c:1011-1026 */
void orc_deemph(int16_t *result, int len, int a, int32_t *avg);
/* src/rtl_fm.c:1028-1041 */
void orc_dc_block_audio(int16_t *result, int len, int k, int32_t *dc_avg);
/* src/rtl_fm.c:755-775; returns new result_len, -1 if fast/slow == 0 */
int orc_low_pass_real(int16_t *result, int len, int fast, int slow,
                      int32_t *now_lpr, int32_t *prev_lpr_index);
/* src/rtl_fm.c:1114-1177 */
void orc_arbitrary_upsample(const int16_t *b1, int16_t *b2, int len1, int len2);
void orc_arbitrary_downsample(const int16_t *b1, int16_t *b2, int len1, int len2);
/* src/rtl_fm.c:1083-1112 */
int orc_rms(const int16_t *samples, int len, int step, int omit_dc_fix);

/* ---- planner ------------------------------------------------------------ */
/* src/rtl_fm.c:1407-1445 */
void orc_optimal_settings(rtlfm_cfg *cfg, uint32_t freq, int rate_in,
                          int min_capture_rate, int use_fifth_order, int edge,
                          uint32_t *capture_freq, uint32_t *capture_rate);
/* src/rtl_fm.c:1929-1931 */
int orc_deemph_a(int rate_out, int tc_us);

/* ---- whole chain -------------------------------------------------------- */
/*
 * One callback buffer through rtlsdr_callback's conversion
 * (src/rtl_fm.c:1326-1338) and full_demod() (src/rtl_fm.c:1179-1272).
 * out must hold rtlfm_result_cap()-many int16.  Returns result_len (>= 0)
 * or a negative value for configurations the reference crashes on.
 */
int orc_block(const rtlfm_cfg *cfg, rtlfm_stream_state *st, const uint8_t *iq,
              uint32_t len, int16_t *out);

/*
 * nblocks consecutive buffers for each of nstreams streams, `nthreads`
 * pthreads each owning a contiguous range of streams.  Layout as
 * rtlfm_gpu_run_device().  out_len[s] receives the total per stream.
 * Returns 0 or the first negative orc_block() result.
 */
int orc_run_batch(const rtlfm_cfg *cfg, rtlfm_stream_state *st, int nstreams,
                  const uint8_t *iq, size_t stream_stride, int nblocks,
                  int16_t *out, size_t out_stride, int32_t *out_len,
                  int nthreads);

int orc_result_cap(const rtlfm_cfg *cfg);

#ifdef __cplusplus
}
#endif
#endif
