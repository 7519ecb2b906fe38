/*
 * rtlpower_oracle.h — CPU restatement of rtl_power's scanner() DSP.
 * TEST INFRASTRUCTURE ONLY (see rtlfm_oracle.h).
 *
 * Pin: every FUNCTION of the path (sine_table, FIX_MPY/fix_fft, the window
 * functions, fifth_order, generic_fir, remove_dc, real_conj, rms_power) is
 * checked against the reference's own code compiled in place into
 * oracle/_ref/libref_rtlpower.so.  scanner() itself cannot be called — its first
 * statement calls into librtlsdr, which this image cannot build and for which
 * no stand-ins are written — so its ~25 lines of inline glue (u8 conversion,
 * boxcar loop, window multiply, accumulation; src/rtl_power.c:666-681, 697-717)
 * are restated in oracle/ref_rtlpower_harness.c around the reference's
 * functions.  Fixtures in tests/golden/power_*.npz come from that build.
 */
#ifndef RTLPOWER_ORACLE_H
#define RTLPOWER_ORACLE_H

#include <stdint.h>
#include "../include/rtlpower_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* src/rtl_power.c:247-261; returns a malloc'd table of 3N/4 entries */
int16_t *orcp_sine_table(int log2n);
/* src/rtl_power.c:271-327 */
int orcp_fix_fft(int16_t *iq, int m, const int16_t *sinewave, int log2_n_wave);
/* src/rtl_power.c:329-408 */
double orcp_window(int window, int i, int length);
void orcp_window_coefs(int window, int length, int32_t *out);
/* src/rtl_power.c:554-579 (stateless, "ease in") */
void orcp_fifth_order(int16_t *data, int length);
/* src/rtl_power.c:598-626 */
void orcp_generic_fir(int16_t *data, int length, int passes);
/* src/rtl_power.c:581-596 */
void orcp_remove_dc(int16_t *data, int length);
/* src/rtl_power.c:410-436 */
void orcp_rms_power(const uint8_t *buf, int buf_len, int peak_hold, int64_t *avg0, int32_t *samples);

/* scanner() for one read of one tuning_state (src/rtl_power.c:657-718) */
int orcp_scan(const rtlpower_cfg *cfg, const uint8_t *buf8, int64_t *avg, int32_t *samples);
/* nreads reads for each of nstreams streams, pthreads over streams */
int orcp_scan_batch(const rtlpower_cfg *cfg, int nstreams, const uint8_t *iq, size_t stream_stride,
                    int nreads, int64_t *avg /* [nstreams][2^bin_e] */, int32_t *samples, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
