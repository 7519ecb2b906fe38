/*
 * rtlsdr_file.h — the 26 librtlsdr entry points rtl_fm, rtl_power and the
 * convenience helpers link against, implemented over a raw
 * uint8 IQ file instead of a USB dongle (librtlsdr_file.so).
 *
 * Signatures and return conventions are those of the reference's public header
 * (include/rtl-sdr.h: open :70, close :78, center_freq :165/:173, gains :258-:341,
 * sample_rate :354, agc :381, direct sampling :393, ds_mode :404-:421,
 * offset tuning :431, reset_buffer :461, read_sync :470, read_async :472-:492,
 * cancel_async :500, bias tee :522, opt_string :559, versions :588/:595,
 * device enumeration :39-:57), so a tool built against that header links here
 * unchanged.  The input is what rtl_sdr writes (src/rtl_sdr.c:97-121): raw
 * interleaved u8 I,Q, optionally behind a RIFF/WAVE header, which is skipped.
 *
 * The file comes from the environment: RTLSDR_FILE=<path> (required),
 * RTLSDR_FILE_LOOP=1 to wrap around at end of file instead of ending the
 * asynchronous read.
 */
#ifndef RTLSDR_FILE_H
#define RTLSDR_FILE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtlsdr_dev rtlsdr_dev_t;
typedef void (*rtlsdr_read_async_cb_t)(unsigned char *buf, uint32_t len, void *ctx);

enum rtlsdr_ds_mode {
	RTLSDR_DS_IQ = 0,
	RTLSDR_DS_I,
	RTLSDR_DS_Q,
	RTLSDR_DS_I_BELOW,
	RTLSDR_DS_Q_BELOW
};

uint32_t rtlsdr_get_device_count(void);
const char *rtlsdr_get_device_name(uint32_t index);
int rtlsdr_get_device_usb_strings(uint32_t index, char *manufact, char *product, char *serial);
int rtlsdr_open(rtlsdr_dev_t **dev, uint32_t index);
int rtlsdr_close(rtlsdr_dev_t *dev);
int rtlsdr_set_center_freq(rtlsdr_dev_t *dev, uint32_t freq);
uint32_t rtlsdr_get_center_freq(rtlsdr_dev_t *dev);
int rtlsdr_set_freq_correction_ppb(rtlsdr_dev_t *dev, int ppb);
int rtlsdr_get_tuner_gains(rtlsdr_dev_t *dev, int *gains);
int rtlsdr_set_tuner_gain(rtlsdr_dev_t *dev, int gain);
int rtlsdr_set_and_get_tuner_bandwidth(rtlsdr_dev_t *dev, uint32_t bw, uint32_t *applied_bw, int apply_bw);
int rtlsdr_set_tuner_bandwidth(rtlsdr_dev_t *dev, uint32_t bw);
int rtlsdr_set_tuner_gain_mode(rtlsdr_dev_t *dev, int manual);
int rtlsdr_set_sample_rate(rtlsdr_dev_t *dev, uint32_t rate);
int rtlsdr_set_agc_mode(rtlsdr_dev_t *dev, int on);
int rtlsdr_set_direct_sampling(rtlsdr_dev_t *dev, int on);
int rtlsdr_set_ds_mode(rtlsdr_dev_t *dev, enum rtlsdr_ds_mode mode, uint32_t freq_threshold);
int rtlsdr_set_offset_tuning(rtlsdr_dev_t *dev, int on);
int rtlsdr_reset_buffer(rtlsdr_dev_t *dev);
int rtlsdr_read_sync(rtlsdr_dev_t *dev, void *buf, int len, int *n_read);
int rtlsdr_read_async(rtlsdr_dev_t *dev, rtlsdr_read_async_cb_t cb, void *ctx, uint32_t buf_num, uint32_t buf_len);
int rtlsdr_cancel_async(rtlsdr_dev_t *dev);

/*
 * Extension (not one of the 26): zero-copy reads.  The reference's USB layer can hand the callback the
 * kernel's own transfer buffers (use_zerocopy, src/librtlsdr.c:2744-2810); the counterpart here lets the
 * consumer say where the next buffer is to be read TO - e.g. rtlfm_gpu_acquire()'s slot of the pinned
 * staging ring.  `source(ctx, &buf, &cap)` is called before every read of rtlsdr_read_async(): 0 = read
 * up to buf_len bytes into buf (cap >= buf_len), anything else = use the library's own buffer for this one.
 * The callback then receives that pointer.  NULL switches it off.
 */
typedef int (*rtlamd_file_buffer_source_t)(void *ctx, unsigned char **buf, uint32_t *cap);
int rtlamd_file_set_buffer_source(rtlsdr_dev_t *dev, rtlamd_file_buffer_source_t source, void *ctx);
int rtlsdr_set_bias_tee(rtlsdr_dev_t *dev, int on);
int rtlsdr_set_opt_string(rtlsdr_dev_t *dev, const char *opts, int verbose);
const char *rtlsdr_get_ver_id(void);
uint32_t rtlsdr_get_version(void);

#ifdef __cplusplus
}
#endif
#endif
