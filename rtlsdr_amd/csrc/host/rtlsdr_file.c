/*
 * rtlsdr_file.c — librtlsdr's tool-facing API over a raw u8 IQ file
 * (see include/rtlsdr_file.h).  Behaviour follows the reference's async reader
 * where a tool can observe it (src/librtlsdr.c:2826-2952):
 *   - buf_num == 0 -> 15 buffers, buf_len == 0 or not a multiple of 512 -> 32768
 *     (:2847-2855, :407-408);
 *   - the callback runs on the thread that called rtlsdr_read_async and gets a
 *     library-owned buffer that is reused as soon as it returns (:2705-2707);
 *   - rtlsdr_read_async blocks until rtlsdr_cancel_async() or, here, end of file;
 *   - -1 for a NULL device, -2 if a read is already running (:2835-2840).
 */
#include "../../../include/rtlsdr_file.h"

#include <arpa/inet.h>
#include <netdb.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

/*
 * RTLSDR_FILE=tcp://host:port makes the device an rtl_tcp client instead
 * (reference protocol_rtl_tcp.txt, src/rtl_tcp.c:86-90, 378-460, include/rtl_tcp.h:34-50):
 * the server first sends the 12-byte dongle_info {"RTL0", tuner type, gain count}
 * (big endian), then raw u8 I,Q; every setter becomes a 5-byte command
 * {id, 32-bit big-endian parameter}.  This is how thousands of remote dongles
 * can feed one GPU box.
 */
struct rtlsdr_dev {
	int sock;               /* >= 0: rtl_tcp client */
	uint32_t tuner_type, tuner_gain_count;
	FILE *f;
	long data_start;
	uint32_t freq, rate, bw;
	int gain, ppb, agc, direct, offset, bias, gain_mode;
	volatile int async_running, cancel;
	int loop;
	rtlamd_file_buffer_source_t source;  /* zero-copy extension: where the next async buffer is read to */
	void *source_ctx;
};

static const char *env_path(void)
{
	const char *p = getenv("RTLSDR_FILE");
	return (p && *p) ? p : NULL;
}

uint32_t rtlsdr_get_device_count(void) { return env_path() ? 1u : 0u; }

const char *rtlsdr_get_device_name(uint32_t index)
{
	return (index == 0 && env_path()) ? "IQ file (rtlsdr_amd file device)" : "";
}

int rtlsdr_get_device_usb_strings(uint32_t index, char *manufact, char *product, char *serial)
{
	if (index != 0 || !env_path()) return -2;
	if (manufact) strcpy(manufact, "rtlsdr_amd");
	if (product) strcpy(product, !strncmp(env_path(), "tcp://", 6) ? "rtl_tcp" : "file");
	if (serial) strcpy(serial, "00000001");
	return 0;
}

static int tcp_command(rtlsdr_dev_t *d, unsigned char id, uint32_t param)
{
	unsigned char c[5] = {id, (unsigned char)(param >> 24), (unsigned char)(param >> 16),
	                      (unsigned char)(param >> 8), (unsigned char)param};
	if (d->sock < 0) return 0;
	return send(d->sock, c, 5, MSG_NOSIGNAL) == 5 ? 0 : -1;
}

static int tcp_open(rtlsdr_dev_t **out, const char *url)
{
	char host[256];
	const char *colon = strrchr(url, ':');
	if (!colon || (size_t)(colon - url) >= sizeof(host)) return -1;
	memcpy(host, url, (size_t)(colon - url));
	host[colon - url] = 0;
	struct addrinfo hints, *res = NULL;
	memset(&hints, 0, sizeof(hints));
	hints.ai_family = AF_UNSPEC;
	hints.ai_socktype = SOCK_STREAM;
	if (getaddrinfo(host, colon + 1, &hints, &res) != 0 || !res) return -1;
	int s = socket(res->ai_family, res->ai_socktype, res->ai_protocol);
	if (s < 0 || connect(s, res->ai_addr, res->ai_addrlen) != 0) {
		if (s >= 0) close(s);
		freeaddrinfo(res);
		return -1;
	}
	freeaddrinfo(res);
	unsigned char info[12];
	size_t got = 0;
	while (got < 12) {
		ssize_t n = recv(s, info + got, 12 - got, 0);
		if (n <= 0) { close(s); return -1; }
		got += (size_t)n;
	}
	if (memcmp(info, "RTL0", 4) != 0) { fprintf(stderr, "rtlsdr_file: not an rtl_tcp server\n"); close(s); return -1; }
	rtlsdr_dev_t *d = (rtlsdr_dev_t *)calloc(1, sizeof(*d));
	d->sock = s;
	d->tuner_type = ((uint32_t)info[4] << 24) | (info[5] << 16) | (info[6] << 8) | info[7];
	d->tuner_gain_count = ((uint32_t)info[8] << 24) | (info[9] << 16) | (info[10] << 8) | info[11];
	d->rate = 2048000;
	d->freq = 100000000;
	*out = d;
	return 0;
}

int rtlsdr_open(rtlsdr_dev_t **out, uint32_t index)
{
	const char *path = env_path();
	if (!out || index != 0 || !path) return -1;
	if (!strncmp(path, "tcp://", 6)) return tcp_open(out, path + 6);
	FILE *f = fopen(path, "rb");
	if (!f) { perror(path); return -1; }
	rtlsdr_dev_t *d = (rtlsdr_dev_t *)calloc(1, sizeof(*d));
	d->sock = -1;
	d->f = f;
	d->rate = 2048000;
	d->freq = 100000000;
	d->loop = getenv("RTLSDR_FILE_LOOP") && atoi(getenv("RTLSDR_FILE_LOOP"));
	/* rtl_sdr -H writes a RIFF/WAVE header (src/convenience/wavewrite.c:109-156): find "data" */
	unsigned char hdr[12];
	if (fread(hdr, 1, 12, f) == 12 && !memcmp(hdr, "RIFF", 4) && !memcmp(hdr + 8, "WAVE", 4)) {
		unsigned char ck[8];
		while (fread(ck, 1, 8, f) == 8) {
			uint32_t sz = ck[4] | (ck[5] << 8) | (ck[6] << 16) | ((uint32_t)ck[7] << 24);
			if (!memcmp(ck, "data", 4)) break;
			fseek(f, (long)(sz + (sz & 1)), SEEK_CUR);
		}
		d->data_start = ftell(f);
	} else {
		d->data_start = 0;
		fseek(f, 0, SEEK_SET);
	}
	*out = d;
	return 0;
}

int rtlsdr_close(rtlsdr_dev_t *d)
{
	if (!d) return -1;
	if (d->f) fclose(d->f);
	if (d->sock >= 0) close(d->sock);
	free(d);
	return 0;
}

int rtlsdr_set_center_freq(rtlsdr_dev_t *d, uint32_t freq) { if (!d) return -1; d->freq = freq; return tcp_command(d, 0x01, freq); }
uint32_t rtlsdr_get_center_freq(rtlsdr_dev_t *d) { return d ? d->freq : 0; }
int rtlsdr_set_freq_correction_ppb(rtlsdr_dev_t *d, int ppb) { if (!d) return -1; d->ppb = ppb; return tcp_command(d, 0x05, (uint32_t)(ppb / 1000)); }

int rtlsdr_get_tuner_gains(rtlsdr_dev_t *d, int *gains)
{
	/* tenths of a dB, an R820T-like ladder so nearest_gain() has something to pick from */
	static const int table[] = {0, 9, 14, 27, 37, 77, 87, 125, 144, 157, 166, 197, 207, 229, 254, 280,
	                            297, 328, 338, 364, 372, 386, 402, 421, 434, 439, 445, 480, 496};
	int n = (int)(sizeof(table) / sizeof(table[0]));
	if (!d) return -1;
	if (gains) memcpy(gains, table, sizeof(table));
	return n;
}

int rtlsdr_set_tuner_gain(rtlsdr_dev_t *d, int gain) { if (!d) return -1; d->gain = gain; return tcp_command(d, 0x04, (uint32_t)gain); }

int rtlsdr_set_and_get_tuner_bandwidth(rtlsdr_dev_t *d, uint32_t bw, uint32_t *applied_bw, int apply_bw)
{
	if (!d) return -1;
	if (apply_bw) d->bw = bw;
	if (applied_bw) *applied_bw = bw;
	return apply_bw ? tcp_command(d, 0x40, bw) : 0;
}

int rtlsdr_set_tuner_bandwidth(rtlsdr_dev_t *d, uint32_t bw) { return rtlsdr_set_and_get_tuner_bandwidth(d, bw, NULL, 1); }
int rtlsdr_set_tuner_gain_mode(rtlsdr_dev_t *d, int manual) { if (!d) return -1; d->gain_mode = manual; return tcp_command(d, 0x03, (uint32_t)manual); }

int rtlsdr_set_sample_rate(rtlsdr_dev_t *d, uint32_t rate)
{
	if (!d) return -1;
	/* the reference's validity window, src/librtlsdr.c:1633-1637 */
	if (rate <= 225000 || rate > 3200000 || (rate > 300000 && rate <= 900000)) return -22;
	d->rate = rate;
	return tcp_command(d, 0x02, rate);
}

int rtlsdr_set_agc_mode(rtlsdr_dev_t *d, int on) { if (!d) return -1; d->agc = on; return tcp_command(d, 0x08, (uint32_t)on); }
int rtlsdr_set_direct_sampling(rtlsdr_dev_t *d, int on) { if (!d) return -1; d->direct = on; return tcp_command(d, 0x09, (uint32_t)on); }
int rtlsdr_set_ds_mode(rtlsdr_dev_t *d, enum rtlsdr_ds_mode mode, uint32_t thr) { (void)thr; if (!d) return -1; d->direct = (int)mode; return 0; }
int rtlsdr_set_offset_tuning(rtlsdr_dev_t *d, int on) { if (!d) return -1; d->offset = on; return tcp_command(d, 0x0A, (uint32_t)on); }
int rtlsdr_set_bias_tee(rtlsdr_dev_t *d, int on) { if (!d) return -1; d->bias = on; return 0; }
int rtlsdr_set_opt_string(rtlsdr_dev_t *d, const char *opts, int verbose) { (void)opts; (void)verbose; return d ? 0 : -1; }
int rtlsdr_reset_buffer(rtlsdr_dev_t *d) { return d ? 0 : -1; }

static size_t read_some(rtlsdr_dev_t *d, unsigned char *buf, size_t len)
{
	if (d->sock >= 0) {
		size_t have = 0;
		while (have < len) {
			ssize_t n = recv(d->sock, buf + have, len - have, 0);
			if (n <= 0) break;  /* server closed: end of stream */
			have += (size_t)n;
		}
		return have;
	}
	size_t got = fread(buf, 1, len, d->f);
	while (got < len && d->loop) {
		if (fseek(d->f, d->data_start, SEEK_SET) != 0) break;
		size_t more = fread(buf + got, 1, len - got, d->f);
		if (more == 0) break;
		got += more;
	}
	return got;
}

int rtlsdr_read_sync(rtlsdr_dev_t *d, void *buf, int len, int *n_read)
{
	if (!d || !buf || len < 0) return -1;
	size_t got = read_some(d, (unsigned char *)buf, (size_t)len);
	if (n_read) *n_read = (int)got;
	return 0;
}

int rtlsdr_read_async(rtlsdr_dev_t *d, rtlsdr_read_async_cb_t cb, void *ctx, uint32_t buf_num, uint32_t buf_len)
{
	if (!d) return -1;
	if (d->async_running) return -2;
	if (buf_num == 0) buf_num = 15;
	if (buf_len == 0 || buf_len % 512 != 0) buf_len = 16 * 32 * 512;
	unsigned char **bufs = (unsigned char **)calloc(buf_num, sizeof(*bufs));
	for (uint32_t i = 0; i < buf_num; i++) bufs[i] = (unsigned char *)malloc(buf_len);
	d->cancel = 0;
	d->async_running = 1;
	uint32_t k = 0;
	while (!d->cancel) {
		unsigned char *dst = bufs[k];
		uint32_t cap = 0;
		if (d->source) {
			/* the consumer's own memory (e.g. a slot of the GPU layer's pinned ring): read straight into it */
			unsigned char *p = NULL;
			if (d->source(d->source_ctx, &p, &cap) == 0 && p) {
				if (cap >= buf_len) dst = p;
				else if (cb) cb(p, 0, ctx);  /* too small for a transfer: handed back unused (0 bytes), our own buffer instead */
			}
		}
		size_t got = read_some(d, dst, buf_len);
		if (got == 0 && dst == bufs[k]) break;  /* end of file */
		if (cb) cb(dst, (uint32_t)got, ctx);    /* may be short at the very end, as actual_length can be; 0 bytes
		                                         * only for a consumer-owned buffer, which has to be given back */
		if (got < buf_len) break;
		k = (k + 1) % buf_num;
	}
	d->async_running = 0;
	for (uint32_t i = 0; i < buf_num; i++) free(bufs[i]);
	free(bufs);
	return 0;
}

int rtlamd_file_set_buffer_source(rtlsdr_dev_t *d, rtlamd_file_buffer_source_t source, void *ctx)
{
	if (!d) return -1;
	if (d->async_running) return -2;
	d->source = source;
	d->source_ctx = ctx;
	return 0;
}

int rtlsdr_cancel_async(rtlsdr_dev_t *d)
{
	if (!d) return -1;
	if (d->async_running) { d->cancel = 1; return 0; }
	return -2;
}

const char *rtlsdr_get_ver_id(void) { return "rtlsdr_amd file device"; }
uint32_t rtlsdr_get_version(void) { return (0u << 24) | (1u << 16) | (0u << 8) | 0u; }

/* ---- the WAV container helpers, exported for tools and tests (not part of the 26) ---- */
#include "wavhdr.h"
static struct rtlamd_wave g_wave;
void rtlamd_wave_write_header_file(unsigned samplerate, unsigned freq, int bits, int channels, FILE *f)
{
	rtlamd_wave_write_header(&g_wave, samplerate, freq, bits, channels, f);
}
void rtlamd_wave_add_data(uint32_t nbytes) { g_wave.data_size += nbytes; }
void rtlamd_wave_finalize_file(FILE *f) { rtlamd_wave_finalize(&g_wave, f); }
