// wavhdr.h — the RIFF/WAVE header rtl_fm -H and rtl_sdr -H write
// (reference src/convenience/wavewrite.c:109-248): 120 bytes, little endian,
//   "RIFF" size "WAVE" | "fmt " 16 PCM(1) channels rate rate*bytes/frame blockAlign bits |
//   "auxi" 68 StartTime[8 x u16] StopTime[8 x u16] centerFreq ADsamplerate IF BW IQOffset 4 x 0 |
//   "data" size
// The "auxi" chunk is what SpectraVue / HDSDR read the tuned frequency from.
// Quirk kept: nBlockAlign is written as the channel count, not bytes per frame
// (wavewrite.c:196).  Header-only so the CLI and the device-layer library share it.
#pragma once

#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/time.h>
#include <time.h>

struct rtlamd_wave {
	unsigned char hdr[120];
	uint32_t data_size;
	int started;
};

static inline void rtlamd_put16(unsigned char *p, uint32_t v) { p[0] = v & 0xff; p[1] = (v >> 8) & 0xff; }
static inline void rtlamd_put32(unsigned char *p, uint32_t v) { rtlamd_put16(p, v & 0xffff); rtlamd_put16(p + 2, v >> 16); }

static inline void rtlamd_wave_time(unsigned char *p)
{
	// Windows SYSTEMTIME in UTC, wavewrite.c:150-172
	struct timeval tv;
	struct tm t;
	gettimeofday(&tv, NULL);
	gmtime_r(&tv.tv_sec, &t);
	rtlamd_put16(p + 0, (uint32_t)t.tm_year + 1900); rtlamd_put16(p + 2, (uint32_t)t.tm_mon + 1);
	rtlamd_put16(p + 4, (uint32_t)t.tm_wday); rtlamd_put16(p + 6, (uint32_t)t.tm_mday);
	rtlamd_put16(p + 8, (uint32_t)t.tm_hour); rtlamd_put16(p + 10, (uint32_t)t.tm_min);
	rtlamd_put16(p + 12, (uint32_t)t.tm_sec); rtlamd_put16(p + 14, (uint32_t)(tv.tv_usec / 1000));
}

// waveWriteHeader(), wavewrite.c:213-221 (+ wavePrepareHeader :174-211); no-op for stdout
static inline void rtlamd_wave_write_header(struct rtlamd_wave *w, unsigned samplerate, unsigned freq,
                                            int bits_per_sample, int channels, FILE *f)
{
	if (f == stdout) return;
	unsigned char *h = w->hdr;
	const int bytes_per_frame = bits_per_sample / 8 * channels;
	memset(h, 0, sizeof(w->hdr));
	memcpy(h + 0, "RIFF", 4); rtlamd_put32(h + 4, 120 - 8); memcpy(h + 8, "WAVE", 4);
	memcpy(h + 12, "fmt ", 4); rtlamd_put32(h + 16, 16);
	rtlamd_put16(h + 20, 1); rtlamd_put16(h + 22, (uint32_t)channels);
	rtlamd_put32(h + 24, samplerate); rtlamd_put32(h + 28, samplerate * (unsigned)bytes_per_frame);
	rtlamd_put16(h + 32, (uint32_t)channels); rtlamd_put16(h + 34, (uint32_t)bits_per_sample);
	memcpy(h + 36, "auxi", 4); rtlamd_put32(h + 40, 68);
	rtlamd_wave_time(h + 44);
	memcpy(h + 60, h + 44, 16);  // StopTime = StartTime until finalized
	rtlamd_put32(h + 76, freq); rtlamd_put32(h + 80, samplerate);
	memcpy(h + 112, "data", 4); rtlamd_put32(h + 116, 0);
	w->data_size = 0;
	w->started = 1;
	fwrite(h, sizeof(w->hdr), 1, f);
}

// waveFinalizeHeader(), wavewrite.c:223-235: data and RIFF sizes, stop time, rewrite in place
static inline void rtlamd_wave_finalize(struct rtlamd_wave *w, FILE *f)
{
	if (f == stdout || !w->started) return;
	rtlamd_wave_time(w->hdr + 60);
	rtlamd_put32(w->hdr + 116, w->data_size);
	rtlamd_put32(w->hdr + 4, 120 - 8 + w->data_size);
	fseek(f, 0, SEEK_SET);
	fwrite(w->hdr, sizeof(w->hdr), 1, f);
	w->started = 0;
}
