// rtl_fm_hip — an rtl_fm-shaped command line over the C ABI.
//
// Keeps the reference tool's structure (src/rtl_fm.c): getopt with the same
// option letters (:1721-1880), the `-M wbfm` preset (:1831-1840), rate planning
// through optimal_settings (:1407-1445) and deemph_a (:1929-1934), a dongle
// thread that sits in rtlsdr_read_async() and hands each buffer over from the
// callback (:1346-1351, :1274-1344), a demod thread (:1353-1391) and an output
// thread that fwrite()s int16 PCM (:1393-1405) — but the callback body and
// full_demod() are rtlfm_gpu_push() / rtlfm_gpu_run() / rtlfm_gpu_fetch().
// The device is whatever exports the rtlsdr_* API; here librtlsdr_file.so
// (RTLSDR_FILE=<raw u8 IQ file>).  Unlike the reference's condvar hand-off
// (:1339-1343) nothing is ever dropped: the callback waits while the queue is
// full, so the output is a deterministic function of the input file.
//
// Not restated (out of scope): frequency scanning / hopping,
// the CSV command file, squelch-driven retuning.  -l / -t hold the output back as
// demod_thread_fn does, -L prints full_demod()'s level lines.
#include <getopt.h>
#include <pthread.h>

#include <cerrno>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/rtlfm_hip.h"
#include "../../../include/rtlsdr_file.h"
#include "wavhdr.h"

namespace {

// atofs(): a number with an optional k / M / G suffix (src/convenience/convenience.c:67-96)
double atofs(const char *s)
{
	std::string t(s);
	double mul = 1.0;
	if (!t.empty()) {
		switch (t.back()) {
		case 'g': case 'G': mul = 1e9; t.pop_back(); break;
		case 'm': case 'M': mul = 1e6; t.pop_back(); break;
		case 'k': case 'K': mul = 1e3; t.pop_back(); break;
		default: break;
		}
	}
	return atof(t.c_str()) * mul;
}

struct Plumbing {
	std::mutex m;
	std::condition_variable cv_room, cv_work, cv_out;
	int queued = 0;          // buffers committed to the ring and not yet handed to rtlfm_gpu_run()
	bool slot_open = false;  // -Z: the device layer is writing into a slot of the ring (acquire ... commit)
	bool want_run = false;   // the demod thread is about to run: no new slot is handed out until it has
	bool eof = false, failed = false;
	std::deque<std::vector<int16_t>> out_q;
	bool out_done = false;
};

struct App {
	rtlfm_cfg cfg;
	rtlfm_gpu *gpu = nullptr;
	rtlsdr_dev_t *dev = nullptr;
	Plumbing p;
	FILE *file = nullptr;
	rtlamd_wave wave{};
	int verbosity = 0;
	int conseq_squelch = 10;
	// -L (src/rtl_fm.c:109-113, 1217-1237)
	int print_levels = 0, print_level_no = 1, level_max = 0, level_max_max = 0;
	double level_sum = 0.0;
	uint32_t user_freq = 0;
	uint64_t blocks_in = 0, samples_out = 0, blocks_squelched = 0;
	bool zero_copy = false;            // -Z: the device layer reads straight into the pinned staging ring
	unsigned char *open_slot = nullptr;  // the slot the device layer is filling (rtlfm_gpu_acquire)
};

// -Z: where the device layer reads its next buffer to (rtlamd_file_set_buffer_source): a slot of the GPU
// layer's pinned ring, the counterpart of the reference's zero-copy USB buffers (src/librtlsdr.c:2744-2810)
int next_slot(void *ctx, unsigned char **buf, uint32_t *cap)
{
	App *a = static_cast<App *>(ctx);
	// rtlfm_gpu_run() refuses to start while a slot is open (it takes every buffer whole or not at all), and
	// this thread asks for its next slot the moment a callback returns: the slot is handed out under the
	// plumbing's lock, and not while the demod thread has announced a run or the ring is full - otherwise
	// the demod thread would hardly ever find the ring closed and committed buffers would sit there.
	std::unique_lock<std::mutex> g(a->p.m);
	for (;;) {
		a->p.cv_room.wait(g, [&] { return (!a->p.want_run && a->p.queued < a->cfg.max_blocks) || a->p.failed; });
		if (a->p.failed) return -1;
		uint8_t *p = nullptr;
		int r = rtlfm_gpu_acquire(a->gpu, 0, &p, cap);
		if (r == 0) { a->open_slot = p; a->p.slot_open = true; *buf = p; return 0; }
		if (r != -ENOSPC) return r;  // not fatal: the device layer falls back to its own buffer and the callback pushes
		// the ring's filling half is full although `queued` says otherwise (a run that has not flipped the halves
		// yet): wait for the demod thread's next notification instead of spinning
		a->p.cv_room.wait_for(g, std::chrono::milliseconds(2));
	}
}

// the rtlsdr_read_async callback (reference rtlsdr_callback, src/rtl_fm.c:1274)
void on_buffer(unsigned char *buf, uint32_t len, void *ctx)
{
	App *a = static_cast<App *>(ctx);
	len -= len % 512;  // actual_length comes in whole USB packets; a file's last bytes may not
	if (a->open_slot && buf == a->open_slot) {
		// the samples are already where they belong: say how many (0 gives the slot back)
		a->open_slot = nullptr;
		int r = rtlfm_gpu_commit(a->gpu, 0, len);
		std::lock_guard<std::mutex> g(a->p.m);
		a->p.slot_open = false;
		if (r < 0) {
			fprintf(stderr, "rtlfm_gpu_commit: %s\n", rtlfm_gpu_strerror(r));
			a->p.failed = true;
			rtlsdr_cancel_async(a->dev);
		} else if (len) {
			a->p.queued++;
			a->blocks_in++;
		}
		a->p.cv_work.notify_all();
		return;
	}
	if (len == 0) return;
	// the copy is made under the plumbing's lock and not while the demod thread has announced a run: `queued`
	// is then exactly what the ring holds when rtlfm_gpu_run() takes it
	std::unique_lock<std::mutex> g(a->p.m);
	for (;;) {
		a->p.cv_room.wait(g, [&] { return (!a->p.want_run && a->p.queued < a->cfg.max_blocks) || a->p.failed; });
		if (a->p.failed) return;
		int r = rtlfm_gpu_push(a->gpu, 0, buf, len);
		if (r == 0) break;
		if (r != -ENOSPC) {
			fprintf(stderr, "rtlfm_gpu_push: %s\n", rtlfm_gpu_strerror(r));
			a->p.failed = true;
			rtlsdr_cancel_async(a->dev);  // the reference's error pattern, src/rtl_sdr.c:109-112
			a->p.cv_work.notify_all();
			return;
		}
		a->p.cv_room.wait_for(g, std::chrono::milliseconds(2));
	}
	a->p.queued++;
	a->blocks_in++;
	a->p.cv_work.notify_all();
}

void dongle_thread(App *a)
{
	if (a->zero_copy) rtlamd_file_set_buffer_source(a->dev, next_slot, a);
	rtlsdr_read_async(a->dev, on_buffer, a, 0, a->cfg.block_len);
	std::lock_guard<std::mutex> g(a->p.m);
	a->p.eof = true;
	a->p.cv_work.notify_all();
}

void demod_thread(App *a)
{
	const int cap = rtlfm_result_cap(&a->cfg) * a->cfg.max_blocks + 16;
	for (;;) {
		int taken = 0;
		{
			std::unique_lock<std::mutex> g(a->p.m);
			a->p.cv_work.wait(g, [&] { return a->p.queued > 0 || a->p.eof || a->p.failed; });
			if (a->p.failed || (a->p.queued == 0 && a->p.eof)) break;
			// -Z: no new slot from here on (next_slot), and the one that is open is committed first
			a->p.want_run = true;
			a->p.cv_work.wait(g, [&] { return !a->p.slot_open || a->p.failed; });
			if (a->p.failed) break;
			taken = a->p.queued;  // rtlfm_gpu_run takes everything that is queued
		}
		// The gate is held around the FLIP of the ring's halves only (rtlfm_gpu_run_begin): the transfer, a first run's
		// allocations and the kernel launches (rtlfm_gpu_run_end) happen with the device thread already filling the other
		// half - a callback never waits for a transfer or a kernel, as include/rtlfm_hip.h promises.
		int r = rtlfm_gpu_run_begin(a->gpu, &taken);
		{
			std::lock_guard<std::mutex> g(a->p.m);
			a->p.want_run = false;
			// the count goes down only for a run that has started: on -EAGAIN the buffers are still in the ring
			if (r == 0) a->p.queued -= taken;
			a->p.cv_room.notify_all();
		}
		if (r == -EAGAIN) { std::this_thread::yield(); continue; }
		if (r == 0) r = rtlfm_gpu_run_end(a->gpu);
		std::vector<int16_t> pcm((size_t)cap);
		int n = 0;
		if (r == 0) r = rtlfm_gpu_fetch(a->gpu, 0, pcm.data(), cap, &n);
		if (r < 0) {
			fprintf(stderr, "rtlfm_gpu_run/fetch: %s\n", rtlfm_gpu_strerror(r));
			std::lock_guard<std::mutex> g(a->p.m);
			a->p.failed = true;
			rtlsdr_cancel_async(a->dev);
			a->p.cv_room.notify_all();
			break;
		}
		pcm.resize((size_t)n);
		if (a->print_levels) {
			// full_demod()'s level printing, src/rtl_fm.c:1217-1237, on the rms() the GPU layer kept
			int32_t sr = 0;
			int nl = 0;
			if (rtlfm_gpu_levels(a->gpu, 0, &sr, 1, &nl) == 0 && nl == 1) {
				--a->print_level_no;
				if (sr >= 0) {
					a->level_sum += sr;
					if (a->level_max < sr) a->level_max = sr;
					if (a->level_max_max < sr) a->level_max_max = sr;
					if (!a->print_level_no) {
						a->print_level_no = a->print_levels;
						const double avg_rms = a->level_sum / a->print_levels;
						fprintf(stderr, "%.3f kHz, %.1f avg rms, %d max rms, %d max max rms, %d squelch rms, %d rms, %.1f dB rms level, %.2f dB avg rms level\n",
						        a->user_freq / 1000.0, avg_rms, a->level_max, a->level_max_max, a->cfg.squelch_level, (int)sr,
						        20.0 * log10(1E-10 + sr), 20.0 * log10(1E-10 + avg_rms));
						a->level_max = 0;
						a->level_sum = 0;
					}
				}
			}
		}
		if (a->cfg.squelch_level) {
			// demod_thread_fn(), src/rtl_fm.c:1366-1370: while the squelch has been closed for more than
			// conseq_squelch buffers nothing goes to the output thread, and the counter is held one above
			// the limit ("hair trigger").  squelch_hits starts at 11 (:1615): silence until it first opens.
			rtlfm_stream_state st;
			if (rtlfm_gpu_state_get(a->gpu, 0, &st) == 0 && st.squelch_hits > a->conseq_squelch) {
				st.squelch_hits = a->conseq_squelch + 1;
				rtlfm_gpu_state_set(a->gpu, 0, &st);
				a->blocks_squelched++;
				continue;
			}
		}
		std::lock_guard<std::mutex> g(a->p.m);
		a->p.out_q.push_back(std::move(pcm));
		a->p.cv_out.notify_one();
	}
	std::lock_guard<std::mutex> g(a->p.m);
	a->p.out_done = true;
	a->p.cv_out.notify_all();
}

void output_thread(App *a)
{
	for (;;) {
		std::vector<int16_t> pcm;
		{
			std::unique_lock<std::mutex> g(a->p.m);
			a->p.cv_out.wait(g, [&] { return !a->p.out_q.empty() || a->p.out_done; });
			if (a->p.out_q.empty()) break;
			pcm = std::move(a->p.out_q.front());
			a->p.out_q.pop_front();
		}
		fwrite(pcm.data(), 2, pcm.size(), a->file);  // src/rtl_fm.c:1400
		a->wave.data_size += 2 * (uint32_t)pcm.size();  // waveDataSize, :1401
		a->samples_out += pcm.size();
	}
	fflush(a->file);
}

void usage()
{
	fprintf(stderr,
	        "rtl_fm_hip, rtl_fm's demodulator on an AMD GPU (one stream; see rtlfm_hip.h for batches)\n"
	        "Use:\trtl_fm_hip -f freq [-options] [filename]   (input: RTLSDR_FILE=<raw u8 IQ file>)\n"
	        "\t-f frequency_to_tune_to [Hz]\n"
	        "\t[-M modulation (default: fm)]  fm, wbfm, raw, am, usb, lsb\n"
	        "\t[-s sample_rate (default: 24k)]  [-r resample_rate (default: none / same as -s)]\n"
	        "\t[-m minimum_capture_rate Hz (default: 1m)]\n"
	        "\t[-F fir_size (default: off)]  enables the fifth-order low pass; 0 or 9 (9 = droop compensation)\n"
	        "\t[-A std/fast/lut choose atan math (default: std)]\n"
	        "\t[-E enable_option]  edge, dc, rdc, deemp, offset\n"
	        "\t[-c de-emphasis_time_constant in us: us (75), eu (50) or a number]\n"
	        "\t[-o oversampling (default: 1)]  [-l squelch_level]  [-t squelch_delay (default: 10)]  [-q rdc_block_const]\n"
	        "\t[-L N  prints levels every N calculations]\n"
	        "\t[-W length of one buffer in units of 512 bytes (default: 32 = 16384 B)]\n"
	        "\t[-H write a wave header with the auxi chunk SDR programs read the frequency from]\n"
	        "\t[-Z zero-copy: the device layer reads straight into the GPU layer's pinned staging ring]\n"
	        "\t[-d device_index] [-g gain] [-p ppm]  accepted and passed to the device layer\n"
	        "\tfilename ('-' means stdout)\n");
	exit(1);
}

}  // namespace

int main(int argc, char **argv)
{
	App a;
	rtlfm_cfg_default(&a.cfg);
	rtlfm_cfg &c = a.cfg;
	int rate_in = 24000, min_capture = 1000000, time_constant = 75;
	int fifth = 0, edge = 0, dev_index = 0, gain = -100, ppm = 0;
	uint32_t freq = 0;
	bool have_freq = false, write_wav = false, wb_mode = false;
	int conseq_squelch = 10;  // demod_init(), src/rtl_fm.c:1613
	c.rate_out = 24000;
	c.max_blocks = 8;
	int opt;
	while ((opt = getopt(argc, argv, "d:f:g:s:l:o:t:r:p:E:F:A:M:hm:L:q:c:W:HvZ")) != -1) {
		switch (opt) {
		case 'd': dev_index = atoi(optarg); break;
		case 'f': freq = (uint32_t)atofs(optarg); have_freq = true; break;
		case 'g': gain = (int)(atof(optarg) * 10); break;
		case 'p': ppm = (int)atof(optarg); break;
		case 'm': min_capture = (int)atofs(optarg); break;
		case 'l': c.squelch_level = (int)atof(optarg); break;
		case 'L': a.print_levels = (int)atof(optarg); break;  // src/rtl_fm.c:1757-1759
		case 't':  // src/rtl_fm.c:1774-1781 (a negative value also asks to terminate on squelch: not restated)
			conseq_squelch = (int)atof(optarg);
			if (conseq_squelch < 0) conseq_squelch = -conseq_squelch;
			break;
		case 's': rate_in = (int)atofs(optarg); c.rate_out = rate_in; break;
		case 'r': c.rate_out2 = (int)atofs(optarg); break;
		case 'o': c.post_downsample = (int)atof(optarg); break;
		case 'q': c.rdc_block_const = atoi(optarg); break;
		case 'E':
			if (!strcmp(optarg, "edge")) edge = 1;
			if (!strcmp(optarg, "dc") || !strcmp(optarg, "adc")) c.dc_block_audio = 1;
			if (!strcmp(optarg, "rdc")) c.dc_block_raw = 1;
			if (!strcmp(optarg, "deemp")) c.deemph = 1;
			if (!strcmp(optarg, "offset")) c.offset_tuning = 1;
			break;
		case 'F': fifth = 1; c.comp_fir_size = atoi(optarg); break;
		case 'A':
			if (!strcmp(optarg, "std")) c.custom_atan = RTLFM_ATAN_STD;
			if (!strcmp(optarg, "fast")) c.custom_atan = RTLFM_ATAN_FAST;
			if (!strcmp(optarg, "lut")) c.custom_atan = RTLFM_ATAN_LUT;
			break;
		case 'M':
			if (!strcmp(optarg, "fm") || !strcmp(optarg, "nbfm") || !strcmp(optarg, "nfm")) c.mode = RTLFM_MODE_FM;
			if (!strcmp(optarg, "raw") || !strcmp(optarg, "iq")) c.mode = RTLFM_MODE_RAW;
			if (!strcmp(optarg, "am")) c.mode = RTLFM_MODE_AM;
			if (!strcmp(optarg, "usb")) c.mode = RTLFM_MODE_USB;
			if (!strcmp(optarg, "lsb")) c.mode = RTLFM_MODE_LSB;
			if (!strcmp(optarg, "wbfm") || !strcmp(optarg, "wfm")) {
				// the preset of src/rtl_fm.c:1831-1840
				c.mode = RTLFM_MODE_FM;
				rate_in = 170000; c.rate_out = 170000; c.rate_out2 = 32000;
				c.custom_atan = RTLFM_ATAN_FAST;
				c.deemph = 1;
				c.squelch_level = 0;
				wb_mode = true;
			}
			break;
		case 'c':
			if (!strcmp(optarg, "us")) time_constant = 75;
			else if (!strcmp(optarg, "eu")) time_constant = 50;
			else time_constant = (int)atof(optarg);
			break;
		case 'W': {
			long v = 512L * atoi(optarg);
			if (v > (long)RTLFM_MAX_BLOCK_LEN) v = RTLFM_MAX_BLOCK_LEN;  // src/rtl_fm.c:1869-1873
			c.block_len = (uint32_t)v;
			break;
		}
		case 'H': write_wav = true; break;
		case 'Z': a.zero_copy = true; break;
		case 'v': a.verbosity++; break;
		default: usage();
		}
	}
	if (!have_freq) { fprintf(stderr, "Please specify a frequency.\n"); return 1; }
	if (wb_mode) freq += 16000;  // controller_thread_fn(), src/rtl_fm.c:1455-1460: "wbfm: adding 16000 Hz to every input frequency"
	a.conseq_squelch = conseq_squelch;
	if (c.squelch_level) c.max_blocks = 1;  // the squelch rule below is the reference's per-buffer rule
	if (a.print_levels > 0) { c.report_levels = 1; c.max_blocks = 1; }
	a.user_freq = freq;  // dongle.userFreq, src/rtl_fm.c:1440
	rate_in *= c.post_downsample;  // src/rtl_fm.c:1886
	const char *filename = optind < argc ? argv[optind] : "-";

	if (rtlsdr_get_device_count() == 0) { fprintf(stderr, "No supported devices found (set RTLSDR_FILE).\n"); return 1; }
	if (rtlsdr_open(&a.dev, (uint32_t)dev_index) < 0) { fprintf(stderr, "Failed to open rtlsdr device #%d.\n", dev_index); return 1; }
	if (c.deemph) c.deemph_a = rtlfm_deemph_a(c.rate_out, time_constant);
	uint32_t capture_freq = 0, capture_rate = 0;
	rtlfm_optimal_settings(&c, freq, rate_in, min_capture, fifth, edge, &capture_freq, &capture_rate);
	if (gain == -100) rtlsdr_set_tuner_gain_mode(a.dev, 0);
	else { rtlsdr_set_tuner_gain_mode(a.dev, 1); rtlsdr_set_tuner_gain(a.dev, gain); }
	rtlsdr_set_freq_correction_ppb(a.dev, ppm * 1000);
	rtlsdr_set_offset_tuning(a.dev, c.offset_tuning);
	rtlsdr_set_center_freq(a.dev, capture_freq);
	if (rtlsdr_set_sample_rate(a.dev, capture_rate) < 0)
		fprintf(stderr, "WARNING: capture rate %u Hz is outside what an RTL2832 can do.\n", capture_rate);
	fprintf(stderr, "Tuned to %u Hz.\nOversampling input by: %ix.\nSampling at %u S/s.\nOutput at %u Hz.\n", capture_freq,
	        c.downsample, capture_rate, (unsigned)(c.rate_out2 > 0 ? c.rate_out2 : c.rate_out));
	if (a.verbosity)
		fprintf(stderr, "downsample_passes = %d, downsample = %d, deemph_a = %d, buffer = %u B\n", c.downsample_passes,
		        c.downsample, c.deemph_a, c.block_len);

	int r = rtlfm_gpu_create(&c, 1, 0, &a.gpu);
	if (r < 0) { fprintf(stderr, "rtlfm_gpu_create: %s\n", rtlfm_gpu_strerror(r)); return 2; }
	a.file = !strcmp(filename, "-") ? stdout : fopen(filename, "wb");
	if (!a.file) { fprintf(stderr, "Failed to open %s\n", filename); return 1; }
	if (write_wav && a.file != stdout)  // src/rtl_fm.c:1990-1995
		rtlamd_wave_write_header(&a.wave, (unsigned)(c.rate_out2 > 0 ? c.rate_out2 : c.rate_out), freq, 16,
		                         c.mode == RTLFM_MODE_RAW ? 2 : 1, a.file);
	rtlsdr_reset_buffer(a.dev);

	std::thread t_out(output_thread, &a), t_demod(demod_thread, &a), t_dongle(dongle_thread, &a);
	t_dongle.join();
	t_demod.join();
	t_out.join();
	if (a.file != stdout) {
		if (write_wav) rtlamd_wave_finalize(&a.wave, a.file);  // src/rtl_fm.c:2041-2045
		fclose(a.file);
	}
	fprintf(stderr, "%llu buffers in, %llu samples out, %llu buffers held back by the squelch%s\n", (unsigned long long)a.blocks_in,
	        (unsigned long long)a.samples_out, (unsigned long long)a.blocks_squelched, a.p.failed ? " (FAILED)" : "");
	rtlfm_gpu_destroy(a.gpu);
	rtlsdr_close(a.dev);
	return a.p.failed ? 3 : 0;
}
