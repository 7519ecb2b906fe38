// rtl_power_hip — an rtl_power-shaped command line over the C ABI.
//
// Same flow as the reference tool (src/rtl_power.c:767-1028): parse -f lower:upper:bin
// (:802-806), plan the hops with frequency_range (:438-540), then repeat scanner()
// (:642-720) — retune, rtlsdr_read_sync one buffer per hop, accumulate — and every
// `interval` seconds print one csv_dbm line per hop (:722-765, :992-1003).  The DSP of
// scanner() is rtlpower_gpu_scan(); every hop is one stream of the handle.
// Deterministic replay: RTLPOWER_PASSES=<n> reports after exactly n passes over the
// hops instead of by wall clock (the file device has no real-time pacing).
#include <getopt.h>
#include <cmath>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "../../../include/rtlpower_hip.h"
#include "../../../include/rtlsdr_file.h"

static double atofs(const std::string &s)
{
	// src/convenience/convenience.c:67-96
	std::string t(s);
	double mul = 1.0;
	if (!t.empty()) switch (t.back()) {
	case 'g': case 'G': mul = 1e9; t.pop_back(); break;
	case 'm': case 'M': mul = 1e6; t.pop_back(); break;
	case 'k': case 'K': mul = 1e3; t.pop_back(); break;
	default: break;
	}
	return mul * atof(t.c_str());
}
static double atoft(const std::string &s)
{
	// src/convenience/convenience.c:98-124
	std::string t(s);
	double mul = 1.0;
	if (!t.empty()) switch (t.back()) {
	case 'h': case 'H': mul = 3600; t.pop_back(); break;
	case 'm': case 'M': mul = 60; t.pop_back(); break;
	case 's': case 'S': mul = 1; t.pop_back(); break;
	default: break;
	}
	return mul * atof(t.c_str());
}
static double atofp(const std::string &s)
{
	// src/convenience/convenience.c:126-144
	if (!s.empty() && s.back() == '%') return 0.01 * atof(s.substr(0, s.size() - 1).c_str());
	return atof(s.c_str());
}

int main(int argc, char **argv)
{
	std::string freq_arg;
	int interval = 10, single = 0, window = RTLPOWER_WIN_RECTANGLE, boxcar = 1, comp_fir = 0, peak_hold = 0;
	int dev_index = 0;
	double crop = 0.0;
	time_t exit_after = 0;
	int opt;
	while ((opt = getopt(argc, argv, "f:i:s:t:d:g:p:e:w:c:F:1POhTD:")) != -1) {
		switch (opt) {
		case 'f': freq_arg = optarg; break;
		case 'd': dev_index = atoi(optarg); break;
		case 'c': crop = atofp(optarg); break;
		case 'i': interval = (int)round(atoft(optarg)); break;
		case 'e': exit_after = (time_t)((int)round(atoft(optarg))); break;
		case 'w':
			if (!strcmp(optarg, "rectangle")) window = RTLPOWER_WIN_RECTANGLE;
			if (!strcmp(optarg, "hamming")) window = RTLPOWER_WIN_HAMMING;
			if (!strcmp(optarg, "blackman")) window = RTLPOWER_WIN_BLACKMAN;
			if (!strcmp(optarg, "blackman-harris")) window = RTLPOWER_WIN_BLACKMAN_HARRIS;
			if (!strcmp(optarg, "hann-poisson")) window = RTLPOWER_WIN_HANN_POISSON;
			if (!strcmp(optarg, "youssef")) window = RTLPOWER_WIN_YOUSSEF;
			if (!strcmp(optarg, "kaiser")) window = RTLPOWER_WIN_KAISER;
			if (!strcmp(optarg, "bartlett")) window = RTLPOWER_WIN_BARTLETT;
			break;
		case 'F': boxcar = 0; comp_fir = atoi(optarg); break;  // src/rtl_power.c:866-869
		case 'P': peak_hold = 1; break;
		case '1': single = 1; break;
		case 'g': case 'p': case 's': case 't': case 'O': case 'T': case 'D': break;  // device-side knobs
		default:
			fprintf(stderr, "rtl_power_hip -f lower:upper:bin_size [-i interval] [-1] [-c crop] [-w window] [-F 0|9] [-P] [file]\n");
			return 1;
		}
	}
	size_t c1 = freq_arg.find(':'), c2 = freq_arg.rfind(':');
	if (freq_arg.empty() || c1 == std::string::npos || c2 == c1) { fprintf(stderr, "No frequency range provided.\n"); return 1; }
	if (crop < 0.0 || crop > 1.0) { fprintf(stderr, "Crop value outside of 0 to 1.\n"); return 1; }
	rtlpower_plan plan;
	int r = rtlpower_frequency_range((int)atofs(freq_arg.substr(0, c1)), (int)atofs(freq_arg.substr(c1 + 1, c2 - c1 - 1)),
	                                 (int)atofs(freq_arg.substr(c2 + 1)), crop, boxcar, &plan);
	if (r < 0 || plan.tune_count == 0) { fprintf(stderr, "Error: bandwidth too wide.\n"); return 1; }
	const int bins = 1 << plan.bin_e;
	fprintf(stderr, "Number of frequency hops: %i\nDongle bandwidth: %iHz\nDownsampling by: %ix\nCropping by: %0.2f%%\n"
	        "Total FFT bins: %i\nLogged FFT bins: %i\nFFT bin size: %0.2fHz\nBuffer size: %i bytes (%0.2fms)\n",
	        plan.tune_count, plan.rate, plan.downsample, plan.crop * 100, plan.tune_count * bins,
	        (int)((double)(plan.tune_count * bins) * (1.0 - plan.crop)), plan.bin_size, plan.buf_len,
	        1000 * 0.5 * (float)plan.buf_len / (float)plan.rate);
	if (interval < 1) interval = 1;
	fprintf(stderr, "Reporting every %i seconds\n", interval);
	const char *filename = optind < argc ? argv[optind] : "-";

	rtlsdr_dev_t *dev = nullptr;
	if (rtlsdr_get_device_count() == 0 || rtlsdr_open(&dev, (uint32_t)dev_index) < 0) {
		fprintf(stderr, "Failed to open rtlsdr device #%d (set RTLSDR_FILE).\n", dev_index);
		return 1;
	}
	rtlpower_cfg cfg;
	rtlpower_plan_cfg(&plan, window, boxcar, comp_fir, peak_hold, &cfg);
	rtlpower_gpu *gpu = nullptr;
	r = rtlpower_gpu_create(&cfg, plan.tune_count, 0, &gpu);
	if (r < 0) { fprintf(stderr, "rtlpower_gpu_create: %d\n", r); return 2; }
	FILE *file = !strcmp(filename, "-") ? stdout : fopen(filename, "wb");
	if (!file) { fprintf(stderr, "Failed to open %s\n", filename); return 1; }
	rtlsdr_reset_buffer(dev);
	rtlsdr_set_sample_rate(dev, (uint32_t)plan.rate);

	const char *pe = getenv("RTLPOWER_PASSES");
	const int passes_per_report = pe ? atoi(pe) : 0;
	std::vector<uint8_t> buf((size_t)plan.buf_len);
	std::vector<int64_t> avg((size_t)bins);
	std::vector<char> line((size_t)bins * 16 + 256);
	time_t next_tick = time(nullptr) + interval;
	if (exit_after) exit_after += time(nullptr);
	bool stop = false;
	int passes = 0;
	while (!stop) {
		// scanner(): one read per hop (src/rtl_power.c:650-719)
		for (int i = 0; i < plan.tune_count && !stop; i++) {
			const int f = rtlpower_tune_freq(&plan, i);
			if ((int)rtlsdr_get_center_freq(dev) != f) {
				rtlsdr_set_center_freq(dev, (uint32_t)f);  // retune(), :542-552 (the settling dump is a hardware matter)
			}
			int n_read = 0;
			rtlsdr_read_sync(dev, buf.data(), plan.buf_len, &n_read);
			if (n_read != plan.buf_len) { fprintf(stderr, "Error: dropped samples.\n"); stop = true; break; }
			r = rtlpower_gpu_scan(gpu, i, buf.data(), (uint32_t)plan.buf_len);
			if (r < 0) { fprintf(stderr, "rtlpower_gpu_scan: %d\n", r); stop = true; }
		}
		passes++;
		const time_t now = time(nullptr);
		const bool report = passes_per_report ? (passes % passes_per_report == 0) : (now >= next_tick);
		if (!report && !stop) continue;
		if (stop && passes_per_report) break;
		char t_str[50];
		strftime(t_str, sizeof(t_str), "%Y-%m-%d, %H:%M:%S", localtime(&now));
		for (int i = 0; i < plan.tune_count; i++) {
			int32_t samples = 0;
			rtlpower_gpu_fetch(gpu, i, avg.data(), &samples);
			if (samples == 0) continue;
			if (rtlpower_csv_dbm(&plan, i, avg.data(), samples, line.data(), line.size()) > 0)
				fprintf(file, "%s, %s", t_str, line.data());
		}
		fflush(file);
		rtlpower_gpu_clear(gpu);  // csv_dbm zeroes the accumulators, :761-764
		while (time(nullptr) >= next_tick) next_tick += interval;
		if (single) stop = true;
		if (exit_after && time(nullptr) >= exit_after) stop = true;
	}
	if (file != stdout) fclose(file);
	rtlpower_gpu_destroy(gpu);
	rtlsdr_close(dev);
	return 0;
}
