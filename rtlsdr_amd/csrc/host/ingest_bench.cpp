// ingest_bench — the PCIe-inclusive rate of the callback boundary, measured from native threads.
//
// What a capture server does with S dongles (src/rtl_fm.c:1326-1343 per dongle thread): T threads
// call rtlfm_gpu_push() for their share of the streams — a memcpy out of a pageable buffer, as
// librtlsdr's transfer buffers are (src/librtlsdr.c:2697-2707) — while the previous run's H2D copy
// and kernels are in flight and the main thread collects its result with rtlfm_gpu_fetch_all().
// bench.py's `e2e` leg starts this program (the same loop from Python threads tops out on the
// interpreter lock at 8 threads).  Prints one JSON line.
//
//   ingest_bench <cfg file: the bytes of a rtlfm_cfg> <streams> <threads> <seconds> [device]
#include <atomic>
#include <barrier>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../../include/rtlfm_hip.h"

int main(int argc, char **argv)
{
	if (argc < 5) {
		fprintf(stderr, "usage: %s cfgfile streams threads seconds [device]\n", argv[0]);
		return 2;
	}
	rtlfm_cfg cfg;
	FILE *f = fopen(argv[1], "rb");
	if (!f || fread(&cfg, 1, sizeof cfg, f) != sizeof cfg) {
		fprintf(stderr, "ingest_bench: %s does not hold a rtlfm_cfg (%zu bytes)\n", argv[1], sizeof cfg);
		return 2;
	}
	fclose(f);
	const int S = atoi(argv[2]), T = atoi(argv[3]);
	const double seconds = atof(argv[4]);
	const int device = argc > 5 ? atoi(argv[5]) : 0;
	if (S < 1 || T < 1) return 2;
	cfg.max_blocks = 1;
	const uint32_t L = cfg.block_len;
	rtlfm_gpu *h = nullptr;
	int r = rtlfm_gpu_create(&cfg, S, device, &h);
	if (r < 0) { fprintf(stderr, "ingest_bench: rtlfm_gpu_create: %d\n", r); return 1; }
	// one transfer buffer per stream, pageable; an FM-ish byte pattern (the content does not matter here)
	std::vector<uint8_t> host((size_t)S * L);
	uint32_t x = 0x5D2000u;
	for (auto &b : host) { x = x * 1664525u + 1013904223u; b = (uint8_t)(96 + ((x >> 24) & 63)); }
	const int cap = rtlfm_result_cap(&cfg) + 16;
	std::vector<int16_t> out((size_t)S * cap);
	std::vector<int32_t> lens(S);
	std::atomic<int> err{0};
	std::atomic<bool> stop{false};
	std::barrier go(T + 1), done(T + 1);
	std::vector<std::thread> th;
	for (int t = 0; t < T; t++)
		th.emplace_back([&, t] {
			for (;;) {
				go.arrive_and_wait();
				if (stop.load()) return;
				for (int s = t; s < S; s += T) {
					const int e = rtlfm_gpu_push(h, s, host.data() + (size_t)s * L, L);
					if (e < 0) err.store(e);
				}
				done.arrive_and_wait();
			}
		});
	auto push_all = [&] { go.arrive_and_wait(); };
	auto pushed = [&] { done.arrive_and_wait(); };
	push_all(); pushed();
	if ((r = rtlfm_gpu_run(h)) < 0) { fprintf(stderr, "ingest_bench: rtlfm_gpu_run: %d\n", r); return 1; }
	push_all(); pushed();
	rtlfm_gpu_fetch_all(h, out.data(), (size_t)cap, lens.data());  // warm: ring, mirrors, clocks
	long runs = 0;
	const auto t0 = std::chrono::steady_clock::now();
	double dt = 0;
	for (;;) {
		if ((r = rtlfm_gpu_run(h)) < 0) break;   // run k in flight ...
		push_all();                                // ... the callbacks fill the other half ...
		r = rtlfm_gpu_fetch_all(h, out.data(), (size_t)cap, lens.data());  // ... and run k's audio comes back
		pushed();
		if (r < 0 || err.load() < 0) break;
		runs++;
		dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if (dt > seconds && runs >= 3) break;
	}
	stop.store(true);
	go.arrive_and_wait();
	for (auto &t : th) t.join();
	if (r < 0 || err.load() < 0) { fprintf(stderr, "ingest_bench: failed (%d / %d)\n", r, err.load()); return 1; }
	long total = 0;
	for (int s = 0; s < S; s++) total += lens[s];
	printf("{\"runs\": %ld, \"seconds\": %.3f, \"streams\": %d, \"threads\": %d, \"block_len\": %u, \"GB/s_in\": %.2f, "
	       "\"Msamples/s\": %.1f, \"pcm_per_run\": %ld}\n",
	       runs, dt, S, T, L, runs * (double)S * L / dt / 1e9, runs * (double)S * (L / 2) / dt / 1e6, total);
	rtlfm_gpu_destroy(h);
	return 0;
}
