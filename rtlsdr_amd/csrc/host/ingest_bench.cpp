// ingest_bench — the PCIe-inclusive rate of the callback boundary, measured from native threads,
// on one GPU or on several from ONE process ("one host thread + HIP stream per GPU").
//
// What a capture server does with S dongles per GPU (src/rtl_fm.c:1326-1343 per dongle thread): per
// device one handle, T producer threads and one run/fetch thread.  The producers hand their share of
// the streams' buffers over while the previous run's H2D copy and kernels are in flight and the
// run/fetch thread collects its audio with rtlfm_gpu_fetch_all():
//   --mode push     rtlfm_gpu_push(): a memcpy out of a pageable buffer, as librtlsdr's transfer
//                   buffers are (src/librtlsdr.c:2697-2707)
//   --mode acquire  rtlfm_gpu_acquire() / _commit(): the producer writes the pinned ring slot itself
//                   (the reference's zero-copy mode, src/librtlsdr.c:2744-2810); the "device" here
//                   writes a repeating 64 KiB pattern, i.e. one streaming write per byte and no read
//                   of a second buffer - what a receiving socket or a DMA engine would leave behind
// Threads of a device are pinned to the CPUs of the device's NUMA node (--pin 0 turns that off).
// bench.py's `e2e` leg starts this program.  Prints one JSON line: aggregate and per-device rates.
//
//   ingest_bench <cfg file: the bytes of a rtlfm_cfg> <streams per device> <threads per device> <seconds>
//                [--devices 0,1,...] [--mode push|acquire] [--pin 0|1] [--depth 1|2]
//   (a bare fifth argument is still taken as the one device to use)
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <barrier>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/rtlfm_hip.h"

namespace {

// CPUs of the NUMA node the device hangs on (rtlfm_gpu_device_numa_node); empty = unknown
std::vector<int> device_cpus(int device)
{
	std::vector<int> cpus;
	const int node = rtlfm_gpu_device_numa_node(device);
	if (node < 0) return cpus;
	char path[256];
	snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
	FILE *f = fopen(path, "r");
	if (!f) return cpus;
	char list[4096] = {0};
	if (!fgets(list, sizeof list, f)) list[0] = 0;
	fclose(f);
	for (char *tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
		int a, b;
		if (sscanf(tok, "%d-%d", &a, &b) == 2) { for (int c = a; c <= b; c++) cpus.push_back(c); }
		else if (sscanf(tok, "%d", &a) == 1) cpus.push_back(a);
	}
	return cpus;
}

void pin_to(const std::vector<int> &cpus)
{
	if (cpus.empty()) return;
	cpu_set_t set;
	CPU_ZERO(&set);
	for (int c : cpus) if (c < CPU_SETSIZE) CPU_SET(c, &set);
	pthread_setaffinity_np(pthread_self(), sizeof set, &set);
}

struct Device {
	int id = 0;
	rtlfm_gpu *h = nullptr;
	std::vector<int> cpus;
	long runs = 0;
	double seconds = 0;
	long pcm = 0;
	int err = 0;
};

}  // namespace

int main(int argc, char **argv)
{
	if (argc < 5) {
		fprintf(stderr, "usage: %s cfgfile streams threads seconds [--devices 0,1,..] [--mode push|acquire] [--pin 0|1]\n", argv[0]);
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
	std::vector<int> ids;
	bool acquire = false, pin = true;
	int depth = 2;  // 1: run / fetch / run; 2: run(k + 1) before the results of run k are collected (rtlfm_gpu_fetch_all_prev)
	for (int i = 5; i < argc; i++) {
		const std::string a = argv[i];
		if (a == "--devices" && i + 1 < argc) {
			char *list = argv[++i];
			for (char *tok = strtok(list, ","); tok; tok = strtok(nullptr, ",")) ids.push_back(atoi(tok));
		} else if (a == "--mode" && i + 1 < argc) {
			acquire = !strcmp(argv[++i], "acquire");
		} else if (a == "--pin" && i + 1 < argc) {
			pin = atoi(argv[++i]) != 0;
		} else if (a == "--depth" && i + 1 < argc) {
			depth = atoi(argv[++i]) >= 2 ? 2 : 1;
		} else if (i == 5 && a[0] != '-') {
			ids.push_back(atoi(argv[i]));
		}
	}
	if (ids.empty()) ids.push_back(0);
	if (S < 1 || T < 1) return 2;
	cfg.max_blocks = 1;
	const uint32_t L = cfg.block_len;
	const int cap = rtlfm_result_cap(&cfg) + 16;
	// one transfer buffer per stream, pageable; an FM-ish byte pattern (the content does not matter here)
	std::vector<uint8_t> host((size_t)S * L);
	uint32_t x = 0x5D2000u;
	for (auto &b : host) { x = x * 1664525u + 1013904223u; b = (uint8_t)(96 + ((x >> 24) & 63)); }
	std::vector<std::unique_ptr<Device>> devs;
	for (int id : ids) {
		auto d = std::make_unique<Device>();
		d->id = id;
		int r = rtlfm_gpu_create(&cfg, S, id, &d->h);
		if (r < 0) { fprintf(stderr, "ingest_bench: rtlfm_gpu_create(device %d): %d\n", id, r); return 1; }
		if (pin) d->cpus = device_cpus(id);
		devs.push_back(std::move(d));
	}
	std::barrier start((int)devs.size() + 1);
	std::vector<std::thread> drivers;
	for (auto &dp : devs) {
		Device *d = dp.get();
		drivers.emplace_back([&, d] {
			pin_to(d->cpus);
			rtlfm_gpu *h = d->h;
			std::vector<int16_t> out((size_t)S * cap);
			std::vector<int32_t> lens(S);
			std::atomic<int> err{0};
			std::atomic<bool> stop{false};
			std::barrier go(T + 1), done(T + 1);
			std::vector<std::thread> th;
			for (int t = 0; t < T; t++)
				th.emplace_back([&, t] {
					pin_to(d->cpus);
					for (;;) {
						go.arrive_and_wait();
						if (stop.load()) return;
						for (int s = t; s < S; s += T) {
							int e;
							if (acquire) {
								uint8_t *slot = nullptr;
								uint32_t room = 0;
								e = rtlfm_gpu_acquire(h, s, &slot, &room);
								if (e == 0) {
									// the producer writes the slot: a 64 KiB piece of the stream's pattern, repeated
									const uint8_t *src = host.data() + (size_t)s * L;
									for (uint32_t at = 0; at < L; at += 65536) memcpy(slot + at, src, L - at < 65536 ? L - at : 65536);
									e = rtlfm_gpu_commit(h, s, L);
								}
							} else {
								e = rtlfm_gpu_push(h, s, host.data() + (size_t)s * L, L);
							}
							if (e < 0) err.store(e);
						}
						done.arrive_and_wait();
					}
				});
			auto push_all = [&] { go.arrive_and_wait(); };
			auto pushed = [&] { done.arrive_and_wait(); };
			int r;
			push_all(); pushed();
			if ((r = rtlfm_gpu_run(h)) < 0) d->err = r;
			push_all(); pushed();
			rtlfm_gpu_fetch_all(h, out.data(), (size_t)cap, lens.data());  // warm: ring, mirrors, clocks
			start.arrive_and_wait();                                        // all devices begin together
			const auto t0 = std::chrono::steady_clock::now();
			if (depth == 2) {
				// two runs in flight: one more run so that there is a "run before the last"
				if ((r = rtlfm_gpu_run(h)) < 0) d->err = r;
				push_all(); pushed();
			}
			while (!d->err) {
				if ((r = rtlfm_gpu_run(h)) < 0) { d->err = r; break; }        // run k (depth 2: k + 1) in flight ...
				push_all();                                                    // ... the producers fill the other half ...
				r = depth == 2 ? rtlfm_gpu_fetch_all_prev(h, out.data(), (size_t)cap, lens.data())   // ... and the audio of run k
				               : rtlfm_gpu_fetch_all(h, out.data(), (size_t)cap, lens.data());       //     comes back meanwhile
				pushed();
				if (r < 0 || err.load() < 0) { d->err = r < 0 ? r : err.load(); break; }
				d->runs++;
				d->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
				if (d->seconds > seconds && d->runs >= 3) break;
			}
			stop.store(true);
			go.arrive_and_wait();
			for (auto &t : th) t.join();
			for (int s = 0; s < S; s++) d->pcm += lens[s];
		});
	}
	start.arrive_and_wait();
	for (auto &t : drivers) t.join();
	double agg_gbs = 0, agg_ms = 0, tmax = 0;
	long runs = 0, pcm = 0;
	std::string per = "[";
	for (auto &d : devs) {
		if (d->err < 0) { fprintf(stderr, "ingest_bench: device %d failed (%d)\n", d->id, d->err); return 1; }
		const double gbs = d->runs * (double)S * L / d->seconds / 1e9;
		agg_gbs += gbs;
		agg_ms += d->runs * (double)S * (L / 2) / d->seconds / 1e6;
		runs += d->runs;
		pcm += d->pcm;
		if (d->seconds > tmax) tmax = d->seconds;
		char one[160];
		snprintf(one, sizeof one, "%s{\"device\": %d, \"GB/s_in\": %.2f, \"runs\": %ld, \"pinned_cpus\": %zu}", per.size() > 1 ? ", " : "", d->id, gbs,
		         d->runs, d->cpus.size());
		per += one;
	}
	per += "]";
	printf("{\"runs\": %ld, \"seconds\": %.3f, \"streams\": %d, \"threads\": %d, \"block_len\": %u, \"devices\": %zu, \"mode\": \"%s\", "
	       "\"depth\": %d, \"GB/s_in\": %.2f, \"Msamples/s\": %.1f, \"pcm_per_run\": %ld, \"per_device\": %s}\n",
	       runs, tmax, S, T, L, devs.size(), acquire ? "acquire" : "push", depth, agg_gbs, agg_ms, devs.empty() ? 0 : pcm / (long)devs.size(), per.c_str());
	for (auto &d : devs) rtlfm_gpu_destroy(d->h);
	return 0;
}
