/*
 * ref_loader.c — opens oracle/_ref/*.so with RTLD_LAZY.
 *
 * TEST INFRASTRUCTURE ONLY.  Python's ctypes always adds RTLD_NOW, which
 * would try to bind the reference's unresolved rtlsdr_* function references
 * (librtlsdr/libusb are not in this image and no stand-ins are written for
 * them).  Lazy binding leaves them untouched because the DSP path never calls
 * them.  The returned handle is wrapped with ctypes.CDLL(None, handle=...).
 */
#include <dlfcn.h>
#include <stdio.h>

void *ref_loader_open(const char *path)
{
	void *h = dlopen(path, RTLD_LAZY | RTLD_LOCAL);
	if (!h)
		fprintf(stderr, "ref_loader_open(%s): %s\n", path, dlerror());
	return h;
}

int ref_loader_close(void *handle)
{
	return handle ? dlclose(handle) : -1;
}

void *ref_loader_sym(void *handle, const char *name)
{
	return dlsym(handle, name);
}

/* ------------------------------------------------------------------------
 * Multi-core timing of the REFERENCE hot path (bench.py's cpu_baseline leg).
 * The reference keeps exactly one stream in file-scope globals (and
 * deemph_filter's avg in a function static), so every thread opens its own
 * private copy of oracle/_ref/libref_rtlfm.so (a copy under another name is
 * a distinct object to the dynamic loader, hence distinct globals) and feeds
 * its own stream through the reference's rtlsdr_callback() + full_demod().
 * ---------------------------------------------------------------------- */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

struct ref_worker {
	void *handle;
	void (*reset)(void);
	int (*configure)(const void *cfg);
	int (*run_stream)(const uint8_t *, uint32_t, int, int16_t *);
	const void *cfg;
	const uint8_t *iq;
	uint32_t block_len;
	int nblocks, reps;
	long long produced;
	pthread_barrier_t *start;
};

static int copy_file(const char *src, const char *dst)
{
	FILE *a = fopen(src, "rb"), *b = fopen(dst, "wb");
	char buf[65536];
	size_t n;
	if (!a || !b) { if (a) fclose(a); if (b) fclose(b); return -1; }
	while ((n = fread(buf, 1, sizeof(buf), a)) > 0) fwrite(buf, 1, n, b);
	fclose(a); fclose(b);
	return 0;
}

static void *ref_worker_main(void *arg)
{
	struct ref_worker *w = (struct ref_worker *)arg;
	w->reset();
	w->configure(w->cfg);
	pthread_barrier_wait(w->start);
	for (int r = 0; r < w->reps; r++) {
		int n = w->run_stream(w->iq, w->block_len, w->nblocks, NULL);
		if (n < 0) { w->produced = n; return NULL; }
		w->produced += n;
	}
	return NULL;
}

/* nthreads streams (stream t at iq + t*stream_stride), each `nblocks` buffers,
 * repeated `reps` times with carried state.  Returns wall seconds of the timed
 * region (threads released together), or a negative value. */
double ref_bench_mt(const char *so_path, const void *cfg, const uint8_t *iq, size_t stream_stride,
                    uint32_t block_len, int nblocks, int nthreads, int reps)
{
	struct ref_worker *w = (struct ref_worker *)calloc((size_t)nthreads, sizeof(*w));
	pthread_t *tid = (pthread_t *)calloc((size_t)nthreads, sizeof(*tid));
	pthread_barrier_t start;
	char name[256];
	pthread_barrier_init(&start, NULL, (unsigned)nthreads + 1);
	for (int t = 0; t < nthreads; t++) {
		snprintf(name, sizeof(name), "/tmp/ref_bench_%d_%d.so", (int)getpid(), t);
		if (copy_file(so_path, name) != 0) return -1.0;
		w[t].handle = dlopen(name, RTLD_LAZY | RTLD_LOCAL);
		unlink(name);
		if (!w[t].handle) { fprintf(stderr, "ref_bench_mt: %s\n", dlerror()); return -2.0; }
		w[t].reset = (void (*)(void))dlsym(w[t].handle, "ref_reset");
		w[t].configure = (int (*)(const void *))dlsym(w[t].handle, "ref_configure");
		w[t].run_stream = (int (*)(const uint8_t *, uint32_t, int, int16_t *))dlsym(w[t].handle, "ref_run_stream");
		if (!w[t].reset || !w[t].configure || !w[t].run_stream) return -3.0;
		w[t].cfg = cfg; w[t].iq = iq + (size_t)t * stream_stride;
		w[t].block_len = block_len; w[t].nblocks = nblocks; w[t].reps = reps;
		w[t].start = &start;
		pthread_create(&tid[t], NULL, ref_worker_main, &w[t]);
	}
	struct timespec a, b;
	pthread_barrier_wait(&start);
	clock_gettime(CLOCK_MONOTONIC, &a);
	for (int t = 0; t < nthreads; t++) pthread_join(tid[t], NULL);
	clock_gettime(CLOCK_MONOTONIC, &b);
	double secs = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
	for (int t = 0; t < nthreads; t++) {
		if (w[t].produced < 0) secs = -4.0;
		dlclose(w[t].handle);
	}
	pthread_barrier_destroy(&start);
	free(w); free(tid);
	return secs;
}

/* ------------------------------------------------------------------------
 * The same for rtl_power: scanner() (src/rtl_power.c:642-720) keeps its tuning state, FFT buffer
 * and tables in file-scope globals, so every thread gets a private copy of
 * oracle/_ref/libref_rtlpower.so.  scanner() reads through rtlsdr_read_sync(): the device layer
 * (the product's file-backed librtlsdr_file.so, loaded RTLD_GLOBAL by the caller) serves one
 * looped capture from the page cache to every thread - a memcpy of buf_len bytes per 2^bin_e-point
 * transform.  RTLSDR_FILE / RTLSDR_FILE_LOOP are set by the caller before this is called.
 * ---------------------------------------------------------------------- */
struct refp_worker {
	void *handle;
	int (*setup)(const void *cfg);
	int (*scan_env)(int nscans);
	const void *cfg;
	int nscans, rc;
	pthread_barrier_t *start;
};

static void *refp_worker_main(void *arg)
{
	struct refp_worker *w = (struct refp_worker *)arg;
	w->setup(w->cfg);
	pthread_barrier_wait(w->start);
	w->rc = w->scan_env(w->nscans);
	return NULL;
}

double ref_power_bench_mt(const char *so_path, const void *cfg, int nthreads, int nscans)
{
	struct refp_worker *w = (struct refp_worker *)calloc((size_t)nthreads, sizeof(*w));
	pthread_t *tid = (pthread_t *)calloc((size_t)nthreads, sizeof(*tid));
	pthread_barrier_t start;
	char name[256];
	pthread_barrier_init(&start, NULL, (unsigned)nthreads + 1);
	for (int t = 0; t < nthreads; t++) {
		snprintf(name, sizeof(name), "/tmp/refp_bench_%d_%d.so", (int)getpid(), t);
		if (copy_file(so_path, name) != 0) return -1.0;
		w[t].handle = dlopen(name, RTLD_LAZY | RTLD_LOCAL);
		unlink(name);
		if (!w[t].handle) { fprintf(stderr, "ref_power_bench_mt: %s\n", dlerror()); return -2.0; }
		w[t].setup = (int (*)(const void *))dlsym(w[t].handle, "ref_power_setup");
		w[t].scan_env = (int (*)(int))dlsym(w[t].handle, "ref_power_scan_env");
		if (!w[t].setup || !w[t].scan_env) return -3.0;
		w[t].cfg = cfg; w[t].nscans = nscans; w[t].start = &start;
		pthread_create(&tid[t], NULL, refp_worker_main, &w[t]);
	}
	struct timespec a, b;
	pthread_barrier_wait(&start);
	clock_gettime(CLOCK_MONOTONIC, &a);
	for (int t = 0; t < nthreads; t++) pthread_join(tid[t], NULL);
	clock_gettime(CLOCK_MONOTONIC, &b);
	double secs = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
	for (int t = 0; t < nthreads; t++) {
		if (w[t].rc != 0) secs = -4.0;
		dlclose(w[t].handle);
	}
	pthread_barrier_destroy(&start);
	free(w); free(tid);
	return secs;
}
