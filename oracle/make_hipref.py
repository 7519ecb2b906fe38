#!/usr/bin/env python3
"""make_hipref.py — the reference's OWN rtl_fm / rtl_power on the HIP layer.

TEST INFRASTRUCTURE (oracle/): executes the drop-in claim of INTEGRATION.md §1 instead of describing
it.  A temporary copy of /root/reference/src/rtl_fm.c (rtl_power.c) gets INTEGRATION.md's edits
applied BY LINE NUMBER — everything this script inserts is this repository's own glue code; no
reference text is stored here — and is compiled against the product's C ABI
(rtlsdr_amd/csrc/librtlfm_hip.so) and the file-backed device layer (librtlsdr_file.so) into

    oracle/_ref/rtl_fm_hipref      oracle/_ref/rtl_power_hipref

(git-ignored, travel to the GPU box like the other _ref binaries; the temporary copy is deleted).
tests/test_hipref_gpu.py runs BASELINE config 0's, C2's and `-M wbfm`'s command lines through
rtl_fm_hipref and compares the PCM with the oracle.

The edit sites (reference old-dab/rtlsdr, src/rtl_fm.c):
  :1273        glue inserted between full_demod() and rtlsdr_callback()
  :1325-1342   rtlsdr_callback: u8->int16 convert, dc_block_raw, rotate, memcpy into demod.lowpassed
               -> rtlfm_gpu_push(gpu, 0, buf, len)
  :1361        demod_thread_fn: full_demod(d) -> rtlfm_gpu_run + rtlfm_gpu_fetch into d->result
  :1387, :1398, :1402   (replay determinism only, RTLFM_HIPREF_LOSSLESS=1) the demod -> output hand-off with a
               predicate: the output thread writes every buffer exactly once
  :1471        controller thread, after optimal_settings(): rtlfm_gpu_create from the demod_state fields
src/rtl_power.c:
  :641         glue inserted in front of scanner()
  :660-718     scanner(): everything behind rtlsdr_read_sync() -> rtlpower_gpu_scan(h, tune, buf8, buf_len)
  :996         main(): before the csv_dbm() loop, bring avg[] / samples of every tune back
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("REF", "/root/reference")
OUT = os.path.join(HERE, "_ref")
CSRC = os.path.join(ROOT, "rtlsdr_amd", "csrc")
HOST = os.path.join(CSRC, "host")

FM_GLUE = r'''
/* ---- inserted by oracle/make_hipref.py: the HIP layer (INTEGRATION.md section 1) ---- */
#include <errno.h>
#include "rtlfm_hip.h"
static rtlfm_gpu *volatile hip_gpu;
static int hip_lossless = -1;     /* RTLFM_HIPREF_LOSSLESS=1: wait instead of dropping (replay from a file) */
static int hip_hits_seen;
static volatile unsigned hip_pushed, hip_taken;   /* replay only: buffers queued by the callback / taken by the demod thread */

static int hip_is_lossless(void)
{
	if (hip_lossless < 0) {
		const char *e = getenv("RTLFM_HIPREF_LOSSLESS");
		hip_lossless = (e && atoi(e)) ? 1 : 0;
	}
	return hip_lossless;
}

/* (a) demod_state / dongle_state -> rtlfm_cfg, once optimal_settings() and deemph_a are known */
static int hip_setup(void)
{
	rtlfm_cfg c;
	rtlfm_gpu *g = NULL;
	int r;
	if (hip_gpu)
		return 0;
	rtlfm_cfg_default(&c);
	c.mode = demod.mode_demod == &fm_demod ? RTLFM_MODE_FM :
	         demod.mode_demod == &am_demod ? RTLFM_MODE_AM :
	         demod.mode_demod == &usb_demod ? RTLFM_MODE_USB :
	         demod.mode_demod == &lsb_demod ? RTLFM_MODE_LSB : RTLFM_MODE_RAW;
	c.downsample = demod.downsample;            c.downsample_passes = demod.downsample_passes;
	c.comp_fir_size = demod.comp_fir_size;      c.custom_atan = demod.custom_atan;
	c.post_downsample = demod.post_downsample;  c.deemph = demod.deemph;
	c.deemph_a = demod.deemph_a;                c.rate_out = demod.rate_out;
	c.rate_out2 = demod.rate_out2;              c.dc_block_audio = demod.dc_block_audio;
	c.adc_block_const = demod.adc_block_const;  c.dc_block_raw = demod.dc_block_raw;
	c.rdc_block_const = demod.rdc_block_const;  c.offset_tuning = dongle.offset_tuning;
	c.output_scale = demod.output_scale;        c.squelch_level = demod.squelch_level;
	c.block_len = dongle.buf_len;               c.max_blocks = 1;
	r = rtlfm_gpu_create(&c, 1, 0, &g);
	if (r < 0) {
		fprintf(stderr, "rtlfm_gpu_create: %s\n", rtlfm_gpu_strerror(r));
		exit(1);
	}
	hip_hits_seen = demod.squelch_hits;
	hip_gpu = g;
	fprintf(stderr, "rtl_fm: demodulating on the HIP layer (librtlfm_hip %06x)\n", rtlfm_gpu_version());
	return 0;
}

/* (b) what the callback does with the driver's buffer */
static int hip_push(struct demod_state *d, unsigned char *buf, uint32_t len)
{
	int r;
	len -= len % 512;   /* a file's last bytes may not fill a USB packet; actual_length always does */
	while (!hip_gpu && hip_is_lossless() && !do_exit)
		usleep(1000);    /* replay: the controller thread is still creating the handle */
	if (!len || !hip_gpu)
		return 0;
	while ((r = rtlfm_gpu_push(hip_gpu, 0, buf, len)) == -ENOSPC && hip_is_lossless() && !do_exit) {
		/* replay: the demod thread has not taken the previous buffer yet (its wake-up may have been
		 * lost: safe_cond_wait has no predicate) */
		safe_cond_signal(&d->ready, &d->ready_m);
		usleep(100);
	}
	if (r == 0 && hip_is_lossless()) {
		/* replay: the wake-up of :1343 is lost when the demod thread is not waiting yet, and after the
		 * file's last buffer nobody would signal again: keep signalling until THIS buffer has been taken */
		const unsigned mine = ++hip_pushed;
		while (!do_exit && hip_taken < mine) {
			safe_cond_signal(&d->ready, &d->ready_m);
			usleep(50);
		}
	}
	if (r == -ENOSPC)
		return 0;        /* live: the reference overwrites the pending buffer (src/rtl_fm.c:1339-1342); here the newer one is dropped */
	if (r < 0) {
		fprintf(stderr, "rtlfm_gpu_push: %s\n", rtlfm_gpu_strerror(r));
		do_exit = 1;
		rtlsdr_cancel_async(dongle.dev);   /* the reference's own error pattern, src/rtl_sdr.c:109-112 */
	}
	return r;
}

/* (c) full_demod(d) */
static void hip_full_demod(struct demod_state *d)
{
	int n = 0, r;
	rtlfm_stream_state st;
	d->result_len = 0;
	if (!hip_gpu)
		return;
	if (d->squelch_level && d->squelch_hits != hip_hits_seen) {
		/* the demod thread clamped the counter (hair trigger, src/rtl_fm.c:1366-1368) */
		if (rtlfm_gpu_state_get(hip_gpu, 0, &st) == 0) {
			st.squelch_hits = d->squelch_hits;
			rtlfm_gpu_state_set(hip_gpu, 0, &st);
		}
	}
	r = rtlfm_gpu_run(hip_gpu);
	if (r == -EAGAIN)
		return;          /* woken without a queued buffer */
	if (r == 0) {
		hip_taken++;
		r = rtlfm_gpu_fetch(hip_gpu, 0, d->result, MAXIMUM_BUF_LENGTH, &n);
	}
	if (r < 0) {
		fprintf(stderr, "rtlfm_gpu_run/fetch: %s\n", rtlfm_gpu_strerror(r));
		d->exit_flag = 1;
		return;
	}
	d->result_len = n;
	if (d->squelch_level && rtlfm_gpu_state_get(hip_gpu, 0, &st) == 0)
		hip_hits_seen = d->squelch_hits = st.squelch_hits;
}

/* replay only: both hand-offs of the reference are lossy by design - the demod -> output one
 * overwrites o->result when the output thread is late (src/rtl_fm.c:1382-1387), and safe_cond_wait
 * has no predicate, so a wake-up can be lost or taken twice.  With RTLFM_HIPREF_LOSSLESS=1 the output
 * thread waits on a flag under the condition's own mutex and the demod thread waits until the buffer
 * has been written; without it both sites behave exactly as the reference. */
static int hip_out_pending;

static int hip_out_take(struct output_state *s)   /* in place of safe_cond_wait(&s->ready, ...) at :1398 */
{
	int took;
	if (!hip_is_lossless()) {
		safe_cond_wait(&s->ready, &s->ready_m);
		return 1;
	}
	pthread_mutex_lock(&s->ready_m);
	while (!hip_out_pending && !do_exit)
		pthread_cond_wait(&s->ready, &s->ready_m);
	took = hip_out_pending;
	pthread_mutex_unlock(&s->ready_m);
	return took;          /* 0: woken to exit, nothing new to write */
}

static void hip_out_done(struct output_state *s)   /* behind the fwrite, :1402 */
{
	pthread_mutex_lock(&s->ready_m);
	hip_out_pending = 0;
	pthread_mutex_unlock(&s->ready_m);
}

static void hip_out_post(struct output_state *o)   /* behind the demod thread's signal, :1387 */
{
	if (!hip_is_lossless())
		return;
	pthread_mutex_lock(&o->ready_m);
	hip_out_pending = 1;
	pthread_cond_signal(&o->ready);
	pthread_mutex_unlock(&o->ready_m);
	while (!do_exit) {
		int p;
		pthread_mutex_lock(&o->ready_m);
		p = hip_out_pending;
		pthread_mutex_unlock(&o->ready_m);
		if (!p)
			break;
		usleep(50);
	}
}
/* ---- end of inserted glue ---- */
'''

POWER_GLUE = r'''
/* ---- inserted by oracle/make_hipref.py: the HIP layer (include/rtlpower_hip.h) ---- */
#include <errno.h>
#include "rtlpower_hip.h"
static rtlpower_gpu *hip_pw;

/* -w lives in a local of main() (src/rtl_power.c:795); what scanner() sees of it is the global
 * window_coefs[] table main() filled (:985-988): find the window that produces it */
static int hip_power_window(void)
{
	int id, j, length = 1 << tunes[0].bin_e;
	int32_t *tmp = malloc(sizeof(int32_t) * (size_t)length);
	for (id = 0; id <= RTLPOWER_WIN_BARTLETT; id++) {
		if (rtlpower_window_coefs(id, length, tmp) < 0)
			continue;
		for (j = 0; j < length && tmp[j] == window_coefs[j]; j++)
			;
		if (j == length)
			break;
	}
	free(tmp);
	if (id > RTLPOWER_WIN_BARTLETT) {
		fprintf(stderr, "rtl_power: window table not recognised\n");
		exit(1);
	}
	return id;
}

static void hip_power_setup(void)
{
	rtlpower_cfg c;
	int r;
	if (hip_pw)
		return;
	memset(&c, 0, sizeof(c));
	c.bin_e = tunes[0].bin_e;
	c.window = hip_power_window();
	c.downsample = tunes[0].downsample;
	c.downsample_passes = tunes[0].downsample_passes;
	c.boxcar = boxcar;
	c.comp_fir_size = comp_fir_size;
	c.peak_hold = peak_hold;
	c.buf_len = (uint32_t)tunes[0].buf_len;
	r = rtlpower_gpu_create(&c, tune_count, 0, &hip_pw);
	if (r < 0) {
		fprintf(stderr, "rtlpower_gpu_create failed: %d\n", r);
		exit(1);
	}
	fprintf(stderr, "rtl_power: scanning on the HIP layer\n");
}

/* one rtlsdr_read_sync() buffer of tune i: everything scanner() does with it */
static void hip_power_scan(int i, struct tuning_state *ts, int buf_len)
{
	int r;
	hip_power_setup();
	r = rtlpower_gpu_scan(hip_pw, i, ts->buf8, (uint32_t)buf_len);
	if (r < 0) {
		fprintf(stderr, "rtlpower_gpu_scan failed: %d\n", r);
		do_exit = 2;
	}
}

/* before csv_dbm(): the accumulators of every tune back into tunes[] (csv_dbm zeroes them) */
static void hip_power_collect(void)
{
	int i;
	if (!hip_pw)
		return;
	for (i = 0; i < tune_count; i++) {
		int32_t n = 0;
		if (sizeof(long) != sizeof(int64_t) ||
		    rtlpower_gpu_fetch(hip_pw, i, (int64_t *)tunes[i].avg, &n) < 0) {
			fprintf(stderr, "rtlpower_gpu_fetch failed\n");
			do_exit = 2;
			return;
		}
		tunes[i].samples = n;
	}
	rtlpower_gpu_clear(hip_pw);
}
/* ---- end of inserted glue ---- */
'''


def expect(lines, no, *tokens):
    """Guard: the line we are about to edit still carries the identifiers this script relies on
    (another revision of the reference would silently mis-patch otherwise)."""
    text = lines[no - 1]
    for t in tokens:
        if t not in text:
            raise SystemExit(f"make_hipref: line {no} does not mention {t!r}: reference revision differs, edit sites must be re-derived")


def patch_rtl_fm(src: str) -> str:
    lines = src.split("\n")
    expect(lines, 1272, "}")
    expect(lines, 1274, "rtlsdr_callback")
    expect(lines, 1325, "convert")
    expect(lines, 1326, "for")
    expect(lines, 1331, "dc_block_raw_filter")
    expect(lines, 1333, "muteLen")
    expect(lines, 1339, "pthread_rwlock_wrlock")
    expect(lines, 1342, "pthread_rwlock_unlock")
    expect(lines, 1343, "safe_cond_signal")
    expect(lines, 1361, "full_demod")
    expect(lines, 1382, "OutputToStdout")
    expect(lines, 1387, "safe_cond_signal")
    expect(lines, 1398, "safe_cond_wait")
    expect(lines, 1400, "fwrite")
    expect(lines, 1402, "pthread_rwlock_unlock")
    expect(lines, 1471, "optimal_settings", "freqs")
    out = []
    for no, text in enumerate(lines, 1):
        if no == 1273:
            out.append(FM_GLUE)
        if 1325 <= no <= 1342:
            # the mute test of :1333-1334 stays; everything else of the range is the push
            if no == 1333 or no == 1334:
                out.append(text)
            if no == 1342:
                out.append("\tif (hip_push(d, buf, len) < 0)\n\t\treturn;")
            continue
        if no == 1361:
            out.append(text.replace("full_demod(d)", "hip_full_demod(d)"))
            continue
        if no == 1387:
            out.append(text)
            out.append("\t\t\thip_out_post(o);")
            continue
        if no == 1398:
            out.append("\t\tif (!hip_out_take(s))\n\t\t\tcontinue;")
            continue
        if no == 1402:
            out.append(text)
            out.append("\t\thip_out_done(s);")
            continue
        if no == 1471:
            out.append(text)
            out.append("\thip_setup();")
            continue
        out.append(text)
    return "\n".join(out)


def patch_rtl_power(src: str) -> str:
    lines = src.split("\n")
    expect(lines, 642, "scanner")
    expect(lines, 657, "rtlsdr_read_sync")
    expect(lines, 660, "rms")
    expect(lines, 717, "samples")
    expect(lines, 718, "}")
    expect(lines, 719, "}")
    expect(lines, 997, "tune_count")
    expect(lines, 999, "csv_dbm")
    out = []
    for no, text in enumerate(lines, 1):
        if no == 642:
            out.append(POWER_GLUE)
        if 660 <= no <= 718:
            if no == 660:
                out.append("\t\thip_power_scan(i, ts, buf_len);")
            continue
        if no == 997:
            out.append("\t\thip_power_collect();")
        out.append(text)
    return "\n".join(out)


def build(verbose: bool = False) -> list[str]:
    fm = os.path.join(REF, "src", "rtl_fm.c")
    if not os.path.exists(fm):
        print(f"make_hipref: {REF} not present; using prebuilt oracle/_ref/*_hipref if any", file=sys.stderr)
        return []
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="hipref_")
    made = []
    try:
        common = ["gcc", "-O3", "-w", f"-I{REF}/include", f"-I{REF}/src", f"-I{ROOT}/include",
                  f"-L{HOST}", f"-L{CSRC}", "-lrtlsdr_file", "-lrtlfm_hip", "-lm", "-lpthread",
                  "-Wl,-rpath,$ORIGIN/../../rtlsdr_amd/csrc/host", "-Wl,-rpath,$ORIGIN/../../rtlsdr_amd/csrc",
                  "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
        jobs = [("rtl_fm.c", patch_rtl_fm, "rtl_fm_hipref", [f"{REF}/src/convenience/convenience.c", f"{REF}/src/convenience/wavewrite.c"]),
                ("rtl_power.c", patch_rtl_power, "rtl_power_hipref", [f"{REF}/src/convenience/convenience.c"])]
        for name, patch, exe, extra in jobs:
            with open(os.path.join(REF, "src", name)) as fh:
                patched = patch(fh.read())
            tsrc = os.path.join(tmp, name)
            with open(tsrc, "w") as fh:
                fh.write(patched)
            cmd = common[:1] + common[1:6] + ["-o", os.path.join(OUT, exe), tsrc] + extra + common[6:]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
            made.append(os.path.join(OUT, exe))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)  # the patched copies never outlive the build
    return made


if __name__ == "__main__":
    for p in build(verbose="-v" in sys.argv):
        print(p)
