// power_kernels.h — rtl_power's scanner() DSP (reference src/rtl_power.c:642-720)
// on gfx950.
//
// A workgroup of 1024 threads walks a run of reads of one stream (one tuning_state).  The reads
// of a stream are independent of each other (every stage is stateless per read) and avg[] += /
// MAX is associative, so a stream's reads are split over `groups` workgroups — with many streams
// one group each, with the tool's real shape (one stream) enough groups to fill the GPU — and
// every group adds its int64 sums with one atomic per bin.  Per read, the whole decimated buffer (<= 16384 complex
// samples, packed int16 I,Q per dword = 64 KiB) lives in LDS next to the
// 3N/4-entry sine table (<= 24 KiB):
//   A. sums of I and Q for remove_dc (:581-596; the sum over N/2 values is divided
//      by N — only half of the DC goes, reproduced as is);
//   B. u8 -> int16 (-127) (:666-668), DC subtract, window multiply with int16 wrap
//      (:697-706), each point stored at its BIT-REVERSED index inside its FFT
//      chunk, which is fix_fft's reordering pass (:282-297) for free;
//   C. fix_fft's log2(N) radix-2 DIT stages (:298-326) — all chunks of the read at
//      once, one __syncthreads per stage, every int16 wrap and the FIX_MPY
//      rounding (:263-269) reproduced bit for bit;
//   D. |X|^2 into int64 accumulators held in registers across all reads
//      (thread t owns points t, t+1024, ...), flushed with one 64-bit atomic per
//      accumulator at the end (add, or max for peak hold, :708-716).
// This path is VALU/LDS bound, not HBM bound: ~250 integer ops per input sample.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtlpower_hip.h"
#include "dsp_device.h"

#ifndef RTLPOWER_WAVE_PRIO
#define RTLPOWER_WAVE_PRIO 1  // the four waves of a SIMD get four priorities (wave >> 2): -0.9 % on C4 (2.825 -> 2.800 ms, alternating); by wave & 3 (one priority per SIMD): +0.9 %
#endif
namespace rtlpower {

using rtlfm::iq16;
using rtlfm::pack_iq;
using rtlfm::unpack_iq;

constexpr int kThreads = 1024;
constexpr int kMaxPoints = 16384;

struct ScanParams {
	const uint8_t *iq8;      // raw reads (used when dec == nullptr)
	size_t stride8;          // bytes between streams
	const int16_t *dec;      // decimated int16 elements [stream][read][dec_elems], or nullptr
	size_t dec_stream_stride, dec_read_stride;  // in elements
	int dec_elems;           // valid elements per read in `dec` (beyond: zeros)
	int nreads;
	int buf_len;             // bytes per raw read
	int len_dec;             // buf_len / ds: what remove_dc and the chunk loop see
	int bin_e, chunks;       // FFT size exponent, chunks per read
	int ds, peak_hold;
	const int32_t *window;   // [N]
	const uint16_t *window16; // [N] the low halves of window[] (k_power_scan_big multiplies in 16 bits)
	const uint32_t *tw;      // [N] per-stage twiddles, see make_twiddles() in rtlpower_hip.hip
	long long *avg;          // [stream][N]
	int32_t *samples;        // [stream]
	int groups;              // workgroups per stream: group g takes reads [g, g + 1) * nreads / groups
	unsigned long long *stamps;  // rtlpower_gpu_clock_probe: [workgroups][4] shader clock first / last, 100 MHz counter first / last
	// k_power_scan_big<14, true> (frames of 2^(14 + comb_c) points, see "transforms that do not fit" below): iq8 = the
	// frames' bytes comb by comb, window16 = the window likewise, ave = remove_dc's averages per frame, work = where
	// the blocks go after stage 13
	int comb_c; size_t comb_blocks; const int2 *ave; uint32_t *work;
	// k_power_scan_frames<E, true>: decimated reads as packed (I, Q) int16 pairs, 2^dec_e points each, a stream's reads back
	// to back (k_power_downsample_iq / k_power_boxcar leave them so)
	const uint32_t *dec32; size_t dec32_stream_stride; int dec_e;
};

// FIX_MPY, src/rtl_power.c:263-269
__device__ __forceinline__ int fix_mpy(int a, int b)
{
	int c = (a * b) >> 14;
	return (int)(int16_t)((c >> 1) + (c & 1));
}

__device__ __forceinline__ int element_at(const ScanParams &p, const uint8_t *raw, const int16_t *dec, int e)
{
	if (dec) return e < p.dec_elems ? (int)dec[e] : 0;
	return e < p.buf_len ? (int)raw[e] - 127 : 0;
}

typedef short pk16_t __attribute__((ext_vector_type(2)));

// One butterfly of fix_fft (src/rtl_power.c:309-321) on packed (re, im) int16 pairs.
//   FIX_MPY(a, b) = ((a*b >> 14) >> 1) + ((a*b >> 14) & 1) = (a*b + 2^14) >> 15 = (a*2b + 2^15) >> 16  (exact)
//   tr = FIX_MPY(wr, b.re) - FIX_MPY(wi, b.im), ti = FIX_MPY(wr, b.im) + FIX_MPY(wi, b.re)
//   q  = a >> 1 (per component);  b' = q - t,  a' = q + t
// Every int16 store of the reference wraps mod 2^16; so do the packed 16-bit adds here.
// The twiddle table holds (2 wr, 2 wi) (both fit int16: wr, wi are the halved sines, -16384 .. 16383), so
// every FIX_MPY is the HIGH HALF of one v_mad_i32_i16 - op_sel takes the 16-bit halves straight out of the
// packed twiddle and the packed point, the rounding constant is the addend - and two v_perm gather the
// four high halves into the packed pairs (m1, m3) and (m2, m4) without a single shift; the one subtraction
// among tr, ti is a packed multiply-add by (-1, +1).  10 instructions per butterfly: 4 v_mad_i32_i16,
// 2 v_perm, v_pk_mad_i16, v_pk_ashr, v_pk_sub, v_pk_add (rounds 1-2: 18, round 3 before this: 14).
__device__ __forceinline__ int mad_i16(uint32_t x, uint32_t y, int c, int xhi, int yhi)
{
	int r;
	if (!xhi && !yhi) asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "=v"(r) : "v"(x), "v"(y), "s"(c));
	else if (xhi && yhi) asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,1,0,0]" : "=v"(r) : "v"(x), "v"(y), "s"(c));
	else if (yhi) asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[0,1,0,0]" : "=v"(r) : "v"(x), "v"(y), "s"(c));
	else asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(x), "v"(y), "s"(c));
	return r;
}
// KIND: 0 = any twiddle; 1 = the twiddle is real (wi == 0: FIX_MPY(0, x) == 0, two products drop out);
// 2 = it is imaginary (wr == 0).  The first three stages know which at compile time: positions 0 and N/4.
template <int KIND = 0>
__device__ __forceinline__ void butterfly(uint32_t &a, uint32_t &b, uint32_t w2)
{
	const pk16_t pm = {(short)-1, (short)1};  // t = (m1 - m2, m3 + m4) = (m2, m4) * (-1, +1) + (m1, m3): one v_pk_mad_i16
	pk16_t tv;
	if (KIND == 1) {
		const uint32_t p1 = (uint32_t)mad_i16(w2, b, 32768, 0, 0);  // high half: FIX_MPY(wr, b.re)
		const uint32_t p3 = (uint32_t)mad_i16(w2, b, 32768, 0, 1);  //            FIX_MPY(wr, b.im)
		tv = __builtin_bit_cast(pk16_t, __builtin_amdgcn_perm(p3, p1, 0x07060302u));
	} else {
		const uint32_t p2 = (uint32_t)mad_i16(w2, b, 32768, 1, 1);  // FIX_MPY(wi, b.im)
		const uint32_t p4 = (uint32_t)mad_i16(w2, b, 32768, 1, 0);  // FIX_MPY(wi, b.re)
		const pk16_t m24 = __builtin_bit_cast(pk16_t, __builtin_amdgcn_perm(p4, p2, 0x07060302u));
		if (KIND == 0) {
			const uint32_t p1 = (uint32_t)mad_i16(w2, b, 32768, 0, 0);  // FIX_MPY(wr, b.re)
			const uint32_t p3 = (uint32_t)mad_i16(w2, b, 32768, 0, 1);  // FIX_MPY(wr, b.im)
			const pk16_t m13 = __builtin_bit_cast(pk16_t, __builtin_amdgcn_perm(p3, p1, 0x07060302u));
			tv = m24 * pm + m13;
		} else {
			tv = m24 * pm;
		}
	}
	const pk16_t q = __builtin_bit_cast(pk16_t, a) >> 1;
	b = __builtin_bit_cast(uint32_t, (pk16_t)(q - tv));
	a = __builtin_bit_cast(uint32_t, (pk16_t)(q + tv));
}

// sum over the wave, valid in lane 63 (inclusive DPP scan)
__device__ __forceinline__ int wave_total(int x)
{
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
	x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
	x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
	return x;
}

// real_conj (src/rtl_power.c:636-640): re^2 + im^2 of a packed int16 pair is at most 2 * 2^30 = 2^31, which
// fits 32 bits unsigned: one v_dot2_i32_i16 of the pair with itself instead of two 64-bit multiplies
__device__ __forceinline__ long long power_of(uint32_t v)
{
	const pk16_t x = __builtin_bit_cast(pk16_t, v);
	return (long long)(uint32_t)__builtin_amdgcn_sdot2(x, x, 0, false);
}

// LDS holds the points skewed by 4 dwords per 32-dword block.  Every access pattern of the transform is
// then bank-conflict free: the stride-8 pass (stages 3-5) puts 32 lanes on 32 banks instead of 8 (what
// the skew was introduced for), and the first pass (stages 0-2), where every lane owns eight CONSECUTIVE
// dwords - two 16-byte accesses at a lane stride of 32 bytes -, no longer folds lanes l and l + 4 onto the
// same four banks (with the 8-per-64 skew of rounds 1-2 that pass ran two-way conflicted both ways:
// a quarter of the LDS time of the whole kernel, SQ_LDS_BANK_CONFLICT).
__device__ __forceinline__ int skew(int a) { return a + ((a >> 5) << 2); }
__host__ __device__ constexpr int skewed_size(int n) { return n + ((n >> 5) << 2) + 8; }
// skew(a + b) == skew(a) + skew(b) whenever (a mod 32) + (b mod 32) < 32 - true for a group's base index
// and its k-th point at distance k h for every stage width the passes use (h = 1, 8, or a multiple of 64)
__host__ __device__ constexpr int skew_c(int a) { return a + ((a >> 5) << 2); }
__host__ __device__ constexpr unsigned brev_c(unsigned v)
{
	unsigned r = 0;
	for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i);
	return r;
}

// stages st .. st+R-1 for every group of 2^R points at stride 2^st (st is a multiple
// of 3).  tw[(1 << stage) - 1 + m] holds the stage's twiddle for butterfly position m
// as packed int16 (2 wr, 2 wi) of the halved values fix_fft uses (src/rtl_power.c:303-308), see butterfly():
// consecutive lanes read consecutive dwords (or the same one), never a strided table.
// ST >= 0: the stage is known at compile time (k_power_scan_big) and every point offset is an immediate.
template <int R, int ST = -1>
__device__ __forceinline__ int fft_group_regs(uint32_t *pts, const uint32_t *tw, int st_rt, int g, uint32_t (&x)[1 << R], int (&off)[1 << R])
{
	constexpr int G = 1 << R;
	const int st = ST >= 0 ? ST : st_rt;
	const int h = 1 << st;
	const int glo = g & (h - 1), ghi = g >> st;
	const int base = skew((ghi << (st + R)) | glo);
#pragma unroll
	for (int k = 0; k < G; k++) off[k] = ST >= 0 ? skew_c(k << (ST >= 0 ? ST : 0)) : skew(k << st);
#pragma unroll
	for (int k = 0; k < G; k++) x[k] = pts[base + off[k]];
#pragma unroll
	for (int r = 0; r < R; r++) {
		const uint32_t *tws = tw + ((1 << (st + r)) - 1) + glo;
#pragma unroll
		for (int k = 0; k < G; k++) {
			if (k & (1 << r)) continue;
			const int kk = k & ((1 << r) - 1);
			const uint32_t w = tws[kk * h];  // position m = glo + (k mod 2^r) * h of the stage's 2^(st+r)
			// stages 0-2: position 0 is the real twiddle (16383, 0), the middle one (0, -16384)
			if (ST == 0 && kk == 0) butterfly<1>(x[k], x[k + (1 << r)], w);
			else if (ST == 0 && r > 0 && kk == (1 << r) / 2) butterfly<2>(x[k], x[k + (1 << r)], w);
			else butterfly<0>(x[k], x[k + (1 << r)], w);
		}
	}
	return base;
}
template <int R, int ST>
__device__ __forceinline__ void fft_group_regs(uint32_t *pts, const uint32_t *tw, int g, uint32_t (&x)[1 << R])
{
	int off[1 << R];
	fft_group_regs<R, ST>(pts, tw, ST, g, x, off);
}

template <int R, int ST = -1>
__device__ __forceinline__ void fft_group(uint32_t *pts, const uint32_t *tw, int st_rt, int g)
{
	constexpr int G = 1 << R;
	uint32_t x[G];
	int off[G];
	const int base = fft_group_regs<R, ST>(pts, tw, st_rt, g, x, off);
#pragma unroll
	for (int k = 0; k < G; k++) pts[base + off[k]] = x[k];
}

template <int R, int ST = -1>
__device__ __forceinline__ void fft_pass(uint32_t *pts, const uint32_t *tw, int M, int st_rt, int t)
{
	// (unrolling k_power_scan_big's two or four rounds so that one round's LDS reads travel under the other's
	// butterflies costs 9 more registers and measured 2 % SLOWER: four waves per SIMD already overlap them)
	for (int g = t; g < (M >> R); g += kThreads) fft_group<R, ST>(pts, tw, st_rt, g);
}

// The LDS traffic of one wave is in order: what its lanes wrote is there for its lanes' later reads.
__device__ __forceinline__ void wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ void __launch_bounds__(kThreads) k_power_scan(const ScanParams p)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
	const int N = 1 << p.bin_e;
	const int M = p.chunks * N;
	uint32_t *pts = sm;                       // [skewed_size(M)]
	uint32_t *tw = sm + skewed_size(M);       // [N]
	__shared__ long long red[2][kThreads / 64];
	__shared__ int ave[2];
	const int t = threadIdx.x;
	const size_t s = blockIdx.x / p.groups;
	const int grp = (int)(blockIdx.x % p.groups);
	const int r_begin = (int)((long long)grp * p.nreads / p.groups), r_end = (int)((long long)(grp + 1) * p.nreads / p.groups);
	for (int k = t; k < N; k += kThreads) tw[k] = p.tw[k];
	const int A = N >= kThreads ? N / kThreads : 1;  // accumulators per thread
	long long acc[16];
#pragma unroll
	for (int k = 0; k < 16; k++) acc[k] = 0;

	for (int r = r_begin; r < r_end; r++) {
		const uint8_t *raw = p.iq8 ? p.iq8 + s * p.stride8 + (size_t)r * p.buf_len : nullptr;
		const int16_t *dec = p.dec ? p.dec + s * p.dec_stream_stride + (size_t)r * p.dec_read_stride : nullptr;
		// ---- A: remove_dc sums over the elements below len_dec --------------------
		long long si = 0, sq = 0;
		for (int pnt = t; 2 * pnt < p.len_dec; pnt += kThreads) {
			si += element_at(p, raw, dec, 2 * pnt);
			if (2 * pnt + 1 < p.len_dec) sq += element_at(p, raw, dec, 2 * pnt + 1);
		}
		for (int off = 32; off > 0; off >>= 1) {
			si += __shfl_down(si, off);
			sq += __shfl_down(sq, off);
		}
		__syncthreads();  // previous read's phase D is done with pts / red
		if ((t & 63) == 0) { red[0][t >> 6] = si; red[1][t >> 6] = sq; }
		__syncthreads();
		if (t == 0) {
			long long a = 0, b = 0;
			for (int k = 0; k < kThreads / 64; k++) { a += red[0][k]; b += red[1][k]; }
			ave[0] = (int)(int16_t)(a / (long long)p.len_dec);
			ave[1] = p.len_dec > 1 ? (int)(int16_t)(b / (long long)(p.len_dec - 1)) : 0;
		}
		__syncthreads();
		const int ai = ave[0], aq = ave[1];
		// ---- B: convert, DC, window, bit-reversed placement -----------------------
		for (int pnt = t; pnt < M; pnt += kThreads) {
			const int c = pnt >> p.bin_e, j = pnt & (N - 1);
			int vi = element_at(p, raw, dec, 2 * pnt), vq = element_at(p, raw, dec, 2 * pnt + 1);
			if (2 * pnt < p.len_dec) vi = (int16_t)(vi - ai);
			if (2 * pnt + 1 < p.len_dec) vq = (int16_t)(vq - aq);
			const int w = p.window[j];
			vi = (int16_t)(vi * w);
			vq = (int16_t)(vq * w);
			const int rj = (int)(__brev((unsigned)j) >> (32 - p.bin_e));
			pts[skew((c << p.bin_e) + rj)] = pack_iq(vi, vq);
		}
		__syncthreads();
		// ---- C: radix-2 DIT stages, up to three per LDS round trip ---------------------
		// A group of 2^R points at stride 2^s stays in registers for stages s..s+R-1;
		// each butterfly is the reference's, bit for bit (see butterfly()).
		for (int st = 0; st < p.bin_e;) {
			const int R = p.bin_e - st >= 3 ? 3 : p.bin_e - st;
			if (R == 3) fft_pass<3>(pts, tw, M, st, t);
			else if (R == 2) fft_pass<2>(pts, tw, M, st, t);
			else fft_pass<1>(pts, tw, M, st, t);
			st += R;
			__syncthreads();
		}
		// ---- D: integrate / peak hold ------------------------------------------------
		auto fold = [&](long long &a, int pnt) {
			const long long pw = power_of(pts[skew(pnt)]);
			a = p.peak_hold ? (pw > a ? pw : a) : a + pw;
		};
		if (N >= kThreads) {
			// accumulator a of thread t is bin t + 1024*a of every chunk
#pragma unroll
			for (int a = 0; a < 16; a++)
				if (a < A)
					for (int c = 0; c < p.chunks; c++) fold(acc[a], (c << p.bin_e) + t + kThreads * a);
		} else {
			// N divides 1024: every point of this thread is bin t mod N
			for (int pnt = t; pnt < M; pnt += kThreads) fold(acc[0], pnt);
		}
	}
	// flush with one 64-bit atomic per accumulator (threads t, t+N, ... share a bin when N < 1024)
	if (t < M) {
#pragma unroll
		for (int a = 0; a < 16; a++) {
			if (a < A) {
				const int bin = (t + kThreads * a) & (N - 1);
				if (p.peak_hold) atomicMax(p.avg + s * N + bin, acc[a]);
				else atomicAdd(reinterpret_cast<unsigned long long *>(p.avg + s * N + bin), (unsigned long long)acc[a]);
			}
		}
	}
	if (t == 0) atomicAdd(p.samples + s, p.ds * p.chunks * (r_end - r_begin));  // :717
}

// The same scan specialised for the large-FFT, undecimated case (BASELINE config 4:
// bin_e = 13 or 14, one FFT per read, raw u8 input).  Differences from k_power_scan:
//  * thread (lane l, wave w) owns the contiguous points j = l<<(E-6) | w<<(E-10) | k:
//    its bytes arrive with one or two 16-byte loads, stay in registers from the DC sums
//    (phase A) to the windowing (phase B), and the NEXT read is fetched while this
//    one is transformed;
//  * with the lane index in the top bits of j the bit-reversed LDS address has the
//    lane in its LOW bits: phase B's scattered stores are bank-conflict free (they were
//    32-way conflicted with the natural mapping).
//
// COMB (E = 14, frames of 2^(14 + c) points): after the bit reversal over all 14 + c bits, block rev_c(b) of a frame
// holds the points j = k << c | b (the comb b of the frame) in the bit-reversed order of k - exactly what this
// kernel's phase B builds in LDS for a read of 16384 points -, and stages 0 .. 13 stay inside the block.  So a
// "read" is one comb (its bytes contiguous: k_power_comb_bytes has gathered them, the window likewise), the averages
// come from k_power_dc_part / _fin (remove_dc runs over the whole frame), and the final pass stores the block into
// the work buffer for the stages beyond 13 instead of accumulating |X|^2.
template <int E, bool COMB = false>
__global__ void __launch_bounds__(kThreads, 4) k_power_scan_big(const ScanParams p)
{
	static_assert(!COMB || E == 14, "a comb is one 16384-point block");
	constexpr int N = 1 << E;
	constexpr int P = N / kThreads;  // points per thread: 8 or 16
	constexpr int V = P / 8;         // uint4 loads per thread
	extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
	uint32_t *pts = sm;                       // [skewed_size(N)]
	uint32_t *tw = sm + skewed_size(N);       // [N]
	__shared__ int red[2][kThreads / 64];
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#if RTLPOWER_WAVE_PRIO
	{
		// sixteen waves in lock step, four per SIMD (wave w on SIMD w & 3): the four of a SIMD get four priorities
		const int w4 = __builtin_amdgcn_readfirstlane(wave) >> 2;
		if (w4 == 0) __builtin_amdgcn_s_setprio(3);
		else if (w4 == 1) __builtin_amdgcn_s_setprio(2);
		else if (w4 == 2) __builtin_amdgcn_s_setprio(1);
	}
#endif
	// COMB: the "reads" are this workgroup's share of all blocks, in order (the combs of a frame follow each other)
	const size_t s = COMB ? 0 : blockIdx.x / p.groups;
	const int grp = COMB ? 0 : (int)(blockIdx.x % p.groups);
	using idx_t = std::conditional_t<COMB, long long, int>;  // (the accumulating kernel has no register to spare for 64-bit counters)
	const idx_t r_begin = COMB ? (idx_t)((unsigned long long)blockIdx.x * p.comb_blocks / gridDim.x)
	                           : (idx_t)((long long)grp * p.nreads / p.groups);
	const idx_t r_end = COMB ? (idx_t)((unsigned long long)(blockIdx.x + 1) * p.comb_blocks / gridDim.x)
	                         : (idx_t)((long long)(grp + 1) * p.nreads / p.groups);
	if (r_begin >= r_end) return;
	unsigned long long st_clk = 0, st_rt = 0;
	if (p.stamps) { st_clk = __builtin_amdgcn_s_memtime(); st_rt = __builtin_amdgcn_s_memrealtime(); }
	for (int k = t; k < N; k += kThreads) tw[k] = p.tw[k];
	// the thread's points: the BIT-REVERSED lane index in the top bits of j, so that the bit-reversed LDS
	// address of point j0 + k has the lane itself in its low bits - consecutive lanes, consecutive dwords
	// (with the lane in the top bits the low address bits were its bit reversal: the first 32 lanes of a
	// wave hit only the even banks)
	const int lane_r = (int)(__brev((unsigned)lane) >> 26);
	const int j0 = (lane_r << (E - 6)) | (wave << (E - 10));
	// the window coefficients as 16-bit halves (only the low 16 bits of a product survive the reference's
	// int16 stores), two to a register
	const uint4 *wp = reinterpret_cast<const uint4 *>(p.window16 + j0 + (COMB ? (size_t)(r_begin & ((1 << p.comb_c) - 1)) << E : 0));
	const int scatter0 = skew((int)(__brev((unsigned)j0) >> (32 - E)));
	long long acc[COMB ? 1 : 16];
#pragma unroll
	for (int k = 0; k < (COMB ? 1 : 16); k++) acc[k] = 0;
	const uint8_t *base = p.iq8 + (COMB ? 0 : s * p.stride8) + 2 * (size_t)j0;
	const size_t read_bytes = COMB ? (size_t)2 * N : (size_t)p.buf_len;
	uint4 cur[V], nxt[V];
#pragma unroll
	for (int v = 0; v < V; v++) cur[v] = reinterpret_cast<const uint4 *>(base + (size_t)r_begin * read_bytes)[v];

	// ---- A: remove_dc sums (all 2N elements are below len_dec here), one read AHEAD ----------------------
	// at most N * 128 = 2^21 in magnitude: 32-bit sums, one v_dot4 per dword and component, the wave's total
	// by DPP (lane 63 holds it) into red[]; after a workgroup barrier every wave adds the sixteen partial
	// sums itself (lanes 0-15: I, 16-31: Q; a DPP row is 16 lanes) and divides by a compile-time constant.
	// Rounds 1-2 did this in 64 bits with an emulated division on one lane between two barriers of its own.
	// The sums of read r + 1 are taken from its registers while read r is transformed and ride on the
	// barriers the transform has anyway.
	auto dc_partial = [&](const uint4 (&d4)[V]) {
		int si = -127 * P, sq = -127 * P;
#pragma unroll
		for (int v = 0; v < V; v++) {
			const uint32_t d[4] = {d4[v].x, d4[v].y, d4[v].z, d4[v].w};
#pragma unroll
			for (int q = 0; q < 4; q++) {
				si = (int)__builtin_amdgcn_udot4(d[q], 0x00010001u, (uint32_t)si, false);
				sq = (int)__builtin_amdgcn_udot4(d[q], 0x01000100u, (uint32_t)sq, false);
			}
		}
		si = wave_total(si);
		sq = wave_total(sq);
		if (lane == 63) { red[0][wave] = si; red[1][wave] = sq; }
	};
	auto dc_average = [&](int &ai, int &aq) {
		int part = lane < 2 * (kThreads / 64) ? (&red[0][0])[lane] : 0;
		part += __builtin_amdgcn_update_dpp(0, part, 0x111, 0xf, 0xf, false);
		part += __builtin_amdgcn_update_dpp(0, part, 0x112, 0xf, 0xf, false);
		part += __builtin_amdgcn_update_dpp(0, part, 0x114, 0xf, 0xf, false);
		part += __builtin_amdgcn_update_dpp(0, part, 0x118, 0xf, 0xf, false);
		ai = (int)(int16_t)(__builtin_amdgcn_readlane(part, 15) / (2 * N));
		aq = (int)(int16_t)(__builtin_amdgcn_readlane(part, 31) / (2 * N - 1));
	};
	// the thread's window coefficients: fetched again for every read (32 bytes from a table that lives in
	// L2; held in registers across the radix-8 passes they pushed those into spilling), but EARLY - in front
	// of the final pass, whose few live registers leave room - so that phase B does not start with an L2
	// round trip.  The asm keeps the loop-invariant loads where they are.
	uint32_t w2[P / 2];
	auto load_window = [&]() {
		const uint4 *wq = wp;
		asm volatile("" : "+v"(wq));
#pragma unroll
		for (int k = 0; k < P / 8; k++) { const uint4 v = wq[k]; w2[4 * k] = v.x; w2[4 * k + 1] = v.y; w2[4 * k + 2] = v.z; w2[4 * k + 3] = v.w; }
	};
	load_window();
	int ai = 0, aq = 0;
	if (!COMB) dc_partial(cur);
	__syncthreads();  // also: the twiddle table is in place
	if (!COMB) dc_average(ai, aq);

	for (idx_t r = r_begin; r < r_end; r++) {
		if (COMB) {
			const int2 a = p.ave[r >> p.comb_c];  // remove_dc's averages of the comb's frame
			ai = a.x; aq = a.y;
		}
		// ---- B: convert, DC, window, bit-reversed placement (conflict-free) ---------
		// on packed pairs: v_perm lifts (I, Q) out of the dword as two zero-extended 16-bit halves, one
		// v_pk_sub takes 127 + average off both, one v_pk_mul_lo_u16 applies the window coefficient (only
		// the low 16 bits of either product survive the reference's int16 stores) - three instructions per
		// point where the scalar form took eight and two quarter-rate 32-bit multiplies
		typedef unsigned short upk16_t __attribute__((ext_vector_type(2)));
		const upk16_t dcw = {(unsigned short)(127 + ai), (unsigned short)(127 + aq)};
#pragma unroll
		for (int k = 0; k < P; k++) {
			const uint32_t d = (&cur[k / 8].x)[(k / 2) & 3];
			const upk16_t iq = __builtin_bit_cast(upk16_t, __builtin_amdgcn_perm(0u, d, (k & 1) ? 0x0c030c02u : 0x0c010c00u));
			const upk16_t wpair = __builtin_bit_cast(upk16_t, w2[k / 2]);
			const unsigned short wk = (k & 1) ? wpair.y : wpair.x;
			const upk16_t ww = {wk, wk};
			// bit reversal of j0 + k: k's E - 10 bits land above bit 10 (a compile-time offset, additive
			// under the skew), the wave's four bits at 6..9, the lane in the low six
			pts[scatter0 + skew_c((int)(brev_c(k) >> (32 - (E - 10))) << 10)] = __builtin_bit_cast(uint32_t, (upk16_t)((upk16_t)(iq - dcw) * ww));
		}
		// the next read's bytes travel while this one is transformed
		const bool more = r + 1 < r_end;
		if (more) {
#pragma unroll
			for (int v = 0; v < V; v++) nxt[v] = reinterpret_cast<const uint4 *>(base + (size_t)(r + 1) * read_bytes)[v];
		}
		__syncthreads();
		// ---- C: E = 13: four radix-8 passes and a radix-2 one; E = 14: four and a radix-4 one ----------
		// Stages 0-8 are 512-point transforms, and group g of passes 0, 1 and 2 reads only what groups of
		// the SAME 64 (g & ~63 ..) wrote in the pass before: a wave's own lanes.  No workgroup barrier
		// until stage 9 - the sixteen waves drift apart, one's LDS traffic under another's butterflies.
		for (int g = t; g < N / 8; g += kThreads) {
			fft_group<3, 0>(pts, tw, 0, g); wave_sync();
			fft_group<3, 3>(pts, tw, 3, g); wave_sync();
			fft_group<3, 6>(pts, tw, 6, g);
		}
		if (more && !COMB) dc_partial(nxt);
		__syncthreads();
		fft_pass<3, 9>(pts, tw, N, 9, t);
		__syncthreads();
		// ---- the last stages and D: the outputs of group g = t + 1024 it of the final pass are the bins
		// g + 4096 k - the thread's own accumulators a = it + 4 k: |X|^2 goes from the butterfly's registers
		// into them, the spectrum is never written back to LDS.  One |X|^2 is at most 2^31: the peak-hold
		// maximum lives in 32 bits, the sum takes one 64-bit add.
		if (more) {
			if (COMB) wp = reinterpret_cast<const uint4 *>(p.window16 + j0 + ((size_t)((r + 1) & ((1 << p.comb_c) - 1)) << E));
			load_window();
		}
		{
			constexpr int R = E - 12, G = 1 << R;
			// COMB: comb b of frame f is block rev_c(b) of the frame
			uint32_t *sink = nullptr;
			if (COMB) {
				const unsigned b = (unsigned)(r & ((1 << p.comb_c) - 1));
				const unsigned br = p.comb_c ? __brev(b) >> (32 - p.comb_c) : 0u;
				sink = p.work + ((size_t)(r >> p.comb_c) << (E + p.comb_c)) + ((size_t)br << E);
			}
			// the LDS addresses of the four rounds are loop-invariant; hoisted out of the read loop they are
			// spilled and reloaded - two integer operations each, recomputed here, are cheaper
			int tl = t;
			asm volatile("" : "+v"(tl));
#pragma unroll
			for (int it = 0; it < 4; it++) {
				uint32_t x[G];
				if (it) __builtin_amdgcn_sched_barrier(0);  // one round's operands at a time: all four at once spill
				fft_group_regs<R, 12>(pts, tw, tl + it * kThreads, x);
				if constexpr (COMB) {
					// group g of the final pass holds the block's points g + 4096 k
#pragma unroll
					for (int k = 0; k < G; k++) sink[tl + it * kThreads + (k << 12)] = x[k];
				} else {
#pragma unroll
					for (int k = 0; k < G; k++) {
						const int a = it + 4 * k;
						if (p.peak_hold) {
							const uint32_t pw = (uint32_t)power_of(x[k]);
							const uint32_t m = (uint32_t)acc[a];
							acc[a] = (long long)(pw > m ? pw : m);
						} else {
							acc[a] += power_of(x[k]);
						}
					}
				}
			}
		}
		if (more && !COMB) dc_average(ai, aq);  // red[] was written before the barrier behind stage 8
		__syncthreads();               // the final pass is done reading pts: the next read may be placed
#pragma unroll
		for (int v = 0; v < V; v++) cur[v] = nxt[v];
	}
	if constexpr (!COMB) {
#pragma unroll
		for (int a = 0; a < P; a++) {
			const int bin = t + kThreads * a;
			if (p.peak_hold) atomicMax(p.avg + s * N + bin, acc[a]);
			else atomicAdd(reinterpret_cast<unsigned long long *>(p.avg + s * N + bin), (unsigned long long)acc[a]);
		}
		if (t == 0) atomicAdd(p.samples + s, p.ds * (int)(r_end - r_begin));
	}
	if (p.stamps && t == 0) {
		unsigned long long *o = p.stamps + (size_t)blockIdx.x * 4;
		o[0] = st_clk; o[1] = __builtin_amdgcn_s_memtime(); o[2] = st_rt; o[3] = __builtin_amdgcn_s_memrealtime();
	}
}

// rtl_power's everyday shape: SEVERAL frames per read.  The planner never reads less than 16384 bytes
// (src/rtl_power.c:501-504), so every scan below 8192 bins - "-f 88M:108M:125k" is 32 of them - hands over reads of
// M = 2^E = 8192 points (or 16384) that hold chunks = M >> bin_e frames.  Stage s < bin_e pairs points inside a
// frame only, so the stages run over all M points at once exactly as k_power_scan_big's do - same thread-owned
// 16-byte loads with the next read in flight, same packed phases A and B (the bit reversal is over bin_e bits inside
// the frame, the window index is j mod N), stages 0 .. bin_e - 1 with the stage a run-time value -, and phase D folds
// the M points onto the N bins: point pos is bin pos mod N, a thread's eight or sixteen points keep their own 64-bit
// sums over the reads, and at the end the workgroup adds them up per bin in LDS (ds_add_u64 / ds_max_u64 on the
// points' area) before N global atomics.  Before: k_power_scan's byte loads and scalar phases, 100-150 Gsamples/s
// where the 8192-bin kernel runs 380.
// SRC16 (round 5): the reads are DECIMATED ones - packed (I, Q) int16 pairs from k_power_downsample_iq (-F) or
// k_power_boxcar, 2^dec_e >= 512 points each - and one "read" of this kernel is the M / 2^dec_e consecutive reads of
// the stream that fill its M points (a trailing group that the stream's reads do not fill is topped up with zeros:
// zero samples with a zero average transform to zero power).  remove_dc (src/rtl_power.c:581-596, called on
// buf_len / ds elements, :692-693) runs per decimated read: a wave's 512 points lie inside one read, and the read's
// sums are those of its 2^(dec_e - 9) waves.  Everything from the placement on is the same.
template <int E, bool SRC16 = false>
__global__ void __launch_bounds__(kThreads) k_power_scan_frames(const ScanParams p)
{
	constexpr int M = 1 << E;
	constexpr int P = M / kThreads;  // points per thread: 8 or 16
	constexpr int V = SRC16 ? P / 4 : P / 8;  // 16-byte loads per thread
	extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
	uint32_t *pts = sm;                       // [skewed_size(M)]
	uint32_t *tw = sm + skewed_size(M);       // [N]
	__shared__ int red[2][kThreads / 64];
	const int BE = p.bin_e, N = 1 << BE;      // 3 <= BE < E (SRC16: <= dec_e)
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const size_t s = blockIdx.x / p.groups;
	const int grp = (int)(blockIdx.x % p.groups);
	// SRC16: the unit of the loop is a group of `per` decimated reads
	const int per = SRC16 ? M >> p.dec_e : 1;
	const int units = SRC16 ? (p.nreads + per - 1) / per : p.nreads;
	const int r_begin = (int)((long long)grp * units / p.groups), r_end = (int)((long long)(grp + 1) * units / p.groups);
	if (r_begin >= r_end) return;
	for (int k = t; k < N; k += kThreads) tw[k] = p.tw[k];
	const int j0 = t * P;                     // the thread's points j0 .. j0 + P - 1: a wave reads 64 P contiguous points
	// their LDS addresses: frame j >> BE stays, the index inside the frame is bit-reversed
	int scatter[P];
#pragma unroll
	for (int k = 0; k < P; k++) {
		const int j = j0 + k;
		scatter[k] = skew((j & ~(N - 1)) | (int)(__brev((unsigned)(j & (N - 1))) >> (32 - BE)));
	}
	// window coefficients (16-bit halves, two to a register): P consecutive ones from j0 mod N (N >= 8 = a multiple of
	// what a 16-byte load holds; N < P wraps inside the thread's points: the table is then read piece by piece)
	uint32_t w2[P / 2];
#pragma unroll
	for (int k = 0; k < P / 8; k++) {
		const uint4 v = *reinterpret_cast<const uint4 *>(p.window16 + ((j0 + 8 * k) & (N - 1)));
		w2[4 * k] = v.x; w2[4 * k + 1] = v.y; w2[4 * k + 2] = v.z; w2[4 * k + 3] = v.w;
	}
	long long acc[P];
#pragma unroll
	for (int k = 0; k < P; k++) acc[k] = 0;
	const uint8_t *base = SRC16 ? reinterpret_cast<const uint8_t *>(p.dec32 + s * p.dec32_stream_stride + (size_t)j0)
	                            : p.iq8 + s * p.stride8 + 2 * (size_t)j0;
	const size_t unit_bytes = SRC16 ? (size_t)4 * M : (size_t)p.buf_len;
	// SRC16: the decimated read this thread's points belong to, inside the unit; reads the stream does not have read as zeros
	const int my_read = SRC16 ? j0 >> p.dec_e : 0;
	uint4 cur[V], nxt[V];
	auto load_unit = [&](uint4 (&d4)[V], int u) {
		const bool there = !SRC16 || u * per + my_read < p.nreads;
#pragma unroll
		for (int v = 0; v < V; v++) d4[v] = there ? reinterpret_cast<const uint4 *>(base + (size_t)u * unit_bytes)[v] : make_uint4(0, 0, 0, 0);
	};
	load_unit(cur, r_begin);
	auto dc_partial = [&](const uint4 (&d4)[V]) {
		int si = SRC16 ? 0 : -127 * P, sq = SRC16 ? 0 : -127 * P;
#pragma unroll
		for (int v = 0; v < V; v++) {
			const uint32_t d[4] = {d4[v].x, d4[v].y, d4[v].z, d4[v].w};
#pragma unroll
			for (int q = 0; q < 4; q++) {
				if (SRC16) {
					const pk16_t one_i = {1, 0}, one_q = {0, 1};
					si = __builtin_amdgcn_sdot2(__builtin_bit_cast(pk16_t, d[q]), one_i, si, false);
					sq = __builtin_amdgcn_sdot2(__builtin_bit_cast(pk16_t, d[q]), one_q, sq, false);
				} else {
					si = (int)__builtin_amdgcn_udot4(d[q], 0x00010001u, (uint32_t)si, false);
					sq = (int)__builtin_amdgcn_udot4(d[q], 0x01000100u, (uint32_t)sq, false);
				}
			}
		}
		si = wave_total(si);
		sq = wave_total(sq);
		if (lane == 63) { red[0][wave] = si; red[1][wave] = sq; }
	};
	auto dc_average = [&](int &ai, int &aq) {  // remove_dc over the whole read: 2 M elements (src/rtl_power.c:581-596)
		int part = lane < 2 * (kThreads / 64) ? (&red[0][0])[lane] : 0;
		part += __builtin_amdgcn_update_dpp(0, part, 0x111, 0xf, 0xf, false);
		part += __builtin_amdgcn_update_dpp(0, part, 0x112, 0xf, 0xf, false);
		part += __builtin_amdgcn_update_dpp(0, part, 0x114, 0xf, 0xf, false);
		part += __builtin_amdgcn_update_dpp(0, part, 0x118, 0xf, 0xf, false);
		if (SRC16) {
			// lanes 0-15 hold the running sums of the sixteen waves' I parts, 16-31 those of the Q parts: this wave's read
			// is the waves [first, first + wpr)
			const int wpr_e = p.dec_e - 9;
			const int first = __builtin_amdgcn_readfirstlane((wave >> wpr_e) << wpr_e), last = first + (1 << wpr_e) - 1;
			int sumI = __builtin_amdgcn_readlane(part, last), sumQ = __builtin_amdgcn_readlane(part, 16 + last);
			if (first > 0) { sumI -= __builtin_amdgcn_readlane(part, first - 1); sumQ -= __builtin_amdgcn_readlane(part, 16 + first - 1); }
			const int len = 2 << p.dec_e;  // elements of the decimated read
			ai = (int)(int16_t)(sumI / len);
			aq = (int)(int16_t)(sumQ / (len - 1));
			return;
		}
		ai = (int)(int16_t)(__builtin_amdgcn_readlane(part, 15) / (2 * M));
		aq = (int)(int16_t)(__builtin_amdgcn_readlane(part, 31) / (2 * M - 1));
	};
	int ai, aq;
	dc_partial(cur);
	__syncthreads();  // also: the twiddle table is in place
	dc_average(ai, aq);

	for (int r = r_begin; r < r_end; r++) {
		// ---- B: convert, DC, window on packed pairs (as k_power_scan_big), placement ----
		typedef unsigned short upk16_t __attribute__((ext_vector_type(2)));
		const upk16_t dcw = {(unsigned short)((SRC16 ? 0 : 127) + ai), (unsigned short)((SRC16 ? 0 : 127) + aq)};
#pragma unroll
		for (int k = 0; k < P; k++) {
			upk16_t iq;
			if (SRC16) {
				iq = __builtin_bit_cast(upk16_t, (&cur[k / 4].x)[k & 3]);
			} else {
				const uint32_t d = (&cur[k / 8].x)[(k / 2) & 3];
				iq = __builtin_bit_cast(upk16_t, __builtin_amdgcn_perm(0u, d, (k & 1) ? 0x0c030c02u : 0x0c010c00u));
			}
			const upk16_t wpair = __builtin_bit_cast(upk16_t, w2[k / 2]);
			const unsigned short wk = (k & 1) ? wpair.y : wpair.x;
			const upk16_t ww = {wk, wk};
			pts[scatter[k]] = __builtin_bit_cast(uint32_t, (upk16_t)((upk16_t)(iq - dcw) * ww));
		}
		const bool more = r + 1 < r_end;
		if (more) load_unit(nxt, r + 1);
		__syncthreads();
		// ---- C: stages 0 .. BE - 1, three per LDS round trip; the radix-8 passes below stage 9 stay inside a wave's
		// own 512-point blocks (k_power_scan_big), every other step needs the workgroup
		bool dc_done = false;
		for (int st = 0; st < BE;) {
			const int R = BE - st >= 3 ? 3 : BE - st;
			if (R == 3) fft_pass<3>(pts, tw, M, st, t);
			else if (R == 2) fft_pass<2>(pts, tw, M, st, t);
			else fft_pass<1>(pts, tw, M, st, t);
			st += R;
			if (more && !dc_done) { dc_partial(nxt); dc_done = true; }  // rides on the barriers the transform has anyway
			const bool own = R == 3 && st + 3 <= 9 && BE - st >= 3;    // the next pass is a radix-8 one inside the same blocks
			if (own) wave_sync(); else __syncthreads();
		}
		// ---- D: point pos is bin pos mod N; the thread's points t + 1024 a (conflict-free under the skew) ----
#pragma unroll
		for (int a = 0; a < P; a++) {
			const uint32_t x = pts[skew(t + kThreads * a)];
			if (p.peak_hold) {
				const uint32_t pw = (uint32_t)power_of(x);
				const uint32_t m = (uint32_t)acc[a];
				acc[a] = (long long)(pw > m ? pw : m);
			} else {
				acc[a] += power_of(x);
			}
		}
		if (more) dc_average(ai, aq);
		__syncthreads();  // phase D is done reading pts: the next read may be placed
#pragma unroll
		for (int v = 0; v < V; v++) cur[v] = nxt[v];
	}
	// ---- the workgroup's sums per bin (the points' area is free now: N 64-bit words), then N global atomics ----
	unsigned long long *bins = reinterpret_cast<unsigned long long *>(sm);
	for (int k = t; k < N; k += kThreads) bins[k] = 0;
	__syncthreads();
#pragma unroll
	for (int a = 0; a < P; a++) {
		const int bin = (t + kThreads * a) & (N - 1);
		if (p.peak_hold) atomicMax(bins + bin, (unsigned long long)acc[a]);
		else atomicAdd(bins + bin, (unsigned long long)acc[a]);
	}
	__syncthreads();
	for (int k = t; k < N; k += kThreads) {
		if (p.peak_hold) atomicMax(p.avg + s * N + k, (long long)bins[k]);
		else atomicAdd(reinterpret_cast<unsigned long long *>(p.avg + s * N + k), bins[k]);
	}
	if (t == 0) {  // :717, once per frame
		if (SRC16) {
			const int reads = min(p.nreads, r_end * per) - r_begin * per;  // the stream's own reads in this group's units
			atomicAdd(p.samples + s, p.ds * ((1 << p.dec_e) >> BE) * reads);
		} else {
			atomicAdd(p.samples + s, p.ds * p.chunks * (r_end - r_begin));
		}
	}
}

// ---- transforms that do not fit one workgroup's LDS: bin_e 15 ... 21, or more frames per read than 16384 points ----
// frequency_range() plans up to 2^21 bins (src/rtl_power.c:483-486) and fix_fft (:271-327) has no size limit.  The
// same radix-2 DIT stages, with the same FIX_MPY rounding and the same ">> 1" per stage, run over a work buffer in
// HBM: after the bit-reversed placement, stages 0 .. 13 stay inside contiguous blocks of 16384 points (one block =
// one workgroup's LDS, the k_power_scan machinery), and stages 14 .. E-1 are passes over the whole frame, three
// stages per pass (a thread holds the eight points of a radix-8 group at stride 2^st in registers).  The integers
// are those of the reference stage by stage; only where the intermediate array lives differs.
//   k_power_dc      remove_dc's averages of every (stream, read)           (:581-596, the half-DC quirk)
//   k_power_place   convert, DC, window (int16 wrap), bit-reversed store   (:666-668, 697-706, 282-297)
//   k_power_fft_lds stages 0 .. min(E, 14) - 1 of every block of 2^min(E,14) points
//   k_power_fft_gl  stages st .. st + R - 1 for st >= 14
//   k_power_accum   avg[] += |X|^2 or peak hold over the batch's reads and frames (:708-716), samples (:717)
struct StagedParams {
	const uint8_t *iq8; size_t stride8;
	const int16_t *dec; size_t dec_stream_stride, dec_read_stride; int dec_elems;
	int nreads;              // reads of this batch
	int buf_len, len_dec, bin_e, chunks, ds, peak_hold;
	const int32_t *window;   // [N]
	const uint32_t *tw;      // [N] per-stage twiddles
	int2 *ave;               // [stream][read] (avgI, avgQ)
	uint32_t *work;          // [stream][read][chunks][N] packed (re, im)
	long long *avg; int32_t *samples;
	int nstreams;
};

__device__ __forceinline__ int staged_element(const StagedParams &p, const uint8_t *raw, const int16_t *dec, int e)
{
	if (dec) return e < p.dec_elems ? (int)dec[e] : 0;
	return e < p.buf_len ? (int)raw[e] - 127 : 0;
}

__global__ void __launch_bounds__(256) k_power_dc(const StagedParams p)
{
	const size_t sr = blockIdx.x;  // (stream, read)
	const size_t s = sr / p.nreads;
	const int r = (int)(sr % p.nreads);
	const uint8_t *raw = p.iq8 ? p.iq8 + s * p.stride8 + (size_t)r * p.buf_len : nullptr;
	const int16_t *dec = p.dec ? p.dec + s * p.dec_stream_stride + (size_t)r * p.dec_read_stride : nullptr;
	long long si = 0, sq = 0;
	for (int pnt = threadIdx.x; 2 * pnt < p.len_dec; pnt += 256) {
		si += staged_element(p, raw, dec, 2 * pnt);
		if (2 * pnt + 1 < p.len_dec) sq += staged_element(p, raw, dec, 2 * pnt + 1);
	}
	__shared__ long long red[2][256];
	red[0][threadIdx.x] = si; red[1][threadIdx.x] = sq;
	__syncthreads();
	for (int off = 128; off > 0; off >>= 1) {
		if ((int)threadIdx.x < off) { red[0][threadIdx.x] += red[0][threadIdx.x + off]; red[1][threadIdx.x] += red[1][threadIdx.x + off]; }
		__syncthreads();
	}
	if (threadIdx.x == 0)  // as k_power_scan's phase A: the sum over N/2 values divided by N (and N - 1)
		p.ave[sr] = make_int2((int)(int16_t)(red[0][0] / (long long)p.len_dec),
		                      p.len_dec > 1 ? (int)(int16_t)(red[1][0] / (long long)(p.len_dec - 1)) : 0);
}

__global__ void __launch_bounds__(256) k_power_place(const StagedParams p)
{
	const int N = 1 << p.bin_e;
	const size_t M = (size_t)p.chunks * N;
	const size_t total = (size_t)p.nstreams * p.nreads * M;
	for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
		const size_t sr = g / M;
		const int pnt = (int)(g % M);
		const size_t s = sr / p.nreads;
		const int r = (int)(sr % p.nreads);
		const uint8_t *raw = p.iq8 ? p.iq8 + s * p.stride8 + (size_t)r * p.buf_len : nullptr;
		const int16_t *dec = p.dec ? p.dec + s * p.dec_stream_stride + (size_t)r * p.dec_read_stride : nullptr;
		const int2 a = p.ave[sr];
		const int c = pnt >> p.bin_e, j = pnt & (N - 1);
		int vi = staged_element(p, raw, dec, 2 * pnt), vq = staged_element(p, raw, dec, 2 * pnt + 1);
		if (2 * pnt < p.len_dec) vi = (int16_t)(vi - a.x);
		if (2 * pnt + 1 < p.len_dec) vq = (int16_t)(vq - a.y);
		const int w = p.window[j];
		vi = (int16_t)(vi * w);
		vq = (int16_t)(vq * w);
		const int rj = (int)(__brev((unsigned)j) >> (32 - p.bin_e));
		p.work[sr * M + ((size_t)c << p.bin_e) + rj] = pack_iq(vi, vq);
	}
}

// The same placement for bin_e >= 12 without the scattered 4-byte stores (every store of k_power_place lands in a line of
// its own: the bit reversal of consecutive j): the index is cut into j = a << (E - 6) | m << 6 | c, and a workgroup takes
// the 64 x 64 tile (a, c) of one m - 64 rows of 64 CONSECUTIVE samples (128 input bytes each) -, turns it over in LDS and
// writes 64 rows of 64 consecutive points: rev(j) = rev6(c) << (E - 6) | rev(m) << 6 | rev6(a), so for a fixed c the
// 64 values of a fill one 256-byte piece of the frame.  Both sides move whole lines.
__global__ void __launch_bounds__(256) k_power_place_tiled(const StagedParams p)
{
	__shared__ uint32_t tile[64][65];
	const int E = p.bin_e, N = 1 << E;
	const int mbits = E - 12;                      // >= 0
	const size_t M = (size_t)p.chunks * N;
	const size_t tiles_per_frame = (size_t)1 << mbits;
	const size_t total = (size_t)p.nstreams * p.nreads * p.chunks * tiles_per_frame;
	const int t = threadIdx.x, c = t & 63, a0 = t >> 6;  // 4 rows per sweep
	for (size_t g = blockIdx.x; g < total; g += gridDim.x) {
		const size_t m = g & (tiles_per_frame - 1);
		const size_t fr = g >> mbits;                // (stream, read, chunk)
		const int ch = (int)(fr % p.chunks);
		const size_t sr = fr / p.chunks;
		const size_t s = sr / p.nreads;
		const int r = (int)(sr % p.nreads);
		const uint8_t *raw = p.iq8 ? p.iq8 + s * p.stride8 + (size_t)r * p.buf_len : nullptr;
		const int16_t *dec = p.dec ? p.dec + s * p.dec_stream_stride + (size_t)r * p.dec_read_stride : nullptr;
		const int2 av = p.ave[sr];
		__syncthreads();
		for (int a = a0; a < 64; a += 4) {
			const int j = (a << (E - 6)) | ((int)m << 6) | c;
			const int pnt = (ch << E) + j;           // point of the read (remove_dc and the data's end go by it)
			int vi = staged_element(p, raw, dec, 2 * pnt), vq = staged_element(p, raw, dec, 2 * pnt + 1);
			if (2 * pnt < p.len_dec) vi = (int16_t)(vi - av.x);
			if (2 * pnt + 1 < p.len_dec) vq = (int16_t)(vq - av.y);
			const int w = p.window[j];
			vi = (int16_t)(vi * w);
			vq = (int16_t)(vq * w);
			tile[a][c] = pack_iq(vi, vq);
		}
		__syncthreads();
		// output row c' = t >> 6 (+ 4 k): the points rev(j) for that c, a = rev6(t & 63) so that consecutive threads write consecutive points
		const int ar = (int)(__brev((unsigned)(t & 63)) >> 26);
		const unsigned mr = mbits ? (__brev((unsigned)m) >> (32 - mbits)) : 0u;
		for (int cc = a0; cc < 64; cc += 4) {
			const unsigned cr = __brev((unsigned)cc) >> 26;
			const size_t rj = ((size_t)cr << (E - 6)) | ((size_t)mr << 6) | (size_t)(t & 63);
			p.work[sr * M + ((size_t)ch << E) + rj] = tile[ar][cc];
		}
	}
}

// stages 0 .. eb - 1 (eb = min(bin_e, 14)) of one block of 2^eb points per workgroup
__global__ void __launch_bounds__(kThreads) k_power_fft_lds(uint32_t *work, const uint32_t *twg, int eb, size_t nblocks_total)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
	const int B = 1 << eb;
	uint32_t *pts = sm;                  // [skewed_size(B)]
	uint32_t *tw = sm + skewed_size(B);  // [B]: the stages below eb use the table's first 2^eb - 1 entries
	const int t = threadIdx.x;
	for (int k = t; k < B; k += kThreads) tw[k] = twg[k];
	for (size_t blk = blockIdx.x; blk < nblocks_total; blk += gridDim.x) {
		uint32_t *w = work + blk * (size_t)B;
		__syncthreads();
		for (int k = t; k < B; k += kThreads) pts[skew(k)] = w[k];
		__syncthreads();
		for (int st = 0; st < eb;) {
			const int R = eb - st >= 3 ? 3 : eb - st;
			if (R == 3) fft_pass<3>(pts, tw, B, st, t);
			else if (R == 2) fft_pass<2>(pts, tw, B, st, t);
			else fft_pass<1>(pts, tw, B, st, t);
			st += R;
			__syncthreads();
		}
		for (int k = t; k < B; k += kThreads) w[k] = pts[skew(k)];
	}
}

// stages st .. st + R - 1 (st >= 14) over frames of N points in HBM: one thread per group of 2^R points at stride 2^st
template <int R>
__global__ void __launch_bounds__(256) k_power_fft_gl(uint32_t *work, const uint32_t *tw, int bin_e, int st, size_t frames)
{
	constexpr int G = 1 << R;
	const size_t per_frame = (size_t)1 << (bin_e - R);
	const size_t total = frames * per_frame;
	const int h = 1 << st;
	for (size_t g0 = (size_t)blockIdx.x * 256 + threadIdx.x; g0 < total; g0 += (size_t)gridDim.x * 256) {
		const size_t f = g0 >> (bin_e - R);
		const int g = (int)(g0 & (per_frame - 1));
		const int glo = g & (h - 1), ghi = g >> st;
		uint32_t *base = work + (f << bin_e) + (((size_t)ghi << (st + R)) | (size_t)glo);
		uint32_t x[G];
#pragma unroll
		for (int k = 0; k < G; k++) x[k] = base[(size_t)k << st];
#pragma unroll
		for (int r = 0; r < R; r++) {
			const uint32_t *tws = tw + (((size_t)1 << (st + r)) - 1) + glo;
#pragma unroll
			for (int k = 0; k < G; k++) {
				if (k & (1 << r)) continue;
				const int kk = k & ((1 << r) - 1);
				butterfly<0>(x[k], x[k + (1 << r)], tws[(size_t)kk * h]);
			}
		}
#pragma unroll
		for (int k = 0; k < G; k++) base[(size_t)k << st] = x[k];
	}
}

// one thread per (stream, bin): the batch's reads and frames in order (no atomics, one owner per accumulator)
__global__ void __launch_bounds__(256) k_power_accum(const StagedParams p)
{
	const int N = 1 << p.bin_e;
	const size_t M = (size_t)p.chunks * N;
	const size_t total = (size_t)p.nstreams * N;
	for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
		const size_t s = g >> p.bin_e;
		const int bin = (int)(g & (size_t)(N - 1));
		long long a = p.avg[g];
		for (int r = 0; r < p.nreads; r++)
			for (int c = 0; c < p.chunks; c++) {
				const long long pw = power_of(p.work[(s * p.nreads + r) * M + ((size_t)c << p.bin_e) + bin]);
				a = p.peak_hold ? (pw > a ? pw : a) : a + pw;
			}
		p.avg[g] = a;
		if (bin == 0) p.samples[s] += p.ds * p.chunks * p.nreads;  // :717, once per frame
	}
}

// ---- the same path for undecimated reads of exactly one frame (rtl_power's fine-bin scans), round 4 ----
// k_power_comb_bytes: the frame's BYTES comb by comb, so that k_power_scan_big<14, true> takes a comb as it takes a
// 16384-point read - and, from the same registers, the sums of its 16 KiB tile for remove_dc (k_power_dc_fin adds a
// frame's tiles up; one workgroup per (stream, read) left 2^21-point reads to 256 threads each); k_power_fft_gl_acc:
// the last pass over HBM accumulates |X|^2 from its registers - the spectrum is never written, k_power_accum never
// reads it.
__global__ void __launch_bounds__(64) k_power_dc_fin(const int2 *part, int slices, int len_dec, size_t nsr, int2 *ave)
{
	const size_t sr = (size_t)blockIdx.x * 64 + threadIdx.x;
	if (sr >= nsr) return;
	long long si = 0, sq = 0;
	for (int k = 0; k < slices; k++) { const int2 v = part[sr * slices + k]; si += v.x; sq += v.y; }
	// as k_power_dc: the sum over N/2 values divided by N (and N - 1)
	ave[sr] = make_int2((int)(int16_t)(si / (long long)len_dec), (int)(int16_t)(sq / (long long)(len_dec - 1)));
}

// out[frame][b][k] = in[frame][k << c | b] (2-byte points): tiles of 8192 consecutive points, K = 8192 >> c values of k
// for every comb; 16-byte loads, 16-byte stores in runs of K points (c <= 7: at least 128 bytes)
__global__ void __launch_bounds__(256) k_power_comb_bytes(const uint8_t *iq8, size_t stride8, int nreads, int buf_len, int bin_e,
                                                          size_t tiles_total, uint8_t *out, int2 *part)
{
	__shared__ __attribute__((aligned(16))) uint16_t tile[8192];
	__shared__ int red[2][4];
	const int c = bin_e - 14, C = 1 << c, kshift = 13 - c, K = 1 << kshift;
	const int tpf = 1 << (bin_e - 13);  // tiles per frame
	const int t = threadIdx.x;
	for (size_t g = blockIdx.x; g < tiles_total; g += gridDim.x) {
		const size_t sr = g >> (bin_e - 13);
		const int T = (int)(g & (size_t)(tpf - 1));
		const size_t s = sr / nreads;
		const int r = (int)(sr % nreads);
		const uint4 *src = reinterpret_cast<const uint4 *>(iq8 + s * stride8 + (size_t)r * buf_len + (size_t)T * 16384);
		uint4 d[4];
#pragma unroll
		for (int i = 0; i < 4; i++) d[i] = src[i * 256 + t];
		// remove_dc's sums of this tile (src/rtl_power.c:581-596): at most 8192 * 128 in magnitude
		int si = -127 * 32, sq = si;
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const uint32_t w[4] = {d[i].x, d[i].y, d[i].z, d[i].w};
#pragma unroll
			for (int q = 0; q < 4; q++) {
				si = (int)__builtin_amdgcn_udot4(w[q], 0x00010001u, (uint32_t)si, false);
				sq = (int)__builtin_amdgcn_udot4(w[q], 0x01000100u, (uint32_t)sq, false);
			}
		}
		si = wave_total(si);
		sq = wave_total(sq);
		__syncthreads();  // the previous tile has been read out
		if ((t & 63) == 63) { red[0][t >> 6] = si; red[1][t >> 6] = sq; }
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const int q0 = (i * 256 + t) * 8;
			const uint32_t w[4] = {d[i].x, d[i].y, d[i].z, d[i].w};
#pragma unroll
			for (int e = 0; e < 8; e++) {
				const int q = q0 + e;
				tile[((q & (C - 1)) << kshift) | (q >> c)] = (uint16_t)(w[e >> 1] >> ((e & 1) * 16));
			}
		}
		__syncthreads();
		if (t == 0) part[g] = make_int2(red[0][0] + red[0][1] + red[0][2] + red[0][3], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
		uint8_t *dst = out + (sr << (bin_e + 1)) + (size_t)T * K * 2;
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const int u = (i * 256 + t) * 8;  // first of eight points of one comb
			const int b = u >> kshift, kl = u & (K - 1);
			*reinterpret_cast<uint4 *>(dst + ((size_t)b << 15) + (size_t)kl * 2) = *reinterpret_cast<const uint4 *>(tile + u);
		}
	}
}

// stages st .. st + R - 1 = the LAST ones (st + R == bin_e), and phase D with them: one thread per (stream, group of
// 2^R points at stride 2^st) walks the batch's frames of its stream in order (the next frame's points in flight), the
// group's twiddles stay in registers, |X|^2 goes from the butterflies' registers into the thread's own accumulators
// (one owner per bin: no atomics).  src/rtl_power.c:309-321, 708-717
template <int R>
__global__ void __launch_bounds__(256) k_power_fft_gl_acc(const StagedParams p, int st)
{
	constexpr int G = 1 << R;
	const int E = p.bin_e;
	const size_t per_frame = (size_t)1 << (E - R);
	const size_t total = (size_t)p.nstreams * per_frame;
	const int h = 1 << st;
	const size_t nfr = (size_t)p.nreads * p.chunks;
	for (size_t g0 = (size_t)blockIdx.x * 256 + threadIdx.x; g0 < total; g0 += (size_t)gridDim.x * 256) {
		const size_t s = g0 >> (E - R);
		const int g = (int)(g0 & (per_frame - 1));
		const int glo = g & (h - 1), ghi = g >> st;
		const size_t pos = ((size_t)ghi << (st + R)) | (size_t)glo;
		uint32_t w[G];  // w[(1 << r) - 1 + kk]
#pragma unroll
		for (int r = 0; r < R; r++)
#pragma unroll
			for (int kk = 0; kk < (1 << r); kk++) w[(1 << r) - 1 + kk] = p.tw[(((size_t)1 << (st + r)) - 1) + glo + (size_t)kk * h];
		long long *av = p.avg + (s << E) + pos;
		long long acc[G];
#pragma unroll
		for (int k = 0; k < G; k++) acc[k] = av[(size_t)k << st];
		const uint32_t *base = p.work + ((s * nfr) << E) + pos;
		uint32_t x[G], nx[G];
#pragma unroll
		for (int k = 0; k < G; k++) x[k] = base[(size_t)k << st];
		for (size_t f = 0; f < nfr; f++) {
			if (f + 1 < nfr) {
#pragma unroll
				for (int k = 0; k < G; k++) nx[k] = base[((f + 1) << E) + ((size_t)k << st)];
			}
#pragma unroll
			for (int r = 0; r < R; r++) {
#pragma unroll
				for (int k = 0; k < G; k++) {
					if (k & (1 << r)) continue;
					butterfly<0>(x[k], x[k + (1 << r)], w[(1 << r) - 1 + (k & ((1 << r) - 1))]);
				}
			}
#pragma unroll
			for (int k = 0; k < G; k++) {
				const long long pw = power_of(x[k]);
				acc[k] = p.peak_hold ? (pw > acc[k] ? pw : acc[k]) : acc[k] + pw;
				x[k] = nx[k];
			}
		}
#pragma unroll
		for (int k = 0; k < G; k++) av[(size_t)k << st] = acc[k];
		if (g == 0) p.samples[s] += p.ds * p.chunks * p.nreads;  // :717, once per frame
	}
}

// rms_power(), src/rtl_power.c:410-436 (bin_e == 0): one workgroup per stream
__global__ void __launch_bounds__(256) k_power_rms(const uint8_t *iq8, size_t stride8, int nreads, int buf_len,
                                                   int peak_hold, long long *avg, int32_t *samples)
{
	__shared__ long long rp[256], rt[256];
	const size_t s = blockIdx.x;
	for (int r = 0; r < nreads; r++) {
		const uint8_t *raw = iq8 + s * stride8 + (size_t)r * buf_len;
		long long p = 0, t = 0;
		for (int i = threadIdx.x; i < buf_len; i += 256) {
			int v = (int)raw[i] - 127;
			t += v;
			p += v * v;
		}
		rp[threadIdx.x] = p; rt[threadIdx.x] = t;
		__syncthreads();
		for (int off = 128; off > 0; off >>= 1) {
			if ((int)threadIdx.x < off) { rp[threadIdx.x] += rp[threadIdx.x + off]; rt[threadIdx.x] += rt[threadIdx.x + off]; }
			__syncthreads();
		}
		if (threadIdx.x == 0) {
			p = rp[0]; t = rt[0];
			double dc = (double)t / (double)buf_len;
			double err = (double)(t * 2) * dc - dc * dc * buf_len;
			p -= (long long)round(err);
			long long *a = avg + s;
			if (!peak_hold) *a += p;
			else if (p > *a) *a = p;
			samples[s] += 1;
		}
		__syncthreads();
	}
}

// ---- decimation ahead of the FFT (only when ds > 1) -----------------------------
// boxcar, src/rtl_power.c:671-681: int16-wrapping sums of ds complex samples
__global__ void __launch_bounds__(256) k_power_boxcar(const uint8_t *iq8, size_t stride8, int nreads, int buf_len,
                                                      int ds, int nstreams, int16_t *dec, size_t dec_stream_stride,
                                                      size_t dec_read_stride, int out_cplx)
{
	const int n = buf_len / 2;
	const size_t total = (size_t)nstreams * nreads * out_cplx;
	for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
		int k = (int)(g % out_cplx);
		size_t sr = g / out_cplx;
		int r = (int)(sr % nreads);
		size_t s = sr / nreads;
		const uint8_t *raw = iq8 + s * stride8 + (size_t)r * buf_len;
		int si = 0, sq = 0;
		for (int x = k * ds; x < (k + 1) * ds && x < n; x++) {
			si += (int)raw[2 * x] - 127;
			sq += (int)raw[2 * x + 1] - 127;
		}
		int16_t *d = dec + s * dec_stream_stride + (size_t)r * dec_read_stride;
		d[2 * k] = (int16_t)si;
		d[2 * k + 1] = (int16_t)sq;
	}
}

// One component value of the input of a fifth_order pass
struct ElemSrc {
	const uint8_t *raw;   // pass 0: u8
	const int16_t *i16;   // later passes
	__device__ __forceinline__ int at(int e) const { return raw ? (int)raw[e] - 127 : (int)i16[e]; }
};

// stateless fifth_order on I and Q (downsample_iq, src/rtl_power.c:554-579, 628-634):
// one thread per complex output; see the oracle for the ease-in / x5-twice pattern
__global__ void __launch_bounds__(256) k_power_fifth(const uint8_t *iq8, size_t stride8, const int16_t *in,
                                                     size_t in_stream_stride, size_t in_read_stride, int nreads,
                                                     int buf_len, int length /* elements going in */, int nstreams,
                                                     int16_t *out, size_t out_stream_stride, size_t out_read_stride)
{
	const int outs = length / 4;  // complex outputs
	const size_t total = (size_t)nstreams * nreads * outs;
	for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
		int m = (int)(g % outs);
		size_t sr = g / outs;
		int r = (int)(sr % nreads);
		size_t s = sr / nreads;
		ElemSrc src;
		src.raw = iq8 ? iq8 + s * stride8 + (size_t)r * buf_len : nullptr;
		src.i16 = in ? in + s * in_stream_stride + (size_t)r * in_read_stride : nullptr;
		int y[2];
		for (int comp = 0; comp < 2; comp++) {
			auto x = [&](int k) { return src.at(2 * k + comp); };
			int v;
			if (m == 0) v = ((x(0) + x(1)) * 10 + (x(2) + x(3)) * 5 + x(3) + x(5)) >> 4;
			else if (m == 1) v = ((x(1) + x(2)) * 10 + (x(0) + x(3)) * 5 + x(4) + x(5)) >> 4;
			else if (m == 2) v = (x(0) + (x(1) + x(4)) * 5 + (x(2) + x(3)) * 10 + x(5)) >> 4;
			else if (m == 3) v = (x(2) + (x(3) + x(5)) * 5 + (x(4) + x(5)) * 10 + x(6)) >> 4;
			else if (m == 4) v = (x(4) + (x(5) + x(7)) * 5 + (x(5) + x(6)) * 10 + x(8)) >> 4;
			else v = (x(2 * m - 5) + (x(2 * m - 4) + x(2 * m - 1)) * 5 + (x(2 * m - 3) + x(2 * m - 2)) * 10 + x(2 * m)) >> 4;
			y[comp] = (int16_t)v;
		}
		int16_t *d = out + s * out_stream_stride + (size_t)r * out_read_stride;
		d[2 * m] = (int16_t)y[0];
		d[2 * m + 1] = (int16_t)y[1];
	}
}

// stateless generic_fir, src/rtl_power.c:598-626: the first nine samples pass
__global__ void __launch_bounds__(256) k_power_fir9(const int16_t *in, int16_t *out, size_t stream_stride,
                                                    size_t read_stride, int nreads, int cplx, int nstreams, int passes)
{
	const size_t total = (size_t)nstreams * nreads * cplx;
	for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
		int n = (int)(g % cplx);
		size_t sr = g / cplx;
		int r = (int)(sr % nreads);
		size_t s = sr / nreads;
		const int16_t *x = in + s * stream_stride + (size_t)r * read_stride;
		int16_t *y = out + s * stream_stride + (size_t)r * read_stride;
		if (n < 9) { y[2 * n] = x[2 * n]; y[2 * n + 1] = x[2 * n + 1]; continue; }
		int hi[9], hq[9];
#pragma unroll
		for (int k = 0; k < 9; k++) { hi[k] = x[2 * (n - 9 + k)]; hq[k] = x[2 * (n - 9 + k) + 1]; }
		y[2 * n] = (int16_t)rtlfm::fir9_tap(hi, rtlfm::k_cic9[passes]);
		y[2 * n + 1] = (int16_t)rtlfm::fir9_tap(hq, rtlfm::k_cic9[passes]);
	}
}

// ---- -F in ONE launch (round 5): u8 -> (-127) -> downsample_iq x passes -> generic_fir -> packed int16 pairs ----
// downsample_iq (src/rtl_power.c:628-634) runs the stateless fifth_order (:554-579) over I and Q: output m of a pass
// needs the inputs 2m - 5 .. 2m of the pass before (the first five outputs of a READ are the ease-in forms below), so
// a tile of final outputs needs 5 (2^passes - 1) more input samples than it decimates - the passes run tile by tile
// in LDS, every level's samples as packed (I, Q) int16 pairs, ping-pong between two areas.  The general kernels
// (k_power_fifth per pass, k_power_fir9) went through HBM once per pass with 2-byte accesses: 57-59 Gsamples/s where
// the undecimated scans run 320-560 (profiles/r04_power_small.txt).
//   * passes 0-2 in packed 16-bit arithmetic: the taps sum to 32, the bytes span -127 .. 128, so the sum of pass p is
//     at most 32 * 128 * 2^p <= 16384 for p <= 2 - three v_pk_add, two v_pk_mad, one v_pk_ashr per complex output;
//     later passes (1/8 of the data and less) in 32 bits, the reference's `int` arithmetic with its int16 stores;
//   * generic_fir (:598-626) on the last level: the first nine samples pass, sample n >= 9 is the tap sum over the
//     nine ORIGINAL samples n - 9 .. n - 1 (`hist` is fed from `temp`);
//   * output: dword n of read r of stream s = the decimated sample n - what k_power_scan_frames<E, true> loads.
struct DecimateParams {
	const uint8_t *iq8; size_t stride8;  // raw reads: 16-byte aligned rows, buf_len a multiple of 16
	int nreads, buf_len;
	int passes;                         // 1 .. kDecMaxPasses
	int fir;                            // 9: generic_fir with cic_9_tables[passes] behind the passes
	uint32_t *out; size_t out_stream_stride;  // dwords
	int out_per_read;                   // (buf_len / 2) >> passes
	int tile_out, tiles_per_read;
	size_t total_tiles;
};
constexpr int kDecMaxPasses = 6;
constexpr int kDecTileIn = 4096;        // input samples a tile decimates (before the halo)
// dwords of LDS: level 0 holds the tile's input samples, the passes' reach-back (five samples per pass, rounded to even
// indices), the filter's nine samples of the last level and the rounding to 16-byte pieces; level 1 half of that
constexpr int kDecLdsA = kDecTileIn + 6 * 63 + 9 * 64 + 64, kDecLdsB = kDecLdsA / 2 + 16;

__device__ __forceinline__ uint32_t fifth_pk(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e, uint32_t f)
{
	const pk16_t t1 = __builtin_bit_cast(pk16_t, b) + __builtin_bit_cast(pk16_t, e);
	const pk16_t t2 = __builtin_bit_cast(pk16_t, c) + __builtin_bit_cast(pk16_t, d);
	const pk16_t s0 = __builtin_bit_cast(pk16_t, a) + __builtin_bit_cast(pk16_t, f);
	const pk16_t five = {5, 5}, ten = {10, 10};
	return __builtin_bit_cast(uint32_t, (pk16_t)((t1 * five + s0 + t2 * ten) >> 4));
}
__device__ __forceinline__ uint32_t fifth_32(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e, uint32_t f)
{
	const iq16 A = unpack_iq(a), B = unpack_iq(b), C = unpack_iq(c), D = unpack_iq(d), E = unpack_iq(e), F = unpack_iq(f);
	const int vi = ((int)A.i + ((int)B.i + E.i) * 5 + ((int)C.i + D.i) * 10 + F.i) >> 4;
	const int vq = ((int)A.q + ((int)B.q + E.q) * 5 + ((int)C.q + D.q) * 10 + F.q) >> 4;
	return pack_iq((int16_t)vi, (int16_t)vq);
}

template <int P>  // passes
__global__ void __launch_bounds__(256) k_power_downsample_iq(const DecimateParams p)
{
	static_assert(P >= 1 && P <= kDecMaxPasses, "the LDS areas are sized for six passes");
	__shared__ __attribute__((aligned(16))) uint32_t bufA[kDecLdsA];
	__shared__ __attribute__((aligned(16))) uint32_t bufB[kDecLdsB];
	const int t = threadIdx.x;
	for (size_t tile = blockIdx.x; tile < p.total_tiles; tile += gridDim.x) {
		const int tl = (int)(tile % p.tiles_per_read);
		const size_t sr = tile / p.tiles_per_read;
		const int r = (int)(sr % p.nreads);
		const size_t s = sr / p.nreads;
		const int o0 = tl * p.tile_out, o1 = min(o0 + p.tile_out, p.out_per_read);
		// what each level must hold: lo[k] .. hi[k] - 1 of level k (level 0 = the input samples, level P = the last pass's outputs)
		int lo[P + 1], hi[P + 1];
		lo[P] = p.fir ? max(0, o0 - 9) : o0;
		hi[P] = o1;
#pragma unroll
		for (int k = P - 1; k >= 0; k--) {
			const int len = (p.buf_len / 2) >> k;
			lo[k] = max(0, 2 * lo[k + 1] - 5) & ~1;            // even: the pairs of a level sit on 8-byte boundaries
			hi[k] = min(len, max(2 * (hi[k + 1] - 1), 5) + 1);
		}
		lo[0] &= ~7;  // whole 16-byte pieces of the read
		__syncthreads();  // the tile before is done with the LDS areas
		// ---- level 0: the bytes, as packed (I - 127, Q - 127)
		{
			const uint8_t *raw = p.iq8 + s * p.stride8 + (size_t)r * p.buf_len;
			const int c0 = lo[0] >> 3, c1 = (hi[0] + 7) >> 3;
			for (int c = c0 + t; c < c1; c += 256) {
				const uint4 v = *reinterpret_cast<const uint4 *>(raw + (size_t)c * 16);
				const uint32_t d[4] = {v.x, v.y, v.z, v.w};
				uint32_t o[8];
				const pk16_t off = {127, 127};
#pragma unroll
				for (int q = 0; q < 4; q++) {
					o[2 * q] = __builtin_bit_cast(uint32_t, (pk16_t)(__builtin_bit_cast(pk16_t, __builtin_amdgcn_perm(0u, d[q], 0x0c010c00u)) - off));
					o[2 * q + 1] = __builtin_bit_cast(uint32_t, (pk16_t)(__builtin_bit_cast(pk16_t, __builtin_amdgcn_perm(0u, d[q], 0x0c030c02u)) - off));
				}
				uint4 *dst = reinterpret_cast<uint4 *>(bufA + (c - c0) * 8);
				dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
				dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
			}
		}
		__syncthreads();
		// ---- the passes: level k in `src` (index 0 = sample lo[k]) -> level k + 1 in `dst`
		uint32_t *src = bufA, *dst = bufB;
#pragma unroll
		for (int k = 0; k < P; k++) {
			const int b0 = lo[k], n0 = lo[k + 1], n1 = hi[k + 1];
			// the general form for every output (indices clamped at the level's first sample: outputs 0 .. 4 of a read
			// are overwritten below)
			for (int m = n0 + t; m < n1; m += 256) {
				const int i5 = max(2 * m - 5 - b0, 0);  // odd unless clamped: b32, two aligned b64, b32
				uint32_t a, b, c, d, e, f;
				if (2 * m - 5 >= b0) {
					a = src[i5];
					const uint2 bc = *reinterpret_cast<const uint2 *>(src + i5 + 1), de = *reinterpret_cast<const uint2 *>(src + i5 + 3);
					b = bc.x; c = bc.y; d = de.x; e = de.y;
					f = src[i5 + 5];
				} else {
					a = b = c = d = e = f = 0;  // m < 5 at the start of a read
				}
				dst[m - n0] = k < 3 ? fifth_pk(a, b, c, d, e, f) : fifth_32(a, b, c, d, e, f);
			}
			if (n0 < 5) {
				// ease-in (src/rtl_power.c:559-569 and the first two rounds of the loop, where d and e are both data[10]):
				// with x(k) = sample k of this level
				__syncthreads();
				if (t < 5 && t >= n0 && t < n1) {
					auto X = [&](int q) { return unpack_iq(src[q - b0]); };  // b0 == 0 here
					int vi, vq;
					const iq16 x0 = X(0), x1 = X(1), x2 = X(2), x3 = X(3), x4 = X(4), x5 = X(5);
					if (t == 0) { vi = ((x0.i + x1.i) * 10 + (x2.i + x3.i) * 5 + x3.i + x5.i) >> 4; vq = ((x0.q + x1.q) * 10 + (x2.q + x3.q) * 5 + x3.q + x5.q) >> 4; }
					else if (t == 1) { vi = ((x1.i + x2.i) * 10 + (x0.i + x3.i) * 5 + x4.i + x5.i) >> 4; vq = ((x1.q + x2.q) * 10 + (x0.q + x3.q) * 5 + x4.q + x5.q) >> 4; }
					else if (t == 2) { vi = (x0.i + (x1.i + x4.i) * 5 + (x2.i + x3.i) * 10 + x5.i) >> 4; vq = (x0.q + (x1.q + x4.q) * 5 + (x2.q + x3.q) * 10 + x5.q) >> 4; }
					else if (t == 3) { const iq16 x6 = X(6); vi = (x2.i + (x3.i + x5.i) * 5 + (x4.i + x5.i) * 10 + x6.i) >> 4; vq = (x2.q + (x3.q + x5.q) * 5 + (x4.q + x5.q) * 10 + x6.q) >> 4; }
					else { const iq16 x6 = X(6), x7 = X(7), x8 = X(8); vi = (x4.i + (x5.i + x7.i) * 5 + (x5.i + x6.i) * 10 + x8.i) >> 4; vq = (x4.q + (x5.q + x7.q) * 5 + (x5.q + x6.q) * 10 + x8.q) >> 4; }
					dst[t - n0] = pack_iq((int16_t)vi, (int16_t)vq);
				}
			}
			__syncthreads();
			uint32_t *tmp = src; src = dst; dst = tmp;
		}
		// ---- generic_fir and the store
		uint32_t *orow = p.out + s * p.out_stream_stride + (size_t)r * p.out_per_read;
		const int bP = lo[P];
		for (int n = o0 + t; n < o1; n += 256) {
			uint32_t v = src[n - bP];
			if (p.fir && n >= 9) {
				int hi_[9], hq_[9];
#pragma unroll
				for (int q = 0; q < 9; q++) { const iq16 x = unpack_iq(src[n - 9 + q - bP]); hi_[q] = x.i; hq_[q] = x.q; }
				v = pack_iq((int16_t)rtlfm::fir9_tap(hi_, rtlfm::k_cic9[P]), (int16_t)rtlfm::fir9_tap(hq_, rtlfm::k_cic9[P]));
			}
			orow[n] = v;
		}
	}
}

}  // namespace rtlpower
