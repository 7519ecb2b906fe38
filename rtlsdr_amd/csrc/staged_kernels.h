// staged_kernels.h — the pass-major ("staged") form of rtl_fm's chain.
//
// Every stage of full_demod() (reference src/rtl_fm.c:1179-1272) is its own
// launch over ALL streams and ALL queued blocks at once; intermediate data
// lives in HBM work buffers as packed int16 I,Q dwords.  Because a stage runs
// for every block before the next stage starts, the history a block needs from
// its predecessor is simply read from the predecessor's input in the work
// buffer; only block 0 reads the carried rtlfm_stream_state.  This form exists
// for every configuration the reference accepts and is the cross-check for the
// fused streaming kernel (fused_kernel.h), which is what the roofline is
// measured on.
//
// Layout: a work buffer holds, for stream s, `xstride` dwords; at decimation
// level p block b's N_p complex samples start at s*xstride + b*N_p (blocks stay
// contiguous at every level).  State is double-buffered: kernels read `sin`,
// write `sout` (which starts as a copy of `sin`).
#pragma once

#include "dsp_device.h"

#include <type_traits>

#ifndef RTLFM_TAIL_PRIO
#define RTLFM_TAIL_PRIO 3
#endif
namespace rtlfm {

using state_t = rtlfm_stream_state;

#define RTLFM_GRID_STRIDE(i, n) \
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)(n); i += (size_t)gridDim.x * blockDim.x)

// ---------------------------------------------------------------- convert ----
// u8 -> int16 (-127) (src/rtl_fm.c:1326-1328), optional raw DC subtraction
// (:1330-1332, second half of dc_block_raw_filter :1058-1061), rotate16_neg90
// (:424-434) whose phase restarts with every buffer.  One thread per 16 input
// bytes = 8 complex samples.
__global__ void __launch_bounds__(256)
k_convert(const uint8_t *__restrict__ iq, size_t stream_stride, uint32_t L, int nblocks, int nstreams,
          uint32_t *__restrict__ X, size_t xstride, int rotate, const int2 *__restrict__ rdc_avg)
{
	const size_t per_block = L / 16;
	const size_t total = (size_t)nstreams * nblocks * per_block;
	RTLFM_GRID_STRIDE(g, total) {
		size_t j = g % per_block;
		size_t sb = g / per_block;
		int b = (int)(sb % nblocks);
		size_t s = sb / nblocks;
		const uint4 raw = *reinterpret_cast<const uint4 *>(iq + s * stream_stride + (size_t)b * L + j * 16);
		uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
		int ai = 0, aq = 0;
		if (rdc_avg) {
			int2 a = rdc_avg[s * nblocks + b];
			ai = a.x; aq = a.y;
		}
		uint32_t o[8];
#pragma unroll
		for (int k = 0; k < 8; k++) {
			uint32_t pair = (w[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
			int16_t re = (int16_t)((int)(pair & 0xff) - 127 - ai);
			int16_t im = (int16_t)((int)(pair >> 8) - 127 - aq);
			int16_t orr = re, oi = im;
			if (rotate) {
				switch (k & 3) {
				case 1: orr = im; oi = (int16_t)(-re); break;
				case 2: orr = (int16_t)(-re); oi = (int16_t)(-im); break;
				case 3: orr = (int16_t)(-im); oi = re; break;
				default: break;
				}
			}
			o[k] = pack_iq(orr, oi);
		}
		uint4 *dst = reinterpret_cast<uint4 *>(X + s * xstride + (size_t)b * (L / 2) + j * 8);
		dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
		dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
	}
}

// First half of dc_block_raw_filter (src/rtl_fm.c:1049-1055): per (stream,
// block) sums of I-127 and Q-127.  One workgroup per (stream, block).
__global__ void __launch_bounds__(256)
k_rdc_sums(const uint8_t *__restrict__ iq, size_t stream_stride, uint32_t L, int nblocks,
           long long *__restrict__ sums /* [s][b][2] */)
{
	const int sb = blockIdx.x;
	const int b = sb % nblocks;
	const size_t s = sb / nblocks;
	const uint8_t *src = iq + s * stream_stride + (size_t)b * L;
	long long si = 0, sq = 0;
	for (uint32_t k = threadIdx.x * 4; k < L; k += blockDim.x * 4) {
		uint32_t w = *reinterpret_cast<const uint32_t *>(src + k);
		si += (int)(w & 0xff) - 127 + (int)((w >> 16) & 0xff) - 127;
		sq += (int)((w >> 8) & 0xff) - 127 + (int)(w >> 24) - 127;
	}
	__shared__ long long red[2][256];
	red[0][threadIdx.x] = si;
	red[1][threadIdx.x] = sq;
	__syncthreads();
	for (int off = 128; off > 0; off >>= 1) {
		if ((int)threadIdx.x < off) {
			red[0][threadIdx.x] += red[0][threadIdx.x + off];
			red[1][threadIdx.x] += red[1][threadIdx.x + off];
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		sums[(size_t)sb * 2] = red[0][0];
		sums[(size_t)sb * 2 + 1] = red[1][0];
	}
}

// The same sums at streaming speed, for the fused front end's -E rdc path: one workgroup per
// (buffer, stream), 16-byte loads, v_dot4 against (1, 0, 1, 0) / (0, 1, 0, 1) byte masks (u8 sums;
// the -127 per sample is taken off once at the end), wave reduction by DPP.
__global__ void __launch_bounds__(256)
k_rdc_sums_wide(const uint8_t *__restrict__ iq, size_t stream_stride, uint32_t L, int nblocks, long long *__restrict__ sums /* [s][b][2] */)
{
	// (stream, buffer) folded into grid.x: grid.y stops at 65535 streams
	const size_t s = blockIdx.x / (unsigned)nblocks;
	const int b = (int)(blockIdx.x % (unsigned)nblocks);
	const uint8_t *src = iq + s * stream_stride + (size_t)b * L;
	unsigned si = 0, sq = 0;  // <= 262144 * 255 / 2: fits
	const uint32_t n16 = L / 16;
	auto add16 = [&](const uint4 &v) {
		si = __builtin_amdgcn_udot4(v.x, 0x00010001u, si, false); sq = __builtin_amdgcn_udot4(v.x, 0x01000100u, sq, false);
		si = __builtin_amdgcn_udot4(v.y, 0x00010001u, si, false); sq = __builtin_amdgcn_udot4(v.y, 0x01000100u, sq, false);
		si = __builtin_amdgcn_udot4(v.z, 0x00010001u, si, false); sq = __builtin_amdgcn_udot4(v.z, 0x01000100u, sq, false);
		si = __builtin_amdgcn_udot4(v.w, 0x00010001u, si, false); sq = __builtin_amdgcn_udot4(v.w, 0x01000100u, sq, false);
	};
	// eight 16-byte loads of the lane in flight (2 KiB of whole lines per wave and instruction), non-temporal: the front
	// end reads the same bytes again right behind this pass and nothing here is read twice
	typedef uint32_t v4u __attribute__((ext_vector_type(4)));
	const v4u *src4 = reinterpret_cast<const v4u *>(src);
	uint32_t k = threadIdx.x;
	for (; k + 7 * 256 < n16; k += 8 * 256) {
		v4u v[8];
#pragma unroll
		for (int j = 0; j < 8; j++) v[j] = __builtin_nontemporal_load(src4 + k + 256 * j);
#pragma unroll
		for (int j = 0; j < 8; j++) add16(make_uint4(v[j].x, v[j].y, v[j].z, v[j].w));
	}
	for (; k < n16; k += 256) add16(reinterpret_cast<const uint4 *>(src)[k]);
	for (uint32_t k = n16 * 16 + threadIdx.x * 2; k < L; k += 512) { si += src[k]; sq += src[k + 1]; }
	for (int off = 32; off > 0; off >>= 1) {
		si += __shfl_down(si, off, 64);
		sq += __shfl_down(sq, off, 64);
	}
	__shared__ unsigned red[2][4];
	const int w = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 0) { red[0][w] = si; red[1][w] = sq; }
	__syncthreads();
	if (threadIdx.x == 0) {
		const long long pairs = L / 2;
		sums[(s * nblocks + b) * 2] = (long long)(red[0][0] + red[0][1] + red[0][2] + red[0][3]) - 127 * pairs;
		sums[(s * nblocks + b) * 2 + 1] = (long long)(red[1][0] + red[1][1] + red[1][2] + red[1][3]) - 127 * pairs;
	}
}

// The same for buffers shorter than a tile of the front ends (-W 1 ... 15: 512 ... 7680 bytes; round 6): a WAVE per
// (stream, buffer), four to a workgroup, no barrier - a workgroup of 256 threads per 512-byte buffer was a million
// workgroups per 4 GiB whose launch alone took longer than the front end.
__global__ void __launch_bounds__(256)
k_rdc_sums_small(const uint8_t *__restrict__ iq, size_t stream_stride, uint32_t L, int nblocks, size_t total, long long *__restrict__ sums /* [s][b][2] */)
{
	const size_t sb = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (sb >= total) return;
	const int lane = threadIdx.x & 63;
	const size_t s = sb / (unsigned)nblocks;
	const int b = (int)(sb % (unsigned)nblocks);
	const uint8_t *src = iq + s * stream_stride + (size_t)b * L;
	unsigned si = 0, sq = 0;
	const uint32_t n16 = L / 16;  // L is a multiple of 512
	typedef uint32_t v4u __attribute__((ext_vector_type(4)));
	const v4u *src4 = reinterpret_cast<const v4u *>(src);
	for (uint32_t k = lane; k < n16; k += 64) {
		const v4u v = __builtin_nontemporal_load(src4 + k);
		si = __builtin_amdgcn_udot4(v.x, 0x00010001u, si, false); sq = __builtin_amdgcn_udot4(v.x, 0x01000100u, sq, false);
		si = __builtin_amdgcn_udot4(v.y, 0x00010001u, si, false); sq = __builtin_amdgcn_udot4(v.y, 0x01000100u, sq, false);
		si = __builtin_amdgcn_udot4(v.z, 0x00010001u, si, false); sq = __builtin_amdgcn_udot4(v.z, 0x01000100u, sq, false);
		si = __builtin_amdgcn_udot4(v.w, 0x00010001u, si, false); sq = __builtin_amdgcn_udot4(v.w, 0x01000100u, sq, false);
	}
	for (int off = 32; off > 0; off >>= 1) {
		si += __shfl_down(si, off, 64);
		sq += __shfl_down(sq, off, 64);
	}
	if (lane == 0) {
		const long long pairs = L / 2;
		sums[sb * 2] = (long long)si - 127 * pairs;
		sums[sb * 2 + 1] = (long long)sq - 127 * pairs;
	}
}


// The smoothing recurrence of dc_block_raw_filter (src/rtl_fm.c:1054-1057, :1062-1063), sequential over a stream's
// buffers.  One WAVE per stream (round 6): the block means - a 64-bit division each, independent of each other - are taken
// by 64 lanes at a time, and only the recurrence itself, avg = (mean + avg k) / (k + 1), runs down one lane, out of LDS,
// with the division by the fixed k + 1 as sdiv_trunc.  (Until then one lane per stream did everything, a dependent load and
// four divisions a step: 1.2 us per buffer - 5 ms of the 5.8 ms of `-E rdc` on 4096-byte buffers, 0.3 ms per 256 buffers.)
__global__ void __launch_bounds__(64)
k_rdc_smooth(const long long *__restrict__ sums, uint32_t L, int nblocks, int nstreams,
             int k, const state_t *__restrict__ sin, state_t *__restrict__ sout,
             int2 *__restrict__ avg)
{
	const size_t s = blockIdx.x;
	if (s >= (size_t)nstreams) return;
	const int lane = (int)threadIdx.x;
	__shared__ int2 m[64];
	int pi = sin[s].dc_avgI, pq = sin[s].dc_avgQ;
	const int pairs = (int)(L / 2);
	for (int base = 0; base < nblocks; base += 64) {
		const int b = base + lane;
		if (b < nblocks) {
			m[lane] = make_int2((int)(sums[(s * nblocks + b) * 2] / pairs), (int)(sums[(s * nblocks + b) * 2 + 1] / pairs));
		}
		__syncthreads();
		if (lane == 0) {
			const int cnt = nblocks - base < 64 ? nblocks - base : 64;
			for (int t = 0; t < cnt; t++) {
				const int2 v = m[t];
				const int ni = (int)((uint32_t)v.x + (uint32_t)pi * (uint32_t)k), nq = (int)((uint32_t)v.y + (uint32_t)pq * (uint32_t)k);
				pi = k + 1 != 0 ? sdiv_trunc(ni, k + 1) : 0;
				pq = k + 1 != 0 ? sdiv_trunc(nq, k + 1) : 0;
				m[t] = make_int2(pi, pq);
			}
		}
		__syncthreads();
		if (b < nblocks) avg[s * nblocks + b] = m[lane];
		__syncthreads();
	}
	if (lane == 0) {
		sout[s].dc_avgI = pi;
		sout[s].dc_avgQ = pq;
	}
}

// ------------------------------------------------------------ fifth_order ----
// e_b[idx] for idx < 0 is the history a block starts with: hist[6+idx] of the
// state for block 0, otherwise sample N-1+idx of the previous block — the
// archive holds x[N-7..N-2] (src/rtl_fm.c:800-805), the final sample x[N-1] is
// never kept.
__device__ __forceinline__ iq16 fifth_fetch(const uint32_t *__restrict__ Xs, int N, int b, int idx,
                                            const state_t *__restrict__ st, int p)
{
	while (idx < 0) {
		if (b == 0) {
			iq16 r;
			r.i = st->lp_i_hist[p][6 + idx];
			r.q = st->lp_q_hist[p][6 + idx];
			return r;
		}
		b -= 1;
		idx = N - 1 + idx;
	}
	return unpack_iq(Xs[(size_t)b * N + idx]);
}

// One pass of fifth_order on I and Q (src/rtl_fm.c:1188-1191, 777-806): one
// thread per complex output.  N = complex samples per block going in.
__global__ void __launch_bounds__(256)
k_fifth(const uint32_t *__restrict__ X, uint32_t *__restrict__ Y, size_t xstride, int N, int nblocks,
        int nstreams, int p, const state_t *__restrict__ sin, state_t *__restrict__ sout)
{
	const int M = N / 2;
	// a workgroup takes 256 consecutive outputs of one (stream, buffer): its place is found once per workgroup with
	// 32-bit divisions of wave-uniform numbers (round 5: two 64-bit divisions per OUTPUT made this kernel and k_fir9
	// cost 0.22 ms per step of a seven-pass chain - on 1 / 64 of the data)
	const unsigned chunks = (unsigned)(M + 255) / 256u;
	const unsigned long long nwg = (unsigned long long)nstreams * nblocks * chunks;
	for (unsigned long long wg = blockIdx.x; wg < nwg; wg += gridDim.x) {
		const unsigned sb = (unsigned)(wg / chunks), ch = (unsigned)(wg - (unsigned long long)sb * chunks);
		const int b = (int)(sb % (unsigned)nblocks);
		const size_t s = sb / (unsigned)nblocks;
		const int m = (int)(ch * 256u + threadIdx.x);
		if (m >= M) continue;
		const uint32_t *Xs = X + s * xstride;
		iq16 e[6];
		if (m >= 3) {
			const uint32_t *src = Xs + (size_t)b * N + 2 * m - 5;
#pragma unroll
			for (int k = 0; k < 6; k++) e[k] = unpack_iq(src[k]);
		} else {
#pragma unroll
			for (int k = 0; k < 6; k++) e[k] = fifth_fetch(Xs, N, b, 2 * m - 5 + k, &sin[s], p);
		}
		int yi = fifth_tap(e[0].i, e[1].i, e[2].i, e[3].i, e[4].i, e[5].i);
		int yq = fifth_tap(e[0].q, e[1].q, e[2].q, e[3].q, e[4].q, e[5].q);
		Y[s * xstride + (size_t)b * M + m] = pack_iq((int16_t)yi, (int16_t)yq);
		if (b == nblocks - 1 && m == M - 1) {
			// archive: the window of the last output (src/rtl_fm.c:800-805)
#pragma unroll
			for (int k = 0; k < 6; k++) {
				sout[s].lp_i_hist[p][k] = e[k].i;
				sout[s].lp_q_hist[p][k] = e[k].q;
			}
		}
	}
}

// ------------------------------------------------------------ generic_fir ----
// Output t is the 9-tap sum over samples t-9..t-1 of the stream (history is
// continuous across blocks, src/rtl_fm.c:808-831).  T = samples per stream.
__global__ void __launch_bounds__(256)
k_fir9(const uint32_t *__restrict__ X, uint32_t *__restrict__ Y, size_t xstride, int T, int nstreams,
       int passes, const state_t *__restrict__ sin, state_t *__restrict__ sout)
{
	const unsigned chunks = (unsigned)(T + 255) / 256u;
	const unsigned long long nwg = (unsigned long long)nstreams * chunks;
	for (unsigned long long wg = blockIdx.x; wg < nwg; wg += gridDim.x) {
		const size_t s = (size_t)(wg / chunks);
		const int t = (int)((unsigned)(wg - (unsigned long long)s * chunks) * 256u + threadIdx.x);
		if (t >= T) continue;
		const uint32_t *Xs = X + s * xstride;
		int hi[9], hq[9];
#pragma unroll
		for (int k = 0; k < 9; k++) {
			int idx = t - 9 + k;
			if (idx >= 0) {
				iq16 v = unpack_iq(Xs[idx]);
				hi[k] = v.i; hq[k] = v.q;
			} else {
				hi[k] = sin[s].droop_i_hist[9 + idx];
				hq[k] = sin[s].droop_q_hist[9 + idx];
			}
		}
		int yi = fir9_tap(hi, k_cic9[passes]);
		int yq = fir9_tap(hq, k_cic9[passes]);
		Y[s * xstride + t] = pack_iq((int16_t)yi, (int16_t)yq);
		if (t == T - 1) {
			iq16 cur = unpack_iq(Xs[t]);
#pragma unroll
			for (int k = 0; k < 8; k++) {
				sout[s].droop_i_hist[k] = (int16_t)hi[k + 1];
				sout[s].droop_q_hist[k] = (int16_t)hq[k + 1];
			}
			sout[s].droop_i_hist[8] = cur.i;
			sout[s].droop_q_hist[8] = cur.q;
		}
	}
}

// ---------------------------------------------------------------- low_pass ----
// Boxcar sum of `D` complex samples (src/rtl_fm.c:461-481) over the stream's
// whole run: with p0 samples already accumulated in (now_r, now_j), output k
// covers run samples [k*D - p0, (k+1)*D - p0).  T = run length in samples.
// cnt[s] receives the number of outputs.
__global__ void __launch_bounds__(256)
k_boxcar(const uint32_t *__restrict__ X, uint32_t *__restrict__ Y, size_t xstride, int T, int nstreams,
         int D, const state_t *__restrict__ sin, state_t *__restrict__ sout, int32_t *__restrict__ cnt)
{
	const int maxout = T / D + 1;
	const size_t total = (size_t)nstreams * (maxout + 1);
	RTLFM_GRID_STRIDE(g, total) {
		int k = (int)(g % (maxout + 1));
		size_t s = g / (maxout + 1);
		const int p0 = sin[s].prev_index;
		const int E = (p0 + T) / D;
		const uint32_t *Xs = X + s * xstride;
		if (k < E) {
			int lo = k * D - p0, hi = lo + D;
			uint32_t ar = 0, aj = 0;
			if (k == 0) { ar = (uint32_t)sin[s].now_r; aj = (uint32_t)sin[s].now_j; lo = 0; }
			for (int n = lo; n < hi; n++) {
				iq16 v = unpack_iq(Xs[n]);
				ar += (uint32_t)(int)v.i; aj += (uint32_t)(int)v.q;
			}
			Y[s * xstride + k] = pack_iq((int16_t)(int)ar, (int16_t)(int)aj);
		} else if (k == E) {
			// the partial sum that stays behind
			int lo = E * D - p0;
			uint32_t ar = 0, aj = 0;
			if (E == 0) { ar = (uint32_t)sin[s].now_r; aj = (uint32_t)sin[s].now_j; lo = 0; }
			for (int n = lo; n < T; n++) {
				iq16 v = unpack_iq(Xs[n]);
				ar += (uint32_t)(int)v.i; aj += (uint32_t)(int)v.q;
			}
			sout[s].now_r = (int)ar;
			sout[s].now_j = (int)aj;
			sout[s].prev_index = p0 + T - E * D;
			cnt[s] = E;
		}
	}
}

// Which block does decimated sample t belong to, and is it that block's first?
// Blocks of N input samples each; boxcar D with p0 pre-accumulated (D == 1,
// p0 == 0 describes the fifth_order path with N already the decimated size).
__device__ __forceinline__ int dec_block_of(int t, int N, int D, int p0)
{
	return (int)((((long long)t + 1) * D - p0 - 1) / N);
}
__device__ __forceinline__ bool dec_block_first(int t, int N, int D, int p0)
{
	if (t == 0) return true;
	return dec_block_of(t - 1, N, D, p0) < dec_block_of(t, N, D, p0);
}
// first decimated index of block b / one past its last
__device__ __forceinline__ int dec_block_begin(int b, int N, int D, int p0)
{
	return (int)(((long long)p0 + (long long)b * N) / D);
}

// ----------------------------------------------------------------- squelch ----
// rms() of a block's decimated samples (src/rtl_fm.c:1083-1112) and the
// squelch decision (:1204-1215).  One workgroup per (stream, block); writes
// mute[s*nblocks+b] = 1 when the block is to be zeroed.
__global__ void __launch_bounds__(256)
k_squelch_rms(const uint32_t *__restrict__ X, size_t xstride, int N, int D, int nblocks,
              const state_t *__restrict__ sin, int level, int omit_dc_fix, int32_t *__restrict__ mute,
              int32_t *__restrict__ rms_out)
{
	const int sb = blockIdx.x;
	const int b = sb % nblocks;
	const size_t s = sb / nblocks;
	const int p0 = D > 1 ? sin[s].prev_index : 0;
	const int t0 = dec_block_begin(b, N, D, p0), t1 = dec_block_begin(b + 1, N, D, p0);
	const int16_t *lp = reinterpret_cast<const int16_t *>(X + s * xstride + t0);
	const int len = 2 * (t1 - t0);
	int step = 1;
	while (len > step * 32768) ++step;
	uint32_t p = 0;
	int32_t t = 0;
	if (step == 1) {
		// both sums are taken modulo 2^32 (the reference's uint32 p wraps, :1093-1098), so their order is free: a packed
		// (I, Q) pair per lane, v_dot2 for the squares and for I + Q
		const uint32_t *z = X + s * xstride + t0;
		typedef short s2_t __attribute__((ext_vector_type(2)));
		const s2_t ones = {(short)1, (short)1};
		for (int i = threadIdx.x; i < t1 - t0; i += blockDim.x) {
			const s2_t v = __builtin_bit_cast(s2_t, z[i]);
			p = (uint32_t)__builtin_amdgcn_sdot2(v, v, (int)p, false);
			t = __builtin_amdgcn_sdot2(v, ones, t, false);
		}
	} else
	for (int i = threadIdx.x * step; i < len; i += blockDim.x * step) {
		int v = lp[i];
		t += v;
		p += (uint32_t)(v * v);
	}
	__shared__ uint32_t rp[256];
	__shared__ int32_t rt[256];
	rp[threadIdx.x] = p; rt[threadIdx.x] = t;
	__syncthreads();
	for (int off = 128; off > 0; off >>= 1) {
		if ((int)threadIdx.x < off) {
			rp[threadIdx.x] += rp[threadIdx.x + off];
			rt[threadIdx.x] = (int32_t)((uint32_t)rt[threadIdx.x] + (uint32_t)rt[threadIdx.x + off]);
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		p = rp[0]; t = rt[0];
		double r;
		if (omit_dc_fix) {
			int num = len / step;
			r = sqrt((double)p / num);
		} else {
			double dc = (double)(int32_t)((uint32_t)t * (uint32_t)step) / (double)len;
			double err = t * 2 * dc - dc * dc * len;
			r = sqrt((p - err) / len);
		}
		// the uint32 sum of squares wraps for large samples, p - err can go negative and the
		// reference's (int)sqrt() of it is x86's "integer indefinite" INT_MIN, which the squelch
		// skips (sr >= 0, src/rtl_fm.c:1206); v_cvt_i32_f64 would turn the NaN into 0
		const int sr = r == r ? (int)r : INT32_MIN;
		mute[sb] = (sr >= 0 && sr < level) ? 1 : (sr >= 0 ? 0 : 2);
		rms_out[sb] = sr;
	}
}

// squelch_hits bookkeeping, sequential over a stream's blocks.
__global__ void k_squelch_hits(const int32_t *__restrict__ mute, int nblocks, int nstreams,
                               const state_t *__restrict__ sin, state_t *__restrict__ sout)
{
	RTLFM_GRID_STRIDE(s, nstreams) {
		int hits = sin[s].squelch_hits;
		for (int b = 0; b < nblocks; b++) {
			int m = mute[s * nblocks + b];
			if (m == 1) hits++;
			else if (m == 0) hits = 0;
		}
		sout[s].squelch_hits = hits;
	}
}

// a workgroup per (stream, block): nothing but a look at the flag for the blocks that stay
__global__ void __launch_bounds__(256)
k_squelch_zero(uint32_t *__restrict__ X, size_t xstride, int N, int D, int nblocks, int nstreams, int T,
               const state_t *__restrict__ sin, const int32_t *__restrict__ mute)
{
	const int sb = blockIdx.x;
	if (mute[sb] != 1) return;
	const int b = sb % nblocks;
	const size_t s = sb / nblocks;
	const int p0 = D > 1 ? sin[s].prev_index : 0;
	const int t0 = dec_block_begin(b, N, D, p0), t1 = dec_block_begin(b + 1, N, D, p0);
	uint32_t *x = X + s * xstride;
	for (int t = t0 + (int)threadIdx.x; t < t1 && t < T; t += 256) x[t] = 0;
}

// The squelch behind a front end that has taken rms()'s sums itself (boxcar_kernel.h, SQ): a workgroup per stream.
// sums[s][b] = (sum of squares, sum) of buffer b's decimated elements modulo 2^32; the level and the decision are
// rms()'s and full_demod()'s (src/rtl_fm.c:1083-1112, 1204-1215), squelch_hits counts over the stream's buffers in
// order, and a muted buffer's PCM is written as what the demodulators make of zeroed samples: zeros - and a zero for
// the first output behind it too where fm_demod pairs it with the zeroed last sample (atan2(0, 0) = 0), which for the
// run's last buffer is the carried pre_r / pre_j.
__global__ void __launch_bounds__(256)
k_squelch_apply(const uint32_t *__restrict__ sums, int16_t *__restrict__ R, size_t rstride, int N, int D, int nblocks, int nstreams,
                int level, int omit_dc_fix, int fm, const int32_t *__restrict__ cnt, int T, const state_t *__restrict__ sin,
                state_t *__restrict__ sout, int32_t *__restrict__ rms_out)
{
	extern __shared__ int32_t sq_mute[];  // [nblocks]
	const size_t s = blockIdx.x;
	const int p0 = D > 1 ? sin[s].prev_index : 0;
	const int Ts = cnt ? cnt[s] : T;
	for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
		const int t0 = dec_block_begin(b, N, D, p0), t1 = dec_block_begin(b + 1, N, D, p0);
		const int len = 2 * (t1 - t0);
		const uint32_t p = sums[(s * nblocks + b) * 2];
		const int32_t t = (int32_t)sums[(s * nblocks + b) * 2 + 1];
		int m = 2;  // 2: no decision (an empty buffer, or a level that is not a number)
		int sr = INT32_MIN;
		if (len > 0) {
			double r;
			if (omit_dc_fix) r = sqrt((double)p / len);
			else {
				const double dc = (double)t / (double)len;
				const double err = t * 2 * dc - dc * dc * len;
				r = sqrt((p - err) / len);
			}
			sr = r == r ? (int)r : INT32_MIN;  // (int)NaN is INT_MIN on the reference's machine: the squelch skips it (sr >= 0)
			m = (sr >= 0 && sr < level) ? 1 : (sr >= 0 ? 0 : 2);
		}
		sq_mute[b] = level ? m : 0;
		if (rms_out) rms_out[s * nblocks + b] = sr;
	}
	__syncthreads();
	if (threadIdx.x == 0 && level) {
		int hits = sin[s].squelch_hits;
		for (int b = 0; b < nblocks; b++) {
			if (sq_mute[b] == 1) hits++;
			else if (sq_mute[b] == 0) hits = 0;
		}
		sout[s].squelch_hits = hits;
		if (fm && sq_mute[nblocks - 1] == 1 && dec_block_begin(nblocks, N, D, p0) > dec_block_begin(nblocks - 1, N, D, p0)) {
			sout[s].pre_r = 0; sout[s].pre_j = 0;  // fm_demod kept the zeroed last sample (:955-956)
		}
	}
	if (!level) return;
	int16_t *r = R + s * rstride;
	for (int b = 0; b < nblocks; b++) {
		if (sq_mute[b] != 1) continue;
		const int t0 = dec_block_begin(b, N, D, p0);
		int t1 = dec_block_begin(b + 1, N, D, p0) + (fm ? 1 : 0);  // + the first output of the next buffer
		if (t1 > Ts) t1 = Ts;
		for (int t = t0 + (int)threadIdx.x; t < t1; t += blockDim.x) r[t] = 0;
	}
}

// ------------------------------------------------------------------ demods ----
// fm_demod (src/rtl_fm.c:932-959): output t pairs sample t with t-1 (or with
// the carried pre_r/pre_j); the FIRST output of every block always uses
// polar_discriminant.  cnt == nullptr: every stream has T samples.
// A workgroup per (stream, block): the block's extent costs two divisions per workgroup instead of four 64-bit ones per
// sample (the per-sample form was 1.24 ms for the 214 M decimated samples of a /10 run of 4 GiB - more than the
// decimator in front of it).
__global__ void __launch_bounds__(256)
k_fm_demod(const uint32_t *__restrict__ X, size_t xstride, int16_t *__restrict__ R, size_t rstride,
           int T, int nstreams, int N, int D, int nblocks, int variant, const int32_t *__restrict__ lut,
           const int32_t *__restrict__ cnt, const state_t *__restrict__ sin, state_t *__restrict__ sout)
{
	const int sb = blockIdx.x;
	const int b = sb % nblocks;
	const size_t s = sb / nblocks;
	const int Ts = cnt ? cnt[s] : T;
	const int p0 = D > 1 ? sin[s].prev_index : 0;
	const int t0 = dec_block_begin(b, N, D, p0);
	int t1 = dec_block_begin(b + 1, N, D, p0);
	if (t1 > Ts) t1 = Ts;
	const uint32_t *Xs = X + s * xstride;
	int16_t *Rs = R + s * rstride;
	for (int t = t0 + (int)threadIdx.x; t < t1; t += 256) {
		const iq16 cur = unpack_iq(Xs[t]);
		int br, bj;
		if (t > 0) {
			const iq16 pv = unpack_iq(Xs[t - 1]);
			br = pv.i; bj = pv.q;
		} else {
			br = sin[s].pre_r; bj = sin[s].pre_j;
		}
		const int v = t == t0 ? disc_std(cur.i, cur.q, br, bj) : discriminate(variant, cur.i, cur.q, br, bj, lut);
		Rs[t] = (int16_t)v;
		if (t == Ts - 1) {
			sout[s].pre_r = cur.i;
			sout[s].pre_j = cur.q;
		}
	}
}

// am_demod / usb_demod / lsb_demod / raw_demod (src/rtl_fm.c:961-1009)
__global__ void __launch_bounds__(256)
k_simple_demod(const uint32_t *__restrict__ X, size_t xstride, int16_t *__restrict__ R, size_t rstride,
               int T, int nstreams, int mode, int output_scale, const int32_t *__restrict__ cnt)
{
	const unsigned chunks = (unsigned)(T + 255) / 256u;
	const unsigned long long nwg = (unsigned long long)nstreams * chunks;
	for (unsigned long long wg = blockIdx.x; wg < nwg; wg += gridDim.x) {
		const size_t s = (size_t)(wg / chunks);
		const int t = (int)((unsigned)(wg - (unsigned long long)s * chunks) * 256u + threadIdx.x);
		const int Ts = cnt ? cnt[s] : T;
		if (t >= Ts) continue;
		uint32_t w = X[s * xstride + t];
		if (mode == RTLFM_MODE_RAW) {
			reinterpret_cast<uint32_t *>(R + s * rstride)[t] = w;
			continue;
		}
		R[s * rstride + t] = simple_demod(mode, w, output_scale);
	}
}

// ---- passes 6 .. P-1, generic_fir and mode_demod behind the six-pass front end, in ONE launch (round 5) --------
// `rtl_fm -s 12k -F 9` plans seven passes, `-s 3k -F 9` nine: k_fused<6> emits the /64 IQ and round 4 finished with one
// k_fifth launch per pass, k_fir9 and k_fm_demod - 0.2 ms per step on 1 / 64 of the data.  Here a workgroup takes one
// (stream, buffer): the buffer's /64 samples and the last `kt` of the buffer before go into LDS, every pass runs there
// (level by level, ping-pong), then the filter and the demodulator.  What a buffer needs of its predecessor - the
// fifth_order history of every pass (x[N-7 .. N-2] of that level: the archive never holds x[N-1], src/rtl_fm.c:800-805),
// generic_fir's nine samples, fm_demod's previous sample - is RECOMPUTED from the predecessor's tail: outputs at negative
// indices, the general formula, no state of the predecessor's own start involved (the host checks that the tail is short
// against the buffer).  Buffer 0 takes them from the carried state, the last buffer leaves them there.
struct DeepRestParams {
	const uint32_t *X; size_t xstride;  // level 6, packed (I, Q): n6 samples per buffer, a stream's buffers back to back
	int n6, nblocks, nstreams, passes;  // passes: 7 .. 10 (6 .. passes - 1 run here)
	int fir, kt;                        // generic_fir behind them; samples of the buffer before taken along at level 6
	uint32_t *Y; size_t ystride;        // the decimated, FIR-compensated IQ (squelch, -L, -M raw behind it), or
	int16_t *R; size_t rstride;         // ... the PCM: fm / am / usb / lsb demodulated here (Y == nullptr)
	int mode, variant, output_scale;
	const int32_t *lut;
	const state_t *sin; state_t *sout;
};
// tail lengths per level for `passes` (index 0 = level 6): what the next level, the filter and the demodulator reach back
__host__ __device__ inline void deep_rest_tails(int passes, int fir, int k[6])
{
	const int L = passes - 6;
	k[L] = fir ? 10 : 1;
	for (int j = L - 1; j >= 0; j--) {
		k[j] = 2 * k[j + 1] + 5;
		if (k[j] < 6) k[j] = 6;
	}
}

__global__ void __launch_bounds__(256) k_deep_rest(const DeepRestParams p)
{
	extern __shared__ uint32_t dr_lds[];
	const int sb = blockIdx.x;
	const int b = sb % p.nblocks;
	const size_t s = sb / p.nblocks;
	const int tid = threadIdx.x;
	const state_t &in = p.sin[s];
	const bool first = b == 0, last = b == p.nblocks - 1;
	int kk[6];
	deep_rest_tails(p.passes, p.fir, kk);
	const int L = p.passes - 6;
	const int cap = p.kt + p.n6 + 8;
	uint32_t *A = dr_lds, *B = dr_lds + cap;
	// level 6: samples [-k, n6) of this buffer (negative: the buffer before), at A[i + k]
	int k = first ? 0 : kk[0], n = p.n6;
	{
		const uint32_t *x = p.X + s * p.xstride + (size_t)b * p.n6;
		for (int i = tid - k; i < n; i += 256) A[i + k] = x[i];
	}
	__syncthreads();
	for (int j = 0; j < L; j++) {
		const int pass = 6 + j;
		const int k2 = first ? 0 : kk[j + 1], n2 = n / 2;
		// sample idx of the level going in, as fifth_order sees it from output m >= 0 of this buffer: behind the buffer's
		// start the archive of the buffer before, x[N-1+idx] (the last sample is never kept)
		auto fetch = [&](int idx) -> iq16 {
			if (idx >= 0) return unpack_iq(A[idx + k]);
			if (!first) return unpack_iq(A[idx - 1 + k]);
			iq16 r; r.i = in.lp_i_hist[pass][6 + idx]; r.q = in.lp_q_hist[pass][6 + idx];
			return r;
		};
		for (int m = tid - k2; m < n2; m += 256) {
			iq16 e[6];
			if (m >= 3) {
#pragma unroll
				for (int q = 0; q < 6; q++) e[q] = unpack_iq(A[2 * m - 5 + q + k]);
			} else if (m >= 0) {
#pragma unroll
				for (int q = 0; q < 6; q++) e[q] = fetch(2 * m - 5 + q);
			} else {
				// an output of the buffer before (its own samples, no archive in between)
#pragma unroll
				for (int q = 0; q < 6; q++) e[q] = unpack_iq(A[2 * m - 5 + q + k]);
			}
			const int yi = fifth_tap(e[0].i, e[1].i, e[2].i, e[3].i, e[4].i, e[5].i);
			const int yq = fifth_tap(e[0].q, e[1].q, e[2].q, e[3].q, e[4].q, e[5].q);
			B[m + k2] = pack_iq((int16_t)yi, (int16_t)yq);
			if (last && m == n2 - 1) {
#pragma unroll
				for (int q = 0; q < 6; q++) { p.sout[s].lp_i_hist[pass][q] = e[q].i; p.sout[s].lp_q_hist[pass][q] = e[q].q; }
			}
		}
		__syncthreads();
		uint32_t *t = A; A = B; B = t;
		k = k2; n = n2;
	}
	// A: the last level, samples [-k, n).  generic_fir (src/rtl_fm.c:808-831): output t = taps over samples t-9 .. t-1
	if (p.fir) {
		auto xs = [&](int idx) -> iq16 {
			if (idx >= 0 || !first) return unpack_iq(A[idx + k]);
			iq16 r; r.i = in.droop_i_hist[9 + idx]; r.q = in.droop_q_hist[9 + idx];
			return r;
		};
		const int t_lo = first ? 0 : -1;  // (the output in front of the buffer: fm_demod's previous sample)
		for (int t = tid + t_lo; t < n; t += 256) {
			int hi[9], hq[9];
#pragma unroll
			for (int q = 0; q < 9; q++) { const iq16 v = xs(t - 9 + q); hi[q] = v.i; hq[q] = v.q; }
			B[t + 1] = pack_iq((int16_t)fir9_tap(hi, k_cic9[p.passes]), (int16_t)fir9_tap(hq, k_cic9[p.passes]));
			if (last && t == n - 1) {
				const iq16 cur = unpack_iq(A[t + k]);
				for (int q = 0; q < 8; q++) { p.sout[s].droop_i_hist[q] = (int16_t)hi[q + 1]; p.sout[s].droop_q_hist[q] = (int16_t)hq[q + 1]; }
				p.sout[s].droop_i_hist[8] = cur.i; p.sout[s].droop_q_hist[8] = cur.q;
			}
		}
		__syncthreads();
		uint32_t *t = A; A = B; B = t;
		k = 1;  // A[t + 1] = filtered sample t, t >= -1
	}
	// A[t + k]: what full_demod() hands to the squelch and mode_demod()
	if (p.Y) {
		uint32_t *y = p.Y + s * p.ystride + (size_t)b * n;
		for (int t = tid; t < n; t += 256) y[t] = A[t + k];
		return;
	}
	int16_t *r = p.R + s * p.rstride + (size_t)b * n;
	for (int t = tid; t < n; t += 256) {
		const uint32_t w = A[t + k];
		if (p.mode != RTLFM_MODE_FM) { r[t] = simple_demod(p.mode, w, p.output_scale); continue; }
		const iq16 cur = unpack_iq(w);
		int br, bj;
		if (t > 0 || !first) { const iq16 pv = unpack_iq(A[t - 1 + k]); br = pv.i; bj = pv.q; }
		else { br = in.pre_r; bj = in.pre_j; }
		// fm_demod (:932-959): the first output of a buffer is always polar_discriminant
		r[t] = (int16_t)(t == 0 ? disc_std(cur.i, cur.q, br, bj) : discriminate(p.variant, cur.i, cur.q, br, bj, p.lut));
		if (last && t == n - 1) { p.sout[s].pre_r = cur.i; p.sout[s].pre_j = cur.q; }
	}
}

// ---- fifth_order on lengths its passes do not divide (round 5; lane-parallel since round 6) --------------------
// `rtl_fm -W n -F 9` with nine or ten passes and n odd (n % 4 != 0 for ten) is something the reference runs
// (src/rtl_fm.c:1188-1191): pass 8 (or 9) is then called with a length that is not a multiple of four elements.  What
// the reference's loops then DO, as functions of the `lowpassed` array they are handed (derived from :777-831, not
// copied from them):
//   * one fifth_order call on (data, length, hist) works on the component x[k] = data[2k] and produces
//     count = max(1, ceil(length / 4)) outputs y[m] = (x[2m-5] + 5 x[2m-4] + 10 x[2m-3] + 10 x[2m-2] + 5 x[2m-1] + x[2m]) >> 4
//     into data[2m], x[-5..-1] = hist[1..5]; it reads every input before the position is overwritten (it writes data[2m]
//     and has then read up to data[4m]), so "all lanes read, then all lanes write" on the one array IS the loop; the archive
//     it leaves is x[2M-5 .. 2M] of its last output M.  The first output is unconditional - a call with length <= 0 (the Q
//     call of a one-element buffer) still reads data[0], writes it and moves the history on;
//   * the Q call (`lowpassed + 1, length - 1`) sees another count than the I call, the last passes stop decimating, and
//     everything behind them works on an ODD number of elements: generic_fir yields ceil(len / 2) outputs per call, each
//     the tap sum over the nine inputs BEFORE it (original values: the history holds them), rms() sums lp_len elements,
//     the demodulators pair lowpassed[i] with lowpassed[i + 1] - I with the Q of another sample, or with whatever the array
//     holds behind lp_len -, fm_demod's pre_r / pre_j are the array's last two elements.
// So the array itself, with what earlier passes and buffers left in it, is part of the semantics: it lives in LDS, one
// WAVE per stream walks the stream's buffers in order (the history chains them), and every stage is a lane-parallel
// pass over it - a lane per output, the repo's own window forms (fifth_tap, fir9_tap, discriminate) - with a barrier
// between its reads and its writes.  A buffer is down to at most 512 samples here.  The regular passes in front
// (k_fused emit / k_fifth) and the audio tail behind (run_tail, result_len = lp_len / 2 per buffer) are the ordinary kernels.
constexpr int kIrregularMaxElems = 1024 + 16;  // block_len >> 8 elements at most (RTLFM_MAX_BLOCK_LEN = 262144)
struct IrregularParams {
	const uint32_t *X; size_t xstride;  // packed (I, Q) of the level in front of pass `first`: n_in samples per buffer
	int n_in, nblocks, nstreams;
	int first, passes;                  // passes first .. passes - 1 run here
	int lp_len;                         // block_len >> passes: elements mode_demod() is given
	int fir, mode, variant, output_scale, squelch_level, report_levels, omit_dc_fix;
	const int32_t *lut;
	int16_t *R; size_t rstride;         // result: lp_len / 2 samples per buffer (raw: lp_len elements)
	int32_t *levels;                    // [stream][nblocks] rms() per buffer (squelch / -L), or nullptr
	const state_t *sin; state_t *sout;
};

__device__ __forceinline__ int wave_sum_i32(int v)
{
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) v = (int)((uint32_t)v + (uint32_t)__shfl_xor(v, off));
	return v;
}

__global__ void __launch_bounds__(64) k_fifth_irregular(const IrregularParams p)
{
	const int s = (int)blockIdx.x, lane = (int)threadIdx.x;  // one wave per stream
	if (s >= p.nstreams) return;
	__shared__ int16_t lp[kIrregularMaxElems + 8];
	__shared__ int16_t hist[2][RTLFM_MAX_PASSES][6];  // [I / Q][pass]: lp_i_hist / lp_q_hist
	__shared__ int16_t droop[2][9];
	__shared__ int carry[3];                           // pre_r, pre_j, squelch_hits
	const state_t &in = p.sin[s];
	state_t &out = p.sout[s];
	for (int k = lane; k < 6 * (p.passes - p.first); k += 64) {
		const int q = p.first + k / 6, j = k % 6;
		hist[0][q][j] = in.lp_i_hist[q][j]; hist[1][q][j] = in.lp_q_hist[q][j];
	}
	if (lane < 9) { droop[0][lane] = in.droop_i_hist[lane]; droop[1][lane] = in.droop_q_hist[lane]; }
	if (lane == 0) { carry[0] = in.pre_r; carry[1] = in.pre_j; carry[2] = in.squelch_hits; }
	for (int k = lane; k < kIrregularMaxElems + 8; k += 64) lp[k] = 0;
	__syncthreads();
	constexpr int kPer5 = (kIrregularMaxElems / 4 + 63) / 64 + 1;  // outputs of one fifth_order call per lane
	constexpr int kPer9 = (kIrregularMaxElems / 2 + 63) / 64 + 1;  // ... of one generic_fir call
	// one fifth_order call: component x[k] = data[2k], x[-5..-1] = h[1..5]
	auto fifth = [&](int16_t *data, int length, int16_t *h) {
		const int count = length > 0 ? (length + 3) / 4 : 1;
		auto X = [&](int k) -> int { return k < 0 ? (int)h[6 + k] : (int)data[2 * k]; };
		int y[kPer5];
#pragma unroll
		for (int r = 0; r < kPer5; r++) {
			const int m = lane + 64 * r;
			y[r] = m < count ? fifth_tap(X(2 * m - 5), X(2 * m - 4), X(2 * m - 3), X(2 * m - 2), X(2 * m - 1), X(2 * m)) : 0;
		}
		int16_t keep = 0;  // the archive: x[2M - 5 + lane] of the call's last output M, lanes 0..5
		if (lane < 6) keep = (int16_t)X(2 * (count - 1) - 5 + lane);
		__syncthreads();
#pragma unroll
		for (int r = 0; r < kPer5; r++) {
			const int m = lane + 64 * r;
			if (m < count) data[2 * m] = (int16_t)y[r];
		}
		if (lane < 6) h[lane] = keep;
		__syncthreads();
	};
	// one generic_fir call: output k (of ceil(length / 2)) = taps over x[k-9 .. k-1], x[-9..-1] = h[0..8]
	auto fir9 = [&](int16_t *data, int length, int16_t *h) {
		const int count = length > 0 ? (length + 1) / 2 : 0;
		auto X = [&](int k) -> int { return k < 0 ? (int)h[9 + k] : (int)data[2 * k]; };
		const int32_t *t = k_cic9[p.passes];
		int y[kPer9];
#pragma unroll
		for (int r = 0; r < kPer9; r++) {
			const int k = lane + 64 * r;
			int w[9];
#pragma unroll
			for (int j = 0; j < 9; j++) w[j] = k < count ? X(k - 9 + j) : 0;
			y[r] = fir9_tap(w, t);
		}
		int16_t keep = 0;  // the last nine of (history, x[0 .. count - 1])
		if (lane < 9) keep = (int16_t)X(count - 9 + lane);
		__syncthreads();
#pragma unroll
		for (int r = 0; r < kPer9; r++) {
			const int k = lane + 64 * r;
			if (k < count) data[2 * k] = (int16_t)y[r];
		}
		if (lane < 9) h[lane] = keep;
		__syncthreads();
	};
	const int len0 = 2 * p.n_in;  // elements pass `first` is given
	const int lp_len = p.lp_len;
	const int n_res = p.mode == RTLFM_MODE_RAW ? lp_len : lp_len / 2;
	for (int b = 0; b < p.nblocks; b++) {
		const uint32_t *x = p.X + (size_t)s * p.xstride + (size_t)b * p.n_in;
		for (int k = lane; k < p.n_in; k += 64) { const iq16 w = unpack_iq(x[k]); lp[2 * k] = w.i; lp[2 * k + 1] = w.q; }
		if (lane < 8) lp[len0 + lane] = 0;  // what the oracle's array holds behind a buffer's elements
		__syncthreads();
		for (int q = p.first; q < p.passes; q++) {  // :1188-1191
			const int length = len0 >> (q - p.first);
			fifth(lp, length, hist[0][q]);
			fifth(lp + 1, length - 1, hist[1][q]);
		}
		if (p.fir) {  // :1193-1199
			fir9(lp, lp_len, droop[0]);
			fir9(lp + 1, lp_len - 1, droop[1]);
		}
		if ((p.squelch_level || p.report_levels) && lp_len > 0) {
			// rms() over the buffer's lp_len elements (:1083-1112; at most 1024 here, so its step is 1) and the squelch, :1204-1215
			int t = 0, pw = 0;
			for (int i = lane; i < lp_len; i += 64) { const int v = lp[i]; t = (int)((uint32_t)t + (uint32_t)v); pw = (int)((uint32_t)pw + (uint32_t)(v * v)); }
			t = wave_sum_i32(t);
			const uint32_t pwu = (uint32_t)wave_sum_i32(pw);
			double r;
			if (p.omit_dc_fix) r = sqrt((double)pwu / lp_len);
			else {
				const double dc = (double)t / (double)lp_len;
				const double err = t * 2 * dc - dc * dc * lp_len;
				r = sqrt((pwu - err) / lp_len);
			}
			const int sr = r == r ? (int)r : INT32_MIN;  // (int)NaN is INT_MIN on the reference's machine: the squelch skips it
			if (lane == 0 && p.levels) p.levels[(size_t)s * p.nblocks + b] = sr;
			if (p.squelch_level && sr >= 0) {  // wave-uniform
				const bool mute = sr < p.squelch_level;
				if (mute) for (int i = lane; i < lp_len; i += 64) lp[i] = 0;
				if (lane == 0) carry[2] = mute ? carry[2] + 1 : 0;
			}
			__syncthreads();
		}
		int16_t *res = p.R + (size_t)s * p.rstride + (size_t)b * n_res;
		if (p.mode == RTLFM_MODE_FM) {
			// fm_demod(), :932-959: output k pairs elements (2k, 2k + 1) with (2k - 2, 2k - 1); the first one with the carried
			// pair and always polar_discriminant.  With fewer than two elements the reference reads lowpassed[-1]: nothing is produced
			if (lp_len >= 2) {
				if (lane == 0) res[0] = (int16_t)disc_std(lp[0], lp[1], carry[0], carry[1]);
				for (int k = 1 + lane; 2 * k < lp_len - 1; k += 64)
					res[k] = (int16_t)discriminate(p.variant, lp[2 * k], lp[2 * k + 1], lp[2 * k - 2], lp[2 * k - 1], p.lut);
				__syncthreads();
				if (lane == 0) { carry[0] = lp[lp_len - 2]; carry[1] = lp[lp_len - 1]; }
			}
		} else if (p.mode == RTLFM_MODE_RAW) {
			for (int i = lane; i < lp_len; i += 64) res[i] = lp[i];
		} else {
			// am_demod / usb_demod / lsb_demod, :961-1000: pairs (i, i + 1) for i = 0, 2, ... < lp_len
			for (int k = lane; 2 * k < lp_len; k += 64)
				if (k < n_res) res[k] = simple_demod(p.mode, pack_iq(lp[2 * k], lp[2 * k + 1]), p.output_scale);
		}
		__syncthreads();
	}
	for (int k = lane; k < 6 * (p.passes - p.first); k += 64) {
		const int q = p.first + k / 6, j = k % 6;
		out.lp_i_hist[q][j] = hist[0][q][j]; out.lp_q_hist[q][j] = hist[1][q][j];
	}
	if (lane < 9) { out.droop_i_hist[lane] = droop[0][lane]; out.droop_q_hist[lane] = droop[1][lane]; }
	if (lane == 0) {
		if (p.mode == RTLFM_MODE_FM) { out.pre_r = carry[0]; out.pre_j = carry[1]; }
		if (p.squelch_level) out.squelch_hits = carry[2];
	}
}

// ------------------------------------------------------------- audio tail ----
// low_pass_simple (src/rtl_fm.c:739-753): sums of `step`, no divide.
__global__ void __launch_bounds__(256)
k_post_downsample(const int16_t *__restrict__ A, size_t astride, int16_t *__restrict__ B, size_t bstride,
                  int Tout, int nstreams, int step)
{
	const size_t total = (size_t)nstreams * Tout;
	RTLFM_GRID_STRIDE(g, total) {
		int t = (int)(g % Tout);
		size_t s = g / Tout;
		const int16_t *src = A + s * astride + (size_t)t * step;
		int acc = 0;
		for (int k = 0; k < step; k++) acc += src[k];
		B[s * bstride + t] = (int16_t)acc;
	}
}

// deemph_filter (src/rtl_fm.c:1011-1026): non-linear one-pole IIR
//     d = x - avg;  avg += d > 0 ? (d + a/2) / a : (d - a/2) / a;  x = (int16_t)avg
// The rounding makes it order-dependent, so one lane walks one stream's run in
// order (len[s] or T samples) and the cost is the length of the dependent chain
// per sample.  It is kept to four instructions:
//   * C's truncating division is odd-symmetric, so the step is sign(d) * ((|d| + a/2) / a);
//   * x and avg are carried biased by +32768, which makes |d| + a/2 one v_sad_u32;
//   * (|d| + a/2) < 2^17, so the quotient is v_mul_hi_u32 with M = ceil(2^32 / a), exact
//     for every a <= 32768 (Granlund-Montgomery: (M*a - 2^32) * 2^17 <= 2^32); MAGIC = false
//     keeps the hardware division for a == 1 and larger a;
//   * add / subtract are both formed, the compare that picks one is off the chain.
// Samples move in 16-byte groups, a 128-byte line at a time with the next line in flight.
// MAGIC: 0 = hardware division, 1 = multiply-high, 2 = a is a power of two (a = 2 at 16 / 24 kHz,
// 4 at 44.1 / 48 kHz): a shift, which takes the quarter-rate v_mul_hi_u32 out of the chain;
// 3 = a == 2 (round 5): avg + sign(d) ((|d| + 1) >> 1) is (x + avg + [x > avg]) >> 1 - the mean of the two, the odd
// half rounded away from avg: add, compare, add-with-carry, shift
struct DeemphStep {
	uint32_t a, half, magic;  // magic: M for MAGIC 1, log2(a) for MAGIC 2
	template <int MAGIC>
	__device__ __forceinline__ uint32_t step(uint32_t xb, uint32_t avgb) const
	{
		if (MAGIC == 3) return (xb + avgb + (xb > avgb ? 1u : 0u)) >> 1;
		const uint32_t n = (xb > avgb ? xb - avgb : avgb - xb) + half;
		const uint32_t q = MAGIC == 2 ? n >> magic : MAGIC == 1 ? __umulhi(n, magic) : n / a;
		const uint32_t up = avgb + q, dn = avgb - q;
		return xb > avgb ? up : dn;
	}
};

// n samples at r from the biased state avgb; WRITE: store the filtered samples.  Samples move in
// 128-byte lines (64 samples) with the next line's eight loads in flight while this one is
// walked: every lane reads its own row, so each line is a separate trip to HBM and one line of
// the chain (~3000 cycles) is about what that trip takes.
template <int MAGIC, bool WRITE>
__device__ __forceinline__ uint32_t deemph_walk(int16_t *r, int n, uint32_t avgb, const DeemphStep &ds)
{
	auto one = [&](int k) {
		avgb = ds.step<MAGIC>((uint32_t)(uint16_t)r[k] ^ 0x8000u, avgb);
		if (WRITE) r[k] = (int16_t)(uint16_t)(avgb ^ 0x8000u);
	};
	int k = 0;
	// head: up to the first 16-byte boundary
	const int head = (int)(((16 - ((uintptr_t)r & 15)) & 15) >> 1);
	for (; k < head && k < n; k++) one(k);
	auto group = [&](uint4 &g) {
		uint32_t w[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const uint32_t b = w[j] ^ 0x80008000u;
			avgb = ds.step<MAGIC>(b & 0xffffu, avgb);
			const uint32_t lo = avgb;
			avgb = ds.step<MAGIC>(b >> 16, avgb);
			w[j] = ((lo & 0xffffu) | (avgb << 16)) ^ 0x80008000u;
		}
		g = make_uint4(w[0], w[1], w[2], w[3]);
	};
	if (k + 64 <= n) {
		uint4 cur[8], nxt[8];
#pragma unroll
		for (int j = 0; j < 8; j++) cur[j] = reinterpret_cast<const uint4 *>(r + k)[j];
		for (; k + 64 <= n; k += 64) {
			const bool more = k + 128 <= n;
			const uint4 *np = reinterpret_cast<const uint4 *>(r + (more ? k + 64 : k));
#pragma unroll
			for (int j = 0; j < 8; j++) nxt[j] = np[j];
#pragma unroll
			for (int j = 0; j < 8; j++) {
				group(cur[j]);
				if (WRITE) reinterpret_cast<uint4 *>(r + k)[j] = cur[j];
			}
#pragma unroll
			for (int j = 0; j < 8; j++) cur[j] = nxt[j];
		}
	}
	for (; k + 8 <= n; k += 8) {
		uint4 g = *reinterpret_cast<const uint4 *>(r + k);
		group(g);
		if (WRITE) *reinterpret_cast<uint4 *>(r + k) = g;
	}
	for (; k < n; k++) one(k);
	return avgb;
}

// the plain form, for a state outside the int16 range (it can only come from rtlfm_gpu_state_set)
__device__ __forceinline__ int deemph_plain(int16_t *r, int n, int avg, int a)
{
	const int half = a / 2;
	for (int k = 0; k < n; k++) {
		int d = r[k] - avg;
		avg += d > 0 ? (d + half) / a : (d - half) / a;
		r[k] = (int16_t)avg;
	}
	return avg;
}

template <int MAGIC>
__global__ void __launch_bounds__(64)
k_deemph(int16_t *__restrict__ R, size_t rstride, int T, const int32_t *__restrict__ cnt, int nstreams,
         DeemphStep ds, const state_t *__restrict__ sin, state_t *__restrict__ sout)
{
	const size_t s = (size_t)blockIdx.x * 64 + threadIdx.x;
	if (s >= (size_t)nstreams) return;
	const int n = cnt ? cnt[s] : T;
	int16_t *r = R + s * rstride;
	if ((uint32_t)(sin[s].deemph_avg + 32768) > 65535u) {
		sout[s].deemph_avg = deemph_plain(r, n, sin[s].deemph_avg, (int)ds.a);
		return;
	}
	const uint32_t avgb = deemph_walk<MAGIC, true>(r, n, (uint32_t)(sin[s].deemph_avg + 32768), ds);
	sout[s].deemph_avg = (int)avgb - 32768;
}

// a == 1 (rate_out up to 13 kHz with 75 us: rtl_fm -s 12k -E deemp): avg += (x - avg) / 1 makes avg = x - the filter
// hands every sample on unchanged and keeps the last one.  Until round 5 this took the one-lane-per-stream walk like
// any other divisor that the time-parallel form does not cover: 2.16 ms of a 2.2 ms step at 256 streams x 65536 samples.
__global__ void k_deemph_identity(const int16_t *__restrict__ R, size_t rstride, int T, const int32_t *__restrict__ cnt, int nstreams,
                                  state_t *__restrict__ sout)
{
	RTLFM_GRID_STRIDE(s, nstreams) {
		const int n = cnt ? cnt[s] : T;
		if (n > 0) sout[s].deemph_avg = R[(size_t)s * rstride + n - 1];
	}
}

// ---- the same filter for few, long streams: exact parallelisation over time ------------------
// One lane per stream leaves the GPU idle when there are a thousand streams of 350 k samples
// (-M wbfm at scale: 9 ms of a 12 ms step).  The step f_x(v) = v + rdiv(x - v, a) is monotone
// and 1-Lipschitz in the state v, so a state interval maps onto the interval between the images
// of its ends, and it contracts (gap -> gap - gap/a + 1): from the whole int16 range to at most
// a few dozen within ~100 samples for the usual a.  Per chunk of L samples (L a multiple of 64):
//   A (all chunks at once): A1 runs the two extreme states through the chunk; if they merge the
//     chunk is done, else A2 takes the interval as it was after k samples (small enough for a
//     lane group) and walks one lane per candidate state to the end of the chunk: table[j] = outgoing state for the state lo + j after k samples;
//   B (one lane per chunk): the chunk's incoming state: look back to the nearest merged chunk and
//     thread the state forward through the unmerged ones (k samples + one table lookup each);
//   C (one lane per chunk, all chunks at once): replay the chunk from its incoming state, writing.
// Every step is the reference's integer step; chunks whose interval does not contract in time
// (or a true state outside the tracked interval) are simply walked in B.
constexpr int kDeemphGap = 62;  // candidates per chunk must fit a wave
struct DeemphChunk {
	int32_t lo;   // biased lower end after k samples
	int32_t k;    // samples consumed before the table applies; -1: not contracted
	int32_t n;    // candidate states in the table; -1: merged, table[0] whatever comes in
	int16_t table[64];
};
// chunk c of a run of n samples whose first 16-byte boundary is `head` samples in: [begin, end)
__device__ __forceinline__ void deemph_chunk_range(int c, int n, int head, int L, int &begin, int &end)
{
	begin = c == 0 ? 0 : head + c * L;
	end = head + (c + 1) * L;
	if (end > n) end = n;
	if (begin > n) begin = n;
}
__device__ __forceinline__ int deemph_chunks(int n, int head, int L)
{
	const int m = n - head;
	return m <= L ? 1 : (m + L - 1) / L;
}

// (stream, chunk) pairs for the kernels of the four-pass filter: every stream of the handle, one pair per
// lane (or per `lanes` lanes); the grid covers nstreams * max_chunks pairs.
template <class Body>
__device__ __forceinline__ void for_stream_chunks(int nstreams, int max_chunks, int lanes, Body body)
{
	const int per_wg = 64 / lanes;
	const int sub = (int)threadIdx.x % lanes, slot = (int)threadIdx.x / lanes;
	const size_t g = (size_t)blockIdx.x * per_wg + slot;
	const size_t s = g / max_chunks;
	if (s < (size_t)nstreams) body(s, (int)(g % max_chunks), sub);
}

// A1: one lane per chunk walks the two extreme states through the whole chunk.  It records where
// the interval first fits a lane group of pass A2 (k, lo_k, hi_k) and whether the two states have
// merged by the end of the chunk - then the outgoing state does not depend on the incoming one
// (n = -1, table[0]) and A2 has nothing to do.  Anything but a silent stream merges.
template <int MAGIC>
__global__ void __launch_bounds__(64)
k_deemph_scan_a1(int16_t *__restrict__ R, size_t rstride, int T, const int32_t *__restrict__ cnt, int nstreams,
                 DeemphStep ds, int max_chunks, int L, int lpc, DeemphChunk *__restrict__ tab)
{
	for_stream_chunks(nstreams, max_chunks, 1, [&](const size_t s, const int c, int) {
	const int n = cnt ? cnt[s] : T;
	int16_t *r = R + s * rstride;
	const int head = (int)(((16 - ((uintptr_t)r & 15)) & 15) >> 1);
	if (c >= deemph_chunks(n, head, L) - 1) return;  // the last chunk's outgoing state comes from pass C
	int begin, end;
	deemph_chunk_range(c, n, head, L, begin, end);
	const uint32_t gap_max = (uint32_t)(lpc - 2);
	uint32_t lo = 0, hi = 65535, lo_k = 0, hi_k = 0;
	int kfit = -1;
	auto one = [&](int k) {
		const uint32_t x = (uint32_t)(uint16_t)r[k] ^ 0x8000u;
		lo = ds.step<MAGIC>(x, lo); hi = ds.step<MAGIC>(x, hi);
	};
	auto note = [&](int k) {  // k: samples consumed so far (relative to begin), a multiple of 8 from the aligned part on
		if (kfit < 0 && hi - lo <= gap_max) { kfit = k; lo_k = lo; hi_k = hi; }
	};
	int k = begin;
	// chunk 0 may start before the first 16-byte boundary
	const int pre = (int)(((16 - ((uintptr_t)(r + k) & 15)) & 15) >> 1);
	for (int j = 0; j < pre && k < end; j++, k++) one(k);
	if (k + 64 <= end) {
		uint4 cur[8], nxt[8];
#pragma unroll
		for (int j = 0; j < 8; j++) cur[j] = reinterpret_cast<const uint4 *>(r + k)[j];
		for (; k + 64 <= end; k += 64) {
			const uint4 *np = reinterpret_cast<const uint4 *>(r + (k + 128 <= end ? k + 64 : k));
#pragma unroll
			for (int j = 0; j < 8; j++) nxt[j] = np[j];
#pragma unroll
			for (int j = 0; j < 8; j++) {
				note(k + 8 * j - begin);
				const uint32_t w[4] = {cur[j].x, cur[j].y, cur[j].z, cur[j].w};
#pragma unroll
				for (int i = 0; i < 4; i++) {
					const uint32_t b2 = w[i] ^ 0x80008000u;
					lo = ds.step<MAGIC>(b2 & 0xffffu, lo); hi = ds.step<MAGIC>(b2 & 0xffffu, hi);
					lo = ds.step<MAGIC>(b2 >> 16, lo); hi = ds.step<MAGIC>(b2 >> 16, hi);
				}
			}
#pragma unroll
			for (int j = 0; j < 8; j++) cur[j] = nxt[j];
		}
	}
	for (; k < end; k++) {
		if (((k - begin - pre) & 7) == 0) note(k - begin);
		one(k);
	}
	DeemphChunk *t = tab + s * max_chunks + c;
	if (lo == hi) {
		t->lo = 0; t->k = 0; t->n = -1;
		t->table[0] = (int16_t)(uint16_t)(lo ^ 0x8000u);
	} else {
		t->lo = (int32_t)lo_k; t->k = kfit; t->n = kfit < 0 ? 0 : (int32_t)(hi_k - lo_k + 1);
	}
	});
}

// A2: for the chunks A1 left unmerged, lpc lanes per chunk (a power of two >= 2a + 3), one lane per
// candidate state lo_k .. hi_k from sample k to the end of the chunk -> table
template <int MAGIC>
__global__ void __launch_bounds__(64)
k_deemph_scan_a2(int16_t *__restrict__ R, size_t rstride, int T, const int32_t *__restrict__ cnt, int nstreams,
                 DeemphStep ds, int max_chunks, int L, int lpc, DeemphChunk *__restrict__ tab)
{
	for_stream_chunks(nstreams, max_chunks, lpc, [&](const size_t s, const int c, const int sub) {
	const int n = cnt ? cnt[s] : T;
	int16_t *r = R + s * rstride;
	const int head = (int)(((16 - ((uintptr_t)r & 15)) & 15) >> 1);
	if (c >= deemph_chunks(n, head, L) - 1) return;
	DeemphChunk *t = tab + s * max_chunks + c;
	const int k = t->k, nc = t->n;
	if (k < 0 || nc <= 0) return;  // not contracted (pass B walks it) or merged
	int begin, end;
	deemph_chunk_range(c, n, head, L, begin, end);
	uint32_t v = (uint32_t)t->lo + (uint32_t)(sub < nc ? sub : nc - 1);
	v = deemph_walk<MAGIC, false>(r + begin + k, end - begin - k, v, ds);
	if (sub < nc) t->table[sub] = (int16_t)(uint16_t)(v ^ 0x8000u);
	});
}

// B: the incoming state of every chunk, one lane per chunk.  A merged chunk fixes its successor's
// incoming state by itself, so a lane looks back to the nearest merged chunk (or the start of the
// run), usually its direct predecessor, and threads the state forward from there through the
// unmerged chunks in between (their first k samples + table).  Lane (stream, 0) also handles a
// stream whose carried state lies outside the int16 range (plain form, whole run).
template <int MAGIC>
__global__ void __launch_bounds__(64)
k_deemph_scan_b(int16_t *__restrict__ R, size_t rstride, int T, const int32_t *__restrict__ cnt, int nstreams,
                DeemphStep ds, int max_chunks, int L, const DeemphChunk *__restrict__ tab, uint32_t *__restrict__ incoming,
                const state_t *__restrict__ sin, state_t *__restrict__ sout)
{
	for_stream_chunks(nstreams, max_chunks, 1, [&](const size_t s, const int c, int) {
	const int n = cnt ? cnt[s] : T;
	int16_t *r = R + s * rstride;
	uint32_t *inc = incoming + s * max_chunks;
	if ((uint32_t)(sin[s].deemph_avg + 32768) > 65535u) {
		if (c == 0) {
			sout[s].deemph_avg = deemph_plain(r, n, sin[s].deemph_avg, (int)ds.a);
			inc[0] = 0xffffffffu;  // pass C leaves this stream alone
		}
		return;
	}
	const int head = (int)(((16 - ((uintptr_t)r & 15)) & 15) >> 1);
	const int nc = deemph_chunks(n, head, L);
	if (c >= nc) return;
	const DeemphChunk *t0 = tab + s * max_chunks;
	int j = c - 1;  // nearest chunk before c whose outgoing state is known without its incoming one
	while (j >= 0 && t0[j].n != -1) j--;
	uint32_t v = j < 0 ? (uint32_t)(sin[s].deemph_avg + 32768) : (uint32_t)(uint16_t)t0[j].table[0] ^ 0x8000u;
	for (int i = j + 1; i < c; i++) {
		int begin, end;
		deemph_chunk_range(i, n, head, L, begin, end);
		const DeemphChunk *t = t0 + i;
		const int k = t->k;
		if (k < 0) {
			v = deemph_walk<MAGIC, false>(r + begin, end - begin, v, ds);
		} else {
			v = deemph_walk<MAGIC, false>(r + begin, k, v, ds);
			const uint32_t q = v - (uint32_t)t->lo;
			if (q < (uint32_t)t->n) v = (uint32_t)(uint16_t)t->table[q] ^ 0x8000u;
			else v = deemph_walk<MAGIC, false>(r + begin + k, end - begin - k, v, ds);  // outside the tracked interval
		}
	}
	inc[c] = v;
	});
}

template <int MAGIC>
__global__ void __launch_bounds__(64)
k_deemph_scan_c(int16_t *__restrict__ R, size_t rstride, int T, const int32_t *__restrict__ cnt, int nstreams,
                DeemphStep ds, int max_chunks, int L, const uint32_t *__restrict__ incoming, state_t *__restrict__ sout)
{
	for_stream_chunks(nstreams, max_chunks, 1, [&](const size_t s, const int c, int) {
	if (incoming[s * max_chunks] == 0xffffffffu) return;
	const int n = cnt ? cnt[s] : T;
	int16_t *r = R + s * rstride;
	const int head = (int)(((16 - ((uintptr_t)r & 15)) & 15) >> 1);
	const int nc = deemph_chunks(n, head, L);
	if (c >= nc) return;
	int begin, end;
	deemph_chunk_range(c, n, head, L, begin, end);
	const uint32_t v = deemph_walk<MAGIC, true>(r + begin, end - begin, incoming[s * max_chunks + c], ds);
	if (c == nc - 1) sout[s].deemph_avg = (int)v - 32768;
	});
}

// floor(a / b) for 0 <= a < 2^52, 0 < b < 2^31: the operands are exact in fp64, the correctly
// rounded quotient is at most one off after truncation, and the remainder says which way (a 64-bit
// integer division expands to ~100 instructions here, this to ~25)
__device__ __forceinline__ long long floor_div_pos(long long a, int b)
{
	long long q = (long long)((double)a / (double)b);
	long long r = a - q * (long long)b;
	if (r < 0) { q -= 1; r += b; }
	else if (r >= b) { q += 1; r -= b; }
	return q;
}

// ---- the replay pass with low_pass_real folded in ---------------------------------------------
// `-M wbfm` is deemph_filter followed by low_pass_real (src/rtl_fm.c:1264-1271): the replay pass
// above would write 2 bytes per filtered sample only for k_low_pass_real to read them back and keep
// one in five.  Here the lane that replays a chunk feeds the filtered samples straight into the
// reference's accumulator (now_lpr += y; prev_lpr_index += slow; emit when it reaches fast): the
// phase and the output index at the chunk's start are closed forms, the sums are linear, so only
// the one output that straddles a chunk boundary needs the neighbour's partial sum - it is put
// together by k_lpr_fixup from head[c] (the sum up to the chunk's first emission) and tail[] (what
// was left in the accumulator at the end of the chunks before).
template <int MAGIC, class Sink>
__device__ __forceinline__ uint32_t deemph_walk_sink(const int16_t *r, int n, uint32_t avgb, const DeemphStep &ds, bool filter,
                                                    Sink &&sink)
{
	auto one = [&](int k) {
		const uint32_t xb = (uint32_t)(uint16_t)r[k] ^ 0x8000u;
		avgb = filter ? ds.step<MAGIC>(xb, avgb) : xb;
		sink((int)(int16_t)(uint16_t)(avgb ^ 0x8000u));
	};
	int k = 0;
	const int head = (int)(((16 - ((uintptr_t)r & 15)) & 15) >> 1);
	for (; k < head && k < n; k++) one(k);
	auto group = [&](const uint4 &g) {
		const uint32_t w[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const uint32_t b = w[j] ^ 0x80008000u;
			avgb = filter ? ds.step<MAGIC>(b & 0xffffu, avgb) : (b & 0xffffu);
			sink((int)(int16_t)(uint16_t)(avgb ^ 0x8000u));
			avgb = filter ? ds.step<MAGIC>(b >> 16, avgb) : (b >> 16);
			sink((int)(int16_t)(uint16_t)(avgb ^ 0x8000u));
		}
	};
	// 16-byte groups / samples in flight ahead of the walk.  (With four groups and 62 registers a wave of
	// this fits beside four waves of k_boxcar_scan<1> on a SIMD: 1 % on the wbfm step, and slower alone.)
	constexpr int NG = 8, NS = 8 * NG;
	if (k + NS <= n) {
		uint4 cur[NG], nxt[NG];
#pragma unroll
		for (int j = 0; j < NG; j++) cur[j] = reinterpret_cast<const uint4 *>(r + k)[j];
		for (; k + NS <= n; k += NS) {
			const bool more = k + 2 * NS <= n;
			const uint4 *np = reinterpret_cast<const uint4 *>(r + (more ? k + NS : k));
#pragma unroll
			for (int j = 0; j < NG; j++) nxt[j] = np[j];
#pragma unroll
			for (int j = 0; j < NG; j++) group(cur[j]);
#pragma unroll
			for (int j = 0; j < NG; j++) cur[j] = nxt[j];
		}
	}
	for (; k + 8 <= n; k += 8) group(*reinterpret_cast<const uint4 *>(r + k));
	for (; k < n; k++) one(k);
	return avgb;
}

// C's int / d for a divisor fixed over a walk (low_pass_real's fast / slow): the magic number of
// d for 31-bit magnitudes, one v_mul_hi per quotient instead of the ~40 instructions of the
// expanded division
struct ConstDiv {
	uint32_t M;
	int sh, d;
	__device__ __forceinline__ explicit ConstDiv(int d_) : M(0), sh(0), d(d_)
	{
		if (d >= 2) {
			const int l = 32 - __clz(d - 1);  // ceil(log2 d)
			M = (uint32_t)(((1ull << (31 + l)) + (unsigned)d - 1u) / (unsigned)d);
			sh = l - 1;
		}
	}
	__device__ __forceinline__ int operator()(int n) const
	{
		if (d < 2) return n;  // fast / slow >= 1
		if (n == INT32_MIN) return n / d;
		const uint32_t a = n < 0 ? 0u - (uint32_t)n : (uint32_t)n;
		const uint32_t q = __umulhi(a, M) >> sh;
		return n < 0 ? (int)(0u - q) : (int)q;
	}
};

struct LprChunk {
	uint32_t head;   // accumulator at the chunk's first emission (the part of the straddling output that lies in this chunk)
	uint32_t tail;   // accumulator at the chunk's end
	int32_t mfirst;  // index of the output of the first emission, -1: the chunk emitted nothing
};

// The reference's accumulator (now_lpr += y; prev_lpr_index += slow; emit when it reaches fast,
// src/rtl_fm.c:755-775) fed by the lane that walks a chunk.  Every lane of a wave writes a row of
// its own, so a 2-byte store per output is 64 partial writes to 64 different lines per
// instruction, and a line collects its 64 outputs over ~20000 cycles - long enough to be evicted
// half written again and again (the emission was 60 % of the one-pass kernel's time).  Outputs are
// therefore shifted into a 16-byte register group and stored eight at a time (`vec`: the output
// rows are 16-byte aligned); the outputs before the chunk's first multiple of eight and after its
// last one go out one by one.  The first emission of a chunk other than the stream's first belongs
// to an output that began in the chunk before: its value is put together by k_lpr_fixup from
// head, what is stored for it here is overwritten there.
// Round 5, `ring`: a lane's outputs wait in 64 entries of LDS that mirror the row's addresses, and whenever the lane
// completes a 64-byte piece of its row (32 outputs, aligned in memory whatever the row's own alignment is) the piece
// leaves as four 16-byte stores in a row - one burst per 64 bytes instead of four 16-byte pieces that a line collected
// over ~20000 cycles (profiles/r04_pmc_wbfm_k_deemph_spec_lpr.txt: 407 MB written for 135 MB of audio).  The outputs in
// front of the chunk's first piece boundary and behind its last go out one by one, as before.
constexpr int kLprRing = 64, kLprRingStride = 72;  // int16 entries per lane / between lanes (144 bytes: 16-byte aligned rows)
struct LprSink {
	int16_t *bo;
	int m, phi, sl, fa, first_full;
	uint32_t acc;
	uint32_t p0, p1, p2, p3;  // the last eight outputs, oldest in the low half of p0
	ConstDiv cdiv;
	LprChunk out;
	int16_t *ring;            // this lane's kLprRing entries in LDS, or nullptr: the register form
	int shift;                // (row address / 2) mod 32: output m sits at 64-byte offset 2 ((m + shift) mod 32)
	__device__ __forceinline__ LprSink(int16_t *bo_, int m_, int phi_, int slow, int fast, uint32_t acc_, bool vec, int16_t *ring_ = nullptr)
	    : bo(bo_), m(m_), phi(phi_), sl(slow), fa(fast), first_full(vec ? ((m_ + 7) & ~7) : INT32_MAX), acc(acc_), p0(0), p1(0),
	      p2(0), p3(0), cdiv(fast / slow), ring(ring_), shift(0)
	{
		out.mfirst = -1; out.head = 0; out.tail = 0;
		if (ring) {
			shift = (int)(((uintptr_t)bo_ >> 1) & 31);
			first_full = m_ + ((32 - ((m_ + shift) & 31)) & 31);  // the chunk's first piece boundary
		}
	}
	__device__ __forceinline__ void operator()(int y)
	{
		acc += (uint32_t)y;
		phi += sl;
		if (phi >= fa) {
			if (out.mfirst < 0) { out.mfirst = m; out.head = acc; }
			const uint32_t q = (uint32_t)cdiv((int)acc);
			if (ring) {
				ring[(m + shift) & (kLprRing - 1)] = (int16_t)q;
				if (m < first_full) bo[m] = (int16_t)q;
				m++;
				if (((m + shift) & 31) == 0 && m > first_full) {
					const uint4 *src = reinterpret_cast<const uint4 *>(ring + ((m - 32 + shift) & (kLprRing - 1)));
					uint4 *dst = reinterpret_cast<uint4 *>(bo + (m - 32));
					const uint4 a = src[0], b = src[1], c = src[2], d = src[3];
					dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d;
				}
			} else {
				p0 = __builtin_amdgcn_alignbit(p1, p0, 16);
				p1 = __builtin_amdgcn_alignbit(p2, p1, 16);
				p2 = __builtin_amdgcn_alignbit(p3, p2, 16);
				p3 = (p3 >> 16) | (q << 16);
				if (m < first_full) bo[m] = (int16_t)q;
				m++;
				if ((m & 7) == 0 && m > first_full) *reinterpret_cast<uint4 *>(bo + m - 8) = make_uint4(p0, p1, p2, p3);
			}
			phi -= fa;
			acc = 0;
		}
	}
	__device__ __forceinline__ void finish()  // the outputs after the last full group
	{
		out.tail = acc;
		if (ring) {
			const int mb = m - ((m + shift) & 31);  // the last piece boundary
			if (mb < first_full) return;            // the chunk never got past its first boundary: everything went out one by one
			for (int j = mb; j < m; j++) bo[j] = ring[(j + shift) & (kLprRing - 1)];
			return;
		}
		const int nrem = m & 7;
		if (m - nrem < first_full) return;  // written one by one already (or `vec` is off)
		const uint32_t p[4] = {p0, p1, p2, p3};
#pragma unroll
		for (int j = 0; j < 8; j++)
			if (j >= 8 - nrem) bo[m - 8 + j] = (int16_t)(p[j >> 1] >> (16 * (j & 1)));
	}
};

template <int MAGIC>
__global__ void __launch_bounds__(64)
k_deemph_scan_c_lpr(const int16_t *__restrict__ R, size_t rstride, int T, const int32_t *__restrict__ cnt, int nstreams,
                    DeemphStep ds, int max_chunks, int L, const uint32_t *__restrict__ incoming, int16_t *__restrict__ B,
                    size_t bstride, int fast, int slow, const state_t *__restrict__ sin, state_t *__restrict__ sout,
                    LprChunk *__restrict__ lc, int vec)
{
	for_stream_chunks(nstreams, max_chunks, 1, [&](const size_t s, const int c, int) {
	const int n = cnt ? cnt[s] : T;
	const int16_t *r = R + s * rstride;
	const int head = (int)(((16 - ((uintptr_t)r & 15)) & 15) >> 1);
	const int nc = deemph_chunks(n, head, L);
	if (c >= nc) return;
	int begin, end;
	deemph_chunk_range(c, n, head, L, begin, end);
	// a stream whose carried deemph state lay outside int16 was filtered in place by pass B (plain form)
	const bool filter = incoming[s * max_chunks] != 0xffffffffu;
	const long long p0 = sin[s].prev_lpr_index;
	const long long idx0 = p0 + (long long)begin * slow;
	const int m0 = (p0 >= 0 && p0 < fast) ? (int)floor_div_pos(idx0, fast) : (int)(idx0 / fast);
	LprSink sink(B + s * bstride, m0, (int)(idx0 - (long long)m0 * fast), slow, fast, c == 0 ? (uint32_t)sin[s].now_lpr : 0u, vec != 0);
	const uint32_t v = deemph_walk_sink<MAGIC>(r + begin, end - begin, filter ? incoming[s * max_chunks + c] : 0u, ds, filter, sink);
	sink.finish();
	lc[s * max_chunks + c] = sink.out;
	if (filter && c == nc - 1) sout[s].deemph_avg = (int)v - 32768;
	});
}

// ---- one pass instead of A1 / A2 / B / C --------------------------------------------------------
// The filter forgets: two walks from the two extreme states meet after ~11 a samples on any signal
// that moves (a = 13: 143 in the median, 232 at most on the wbfm test signal; a = 2: 17 / 23).  So a
// chunk does not need the tables of passes A and B to learn its incoming state: it walks both
// extremes over the W samples before its own first one, and if they have met, that IS the state the
// chunk starts from, exactly, whatever came before.  Then it replays its samples into the
// resampler as k_deemph_scan_c_lpr does.  One read of the run plus W / L instead of two.  A chunk
// whose extremes have not met (a silent stream) raises its stream's flag, and the lane that is last
// to finish a flagged stream redoes it with the reference's sequential loop; a filter state outside int16 likewise.
template <int MAGIC>
__device__ __forceinline__ void deemph_walk_pair(const int16_t *r, int n, uint32_t &lo, uint32_t &hi, const DeemphStep &ds)
{
	int k = 0;
	const int pre = (int)(((16 - ((uintptr_t)r & 15)) & 15) >> 1);
	for (; k < pre && k < n; k++) {
		const uint32_t x = (uint32_t)(uint16_t)r[k] ^ 0x8000u;
		lo = ds.step<MAGIC>(x, lo); hi = ds.step<MAGIC>(x, hi);
	}
	auto group = [&](const uint4 &g) {
		const uint32_t w[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const uint32_t b2 = w[i] ^ 0x80008000u;
			lo = ds.step<MAGIC>(b2 & 0xffffu, lo); hi = ds.step<MAGIC>(b2 & 0xffffu, hi);
			lo = ds.step<MAGIC>(b2 >> 16, lo); hi = ds.step<MAGIC>(b2 >> 16, hi);
		}
	};
	// four 16-byte groups in flight ahead of the walk, as deemph_walk_sink has eight: a lane's loads are its
	// own (a chunk per lane), and one L2 round trip per eight samples was most of the settle walk's time
	constexpr int NG = 4, NS = 8 * NG;
	if (k + NS <= n) {
		uint4 cur[NG], nxt[NG];
#pragma unroll
		for (int j = 0; j < NG; j++) cur[j] = reinterpret_cast<const uint4 *>(r + k)[j];
		for (; k + NS <= n; k += NS) {
			const bool more = k + 2 * NS <= n;
			const uint4 *np = reinterpret_cast<const uint4 *>(r + (more ? k + NS : k));
#pragma unroll
			for (int j = 0; j < NG; j++) nxt[j] = np[j];
#pragma unroll
			for (int j = 0; j < NG; j++) group(cur[j]);
#pragma unroll
			for (int j = 0; j < NG; j++) cur[j] = nxt[j];
		}
	}
	for (; k + 8 <= n; k += 8) group(*reinterpret_cast<const uint4 *>(r + k));
	for (; k < n; k++) {
		const uint32_t x = (uint32_t)(uint16_t)r[k] ^ 0x8000u;
		lo = ds.step<MAGIC>(x, lo); hi = ds.step<MAGIC>(x, hi);
	}
}

// The one-pass tail kernels (k_deemph_spec_lpr, k_deemph_spec_arb) are ONE launch with no helpers around them: a
// workgroup owns whole streams, so a stream that cannot be settled (a chunk whose two extreme walks do not meet:
// silence; a carried filter state outside int16) is redone inside the workgroup that found out, after a workgroup
// barrier, with the reference's own sequential loop - nothing crosses workgroups.  (Round 3 had two memsets,
// k_flag_list, the four passes of the time-parallel filter, k_arb_upsample_only and a copy of the counts around
// these kernels: ten operations per step on the tail's stream, each of which waited for room beside the next
// step's front end.  A first one-launch form of round 4 let the workgroups of a stream meet through a per-stream
// ticket in global memory: on this part an agent-scope release / acquire is a write-back / invalidate of the
// XCD's whole L2, and sixteen thousand of them made the kernel ten times slower - and the front end beside it.)
// low_pass_real's totals over a run of n samples from the carried phase p0 (src/rtl_fm.c:755-775): outputs, phase left
__device__ __forceinline__ void lpr_totals(long long p0, int n, int slow, int fast, int &E, int &phase)
{
	const long long tot = p0 + (long long)n * slow;
	E = (p0 >= 0 && p0 < fast) ? (int)floor_div_pos(tot, fast) : (int)(tot / fast);
	phase = (int)(tot - (long long)E * fast);
}

// A workgroup of `nthreads` lanes owns spw = max(1, nthreads / max_chunks) whole streams (one stream and a loop over
// its chunks when it has more chunks than the workgroup has lanes).  Lane (stream, chunk) settles its chunk's
// incoming filter state from the W samples before it and walks the chunk into the resampler; after a workgroup
// barrier the lane of chunk 0 finishes its stream: the outputs that straddle chunk boundaries, the carried
// accumulator, phase and count (what k_lpr_fixup does for the four-pass route) - or, if a chunk could not settle
// (flag in LDS) or the carried filter state lies outside int16, the reference's own sequential loop over the run.
// (Not VALU-bound: a form of this kernel with the filter's state in fp32 - six full-rate instructions per sample instead
// of seven and a quarter-rate multiply-high, exact for every a the host checked - ran the wbfm step in 1.377 ms
// against 1.378, same box, alternating.  Every lane walks a row of its own, 16 bytes at a time: the kernel waits for
// its 64 separate lines per load instruction.)
constexpr int kSpecLprThreads = 256;
#ifdef RTLFM_LPR_WAVES4
#define RTLFM_LPR_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define RTLFM_LPR_ATTR
#endif
template <int MAGIC>
__global__ void __launch_bounds__(kSpecLprThreads) RTLFM_LPR_ATTR
k_deemph_spec_lpr(int16_t *R, size_t rstride, int T, const int32_t *__restrict__ cnt, int nstreams,
                  DeemphStep ds, int max_chunks, int L, int W, int16_t *__restrict__ B, size_t bstride, int fast, int slow,
                  const state_t *__restrict__ sin, state_t *__restrict__ sout, LprChunk *lc, int vec,
                  int32_t *__restrict__ cnt_out)
{
#if RTLFM_TAIL_PRIO >= 0
	__builtin_amdgcn_s_setprio(RTLFM_TAIL_PRIO);
#endif
	__shared__ int unsettled[kSpecLprThreads];  // per stream of this workgroup
	extern __shared__ __attribute__((aligned(16))) int16_t lpr_rings[];  // kLprRingStride entries per lane, or nothing (vec != 2)
	const int nthreads = (int)blockDim.x, tid = (int)threadIdx.x;
	int16_t *const my_ring = vec == 2 ? lpr_rings + (size_t)tid * kLprRingStride : nullptr;
	const int spw = max_chunks >= nthreads ? 1 : nthreads / max_chunks;  // streams per workgroup
	unsettled[tid] = 0;
	__syncthreads();
	const int sl = spw == 1 ? 0 : tid / max_chunks;                    // this lane's stream within the workgroup
	const size_t s = (size_t)blockIdx.x * spw + sl;
	const bool live = sl < spw && s < (size_t)nstreams;
	int n = 0, nc = 0, head = 0, carried = 0;
	int16_t *r = nullptr;
	long long p0 = 0;
	bool plain = false;
	if (live) {
		n = cnt ? cnt[s] : T;
		r = R + s * rstride;
		head = (int)(((16 - ((uintptr_t)r & 15)) & 15) >> 1);
		nc = deemph_chunks(n, head, L);
		carried = sin[s].deemph_avg;
		plain = (uint32_t)(carried + 32768) > 65535u;  // only rtlfm_gpu_state_set can do that
		p0 = sin[s].prev_lpr_index;
	}
	const int c_first = spw == 1 ? tid : tid - sl * max_chunks, c_step = spw == 1 ? nthreads : max_chunks;
	if (live && !plain) {
		for (int c = c_first; c < nc; c += c_step) {
			int begin, end;
			deemph_chunk_range(c, n, head, L, begin, end);
			const long long idx0 = p0 + (long long)begin * slow;
			const int m0 = (p0 >= 0 && p0 < fast) ? (int)floor_div_pos(idx0, fast) : (int)(idx0 / fast);
			LprSink sink(B + s * bstride, m0, (int)(idx0 - (long long)m0 * fast), slow, fast, c == 0 ? (uint32_t)sin[s].now_lpr : 0u, vec != 0, my_ring);
			bool settled = true;
			uint32_t v = (uint32_t)(carried + 32768);
			if (begin > 0) {
				if (begin <= W) {
					// close to the start of the run: from the carried state itself
					uint32_t v2 = v;
					deemph_walk_pair<MAGIC>(r, begin, v, v2, ds);
				} else {
					uint32_t lo = 0, hi = 65535;
					deemph_walk_pair<MAGIC>(r + begin - W, W, lo, hi, ds);
					settled = lo == hi;
					v = lo;
				}
			}
			if (!settled) { unsettled[sl] = 1; continue; }
			const int v_end = (int)deemph_walk_sink<MAGIC>(r + begin, end - begin, v, ds, true, sink) - 32768;
			sink.finish();
			lc[s * max_chunks + c] = sink.out;
			if (c == nc - 1) sout[s].deemph_avg = v_end;
		}
	}
	__threadfence_block();  // the chunk records and outputs of this workgroup's lanes, for its finishing lanes
	__syncthreads();
	if (!live) return;
	if (plain || unsettled[sl]) {
		if (c_first != 0) return;
		// the reference's loop over the whole run, on the stream's first lane
		int E, phase;
		lpr_totals(p0, n, slow, fast, E, phase);
		const int m0 = (p0 >= 0 && p0 < fast) ? (int)floor_div_pos(p0, fast) : (int)(p0 / fast);
		LprSink sink(B + s * bstride, m0, (int)(p0 - (long long)m0 * fast), slow, fast, (uint32_t)sin[s].now_lpr, vec != 0, my_ring);
		if (plain) {
			sout[s].deemph_avg = deemph_plain(r, n, carried, (int)ds.a);
			deemph_walk_sink<MAGIC>(r, n, 0u, ds, false, sink);
		} else {
			sout[s].deemph_avg = (int)deemph_walk_sink<MAGIC>(r, n, (uint32_t)(carried + 32768), ds, true, sink) - 32768;
		}
		sink.finish();
		sout[s].now_lpr = (int)sink.out.tail;
		sout[s].prev_lpr_index = phase;
		if (cnt_out) cnt_out[s] = E;
		return;
	}
	// Every lane finishes its own chunks (as k_lpr_fixup does for the four-pass route): the one output that began in
	// the chunks before - its value needs what they left in the accumulator since their last emission, normally the
	// direct predecessor's tail -, and the run's last chunk leaves the carried accumulator, phase and count.  (One lane
	// walking all chunk records of its stream here kept the other 255 lanes of the workgroup waiting for ~130 dependent
	// trips to L2: a quarter of the kernel's time at the wbfm shape.)
	const LprChunk *l0 = lc + s * max_chunks;
	const int div = fast / slow;
	auto left_by = [&](int upto) {
		uint32_t sum = 0;
		for (int j = upto; j >= 0; j--) {
			const LprChunk k = l0[j];
			sum += k.tail;
			if (k.mfirst >= 0) break;
		}
		return sum;
	};
	for (int c = c_first; c < nc; c += c_step) {
		const LprChunk k = l0[c];
		if (c > 0 && k.mfirst >= 0) B[s * bstride + k.mfirst] = (int16_t)((int)(k.head + left_by(c - 1)) / div);
		if (c == nc - 1) {
			int E, phase;
			lpr_totals(p0, n, slow, fast, E, phase);
			sout[s].now_lpr = (int)left_by(c);
			sout[s].prev_lpr_index = phase;
			if (cnt_out) cnt_out[s] = E;
		}
	}
}

// ---- -M wbfm's tail where it takes no wave slot from the front end (round 6) -----------------------------------
// k_deemph_spec_lpr above is 146 registers per lane in 256-thread workgroups.  Beside the NEXT step's front end - four
// waves of k_boxcar_scan per SIMD, 120 registers each - one of its waves only fits where a front-end wave has left, a
// workgroup only where four have left at once, and every wave of the tail that runs displaces a whole wave of a front end
// whose speed is the number of tiles it has in flight: a step cost front end + 0.27 ms (LAB.md I.1, I.19; round 6: I.21).
// But 4 x 120 registers leave 32 of a SIMD's 512, a wave slot (eight per SIMD) and 13 KB of the CU's LDS unused.  This
// kernel is the same arithmetic made to fit THERE: at most 32 registers, no LDS, one-wave workgroups, no barrier - a fifth
// wave beside four front-end waves that costs them issue cycles and memory bandwidth, not a slot.
// What had to go for that:
//   * the chunk records and the finishing pass behind a workgroup barrier.  low_pass_real's accumulator is empty right
//     behind every emission (src/rtl_fm.c:755-775), and where the emissions fall is a closed form of the carried phase: a
//     lane's stretch runs from the last emission at or before its chunk's first sample to the last one at or before the
//     chunk's end, so every output belongs to exactly one lane and nothing is put together afterwards.  The few samples in
//     front of the chunk (fewer than fast / slow + 1) come out of the settle walk, which has to have met by then;
//   * the walk's sixteen 16-byte groups in flight: one group is walked while the next is loaded;
//   * the 64-byte output pieces through LDS: outputs leave in 16-byte groups from four registers (rows 16-byte aligned),
//     the odd ones at a stretch's ends one by one;
//   * the redo of a stream that cannot settle (silence: the two extreme walks stop a / 2 either side of the input) by its
//     workgroup: the lane walks from the stream's carried state to its own stretch instead - every lane of such a stream
//     does, so a silent stream costs up to chunks / 2 times the walk, and is as slow as its last lane (as before);
//   * a carried state outside its range (filter state beyond int16, resampler phase outside [0, fast): only
//     rtlfm_gpu_state_set can do that): the stream's first lane runs the reference's two loops over the whole run.
struct SlimSink {
	int16_t *bof;  // the output row advanced to the stretch's first 16-byte group boundary: output m lives at bof[m - first_full]
	int mrel;      // m - first_full (negative in front of the boundary: those outputs leave one by one)
	int phi;
	uint32_t acc, p0, p1, p2, p3;  // the accumulator; the last eight outputs, oldest in the low half of p0
	__device__ __forceinline__ void init(int16_t *bo, int m, int phi_, uint32_t acc_, bool vec)
	{
		const int first_full = vec ? ((m + 7) & ~7) : (1 << 30);
		bof = bo + first_full; mrel = m - first_full; phi = phi_; acc = acc_; p0 = p1 = p2 = p3 = 0;
	}
	__device__ __forceinline__ void put(int y, int sl, int fa, const ConstDiv &cdiv)
	{
		acc += (uint32_t)y;
		phi += sl;
		if (phi >= fa) {
			const uint32_t q = (uint32_t)cdiv((int)acc);
			p0 = __builtin_amdgcn_alignbit(p1, p0, 16);
			p1 = __builtin_amdgcn_alignbit(p2, p1, 16);
			p2 = __builtin_amdgcn_alignbit(p3, p2, 16);
			p3 = (p3 >> 16) | (q << 16);
			if (mrel < 0) bof[mrel] = (int16_t)q;
			mrel++;
			if ((mrel & 7) == 0 && mrel > 0) *reinterpret_cast<uint4 *>(bof + mrel - 8) = make_uint4(p0, p1, p2, p3);
			phi -= fa;
			acc = 0;
		}
	}
	__device__ __forceinline__ void finish()
	{
		const int nrem = mrel & 7;
		if (mrel - nrem < 0) return;  // written one by one already
		const uint32_t p[4] = {p0, p1, p2, p3};
#pragma unroll
		for (int j = 0; j < 8; j++)
			if (j >= 8 - nrem) bof[mrel - 8 + j] = (int16_t)(p[j >> 1] >> (16 * (j & 1)));
	}
};

// `count` samples at p through the filter from the biased state v: one 16-byte group walked while the next is loaded.
// PAIR: two states (the settle walk's extremes), nothing emitted; else the filtered samples go into the sink.
template <int MAGIC, bool PAIR>
__device__ __forceinline__ void slim_walk(const int16_t *p, int count, uint32_t &v, uint32_t &v2, const DeemphStep &ds,
                                          SlimSink &sink, int sl, int fa, const ConstDiv &cdiv)
{
	auto one = [&](uint32_t xb) {
		v = ds.step<MAGIC>(xb, v);
		if (PAIR) v2 = ds.step<MAGIC>(xb, v2);
		else sink.put((int)(int16_t)(uint16_t)(v ^ 0x8000u), sl, fa, cdiv);
	};
	for (; count > 0 && (((uintptr_t)p) & 15); count--, p++) one((uint32_t)(uint16_t)*p ^ 0x8000u);
	if (count >= 8) {
		uint4 cur = *reinterpret_cast<const uint4 *>(p);
		for (; count >= 8; count -= 8) {
			p += 8;
			const uint4 nxt = *reinterpret_cast<const uint4 *>(count >= 16 ? p : p - 8);
			one((cur.x ^ 0x80008000u) & 0xffffu); one((cur.x ^ 0x80008000u) >> 16);
			one((cur.y ^ 0x80008000u) & 0xffffu); one((cur.y ^ 0x80008000u) >> 16);
			one((cur.z ^ 0x80008000u) & 0xffffu); one((cur.z ^ 0x80008000u) >> 16);
			one((cur.w ^ 0x80008000u) & 0xffffu); one((cur.w ^ 0x80008000u) >> 16);
			cur = nxt;
		}
	}
	for (; count > 0; count--, p++) one((uint32_t)(uint16_t)*p ^ 0x8000u);
}

// What a lane of k_deemph_lpr_slim is to do, worked out by k_lpr_slim_plan in front of it (the 64-bit divisions of the
// closed forms cost forty registers; the walk itself must fit 32): samples [j0, j0 + count) of its stream into the
// resampler, the first output being m0 at phase phi0.  count bit 30: the stream's last chunk (it leaves the carried
// state); count < 0: nothing to do (no such chunk, or a stream the plan kernel has done itself).
struct SlimPlan { int32_t j0, count, m0, phi0; };
constexpr int kSlimLast = 1 << 30;

// One lane per (stream, chunk).  Outputs emitted before sample i: m(i) = floor((p0 + i slow) / fast); the accumulator is
// empty at the first sample j with m(j) == m(i): j = ceil((m(i) fast - p0) / slow) - that is where a lane's stretch begins
// and where its predecessor's ends.  The run's totals (count of outputs, phase left) are closed forms too and are written
// here.  A stream whose carried state no run of the chain produces (filter state beyond int16, resampler phase outside
// [0, fast): only rtlfm_gpu_state_set can do that) is done here, by its first lane, with the reference's two loops
// (src/rtl_fm.c:1011-1026, 755-775).
__global__ void __launch_bounds__(64)
k_lpr_slim_plan(const int16_t *__restrict__ R, size_t rstride, int T, const int32_t *__restrict__ cnt, int nstreams, int a,
                int max_chunks, int L, int16_t *__restrict__ B, size_t bstride, int fast, int slow,
                const state_t *__restrict__ sin, state_t *__restrict__ sout, int32_t *__restrict__ cnt_out, SlimPlan *__restrict__ plan)
{
	const uint32_t g = blockIdx.x * 64u + threadIdx.x;
	const int s = (int)(g / (uint32_t)max_chunks);
	const int c = (int)(g - (uint32_t)s * (uint32_t)max_chunks);
	if (s >= nstreams) return;
	SlimPlan out{0, -1, 0, 0};
	const int n = cnt ? cnt[s] : T;
	const int16_t *r = R + (size_t)s * rstride;
	const int head = (int)(((16 - ((uintptr_t)r & 15)) & 15) >> 1);
	const int nc = deemph_chunks(n, head, L);
	const int carried = sin[s].deemph_avg;
	const long long p0 = sin[s].prev_lpr_index;
	if ((uint32_t)(carried + 32768) > 65535u || p0 < 0 || p0 >= fast) {
		plan[g] = out;
		if (c != 0) return;
		int16_t *bo = B + (size_t)s * bstride;
		int avg = carried, now = sin[s].now_lpr, i2 = 0;
		long long idx = p0;
		const int half = a / 2, div = fast / slow;
		for (int i = 0; i < n; i++) {
			const int d = r[i] - avg;
			avg += d > 0 ? (d + half) / a : (d - half) / a;
			now = (int)((uint32_t)now + (uint32_t)(int)(int16_t)avg);
			idx += slow;
			if (idx < fast) continue;
			bo[i2++] = (int16_t)(now / div);
			idx -= fast;
			now = 0;
		}
		sout[s].deemph_avg = avg; sout[s].now_lpr = now; sout[s].prev_lpr_index = (int)idx;
		if (cnt_out) cnt_out[s] = i2;
		return;
	}
	if (c < nc) {
		int begin, end;
		deemph_chunk_range(c, n, head, L, begin, end);
		auto stretch_start = [&](int i, int &mi) -> int {
			const long long idx = p0 + (long long)i * slow;
			mi = (int)floor_div_pos(idx, fast);
			if (mi == 0) return 0;
			const long long need = (long long)mi * fast - p0;  // > 0
			return (int)floor_div_pos(need + slow - 1, slow);
		};
		int m0 = 0, m1 = 0;
		const int j0 = c == 0 ? 0 : stretch_start(begin, m0);
		const int j1 = c == nc - 1 ? n : stretch_start(end, m1);
		if (c == 0) m0 = 0;
		out.j0 = j0; out.count = (j1 - j0) | (c == nc - 1 ? kSlimLast : 0);
		out.m0 = m0; out.phi0 = (int)(p0 + (long long)j0 * slow - (long long)m0 * fast);
		if (c == nc - 1) {
			int E, phase;
			lpr_totals(p0, n, slow, fast, E, phase);
			sout[s].prev_lpr_index = phase;
			if (cnt_out) cnt_out[s] = E;
		}
	}
	plan[g] = out;
}

template <int MAGIC>
__global__ void __launch_bounds__(64)
k_deemph_lpr_slim(const int16_t *__restrict__ R, size_t rstride, int nstreams, DeemphStep ds, int max_chunks, uint32_t chunks_magic,
                  int W, int16_t *__restrict__ B, size_t bstride, int fast, int slow, const state_t *__restrict__ sin,
                  state_t *__restrict__ sout, int vec, const SlimPlan *__restrict__ plan, int prio)
{
	// the front end's waves run at priorities 2 / 1 / 0 by their progress when a tail follows and leave 3 to the tail
	// (fused_kernel.h, ProgressPrio): a lane's walk is one dependent chain, and at the lowest priority it got an issue slot
	// in five - the tail then outlasted the front end it ran beside (LAB.md I.22)
	if (prio >= 3) __builtin_amdgcn_s_setprio(3);
	else if (prio == 2) __builtin_amdgcn_s_setprio(2);
	else if (prio == 1) __builtin_amdgcn_s_setprio(1);
	else __builtin_amdgcn_s_setprio(0);
	const uint32_t g = blockIdx.x * 64u + threadIdx.x;
	const int s = (int)__umulhi(g, chunks_magic);  // g / max_chunks (the host checked the magic number over the grid)
	if (s >= nstreams) return;
	const SlimPlan pl = plan[g];
	if (pl.count < 0) return;
	const int16_t *r = R + (size_t)s * rstride;
	const ConstDiv cdiv(fast / slow);
	SlimSink sink;
	const int carried = sin[s].deemph_avg;
	// the filter's state at j0
	uint32_t v = (uint32_t)(carried + 32768), v2 = v;
	if (pl.j0 > 0) {
		bool settled = false;
		if (pl.j0 > W) {
			v = 0; v2 = 65535;
			slim_walk<MAGIC, true>(r + pl.j0 - W, W, v, v2, ds, sink, slow, fast, cdiv);
			settled = v == v2;
		}
		if (!settled) {
			// close to the run's start, or a stretch of the stream the filter does not forget over: from the carried state
			v = (uint32_t)(carried + 32768); v2 = v;
			slim_walk<MAGIC, true>(r, pl.j0, v, v2, ds, sink, slow, fast, cdiv);
		}
	}
	sink.init(B + (size_t)s * bstride, pl.m0, pl.phi0, pl.m0 == 0 && pl.j0 == 0 ? (uint32_t)sin[s].now_lpr : 0u, vec != 0);
	slim_walk<MAGIC, false>(r + pl.j0, pl.count & (kSlimLast - 1), v, v2, ds, sink, slow, fast, cdiv);
	sink.finish();
	if (pl.count & kSlimLast) {
		sout[s].deemph_avg = (int)v - 32768;
		sout[s].now_lpr = (int)sink.acc;
	}
}

// ---- deemph_filter alone, in one pass (round 5) ---------------------------------------------------------------
// `rtl_fm -M fm -s 24k -E deemp` - the everyday narrow-band line - ends in deemph_filter with nothing behind it, and took
// the four passes of the time-parallel form (A1, A2, B, C: the run read three times, written once, 0.77 ms of kernel time
// per step of 256 streams x 200 K samples beside a 0.77 ms front end).  The same speculation as k_deemph_spec_lpr: a lane
// per chunk settles its incoming state over the W samples before the chunk (two walks from the two extreme states: where
// they have met, that is the state) and then filters its chunk - OUT of place (the neighbour's settling reads this
// chunk's unfiltered samples): 128 bytes in, 128 bytes out per lane and round, the stores of a line back to back.  A
// stream that cannot settle (silence) or whose carried state lies outside int16 is redone by its workgroup's first lane
// with the reference's loop.  src and dst rows share their alignment modulo 16 bytes (the host checks).
template <int MAGIC>
__device__ __forceinline__ uint32_t deemph_walk_copy(const int16_t *r, int16_t *d, int n, uint32_t avgb, const DeemphStep &ds)
{
	auto one = [&](int k) {
		avgb = ds.step<MAGIC>((uint32_t)(uint16_t)r[k] ^ 0x8000u, avgb);
		d[k] = (int16_t)(uint16_t)(avgb ^ 0x8000u);
	};
	int k = 0;
	const int head = (int)(((16 - ((uintptr_t)r & 15)) & 15) >> 1);
	for (; k < head && k < n; k++) one(k);
	auto group = [&](uint4 &g) {
		uint32_t w[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const uint32_t b = w[j] ^ 0x80008000u;
			avgb = ds.step<MAGIC>(b & 0xffffu, avgb);
			const uint32_t lo = avgb;
			avgb = ds.step<MAGIC>(b >> 16, avgb);
			w[j] = ((lo & 0xffffu) | (avgb << 16)) ^ 0x80008000u;
		}
		g = make_uint4(w[0], w[1], w[2], w[3]);
	};
	if (k + 64 <= n) {
		uint4 cur[8], nxt[8];
#pragma unroll
		for (int j = 0; j < 8; j++) cur[j] = reinterpret_cast<const uint4 *>(r + k)[j];
		for (; k + 64 <= n; k += 64) {
			const bool more = k + 128 <= n;
			const uint4 *np = reinterpret_cast<const uint4 *>(r + (more ? k + 64 : k));
#pragma unroll
			for (int j = 0; j < 8; j++) nxt[j] = np[j];
#pragma unroll
			for (int j = 0; j < 8; j++) group(cur[j]);
#pragma unroll
			for (int j = 0; j < 8; j++) reinterpret_cast<uint4 *>(d + k)[j] = cur[j];
#pragma unroll
			for (int j = 0; j < 8; j++) cur[j] = nxt[j];
		}
	}
	for (; k + 8 <= n; k += 8) {
		uint4 g = *reinterpret_cast<const uint4 *>(r + k);
		group(g);
		*reinterpret_cast<uint4 *>(d + k) = g;
	}
	for (; k < n; k++) one(k);
	return avgb;
}

template <int MAGIC>
__global__ void __launch_bounds__(kSpecLprThreads)
k_deemph_spec(const int16_t *R, size_t rstride, int16_t *__restrict__ B, size_t bstride, int T, const int32_t *__restrict__ cnt,
              int nstreams, DeemphStep ds, int max_chunks, int L, int W, const state_t *__restrict__ sin, state_t *__restrict__ sout)
{
#if RTLFM_TAIL_PRIO >= 0
	__builtin_amdgcn_s_setprio(RTLFM_TAIL_PRIO);
#endif
	__shared__ int unsettled[kSpecLprThreads];  // per stream of this workgroup
	const int nthreads = (int)blockDim.x, tid = (int)threadIdx.x;
	const int spw = max_chunks >= nthreads ? 1 : nthreads / max_chunks;  // streams per workgroup
	unsettled[tid] = 0;
	__syncthreads();
	const int sl = spw == 1 ? 0 : tid / max_chunks;
	const size_t s = (size_t)blockIdx.x * spw + sl;
	const bool live = sl < spw && s < (size_t)nstreams;
	int n = 0, nc = 0, head = 0, carried = 0;
	const int16_t *r = nullptr;
	int16_t *d = nullptr;
	bool plain = false;
	if (live) {
		n = cnt ? cnt[s] : T;
		r = R + s * rstride;
		d = B + s * bstride;
		head = (int)(((16 - ((uintptr_t)r & 15)) & 15) >> 1);
		nc = deemph_chunks(n, head, L);
		carried = sin[s].deemph_avg;
		plain = (uint32_t)(carried + 32768) > 65535u;
	}
	const int c_first = spw == 1 ? tid : tid - sl * max_chunks, c_step = spw == 1 ? nthreads : max_chunks;
	if (live && !plain) {
		for (int c = c_first; c < nc; c += c_step) {
			int begin, end;
			deemph_chunk_range(c, n, head, L, begin, end);
			bool settled = true;
			uint32_t v = (uint32_t)(carried + 32768);
			if (begin > 0) {
				if (begin <= W) {
					uint32_t v2 = v;
					deemph_walk_pair<MAGIC>(r, begin, v, v2, ds);  // close to the start of the run: from the carried state itself
				} else {
					uint32_t lo = 0, hi = 65535;
					deemph_walk_pair<MAGIC>(r + begin - W, W, lo, hi, ds);
					settled = lo == hi;
					v = lo;
				}
			}
			if (!settled) { unsettled[sl] = 1; continue; }
			const int v_end = (int)deemph_walk_copy<MAGIC>(r + begin, d + begin, end - begin, v, ds) - 32768;
			if (c == nc - 1) sout[s].deemph_avg = v_end;
		}
	}
	__syncthreads();
	if (!live || c_first != 0 || !(plain || unsettled[sl])) return;
	// the reference's loop over the whole run, on the stream's first lane (deemph_filter, src/rtl_fm.c:1011-1026)
	if (plain) {
		int avg = carried;
		const int a = (int)ds.a, half = a / 2;
		for (int k = 0; k < n; k++) {
			const int dd = r[k] - avg;
			avg += dd > 0 ? (dd + half) / a : (dd - half) / a;
			d[k] = (int16_t)avg;
		}
		sout[s].deemph_avg = avg;
	} else {
		sout[s].deemph_avg = (int)deemph_walk_copy<MAGIC>(r, d, n, (uint32_t)(carried + 32768), ds) - 32768;
	}
}

// the outputs that straddle chunk boundaries, the carried accumulator and the output count
__global__ void __launch_bounds__(64)
k_lpr_fixup(const int16_t *__restrict__ R, size_t rstride, int T, const int32_t *__restrict__ cnt, int nstreams, int max_chunks,
            int L, const LprChunk *__restrict__ lc, int16_t *__restrict__ B, size_t bstride, int fast, int slow,
            const state_t *__restrict__ sin, state_t *__restrict__ sout, int32_t *__restrict__ cnt_out)
{
	for_stream_chunks(nstreams, max_chunks, 1, [&](const size_t s, const int c, int) {
	const int n = cnt ? cnt[s] : T;
	const int16_t *r = R + s * rstride;
	const int head = (int)(((16 - ((uintptr_t)r & 15)) & 15) >> 1);
	const int nc = deemph_chunks(n, head, L);
	if (c >= nc) return;
	const LprChunk *l0 = lc + s * max_chunks;
	const int div = fast / slow;
	// what the chunks before c left in the accumulator since their last emission
	auto carried = [&](int upto) {
		uint32_t sum = 0;
		for (int j = upto; j >= 0; j--) {
			sum += l0[j].tail;
			if (l0[j].mfirst >= 0) break;
		}
		return sum;
	};
	if (c > 0 && l0[c].mfirst >= 0) B[s * bstride + l0[c].mfirst] = (int16_t)((int)(l0[c].head + carried(c - 1)) / div);
	if (c == nc - 1) {
		const long long p0 = sin[s].prev_lpr_index;
		const long long tot = p0 + (long long)n * slow;
		const int E = (p0 >= 0 && p0 < fast) ? (int)floor_div_pos(tot, fast) : (int)(tot / fast);
		sout[s].now_lpr = (int)carried(c);
		sout[s].prev_lpr_index = (int)(tot - (long long)E * fast);
		cnt_out[s] = E;
	}
	});
}

// dc_block_audio_filter (src/rtl_fm.c:1028-1041) works on whatever result_len a buffer has.
// Buffer b of a stream owns the decimated samples [dec_block_begin(b), dec_block_begin(b+1)) of
// the run (N input samples per buffer, boxcar D with prev_index carried in; D == 1 describes a
// uniform count N per buffer).  Block sums in parallel ...
__global__ void __launch_bounds__(256)
k_adc_sums(const int16_t *__restrict__ R, size_t rstride, int N, int D, int nblocks,
           const state_t *__restrict__ sin, long long *__restrict__ sums)
{
	const int sb = blockIdx.x;
	const int b = sb % nblocks;
	const size_t s = sb / nblocks;
	const int p0 = D > 1 ? sin[s].prev_index : 0;
	const int t0 = dec_block_begin(b, N, D, p0), t1 = dec_block_begin(b + 1, N, D, p0);
	const int16_t *r = R + s * rstride + t0;
	int acc = 0;  // at most 2^17 samples of 16 bits per lane: 32 bits hold it
	for (int k = threadIdx.x; k < t1 - t0; k += blockDim.x) acc += r[k];
	long long a64 = acc;
	for (int off = 32; off > 0; off >>= 1) a64 += __shfl_down(a64, off, 64);
	__shared__ long long red[4];
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a64;
	__syncthreads();
	if (threadIdx.x == 0) sums[sb] = red[0] + red[1] + red[2] + red[3];
}
// ... the smoothing recurrence sequentially per stream ...
__global__ void k_adc_smooth(const long long *__restrict__ sums, int N, int D, int nblocks, int nstreams,
                             int k, const state_t *__restrict__ sin, state_t *__restrict__ sout,
                             int32_t *__restrict__ avg)
{
	RTLFM_GRID_STRIDE(s, nstreams) {
		int prev = sin[s].dc_avg;
		const int p0 = D > 1 ? sin[s].prev_index : 0;
		for (int b = 0; b < nblocks; b++) {
			const int len = dec_block_begin(b + 1, N, D, p0) - dec_block_begin(b, N, D, p0);
			int m = (int)(sums[s * nblocks + b] / len);
			m = (m + prev * k) / (k + 1);
			avg[s * nblocks + b] = m;
			prev = m;
		}
		sout[s].dc_avg = prev;
	}
}
// ... and the subtraction (T = upper bound of the per-stream count).
__global__ void __launch_bounds__(256)
k_adc_apply(int16_t *__restrict__ R, size_t rstride, int N, int D, int nblocks, int nstreams, int T,
            const state_t *__restrict__ sin, const int32_t *__restrict__ avg)
{
	const size_t total = (size_t)nstreams * T;
	RTLFM_GRID_STRIDE(g, total) {
		const int t = (int)(g % T);
		const size_t s = g / T;
		const int p0 = D > 1 ? sin[s].prev_index : 0;
		const int b = dec_block_of(t, N, D, p0);
		if (b >= nblocks) continue;
		int16_t *r = R + s * rstride + t;
		*r = (int16_t)(*r - avg[s * nblocks + b]);
	}
}

// Round 5: the smoothing and the subtraction in ONE launch behind the sums, a workgroup per (stream, buffer).  The
// recurrence avg_b = (mean_b + k avg_{b-1}) / (k + 1) over a stream's buffers is two integer operations per buffer, so
// every workgroup simply runs it from the carried dc_avg up to its own buffer (k_adc_smooth did that in a launch of its
// own, one lane per stream), and the subtraction knows its buffer's extent instead of finding every sample's buffer with a
// 64-bit division (k_adc_apply).  The last buffer's workgroup leaves dc_avg.  Same integers.
// (A first form - one workgroup per stream doing all three steps - was SLOWER at 256 streams x 64 buffers: 256 workgroups
// do not fill the GPU; rtl_fm -M am -s 12k -E dc 0.82 -> 0.90 ms per step.)
__global__ void __launch_bounds__(256)
k_adc_smooth_apply(int16_t *__restrict__ R, size_t rstride, int N, int D, int nblocks, int k, const long long *__restrict__ sums,
                   const state_t *__restrict__ sin, state_t *__restrict__ sout)
{
	extern __shared__ int32_t adc_means[];  // [nblocks] block means, then the workgroup's average in adc_means[nblocks]
	const int sb = blockIdx.x;
	const int b = sb % nblocks;
	const size_t s = sb / nblocks;
	const int p0 = D > 1 ? sin[s].prev_index : 0;
	// the means of buffers 0 .. b, a lane each (the 64-bit division of a sum by its count is the expensive part: in
	// parallel), then the recurrence on one lane: two 32-bit operations per buffer
	for (int j = threadIdx.x; j <= b; j += 256) {
		const int len = dec_block_begin(j + 1, N, D, p0) - dec_block_begin(j, N, D, p0);
		adc_means[j] = (int)(sums[s * nblocks + j] / len);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		int avg = sin[s].dc_avg;
		for (int j = 0; j <= b; j++) avg = (adc_means[j] + avg * k) / (k + 1);
		adc_means[nblocks] = avg;
		if (b == nblocks - 1) sout[s].dc_avg = avg;
	}
	__syncthreads();
	const int avg = adc_means[nblocks];
	const int t0 = dec_block_begin(b, N, D, p0), t1 = dec_block_begin(b + 1, N, D, p0);
	int16_t *r = R + s * rstride;
	for (int t = t0 + (int)threadIdx.x; t < t1; t += 256) r[t] = (int16_t)(r[t] - avg);
}

// low_pass_real (src/rtl_fm.c:755-775) in closed form over the stream's run:
// with phase p0 = prev_lpr_index (< fast) and slow <= fast, the number of
// outputs after i+1 inputs is floor((p0 + (i+1)*slow)/fast); output m is
// emitted at input ceil(((m+1)*fast - p0)/slow) - 1 and is the sum since the
// previous emission (plus the carried now_lpr for m == 0) divided by
// fast/slow (truncating).  n_in[s] (or T) inputs; cnt_out[s] outputs.
// One thread per output m: the number of inputs consumed when output m - 1 left costs one division;
// from there the thread walks the reference's own phase accumulator (prev_lpr_index += slow, emit
// when it reaches fast) until its output leaves - or the inputs run out, which makes it the thread
// that keeps the state.  Neighbouring threads read neighbouring inputs.
__global__ void __launch_bounds__(256)
k_low_pass_real(const int16_t *__restrict__ A, size_t astride, int16_t *__restrict__ B, size_t bstride,
                int T, const int32_t *__restrict__ n_in, int nstreams, int fast, int slow,
                const state_t *__restrict__ sin, state_t *__restrict__ sout, int32_t *__restrict__ cnt_out)
{
	const int div = fast / slow;
	const int maxout = (int)(((long long)fast - 1 + (long long)T * slow) / fast);
	const size_t total = (size_t)nstreams * (maxout + 1);
	const bool small = total < (1ull << 32);
	RTLFM_GRID_STRIDE(g, total) {
		int m;
		size_t s;
		if (small) {
			const uint32_t s32 = (uint32_t)g / (uint32_t)(maxout + 1);
			m = (int)((uint32_t)g - s32 * (uint32_t)(maxout + 1));
			s = s32;
		} else {
			m = (int)(g % (maxout + 1));
			s = g / (maxout + 1);
		}
		const int n = n_in ? n_in[s] : T;
		const long long p0 = sin[s].prev_lpr_index;
		// p0 < fast and n * slow < 2^31 * 2^20: everything below stays under 2^52 for any run the
		// library accepts (p0 comes from the carried state: outside [0, fast) only if injected)
		const bool fits = p0 >= 0 && p0 < fast;
		long long k = 0;  // inputs consumed once output m - 1 is out
		if (m > 0) {
			const long long need = (long long)m * fast - p0;
			k = fits ? floor_div_pos(need + slow - 1, slow) : (need + slow - 1) / slow;
			if (k > n) continue;  // output m - 1 does not exist
		}
		long long ph = p0 + k * slow - (long long)m * fast;  // the accumulator's phase before input k
		uint32_t acc = m == 0 ? (uint32_t)sin[s].now_lpr : 0u;
		const int16_t *a = A + s * astride;
		bool out = false;
		while (k < n) {
			acc += (uint32_t)(int)a[k++];
			ph += slow;
			if (ph >= fast) { out = true; break; }
		}
		if (out) {
			B[s * bstride + m] = (int16_t)((int)acc / div);
		} else {
			// the inputs ran out before output m: it is what stays in the accumulator
			sout[s].now_lpr = (int)acc;
			sout[s].prev_lpr_index = (int)ph;
			cnt_out[s] = m;
		}
	}
}

// Extents of buffer b for the per-buffer resampler arbitrary_resample(result, result, len1,
// len1 * rate_out2 / rate_out) (the commented-out call, src/rtl_fm.c:1270): len1 is whatever
// result_len the buffer has.  Behind a boxcar that does not divide the buffer len1 takes the two
// values nlo = N / D and nlo + 1, so the start of buffer b's output in the concatenated result is
// b * len2(nlo) + (#long buffers before b) * (len2(nlo + 1) - len2(nlo)).
struct ArbExtent { int in0, len1, out0, len2; };
struct ArbPlan { int N, D, nlo, l2lo, l2hi; };  // nlo = N / D, l2lo = len2(nlo), l2hi = len2(nlo + 1): from the host
__device__ __forceinline__ ArbExtent arb_extent(int b, const ArbPlan &a, int p0)
{
	ArbExtent e;
	if (a.D == 1) {  // uniform count: no boxcar phase, no division
		e.in0 = b * a.N; e.len1 = a.N; e.out0 = b * a.l2lo; e.len2 = a.l2lo;
		return e;
	}
	e.in0 = dec_block_begin(b, a.N, a.D, p0);
	e.len1 = dec_block_begin(b + 1, a.N, a.D, p0) - e.in0;
	const int nlong = e.in0 - b * a.nlo;
	e.out0 = b * a.l2lo + nlong * (a.l2hi - a.l2lo);
	e.len2 = e.len1 == a.nlo ? a.l2lo : a.l2hi;
	return e;
}

// arbitrary_upsample (src/rtl_fm.c:1114-1135), stateless per block, closed
// form per output j.  The reference advances (i, tick) after each output:
// tick += len1; if (tick > len2) {tick -= len2; i++}; clamp at the end.  Before
// the clamp engages, after j outputs the total advance is j*len1 = (i-1)*len2 +
// tick with 0 < tick <= len2 (tick == 0 only for j == 0).  The clamp sets
// (i = len1-1, tick = len2) and is sticky.  len2max = the larger of the two per-buffer lengths.
__global__ void __launch_bounds__(256)
k_arb_upsample(const int16_t *__restrict__ A, size_t astride, int16_t *__restrict__ B, size_t bstride,
               ArbPlan ap, int len2max, int nblocks, int nstreams,
               const state_t *__restrict__ sin, int32_t *__restrict__ cnt_out)
{
	const size_t total = (size_t)nstreams * nblocks * len2max;
	const bool small = (long long)(ap.nlo + 1) * len2max < (1ll << 31) && total < (1ull << 32);
	RTLFM_GRID_STRIDE(g, total) {
		int j, b;
		size_t s;
		if (small) {
			const uint32_t g32 = (uint32_t)g, sb = g32 / (uint32_t)len2max;
			j = (int)(g32 - sb * (uint32_t)len2max);
			s = sb / (uint32_t)nblocks;
			b = (int)(sb - (uint32_t)s * (uint32_t)nblocks);
		} else {
			j = (int)(g % len2max);
			const size_t sb = g / len2max;
			b = (int)(sb % nblocks);
			s = sb / nblocks;
		}
		const int p0 = ap.D > 1 ? sin[s].prev_index : 0;
		const ArbExtent e = arb_extent(b, ap, p0);
		if (b == nblocks - 1 && j == 0 && cnt_out) cnt_out[s] = e.out0 + e.len2;
		if (j >= e.len2 || !(e.len1 < e.len2)) continue;  // len1 >= len2: arbitrary_downsample's buffer
		const int len1 = e.len1, len2 = e.len2;
		const int16_t *a = A + s * astride + e.in0;
		int i, tick;
		if (j == 0) {
			i = 1; tick = 0;
		} else if (small) {
			// j * len1 < len1 * len2 < 2^31: one 32-bit division instead of a 64-bit one
			const uint32_t adv = (uint32_t)j * (uint32_t)len1;
			const uint32_t q = (adv - 1u) / (uint32_t)len2;
			i = 1 + (int)q;
			tick = (int)(adv - q * (uint32_t)len2);
		} else {
			const long long adv = (long long)j * len1;
			const long long q = (adv - 1) / len2;
			i = 1 + (int)q;
			tick = (int)(adv - q * len2);
		}
		if (i >= len1) { i = len1 - 1; tick = len2; }
		double frac = (double)tick / (double)len2;
		B[s * bstride + (size_t)e.out0 + j] = (int16_t)(a[i - 1] * (1 - frac) + a[i] * frac);
	}
}

// ---- deemph_filter + arbitrary_upsample in one pass (config 3's audio tail) ---------------------
// deemph_filter in place (four passes: the run is read twice and written once) followed by
// k_arb_upsample (read again, written at len2 / len1) moves 5.4 bytes per demodulated sample; this
// moves 3.4: the run is read once and only the resampled output is written.  A one-wave workgroup
// takes a span of 64 chunks of 32 samples of one stream, plus the W samples before it, into LDS
// with 16-byte loads; lane l settles the filter state at the start of chunk l from the W samples
// before it as k_deemph_spec_lpr does (both extreme states walked until they meet; a stream where
// they do not raises fallback[] and goes through the separate kernels afterwards), filters its
// chunk in place in LDS, and then all lanes produce the span's outputs, coalesced: output j of a
// buffer needs the filtered samples i - 1 and i with (i, tick) as the reference's loop has them
// after j rounds, here advanced by 64 outputs at a time per lane.  frac = tick / len2 without a
// division: q0 = tick * rinv, q = fma(fma(-q0, len2, tick), rinv, q0) with rinv = RN(1 / len2) is
// the correctly rounded quotient - the host checks that against the division for every tick of
// this len2 before it passes `fast` (0: divide).  Uniform buffers only (N samples each).
// LDS: (W / 32 + 64) chunks at a stride of 40 int16 (16-byte accesses of 16 consecutive lanes fall
// on different banks); 5.4 KB at W = 96, little enough to find room on a CU next to the front
// end's waves, which hold nearly all of its LDS (a first form with 64 x 128 samples and a table of
// fractions, 40 KB, waited for the front end to finish instead of running beside it).
constexpr int kArbChunk = 32, kArbStride = 40;
__device__ __forceinline__ int arb_first_output(int i, int len1, int len2)
{
	// first output whose left neighbour index is i - 1 (i.e. that reads samples i - 1 and i)
	if (i <= 1) return 0;
	if (i >= len1) return len2;
	return (int)(((uint32_t)(i - 1) * (uint32_t)len2) / (uint32_t)len1) + 1;
}

// arbitrary_upsample (src/rtl_fm.c:1114-1135) of buffer b of a stream by one wave: N samples at a -> len2 at bo
__device__ __forceinline__ void arb_upsample_wave(const int16_t *a, int16_t *bo, int len1, int len2, int lane)
{
	for (int j = lane; j < len2; j += 64) {
		int i = 1, tick = 0;
		if (j) {
			const uint32_t adv = (uint32_t)j * (uint32_t)len1;
			const uint32_t q = (adv - 1u) / (uint32_t)len2;
			i = 1 + (int)q;
			tick = (int)(adv - q * (uint32_t)len2);
		}
		if (i >= len1) { i = len1 - 1; tick = len2; }
		const double frac = (double)tick / (double)len2;
		bo[j] = (int16_t)(a[i - 1] * (1 - frac) + a[i] * frac);
	}
}

// One workgroup of `wpw` waves per stream; wave w takes the spans w, w + wpw, ... of its stream, each wave with an
// LDS region of its own (the span's traffic is wave-private: wave barriers, no workgroup barrier inside the loop).
constexpr int kSpecArbMaxWaves = 8;
// (i, frac) of one output as one 16-byte entry (k_deemph_arb_span; the host fills it beside tab_i / tab_frac)
struct alignas(16) ArbTab { double frac; int32_t i; int32_t pad; };
constexpr int kArbTabGap = 64;  // tab_i: i of every output, a gap, then k_deemph_spec_arb's padded offset 2 i + 16 (i >> 5) - 2
static_assert(sizeof(ArbTab) == 16, "one 16-byte load per output");
// (A/B builds, tools/build_variant.sh arb64 -DRTLFM_ARB_WAVES8: the kernel held to 64 registers so that TWO of its waves fit
// the hole one front-end wave leaves on a SIMD - LAB.md I.28)
#ifndef RTLFM_ARB_LOOP
#define RTLFM_ARB_LOOP 1  // 0: round 5's loads and resampling loop (A/B builds, LAB.md I.31)
#endif
#ifndef RTLFM_ARB_BATCH
#define RTLFM_ARB_BATCH 4
#endif
#ifndef RTLFM_ARB_SCALAR_WAVE
#define RTLFM_ARB_SCALAR_WAVE 1
#endif
#ifndef RTLFM_ARB_ADDITIVE
#define RTLFM_ARB_ADDITIVE 1
#endif
#ifndef RTLFM_ARB_ABLATE
#define RTLFM_ARB_ABLATE 0  // timing builds only: 1 no resampling, 2 no settling walk, 4 no filter walk, 8 no loads
#endif
#ifdef RTLFM_ARB_WAVES8
#define RTLFM_ARB_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))
#else
#define RTLFM_ARB_ATTR
#endif
template <int MAGIC>
__global__ void __launch_bounds__(64 * kSpecArbMaxWaves) RTLFM_ARB_ATTR
k_deemph_spec_arb(int16_t *R, size_t rstride, int T, int nstreams, DeemphStep ds, int W, int spans,
                  int N, int len2, int nblocks, const int32_t *__restrict__ tab_i, const double *__restrict__ tab_frac,
                  int16_t *__restrict__ B, size_t bstride,
                  const state_t *__restrict__ sin, state_t *__restrict__ sout, size_t lds_per_wave,
                  int32_t *__restrict__ cnt_out)
{
#if RTLFM_TAIL_PRIO >= 0
	__builtin_amdgcn_s_setprio(RTLFM_TAIL_PRIO);
#endif
	extern __shared__ uint4 arb_lds[];
	__shared__ int wg_unsettled;
	// (the wave's number in a scalar register: everything a span derives from it - k0, the buffers it intersects, their
	// first and last outputs with their divisions - is then scalar work)
	const int lane = (int)threadIdx.x & 63, wave = RTLFM_ARB_SCALAR_WAVE ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : (int)threadIdx.x >> 6, wpw = (int)blockDim.x >> 6;
	int16_t *y = reinterpret_cast<int16_t *>(reinterpret_cast<char *>(arb_lds) + (size_t)wave * lds_per_wave);
	const size_t s = blockIdx.x;
	constexpr int C = kArbChunk, Cp = kArbStride, span = 64 * C;
	const int pre = W / C;  // chunks before the span
	int16_t *r = R + s * rstride;
	const int carried = sin[s].deemph_avg;
	if (threadIdx.x == 0) {
		wg_unsettled = 0;
		if (cnt_out) cnt_out[s] = nblocks * len2;
	}
	__syncthreads();
	// a carried state outside int16 (only rtlfm_gpu_state_set can do that) has no biased form: the plain loop below
	const bool plain = (uint32_t)(carried + 32768) > 65535u;
	auto wave_sync = [&]() {  // LDS traffic of one wave is in order: a wave barrier + the fences the compiler needs
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	};
	for (int sp = wave; sp < spans && !plain; sp += wpw) {
	const int k0 = sp * span, k1 = min(k0 + span, T);
	bool unsettled = false;
	wave_sync();  // the span before is done with this wave's LDS region
	// samples [k0 - W, k0 + span), zero outside the run
	if (!(RTLFM_ARB_ABLATE & 8)) {
#if RTLFM_ARB_LOOP == 1
	constexpr int kGroupsAtOnce = 5;  // W = 128: 272 groups of eight samples, 4.25 per lane
	const int G = (pre + 64) * (C / 8);
	if (k0 >= W && k0 + span <= T && G <= kGroupsAtOnce * 64) {
		// the window lies inside the run (all but a stream's first and last span): a lane's loads in flight together,
		// then its writes - one trip to memory per span instead of one per group
		uint4 v5[kGroupsAtOnce];
		const uint4 *src = reinterpret_cast<const uint4 *>(r + (k0 - W));
#pragma unroll
		for (int u = 0; u < kGroupsAtOnce; u++) {
			const int g = lane + 64 * u;
			v5[u] = g < G ? src[g] : make_uint4(0, 0, 0, 0);
		}
#pragma unroll
		for (int u = 0; u < kGroupsAtOnce; u++) {
			const int g = lane + 64 * u;
			if (g < G) *reinterpret_cast<uint4 *>(y + (g >> 2) * Cp + (g & 3) * 8) = v5[u];
		}
	} else
#endif
	for (int g = lane; g < (pre + 64) * (C / 8); g += 64) {
		const int q = g >> 2, w = g & 3;
		const int k = k0 - W + g * 8;
		uint4 v = make_uint4(0, 0, 0, 0);
		if (k >= 0 && k + 8 <= T) v = *reinterpret_cast<const uint4 *>(r + k);
		else if (k + 8 > 0 && k < T) {
			uint32_t t[4] = {0, 0, 0, 0};
			for (int j = 0; j < 8; j++)
				if (k + j >= 0 && k + j < T) t[j >> 1] |= (uint32_t)(uint16_t)r[k + j] << (16 * (j & 1));
			v = make_uint4(t[0], t[1], t[2], t[3]);
		}
		*reinterpret_cast<uint4 *>(y + q * Cp + w * 8) = v;
	}
	}
	wave_sync();
	// the state at the start of this lane's chunk (chunk pre + lane of the array): from the `pre` chunks before it
	const int begin = k0 + lane * C, end = min(begin + C, T);
	uint32_t v = (uint32_t)(carried + 32768);
	{
		// Both extreme states are walked over the samples before the chunk; where they have met, that is the
		// state, whatever came earlier.  The window GROWS: one chunk (32 samples) first - with a = 2 the two
		// walks of a moving signal meet after 17-23 samples -, and only if some lane of the wave is still
		// undecided two chunks, then all `pre` of them (W samples).  Before, every lane walked W = 128
		// samples twice for its 32: nine filter steps per sample where three do.
		const bool mine = begin > 0 && begin < T;
		uint32_t lo = 0, hi = 65535;
		for (int nch = 1; !(RTLFM_ARB_ABLATE & 2); nch = nch * 2 < pre ? nch * 2 : pre) {
			lo = 0; hi = 65535;
			if (begin < nch * C) lo = hi = v;  // the run starts inside the window: from the carried state, over the samples there are
			for (int c = pre - nch; c < pre; c++) {
				if (begin - (pre - c) * C < 0) continue;
				const uint4 *wp = reinterpret_cast<const uint4 *>(y + (lane + c) * Cp);
#pragma unroll
				for (int g = 0; g < C / 8; g++) {
					const uint4 q4 = wp[g];
					const uint32_t w4[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
					for (int i = 0; i < 4; i++) {
						const uint32_t b2 = w4[i] ^ 0x80008000u;
						lo = ds.step<MAGIC>(b2 & 0xffffu, lo); hi = ds.step<MAGIC>(b2 & 0xffffu, hi);
						lo = ds.step<MAGIC>(b2 >> 16, lo); hi = ds.step<MAGIC>(b2 >> 16, hi);
					}
				}
			}
			if (nch >= pre || !__any(mine && lo != hi)) break;
		}
		if (RTLFM_ARB_ABLATE & 2) lo = hi = v;
		if (mine) {
			unsettled = lo != hi;  // what this workgroup writes for the stream is replaced afterwards
			v = lo;
		}
	}
	wave_sync();
	// the filtered sample before the span (left neighbour of its first sample), then the chunk in place
	if (lane == 0) y[pre * Cp - (Cp - C) - 1] = (int16_t)(uint16_t)(v ^ 0x8000u);
	if (begin < T && !(RTLFM_ARB_ABLATE & 4)) {
		int16_t *cp = y + (pre + lane) * Cp;
		const int cntc = end - begin;
		int k = 0;
		for (; k + 8 <= cntc; k += 8) {
			uint4 q4 = *reinterpret_cast<uint4 *>(cp + k);
			uint32_t w4[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const uint32_t b2 = w4[i] ^ 0x80008000u;
				v = ds.step<MAGIC>(b2 & 0xffffu, v);
				const uint32_t l16 = v;
				v = ds.step<MAGIC>(b2 >> 16, v);
				w4[i] = ((l16 & 0xffffu) | (v << 16)) ^ 0x80008000u;
			}
			*reinterpret_cast<uint4 *>(cp + k) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
		}
		for (; k < cntc; k++) {
			v = ds.step<MAGIC>((uint32_t)(uint16_t)cp[k] ^ 0x8000u, v);
			cp[k] = (int16_t)(uint16_t)(v ^ 0x8000u);
		}
		if (end == T) sout[s].deemph_avg = (int)v - 32768;
#if RTLFM_ARB_LOOP == 1
		// ... and the chunk's last filtered sample once more in the padding in front of the next chunk: the resampler
		// finds sample p - 1 two bytes below sample p wherever p lies
		cp[Cp - 1] = (int16_t)(uint16_t)(v ^ 0x8000u);
#endif
	}
#if RTLFM_ARB_LOOP == 1
	if (lane == 0) y[pre * Cp - 1] = y[pre * Cp - (Cp - C) - 1];
#endif
	wave_sync();
	// arbitrary_upsample (src/rtl_fm.c:1114-1135) of the buffers that intersect the span
	const int poff = pre * C;
	auto at = [&](int rel) {  // filtered sample k0 + rel, rel >= -1
		const int p = rel + poff;
		return (int)y[(p >> 5) * Cp + (p & (C - 1))];
	};
	// (i, frac) of output j - what the reference's loop holds when it writes buf2[j] - do not depend on the data:
	// the host walks that loop once per (len1, len2) and leaves them in tab_i / tab_frac (rtlfm_hip.hip), 12 bytes
	// per output that every stream shares.  Computing them per output (a 32-bit division to start, then a carry
	// chain and a two-fma quotient for frac) was 18 of the 35 instructions an output cost.
	const int len1 = N;
	for (int b = k0 / N; b <= (k1 - 1) / N && !(RTLFM_ARB_ABLATE & 1); b++) {
		const int base = b * N;
		const int ia = max(k0, base) - base, ib = min(k1, base + N) - base;
		const int j0 = arb_first_output(ia, len1, len2), j1 = arb_first_output(ib, len1, len2);
		int j = j0 + lane;
		if (j >= j1) continue;
		int16_t *bo = B + s * bstride + (size_t)b * len2;
		const int rel0 = base - k0;
		const int32_t *ti = tab_i + j;
		const double *tf = tab_frac + j;
#if RTLFM_ARB_LOOP == 1
		// the padded address of sample p is 2 p + 16 (p >> 5), sample p - 1 two bytes below it (the filter left every
		// chunk's last sample in the padding behind it as well).  Where the buffer starts on a chunk boundary of the span
		// (buffers of a multiple of 32 samples) that address is additive: the host left 2 i + 16 (i >> 5) - 2 behind the
		// i of tab_i (from entry len2 + kArbTabGap on), and a sample's address is one addition.
		// (i and frac in tables of their own: a wave's loads touch 2 + 4 lines per output; as one 16-byte entry 8 - and
		// the loop is as much the L1's as the VALU's: with the entry split 4 + 8 + 4 into two loads of eight lines each,
		// 30 us of 65 more, LAB.md I.31)
		static_assert(C == 32 && Cp == 40, "the address form");
		const char *yb = reinterpret_cast<const char *>(y);
		const int off = rel0 + poff;
		const bool additive = RTLFM_ARB_ADDITIVE && (off & (C - 1)) == 0;
		const char *yo = yb + (2 * off + ((off >> 5) << 4));
		// whole rounds of U outputs per lane while the wave has them (a uniform test: no guard per output) - the U entries
		// requested together, then the arithmetic -, the rest one by one
		auto rounds = [&](auto additive_t) {
			constexpr bool ADD = decltype(additive_t)::value;
			const int32_t *tq = ADD ? tab_i + (len2 + kArbTabGap) : tab_i;
			auto emit = [&](int ii, double frac, int jj) {
				const int p = off + ii;
				const char *q = ADD ? yo + ii : yb + (2 * p + ((p >> 5) << 4) - 2);
				const int a0 = *reinterpret_cast<const int16_t *>(q), a1 = *reinterpret_cast<const int16_t *>(q + 2);
				bo[jj] = (int16_t)(a0 * (1 - frac) + a1 * frac);
			};
			constexpr int U = RTLFM_ARB_BATCH;
			int jw = j0;
			for (; jw + 64 * U <= j1; jw += 64 * U) {
				const int32_t *pi = tq + jw + lane;
				const double *pf = tab_frac + jw + lane;
				int ei[U];
				double ef[U];
#pragma unroll
				for (int u = 0; u < U; u++) { ei[u] = pi[64 * u]; ef[u] = pf[64 * u]; }
#pragma unroll
				for (int u = 0; u < U; u++) emit(ei[u], ef[u], jw + lane + 64 * u);
			}
			for (j = jw + lane; j < j1; j += 64) emit(tq[j], tab_frac[j], j);
		};
		if (additive) rounds(std::true_type{}); else rounds(std::false_type{});
#else
		for (; j < j1; j += 64, ti += 64, tf += 64) {
			const int ii = *ti;
			const double frac = *tf;
			bo[j] = (int16_t)(at(rel0 + ii - 1) * (1 - frac) + at(rel0 + ii) * frac);
		}
#endif
	}
	if (__any(unsettled) && lane == 0) wg_unsettled = 1;
	}  // spans of this wave
	// ---- anybody who could not settle (or a state outside int16): the reference's sequential loop, by wave 0 ------
	__syncthreads();
	if (wave != 0 || !(plain || wg_unsettled)) return;
	if (lane == 0) {
		// deemph_filter (src/rtl_fm.c:1011-1026) over the run, in place
		if (plain) sout[s].deemph_avg = deemph_plain(r, T, carried, (int)ds.a);
		else sout[s].deemph_avg = (int)deemph_walk<MAGIC, true>(r, T, (uint32_t)(carried + 32768), ds) - 32768;
	}
	__threadfence_block();
	wave_sync();
	for (int b = 0; b < nblocks; b++)
		arb_upsample_wave(r + (size_t)b * N, B + s * bstride + (size_t)b * len2, N, len2, lane);
}

// ---- round 5: the same tail at half the instructions (k_deemph_spec_arb above stays as the cross-check, option
// "arb_span" = 0).  profiles/r04_pmc_c3_k_deemph_spec_arb.txt counted 65 lane-operations per demodulated sample on 1 / 64
// of the data - 16 % of config 3's step, and the step pays a tail's WORK in full.  Where they went and
// what is different here:
//   * the resampler's two samples per output came out of a padded chunk array: shift, multiply, mask and add per look-up,
//     twice per output.  The span lies LINEAR in LDS here (sample k at y[k - (k0 - W)]): one address per output, the two
//     samples at offsets -2 and 0.  The lane-per-chunk walks then read and write 16 bytes at a lane stride of 2 C bytes,
//     four-way conflicted - eight LDS instructions per lane and chunk, against the hundreds of VALU ones saved;
//   * (i, frac) of an output as ONE 16-byte table entry (one load, one pointer to advance) instead of two arrays;
//   * a = 2 (16 and 24 kHz) takes the four-instruction step (DeemphStep::step<3>) instead of six;
//   * the settle window grows in steps of 32 samples whatever the chunk length C is, so C = 64 halves the settling per
//     sample (two walks over 32 samples per chunk of 64).
constexpr int kArbSettle = 32;  // samples per settling step

template <int MAGIC, int C>
__global__ void __launch_bounds__(64 * kSpecArbMaxWaves)
k_deemph_arb_span(int16_t *R, size_t rstride, int T, int nstreams, DeemphStep ds, int W, int spans,
                  int N, int len2, int nblocks, const ArbTab *__restrict__ tab,
                  int16_t *__restrict__ B, size_t bstride,
                  const state_t *__restrict__ sin, state_t *__restrict__ sout, size_t lds_per_wave,
                  int32_t *__restrict__ cnt_out)
{
#if RTLFM_TAIL_PRIO >= 0
	__builtin_amdgcn_s_setprio(RTLFM_TAIL_PRIO);
#endif
	static_assert(C % kArbSettle == 0, "the settling steps end at the chunk's first sample");
	extern __shared__ uint4 arb_lds[];
	__shared__ int wg_unsettled;
	const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6, wpw = (int)blockDim.x >> 6;
	int16_t *y = reinterpret_cast<int16_t *>(reinterpret_cast<char *>(arb_lds) + (size_t)wave * lds_per_wave);
	const size_t s = blockIdx.x;
	constexpr int span = 64 * C;
	int16_t *r = R + s * rstride;
	const int carried = sin[s].deemph_avg;
	if (threadIdx.x == 0) {
		wg_unsettled = 0;
		if (cnt_out) cnt_out[s] = nblocks * len2;
	}
	__syncthreads();
	const bool plain = (uint32_t)(carried + 32768) > 65535u;  // no biased form: the reference's loop below
	auto wave_sync = [&]() {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	};
	auto walk8 = [&](const uint4 &q4, uint32_t &lo, uint32_t &hi) {
		const uint32_t w4[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const uint32_t b2 = w4[i] ^ 0x80008000u;
			lo = ds.step<MAGIC>(b2 & 0xffffu, lo); hi = ds.step<MAGIC>(b2 & 0xffffu, hi);
			lo = ds.step<MAGIC>(b2 >> 16, lo); hi = ds.step<MAGIC>(b2 >> 16, hi);
		}
	};
	for (int sp = wave; sp < spans && !plain; sp += wpw) {
		const int k0 = sp * span, k1 = min(k0 + span, T);
		const int kbase = k0 - W;  // y[p] = sample kbase + p
		bool unsettled = false;
		wave_sync();  // the span before is done with this wave's LDS region
		// samples [k0 - W, k0 + span), zero outside the run: coalesced 16-byte loads, linear in LDS
		for (int g = lane; g < (W + span) / 8; g += 64) {
			const int k = kbase + g * 8;
			uint4 v = make_uint4(0, 0, 0, 0);
			if (k >= 0 && k + 8 <= T) v = *reinterpret_cast<const uint4 *>(r + k);
			else if (k + 8 > 0 && k < T) {
				uint32_t t[4] = {0, 0, 0, 0};
				for (int j = 0; j < 8; j++)
					if (k + j >= 0 && k + j < T) t[j >> 1] |= (uint32_t)(uint16_t)r[k + j] << (16 * (j & 1));
				v = make_uint4(t[0], t[1], t[2], t[3]);
			}
			*reinterpret_cast<uint4 *>(y + g * 8) = v;
		}
		wave_sync();
		const int begin = k0 + lane * C, end = min(begin + C, T);
		uint32_t v = (uint32_t)(carried + 32768);
		{
			// both extreme states over the samples before the chunk: where they have met, that is the state, whatever came
			// earlier (k_deemph_spec_arb).  32 samples first, more only while some lane of the wave is undecided.
			const bool mine = begin > 0 && begin < T;
			uint32_t lo = 0, hi = 65535;
			const int steps_max = W / kArbSettle;
			for (int nw = 1;; nw = nw * 2 < steps_max ? nw * 2 : steps_max) {
				lo = 0; hi = 65535;
				if (begin < nw * kArbSettle) lo = hi = v;  // the run starts inside the window: from the carried state, over the samples there are
				for (int b = nw; b >= 1; b--) {
					const int st = begin - b * kArbSettle;
					if (st < 0) continue;
					const uint4 *wp = reinterpret_cast<const uint4 *>(y + (st - kbase));
#pragma unroll
					for (int g = 0; g < kArbSettle / 8; g++) walk8(wp[g], lo, hi);
				}
				if (nw >= steps_max || !__any(mine && lo != hi)) break;
			}
			if (mine) {
				unsettled = lo != hi;  // what this workgroup writes for the stream is replaced afterwards
				v = lo;
			}
		}
		wave_sync();  // every lane has read what it settles on: the chunks may be filtered in place
		if (lane == 0) y[W - 1] = (int16_t)(uint16_t)(v ^ 0x8000u);  // the filtered sample before the span
		if (begin < T) {
			int16_t *cp = y + W + lane * C;
			const int cntc = end - begin;
			int k = 0;
			for (; k + 8 <= cntc; k += 8) {
				uint4 q4 = *reinterpret_cast<uint4 *>(cp + k);
				uint32_t w4[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
				for (int i = 0; i < 4; i++) {
					const uint32_t b2 = w4[i] ^ 0x80008000u;
					v = ds.step<MAGIC>(b2 & 0xffffu, v);
					const uint32_t l16 = v;
					v = ds.step<MAGIC>(b2 >> 16, v);
					w4[i] = ((l16 & 0xffffu) | (v << 16)) ^ 0x80008000u;
				}
				*reinterpret_cast<uint4 *>(cp + k) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
			}
			for (; k < cntc; k++) {
				v = ds.step<MAGIC>((uint32_t)(uint16_t)cp[k] ^ 0x8000u, v);
				cp[k] = (int16_t)(uint16_t)(v ^ 0x8000u);
			}
			if (end == T) sout[s].deemph_avg = (int)v - 32768;
		}
		wave_sync();
		// arbitrary_upsample (src/rtl_fm.c:1114-1135) of the buffers that intersect the span: output j of a buffer from the
		// filtered samples i - 1 and i, (i, frac) as the reference's loop holds them when it writes buf2[j] (the host walks
		// that loop once per (len1, len2): rtlfm_hip.hip)
		for (int b = k0 / N; b <= (k1 - 1) / N; b++) {
			const int base = b * N;
			const int ia = max(k0, base) - base, ib = min(k1, base + N) - base;
			const int j0 = arb_first_output(ia, N, len2), j1 = arb_first_output(ib, N, len2);
			int16_t *bo = B + s * bstride + (size_t)b * len2;
			const int16_t *yb = y + (base - kbase);  // yb[i] = filtered sample i of buffer b
			// The table entries of the NEXT four outputs of the lane travel while these four are computed: a tail wave beside
			// the front end costs the step the wave slot it holds, not its instructions, and with one dependent
			// trip to L2 per output (44 per span and lane) the wave spent most of its life waiting.
			constexpr int PF = 4;
			ArbTab cur[PF], nxt[PF];
			int j = j0 + lane;
#pragma unroll
			for (int u = 0; u < PF; u++) cur[u] = tab[min(j + 64 * u, len2 - 1)];
			for (; j < j1; j += 64 * PF) {
#pragma unroll
				for (int u = 0; u < PF; u++) nxt[u] = tab[min(j + 64 * (PF + u), len2 - 1)];
#pragma unroll
				for (int u = 0; u < PF; u++) {
					const int ju = j + 64 * u;
					if (ju < j1) {
						const int16_t *pp = yb + cur[u].i;
						bo[ju] = (int16_t)((double)pp[-1] * (1 - cur[u].frac) + (double)pp[0] * cur[u].frac);
					}
				}
#pragma unroll
				for (int u = 0; u < PF; u++) cur[u] = nxt[u];
			}
		}
		if (__any(unsettled) && lane == 0) wg_unsettled = 1;
	}  // spans of this wave
	// ---- anybody who could not settle (or a state outside int16): the reference's sequential loop, by wave 0 ------
	__syncthreads();
	if (wave != 0 || !(plain || wg_unsettled)) return;
	if (lane == 0) {
		// deemph_filter (src/rtl_fm.c:1011-1026) over the run, in place
		if (plain) sout[s].deemph_avg = deemph_plain(r, T, carried, (int)ds.a);
		else sout[s].deemph_avg = (int)deemph_walk<MAGIC, true>(r, T, (uint32_t)(carried + 32768), ds) - 32768;
	}
	__threadfence_block();
	wave_sync();
	for (int b = 0; b < nblocks; b++)
		arb_upsample_wave(r + (size_t)b * N, B + s * bstride + (size_t)b * len2, N, len2, lane);
}

// arbitrary_downsample (src/rtl_fm.c:1137-1166): the double remainder makes it
// order-dependent, so one lane walks one (stream, block) in order.
__global__ void k_arb_downsample(const int16_t *__restrict__ A, size_t astride, int16_t *__restrict__ B,
                                 size_t bstride, ArbPlan ap, int nblocks,
                                 int nstreams, const state_t *__restrict__ sin, int32_t *__restrict__ cnt_out)
{
	const size_t total = (size_t)nstreams * nblocks;
	RTLFM_GRID_STRIDE(g, total) {
		int b = (int)(g % nblocks);
		size_t s = g / nblocks;
		const int p0 = ap.D > 1 ? sin[s].prev_index : 0;
		const ArbExtent e = arb_extent(b, ap, p0);
		if (b == nblocks - 1 && cnt_out) cnt_out[s] = e.out0 + e.len2;
		if (e.len1 < e.len2) continue;  // arbitrary_upsample's buffer
		const int len1 = e.len1, len2 = e.len2;
		const int16_t *b1 = A + s * astride + e.in0;
		int16_t *b2 = B + s * bstride + e.out0;
		int src = 1, j = 0, tick = 0;
		double carry = 0;
		int16_t cur = 0;  // b2[j] while it is being accumulated
		while (j < len2) {
			double frac = 1.0;
			if (tick + len2 > len1) frac = (double)(len1 - tick) / (double)len2;
			cur = (int16_t)(cur + (int16_t)((double)b1[src] * frac + carry));
			carry = (double)b1[src] * (1.0 - frac);
			tick += len2;
			src++;
			if (tick > len1) {
				b2[j] = (int16_t)(cur * len2 / len1);
				j++;
				cur = 0;
				tick -= len1;
			}
			if (src >= len1) { src = len1 - 1; tick = len1; }
		}
	}
}

// Plain strided copy of per-stream results (used when a tail stage left the
// final result in a work buffer).
__global__ void __launch_bounds__(256)
k_copy_i16(const int16_t *__restrict__ A, size_t astride, int16_t *__restrict__ B, size_t bstride, int T,
           const int32_t *__restrict__ cnt, int nstreams)
{
	const size_t total = (size_t)nstreams * T;
	RTLFM_GRID_STRIDE(g, total) {
		int t = (int)(g % T);
		size_t s = g / T;
		if (cnt && t >= cnt[s]) continue;
		B[s * bstride + t] = A[s * astride + t];
	}
}

__global__ void k_fill_cnt(int32_t *cnt, int nstreams, int v)
{
	RTLFM_GRID_STRIDE(s, nstreams) cnt[s] = v;
}
__global__ void k_scale_cnt(int32_t *cnt, int nstreams, int mul, int div)
{
	RTLFM_GRID_STRIDE(s, nstreams) cnt[s] = cnt[s] * mul / div;
}

// The fields the audio tail owns (low_pass_real, deemph_filter, dc_block_audio_filter), carried
// from the previous state copy at the start of a tail that runs on its own stream (rtlfm_hip.hip,
// run_tail): the front end's whole-record copy may have read them while the previous tail was
// still writing.
__global__ void k_tail_state_copy(const state_t *__restrict__ sin, state_t *__restrict__ sout, int nstreams)
{
	RTLFM_GRID_STRIDE(s, nstreams) {
		sout[s].now_lpr = sin[s].now_lpr;
		sout[s].prev_lpr_index = sin[s].prev_lpr_index;
		sout[s].deemph_avg = sin[s].deemph_avg;
		sout[s].dc_avg = sin[s].dc_avg;
	}
}

}  // namespace rtlfm
