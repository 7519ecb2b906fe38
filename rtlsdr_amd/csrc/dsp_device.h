// dsp_device.h — device-side arithmetic shared by the staged and the fused
// kernels.  gfx950 only.  Each helper names the reference lines whose
// arithmetic it reproduces (reference = old-dab/rtlsdr, src/rtl_fm.c).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rtlfm_hip.h"

namespace rtlfm {

// One complex sample as the reference stores it: int16 I at [2n], int16 Q at
// [2n+1], moved as one 32-bit word.
struct iq16 {
	int16_t i, q;
};
static_assert(sizeof(iq16) == 4, "iq16 must be one dword");

__device__ __forceinline__ iq16 unpack_iq(uint32_t w)
{
	iq16 r;
	r.i = (int16_t)(w & 0xffffu);
	r.q = (int16_t)(w >> 16);
	return r;
}
__device__ __forceinline__ uint32_t pack_iq(int i, int q)
{
	return ((uint32_t)i & 0xffffu) | ((uint32_t)q << 16);
}

// cic_9_tables rows, taps 1..5 (src/rtl_fm.c:355-367); symmetry gives the rest.
__constant__ const int32_t k_cic9[11][5] = {
	{0, 0, 0, 0, 0},
	{-156, -97, 2798, -15489, 61019},
	{-128, -568, 5593, -24125, 74126},
	{-129, -639, 6187, -26281, 77511},
	{-122, -612, 6082, -26353, 77818},
	{-120, -602, 6015, -26269, 77757},
	{-120, -582, 5951, -26128, 77542},
	{-119, -580, 5931, -26094, 77505},
	{-119, -578, 5921, -26077, 77484},
	{-119, -577, 5917, -26067, 77473},
	{-199, -362, 5303, -25505, 77489},
};

// fifth_order's tap sum (src/rtl_fm.c:787, :797): int arithmetic, >> 4
// arithmetic, int16 store (wraps from the 8th pass on).
__device__ __forceinline__ int fifth_tap(int a, int b, int c, int d, int e, int f)
{
	return (a + f + 5 * (b + e) + 10 * (c + d)) >> 4;
}

// generic_fir's sum over the nine samples before the current one
// (src/rtl_fm.c:815-821), 32-bit wrap, >> 15.
// The samples are int16 and the taps below 2^17 in magnitude: every factor fits 24 bits, and the low 32 bits of the
// 48-bit product are the reference's wrapping 32-bit product - v_mul_i32_i24 / v_mad_i32_i24 issue at full rate where
// the 32-bit multiply takes four times as long (said explicitly here; where the samples visibly come out of 16-bit
// halves the compiler finds the 24-bit forms by itself).
__device__ __forceinline__ int fir9_tap(const int h[9], const int32_t *t)
{
	int acc = __mul24(h[0] + h[8], t[0]);
	acc = (int)((uint32_t)acc + (uint32_t)__mul24(h[1] + h[7], t[1]));
	acc = (int)((uint32_t)acc + (uint32_t)__mul24(h[2] + h[6], t[2]));
	acc = (int)((uint32_t)acc + (uint32_t)__mul24(h[3] + h[5], t[3]));
	acc = (int)((uint32_t)acc + (uint32_t)__mul24(h[4], t[4]));
	return acc >> 15;
}

// multiply(ar, aj, br, -bj) (src/rtl_fm.c:836-840, called at :846, :877, :898)
__device__ __forceinline__ void conj_product(int ar, int aj, int br, int bj, int &cr, int &cj)
{
	cr = (int)((uint32_t)ar * (uint32_t)br + (uint32_t)aj * (uint32_t)bj);
	cj = (int)((uint32_t)aj * (uint32_t)br - (uint32_t)ar * (uint32_t)bj);
}

// ---- atan2 -> Q14 ----------------------------------------------------------
// am_demod / usb_demod / lsb_demod (src/rtl_fm.c:961-1007) on one decimated sample w = (I, Q):
// the int16 cast comes before the multiplication by output_scale, the product is stored as int16
__device__ __forceinline__ int16_t simple_demod(int mode, uint32_t w, int output_scale)
{
	const iq16 v = unpack_iq(w);
	int16_t base;
	if (mode == RTLFM_MODE_AM) {
		const int pcm = v.i * v.i + v.q * v.q;
		base = (int16_t)sqrt((double)pcm);
	} else if (mode == RTLFM_MODE_USB) {
		base = (int16_t)(v.i + v.q);
	} else {
		base = (int16_t)(v.i - v.q);
	}
	return (int16_t)(int)((uint32_t)(int)base * (uint32_t)output_scale);
}

// polar_discriminant (src/rtl_fm.c:842-849) ends in
//     (int)(atan2((double)cj, (double)cr) / 3.14159 * (1<<14))
// i.e. trunc(theta * K) with K = 16384/3.14159 (note the literal).  The library
// atan2 costs ~135 VALU instructions per call on gfx950; at one call per
// decimated sample it would dominate the fused kernel.  atan2_q14 computes the
// same integer with ~41 instructions, everything in fp64:
//   * octant folding to w = atan(mn/mx), mn <= mx;
//   * mn/mx is located in one of 17 nodes c = i/16 (fp32 estimate), and
//     atan(mn/mx) = atan(c) + atan(t), t = (mn - c*mx)/(mx + c*mn), |t| <= 1/32
//     (numerator and denominator are exact in fp64 for 32-bit inputs);
//   * K*atan(t) = t*(K - K/3 t^2 + K/5 t^4 - K/7 t^6), truncation error
//     K*t^9/9 < 2e-11 LSB; K*atan(c) comes from a 17-entry table rounded from
//     long double.
// The result differs from trunc() of the correctly rounded chain only when
// theta*K lies within ~1e-11 of an integer.
#define RTLFM_ATAN_K 0x1.45f318e7adaf4p+12     /* 16384/3.14159            */
#define RTLFM_ATAN_K3 -0x1.b299768a3ce9bp+10   /* -K/3                     */
#define RTLFM_ATAN_K5 0x1.04c27a52f1590p+10    /*  K/5                     */
#define RTLFM_ATAN_K7 -0x1.74838a2d58c85p+9    /* -K/7                     */
#define RTLFM_ATAN_H 0x1.00000e2bce869p+13     /* (pi/2)*K = 8192.0069...  */
#define RTLFM_ATAN_PIK 0x1.00000e2bce869p+14   /* pi*K     = 16384.0138... */

// K*atan(i/16), i = 0..16
__constant__ const double k_atan_nodes[17] = {
	0x0.0p+0, 0x1.4586b38c3d5d7p+8, 0x1.444486fab2e62p+9, 0x1.e3500e621f45ep+9, 0x1.3f671d1a270f7p+10,
	0x1.8ae69b2cc786ap+10, 0x1.d3c3be67b1e47p+10, 0x1.0cd99bf47b146p+11, 0x1.2e4062951c4edp+11,
	0x1.4e06ba27c2404p+11, 0x1.6c2683974a150p+11, 0x1.88a17177cd652p+11, 0x1.a37f7385017cap+11,
	0x1.bccd369d2515fp+11, 0x1.d49acd9d30f62p+11, 0x1.eafa8d1c75877p+11, 0x1.00000e2bce869p+12,
};

struct AtanNodesConst {
	__device__ __forceinline__ double operator()(int i) const { return k_atan_nodes[i]; }
};

template <class Nodes>
__device__ __forceinline__ int atan2_q14(int y, int x, Nodes nodes)
{
	const double fx = (double)x, fy = (double)y;
	const double ax = fabs(fx), ay = fabs(fy);
	const double mx = fmax(ax, ay), mn = fmin(ax, ay);
	// node index from an fp32 estimate of mn/mx: q <= 1 + 2^-22, so i <= 16;
	// mx == 0 gives NaN -> i = 0 and, further down, NaN -> result 0 (v_cvt_i32_f64)
	const float q = (float)mn * __builtin_amdgcn_rcpf((float)mx);
	const int i = (int)__builtin_fmaf(q, 16.0f, 0.5f);
	const double c = (double)i * 0.0625;
	const double num = __builtin_fma(-c, mx, mn);  // exact
	const double den = __builtin_fma(c, mn, mx);   // exact
	// v_rcp_f64 is good to ~2^-26; one Newton step squares that, which leaves t with a relative
	// error of ~1e-15 and K*atan(t) <= 163 with ~2e-13 absolute - far inside the 1e-11 that the
	// truncated series already costs (RTLFM_ATAN_TWO_NEWTON restores the second step)
	double r = __builtin_amdgcn_rcp(den);
	r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
#if defined(RTLFM_ATAN_TWO_NEWTON)
	r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
#endif
	const double t = num * r;
	const double t2 = t * t;
	double p = __builtin_fma(t2, RTLFM_ATAN_K7, RTLFM_ATAN_K5);
	p = __builtin_fma(t2, p, RTLFM_ATAN_K3);
	p = __builtin_fma(t2, p, RTLFM_ATAN_K);
	const double w = __builtin_fma(t, p, nodes(i));  // K*atan(mn/mx) in [0, 4096.004]
	const bool swap = ay > ax, neg = x < 0;
	// x>=0: swap ? H - w : w ;  x<0: swap ? H + w : PIK - w
	const double base = neg ? (swap ? RTLFM_ATAN_H : RTLFM_ATAN_PIK) : (swap ? RTLFM_ATAN_H : 0.0);
	const double v = (neg != swap) ? base - w : base + w;
	// trunc toward zero is odd-symmetric: give v the sign of y, then convert
	return (int)__builtin_copysign(v, fy);
}

// The library path, kept for A/B checks of atan2_q14 (rtlfm_gpu_selftest_atan2).
__device__ __forceinline__ int atan2_q14_libm(int y, int x)
{
	double angle = atan2((double)y, (double)x);
	return (int)(angle / 3.14159 * 16384.0);
}

// polar_discriminant (src/rtl_fm.c:842-849)
__device__ __forceinline__ int disc_std(int ar, int aj, int br, int bj)
{
	int cr, cj;
	conj_product(ar, aj, br, bj, cr, cj);
	return atan2_q14(cj, cr, AtanNodesConst());
}

// C's truncating int / int (d != 0, not INT_MIN / -1) in ~16 instructions instead of the ~40 of the
// expanded 32-bit division: the quotient of the magnitudes from an fp64 reciprocal (one Newton step,
// off by at most one), the remainder exactly with one fma, one correction either way.
__device__ __forceinline__ int sdiv_trunc(int n, int d)
{
#if defined(RTLFM_SDIV_NATIVE)  // A/B builds: the compiler's expansion
	return n / d;
#endif
	const double an = fabs((double)n), ad = fabs((double)d);
	double r = __builtin_amdgcn_rcp(ad);
	r = __builtin_fma(__builtin_fma(-ad, r, 1.0), r, r);
	double q = __builtin_trunc(an * r);
	const double rem = __builtin_fma(-q, ad, an);  // exact: |rem| < 2 ad
	q = rem < 0.0 ? q - 1.0 : (rem >= ad ? q + 1.0 : q);
	const int qi = (int)(unsigned)q;  // up to 2^31 (INT_MIN / 1)
	return ((n ^ d) < 0) ? (int)(0u - (unsigned)qi) : qi;
}

// fast_atan2 (src/rtl_fm.c:851-872); the 4096*(...) products wrap in 32 bits
// exactly as the x86 build does.
__device__ __forceinline__ int fast_atan2_q14(int y, int x)
{
	// Without branches: the reference's two arms differ in the sign of yabs in the numerator and in
	// the constant, and both denominators are |x| + yabs.  (As `if`s, with the division under one of
	// them, the function cost 41 scalar instructions of exec-mask handling per call and made the
	// -A fast front ends slower than -A std.)  x == y == 0 returns 0 (src/rtl_fm.c:855); a zero
	// denominator otherwise needs yabs and |x| to wrap to 2^31 each, where the reference divides by zero.
	const uint32_t ay = y < 0 ? 0u - (uint32_t)y : (uint32_t)y;
	const bool neg = x < 0;
	const uint32_t ax = neg ? 0u - (uint32_t)x : (uint32_t)x;
#if !defined(RTLFM_FAST_ATAN_GENERAL_ONLY)
	// Round 5 (LAB.md I.12, I.15).  Where nothing wraps - |x|, |y| < 2^18, i.e. everywhere the reference's own arithmetic is still what it was meant to
	// be (4096 (x - yabs) overflows from 2^19 on) - the quotient is at most 4096 and the division is a float reciprocal plus
	// an exact remainder on the 24-bit multiplier: 13 instructions, no fp64 reciprocal, against the 22 of sdiv_trunc.  Both
	// arms are +-(|x| - yabs) over |x| + yabs.  Wave-uniform: one lane outside the range sends the wave through the general
	// form below (the parity suite's full-scale inputs do).  While k_boxcar_scan waited for memory this bought nothing
	// (-0.7 % at /10, 0 at /6); since its outputs leave as whole lines the /6 kernel is bound by its output loop:
	// 1.007 -> 0.929 ms per 4 GiB, the -M wbfm step 1.274 -> 1.19-1.21 (profiles/r05_ab_fast_atan_f32.txt).
	if (__builtin_amdgcn_ballot_w64((ax | ay) >= (1u << 18)) == 0) {
		const int dif = (int)ax - (int)ay;
		const uint32_t un12 = (uint32_t)(dif < 0 ? -dif : dif) << 12, den = ax + ay;
		uint32_t q = (uint32_t)((float)un12 * __builtin_amdgcn_rcpf((float)den));  // off by at most one either way
		const int rem = (int)(un12 - __umul24(q, den));
		q = q - (rem < 0 ? 1u : 0u) + (rem >= (int)den ? 1u : 0u);
		const int sq = ((dif < 0) != neg) ? -(int)q : (int)q;
		const int angle = (neg ? 12288 : 4096) - sq;
		const int r = y < 0 ? -angle : angle;
		return (x | y) == 0 ? 0 : r;
	}
#endif
	const int num = (int)((uint32_t)x + (neg ? ay : 0u - ay));
	const int den = (int)(ax + ay);
	const int prod = (int)(4096u * (uint32_t)num);
	const int q = sdiv_trunc(prod, den);  // den == 0: no trap on the GPU, the value is dropped
	const int angle = (neg ? 12288 : 4096) - (den != 0 ? q : 0);
	const int r = y < 0 ? -angle : angle;
	return (x | y) == 0 ? 0 : r;
}

// polar_disc_fast (src/rtl_fm.c:874-879)
__device__ __forceinline__ int disc_fast(int ar, int aj, int br, int bj)
{
	int x, y;
	conj_product(ar, aj, br, bj, x, y);
	return fast_atan2_q14(y, x);
}

// the body of polar_disc_lut (src/rtl_fm.c:899-930) over the host-built atan_lut
// (src/rtl_fm.c:881-892), including the x == 0 fall-through
__device__ __forceinline__ int lut_atan2_q14(int cj, int cr, const int32_t *__restrict__ lut)
{
	if (cr == 0 || cj == 0) {
		if (cr == 0 && cj == 0) return 0;
		if (cr == 0) return cj > 0 ? 8192 : -8192;
		return cr > 0 ? 0 : 16384;
	}
	int scaled = (int)((uint32_t)cj << 8);
	int x = (scaled == INT32_MIN && cr == -1) ? INT32_MIN : scaled / cr;
	long long mag = x < 0 ? -(long long)x : (long long)x;
	if (mag >= 131072)
		return cj > 0 ? 8192 : -8192;
	if (x > 0)
		return cj > 0 ? lut[x] : lut[x] - 16384;
	return cj > 0 ? 16384 - lut[-x] : -lut[-x];
}

// The same function with the table entry computed instead of gathered:
// atan_lut[i] = (int)(atan(i/256.0)/3.14159*16384) == atan2_q14(i, 256) for every
// i < 131072 (checked exhaustively, tests/test_parity_gpu.py).  Used by the fused kernel,
// where a gather would share the in-order vmcnt with the prefetched tile and force it to
// land.  Branch structure and quirks (x == 0 fall-through, cj << 8 wrap) as above.
template <class Nodes>
__device__ __forceinline__ int lut_atan2_q14_direct(int cj, int cr, Nodes nodes)
{
	if (cr == 0 || cj == 0) {
		if (cr == 0 && cj == 0) return 0;
		if (cr == 0) return cj > 0 ? 8192 : -8192;
		return cr > 0 ? 0 : 16384;
	}
	int scaled = (int)((uint32_t)cj << 8);
	int x = (scaled == INT32_MIN && cr == -1) ? INT32_MIN : sdiv_trunc(scaled, cr);
	long long mag = x < 0 ? -(long long)x : (long long)x;
	if (mag >= 131072)
		return cj > 0 ? 8192 : -8192;
	const int e = atan2_q14((int)mag, 256, nodes);  // atan_lut[|x|]
	if (x > 0)
		return cj > 0 ? e : e - 16384;
	return cj > 0 ? 16384 - e : -e;
}

__device__ __forceinline__ int disc_lut(int ar, int aj, int br, int bj, const int32_t *__restrict__ lut)
{
	int cr, cj;
	conj_product(ar, aj, br, bj, cr, cj);
	return lut_atan2_q14(cj, cr, lut);
}

__device__ __forceinline__ int discriminate(int variant, int ar, int aj, int br, int bj,
                                            const int32_t *__restrict__ lut)
{
	if (variant == RTLFM_ATAN_FAST) return disc_fast(ar, aj, br, bj);
	if (variant == RTLFM_ATAN_LUT) return disc_lut(ar, aj, br, bj, lut);
	return disc_std(ar, aj, br, bj);
}

}  // namespace rtlfm
