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
__device__ __forceinline__ int fir9_tap(const int h[9], const int32_t *t)
{
	uint32_t acc = 0;
	acc += (uint32_t)(h[0] + h[8]) * (uint32_t)t[0];
	acc += (uint32_t)(h[1] + h[7]) * (uint32_t)t[1];
	acc += (uint32_t)(h[2] + h[6]) * (uint32_t)t[2];
	acc += (uint32_t)(h[3] + h[5]) * (uint32_t)t[3];
	acc += (uint32_t)h[4] * (uint32_t)t[4];
	return (int32_t)acc >> 15;
}

// multiply(ar, aj, br, -bj) (src/rtl_fm.c:836-840, called at :846, :877, :898)
__device__ __forceinline__ void conj_product(int ar, int aj, int br, int bj, int &cr, int &cj)
{
	cr = (int)((uint32_t)ar * (uint32_t)br + (uint32_t)aj * (uint32_t)bj);
	cj = (int)((uint32_t)aj * (uint32_t)br - (uint32_t)ar * (uint32_t)bj);
}

// polar_discriminant (src/rtl_fm.c:842-849): fp64 atan2, the literal 3.14159,
// truncation toward zero.
__device__ __forceinline__ int disc_std(int ar, int aj, int br, int bj)
{
	int cr, cj;
	conj_product(ar, aj, br, bj, cr, cj);
	double angle = atan2((double)cj, (double)cr);
	return (int)(angle / 3.14159 * 16384.0);
}

// fast_atan2 (src/rtl_fm.c:851-872) behind polar_disc_fast (:874-879); the
// 4096*(...) products wrap in 32 bits exactly as the x86 build does.
__device__ __forceinline__ int disc_fast(int ar, int aj, int br, int bj)
{
	int x, y;
	conj_product(ar, aj, br, bj, x, y);
	if (x == 0 && y == 0)
		return 0;
	int ay = y < 0 ? (int)(0u - (uint32_t)y) : y;
	int num, den, base;
	if (x >= 0) {
		num = (int)((uint32_t)x - (uint32_t)ay);
		den = (int)((uint32_t)x + (uint32_t)ay);
		base = 4096;
	} else {
		num = (int)((uint32_t)x + (uint32_t)ay);
		den = (int)((uint32_t)ay - (uint32_t)x);
		base = 12288;
	}
	int prod = (int)(4096u * (uint32_t)num);
	int angle = base - (den != 0 ? prod / den : 0);
	return y < 0 ? -angle : angle;
}

// polar_disc_lut (src/rtl_fm.c:894-930) over the host-built atan_lut
// (src/rtl_fm.c:881-892), including the x == 0 fall-through.
__device__ __forceinline__ int disc_lut(int ar, int aj, int br, int bj, const int32_t *__restrict__ lut)
{
	int cr, cj;
	conj_product(ar, aj, br, bj, cr, cj);
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

__device__ __forceinline__ int discriminate(int variant, int ar, int aj, int br, int bj,
                                            const int32_t *__restrict__ lut)
{
	if (variant == RTLFM_ATAN_FAST) return disc_fast(ar, aj, br, bj);
	if (variant == RTLFM_ATAN_LUT) return disc_lut(ar, aj, br, bj, lut);
	return disc_std(ar, aj, br, bj);
}

}  // namespace rtlfm
