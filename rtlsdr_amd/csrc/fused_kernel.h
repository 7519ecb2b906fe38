// fused_kernel.h — the streaming form of rtl_fm's FM chain for gfx950.
//
// ONE launch does, for every stream and every queued callback buffer,
//   u8 -> int16 (-127)        src/rtl_fm.c:1326-1328
//   rotate16_neg90            src/rtl_fm.c:424-434 (phase restarts per buffer)
//   fifth_order x P           src/rtl_fm.c:777-806, 1188-1191 (incl. the per-buffer
//                             history quirk: x[N-1] never reaches the next buffer)
//   generic_fir (9 taps)      src/rtl_fm.c:808-831 (optional)
//   fm_demod                  src/rtl_fm.c:932-959 (first output of a buffer: atan2)
// reading each input byte from HBM exactly once and writing only int16 PCM.
//
// Mapping (CDNA4, wave64).  The kernel is VALU-bound before it is HBM-bound
// (integer VALU issues 16 lanes/clk/SIMD), so everything below is about
// instructions per input sample:
//  * One wave owns one (stream, run of consecutive buffers) and walks it tile by
//    tile; there is no workgroup barrier anywhere (workgroup == one wave).
//  * A tile is 4096 complex samples = 8 KiB.  Lane l owns the contiguous run of
//    64 samples [64l, 64l+64): eight 16-byte loads per lane (this pattern reads
//    HBM as fast as a fully coalesced one, tools/bw_probe.hip), re-issued into
//    the same registers as soon as pass 0 has consumed them.
//  * Pass 0 works on the raw bytes: two v_perm_b32 per input dword gather the
//    bytes one output component needs, then each output is two v_dot4_i32_i8
//    per component against tap vectors that already contain the -j^n rotation
//    signs and the I/Q swap; the "-127" is folded into the accumulator's start
//    value.  No unpack, no separate rotate.
//  * Passes 1-2 (and 3 when rotating) run on packed (I,Q) int16 pairs with
//    v_pk_* arithmetic (the sums stay below 2^15 there); later passes, the
//    droop FIR and the tails use 32-bit lanes.
//  * The conjugate product of the discriminator is two v_dot2_i32_i16; the
//    angle is atan2_q14 (dsp_device.h) with its node table in LDS.
//  * The only cross-lane traffic is each lane's last five outputs per pass,
//    handed to lane l+1 through a small wave-private LDS slot array; lane 63
//    leaves what lane 0 of the next tile needs in slot 0, already shifted by
//    one when that tile starts a buffer.  Arrays with fewer than 5 values per
//    lane go through a linear LDS ring instead.
//  * A wave that starts in the middle of a stream first runs ONE warm-up tile
//    (the last tile of the previous buffer, outputs discarded): every carried
//    quantity depends on fewer than 4096 earlier samples for P <= 6.
#pragma once

#include <string.h>

#include <type_traits>

#include "dsp_device.h"

namespace rtlfm {
namespace fused {

using state_t = rtlfm_stream_state;

constexpr int kLaneSamples = 64;
constexpr int kTileSamples = 64 * kLaneSamples;  // 4096
constexpr int kTileBytes = 2 * kTileSamples;     // 8192
constexpr int kPre = 16;                         // prefix entries of a linear ring
constexpr int kMaxP = 6;
constexpr int kMaxSegList = 128;

#ifdef RTLFM_FUSED_MARKS  // analysis builds only: section markers in the .s
#define RTLFM_MARK(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; MARK " name); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RTLFM_MARK(name) do { } while (0)
#endif
// Measurement builds only (tools/build_variant.sh fused_phases -DRTLFM_FUSED_PHASES, tools/box_phases.py): with clock stamps on,
// a wave leaves the shader cycles it spent per phase of the tile loop - 0 until the tile has arrived and is staged, 1 pass 0,
// 2 the other passes (+ FIR), 3 discriminator and the rest - in place of its start / end stamps.
#ifdef RTLFM_FUSED_PHASES
#define FUSED_PHASE(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph_t[k] += now_ - ph_tp; ph_tp = now_; } while (0)
#else
#define FUSED_PHASE(k) do { } while (0)
#endif

typedef short short2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ short2_t as_s2(uint32_t v) { return __builtin_bit_cast(short2_t, v); }
__device__ __forceinline__ uint32_t as_u32(short2_t v) { return __builtin_bit_cast(uint32_t, v); }

// fifth_order on a packed (I,Q) pair, 16-bit lanes (valid while 32*|x| < 2^15)
__device__ __forceinline__ uint32_t tap_pk16(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e,
                                             uint32_t f)
{
	short2_t s = (as_s2(a) + as_s2(f)) + (as_s2(b) + as_s2(e)) * (short)5 + (as_s2(c) + as_s2(d)) * (short)10;
	s = s >> 4;
	return as_u32(s);
}

// The same for inputs beyond the 16-bit form's range (32 |x| >= 2^15), up to |x| <= 8192 - every
// pass this kernel runs, with or without the raw DC block (|x| <= 256 * 2^p, p <= 5): the three pair
// sums a + f, b + e, c + d still fit 16-bit lanes (one v_pk_add_i16 each), and
// sum = (a + f) + 5 (b + e) + 10 (c + d) is one v_dot2_i32_i16 per component with the taps (5, 10) and
// the sign-extended a + f as the accumulator: 12 instructions instead of 26 (12 sign extensions, two
// multiply-add chains).  ">> 4" in 32 bits, the store keeps the low 16 bits as the reference's int16 does.
__device__ __forceinline__ uint32_t tap_i32(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e,
                                            uint32_t f)
{
	const uint32_t s1 = as_u32(as_s2(a) + as_s2(f)), s5 = as_u32(as_s2(b) + as_s2(e)), s10 = as_u32(as_s2(c) + as_s2(d));
	const short2_t taps = {(short)5, (short)10};
	const uint32_t pi = __builtin_amdgcn_perm(s10, s5, 0x05040100u);  // (s5.I, s10.I)
	const uint32_t pq = __builtin_amdgcn_perm(s10, s5, 0x07060302u);  // (s5.Q, s10.Q)
	const int yi = __builtin_amdgcn_sdot2(as_s2(pi), taps, (int)(int16_t)(s1 & 0xffffu), false) >> 4;
	const int yq = __builtin_amdgcn_sdot2(as_s2(pq), taps, (int)s1 >> 16, false) >> 4;
	return __builtin_amdgcn_perm((uint32_t)yq, (uint32_t)yi, 0x05040100u);
}

// Pass 0 constants.  S_j = raw dword j (I_2j, Q_2j, I_2j+1, Q_2j+1) XOR 0x7f7f7f7f:
// as int8 every byte is then exactly -(u - 127), the NEGATED converted sample
// of src/rtl_fm.c:1326-1328 (u = 0 -> +127, u = 255 -> -128), so the taps below
// are the negated filter taps and no bias term is left.
//   G_j = perm(S_j-1, S_j, selG), H_j = perm(S_j-1, S_j, selH)
//   I'[m] = dot4(G_m, ti[par][0]) + dot4(G_m-2, ti[par][1])      (par = m & 1)
//   Q'[m] = dot4(H_m, tq[par][0]) + dot4(H_m-2, tq[par][1])
// With rotation, sample n of a buffer is multiplied by (-j)^n:
//   I'(n) = +I,+Q,-I,-Q and Q'(n) = +Q,-I,-Q,+I for n%4 = 0..3,
// so G gathers (I_2j, Q_2j+1) and H gathers (Q_2j, I_2j+1) and the signs sit in
// the taps; without it G gathers the I bytes, H the Q bytes.
struct Pass0Taps {
	uint32_t selG, selH;
	int32_t ti[2][2], tq[2][2];
};

inline Pass0Taps make_taps(bool rotate)
{
	Pass0Taps t{};
	if (rotate) {
		t.selG = 0x07040300u; t.selH = 0x06050201u;
		// filter taps for even m: I (1,0,-10,-5) (5,10,0,-1); Q (1,0,-10,5) (5,-10,0,1);
		// odd m: the negatives.  Stored negated (see above).
		t.ti[0][0] = 0x050A00FF; t.ti[0][1] = 0x0100F6FB;
		t.ti[1][0] = (int32_t)0xFBF60001; t.ti[1][1] = (int32_t)0xFF000A05;
		t.tq[0][0] = (int32_t)0xFB0A00FF; t.tq[0][1] = (int32_t)0xFF000AFB;
		t.tq[1][0] = 0x05F60001; t.tq[1][1] = 0x0100F605;
	} else {
		t.selG = 0x06040200u; t.selH = 0x07050301u;
		for (int p = 0; p < 2; p++) {
			// filter taps (1,0,10,5) (5,10,0,1), negated
			t.ti[p][0] = (int32_t)0xFBF600FF; t.ti[p][1] = (int32_t)0xFF00F6FB;
			t.tq[p][0] = (int32_t)0xFBF600FF; t.tq[p][1] = (int32_t)0xFF00F6FB;
		}
	}
	return t;
}

// Pass 0 as a Toeplitz product on the int8 matrix pipe (optional engine, see k_fused):
//   D[16 x 16] = A[16 x 64] * B[64 x 16]   (v_mfma_i32_16x16x64_i8, operand maps verified
//   with tools/mfma_probe.hip: A[row l&15][k 16(l>>4)+byte], B[k][col l&15], D[row 4(l>>4)+reg][col l&15])
// Column n of B is a 64-byte window of S (raw ^ 0x7f) starting at dword 8n-4 of a
// 128-dword segment; row 2o+c of A holds the taps of output o (0..7) of the window,
// component c (0 = I', 1 = Q'): the four dwords m-3..m of output m = 8n+o sit at window
// dwords o+1..o+4.  Same taps as the dot4 form (negated for S = -x, sign (-1)^o when
// rotating), so the 32-bit sums are identical integers.  The taps are stored times 8
// and every segment runs through two chained MFMAs (the second accumulates onto the
// first), which yields 16*sum: the reference's ">> 4" is then a byte select (one
// v_perm per output instead of v_perm + v_pk_ashrrev; the matrix pipe has the slack).
inline void make_mfma_taps(bool rotate, uint32_t out[256])
{
	static const int TI_rot[4][4] = {{0, 0, 0, -1}, {5, 0, 0, 10}, {-10, 0, 0, -5}, {1, 0, 0, 0}};
	static const int TQ_rot[4][4] = {{0, 0, 1, 0}, {0, 5, -10, 0}, {0, -10, 5, 0}, {0, 1, 0, 0}};
	static const int TI_raw[4][4] = {{0, 0, 1, 0}, {5, 0, 10, 0}, {10, 0, 5, 0}, {1, 0, 0, 0}};
	static const int TQ_raw[4][4] = {{0, 0, 0, 1}, {0, 5, 0, 10}, {0, 10, 0, 5}, {0, 1, 0, 0}};
	signed char A[16][64];
	memset(A, 0, sizeof(A));
	for (int o = 0; o < 8; o++) {
		const int g = (rotate && (o & 1)) ? -1 : 1;
		for (int i = 0; i < 4; i++)
			for (int by = 0; by < 4; by++) {
				const int k = 4 * (o + 1 + i) + by;
				A[2 * o][k] = (signed char)(-8 * g * (rotate ? TI_rot : TI_raw)[i][by]);
				A[2 * o + 1][k] = (signed char)(-8 * g * (rotate ? TQ_rot : TQ_raw)[i][by]);
			}
	}
	for (int l = 0; l < 64; l++)
		for (int w = 0; w < 4; w++) {
			uint32_t v = 0;
			for (int by = 0; by < 4; by++) v |= (uint32_t)(unsigned char)A[l & 15][16 * (l >> 4) + 4 * w + by] << (8 * by);
			out[4 * l + w] = v;
		}
}

// 16 bytes of the input stream, read once: a non-temporal load (global_load_dwordx4 ... nt).
// The 4 GiB stream otherwise allocates in L2 / the Infinity Cache on its way through and
// pushes the kernel's own PCM stores out to HBM in the middle of the read stream; with nt
// loads the same read+write skeleton runs 9 % faster (tools/bw_probe.hip).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_stream16(const uint8_t *p)
{
	const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
	return make_uint4(v.x, v.y, v.z, v.w);
}

// With four or five passes the PCM (3-6 % of the bytes) leaves with the sc1 bit: written through
// and dropped from the XCD's L2 instead of sitting there dirty until the read stream pushes it out
// (interleaved A/B: -1.8 % at 4 passes, -1.4 % at 5, nothing at 6; nt stores +7 %; sc0 nothing;
// where the PCM is a larger share - 2 passes, boxcar /6 - write-through costs 6-16 %).
// Inline assembly, so the compiler's vmcnt bookkeeping does not see these stores: they are always
// issued BEFORE the next tile's loads, and an older operation only makes an in-order vmcnt wait
// wait for more, never for less.
#ifndef RTLFM_PCM_STORE_POLICY
#define RTLFM_PCM_STORE_POLICY 1  // 1 = sc1 (inline assembly), 0 = plain compiler-tracked stores, 2 = nt, 3 = sc1 without the memory clobber
#endif
__device__ __forceinline__ void store_out8(void *p, uint32_t a, uint32_t b)
{
	typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
	const u32x2_t v = {a, b};
	if (RTLFM_PCM_STORE_POLICY == 1) asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
	else if (RTLFM_PCM_STORE_POLICY == 3) asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v));
	else if (RTLFM_PCM_STORE_POLICY == 2) __builtin_nontemporal_store(v, reinterpret_cast<u32x2_t *>(p));
	else *reinterpret_cast<u32x2_t *>(p) = v;
}
__device__ __forceinline__ void store_out4(void *p, uint32_t a)
{
	if (RTLFM_PCM_STORE_POLICY == 1) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(a) : "memory");
	else if (RTLFM_PCM_STORE_POLICY == 3) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(a));
	else if (RTLFM_PCM_STORE_POLICY == 2) __builtin_nontemporal_store(a, reinterpret_cast<uint32_t *>(p));
	else *reinterpret_cast<uint32_t *>(p) = a;
}

// dot4 / dot2 with a zero accumulator in the VOP3 form (inline constant 0):
// hipcc otherwise emits v_mov 0 + the accumulate-in-place VOP2 form.
__device__ __forceinline__ int dot4_first(uint32_t a, int32_t taps)
{
	// (round 5: with its own wait states.  gfx950 does NOT interlock a VALU read of a v_dot4_i32_i8 result: read at once or one
	// wait state later it is wrong every time, three later it is right - tools/dot_hazard_probe.hip -, and the compiler only
	// keeps that distance for dots it has emitted itself, not for a line of assembly.  Until now the consumers of this one
	// were three or more slots away by the scheduler's grace.  The builtin instead (v_mov 0 + v_dot4c) moves the v_dot4
	// engine's P = 1 kernels' tile loads: tools/check_prefetch.py.  v_dot2_i32_i16: see dot2_pair.)
	int r;
	asm("v_dot4_i32_i8 %0, %1, %2, 0\n\ts_nop 2" : "=v"(r) : "v"(a), "s"(taps));
	return r;
}
// The conjugate product of a discriminator: r0 = dot2(a, b0), r1 = dot2(a, b1), zero accumulators, with their own wait states.
// (Round 6.  LLVM's hazard recogniser keeps ANY VALU read of a dot result three wait states behind the dot and a write of
// another opcode two - v_dot2 included - but only for dots it emitted itself.  tools/dot_hazard_probe.hip never caught
// v_dot2_i32_i16 -> v_add_u32 wrong, so round 5 left the bare one-instruction wrapper in; tools/check_dot_hazard.py then
// found its result read by v_cvt_f64_i32 one and two wait states later, and overwritten one later, in 37 shipped kernels:
// every boxcar front end and the 1-, 2-, 5- and 6-pass fifth_order ones.  No test ever failed on it - and LAB.md I.17's one
// unexplained wrong sample came out of one of those kernels.  Whatever the hardware does most of the time, the documented
// distance is now kept: the second dot and s_nop 2 put four and three wait states behind the two results, and the lint
// runs in the CPU suite, tests/test_isa_lint.py.)
__device__ __forceinline__ void dot2_pair(uint32_t a, uint32_t b0, uint32_t b1, int &r0, int &r1)
{
	asm("v_dot2_i32_i16 %0, %2, %3, 0\n\tv_dot2_i32_i16 %1, %2, %4, 0\n\ts_nop 2" : "=&v"(r0), "=&v"(r1) : "v"(a), "v"(b0), "v"(b1));
}

struct Params {
	const uint8_t *iq;
	size_t stream_stride;
	uint32_t block_len;
	int nblocks, nstreams;
	int16_t *out;
	size_t out_stride;
	const state_t *sin;
	state_t *sout;
	const int32_t *lut;
	int variant, rotate;
	int mode, output_scale;  // RTLFM_MODE_FM, or AM / USB / LSB (run-time discriminator kernels only)
	// emit mode: stop after the FIR and store the decimated IQ (packed int16 pairs in time order)
	// instead of PCM: -M raw, and the input of the staged kernels that finish 7..10 passes
	uint32_t *emit_iq;
	size_t emit_iq_stride;  // dwords between streams
	const uint8_t *dummy_tile;  // 8 KiB, what the reload reads after a segment's last tile
	// A segment is a run of tiles of one stream (it may begin and end inside a buffer); wave w takes
	// segment w / nstreams of stream w % nstreams.  Uniform: segment j = tiles [j tiles_per_seg, ...);
	// nlist > 0: segment j = [seg_start[j], seg_start[j + 1]) - long segments first, short ones last
	// (plan_segments).
	int segs, tiles_per_seg, nlist;
	int seg_start[kMaxSegList + 1];
	const uint32_t *mfma_taps;  // [64 lanes][4] A operand of the pass-0 MFMA (make_mfma_taps)
	const int2 *rdc_avg;        // RDC kernels: [stream][nblocks] (avgI, avgQ) of dc_block_raw_filter, from k_rdc_sums / k_rdc_smooth
	// the power squelch / -L without an emit mode (round 5): [stream][nblocks] (sum of squares, sum) of every buffer's
	// decimated, FIR-compensated elements modulo 2^32, what rms() sums (src/rtl_fm.c:1093-1098); zeroed by the host, added
	// to with two atomics per tile; k_squelch_apply (staged_kernels.h) makes the decisions.  nullptr: nothing is added
	uint32_t *sq_sums;
	int debug;  // timing experiments only (RTLFM_FUSED_DEBUG): 2 = clock stamps (printed per launch; +16 = 18:
	            // only on timing_read, i.e. for the last launch of an uninterrupted run), 4 = reload one (cached) tile
	unsigned long long *stamps;  // [waves][4] when debug & 2
	Pass0Taps taps;
};

// tiles [t0, t1) of segment `seg` (wave-uniform: scalar loads from the kernel arguments)
template <typename ParamsT>
__device__ __forceinline__ void segment_bounds(const ParamsT &p, int seg, int total_tiles, int &t0, int &t1)
{
	if (p.nlist > 0) {
		t0 = p.seg_start[seg];
		t1 = p.seg_start[seg + 1];
	} else {
		t0 = seg * p.tiles_per_seg;
		t1 = t0 + p.tiles_per_seg;
	}
	if (t1 > total_tiles) t1 = total_tiles;
}

// ---- LDS layout of one wave (dword offsets) -----------------------------------
template <int P, bool FIR9, bool MFMA0 = false>
struct Lds {
	static constexpr int cz = 64 >> P;  // values per lane of the decimated array Z = Y[P-1]
	// 28 small persistent dwords first, so that the MFMA engine's chunk 0 lands on a
	// 128-byte boundary.  c_*: the predecessor values lane 0 sees in the next tile.
	static constexpr int xh = 0;        // [8] archived x' of the previous buffer
	static constexpr int c_y0 = 8;      // [5]
	static constexpr int c_y1 = 13;     // [5]
	static constexpr int c_y2 = 18;     // [5]
	static constexpr int c_zd = 23;     // [1]
	static constexpr int c_raw = 24;    // [3]
	// MFMA engine: the XORed tile as 16-byte chunks -1..512 (chunk -1 = last chunk of the
	// previous tile, chunk 512 = slack), reused in place for pass 0's output
	static constexpr int rawbuf = 28;
	static constexpr bool fz_slots = cz >= 9;                     // FIR history by hand-off, else ring
	static constexpr int tr_w = (FIR9 && fz_slots) ? 9 : 5;
	// Transient lane-to-lane hand-off slots [65][tr_w], shared by every hand-off of a tile.
	// With the MFMA engine they live inside the chunk area, which is idle between pass 0's
	// read-back and the next tile's staging (one wave's LDS queue is in order).
	static constexpr int tr = MFMA0 ? 32 + 64 : 32;
	// The body of a linear ring (ring_exchange: 64 * C values of one tile, C <= 4) is transient as well and
	// shares one region behind the hand-off slots; only a ring's 16-entry prefix - the previous tile's tail -
	// lives from tile to tile.  (With the bodies persistent the 6-pass kernel needed 10.5 KB per wave: 15
	// waves per CU instead of 16.)
	static constexpr int ring_body = ((tr + 65 * tr_w + 3) & ~3) + kPre;  // kPre entries in front of it: the prefix's transient copy
	static constexpr int fz_c = (FIR9 && !fz_slots) ? cz : 0;     // values per lane in the FIR input ring
	static constexpr int ring_c = (P >= 5 ? 4 : 0) > fz_c ? (P >= 5 ? 4 : 0) : fz_c;  // the widest ring of this instantiation
	static constexpr int after = MFMA0 ? 32 + 4 * 513 : ring_body + 64 * ring_c;
	static_assert(!MFMA0 || ring_body + 64 * ring_c <= 32 + 4 * 513, "the ring body must fit the chunk area");
	static constexpr int atan = (after + 1) & ~1;                 // 17 doubles
	static constexpr int c_fz = atan + 34;                        // [9]
	static constexpr int y3 = c_fz + ((FIR9 && fz_slots) ? 9 : 0);  // prefix of the Y3 ring, c=4 (P >= 5)
	static constexpr int y4 = y3 + (P >= 5 ? kPre : 0);           // prefix of the Y4 ring, c=2 (P >= 6)
	static constexpr int fz = y4 + (P >= 6 ? kPre : 0);           // prefix of the FIR input ring (!fz_slots)
	static constexpr int fz_size = (FIR9 && !fz_slots) ? kPre : 0;
	static constexpr int total = fz + fz_size;
};

// Lane l publishes `mine` for lane l+1 in the transient slots and receives lane
// l-1's; lane 0 receives what lane 63 left in `cr` during the previous tile
// (leave_carry, stored after the reads).  Wave-private, no barrier: one wave's
// LDS queue is in order.
template <int W>
__device__ __forceinline__ void hand_off(uint32_t *slots, const uint32_t *cr, const uint32_t (&mine)[W],
                                         uint32_t (&prev)[W], int lane)
{
#pragma unroll
	for (int k = 0; k < W; k++) slots[(lane + 1) * W + k] = mine[k];
	__builtin_amdgcn_wave_barrier();
	const uint32_t *rp = lane ? slots + lane * W : cr;
#pragma unroll
	for (int k = 0; k < W; k++) prev[k] = rp[k];
	__builtin_amdgcn_wave_barrier();
}
// `last_lane`: the lane that holds the tile's last samples - 63, or (partial tiles) the last active one
template <int W>
__device__ __forceinline__ void leave_carry(uint32_t *cr, const uint32_t (&carry)[W], int lane, int last_lane = 63)
{
	if (lane == last_lane) {
#pragma unroll
		for (int k = 0; k < W; k++) cr[k] = carry[k];
	}
	__builtin_amdgcn_wave_barrier();
}

// The five predecessors e[-5..-1] of a lane for one fifth_order pass over an
// array with C >= 6 values per lane.  `drop_newest_next`: the next tile starts
// a buffer, whose history is one sample older (the archive of
// src/rtl_fm.c:800-805 never holds the newest input).
template <int C>
__device__ __forceinline__ void fifth_history(uint32_t *slots, uint32_t *cr, const uint32_t (&Y)[C], uint32_t (&h)[5],
                                              int lane, bool drop_newest_next, int last_lane = 63)
{
	uint32_t mine[5] = {Y[C - 5], Y[C - 4], Y[C - 3], Y[C - 2], Y[C - 1]};
	hand_off<5>(slots, cr, mine, h, lane);
	if (drop_newest_next) {
		uint32_t carry[5] = {Y[C - 6], Y[C - 5], Y[C - 4], Y[C - 3], Y[C - 2]};
		leave_carry<5>(cr, carry, lane, last_lane);
	} else {
		leave_carry<5>(cr, mine, lane, last_lane);
	}
}

// The same exchanges without LDS (RTLFM_DPP_EXCHANGE): lane l takes lane l - 1's values through DPP wave_shr:1, lane 0
// takes the carry - what the last lane held one tile earlier, kept in SGPRs (v_readlane at the end of the exchange).
// Three VALU instructions per dword (shift, select for lane 0, read-lane for the next tile) instead of an LDS store, a
// load, the carry's store and two waits on the LDS queue in the middle of the tile's dependent chain.
template <int W>
__device__ __forceinline__ void hand_off_dpp(const uint32_t (&mine)[W], uint32_t (&prev)[W], const uint32_t (&carry)[W], int lane)
{
#pragma unroll
	for (int k = 0; k < W; k++) {
		const uint32_t sh = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine[k], 0x138, 0xf, 0xf, false);  // wave_shr:1
		prev[k] = lane == 0 ? carry[k] : sh;
	}
}
template <int C>
__device__ __forceinline__ void fifth_history_dpp(const uint32_t (&Y)[C], uint32_t (&h)[5], uint32_t (&carry)[5], int lane,
                                                  bool drop_newest_next, int last_lane)
{
	const uint32_t mine[5] = {Y[C - 5], Y[C - 4], Y[C - 3], Y[C - 2], Y[C - 1]};
	hand_off_dpp<5>(mine, h, carry, lane);
	// what the next tile's lane 0 sees: the last lane's newest five, or (the next tile starts a buffer) the five before
	// the newest - the archive of src/rtl_fm.c:800-805 never holds the newest input
	uint32_t t[6];
#pragma unroll
	for (int k = 0; k < 6; k++) t[k] = (uint32_t)__builtin_amdgcn_readlane((int)Y[C - 6 + k], last_lane);
#pragma unroll
	for (int k = 0; k < 5; k++) carry[k] = drop_newest_next ? t[k] : t[k + 1];
}

// linear ring: this tile's C values per lane go into the transient body, every lane reads the H entries
// before its first - from the body, or, for positions before the tile, from the ring's prefix (the last
// kPre entries of the previous tile) - and then the lanes that hold the tile's last kPre entries leave
// them in the prefix, straight from their registers.  Two barriers.
// QUIRK: at a buffer start the entries that lie before the buffer are read one
// position further back (see fifth_history); with fewer than five values per
// lane that reaches lanes 1 and 2 as well.  SHIFT (whole-tile kernels): the kept prefix is then copied in front of
// the body one slot LATER (its newest entry, the x[N-1] the reference never saves, falls off), so that every lane still
// reads at one base address + constants; without it (partial tiles, whose next prefix needs that entry) the
// position is chosen per read - a compare and a select per history entry.
// nlanes < 64 (a partial tile): the ring's new prefix is the last kPre entries of (old prefix ++ the nlanes * C
// valid entries), copied inside LDS instead of left by the last lanes' registers.
template <int C, int H, bool QUIRK, bool SHIFT = false>
__device__ __forceinline__ void ring_exchange(uint32_t *lds, int prefix, int body, const uint32_t (&mine)[C], uint32_t (&hist)[H],
                                              int lane, bool buffer_start, int nlanes = 64)
{
	static_assert(kPre % C == 0 && H + 1 <= kPre, "the prefix holds whole lanes and reaches back far enough");
	// the kept prefix goes in front of the transient body, so that every lane reads its history at
	// one base address + constants (a per-position choice between two places cost a register per
	// position and pushed the 6-pass kernels into spilling)
	uint32_t *ring = lds + body - kPre;
	uint32_t carried = 0;
	if (lane < kPre) carried = lds[prefix + lane];
#pragma unroll
	for (int k = 0; k < C; k++) ring[kPre + lane * C + k] = mine[k];
	const int sh = (QUIRK && SHIFT && buffer_start) ? 1 : 0;  // wave-uniform
	if (lane < kPre - sh) ring[lane + sh] = carried;
	__builtin_amdgcn_wave_barrier();
#pragma unroll
	for (int k = 0; k < H; k++) {
		int pos = lane * C - H + k;
		if (QUIRK && !SHIFT && buffer_start && pos < 0) pos -= 1;
		hist[k] = ring[kPre + pos];
	}
	__builtin_amdgcn_wave_barrier();
	if (nlanes == 64) {
		if (lane >= 64 - kPre / C) {
#pragma unroll
			for (int k = 0; k < C; k++) lds[prefix + (lane - (64 - kPre / C)) * C + k] = mine[k];
		}
	} else {
		uint32_t keep = 0;
		if (lane < kPre) keep = ring[nlanes * C + lane];
		__builtin_amdgcn_wave_barrier();
		if (lane < kPre) lds[prefix + lane] = keep;
	}
	__builtin_amdgcn_wave_barrier();
}

// unpack one raw dword (samples 2j, 2j+1 of a buffer) into two rotated packed
// (I,Q) int16 pairs: reference convert (:1326-1328) + rotate16_neg90 (:424-434)
// dcI / dcQ: what dc_block_raw_filter (:1043-1065) subtracts from this buffer's I / Q samples before the rotation
__device__ __forceinline__ void unpack_rot(uint32_t raw, int j_odd, int rotate, uint32_t &s0, uint32_t &s1, int dcI = 0, int dcQ = 0)
{
	int a0 = (int)(raw & 0xff) - 127 - dcI, b0 = (int)((raw >> 8) & 0xff) - 127 - dcQ;
	int a1 = (int)((raw >> 16) & 0xff) - 127 - dcI, b1 = (int)(raw >> 24) - 127 - dcQ;
	if (!rotate) {
		s0 = pack_iq(a0, b0); s1 = pack_iq(a1, b1);
	} else if (!j_odd) {
		s0 = pack_iq(a0, b0);    // n%4 == 0
		s1 = pack_iq(b1, -a1);   // n%4 == 1: (b, -a)
	} else {
		s0 = pack_iq(-a0, -b0);  // n%4 == 2
		s1 = pack_iq(-b1, a1);   // n%4 == 3: (-b, a)
	}
}

// fifth_order over a lane's CIN inputs with its five predecessors e[-5..-1]
template <int CIN, bool PK16>
__device__ __forceinline__ void fifth_lane(const uint32_t (&x)[CIN], const uint32_t (&h)[5],
                                           uint32_t (&y)[CIN / 2])
{
	auto e = [&](int idx) -> uint32_t { return idx < 0 ? h[5 + idx] : x[idx]; };
#pragma unroll
	for (int m = 0; m < CIN / 2; m++) {
		if constexpr (PK16)
			y[m] = tap_pk16(e(2 * m - 5), e(2 * m - 4), e(2 * m - 3), e(2 * m - 2), e(2 * m - 1), e(2 * m));
		else
			y[m] = tap_i32(e(2 * m - 5), e(2 * m - 4), e(2 * m - 3), e(2 * m - 2), e(2 * m - 1), e(2 * m));
	}
}

#ifndef RTLFM_PRIO_BY_PROGRESS
#define RTLFM_PRIO_BY_PROGRESS 1
#endif
// The further a wave is through its segment, the lower its priority.  Left alone, the SIMD's arbiter prefers its oldest
// wave: four waves that start together finish one after the other (4.2a: 563 / 631 / 735 / 839 us for the four slots),
// the launch drains for a third of its length, and a second-round wave - the youngest on its SIMD - crawls beside three
// old ones.  With the priority following the progress (s_setprio 3 / 2 / 1 / 0 by quarter of the segment; a scalar
// instruction per tile) the waves of a SIMD advance together and end together: -2.5 % at four buffers per launch, -5 %
// at one (0.2068 -> 0.1960 ms), same box, old / new library alternating.  Two levels by half gain half of it, the
// inverse order loses 2.5-6 %.
struct ProgressPrio {
	int th1, th2, th3, base;  // loop-invariant scalars: no state is carried from tile to tile
	// total tiles of the segment; skip_top: priority 3 is left to the audio tail's kernel (three levels by third)
	__device__ __forceinline__ ProgressPrio(int total, int skip_top)
	{
		const int levels = skip_top ? 3 : 4;
		const int step = __builtin_amdgcn_readfirstlane((total + levels - 1) / levels);
		th1 = step; th2 = 2 * step; th3 = skip_top ? 0x7fffffff : 3 * step;
		base = skip_top ? 1 : 0;
	}
	// at the top of every tile (done = tiles behind the wave): three scalar compares and an s_setprio
	__device__ __forceinline__ void at(int done) const
	{
#if RTLFM_PRIO_BY_PROGRESS
		const int q4 = __builtin_amdgcn_readfirstlane(base + (done >= th1 ? 1 : 0) + (done >= th2 ? 1 : 0) + (done >= th3 ? 1 : 0));
		if (q4 <= 0) __builtin_amdgcn_s_setprio(3);
		else if (q4 == 1) __builtin_amdgcn_s_setprio(2);
		else if (q4 == 2) __builtin_amdgcn_s_setprio(1);
		else __builtin_amdgcn_s_setprio(0);
#endif
	}
};

struct AtanNodesLds {
	const double *t;
	__device__ __forceinline__ double operator()(int i) const { return t[i]; }
};


#ifndef RTLFM_DPP_EXCHANGE
#define RTLFM_DPP_EXCHANGE 1  // lane-to-lane hand-offs of the first passes and the discriminator through DPP + SGPR carries instead of LDS
#endif
#ifndef RTLFM_FUSED_WAVES_PER_SIMD
#define RTLFM_FUSED_WAVES_PER_SIMD 4
#endif
// (Five waves per SIMD for the 4- and 5-pass polar_discriminant kernels - 96 VGPRs by recomputing the
// lane-derived LDS addresses per tile - were built and measured with the wave count matched to the 4864
// slots, tools/ab_engines.py --waves: no faster than four, and the recomputation costs 2-3 % with four.
// DESIGN.md section 4.2a.)
#ifndef RTLFM_PASS0_DEFAULT
#define RTLFM_PASS0_DEFAULT 1  // 0: always v_dot4 on the VALU, 1: int8 MFMA where it is faster (RTLFM_PASS0=valu|mfma overrides)
#endif
// Where the MFMA variant issues the next tile's loads: 0 = after the MFMA phase, 1 = after
// pass 0's read-back and the deferred PCM store, 2 = after pass 1 (P >= 2 only).
#ifndef RTLFM_MFMA_RELOAD_AT
#define RTLFM_MFMA_RELOAD_AT 1
#endif
#ifndef RTLFM_FUSED_EARLY_RELOAD
#define RTLFM_FUSED_EARLY_RELOAD 0
#endif
// STD: the discriminator is known to be polar_discriminant at compile time (the
// common case and the one the roofline is quoted on); otherwise p.variant picks
// fast / lut at run time.
// MFMA0: pass 0 on the int8 matrix pipe instead of v_dot4 (see make_mfma_taps).
// sum over the wave, valid in lane 63 (inclusive DPP scan)
__device__ __forceinline__ int wave_sum_to_last(int x)
{
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
	x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
	x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
	return x;
}

// RDC: dc_block_raw_filter (-E rdc, src/rtl_fm.c:1043-1065, 1330-1332) in front of the chain.  The
// filter subtracts one (avgI, avgQ) per buffer - the block mean smoothed over the blocks, which a
// pre-pass has to know before the first sample (k_rdc_sums / k_rdc_smooth: one more read of the
// input) - and everything up to pass 0's ">> 4" is linear: the six-tap sum of the rotated constant is
// C_I = -/+ 4 (aI - aQ), C_Q = -/+ 4 (aI + aQ) for even / odd outputs (32 aI, 32 aQ without the
// rotation), and -16 C is what the MFMA accumulators start from instead of zero.  No instruction is
// added to the tile; samples now span +-255, so pass 3 takes the 32-bit form as it does without rotation.
// PT: callback buffers that are not whole tiles (-W n: any 512 n bytes, src/rtl_fm.c:1869-1873).  A buffer is then
// ceil(block_len / 8192) tiles whose last one is PARTIAL - 256 k samples, lanes 0 .. 4k - 1 active, the others
// compute on stale data that nobody reads: the buffer-boundary rules (rotation phase, the fifth_order quirk,
// fm_demod's first sample) keep falling on tile starts.  What changes for such a tile: its rows behind the
// buffer's end are not loaded (load rows are chosen per row, the one the end cuts in two is read 512 bytes early
// and put right in LDS), "the tile's last lane" is lane 4k - 1 instead of 63 wherever a carry is left for the next
// tile or the state is archived, a ring's new prefix is copied inside LDS, and a wave that starts mid-stream
// warms up on as many tiles as hold 1024 samples (every carried quantity depends on fewer than
// 15 * 2^P - 5 <= 955 earlier samples).  MFMA engine only; the PT = false kernels are untouched.  Round 5: PT and RDC
// together (the averages are per BUFFER and a partial tile ends its buffer: a tile never holds two buffers' samples).
template <int P, bool FIR9, bool STD, bool MFMA0, bool RDC = false, bool PT = false>
__global__ void __launch_bounds__(64, (P == 1 ? 2 : P == 2 ? 3 : RTLFM_FUSED_WAVES_PER_SIMD)) k_fused(const Params p)
{
	static_assert(!RDC || MFMA0, "the raw DC block rides on the MFMA accumulators");
	static_assert(!PT || MFMA0, "partial tiles: MFMA engine");
	using L = Lds<P, FIR9, MFMA0>;
	constexpr int CZ = L::cz;
	__shared__ __attribute__((aligned(128))) uint32_t lds[L::total];
	const int lane = threadIdx.x;
	const int wave = blockIdx.x;
	const int seg = wave / p.nstreams;
	const int s = wave - seg * p.nstreams;
	if (seg >= p.segs) return;
	const int tpb = PT ? (int)((p.block_len + kTileBytes - 1) / kTileBytes) : (int)(p.block_len / kTileBytes);
	const int total_tiles = p.nblocks * tpb;
	// PT: bytes of a buffer's last tile (a multiple of 512), and what one buffer yields
	const int last_tile_bytes = PT ? (int)(p.block_len - (uint32_t)(tpb - 1) * kTileBytes) : kTileBytes;
	const int out_per_buffer = (int)((p.block_len / 2) >> P);
	int t0, t1;
	segment_bounds(p, seg, total_tiles, t0, t1);
	if (t0 >= t1) return;
	const bool from_state = (t0 == 0);
	// A segment that starts behind the run's first tile runs the tile before it as a warm-up (outputs
	// discarded) - the last tile of the previous buffer or an earlier tile of the same buffer alike:
	// bs / next_bs below follow from the tile's position, so the buffer-boundary rules apply where
	// they belong whatever the segmentation is.
	const int gt_first = t0;
	int gt_begin = from_state ? gt_first : gt_first - 1;  // one warm-up tile
	if (PT && !from_state) {
		// ... or as many as hold 1024 samples, where tiles are short
		int have = ((gt_begin % tpb) == tpb - 1 ? last_tile_bytes : kTileBytes) / 2;
		while (gt_begin > 0 && have < 1024) {
			gt_begin--;
			have += ((gt_begin % tpb) == tpb - 1 ? last_tile_bytes : kTileBytes) / 2;
		}
	}
	const int gt_end = t1;
	// PT: a warm-up that reaches back to the run's first tile may be shorter than 1024 samples - it starts from the
	// carried state, like the stream's first segment (whole tiles: a full warm-up tile needs none)
	const bool load_state = from_state || (PT && gt_begin == 0);
	const bool writes_state = (t1 == total_tiles);
	const state_t *sin = p.sin + s;
	state_t *sout = p.sout + s;
	const int rotate = p.rotate;

	// The carried state is double-buffered (read sin, write sout).  The wave that ends a
	// stream's run first copies the whole record, so that fields this kernel does not own
	// (boxcar, resampler, deemph, DC state) carry over without a separate copy kernel per run.
	if (writes_state) {
		const uint32_t *a = reinterpret_cast<const uint32_t *>(sin);
		uint32_t *b = reinterpret_cast<uint32_t *>(sout);
		static_assert(sizeof(state_t) % 4 == 0, "state record is copied dword-wise");
		for (int k = lane; k < (int)(sizeof(state_t) / 4); k += 64) b[k] = a[k];
	}
	unsigned long long st_clk = 0, st_rt = 0;
	if (p.debug & 2) { st_clk = __builtin_amdgcn_s_memtime(); st_rt = __builtin_amdgcn_s_memrealtime(); }
	// ---- carried history at the start of the segment ---------------------------
	for (int k = lane; k < L::total; k += 64) lds[k] = 0;
	__builtin_amdgcn_wave_barrier();
	if (lane < 17) reinterpret_cast<double *>(lds + L::atan)[lane] = k_atan_nodes[lane];
	if (load_state && lane == 0) {
		for (int j = 0; j < 6; j++) lds[L::xh + j] = pack_iq(sin->lp_i_hist[0][j], sin->lp_q_hist[0][j]);
		auto put_slots = [&](int off, int pass) {  // e[-5..-1] = hist[1..5]
			for (int k = 0; k < 5; k++) lds[off + k] = pack_iq(sin->lp_i_hist[pass][k + 1], sin->lp_q_hist[pass][k + 1]);
		};
		auto put_ring = [&](int off, int pass) {  // A[kPre-7+j] = hist[j]
			for (int j = 0; j < 6; j++) lds[off + kPre - 7 + j] = pack_iq(sin->lp_i_hist[pass][j], sin->lp_q_hist[pass][j]);
		};
		if (P >= 2) put_slots(L::c_y0, 1);
		if (P >= 3) put_slots(L::c_y1, 2);
		if (P >= 4) put_slots(L::c_y2, 3);
		if (P >= 5) put_ring(L::y3, 4);
		if (P >= 6) put_ring(L::y4, 5);
		if (FIR9) {
			for (int j = 0; j < 9; j++) {
				uint32_t v = pack_iq(sin->droop_i_hist[j], sin->droop_q_hist[j]);
				if (L::fz_slots) lds[L::c_fz + j] = v; else lds[L::fz + kPre - 9 + j] = v;
			}
		}
		lds[L::c_zd] = pack_iq((int16_t)sin->pre_r, (int16_t)sin->pre_j);
	}
	__builtin_amdgcn_wave_barrier();
	const AtanNodesLds nodes{reinterpret_cast<const double *>(lds + L::atan)};
	// RTLFM_DPP_EXCHANGE: the carries of the lane-to-lane hand-offs live in SGPRs from here on
	// (from three passes on: the one- and two-pass kernels carry 32 / 16 outputs per lane through the tile, and with the
	// carries in SGPRs the allocator of four of their RDC variants copied prefetched registers in front of the back-edge -
	// a wait on the tile loads, tools/check_prefetch.py; their exchanges are a small share of the tile anyway)
	constexpr bool DPPX = RTLFM_DPP_EXCHANGE && P >= 3;
	uint32_t cy0[5] = {0, 0, 0, 0, 0}, cy1[5] = {0, 0, 0, 0, 0}, cy2[5] = {0, 0, 0, 0, 0}, czd[1] = {0};
	if constexpr (DPPX) {
#pragma unroll
		for (int k = 0; k < 5; k++) {
			if (P >= 2) cy0[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds[L::c_y0 + k]);
			if (P >= 3) cy1[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds[L::c_y1 + k]);
			if (P >= 4) cy2[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds[L::c_y2 + k]);
		}
		czd[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds[L::c_zd]);
	}

	const uint8_t *stream_base = p.iq + (size_t)s * p.stream_stride;
	const int out_per_tile = 64 * CZ;
	int16_t *out_base = p.out + (size_t)s * p.out_stride;

	// dot4 engine: lane l holds the contiguous dwords 32l..32l+31 (cur[k] = dwords 32l+4k..);
	// MFMA engine: cur[k] = 16-byte chunk 64k + l of the tile (fully coalesced 1 KiB per load)
	uint4 cur[8];
	auto load_from = [&](const uint8_t *tb) {
#pragma unroll
		for (int k = 0; k < 8; k++) {
			// non-temporal only where one instruction covers whole lines: the lane-contiguous
			// pattern touches each 128-byte line eight times and would refetch it (2x slower)
			if constexpr (MFMA0) cur[k] = load_stream16(tb + k * 1024 + lane * 16);
			else cur[k] = *reinterpret_cast<const uint4 *>(tb + lane * 128 + k * 16);
		}
	};
	// PT: tile `tile` = tile tib of buffer b; rows behind the buffer's end come from the dummy tile, the row the
	// end cuts in two is read 512 bytes early (nothing outside the run is touched; staging puts it right)
	auto load_tile_pt = [&](int tile) {
		const bool real = tile < gt_end;
		const int b = tile / tpb, tib = tile - b * tpb;
		const uint8_t *tb = real ? stream_base + (size_t)b * p.block_len + (size_t)tib * kTileBytes : p.dummy_tile;
		const int valid = real && tib == tpb - 1 ? last_tile_bytes : kTileBytes;  // wave-uniform
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const uint8_t *row = tb + k * 1024;
			if (k * 1024 >= valid) row = p.dummy_tile + k * 1024;
			else if (k * 1024 + 512 == valid) {
				if (tile > 0 || k > 0) row -= 512;
				else row = lane < 32 ? row : p.dummy_tile;  // a run that begins with a 512-byte buffer: nothing before it
			}
			cur[k] = load_stream16(row + lane * 16);
		}
	};
	auto load_tile = [&](int tile) {
		if constexpr (PT) load_tile_pt(tile);
		else load_from(stream_base + (size_t)tile * kTileBytes);
	};
	load_tile(gt_begin);
	// The reload is unconditional: were it skipped for the last tile of a segment, the
	// loop-carried registers would be a merge of "kept" and "loaded" values and the register
	// allocator would copy loaded registers right after the loads, i.e. wait for them
	// (tools/check_prefetch.py).  After the last tile it reads a dummy tile that every wave
	// shares and that therefore stays in the caches (re-reading the last tile itself would go
	// to HBM again, the stream being loaded non-temporally: 1.5 % of the traffic).
	auto reload = [&](int gt, bool more) {
		if constexpr (PT) {
			load_tile_pt(gt + 1);  // tile gt_end reads the dummy tile
		} else {
			const uint8_t *next = stream_base + (size_t)((p.debug & 4) ? gt_begin : gt + 1) * kTileBytes;
			load_from(more || (p.debug & 4) ? next : p.dummy_tile);
		}
	};
	typedef int v4i_t __attribute__((ext_vector_type(4)));
	v4i_t mfma_a = {0, 0, 0, 0};
	if constexpr (MFMA0) {
		const uint4 a = reinterpret_cast<const uint4 *>(p.mfma_taps)[lane];
		mfma_a = v4i_t{(int)a.x, (int)a.y, (int)a.z, (int)a.w};
	}
	int dcI = 0, dcQ = 0;             // RDC: this buffer's (avgI, avgQ)
	v4i_t dc_acc = {0, 0, 0, 0};      // -16 x the six-tap sum of the rotated constant: rows (I', Q') of an even and an odd output

	// The PCM of tile t is stored just before tile t+2's loads are issued, never
	// after them: loads and stores share the in-order vmcnt counter on gfx9-family
	// parts, so a store younger than the loads would put the whole HBM write latency
	// into the wait that precedes pass 0 of the next tile.
	int16_t held[CZ];
	int16_t *held_dst = nullptr;
	auto flush_held = [&]() {
		if (held_dst) {
			int16_t *dst = held_dst;
			if constexpr (CZ >= 8) {
				// 1-3 passes: the PCM is 12-50 % of the bytes, and written through it costs (P = 2: +16 %)
				// (debug bit 256: non-temporal - set by the host for outputs that do not fit the Infinity Cache, LAB.md I.12)
				uint4 *d4 = reinterpret_cast<uint4 *>(dst);
				const bool nt = (p.debug & 256) != 0;
#pragma unroll
				for (int k = 0; k < CZ / 8; k++) {
					u32x4_t w;
					w.x = pack_iq(held[8 * k], held[8 * k + 1]); w.y = pack_iq(held[8 * k + 2], held[8 * k + 3]);
					w.z = pack_iq(held[8 * k + 4], held[8 * k + 5]); w.w = pack_iq(held[8 * k + 6], held[8 * k + 7]);
					// (s_nop: a store of more than 64 bits reads its data over two cycles, and a VALU write to those registers must
					// stay one wait state behind it - the compiler keeps that distance for stores it knows, not for this line)
					if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 0" : : "v"(d4 + k), "v"(w) : "memory");
					else d4[k] = make_uint4(w.x, w.y, w.z, w.w);
				}
			} else if constexpr (CZ == 4) {
				store_out8(dst, pack_iq(held[0], held[1]), pack_iq(held[2], held[3]));
			} else if constexpr (CZ == 2) {
				store_out4(dst, pack_iq(held[0], held[1]));
			} else {
				dst[0] = held[0];
			}
		}
	};

	ProgressPrio prio(gt_end - gt_begin, __builtin_amdgcn_readfirstlane(p.debug & 128));  // 128: an audio tail follows
#ifdef RTLFM_FUSED_PHASES
	unsigned long long ph_t[4] = {0, 0, 0, 0}, ph_tp = __builtin_amdgcn_s_memtime();
#endif
	for (int gt = gt_begin; gt < gt_end; gt++) {
		const bool more = gt + 1 < gt_end;
		const int tib = gt % tpb;
		const bool bs = tib == 0;                       // this tile starts a buffer
		const bool next_bs = ((gt + 1) % tpb) == 0;     // the next one does
		const bool q0 = bs && lane == 0;
		const bool emit = gt >= gt_first;
		const bool last = gt + 1 == gt_end;
		// PT: a buffer's last tile holds tile_bytes / 128 active lanes (64 samples = 128 bytes each)
		const int tile_bytes = PT && next_bs ? last_tile_bytes : kTileBytes;
		const int nlanes = PT ? tile_bytes >> 7 : 64;
		const int last_lane = nlanes - 1;
		const bool archive = last && writes_state && lane == last_lane;

		RTLFM_MARK("tile_begin");
		FUSED_PHASE(3);  // (what follows the discriminator: the hold of the PCM, the loop's own bookkeeping)
		prio.at(gt - gt_begin);
		if constexpr (RDC) {
			if (bs || gt == gt_begin) {
				// a scalar load (the index is wave-uniform; readfirstlane says so to the compiler): a vector load
				// would share the in-order vmcnt with the tile prefetch and force it to land (tools/check_prefetch.py)
				const int idx = __builtin_amdgcn_readfirstlane(s * p.nblocks + gt / tpb);
				const int2 a = p.rdc_avg[idx];
				dcI = __builtin_amdgcn_readfirstlane(a.x); dcQ = __builtin_amdgcn_readfirstlane(a.y);
				if (rotate) dc_acc = v4i_t{64 * (dcI - dcQ), 64 * (dcI + dcQ), -64 * (dcI - dcQ), -64 * (dcI + dcQ)};
				else dc_acc = v4i_t{-512 * dcI, -512 * dcQ, -512 * dcI, -512 * dcQ};
			}
		}
		// ---------------------------------------------------------------- pass 0 ----
		uint32_t Y0[32];
		if constexpr (MFMA0) {
			// The first / last raw dwords of the tile are needed by the two rare paths below;
			// in this layout they sit in lane 0's cur[0] and lane 63's cur[7].
			uint32_t fix0 = 0, fix1 = 0, fix2 = 0;
			if (!PT && bs) {
				uint32_t e[11];
#pragma unroll
				for (int k = 0; k < 5; k++) e[k] = lds[L::xh + 1 + k];
				unpack_rot(cur[0].x, 0, rotate, e[5], e[6], dcI, dcQ);
				unpack_rot(cur[0].y, 1, rotate, e[7], e[8], dcI, dcQ);
				unpack_rot(cur[0].z, 0, rotate, e[9], e[10], dcI, dcQ);
				fix0 = tap_pk16(e[0], e[1], e[2], e[3], e[4], e[5]);
				fix1 = tap_pk16(e[2], e[3], e[4], e[5], e[6], e[7]);
				fix2 = tap_pk16(e[4], e[5], e[6], e[7], e[8], e[9]);
			}
			__builtin_amdgcn_wave_barrier();
			if (!PT && next_bs) {
				uint32_t a0, a1, a2, a3, a4, a5, a6, a7;
				unpack_rot(cur[7].x, 0, rotate, a0, a1, dcI, dcQ);
				unpack_rot(cur[7].y, 1, rotate, a2, a3, dcI, dcQ);
				unpack_rot(cur[7].z, 0, rotate, a4, a5, dcI, dcQ);
				unpack_rot(cur[7].w, 1, rotate, a6, a7, dcI, dcQ);
				if (lane == 63) {
					lds[L::xh + 0] = a1; lds[L::xh + 1] = a2; lds[L::xh + 2] = a3;
					lds[L::xh + 3] = a4; lds[L::xh + 4] = a5; lds[L::xh + 5] = a6;
					if (archive) {
						uint32_t v[6] = {a1, a2, a3, a4, a5, a6};
						for (int j = 0; j < 6; j++) { iq16 w = unpack_iq(v[j]); sout->lp_i_hist[0][j] = w.i; sout->lp_q_hist[0][j] = w.q; }
					}
				}
			}
			__builtin_amdgcn_wave_barrier();
			// stage S = raw ^ 0x7f7f7f7f as chunks 0..511 (chunk -1 is still the previous tile's 511)
			uint4 *chunks = reinterpret_cast<uint4 *>(lds + L::rawbuf + 4);
			uint4 last;
#pragma unroll
			for (int k = 0; k < 8; k++) {
				uint4 v = cur[k];
				v.x ^= 0x7f7f7f7fu; v.y ^= 0x7f7f7f7fu; v.z ^= 0x7f7f7f7fu; v.w ^= 0x7f7f7f7fu;
				chunks[64 * k + lane] = v;
				if (k == 7) last = v;
			}
			__builtin_amdgcn_wave_barrier();
			FUSED_PHASE(0);  // the tile has arrived and is staged
			if constexpr (PT) {
				const int nv = tile_bytes >> 4;  // valid 16-byte chunks: a multiple of 32
				if (nv < 512) {
					// the row the buffer's end cuts in two was read 512 bytes early (load_tile_pt): its second half is
					// the row's first half; what follows is not part of the buffer (left as sample 0)
					const int kh = tile_bytes >> 10;
					if ((tile_bytes & 1023) && (gt > 0 || kh > 0)) {
						uint4 v = make_uint4(0, 0, 0, 0);
						if (lane < 32) v = chunks[64 * kh + lane + 32];
						__builtin_amdgcn_wave_barrier();
						chunks[64 * kh + lane] = v;
						__builtin_amdgcn_wave_barrier();
					}
				}
				if (bs) {
					// the first three outputs of a buffer, from the archived x' of the buffer before and this tile's
					// first dwords - read back from the staged tile (a row may have been loaded 512 bytes early)
					const uint4 first = chunks[0];
					uint32_t e[11];
#pragma unroll
					for (int k = 0; k < 5; k++) e[k] = lds[L::xh + 1 + k];
					unpack_rot(first.x ^ 0x7f7f7f7fu, 0, rotate, e[5], e[6], dcI, dcQ);
					unpack_rot(first.y ^ 0x7f7f7f7fu, 1, rotate, e[7], e[8], dcI, dcQ);
					unpack_rot(first.z ^ 0x7f7f7f7fu, 0, rotate, e[9], e[10], dcI, dcQ);
					fix0 = tap_pk16(e[0], e[1], e[2], e[3], e[4], e[5]);
					fix1 = tap_pk16(e[2], e[3], e[4], e[5], e[6], e[7]);
					fix2 = tap_pk16(e[4], e[5], e[6], e[7], e[8], e[9]);
					__builtin_amdgcn_wave_barrier();  // xh is read before the block below rewrites it
				}
				last = chunks[nv - 1];  // the tile's last valid chunk (all lanes read it: a broadcast)
				if (next_bs) {
					// archive x'[N-7..N-2] of the buffer that ends here: the last eight samples' raw bytes again
					uint32_t a0, a1, a2, a3, a4, a5, a6, a7;
					unpack_rot(last.x ^ 0x7f7f7f7fu, 0, rotate, a0, a1, dcI, dcQ);
					unpack_rot(last.y ^ 0x7f7f7f7fu, 1, rotate, a2, a3, dcI, dcQ);
					unpack_rot(last.z ^ 0x7f7f7f7fu, 0, rotate, a4, a5, dcI, dcQ);
					unpack_rot(last.w ^ 0x7f7f7f7fu, 1, rotate, a6, a7, dcI, dcQ);
					if (lane == last_lane) {
						lds[L::xh + 0] = a1; lds[L::xh + 1] = a2; lds[L::xh + 2] = a3;
						lds[L::xh + 3] = a4; lds[L::xh + 4] = a5; lds[L::xh + 5] = a6;
						if (archive) {
							uint32_t v[6] = {a1, a2, a3, a4, a5, a6};
							for (int j = 0; j < 6; j++) { iq16 w = unpack_iq(v[j]); sout->lp_i_hist[0][j] = w.i; sout->lp_q_hist[0][j] = w.q; }
						}
					}
				}
				__builtin_amdgcn_wave_barrier();
			}
			// 16 segments of 128 outputs: lane (n = l&15, q = l>>4) feeds window n's bytes
			// 16q..16q+15 = chunk 32s + 2n + q - 1 and receives outputs 128s + 8n + 2q + {0,1}
			const int ln = lane;
			const int n = ln & 15, q = ln >> 4;
			const uint4 *rd = chunks + (2 * n + q - 1);
			// Outputs go back in place, but the eight 16-byte slots of each lane's later
			// 128-byte run (run L = m >> 5) are stored XOR-swizzled, slot k at k ^ ((L >> 1) & 7):
			// the read-back below (lane stride 128 B) is then bank-conflict free instead of
			// 8-way conflicted.  For this lane L = 4*sgm + (n >> 2), k = (2n + (q >> 1)) & 7, so
			// the swizzle is ((2*sgm) & 7) | (n >> 3): four distinct store offsets, by sgm & 3.
			// The chunk area starts 128-byte aligned, so the slot index is bits 4..6 of the byte
			// address and the four offsets are wr0 ^ 32a: one v_xor each, no address registers held.
			const int yk = ((2 * n + (q >> 1)) & 7) ^ (n >> 3);
			char *const lds_b = reinterpret_cast<char *>(lds);
			const uint32_t wr0 = 4u * (uint32_t)(L::rawbuf + 4 + 32 * (n >> 2) + ((2 * q) & 3) + 4 * yk);
			auto wr_sw = [&](int a) { return reinterpret_cast<uint2 *>(lds_b + (wr0 ^ (32u * (uint32_t)a))); };
			// Two batches of eight segments.  Segment s+1's lane (0,0) needs chunk 32s+31, which
			// segment s's outputs overwrite in place, so each batch reads all of its operands
			// (plus the first one of the next batch) before it writes anything; inside a batch
			// the nine LDS reads, eight MFMAs and eight stores are independent of each other.
			uint4 bop[9];
#pragma unroll
			for (int k = 0; k < 9; k++) bop[k] = rd[32 * k];
			__builtin_amdgcn_wave_barrier();
			if (lane == (PT ? 0 : 63)) chunks[-1] = last;  // for the next tile; segment 0 has read the old one (PT: every lane holds it)
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int half = 0; half < 2; half++) {
				uint2 yv[8];
#pragma unroll
				for (int g = 0; g < 8; g += 4) {
					v4i_t acc[4];
#pragma unroll
					for (int k = 0; k < 4; k++) {
						const uint4 o = bop[g + k];
						acc[k] = __builtin_amdgcn_mfma_i32_16x16x64_i8(mfma_a, v4i_t{(int)o.x, (int)o.y, (int)o.z, (int)o.w},
						                                                RDC ? dc_acc : v4i_t{0, 0, 0, 0}, 0, 0, 0);
					}
#pragma unroll
					for (int k = 0; k < 4; k++) {
						const uint4 o = bop[g + k];
						const v4i_t d = __builtin_amdgcn_mfma_i32_16x16x64_i8(mfma_a, v4i_t{(int)o.x, (int)o.y, (int)o.z, (int)o.w},
						                                                       acc[k], 0, 0, 0);
						// d = 16*sum: bits 8..23 are (sum >> 4) as int16
						yv[g + k] = make_uint2(__builtin_amdgcn_perm((uint32_t)d.y, (uint32_t)d.x, 0x06050201u),
						                       __builtin_amdgcn_perm((uint32_t)d.w, (uint32_t)d.z, 0x06050201u));
					}
				}
				if (half == 0) {
					bop[0] = bop[8];
#pragma unroll
					for (int k = 1; k < 8; k++) bop[k] = rd[32 * (8 + k)];  // chunks >= 287: not touched by batch 0's stores
					__builtin_amdgcn_wave_barrier();
				}
#pragma unroll
				for (int k = 0; k < 8; k++) wr_sw(k & 3)[64 * (8 * half + k)] = yv[k];
				__builtin_amdgcn_wave_barrier();
			}
			// the raw registers are free since the staging; the next tile's loads go out once
			// the MFMA phase no longer needs the register file for operands and accumulators
			if (RTLFM_MFMA_RELOAD_AT == 0) reload(gt, more);
			// back to the lane-contiguous form the later passes use
			const uint32_t yl0 = 4u * (uint32_t)(L::rawbuf + 4) + 128u * ln + 16u * ((ln >> 1) & 7);
#pragma unroll
			for (int k = 0; k < 8; k++) {
				const uint4 v = *reinterpret_cast<const uint4 *>(lds_b + (yl0 ^ (16u * k)));
				Y0[4 * k] = v.x; Y0[4 * k + 1] = v.y; Y0[4 * k + 2] = v.z; Y0[4 * k + 3] = v.w;
			}
			__builtin_amdgcn_wave_barrier();
			if (bs && lane == 0) { Y0[0] = fix0; Y0[1] = fix1; Y0[2] = fix2; }
		} else
		{
			// S = raw ^ 0x80808080 is the only form of the tile the math needs, so the
			// raw registers are free again after 35 XORs: the next tile's loads go out
			// here and have the whole tile's arithmetic to land.
			uint32_t sx[35];  // S[-3..31]
			{
				uint32_t mine[3] = {cur[7].y, cur[7].z, cur[7].w}, prev[3];
				hand_off<3>(lds + L::tr, lds + L::c_raw, mine, prev, lane);
				leave_carry<3>(lds + L::c_raw, mine, lane);
				sx[0] = prev[0] ^ 0x7f7f7f7fu; sx[1] = prev[1] ^ 0x7f7f7f7fu; sx[2] = prev[2] ^ 0x7f7f7f7fu;
			}
#pragma unroll
			for (int k = 0; k < 8; k++) {
				sx[3 + 4 * k] = cur[k].x ^ 0x7f7f7f7fu; sx[4 + 4 * k] = cur[k].y ^ 0x7f7f7f7fu;
				sx[5 + 4 * k] = cur[k].z ^ 0x7f7f7f7fu; sx[6 + 4 * k] = cur[k].w ^ 0x7f7f7f7fu;
			}
			if (RTLFM_FUSED_EARLY_RELOAD) reload(gt, more);
			RTLFM_MARK("xor_done");
			// Outputs are produced eight at a time so that only a window of the
			// gathered registers is live (keeps the kernel at 4 waves per SIMD).
			{
				uint32_t g0 = __builtin_amdgcn_perm(sx[0], sx[1], p.taps.selG), g1 = __builtin_amdgcn_perm(sx[1], sx[2], p.taps.selG);
				uint32_t h0 = __builtin_amdgcn_perm(sx[0], sx[1], p.taps.selH), h1 = __builtin_amdgcn_perm(sx[1], sx[2], p.taps.selH);
#pragma unroll
				for (int m = 0; m < 32; m++) {
					// G[m+2] in the notation above = perm(S[m-1], S[m]) = perm(sx[m+2], sx[m+3])
					const uint32_t g2 = __builtin_amdgcn_perm(sx[m + 2], sx[m + 3], p.taps.selG);
					const uint32_t h2 = __builtin_amdgcn_perm(sx[m + 2], sx[m + 3], p.taps.selH);
					const int par = m & 1;
					int ai = dot4_first(g2, p.taps.ti[par][0]);
					ai = __builtin_amdgcn_sdot4((int)g0, p.taps.ti[par][1], ai, false);
					int aq = dot4_first(h2, p.taps.tq[par][0]);
					aq = __builtin_amdgcn_sdot4((int)h0, p.taps.tq[par][1], aq, false);
					uint32_t pk = __builtin_amdgcn_perm((uint32_t)aq, (uint32_t)ai, 0x05040100u);
					Y0[m] = as_u32(as_s2(pk) >> 4);
					g0 = g1; g1 = g2; h0 = h1; h1 = h2;
				}
			}
			RTLFM_MARK("pass0_done");
			if (bs) {
				// first three outputs of a buffer: history is the archived x' of the
				// previous buffer (one sample older, previous buffer's rotation phase)
				uint32_t e[11];
#pragma unroll
				for (int k = 0; k < 5; k++) e[k] = lds[L::xh + 1 + k];
				unpack_rot(sx[3] ^ 0x7f7f7f7fu, 0, rotate, e[5], e[6]);
				unpack_rot(sx[4] ^ 0x7f7f7f7fu, 1, rotate, e[7], e[8]);
				unpack_rot(sx[5] ^ 0x7f7f7f7fu, 0, rotate, e[9], e[10]);
				uint32_t f0 = tap_pk16(e[0], e[1], e[2], e[3], e[4], e[5]);
				uint32_t f1 = tap_pk16(e[2], e[3], e[4], e[5], e[6], e[7]);
				uint32_t f2 = tap_pk16(e[4], e[5], e[6], e[7], e[8], e[9]);
				if (lane == 0) { Y0[0] = f0; Y0[1] = f1; Y0[2] = f2; }
			}
			__builtin_amdgcn_wave_barrier();
			if (next_bs) {
				// archive x'[N-7..N-2] of the buffer that ends here (lane 63's samples 57..62)
				uint32_t a0, a1, a2, a3, a4, a5, a6, a7;
				unpack_rot(sx[31] ^ 0x7f7f7f7fu, 0, rotate, a0, a1);
				unpack_rot(sx[32] ^ 0x7f7f7f7fu, 1, rotate, a2, a3);
				unpack_rot(sx[33] ^ 0x7f7f7f7fu, 0, rotate, a4, a5);
				unpack_rot(sx[34] ^ 0x7f7f7f7fu, 1, rotate, a6, a7);
				if (lane == 63) {
					lds[L::xh + 0] = a1; lds[L::xh + 1] = a2; lds[L::xh + 2] = a3;
					lds[L::xh + 3] = a4; lds[L::xh + 4] = a5; lds[L::xh + 5] = a6;
					if (archive) {
						uint32_t v[6] = {a1, a2, a3, a4, a5, a6};
						for (int j = 0; j < 6; j++) { iq16 w = unpack_iq(v[j]); sout->lp_i_hist[0][j] = w.i; sout->lp_q_hist[0][j] = w.q; }
					}
				}
			}
			__builtin_amdgcn_wave_barrier();
		}

		flush_held();
		held_dst = nullptr;
		if (!MFMA0 ? !RTLFM_FUSED_EARLY_RELOAD : RTLFM_MFMA_RELOAD_AT == 1) reload(gt, more);
		// hist[pass] = Y[c-7..c-2] of lane 63 (registers) / of the ring's prefix
		auto archive_regs = [&](auto &Y, auto cc, int pass) {
			constexpr int c = decltype(cc)::value;
			if (archive) {
#pragma unroll
				for (int j = 0; j < 6; j++) {
					iq16 w = unpack_iq(Y[c - 7 + j]);
					sout->lp_i_hist[pass][j] = w.i; sout->lp_q_hist[pass][j] = w.q;
				}
			}
		};
		auto archive_ring = [&](int off, int pass) {
			if (archive) {
				for (int j = 0; j < 6; j++) {
					iq16 w = unpack_iq(lds[off + kPre - 7 + j]);
					sout->lp_i_hist[pass][j] = w.i; sout->lp_q_hist[pass][j] = w.q;
				}
			}
		};

		const int lz = lane;
		RTLFM_MARK("pass0_special_done");
		FUSED_PHASE(1);
		// ------------------------------------------------------------ passes 1.. ----
		uint32_t Z[CZ];  // output of the last pass
		if constexpr (P == 1) {
#pragma unroll
			for (int k = 0; k < 32; k++) Z[k] = Y0[k];
		} else {
			uint32_t h5[5];
			uint32_t Y1[16];
			if constexpr (DPPX) fifth_history_dpp<32>(Y0, h5, cy0, lz, next_bs, last_lane);
			else fifth_history<32>(lds + L::tr, lds + L::c_y0, Y0, h5, lz, next_bs, last_lane);
			fifth_lane<32, true>(Y0, h5, Y1);
			archive_regs(Y0, std::integral_constant<int, 32>(), 1);
			if (MFMA0 && RTLFM_MFMA_RELOAD_AT == 2) reload(gt, more);
			if constexpr (P == 2) {
#pragma unroll
				for (int k = 0; k < 16; k++) Z[k] = Y1[k];
			} else {
				uint32_t Y2[8];
				if constexpr (DPPX) fifth_history_dpp<16>(Y1, h5, cy1, lz, next_bs, last_lane);
				else fifth_history<16>(lds + L::tr, lds + L::c_y1, Y1, h5, lz, next_bs, last_lane);
				fifth_lane<16, true>(Y1, h5, Y2);
				archive_regs(Y1, std::integral_constant<int, 16>(), 2);
				if constexpr (P == 3) {
#pragma unroll
					for (int k = 0; k < 8; k++) Z[k] = Y2[k];
				} else {
					uint32_t Y3[4];
					if constexpr (DPPX) fifth_history_dpp<8>(Y2, h5, cy2, lz, next_bs, last_lane);
					else fifth_history<8>(lds + L::tr, lds + L::c_y2, Y2, h5, lz, next_bs, last_lane);
					// with rotation |x| <= 1023 here, so the 16-bit form cannot overflow;
					// without it an all-255 input reaches exactly 2^15
					if (rotate && !RDC) fifth_lane<8, true>(Y2, h5, Y3);
					else fifth_lane<8, false>(Y2, h5, Y3);
					archive_regs(Y2, std::integral_constant<int, 8>(), 3);
					if constexpr (P == 4) {
#pragma unroll
						for (int k = 0; k < 4; k++) Z[k] = Y3[k];
					} else {
						uint32_t Y4[2];
						ring_exchange<4, 5, true, !PT>(lds, L::y3, L::ring_body, Y3, h5, lz, bs, nlanes);
						fifth_lane<4, false>(Y3, h5, Y4);
						archive_ring(L::y3, 4);
						if constexpr (P == 5) {
							Z[0] = Y4[0]; Z[1] = Y4[1];
						} else {
							uint32_t Y5[1];
							ring_exchange<2, 5, true, !PT>(lds, L::y4, L::ring_body, Y4, h5, lz, bs, nlanes);
							fifth_lane<2, false>(Y4, h5, Y5);
							archive_ring(L::y4, 5);
							Z[0] = Y5[0];
						}
					}
				}
			}
		}

		RTLFM_MARK("passes_done");
		FUSED_PHASE(2);
		// --------------------------------------------------------- generic_fir ----
		uint32_t V[CZ];  // what fm_demod sees
		if constexpr (FIR9) {
			uint32_t h9[9];
			if constexpr (L::fz_slots) {
				uint32_t mine[9];
#pragma unroll
				for (int k = 0; k < 9; k++) mine[k] = Z[CZ - 9 + k];
				hand_off<9>(lds + L::tr, lds + L::c_fz, mine, h9, lz);
				leave_carry<9>(lds + L::c_fz, mine, lane, last_lane);
				if (archive) {
#pragma unroll
					for (int j = 0; j < 9; j++) { iq16 w = unpack_iq(mine[j]); sout->droop_i_hist[j] = w.i; sout->droop_q_hist[j] = w.q; }
				}
			} else {
				ring_exchange<CZ, 9, false>(lds, L::fz, L::ring_body, Z, h9, lz, false, nlanes);
				if (archive) {
					for (int j = 0; j < 9; j++) { iq16 w = unpack_iq(lds[L::fz + kPre - 9 + j]); sout->droop_i_hist[j] = w.i; sout->droop_q_hist[j] = w.q; }
				}
			}
			// output n = taps over the nine samples before n (src/rtl_fm.c:815-821)
			auto e = [&](int idx) -> uint32_t { return idx < 0 ? h9[9 + idx] : Z[idx]; };
#pragma unroll
			for (int n = 0; n < CZ; n++) {
				int hi[9], hq[9];
#pragma unroll
				for (int k = 0; k < 9; k++) { iq16 w = unpack_iq(e(n - 9 + k)); hi[k] = w.i; hq[k] = w.q; }
				int yi = fir9_tap(hi, k_cic9[P]);
				int yq = fir9_tap(hq, k_cic9[P]);
				V[n] = pack_iq((int16_t)yi, (int16_t)yq);
			}
		} else {
#pragma unroll
			for (int k = 0; k < CZ; k++) V[k] = Z[k];
		}

		if constexpr (!STD) {
			// emit mode: what full_demod() would hand to mode_demod (the decimated, FIR-compensated
			// IQ), tile-major and lane-contiguous = time order; used for -M raw and as the input of
			// the staged kernels that finish 7..10 passes or apply the squelch
			if (p.emit_iq) {
				if (emit && (!PT || lane < nlanes)) {
					// stored right away (no deferral as for the PCM: keeping CZ more registers alive
					// through the next tile would cost every kernel of this family its occupancy)
					const size_t at = PT ? (size_t)(gt / tpb) * out_per_buffer + (size_t)tib * out_per_tile + (size_t)lane * CZ
					                     : ((size_t)gt * 64 + lane) * CZ;
					uint32_t *dst = p.emit_iq + (size_t)s * p.emit_iq_stride + at;
					if constexpr (CZ >= 4) {
#pragma unroll
						for (int k = 0; k < CZ / 4; k++)
							reinterpret_cast<uint4 *>(dst)[k] = make_uint4(V[4 * k], V[4 * k + 1], V[4 * k + 2], V[4 * k + 3]);
					} else if constexpr (CZ == 2) {
						*reinterpret_cast<uint2 *>(dst) = make_uint2(V[0], V[1]);
					} else {
						*dst = V[0];
					}
				}
				continue;
			}
		}
		RTLFM_MARK("fir_done");
		if (p.sq_sums && emit) {  // wave-uniform
			// rms()'s two sums over this tile's decimated elements (a tile lies inside one buffer)
			const short2_t ones = {(short)1, (short)1};
			int sq_p = 0, sq_t = 0;
			if (!PT || lane < nlanes) {
#pragma unroll
				for (int k = 0; k < CZ; k++) {
					sq_p = __builtin_amdgcn_sdot2(as_s2(V[k]), as_s2(V[k]), sq_p, false);
					sq_t = __builtin_amdgcn_sdot2(as_s2(V[k]), ones, sq_t, false);
				}
			}
			sq_p = wave_sum_to_last(sq_p); sq_t = wave_sum_to_last(sq_t);
			if (lane == 63) {
				uint32_t *d = p.sq_sums + ((size_t)s * p.nblocks + (size_t)(gt / tpb)) * 2;
				atomicAdd(d, (uint32_t)sq_p); atomicAdd(d + 1, (uint32_t)sq_t);
			}
		}
		// ------------------------------------------------------------ fm_demod ----
		uint32_t pv;
		{
			uint32_t mine[1] = {V[CZ - 1]}, prev[1];
			if constexpr (DPPX) {
				hand_off_dpp<1>(mine, prev, czd, lz);
				czd[0] = (uint32_t)__builtin_amdgcn_readlane((int)mine[0], last_lane);
			} else {
				hand_off<1>(lds + L::tr, lds + L::c_zd, mine, prev, lz);
				leave_carry<1>(lds + L::c_zd, mine, lane, last_lane);
			}
			pv = prev[0];
			if (archive) {
				iq16 w = unpack_iq(V[CZ - 1]);
				if (STD || p.mode == RTLFM_MODE_FM) { sout->pre_r = w.i; sout->pre_j = w.q; }  // only fm_demod keeps them
			}
		}
		int16_t pcm[CZ];
#pragma unroll
		for (int n = 0; n < CZ; n++) {
			const uint32_t c = V[n];
			const uint32_t b = n == 0 ? pv : V[n > 0 ? n - 1 : 0];
			// multiply(a, conj(b)) (src/rtl_fm.c:836-840): |values| <= 3*8192 here, so
			// the int16 negation and the 32-bit dot products are exact
			const uint32_t bsw = __builtin_amdgcn_alignbit(b, b, 16);                      // (bq, bi)
			const uint32_t bx = as_u32(as_s2(bsw) * short2_t{(short)-1, (short)1});          // (-bq, bi)
			int cr, cj;
			dot2_pair(c, b, bx, cr, cj);
			int v;
			if (STD) {
				v = atan2_q14(cj, cr, nodes);
			} else {
				if (p.mode != RTLFM_MODE_FM) v = simple_demod(p.mode, c, p.output_scale);
				else if (p.variant == RTLFM_ATAN_FAST) v = fast_atan2_q14(cj, cr);
				else v = lut_atan2_q14_direct(cj, cr, nodes);
				if (n == 0 && bs && p.mode == RTLFM_MODE_FM) {
					// first output of a buffer is always polar_discriminant (:935-937)
					int vs = atan2_q14(cj, cr, nodes);
					if (lane == 0) v = vs;
				}
			}
			pcm[n] = (int16_t)v;
		}
		(void)q0;
		RTLFM_MARK("demod_done");
		FUSED_PHASE(3);
		if (emit && (!PT || lane < nlanes)) {
			held_dst = PT ? out_base + (size_t)(gt / tpb) * out_per_buffer + (size_t)tib * out_per_tile + lane * CZ
			              : out_base + (size_t)gt * out_per_tile + lane * CZ;
#pragma unroll
			for (int k = 0; k < CZ; k++) held[k] = pcm[k];
		}
	}
	flush_held();
	if constexpr (RDC) {
		// dc_block_raw_filter keeps its smoothed averages (src/rtl_fm.c:1062-1063): those of the run's last buffer
		if (writes_state && lane == 0) { sout->dc_avgI = dcI; sout->dc_avgQ = dcQ; }
	}
	if ((p.debug & 2) && lane == 0) {
		unsigned long long e_clk = __builtin_amdgcn_s_memtime(), e_rt = __builtin_amdgcn_s_memrealtime();
		if (p.debug & 32) {
			// where the wave ran, in place of its clock stamps: HW_ID (wave / SIMD / CU / SH / SE) and XCC_ID
			const unsigned long long hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID[31:0]
			const unsigned long long xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID[3:0]
			st_clk = hw | (xcc << 32);
			e_clk = st_clk;
		}
		p.stamps[(size_t)wave * 4 + 0] = st_clk; p.stamps[(size_t)wave * 4 + 1] = e_clk;
		p.stamps[(size_t)wave * 4 + 2] = st_rt; p.stamps[(size_t)wave * 4 + 3] = e_rt;
#ifdef RTLFM_FUSED_PHASES
		for (int k = 0; k < 4; k++) p.stamps[(size_t)wave * 4 + k] = ph_t[k];
#endif
	}
}

struct Workspace {
	unsigned long long *stamps = nullptr;
	int stamp_waves = 0;                          // capacity of stamps[]
	int stamp_last = 0;                           // waves of the last launch that stamped
	unsigned stamp_seq = 0;                       // stamped launches so far: they alternate between the two halves of stamps[]
	// the half the last (back = 0) or the one before the last (back = 1) stamped launch wrote
	const unsigned long long *stamp_half(int back) const { return stamps + (size_t)((stamp_seq + 1 + back) & 1) * stamp_waves * 4; }
	uint32_t *mfma_taps[2] = {nullptr, nullptr};  // [rotate]
	uint8_t *dummy_tile = nullptr;
	int pass0_engine = -1;                        // 0 = v_dot4 (VALU), 1 = int8 MFMA; -1 = the compiled default (see launch)
	bool want_stamps = false;                     // rtlfm_gpu_clock_probe(): every wave leaves its clock stamps
	// segmentation (rtlfm_gpu_set_option: fused_waves, fused_min_tiles, fused_tiles_per_seg)
	int target_waves = 8192;                      // enough waves to fill the 4096 wave slots of 256 CUs twice
	// ... and when an audio tail follows the front end (its kernels share the GPU with the next step's
	// front end): shorter segments, so that wave slots come free more often and the tail's waves get in.
	// Same-box sweep of 8192 .. 32768: c3 step 0.907-0.926 -> 0.872-0.876 ms, wbfm
	// 1.418 -> 1.309-1.315 ms at 20480; without a tail the launch time is flat from 8192 to 24576.
	int target_waves_tail = 20480;
	// ... for the fifth_order front ends 12288 since the tail is ONE kernel (round 4, same box, two alternations: c3 step
	// 0.879 / 0.876 / 0.873 / 0.884 ms at 20480 / 16384 / 12288 / 24576; the boxcar front end of -M wbfm still wants
	// 20480: 1.392 / 1.408 / 1.424 / 1.392).  With the DPP hand-offs and the priority that follows the progress (the end of
	// round 4) the optimum moved: c3 step 0.886 / 0.884 / 0.870 / 0.857 / 0.866 / 0.880 ms at 12288 / 16384 / 20480 / 24576 /
	// 32768 / 40960, two alternations on one box; wbfm still 20480 (1.40 against 1.42-1.46 either side).
	// 0 = as target_waves_tail (set together with it by fused_waves).
	int target_waves_tail_fifth = 24576;
	bool tail_follows = false;                    // set by the host per run (rtlfm_hip.hip: plan_tail)
	int min_tiles = 8;                            // a segment pays one warm-up tile: at most 1/8 on top
	bool plan_by_caller = false;                  // fused_waves / fused_min_tiles were set: the planner's own rules of thumb stand back
	int tiles_per_seg = 0;                        // > 0: exactly this (tests)
	int gss_x10 = 0;                              // guided segment lengths: remaining / (gss R) per round, x10 (0 = equal segments)
	int fused_store = -1;                         // k_fused's PCM stores with 1-3 passes: -1 by the launch's output size, 0 plain, 1 non-temporal
	int box_store = -1;                           // k_boxcar_scan's output stores: -1 by the launch's output size, 0 plain, 1 whole lines + nt
	int debug = 0;                                // fused_debug: 2 = clock stamps per launch (+16: only on timing_read), 4 = reload a cached tile, 32 = stamp HW_ID / XCC_ID instead of the shader clock
	void release()
	{
		if (stamps) hipFree(stamps);
		stamps = nullptr;
		for (auto &t : mfma_taps) { if (t) hipFree(t); t = nullptr; }
		if (dummy_tile) hipFree(dummy_tile);
		dummy_tile = nullptr;
	}
};

// How a run of `total_tiles` tiles per stream is cut into segments (one wave each).  Every segment
// but a stream's first re-reads and recomputes one warm-up tile, so segments are kept at
// min_tiles or more - unless the launch cannot fill the GPU's wave slots anyway: then the extra
// waves run where nothing else would, and shorter segments only shorten the launch (one callback
// buffer of one stream, the reference's own shape, gets a wave per tile).
//
// Segment lengths DEcrease along the stream ("guided" scheduling, option fused_gss): the SIMD's
// instruction arbiter favours its oldest wave, so four equal waves that start together finish one
// after the other (measured: 563 / 631 / 735 / 839 us for the four slots of a SIMD, tools/
// wave_placement.py) and with equal segments the launch ends in a long tail of half-empty SIMDs.
// The dispatcher hands out workgroups in index order as slots free up, and wave w is segment
// w / nstreams: the long segments of every stream go first, ever shorter ones fill the freed slots,
// and the tail is as long as the SHORTEST segment.  Round r gives each stream R segments (R = the
// stream's share of the wave slots) of remaining / (gss R) tiles.
constexpr int kWaveSlots = 256 * 4 * 4;  // CUs x SIMDs x resident waves of these kernels
struct SegPlan {
	int segs = 1, tiles_per_seg = 0, nlist = 0;
	int start[kMaxSegList + 1];
};
inline SegPlan plan_segments(const Workspace &ws, int nstreams, int total_tiles, bool fifth_order = false)
{
	SegPlan sp;
	sp.tiles_per_seg = total_tiles;
	if (ws.tiles_per_seg > 0) {
		sp.tiles_per_seg = ws.tiles_per_seg < total_tiles ? ws.tiles_per_seg : total_tiles;
		sp.segs = (total_tiles + sp.tiles_per_seg - 1) / sp.tiles_per_seg;
		return sp;
	}
	int min_tiles = ws.min_tiles > 0 ? ws.min_tiles : 1;
	// Where the streams alone fill every wave slot, a second wave per stream buys no parallelism and pays its
	// warm-up tile: north_star's live shape - 4096 streams x ONE 262144-B buffer per launch, 32 tiles per stream -
	// runs one wave per stream (0.2050 against 0.2086 ms, 0.2128 against 0.2158 on another box, interleaved:
	// tools/launch_gap.py, gpurun_out/r04d/x1_waves.txt); from four buffers per launch on, two waves per stream win
	// again (0.8283 against 0.8383 ms): the longer the waves, the more the SIMD's preference for its oldest wave
	// spreads their ends (4.2a).  (Until the end of round 4 this rule said "segments of sixteen tiles at least", which
	// at 32 tiles per stream still cut two - bench.py's also.ns4096x1.waves_per_stream said so all along.)
	if (nstreams >= kWaveSlots && !ws.tail_follows && !ws.plan_by_caller) min_tiles = total_tiles <= 48 ? total_tiles : 16;
	const int target = !ws.tail_follows ? ws.target_waves
	                   : (fifth_order && ws.target_waves_tail_fifth > 0 ? ws.target_waves_tail_fifth : ws.target_waves_tail);
	int segs = (target + nstreams - 1) / nstreams;
	int cap = total_tiles / min_tiles;
	if (cap < 1) cap = 1;
	const bool underfilled = (long long)nstreams * cap < kWaveSlots;
	if (underfilled) {
		cap = (kWaveSlots + nstreams - 1) / nstreams;
		if (cap > total_tiles) cap = total_tiles;
	}
	if (segs > cap) segs = cap;
	if (segs < 1) segs = 1;
	sp.tiles_per_seg = (total_tiles + segs - 1) / segs;
	sp.segs = (total_tiles + sp.tiles_per_seg - 1) / sp.tiles_per_seg;
	if (ws.gss_x10 <= 0 || underfilled || sp.segs < 2) return sp;
	// guided: R segments per stream and round
	const int R = (kWaveSlots + nstreams - 1) / nstreams;
	int n = 0, at = 0;
	sp.start[0] = 0;
	while (at < total_tiles) {
		int len = (int)((long long)(total_tiles - at) * 10 / ((long long)ws.gss_x10 * R));
		if (len < min_tiles) len = min_tiles;
		for (int k = 0; k < R && at < total_tiles; k++) {
			if (n == kMaxSegList) return sp;  // too many pieces for the argument block: stay uniform
			int l = len;
			if (total_tiles - at - l < min_tiles) l = total_tiles - at;  // no crumbs at the end
			at += l;
			sp.start[++n] = at;
		}
	}
	sp.nlist = n;
	sp.segs = n;
	return sp;
}

// the pass-0 engine a launch will use: the handle's choice, else the compiled default
inline int effective_engine(const Workspace &ws)
{
	return ws.pass0_engine >= 0 ? ws.pass0_engine : (RTLFM_PASS0_DEFAULT ? 1 : 0);
}

// What the front end in emit mode plus staged kernels covers beyond supported(): 7..10 passes,
// -M raw, and the squelch (rtlfm_hip.hip: run_fused_emit)
// buffers that are not whole 8 KiB tiles (-W n) take the partial-tile kernels: MFMA engine
inline bool needs_partial_tiles(const rtlfm_cfg &c) { return (c.block_len % kTileBytes) != 0; }

inline bool supported_emit(const rtlfm_cfg &c)
{
	if (c.downsample_passes < 1 || c.downsample_passes > RTLFM_MAX_PASSES) return false;
	return c.downsample_passes > kMaxP || c.mode == RTLFM_MODE_RAW || c.squelch_level != 0 || c.report_levels != 0;
}

// the power squelch / -L with rms()'s sums taken by the front end itself and k_squelch_apply behind it (as
// boxfused::supported_sq): up to six passes, at most 32768 decimated elements per buffer (beyond that rms() looks at
// every step-th element only, src/rtl_fm.c:1090-1092)
inline bool supported_sq(const rtlfm_cfg &c)
{
	if (c.mode != RTLFM_MODE_FM && c.mode != RTLFM_MODE_AM && c.mode != RTLFM_MODE_USB && c.mode != RTLFM_MODE_LSB)
		return false;
	if (!c.squelch_level && !c.report_levels) return false;
	if (c.downsample_passes < 1 || c.downsample_passes > kMaxP) return false;
	return (c.block_len >> c.downsample_passes) <= 32768;
}

inline int ensure_dummy_tile(Workspace &ws)
{
	if (ws.dummy_tile) return 0;
	if (hipMalloc(&ws.dummy_tile, kTileBytes) != hipSuccess) return -ENOMEM;
	// hipMemset of device memory returns before the fill has run (it is a kernel on the null stream), and the handle's
	// own stream is non-blocking: without the wait a first launch could read the tile before it is filled - the
	// boxcar front end's partial last tile (-W n) then summed garbage behind the run's end into (now_r, now_j).
	// Found by the extended sweep with four test processes sharing the GPU, where the fill was late often enough.
	if (hipMemset(ws.dummy_tile, 0x7f, kTileBytes) != hipSuccess) return -EIO;
	if (hipStreamSynchronize(nullptr) != hipSuccess) return -EIO;
	return 0;
}

inline bool supported(const rtlfm_cfg &c, int nblocks)
{
	if (c.mode != RTLFM_MODE_FM && c.mode != RTLFM_MODE_AM && c.mode != RTLFM_MODE_USB && c.mode != RTLFM_MODE_LSB)
		return false;
	if (c.downsample_passes < 1 || c.downsample_passes > kMaxP) return false;
	if (c.squelch_level || c.report_levels) return false;
	(void)nblocks;
	// -E rdc: the RDC instantiations; -W n: the PT ones (both MFMA pass 0 only; rtlfm_hip.hip checks the engine)
	return true;
}

template <int P, bool FIR9>
static int launch_one(const Params &p, int waves, hipStream_t q)
{
	if (p.block_len % kTileBytes) {
		if (!p.mfma_taps) return -ENOTSUP;
		if (p.rdc_avg) {
			if (p.variant == RTLFM_ATAN_STD) hipLaunchKernelGGL((k_fused<P, FIR9, true, true, true, true>), dim3(waves), dim3(64), 0, q, p);
			else hipLaunchKernelGGL((k_fused<P, FIR9, false, true, true, true>), dim3(waves), dim3(64), 0, q, p);
		} else if (p.variant == RTLFM_ATAN_STD) hipLaunchKernelGGL((k_fused<P, FIR9, true, true, false, true>), dim3(waves), dim3(64), 0, q, p);
		else hipLaunchKernelGGL((k_fused<P, FIR9, false, true, false, true>), dim3(waves), dim3(64), 0, q, p);
		return hipGetLastError() == hipSuccess ? 0 : -EIO;
	}
	if (p.rdc_avg) {
		if (!p.mfma_taps) return -ENOTSUP;
		if (p.variant == RTLFM_ATAN_STD) hipLaunchKernelGGL((k_fused<P, FIR9, true, true, true>), dim3(waves), dim3(64), 0, q, p);
		else hipLaunchKernelGGL((k_fused<P, FIR9, false, true, true>), dim3(waves), dim3(64), 0, q, p);
		return hipGetLastError() == hipSuccess ? 0 : -EIO;
	}
	if (p.mfma_taps) {
		if (p.variant == RTLFM_ATAN_STD) hipLaunchKernelGGL((k_fused<P, FIR9, true, true>), dim3(waves), dim3(64), 0, q, p);
		else hipLaunchKernelGGL((k_fused<P, FIR9, false, true>), dim3(waves), dim3(64), 0, q, p);
	} else {
		if (p.variant == RTLFM_ATAN_STD) hipLaunchKernelGGL((k_fused<P, FIR9, true, false>), dim3(waves), dim3(64), 0, q, p);
		else hipLaunchKernelGGL((k_fused<P, FIR9, false, false>), dim3(waves), dim3(64), 0, q, p);
	}
	return hipGetLastError() == hipSuccess ? 0 : -EIO;
}

// emit_iq != nullptr: store the decimated (and FIR-compensated) IQ instead of PCM; with 7..10
// passes configured, the first six run here (without the FIR)
inline int launch(Workspace &ws, const rtlfm_cfg &c, int nstreams, const uint8_t *d_iq, size_t stream_stride,
                  int nblocks, int16_t *d_out, size_t out_stride, const state_t *sin, state_t *sout,
                  const int32_t *lut, hipStream_t q, uint32_t *emit_iq = nullptr, size_t emit_iq_stride = 0,
                  const int2 *rdc_avg = nullptr, uint32_t *sq_sums = nullptr)
{
	if (!emit_iq && (((uintptr_t)d_out & 15) || (out_stride & 7))) return -EINVAL;
	if (emit_iq && sq_sums) return -EINVAL;
	Params p{};
	p.emit_iq = emit_iq; p.emit_iq_stride = emit_iq_stride;
	p.rdc_avg = rdc_avg;
	p.sq_sums = sq_sums;
	if (int r = ensure_dummy_tile(ws)) return r;
	p.dummy_tile = ws.dummy_tile;
	p.iq = d_iq; p.stream_stride = stream_stride; p.block_len = c.block_len;
	p.nblocks = nblocks; p.nstreams = nstreams;
	p.out = d_out; p.out_stride = out_stride;
	p.sin = sin; p.sout = sout; p.lut = lut;
	p.variant = c.custom_atan; p.rotate = c.offset_tuning ? 0 : 1;
	p.mode = c.mode; p.output_scale = c.output_scale;
	if (c.mode != RTLFM_MODE_FM) p.variant = RTLFM_ATAN_FAST;  // any value but STD: the run-time kernels carry the mode switch
	p.taps = make_taps(p.rotate != 0);
	// Pass-0 engine: forced by rtlfm_gpu_set_path(3|4) / the pass0_engine option, else the MFMA
	// form: its coalesced tile loads can be non-temporal (load_stream16), which removes the cost
	// of mixing the PCM stores into the read stream, and with that it is the faster engine at
	// every decimation depth on MI355X (tools/ab_engines.py, interleaved launches: 6 % at 4
	// passes, 11 % at 5).  The v_dot4 form needs no matrix pipe and no LDS staging.
	const int engine = effective_engine(ws);
	if (rdc_avg && engine != 1) return -ENOTSUP;
	if (engine == 1) {
		uint32_t *&t = ws.mfma_taps[p.rotate];
		if (!t) {
			uint32_t host[256];
			make_mfma_taps(p.rotate != 0, host);
			if (hipMalloc(&t, sizeof(host)) != hipSuccess) return -ENOMEM;
			if (hipMemcpy(t, host, sizeof(host), hipMemcpyHostToDevice) != hipSuccess) return -EIO;
		}
		p.mfma_taps = t;
	}
	p.debug = ws.debug | (ws.tail_follows ? 128 : 0);
	if (!emit_iq && c.downsample_passes <= 3) {
		// Three passes write 1 KiB of PCM per 8 KiB tile, 16 bytes per lane: whole lines per instruction.  Beyond what the
		// Infinity Cache keeps they leave non-temporal (1.041 -> 1.008 ms per 4 GiB).  With two passes or one a lane holds 32
		// or 64 bytes and an instruction writes every second or fourth 16-byte piece of a line: non-temporal that is 3 %
		// and 2.4 x SLOWER (profiles/r05_ab_fused_store.txt) - only the option forces it there (tests).
		const double out_bytes = (double)nstreams * nblocks * (double)(c.block_len / 2) / (double)(1 << c.downsample_passes) * 2.0;
		if (ws.fused_store >= 0 ? ws.fused_store != 0 : (c.downsample_passes == 3 && out_bytes > 192.0 * 1048576.0)) p.debug |= 256;
	}
	if (ws.want_stamps) p.debug |= 2;
	if (needs_partial_tiles(c) && engine != 1) return -ENOTSUP;
	const SegPlan sp = plan_segments(ws, nstreams, nblocks * (int)((c.block_len + kTileBytes - 1) / kTileBytes), true);
	p.segs = sp.segs; p.tiles_per_seg = sp.tiles_per_seg; p.nlist = sp.nlist;
	if (sp.nlist) memcpy(p.seg_start, sp.start, sizeof(int) * (size_t)(sp.nlist + 1));
	const int waves = nstreams * sp.segs;
	if (p.debug & 2) {
		if (ws.stamp_waves < waves) { if (ws.stamps) hipFree(ws.stamps); ws.stamps = nullptr; ws.stamp_waves = 0; if (hipMalloc(&ws.stamps, (size_t)waves * 32 * 2) != hipSuccess) return -ENOMEM; ws.stamp_waves = waves; }
		p.stamps = ws.stamps + (size_t)(ws.stamp_seq & 1) * ws.stamp_waves * 4;
		ws.stamp_seq++;
		ws.stamp_last = waves;
	}
	if (emit_iq) p.variant = RTLFM_ATAN_FAST;  // the emit path lives in the run-time-discriminator kernels
	if (emit_iq && c.downsample_passes > kMaxP) return launch_one<6, false>(p, waves, q);  // the rest is staged
	const bool fir = c.comp_fir_size == 9;
	switch (c.downsample_passes * 2 + (fir ? 1 : 0)) {
	case 2: return launch_one<1, false>(p, waves, q);
	case 3: return launch_one<1, true>(p, waves, q);
	case 4: return launch_one<2, false>(p, waves, q);
	case 5: return launch_one<2, true>(p, waves, q);
	case 6: return launch_one<3, false>(p, waves, q);
	case 7: return launch_one<3, true>(p, waves, q);
	case 8: return launch_one<4, false>(p, waves, q);
	case 9: return launch_one<4, true>(p, waves, q);
	case 10: return launch_one<5, false>(p, waves, q);
	case 11: return launch_one<5, true>(p, waves, q);
	case 12: return launch_one<6, false>(p, waves, q);
	case 13: return launch_one<6, true>(p, waves, q);
	default: return -ENOTSUP;
	}
}

}  // namespace fused
}  // namespace rtlfm
