// boxcar_kernel.h — rtl_fm's default decimator in one launch:
//   u8 IQ -> (-127, rotate16_neg90) -> low_pass boxcar sum of D samples -> fm_demod -> int16 PCM
// (src/rtl_fm.c:1326-1338, :461-481, :932-959), the path of `rtl_fm` without -F, of config 1
// and of the `-M wbfm` preset.  Same work decomposition as fused_kernel.h: one wave walks a
// run of one stream in 8 KiB tiles (4096 complex samples), a wave that starts mid-stream runs one
// warm-up tile, the next tile's loads are in flight while a tile is processed, no workgroup barriers.
// Callback buffers need not be whole tiles (-W n: any 512 n bytes, src/rtl_fm.c:1869-1873): low_pass carries
// its sum and count across buffers, and rotate16_neg90's phase restarts every buffer but buffers are
// multiples of four samples - so the run is ONE continuous sample stream here, whose last tile may be
// partial (its missing bytes read as 127 = sample 0: the sums do not move); a buffer boundary only
// matters to fm_demod, whose first output of every buffer is polar_discriminant whatever -A says.
//
// low_pass keeps a running sum (now_r, now_j) and a count prev_index across buffers, so with
// p0 samples already accumulated at the start of the run, output k covers run samples
// [k*D - p0, (k+1)*D - p0): boundaries fall anywhere and the number of outputs per tile and
// per buffer varies.  A boxcar is P(hi) - P(lo) of a prefix sum P, so the cost of a tile does
// not depend on D: see k_boxcar_scan below.
#pragma once
#include <hip/hip_runtime.h>
#include <cerrno>
#include <cstdint>
#include <cstdlib>

#include "dsp_device.h"
#include "fused_kernel.h"

namespace rtlfm {
namespace boxfused {

constexpr int kRdcMany = 18;  // buffers of 512 bytes: sixteen in a tile, and the two it may begin and end inside
constexpr int kMaxD = 2047;
constexpr int kTileBytes = 8192;
constexpr int kTileSamples = 4096;
constexpr int kTileDwords = 2048;

struct Params {
	const uint8_t *iq;
	size_t stream_stride;
	uint32_t block_len;
	int nblocks, nstreams;
	int16_t *out;
	size_t out_stride;
	int32_t *cnt;  // [nstreams] outputs of the run
	const state_t *sin;
	state_t *sout;
	int variant, rotate;
	int mode, output_scale;
	int D, q4096, r4096;  // 4096 = q4096 * D + r4096
	uint32_t D_magic;     // ceil(2^32 / D): n / D = mulhi(n, D_magic) for every n a tile can ask about (n D < 2^32)
	int out_cap;          // 4096 / D + 2
	int segs, tiles_per_seg, nlist;  // as in fused_kernel.h: runs of tiles, long segments first
	int seg_start[fused::kMaxSegList + 1];
	const uint8_t *dummy_tile;  // what the reload reads after a segment's last tile (fused_kernel.h)
	int has_first;              // k_boxcar_scan: the LDS copy of each dword's first sample exists (odd D)
	int R;                      // k_boxcar_scan: outputs per lane and tile, ceil((4096 / D + 1) / 64)
	// emit mode (k_boxcar_scan<3>): the decimated IQ itself - what full_demod() hands on after low_pass()
	// (src/rtl_fm.c:1200-1202) - as packed int16 pairs, output k of the run at emit_iq[s * emit_iq_stride + k]:
	// the input of the squelch / -L level / -M raw kernels (rtlfm_hip.hip, run_boxfused_emit)
	uint32_t *emit_iq;
	size_t emit_iq_stride;      // dwords between streams
	const int2 *rdc_avg;        // RDC kernels: [stream][nblocks] (avgI, avgQ) of dc_block_raw_filter (k_rdc_sums_wide / k_rdc_smooth)
	// RDC on buffers shorter than a tile (round 6): a tile then holds up to kRdcMany buffers' averages - a table in LDS at
	// dword rdc_tab ((a_j, base_j) per buffer of the tile), and x / (N0 / 256) for x < 64 as a multiply and a shift
	int rdc_many, rdc_tab;
	uint32_t rdc_m_magic;       // ceil(65536 / (N0 / 256))
	// SQ kernels (the power squelch / -L behind the boxcar, round 5): [stream][nblocks] (sum of squares, sum) of every
	// buffer's decimated elements, both modulo 2^32 as rms() has them (src/rtl_fm.c:1093-1098) - zeroed by the host, added
	// to with atomics (a buffer's outputs come from several waves); k_squelch_apply makes the decisions
	uint32_t *sq_sums;
	// how the outputs leave (round 5, tools/write_share_probe.hip): 0 = plain stores, whatever a tile has; 1 = whole 128-byte
	// lines only, non-temporal - what is left of a tile's last line waits in LDS for the next tile.  For outputs that do
	// not fit the 256 MiB Infinity Cache anyway (the launch writes more than that): the skeleton of this kernel with 816 /
	// 1360 bytes per tile (/10, /6) takes 0.868 / 0.960 ms per 4 GiB with plain stores and 0.795 / 0.882 this way.
	int store_lines_nt;
	unsigned long long *stamps;  // RTLFM_BOX_PHASES builds (tools/box_phases.py): [waves][4] shader cycles by phase of the tile loop
};

// Per 8 KiB tile (round 1 walked every window: O(D/2) LDS gathers and dot products per output and
// per component, 1619 VALU instructions per tile at /6):
//   1. the coalesced (non-temporal) loads are staged through LDS as S = raw ^ 0x7f and read back
//      lane-contiguous: lane l owns dwords [32l, 32l + 32) = samples [64l, 64l + 64).  The 16-byte
//      slots of a row are XOR-swizzled with the row number (row_at), which makes both the 16-byte
//      stores (eight consecutive lanes fill one row) and the 16-byte read-backs (eight lanes, eight
//      rows) cover all 32 banks exactly once;
//   2. a running sum along the lane's 32 dwords: two v_dot4_i32_i8 per dword whose +-1 taps carry
//      the (-j)^n rotation and whose third operand is the sum so far - the chain IS the prefix.
//      Before each dword the pair (I, Q) is packed to 16 + 16 bits with one v_perm: the reference
//      stores the sums as int16 (src/rtl_fm.c:473-474), and truncation is a ring homomorphism, so
//      every later add / subtract may wrap in 16-bit lanes (v_pk_add_u16 / v_pk_sub_u16);
//   3. an exclusive wave scan (DPP row_shr / row_bcast) of the 64 lane totals in 32 bits, added to
//      the lane's packed prefixes, which go back to the lane's own LDS row: P2[d] = sum of the
//      rotated samples [0, 2d) of the tile;
//   4. output e ends at sample n = (e + 1) D - phase: one LDS read of P2[n >> 1]; an odd n (odd
//      D, or an odd phase injected through the state) adds the dword's first sample from a
//      compact byte-pair copy of the tile.  The window's other end is the neighbouring lane's
//      value (DPP wave_shr:1), the previous output for the discriminator likewise, so there is
//      no LDS store -> load dependency between the rounds;
//   5. the unfinished window: tile total - P(last boundary), exact in 16 bits for D <= 256; for
//      larger D the run's final partial sum is redone in 32 bits by k_boxcar_partial32.
struct ScanLds {
	static constexpr int atan = 0;                   // 17 doubles
	static constexpr int scratch = 34;               // boundary and value of the tile's last complete output
	static constexpr int rows = 36;                  // S, then P2: 64 rows of 32 dwords, 36 apart; + P2[2048]
	static constexpr int row_stride = 32;            // rows back to back, the 16-byte slot index XORed with the row (row_at)
	static constexpr int first = rows + 65 * row_stride;  // byte pairs of each dword's first sample: 64 rows of 16 dwords, 20 apart
	static constexpr int first_stride = 20;
	// the byte-pair copy is only allocated for odd D.  With an even D the window ends are all even or
	// all odd for the whole run (4096 and the buffer are even), and odd only if an odd prev_index was
	// injected through rtlfm_gpu_state_set: that run fetches the bytes from the input itself.
	__host__ __device__ static int pcm(bool has_first) { return has_first ? first + 64 * first_stride : first; }
	// the tile's outputs waiting for their aligned 16-byte stores: int16 PCM, or (emit mode) packed IQ dwords
	__host__ __device__ static int total(int out_cap, bool has_first, bool emit = false)
	{
		// + the piece of a line carried from tile to tile (store_lines_nt): fewer than 128 bytes
		return pcm(has_first) + (emit ? out_cap + 8 + 32 : (out_cap + 16 + 64) / 2);
	}
};

// inclusive scan over the 64 lanes of a wave
__device__ __forceinline__ int wave_inclusive_scan(int x)
{
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
	x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
	x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
	return x;
}

// One dword of the running sum: pk = the sums BEFORE this dword packed to 16 + 16 bits (the
// exclusive prefix), then both components advanced by the dword's two rotated samples.
// Inline assembly because (a) hipcc only emits the accumulate-in-place v_dot4c here and copies the
// old sums first (two v_mov and wait states per dword), and (b) on gfx950 a VALU instruction that
// is not the same dot opcode must stay three instructions behind the dot whose result it reads —
// the hazard recogniser does not look into inline assembly, so the order is fixed here: the v_perm
// reads sums that were written by the previous block's dots, three instructions back.  The dots
// themselves read the previous sums as their accumulate operand, which forwards without a wait.
__device__ __forceinline__ void chain_step(uint32_t w, int tapI, int tapQ, int &accI, int &accQ, uint32_t &pk)
{
	int ni, nq;
	uint32_t k;
	asm("v_dot4_i32_i8 %0, %3, %4, %6\n\t"
	    "v_dot4_i32_i8 %1, %3, %5, %7\n\t"
	    "v_perm_b32 %2, %7, %6, %8"
	    : "=&v"(ni), "=&v"(nq), "=&v"(k)
	    : "v"(w), "s"(tapI), "s"(tapQ), "v"(accI), "v"(accQ), "s"(0x05040100u));
	accI = ni; accQ = nq; pk = k;
}
// after the last dword the totals go into DPP instructions: three wait states for the dot result
// plus the two a DPP source needs
__device__ __forceinline__ void chain_settle(int &accI, int &accQ)
{
	asm volatile("s_nop 4" : "+v"(accI), "+v"(accQ));
}

// Where dword d of the tile-sized S / P2 area lives: row r = d >> 5 (one lane's 32 dwords) keeps its
// place, its eight 16-byte slots are permuted by r & 7.  Eight consecutive lanes reading or writing
// "their" slot j then touch eight different slots = all 32 banks once, as do eight lanes that fill
// one row - without the four padding dwords per row that a skewed layout costs: 9.2 KiB per wave
// instead of 10.4, which is the difference between three and four waves per SIMD (-7 % at /10,
// -9 % at /6).
__device__ __forceinline__ int row_at(int d)
{
	return d ^ ((int)__builtin_amdgcn_ubfe((uint32_t)d, 5, 3) << 2);
}

__device__ __forceinline__ uint32_t pk_add16(uint32_t a, uint32_t b) { return fused::as_u32(fused::as_s2(a) + fused::as_s2(b)); }
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b) { return fused::as_u32(fused::as_s2(a) - fused::as_s2(b)); }

// Measurement builds only (tools/build_variant.sh box_phases -DRTLFM_BOX_PHASES): every wave adds up the shader cycles
// it spends in the four phases of the tile loop - 0 stage + flush + issue of the next tile's loads (incl. the wait for this
// tile's), 1 the running sums, 2 scan + prefixes back to LDS, 3 the output loop and what follows it.
#ifdef RTLFM_BOX_PHASES
#define BOX_PHASE(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph_t[k] += now_ - ph_tp; ph_tp = now_; } while (0)
#else
#define BOX_PHASE(k) do { } while (0)
#endif

#ifndef RTLFM_BOX_UNROLL2
#define RTLFM_BOX_UNROLL2 0
#endif
#ifndef RTLFM_BOXSCAN_WAVES_PER_SIMD
#define RTLFM_BOXSCAN_WAVES_PER_SIMD 4
#endif

// V: 1 = -M fm -A std, 2 = -M fm -A fast (each with its discriminator compiled in: as run-time
// choices inside the output loop they cost ~40 scalar instructions per output), 0 = everything else
// (-A lut, AM / USB / LSB), chosen at run time; 3 = emit mode: no demodulator, the decimated IQ is stored
// (the power squelch, -L and -M raw work on it: src/rtl_fm.c:1204-1237, 1006-1009)
// RDC: dc_block_raw_filter (-E rdc, src/rtl_fm.c:1043-1065) in front of the boxcar.  It subtracts one (aI, aQ) per
// buffer before the rotation, and the rotated constant sums to zero over every four samples: with c = aI + j aQ,
// sum_{m<n} (-j)^m c = c G(n & 3), G = 0, 1, 1 - j, -j.  So P~(n) = P(n) - c G(n & 3) is the prefix sum of the filtered
// samples, a tile starts and ends with a correction of zero, and the only change is one subtraction where a prefix is
// looked up.  Buffers are multiples of four samples, so where a buffer ends INSIDE a tile (round 5: any buffer of at
// least 8192 bytes, not only whole tiles - a tile then holds samples of at most two buffers) the samples before the
// boundary have summed to zero as well, and a look-up behind the boundary takes the second buffer's averages.  The
// averages come from a pre-pass over the input (k_rdc_sums_wide, k_rdc_smooth), as for the fifth_order front end.
// Rotating chains only (no offset tuning).
// SQ: the launch also leaves what rms() needs of every buffer's decimated IQ (Params::sq_sums), so that the squelch needs
// no emit mode: a muted buffer's PCM is zero whatever the samples were (the discriminator of zeroed samples, and of the
// first sample behind them, is atan2(0, 0) = 0; am / usb / lsb of zeros are zeros), and k_squelch_apply writes those
// zeros once it knows the levels.  Buffers of at least 8192 samples (a tile's outputs belong to two buffers at most).
template <int V, bool RDC = false, bool SQ = false>
__global__ void __launch_bounds__(64, RTLFM_BOXSCAN_WAVES_PER_SIMD) k_boxcar_scan(const Params p)
{
	static_assert(!SQ || V != 3, "emit mode hands the samples themselves on");
	extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
	const int lane = threadIdx.x;
	const int wave = blockIdx.x;
	const int seg = wave / p.nstreams;
	const int s = wave - seg * p.nstreams;
	if (seg >= p.segs) return;
	const long long run_bytes = (long long)p.nblocks * p.block_len;
	const int total_tiles = (int)((run_bytes + kTileBytes - 1) / kTileBytes);
	const int N0 = (int)(p.block_len / 2);  // samples per buffer
	// only the run's last tile can be cut short: everything the tile loop asks about a tile's extent is a 32-bit
	// compare against these two (round 5: the 64-bit products and divisions per tile were 250 scalar instructions)
	const int last_tile = total_tiles - 1;
	const int last_valid = (int)(run_bytes - (long long)last_tile * kTileBytes);  // bytes of the last tile, 512 .. 8192
	int t0, t1;
	fused::segment_bounds(p, seg, total_tiles, t0, t1);
	if (t0 >= t1) return;
	const bool from_state = (t0 == 0);
	const bool writes_state = (t1 == total_tiles);
	const state_t *sin = p.sin + s;
	state_t *sout = p.sout + s;
	const int D = p.D;
	const int pcm_at = ScanLds::pcm(p.has_first != 0);
	uint16_t *pcm = reinterpret_cast<uint16_t *>(lds + pcm_at);

	if (writes_state) {
		const uint32_t *a = reinterpret_cast<const uint32_t *>(sin);
		uint32_t *b = reinterpret_cast<uint32_t *>(sout);
		for (int k = lane; k < (int)(sizeof(state_t) / 4); k += 64) b[k] = a[k];
	}
	if (lane < 17) reinterpret_cast<double *>(lds + ScanLds::atan)[lane] = k_atan_nodes[lane];
	const int p0 = sin->prev_index;
	const int gt_first = t0;
	const int gt_begin = from_state ? gt_first : gt_first - 1;  // one warm-up tile
	const int gt_end = t1;
	int ph, kb;
	{
		const long long n = (long long)p0 + (long long)gt_begin * kTileSamples;
		ph = (int)(n % D);
		kb = (int)(n / D);
	}
	// (ph + x) / D for a position inside the tile's range of outputs: a multiply-high while n D < 2^32; beyond /2047
	// (round 6) a tile completes two outputs at most, so the quotient is 0, 1 or 2: two compares
	auto div_D = [&](int n) -> int {
		if (D == 1) return n;
		if (D > kMaxD) return (n >= D ? 1 : 0) + (n >= 2 * D ? 1 : 0);
		return (int)__umulhi((uint32_t)n, p.D_magic);
	};
	// 4096 is even, so the parity of the phase is that of p0 when D is even: wave-uniform for the run
	const bool need_odd = (D & 1) || (ph & 1);
	const bool first_in_lds = need_odd && p.has_first;
	// the unfinished window carried into the tile (now_r, now_j) and the last complete output (I, Q packed)
	int carry_r = from_state ? sin->now_r : 0, carry_j = from_state ? sin->now_j : 0;
	uint32_t last_out = from_state ? pack_iq((int16_t)sin->pre_r, (int16_t)sin->pre_j) : 0u;
	__builtin_amdgcn_wave_barrier();
	const fused::AtanNodesLds nodes{reinterpret_cast<const double *>(lds + ScanLds::atan)};

	const int tI_even = p.rotate ? (int)0xFF0000FFu : (int)0x00FF00FFu;
	const int tQ_even = p.rotate ? (int)0x0001FF00u : (int)0xFF00FF00u;
	const int tI_odd = p.rotate ? (int)0x01000001u : tI_even;
	const int tQ_odd = p.rotate ? (int)0x00FF0100u : tQ_even;

	const uint8_t *stream_base = p.iq + (size_t)s * p.stream_stride;
	constexpr bool EMIT = V == 3;
	int16_t *out_base = EMIT ? nullptr : p.out + (size_t)s * p.out_stride;
	uint32_t *emit_base = EMIT ? p.emit_iq + (size_t)s * p.emit_iq_stride : nullptr;
	uint4 cur[8];
	// The tile's eight 1 KiB rows.  Only the run's last tile can be partial (a multiple of 512 bytes): whole
	// rows behind its end come from the dummy tile (a scalar select per row), and the row its end cuts in
	// two is read 512 bytes early, so that no byte outside the run is touched - stage() puts it right.
	auto load_tile = [&](int tile) {
		const bool real = tile < gt_end;
		const uint8_t *tb = real ? stream_base + (size_t)tile * kTileBytes : p.dummy_tile;
		const int valid = tile == last_tile ? last_valid : kTileBytes;  // wave-uniform
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const uint8_t *row = tb + k * 1024;
			if (k * 1024 >= valid) row = p.dummy_tile + k * 1024;
			else if (k * 1024 + 512 == valid) {
				if (tile > 0 || k > 0) row -= 512;
				else row = lane < 32 ? row : p.dummy_tile;  // a run of 512 bytes in all: nothing before it to read early
			}
			cur[k] = fused::load_stream16(row + lane * 16);
		}
	};
	load_tile(gt_begin);

	// The outputs of a tile wait in LDS and leave at the top of the next iteration (before the loads of the tile after it).
	// LDS element i mirrors row element lbase + i, and lbase is chosen so that 16-byte pieces of the LDS area are aligned
	// 16-byte pieces of the row: they leave as one dwordx4 per lane (all read first, then stored), the partial pieces at
	// both ends element by element.  store_lines_nt: lbase is a line boundary of the row, a flush only stores whole
	// 128-byte lines (non-temporal) and moves what is left - the beginning of a line the next tile will complete - to the
	// front of the area; only a segment's first and last flush have partial lines (plain stores).
	constexpr int ES = EMIT ? 4 : 2;        // bytes per element: packed IQ dwords / int16 PCM
	constexpr int PE = 16 / ES, LE = 128 / ES;  // elements per 16-byte piece, per line
	const bool lines_nt = p.store_lines_nt != 0;
	const uint32_t row_el = (uint32_t)((EMIT ? (uintptr_t)emit_base : (uintptr_t)out_base) / ES);  // the row's address in elements (low bits)
	int lbase = 0;      // the row element LDS element 0 stands for
	int done = -1;      // row elements below this one are stored (-1: the segment has not begun)
	int flush_end = 0;  // row elements below this one are in LDS
	bool flush_last = false;
	auto pcm_idx = [&](int e) -> int { return kb - lbase + e; };
	auto store16 = [&](void *g, const uint4 v) {
		typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
		const u32x4v vv = {v.x, v.y, v.z, v.w};
		// (s_nop: a VALU write to the data registers of a store of more than 64 bits must stay one wait state behind it, and
		// the compiler does not know that this line is one)
		if (lines_nt) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 0" : : "v"(g), "v"(vv) : "memory");
		else *reinterpret_cast<uint4 *>(g) = v;
	};
	auto flush = [&]() {
		if (done < 0 || flush_end <= done) return;
		uint8_t *rowb = EMIT ? reinterpret_cast<uint8_t *>(emit_base) : reinterpret_cast<uint8_t *>(out_base);
		typedef uint4 __attribute__((may_alias)) u128_alias;  // the PCM is written as uint16
		const u128_alias *l128 = reinterpret_cast<const u128_alias *>(lds + pcm_at);
		const int gran = lines_nt && !flush_last ? LE : 1;
		// [done, upto) leaves now; whole pieces [h, u) of it as 16-byte stores
		const int upto = gran == 1 ? flush_end : flush_end - (int)((row_el + (uint32_t)flush_end) & (LE - 1));
		const bool leaves = upto > done;  // (lines: a tile may complete no line - / 334 has 24 bytes per tile)
		if (leaves) {
			int h = done + (int)((0u - (row_el + (uint32_t)done)) & (PE - 1));
			if (h > upto) h = upto;
			const int u = h + ((upto - h) & ~(PE - 1));
			const int j0 = (h - lbase) / PE, j1 = (u - lbase) / PE;  // lbase is a piece boundary
			for (int j = j0 + lane; j < j1; j += 64) store16(rowb + ((size_t)lbase + (size_t)j * PE) * ES, l128[j]);
			// the partial pieces: fewer than PE elements each
			if (lane < PE && done + lane < h) {
				if (EMIT) reinterpret_cast<uint32_t *>(rowb)[done + lane] = lds[pcm_at + done - lbase + lane];
				else reinterpret_cast<int16_t *>(rowb)[done + lane] = (int16_t)pcm[done - lbase + lane];
			}
			if (lane >= PE && lane < 2 * PE && u + (lane - PE) < upto) {
				const int k = u + (lane - PE);
				if (EMIT) reinterpret_cast<uint32_t *>(rowb)[k] = lds[pcm_at + k - lbase];
				else reinterpret_cast<int16_t *>(rowb)[k] = (int16_t)pcm[k - lbase];
			}
			done = upto;
		}
		if (gran > 1 && leaves) {  // done is a line boundary now
			// the beginning of the next line to the front: at most eight pieces
			const int pieces = (flush_end - done + PE - 1) / PE, from = (done - lbase) / PE;
			uint4 keep = make_uint4(0, 0, 0, 0);
			if (lane < pieces) keep = l128[from + lane];
			__builtin_amdgcn_wave_barrier();
			if (lane < pieces) *reinterpret_cast<uint4 *>(lds + pcm_at + 4 * lane) = keep;
			__builtin_amdgcn_wave_barrier();
			lbase = done;
		}
	};

	// (RDC) the buffer tile gt's first sample lies in and that sample's place inside it, carried from tile to tile
	int rdc_b = 0, rdc_off = 0;
	// (RDC, buffers shorter than a tile, round 6) lane j's averages of the j-th buffer of the NEXT tile to be worked on: loaded a
	// tile ahead and in front of the tile prefetch, so that the in-order vmcnt never makes the prefetch land for them
	// (the table behind p.rdc_avg has kRdcMany entries of slack behind the last stream's last buffer: rtlfm_hip.hip)
	int2 a_tab = make_int2(0, 0);
	if constexpr (RDC) {
		const long long g = (long long)gt_begin * kTileSamples;
		rdc_b = (int)(g / N0);
		rdc_off = (int)(g - (long long)rdc_b * N0);
		if (p.rdc_many && lane < kRdcMany) a_tab = p.rdc_avg[(size_t)s * p.nblocks + rdc_b + lane];
	}
	// fm_demod's first output of every buffer (below, behind the output loop): xs = where the next buffer start that no
	// tile has looked at yet lies, as a sample position relative to tile gt's first sample.  The outputs that complete
	// in tile gt cover the samples [-ph, Et D - ph) of it; consecutive tiles' ranges follow each other without a gap.
	int xs;
	int sq_b;  // SQ: the buffer the outputs in front of that start belong to (-1: there are none)
	{
		const long long g0 = (long long)gt_begin * kTileSamples - ph;          // the only 64-bit division of the wave
		const long long bb0 = g0 <= 0 ? 0 : (g0 + N0 - 1) / N0;
		xs = (int)(bb0 * N0 - (long long)gt_begin * kTileSamples);
		sq_b = (int)bb0 - 1;
	}
	fused::ProgressPrio prio(gt_end - gt_begin, 0);
#ifdef RTLFM_BOX_PHASES
	unsigned long long ph_t[4] = {0, 0, 0, 0}, ph_tp = __builtin_amdgcn_s_memtime();
#endif
	for (int gt = gt_begin; gt < gt_end; gt++) {
		prio.at(gt - gt_begin);
		const bool emit = gt >= gt_first;
		const bool partial = gt == last_tile && last_valid < kTileBytes;  // the run's last tile, cut short
		const int vs = partial ? last_valid / 2 : kTileSamples;            // samples in this tile
		int Et, ph_next;
		if (!partial) {
			const int wrap = (ph + p.r4096 >= D) ? 1 : 0;
			Et = p.q4096 + wrap;
			ph_next = ph + p.r4096 - (wrap ? D : 0);
		} else {
			Et = __builtin_amdgcn_readfirstlane((ph + vs) / D);
			ph_next = ph + vs - Et * D;
		}
		// c G(1), c G(2), c G(3) as packed int16 pairs, for the buffer the tile starts in (lo) and the one that follows (hi);
		// dc_nb: the tile's first sample of the second buffer (beyond the tile when it holds one buffer's samples only)
		uint32_t dc1 = 0, dc2 = 0, dc3 = 0, dh1 = 0, dh2 = 0, dh3 = 0, dc_base = 0;
		int dc_nb = 1 << 30;
		int rdc_off0 = 0;  // (RDC, many buffers per tile) this tile's first sample's place in its buffer
		if constexpr (RDC) {
		if (p.rdc_many) {
			// Buffers shorter than a tile: the tile holds up to kRdcMany buffers' averages.  Entry j of a table in LDS =
			// (a_j, base_j) of the j-th buffer that has samples in this tile; P_n finds j from n with a shift and a multiply.
			// base_j (offset tuning only: the constant adds up) = what the buffers in front of j contributed minus start_j a_j,
			// so that the correction of a prefix of n samples is n a_j + base_j; all in 16-bit lanes, which wrap as the sums do.
			rdc_off0 = rdc_off;
			uint32_t *tab = lds + p.rdc_tab;
			__builtin_amdgcn_wave_barrier();
			const uint32_t apk = pack_iq((int16_t)a_tab.x, (int16_t)a_tab.y);
			if (lane < kRdcMany) tab[2 * lane] = apk;
			__builtin_amdgcn_wave_barrier();
			if (!p.rotate && lane < kRdcMany) {
				fused::short2_t acc = {(short)0, (short)0};
				for (int i = 0; i < lane; i++) {
					const uint32_t len = (uint32_t)(i == 0 ? N0 - rdc_off : N0) & 0xffffu;
					acc = acc + fused::as_s2(len * 0x00010001u) * fused::as_s2(tab[2 * i]);
				}
				const uint32_t start = (uint32_t)(lane == 0 ? 0 : lane * N0 - rdc_off) & 0xffffu;
				tab[2 * lane + 1] = fused::as_u32(acc - fused::as_s2(start * 0x00010001u) * fused::as_s2(apk));
			}
			__builtin_amdgcn_wave_barrier();
			rdc_off += kTileSamples;
			while (rdc_off >= N0) { rdc_off -= N0; rdc_b++; }  // wave-uniform: sixteen times at most
			// the next tile's averages, in front of the tile prefetch below
			if (lane < kRdcMany) a_tab = p.rdc_avg[(size_t)s * p.nblocks + rdc_b + lane];
		} else {
			const int idx = __builtin_amdgcn_readfirstlane(s * p.nblocks + rdc_b);
			const int idx2 = __builtin_amdgcn_readfirstlane(s * p.nblocks + (rdc_b + 1 < p.nblocks ? rdc_b + 1 : rdc_b));
			const int2 a = p.rdc_avg[idx], a2 = p.rdc_avg[idx2];  // scalar loads: a vector one would share the in-order vmcnt with the tile prefetch
			const int dcI = __builtin_amdgcn_readfirstlane(a.x), dcQ = __builtin_amdgcn_readfirstlane(a.y);
			const int dhI = __builtin_amdgcn_readfirstlane(a2.x), dhQ = __builtin_amdgcn_readfirstlane(a2.y);
			dc1 = pack_iq((int16_t)dcI, (int16_t)dcQ);
			dc2 = pack_iq((int16_t)(dcI + dcQ), (int16_t)(dcQ - dcI));
			dc3 = pack_iq((int16_t)dcQ, (int16_t)-dcI);
			dh1 = pack_iq((int16_t)dhI, (int16_t)dhQ);
			dh2 = pack_iq((int16_t)(dhI + dhQ), (int16_t)(dhQ - dhI));
			dh3 = pack_iq((int16_t)dhQ, (int16_t)-dhI);
			dc_nb = N0 - rdc_off;
			if (!p.rotate && dc_nb < kTileSamples)  // nb (a_lo - a_hi), 16-bit lanes
				dc_base = fused::as_u32(fused::as_s2((uint32_t)dc_nb * 0x00010001u) * (fused::as_s2(dc1) - fused::as_s2(dh1)));
			rdc_off += kTileSamples;
			if (rdc_off >= N0) { rdc_off -= N0; rdc_b++; }  // N0 >= 4096: once at most
		}
		}
		// (RDC) what dc_block_raw_filter has taken off the tile's first n samples' sum, as packed (I, Q) in 16-bit lanes
		auto rdc_corr = [&](int n) -> uint32_t {
			if (p.rdc_many) {
				const uint32_t x = (uint32_t)((n > 0 ? n : 1) - 1 + rdc_off0) >> 8;   // < 64
				const uint32_t j = (x * p.rdc_m_magic) >> 16;                          // x / (N0 / 256): the buffer sample n - 1 lies in
				const uint2 e = *reinterpret_cast<const uint2 *>(lds + p.rdc_tab + 2 * j);
				if (p.rotate) {
					// the rotated constant sums to zero over every four samples, and buffers are whole groups of four:
					// a G(n & 3) of the buffer the prefix ends in - a, (aI + aQ, aQ - aI), (aQ, -aI)
					const int k = n & 3;
					const uint32_t c3 = fused::as_u32(fused::as_s2(__builtin_amdgcn_alignbit(e.x, e.x, 16)) * fused::short2_t{(short)1, (short)-1});
					return k == 0 ? 0u : (k == 1 ? e.x : (k == 2 ? pk_add16(e.x, c3) : c3));
				}
				const uint32_t npk = (uint32_t)n * 0x00010001u;
				return n > 0 ? fused::as_u32(fused::as_s2(npk) * fused::as_s2(e.x) + fused::as_s2(e.y)) : 0u;
			}
			if (p.rotate) {  // (wave-uniform)
				const int k = n & 3;
				const bool hi = n > dc_nb;  // (at the boundary itself k == 0: no correction either way)
				const uint32_t c1 = hi ? dh1 : dc1, c2 = hi ? dh2 : dc2, c3 = hi ? dh3 : dc3;
				return k == 0 ? 0u : (k == 1 ? c1 : (k == 2 ? c2 : c3));
			}
			// offset tuning (round 6): nothing rotates, the constant simply adds up - n samples of the tile hold
			// min(n, nb) times the first buffer's averages and the rest times the second's: one packed multiply-add
			// (16-bit lanes wrap as the sums do), base = nb (a_lo - a_hi) behind the boundary
			const bool hi = n > dc_nb;
			const uint32_t npk = (uint32_t)n * 0x00010001u;  // n < 4097: (n, n)
			return fused::as_u32(fused::as_s2(npk) * fused::as_s2(hi ? dh1 : dc1) + fused::as_s2(hi ? dc_base : 0u));
		};

		// ---- 1. stage S: chunk c = 64k + lane -> row c >> 3, 16-byte slot c & 7
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const int c = 64 * k + lane;
			uint4 v = cur[k];
			v.x ^= 0x7f7f7f7fu; v.y ^= 0x7f7f7f7fu; v.z ^= 0x7f7f7f7fu; v.w ^= 0x7f7f7f7fu;
			*reinterpret_cast<uint4 *>(lds + ScanLds::rows + row_at(4 * c)) = v;
		}
		__builtin_amdgcn_wave_barrier();
		if (partial) {
			// the row the run's end cuts in two was read 512 bytes early (load_tile): its second half is the
			// row's first half, and what follows the end is sample 0 (S = 0), like the dummy rows behind it
			const int valid = 2 * vs, kh = valid >> 10;
			if ((valid & 1023) && (gt > 0 || kh > 0)) {
				const int c = 64 * kh + lane;
				uint4 v = make_uint4(0, 0, 0, 0);
				if (lane < 32) v = *reinterpret_cast<const uint4 *>(lds + ScanLds::rows + row_at(4 * (c + 32)));
				__builtin_amdgcn_wave_barrier();
				*reinterpret_cast<uint4 *>(lds + ScanLds::rows + row_at(4 * c)) = v;
			}
			__builtin_amdgcn_wave_barrier();
		}
		flush();
		// where this tile's outputs wait in LDS
		if (!emit) lbase = kb;  // the warm-up tile: nothing leaves
		else if (done < 0) {    // the segment's first tile with outputs
			done = kb;
			lbase = kb - (int)((row_el + (uint32_t)kb) & (uint32_t)((lines_nt ? LE : PE) - 1));
		} else if (!lines_nt) lbase = kb - (int)((row_el + (uint32_t)kb) & (uint32_t)(PE - 1));  // (everything before kb has left)
		load_tile(gt + 1);  // unconditional: behind the segment's last tile it reads the dummy tile (fused_kernel.h)
		BOX_PHASE(0);

		// ---- 2. the lane's running sum = exclusive prefix before each of its dwords
		uint32_t pk[32];
		int accI = 0, accQ = 0;
#pragma unroll
		for (int j = 0; j < 8; j++) {
			const uint4 v = *reinterpret_cast<const uint4 *>(lds + ScanLds::rows + row_at(32 * lane + 4 * j));
			const uint32_t w[4] = {v.x, v.y, v.z, v.w};
			if (first_in_lds) {
				// bytes 0, 1 of every dword (its first sample), two dwords per dword
				const uint32_t f0 = __builtin_amdgcn_perm(w[1], w[0], 0x05040100u);
				const uint32_t f1 = __builtin_amdgcn_perm(w[3], w[2], 0x05040100u);
				*reinterpret_cast<uint2 *>(lds + ScanLds::first + ScanLds::first_stride * lane + 2 * j) = make_uint2(f0, f1);
			}
#pragma unroll
			for (int q = 0; q < 4; q++) {
				chain_step(w[q], (q & 1) ? tI_odd : tI_even, (q & 1) ? tQ_odd : tQ_even, accI, accQ, pk[4 * j + q]);
			}
		}
		chain_settle(accI, accQ);
		BOX_PHASE(1);
		// ---- 3. exclusive scan of the lane totals, prefixes back to the lane's row
		const int incI = wave_inclusive_scan(accI), incQ = wave_inclusive_scan(accQ);
		const uint32_t off = __builtin_amdgcn_perm((uint32_t)(incQ - accQ), (uint32_t)(incI - accI), 0x05040100u);
		const int totI = __builtin_amdgcn_readlane(incI, 63), totQ = __builtin_amdgcn_readlane(incQ, 63);
		const uint32_t tot = pack_iq((int16_t)totI, (int16_t)totQ);
#pragma unroll
		for (int j = 0; j < 8; j++) {
			uint4 v;
			v.x = pk_add16(pk[4 * j], off); v.y = pk_add16(pk[4 * j + 1], off);
			v.z = pk_add16(pk[4 * j + 2], off); v.w = pk_add16(pk[4 * j + 3], off);
			*reinterpret_cast<uint4 *>(lds + ScanLds::rows + row_at(32 * lane + 4 * j)) = v;
		}
		if (lane == 0) lds[ScanLds::rows + row_at(2048)] = tot;  // P2[2048]
		__builtin_amdgcn_wave_barrier();
		BOX_PHASE(2);

		// ---- 4. lane l takes the R consecutive outputs e = l R .. l R + R - 1: the window's other end
		// P(n_{e-1}) and the previous output are the lane's own values of the iteration before, and for
		// its first output two more look-ups, so nothing is exchanged between lanes and the look-up of
		// output e + 1 is in flight while output e goes through the discriminator.
		// P(n_{-1}) = -(carried partial sum), so that out[0] = carry + P(n_0); z_{-1} = the last output
		// of the previous tile.
		const uint32_t edgeP = pk_sub16(0u, pack_iq((int16_t)carry_r, (int16_t)carry_j));
		auto P_n = [&](int n) -> uint32_t {  // P(n): the prefix sum of the tile's first n samples
			if (n > kTileSamples) n = kTileSamples;  // outputs past Et
			const int d = n >> 1;
			uint32_t Pv = lds[ScanLds::rows + row_at(d)];
			if (need_odd) {
				uint32_t fp;
				if (first_in_lds) fp = reinterpret_cast<const uint16_t *>(lds + ScanLds::first)[d + ((d >> 5) << 3)];
				else fp = (uint32_t)*reinterpret_cast<const uint16_t *>(stream_base + (size_t)gt * kTileBytes + 4 * (d < kTileDwords ? d : 0)) ^ 0x7f7fu;
				// (S_a0 | S_b0 << 8) -> sign-extended 16-bit lanes; S = -(u - 127): the first sample of an
				// even dword (n % 4 == 0) is (a, b) = -S, of an odd one (n % 4 == 2) (-a, -b) = +S
				const uint32_t hi8 = __builtin_amdgcn_perm(0u, fp, 0x010c000cu);  // bytes: 0, a0, 0, b0
				uint32_t fs = fused::as_u32(fused::as_s2(hi8) >> 8);
				const bool plus = p.rotate && (d & 1);
				fs = plus ? fs : pk_sub16(0u, fs);
				Pv = (n & 1) ? pk_add16(Pv, fs) : Pv;
			}
			if constexpr (RDC) Pv = pk_sub16(Pv, rdc_corr(n));
			return Pv;
		};
		auto P_at = [&](int e) -> uint32_t { return P_n((e + 1) * D - ph); };  // P(n_e) for e >= 0
		const int R = p.R;
		const int e0 = lane * R;
		// SQ: the tile's first output of the NEXT buffer (the outputs that complete in this tile cover its samples
		// [-ph, Et D - ph); xs = where the next buffer starts), and the lane's sums for both buffers
		int sq_eb = 1 << 30;
		uint32_t sq_p0 = 0, sq_t0 = 0, sq_p1 = 0, sq_t1 = 0;
		if constexpr (SQ) {
			if (xs < Et * D - ph) sq_eb = div_D(ph + xs);
		}
		uint32_t prevP, b;
		{
			const uint32_t a1 = P_at(e0 > 0 ? e0 - 1 : 0), a2 = P_at(e0 > 1 ? e0 - 2 : 0);
			prevP = e0 > 0 ? a1 : edgeP;
			b = e0 > 0 ? pk_sub16(a1, e0 > 1 ? a2 : edgeP) : last_out;
		}
		uint32_t curP = P_at(e0);
		int n_next = (e0 + 2) * D - ph;  // n of output e + 1: D further per round (a 32-bit multiply per look-up issues at quarter rate)
		// (The per-lane `if`s below are cheaper than they look: a branch-free form - outputs past Et computed
		// and dumped, the last output's boundary kept in registers - measured 8-13 % SLOWER.)
#if RTLFM_BOX_UNROLL2
		// (A/B builds, tools/build_variant.sh box_unroll2 -DRTLFM_BOX_UNROLL2=1: two outputs per round - the three register
		// moves that hand (previous boundary, previous output, next boundary) on to the next round and half of the loop's
		// own bookkeeping go; LAB.md I.29)
		auto out1 = [&](const int e, const uint32_t cP, const uint32_t pP, const uint32_t bb, const bool on) -> uint32_t {
			const uint32_t z = pk_sub16(cP, pP);  // lowpassed[] is int16 (src/rtl_fm.c:473-474)
			if (EMIT) {
				if (on && e < Et) {
					lds[pcm_at + pcm_idx(e)] = z;
					if (e == Et - 1) { lds[ScanLds::scratch] = cP; lds[ScanLds::scratch + 1] = z; }
				}
			} else if (on && e < Et) {
				if constexpr (SQ) {
					// rms()'s two sums over the elements of this output (I and Q), by the buffer the output belongs to
					const fused::short2_t zz = fused::as_s2(z), ones = {(short)1, (short)1};
					const uint32_t sq = (uint32_t)__builtin_amdgcn_sdot2(zz, zz, 0, false), sm = (uint32_t)__builtin_amdgcn_sdot2(zz, ones, 0, false);
					if (e < sq_eb) { sq_p0 += sq; sq_t0 += sm; } else { sq_p1 += sq; sq_t1 += sm; }
				}
				const uint32_t bsw = __builtin_amdgcn_alignbit(bb, bb, 16);
				const uint32_t bx = fused::as_u32(fused::as_s2(bsw) * fused::short2_t{(short)-1, (short)1});
				int cr, cj;
				fused::dot2_pair(z, bb, bx, cr, cj);
				int v;
				if (V == 0 && p.mode != RTLFM_MODE_FM) v = simple_demod(p.mode, z, p.output_scale);
				else if (V == 1) v = atan2_q14(cj, cr, nodes);
				else if (V == 2) v = fast_atan2_q14(cj, cr);
				else if (p.variant == RTLFM_ATAN_FAST) v = fast_atan2_q14(cj, cr);
				else v = lut_atan2_q14_direct(cj, cr, nodes);
				pcm[pcm_idx(e)] = (uint16_t)(int16_t)v;
				if (e == Et - 1) {  // the last complete output: its boundary and its value
					lds[ScanLds::scratch] = cP;
					lds[ScanLds::scratch + 1] = z;
				}
			}
			return z;
		};
		for (int r = 0; r < R; r += 2) {
			const uint32_t P1 = P_n(n_next), P2 = P_n(n_next + D);
			n_next += 2 * D;
			const uint32_t z0 = out1(e0 + r, curP, prevP, b, true);
			const uint32_t z1 = out1(e0 + r + 1, P1, curP, z0, r + 1 < R);  // (an odd R: the lane's range ends in the middle of the round)
			prevP = P1; b = z1; curP = P2;
		}
#else
		for (int r = 0; r < R; r++) {
			const int e = e0 + r;
			const uint32_t nxtP = P_n(n_next);
			n_next += D;
			const uint32_t z = pk_sub16(curP, prevP);  // lowpassed[] is int16 (src/rtl_fm.c:473-474)
			if (EMIT) {
				if (e < Et) {
					lds[pcm_at + pcm_idx(e)] = z;
					if (e == Et - 1) { lds[ScanLds::scratch] = curP; lds[ScanLds::scratch + 1] = z; }
				}
			} else if (e < Et) {
				if constexpr (SQ) {
					// rms()'s two sums over the elements of this output (I and Q), by the buffer the output belongs to
					const fused::short2_t zz = fused::as_s2(z), ones = {(short)1, (short)1};
					const uint32_t sq = (uint32_t)__builtin_amdgcn_sdot2(zz, zz, 0, false), sm = (uint32_t)__builtin_amdgcn_sdot2(zz, ones, 0, false);
					if (e < sq_eb) { sq_p0 += sq; sq_t0 += sm; } else { sq_p1 += sq; sq_t1 += sm; }
				}
				const uint32_t bsw = __builtin_amdgcn_alignbit(b, b, 16);
				const uint32_t bx = fused::as_u32(fused::as_s2(bsw) * fused::short2_t{(short)-1, (short)1});
				int cr, cj;
				fused::dot2_pair(z, b, bx, cr, cj);
				int v;
				if (V == 0 && p.mode != RTLFM_MODE_FM) v = simple_demod(p.mode, z, p.output_scale);
				else if (V == 1) v = atan2_q14(cj, cr, nodes);
				else if (V == 2) v = fast_atan2_q14(cj, cr);
				else if (p.variant == RTLFM_ATAN_FAST) v = fast_atan2_q14(cj, cr);
				else v = lut_atan2_q14_direct(cj, cr, nodes);
				pcm[pcm_idx(e)] = (uint16_t)(int16_t)v;
				if (e == Et - 1) {  // the last complete output: its boundary and its value
					lds[ScanLds::scratch] = curP;
					lds[ScanLds::scratch + 1] = z;
				}
			}
			prevP = curP; b = z; curP = nxtP;
		}
#endif
		if constexpr (SQ) {
			if (emit && Et > 0) {
				// the wave's sums into the buffers' (an atomic per sum: four per tile at most; buffers whose outputs this tile
				// does not hold add nothing)
				const int a0 = wave_inclusive_scan((int)sq_p0), a1 = wave_inclusive_scan((int)sq_t0);
				if (lane == 63 && sq_b >= 0 && sq_eb > 0) {
					uint32_t *d = p.sq_sums + ((size_t)s * p.nblocks + sq_b) * 2;
					atomicAdd(d, (uint32_t)a0); atomicAdd(d + 1, (uint32_t)a1);
				}
				if (sq_eb < Et) {
					const int c0 = wave_inclusive_scan((int)sq_p1), c1 = wave_inclusive_scan((int)sq_t1);
					if (lane == 63 && sq_b + 1 < p.nblocks) {
						uint32_t *d = p.sq_sums + ((size_t)s * p.nblocks + sq_b + 1) * 2;
						atomicAdd(d, (uint32_t)c0); atomicAdd(d + 1, (uint32_t)c1);
					}
				}
			}
		}
		if (Et > 0) {
			const int lim = Et * D - ph;
			if (xs < lim) {
				if (V != 1 && !EMIT && (V != 0 || p.mode == RTLFM_MODE_FM)) {
					// fm_demod's first sample of a buffer is always polar_discriminant, whatever -A says
					// (src/rtl_fm.c:935-937): redone here, behind the loop, instead of as a second discriminator under a
					// per-lane condition inside it (where it cost every output of a -A fast run the std path's
					// instructions as well).  Buffer b starts at run sample b N0 and its first output is output
					// (p0 + b N0) / D of the run: lane c looks at the c-th buffer start that falls into this tile's range.
					// (p0 + b N0) / D - kb = (ph + x) / D with x = b N0 - 4096 gt, the buffer start relative to this tile: small
					// numbers, one multiply-high per buffer start (round 5; three 64-bit divisions per tile until then).
					__builtin_amdgcn_wave_barrier();
					for (int x = xs + lane * N0; x < lim; x += 64 * N0) {
						const int e = div_D(ph + x);
						const uint32_t Pe = P_at(e), P1 = e > 0 ? P_at(e - 1) : edgeP, P2 = e > 1 ? P_at(e - 2) : edgeP;
						const uint32_t z0 = pk_sub16(Pe, P1), b0 = e > 0 ? pk_sub16(P1, P2) : last_out;
						const uint32_t bsw = __builtin_amdgcn_alignbit(b0, b0, 16);
						const uint32_t bx = fused::as_u32(fused::as_s2(bsw) * fused::short2_t{(short)-1, (short)1});
						int cr0, cj0;
						fused::dot2_pair(z0, b0, bx, cr0, cj0);
						const int v0 = atan2_q14(cj0, cr0, nodes);
						pcm[pcm_idx(e)] = (uint16_t)(int16_t)v0;
					}
				}
				do { xs += N0; sq_b++; } while (xs < lim);  // wave-uniform
			}
		}
		xs -= kTileSamples;
		__builtin_amdgcn_wave_barrier();
		// (a tile that completes no output - boxcars beyond /4096, round 6 - leaves the last output where it was and adds
		// all of itself to the unfinished window: "the boundary" is then the carried sum's negative, as at the tile's start)
		const uint32_t Plast = Et > 0 ? lds[ScanLds::scratch] : edgeP;
		if (Et > 0) last_out = lds[ScanLds::scratch + 1];
		// ---- 5. the window the tile leaves unfinished: fewer than D <= 256 samples, exact in 16 bits
		{
			uint32_t tot_c = tot;
			// (with the rotation the constant has summed to zero over the tile's whole groups of four - vs is a multiple of
			// four -; without it - offset tuning - the tile's vs samples hold their buffers' averages times their counts)
			if constexpr (RDC) tot_c = pk_sub16(tot, rdc_corr(vs));
			const iq16 part = unpack_iq(pk_sub16(tot_c, Plast));
			carry_r = part.i; carry_j = part.q;
		}
		__builtin_amdgcn_wave_barrier();
		if (emit) { flush_end = kb + Et; flush_last = gt + 1 == gt_end; }
		kb += Et;
		ph = ph_next;
		BOX_PHASE(3);
	}
#ifdef RTLFM_BOX_PHASES
	if (p.stamps && lane == 0) for (int k = 0; k < 4; k++) p.stamps[(size_t)wave * 4 + k] = ph_t[k];
#endif
	flush();
	if (writes_state && lane == 0) {
		if constexpr (RDC) {  // dc_block_raw_filter keeps the averages of the buffer it saw last (src/rtl_fm.c:1062-1063)
			const int2 a = p.rdc_avg[(size_t)s * p.nblocks + (p.nblocks - 1)];
			sout->dc_avgI = a.x; sout->dc_avgQ = a.y;
		}
		sout->prev_index = ph;
		sout->now_r = carry_r;
		sout->now_j = carry_j;
		const iq16 w = unpack_iq(last_out);
		// only fm_demod keeps them (in emit mode the demodulating kernel behind this one does)
		if (!EMIT && (V != 0 || p.mode == RTLFM_MODE_FM)) { sout->pre_r = w.i; sout->pre_j = w.q; }
		p.cnt[s] = kb;
	}
}

// D > 256: the partial sum the run leaves in (now_r, now_j) no longer fits the 16-bit lanes the
// prefix sums live in (the outputs themselves are int16 by definition, src/rtl_fm.c:473-474, so
// they are unaffected).  One lane per stream adds up the samples behind the last complete output
// again, in 32 bits: fewer than D of them.
__global__ void __launch_bounds__(64) k_boxcar_partial32(const Params p)
{
	const int s = (int)(blockIdx.x * 64 + threadIdx.x);
	if (s >= p.nstreams) return;
	const long long total = (long long)p.nblocks * (p.block_len / 2);
	const int left = p.sout[s].prev_index;  // samples in the unfinished window, written by k_boxcar_scan
	long long first = total - left;         // its first sample of this run (negative: it began before the run)
	int ar = 0, aj = 0;
	if (first < 0) { ar = p.sin[s].now_r; aj = p.sin[s].now_j; first = 0; }
	const uint8_t *src = p.iq + (size_t)s * p.stream_stride;
	const long long n0 = p.block_len / 2;
	for (long long i = first; i < total; i++) {
		int a = (int)src[2 * i] - 127, b = (int)src[2 * i + 1] - 127;
		if (p.rdc_avg) {  // dc_block_raw_filter: this buffer's averages off, before the rotation
			const int2 av = p.rdc_avg[(size_t)s * p.nblocks + (size_t)(i / n0)];
			a -= av.x; b -= av.y;
		}
		// rotate16_neg90 restarts with every buffer, and buffers are multiples of four samples
		switch (p.rotate ? (int)(i & 3) : 0) {
		case 0: ar += a; aj += b; break;
		case 1: ar += b; aj -= a; break;
		case 2: ar -= a; aj -= b; break;
		default: ar -= b; aj += a; break;
		}
	}
	p.sout[s].now_r = ar;
	p.sout[s].now_j = aj;
}

// the decimator itself: what both forms of the launch need
inline bool supported_front(const rtlfm_cfg &c)
{
	// at least two outputs per 4096-sample tile: a wave that starts mid-stream takes its first
	// "previous output" from its warm-up tile
	// (downsample == 1, rtl_fm -s 1.2M: low_pass() hands every sample on - 4096 outputs per tile, the same kernel)
	// (beyond /2047 - rtl_fm -s 400 - a tile completes two outputs at most and a mid-stream wave could not warm up on one
	// tile: those runs take ONE wave per stream, launch(); round 6)
	if (c.downsample_passes != 0 || c.downsample < 1) return false;
	if (c.comp_fir_size) return false;
	// -E rdc: any buffer size (round 6: a tile of buffers shorter than itself looks its averages up in a table in LDS);
	// with the rotation the constant sums to zero over every four samples, with offset tuning - no rotation - it adds up
	// linearly: one packed multiply-add per look-up
	return true;  // else any buffer length (a multiple of 512 bytes): the run is one continuous sample stream here
}

// one launch from the bytes to the PCM
inline bool supported(const rtlfm_cfg &c)
{
	if (c.mode != RTLFM_MODE_FM && c.mode != RTLFM_MODE_AM && c.mode != RTLFM_MODE_USB && c.mode != RTLFM_MODE_LSB)
		return false;
	if (c.squelch_level || c.report_levels) return false;
	return supported_front(c);
}

// the power squelch / -L with the sums taken by the front end itself (SQ kernels) and k_squelch_apply behind it: one launch
// over the input + a small one over the PCM, no emit mode.  Buffers of at least 8192 samples (a tile's outputs then
// belong to two buffers at most); -M raw keeps the emit mode (the samples themselves are the output).
inline bool supported_sq(const rtlfm_cfg &c)
{
	if (c.mode != RTLFM_MODE_FM && c.mode != RTLFM_MODE_AM && c.mode != RTLFM_MODE_USB && c.mode != RTLFM_MODE_LSB)
		return false;
	if (!c.squelch_level && !c.report_levels) return false;
	if (c.block_len < 16384) return false;
	// rms() looks at every step-th element once a buffer holds more than 32768 of them (src/rtl_fm.c:1090-1092): the sums
	// taken here are over all of them
	if (c.downsample < 1 || 2 * ((int)(c.block_len / 2) / c.downsample + 1) > 32768) return false;
	return supported_front(c);
}

// emit mode: the launch stores the decimated IQ, and the squelch (src/rtl_fm.c:1204-1215), the -L levels
// (:1217-1237) and mode_demod incl. -M raw (:1006-1009, 1256-1259) follow on 1 / D of the data
inline bool supported_emit(const rtlfm_cfg &c)
{
	if (!supported_front(c)) return false;
	return c.mode == RTLFM_MODE_RAW || c.squelch_level != 0 || c.report_levels != 0;
}

inline int launch(fused::Workspace &ws, const rtlfm_cfg &c, int nstreams, const uint8_t *d_iq, size_t stream_stride,
                  int nblocks, int16_t *d_out, size_t out_stride, int32_t *d_cnt, const state_t *sin, state_t *sout,
                  hipStream_t q, uint32_t *emit_iq = nullptr, size_t emit_iq_stride = 0, const int2 *rdc_avg = nullptr,
                  uint32_t *sq_sums = nullptr)
{
	Params p{};
	p.emit_iq = emit_iq; p.emit_iq_stride = emit_iq_stride;
	p.rdc_avg = rdc_avg;
	p.sq_sums = sq_sums;
	if (sq_sums && emit_iq) return -EINVAL;
	if (int r = fused::ensure_dummy_tile(ws)) return r;
	p.dummy_tile = ws.dummy_tile;
	p.iq = d_iq; p.stream_stride = stream_stride; p.block_len = c.block_len;
	p.nblocks = nblocks; p.nstreams = nstreams;
	p.out = d_out; p.out_stride = out_stride; p.cnt = d_cnt;
	p.sin = sin; p.sout = sout;
	p.variant = c.custom_atan; p.rotate = c.offset_tuning ? 0 : 1;
	p.mode = c.mode; p.output_scale = c.output_scale;
	p.D = c.downsample; p.q4096 = kTileSamples / p.D; p.r4096 = kTileSamples % p.D;
	p.D_magic = p.D == 1 ? 0u : (uint32_t)((0x100000000ull + (uint64_t)p.D - 1) / (uint64_t)p.D);  // (D == 1: no division)
	p.out_cap = kTileSamples / p.D + 2;
	const long long run_bytes = (long long)nblocks * c.block_len;
	const int total_tiles_run = (int)((run_bytes + kTileBytes - 1) / kTileBytes);
	fused::SegPlan sp;
	if (p.D > kMaxD) { sp.segs = 1; sp.tiles_per_seg = total_tiles_run; sp.nlist = 0; }  // every carried quantity reaches back up to 2 D samples
	else sp = fused::plan_segments(ws, nstreams, total_tiles_run);
	p.segs = sp.segs; p.tiles_per_seg = sp.tiles_per_seg; p.nlist = sp.nlist;
	if (sp.nlist) memcpy(p.seg_start, sp.start, sizeof(int) * (size_t)(sp.nlist + 1));
	const int waves = nstreams * sp.segs;
#ifdef RTLFM_BOX_PHASES
	if (ws.want_stamps) {
		if (ws.stamp_waves < waves) { if (ws.stamps) hipFree(ws.stamps); ws.stamps = nullptr; ws.stamp_waves = 0; if (hipMalloc(&ws.stamps, (size_t)waves * 32 * 2) != hipSuccess) return -ENOMEM; ws.stamp_waves = waves; }
		p.stamps = ws.stamps + (size_t)(ws.stamp_seq & 1) * ws.stamp_waves * 4;
		ws.stamp_seq++;
		ws.stamp_last = waves;
	}
#endif
	const bool std_fm = c.custom_atan == RTLFM_ATAN_STD && c.mode == RTLFM_MODE_FM;
	p.has_first = (p.D & 1) ? 1 : 0;
	{
		// outputs that cannot stay in the Infinity Cache anyway leave as whole lines, non-temporal (Params::store_lines_nt)
		const double out_bytes = (double)nstreams * (double)(run_bytes / 2) / p.D * (emit_iq ? 4.0 : 2.0);
		p.store_lines_nt = ws.box_store >= 0 ? (ws.box_store ? 1 : 0) : (out_bytes > 192.0 * 1048576.0 ? 1 : 0);
	}
	p.R = (p.q4096 + 1 + 63) / 64;
	size_t lds_bytes = (size_t)ScanLds::total(p.out_cap, p.has_first != 0, emit_iq != nullptr) * 4;
	if (rdc_avg && c.block_len < (uint32_t)kTileBytes) {
		// buffers shorter than a tile: the tile's table of averages behind everything else in LDS
		p.rdc_many = 1;
		p.rdc_tab = (int)(((lds_bytes / 4) + 1) & ~(size_t)1);
		lds_bytes = (size_t)(p.rdc_tab + 2 * kRdcMany) * 4;
		const uint32_t m = c.block_len / 512;  // N0 / 256, 1 .. 15
		p.rdc_m_magic = (65536u + m - 1) / m;
	}
	const bool fast_fm = c.custom_atan == RTLFM_ATAN_FAST && c.mode == RTLFM_MODE_FM;
#define RTLFM_BOX_GO(VV, RR, SS) hipLaunchKernelGGL((k_boxcar_scan<VV, RR, SS>), dim3(waves), dim3(64), lds_bytes, q, p)
	const int vsel = emit_iq ? 3 : std_fm ? 1 : fast_fm ? 2 : 0;
	if (sq_sums) {
		if (rdc_avg) { if (vsel == 1) RTLFM_BOX_GO(1, true, true); else if (vsel == 2) RTLFM_BOX_GO(2, true, true); else RTLFM_BOX_GO(0, true, true); }
		else { if (vsel == 1) RTLFM_BOX_GO(1, false, true); else if (vsel == 2) RTLFM_BOX_GO(2, false, true); else RTLFM_BOX_GO(0, false, true); }
	} else if (rdc_avg) {
		if (vsel == 3) RTLFM_BOX_GO(3, true, false); else if (vsel == 1) RTLFM_BOX_GO(1, true, false);
		else if (vsel == 2) RTLFM_BOX_GO(2, true, false); else RTLFM_BOX_GO(0, true, false);
	} else {
		if (vsel == 3) RTLFM_BOX_GO(3, false, false); else if (vsel == 1) RTLFM_BOX_GO(1, false, false);
		else if (vsel == 2) RTLFM_BOX_GO(2, false, false); else RTLFM_BOX_GO(0, false, false);
	}
#undef RTLFM_BOX_GO
	if (p.D > 256) hipLaunchKernelGGL(k_boxcar_partial32, dim3((nstreams + 63) / 64), dim3(64), 0, q, p);
	return hipGetLastError() == hipSuccess ? 0 : -EIO;
}

}  // namespace boxfused
}  // namespace rtlfm
