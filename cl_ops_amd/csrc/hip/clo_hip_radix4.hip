// clo_hip_radix4.hip — the LSD radix sort ("satradix") passes for gfx950:
// ranking with thread-private packed counters, the pass kernel, the one-launch
// kernel for small arrays, the MSD bucket split, and their host side.
//
// Structure per pass, the reference's own (sort/clo_sort_satradix.c:264-313):
//   per-tile digit histogram -> scan of the counters in digit-major order
//   (upstream's counters_sum) -> tile-local stable sort + scatter,
// with every element read twice and written once per pass, and TWO digits of
// <= 4 bits per pass: a tile is split by the low digit and then by the high
// digit inside the work-group (two stable local splits through an LDS stage),
// which leaves it sorted by the combined digit D = hi:lo, and the global step
// runs once for D. For the default radix 16 on 32-bit keys: 4 trips through
// HBM, 12 element streams (upstream: 8 digit passes of ~5 element streams + 6
// counter streams each). A requested digit of 5..8 bits is one pass, split in
// two halves. No kernel waits on another work-group: no look-back, no tickets,
// no spinning. The histogram and counter-scan kernels are in
// clo_hip_radixw.hip.
//
// History of the pass, all measured on 2^28 uint32 keys, radix 16 (docs/lab_notebook.md
// §4.1 has the numbers): chained look-back passes (latency-bound, 6.4 ms) ->
// one digit per pass with the next digit's histogram fused into the scatter
// (match-any ranking 5.3 ms, packed counters 3.94 ms) -> digit pairs with the
// tile -> XCD mapping below (3.43 ms; 16 elements per thread 2.94 ms).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "clo_hip.h"
#include "clo_hip_internal.h"
#include "clo_hip_radix_rank.h"

namespace {

// ---------------------------------------------------------------------------
// The pass kernel: two digit steps per trip through HBM.
//
// A tile is split by digit `lo` and then by digit `hi` inside the work-group
// (two stable local splits through the LDS stage), which leaves it sorted by
// the combined digit D = hi:lo; run starts come from the tile's own histogram
// row, global offsets from the scanned counters, and every run leaves the CU as
// contiguous stores. Stability of both splits and of the scatter makes the
// result the one the reference's digit-by-digit passes produce.
//
// Work-group -> tile mapping: work-groups are dealt round-robin over the 8 XCDs
// (observed, not promised — used for speed only: any mapping is correct, tiles
// are independent). Giving each residue class of blockIdx a contiguous range of
// tiles puts neighbouring tiles, whose digit runs share their boundary cache
// lines in the output, behind the same L2, where the two partial writes merge:
// 2^28 8-byte elements 1.84 -> 0.86 ms per pass, 4-byte 0.79 -> 0.69 ms.
// ---------------------------------------------------------------------------
// DIG: every element's NEXT combined digit (8 bits at next_shift) also goes out, one
// byte per element in output order: the next pass's histogram then reads n bytes
// instead of n elements (clo_radixw_launch_tilehist_bytes).
// SEG: a segmented launch (clo_hip_radix_sort_segmented): `n` is the number of TILES of the launch, `tdesc` says which
// segment a tile belongs to and where it lies; positions, offsets and the row of digit bases are the segment's own.
template <typename E, int LB, int HB, bool BIG, bool DIG = false, bool SEG = false>
__global__ __launch_bounds__((pair_shape<E, BIG>::THREADS), (pair_shape<E, BIG>::THREADS >= 1024 ? 8 : 6))   // 3 work-groups per CU (LDS): 6 waves per SIMD, <= 80 VGPRs (59 used); BIG: 2 x 16 waves, <= 64
void clo_radix4_pair_kernel(const E* __restrict__ in, E* __restrict__ out, size_t n,
	unsigned shift, unsigned mask_lo, unsigned mask_hi,
	const unsigned* __restrict__ thist, const unsigned* __restrict__ toff, const unsigned* __restrict__ dbase,
	const unsigned* __restrict__ tinfo, int aligned,
	clo_keyx kx_in, clo_keyx kx_out, unsigned char* __restrict__ dig_out = nullptr, unsigned next_shift = 0,
	const clo_seg_tile* __restrict__ tdesc = nullptr, const E* __restrict__ in2 = nullptr) {

	constexpr int THREADS = pair_shape<E, BIG>::THREADS;
	constexpr int ITEMS = pair_shape<E, BIG>::ITEMS;
	constexpr int TILE = THREADS * ITEMS;
	constexpr int WAVES = THREADS / 64;
	constexpr int R2 = 1 << (LB + HB);
	constexpr int HMAX = pc_words<(LB > HB ? LB : HB)>::H;
	static_assert(R2 <= 256 && R2 <= THREADS, "one thread per combined digit, scanned by the first four waves");

	__shared__ __attribute__((aligned(16))) E s_stage[TILE];
	constexpr bool ALIAS = pair_shape<E, BIG>::ALIAS;   // the table of ends inside the stage
	__shared__ unsigned s_end[ALIAS ? 1 : THREADS * PC_END_STRIDE];
	__shared__ unsigned s_wtot[WAVES][HMAX];
	__shared__ __attribute__((aligned(16))) unsigned s_wbase[WAVES][HMAX];
	__shared__ unsigned s_delta[R2];   // global index = tile-local position + delta[D]
	__shared__ unsigned s_w4[4];

	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	// neighbouring tiles behind the same L2 (see above)
	const unsigned per_xcd = SEG ? (unsigned) ((n + 7) / 8) : (unsigned) ((n + (size_t) TILE * 8 - 1) / ((size_t) TILE * 8));
	const unsigned tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
	size_t base = (size_t) tile * TILE;
	unsigned count, n32;
	if constexpr (SEG) {
		if (tile >= (unsigned) n) return;
		const clo_seg_tile td = tdesc[tile];   // (one 16-byte load, the same for the whole work-group)
		base = (size_t) td.in_base;
		count = td.count_seg & 0xffffu;
		n32 = td.seg_n;
		out += td.out_base;
		if constexpr (DIG) dig_out += td.out_base;
		dbase += (size_t) ((td.count_seg >> 16) & 0x7fffu) * R2;
		if (td.count_seg >> 31) in = in2;   // (a piece in the call's second source; the same for the whole work-group)
	} else {
		if (base >= n) return;
		count = (n - base) < (size_t) TILE ? (unsigned) (n - base) : (unsigned) TILE;
		n32 = n > 0xffffffffull ? 0xffffffffu : (unsigned) n;   // global indices are 32-bit here
	}
	const bool full = count == (unsigned) TILE;
	const unsigned tbase = tid * ITEMS;
	const unsigned mask2 = (mask_hi << LB) | mask_lo;

	// A wave that is about to request its tile goes first on its SIMD (until the requests are out): the other work-group of
	// the CU is in its splits then, and every cycle its VALU work delays these loads is a cycle of HBM time lost. 8-byte
	// elements: uint64 2^28 8.04-8.14 -> 7.90-7.94 ms, pairs 4.15-4.21 -> 4.10-4.12; 4-byte ones: no difference
	// (profiles/r05_ab_wave_priority.txt). The same priority for the scatter's stores too: no further gain.
	__builtin_amdgcn_s_setprio(3);
	// the tile's counters (upstream's counters / counters_sum), requested before the keys
	unsigned h2 = 0, goff = 0;
	if (tid < (unsigned) R2) {
		h2 = thist[(size_t) tile * R2 + tid];
		goff = toff[(size_t) tile * R2 + tid];
		if (dbase) goff += dbase[tid];   // (the one-launch counter scan keeps the digit bases in a row of their own)
	}
	E key[ITEMS];
	if (full) {
		if constexpr (SEG) load_blocked_unaligned<E, ITEMS>(in + base + tbase, key);   // (a segment starts at any element)
		else load_blocked<E, ITEMS>(in + base + tbase, key, aligned != 0);
	} else {
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) key[i] = (tbase + i < count) ? in[base + tbase + i] : (E) 0;
	}
	__builtin_amdgcn_s_setprio(0);
	if (kx_in.kind) {
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) key[i] = clo_keyx_fwd<E>(key[i], kx_in);
	}
	// tile-local start of every combined digit: exclusive scan of the tile's histogram
	{
		const unsigned incl2 = clo_wave_scan_inclusive<unsigned>(h2, lane);
		if (lane == 63 && wave < 4) s_w4[wave] = incl2;
		// (parked in LDS across the first split: three registers fewer while it runs)
		if (tid < (unsigned) R2) s_delta[tid] = goff - (incl2 - h2);
	}

	// Keys that repeat. A tile whose elements all carry ONE combined digit (the histogram kernel has seen
	// it: a constant byte of the key — small keys, a shared prefix, equal keys) is in order already, and
	// splitting it is the worst case of the split: every lane writes its 16 consecutive positions of the stage,
	// a stride of 64 bytes, 16 lanes to a bank (2^28 equal uint32 keys sorted in 3.5 ms against 2.7 for
	// uniform ones, every pass with a constant digit +0.2 ms; now 2.2 ms: profiles/r03_skew_probe.txt).
	// Such a tile goes to the stage as it stands — 16-byte LDS stores — and straight to the scatter.
	const bool single = tinfo[tile] != 0u;   // (one scalar load; the same for the whole work-group)

	if (!single) {
		pc_local_split<E, LB, THREADS, ITEMS, HMAX, pc_no_mid, pc_no_counted, ALIAS>(key, shift, mask_lo, count, s_stage, s_end, s_wtot, s_wbase);
	} else {
		if (full) {
			constexpr int PER = ITEMS * (int) sizeof(E) >= 16 ? 16 / (int) sizeof(E) : ITEMS;
			typedef E vec16 __attribute__((ext_vector_type(PER)));
			#pragma unroll
			for (int k = 0; k < ITEMS / PER; ++k) {
				vec16 t;
				#pragma unroll
				for (int q = 0; q < PER; ++q) t[q] = key[k * PER + q];
				*reinterpret_cast<vec16*>(&s_stage[tbase + k * PER]) = t;
			}
		} else {
			#pragma unroll
			for (int i = 0; i < ITEMS; ++i) if (tbase + i < count) s_stage[tbase + i] = key[i];
		}
		clo_lds_barrier();   // (the wave sums of the histogram row, s_w4, are in: a split has barriers of its own)
	}

	if (tid < (unsigned) R2) {
		unsigned before = 0;   // counts of the lower combined digits in the earlier waves' share of the histogram row
		#pragma unroll
		for (unsigned w = 0; w < 4; ++w) if (w < wave) before += s_w4[w];
		s_delta[tid] -= before;
	}
	if (single) {
		__syncthreads();
	} else if (mask_hi != 0) {
		if (full) {   // 16-byte LDS reads (scalar reads at this lane stride would conflict 8-way)
			constexpr int PER = ITEMS * (int) sizeof(E) >= 16 ? 16 / (int) sizeof(E) : ITEMS;
			typedef E vec16 __attribute__((ext_vector_type(PER)));
			#pragma unroll
			for (int k = 0; k < ITEMS / PER; ++k) {
				const vec16 t = *reinterpret_cast<const vec16*>(&s_stage[tbase + k * PER]);
				#pragma unroll
				for (int q = 0; q < PER; ++q) key[k * PER + q] = t[q];
			}
		} else {
			#pragma unroll
			for (int i = 0; i < ITEMS; ++i) if (tbase + i < count) key[i] = s_stage[tbase + i];
		}
		pc_local_split<E, HB, THREADS, ITEMS, HMAX, pc_no_mid, pc_no_counted, ALIAS>(key, shift + LB, mask_hi, count, s_stage, s_end, s_wtot, s_wbase);
	} else {
		__syncthreads();
	}

	// contiguous runs to HBM: a thread takes VEC consecutive positions of the
	// sorted tile; inside a run they go out as one 16-byte store.
	// Compiled twice, with and without the key transform of the last pass (`kx_out`: signed / floating-point keys):
	// tested per element it put a chain of scalar branches — and two 64-bit register copies per store — into the
	// path of every sort (round 5).
	constexpr int VEC = sizeof(E) >= 8 ? 1 : 4;
	typedef E vecE __attribute__((ext_vector_type(VEC)));
	typedef E vecE_u __attribute__((ext_vector_type(VEC), aligned(sizeof(E))));
	// Four next digits as one dword. When the next digit is a whole byte of the key (next_shift a multiple of 8: any sort
	// whose key starts at bit 0, 8, ...) three byte permutes gather them; otherwise shifts and masks.
	const bool dig_bytes = DIG && VEC == 4 && (next_shift & 7u) == 0u && next_shift < 8u * (unsigned) sizeof(E);
	const unsigned dig_b = next_shift >> 3;   // (wave-uniform: the selectors live in scalar registers)
	const unsigned dig_sel_lo = 0x0c0c0400u + dig_b * 0x0101u, dig_sel_hi = 0x04000c0cu + dig_b * 0x01010000u;
	const unsigned nbits2 = (unsigned) __builtin_popcount(mask2);   // (mask2 is a run of ones from bit 0: a bit-field extract per digit)
	const auto scatter = [&](auto kx_tag) {
		constexpr bool KX = decltype(kx_tag)::value;
		#pragma unroll
		for (int j = 0; j < ITEMS / VEC; ++j) {
			const unsigned p = (j * THREADS + tid) * VEC;
			if (full) {
				const vecE v = *reinterpret_cast<const vecE*>(&s_stage[p]);
				const unsigned d0 = pc_digit<E>(v[0], shift, mask2, nbits2), dl = pc_digit<E>(v[VEC - 1], shift, mask2, nbits2);
				const unsigned gi0 = p + s_delta[d0];
				if (d0 == dl && gi0 <= n32 - VEC) {
					vecE vo = v;
					if constexpr (KX) {
						#pragma unroll
						for (int k = 0; k < VEC; ++k) vo[k] = clo_keyx_inv<E>(v[k], kx_out);
					}
					*reinterpret_cast<vecE_u*>(&out[gi0]) = vo;
					if constexpr (DIG) {
						if constexpr (VEC == 4) {
							typedef unsigned u32_u __attribute__((aligned(1)));
							unsigned dg;
							if (dig_bytes)
								dg = __builtin_amdgcn_perm((unsigned) v[1], (unsigned) v[0], dig_sel_lo) | __builtin_amdgcn_perm((unsigned) v[3], (unsigned) v[2], dig_sel_hi);
							else
								dg = ((unsigned) (v[0] >> next_shift) & 255u) | (((unsigned) (v[1] >> next_shift) & 255u) << 8)
									| (((unsigned) (v[2] >> next_shift) & 255u) << 16) | ((unsigned) (v[3] >> next_shift) << 24);
							*reinterpret_cast<u32_u*>(dig_out + gi0) = dg;
						} else {
							dig_out[gi0] = (unsigned char) (v[0] >> next_shift);
						}
					}
				} else {
					#pragma unroll
					for (int k = 0; k < VEC; ++k) {
						const unsigned gi = p + k + s_delta[pc_digit<E>(v[k], shift, mask2, nbits2)];
						if (gi < n32) {
							out[gi] = KX ? clo_keyx_inv<E>(v[k], kx_out) : v[k];
							if constexpr (DIG) dig_out[gi] = (unsigned char) (v[k] >> next_shift);
						}
					}
				}
			} else {
				#pragma unroll
				for (int k = 0; k < VEC; ++k) {
					if (p + k < count) {
						const E e = s_stage[p + k];
						const unsigned gi = p + k + s_delta[pc_digit<E>(e, shift, mask2, nbits2)];
						if (gi < n32) {
							out[gi] = KX ? clo_keyx_inv<E>(e, kx_out) : e;
							if constexpr (DIG) dig_out[gi] = (unsigned char) (e >> next_shift);
						}
					}
				}
			}
		}
	};
	if (kx_out.kind) scatter(std::true_type()); else scatter(std::false_type());
}

// ---------------------------------------------------------------------------
// Arrays of at most one tile: every digit inside ONE work-group, one launch for
// the whole sort (upstream's harness sweeps sizes from 2^4 up; a multi-kernel
// sort costs a dozen dependent launches however small the array). One local
// split per digit, the tile stays in LDS / registers in between.
// ---------------------------------------------------------------------------
// Two shapes: 512 threads x 8 elements (up to 4096 elements) and 1024 x 16 (1024
// x 8 for 8-byte elements: a 64 KiB stage either way) for up to 16384 / 8192.
constexpr int SMALL_TILE = 512 * 8;
template <typename E> struct small_big { static constexpr int THREADS = 1024, ITEMS = sizeof(E) == 8 ? 8 : 16, TILE = THREADS * ITEMS; };

template <typename E, int BITS, int THREADS, int ITEMS>
__global__ __launch_bounds__(THREADS)
void clo_radix4_small_kernel(const E* in, E* out, unsigned n, unsigned key_shift, unsigned key_bits, clo_keyx kx) {
	constexpr int H = pc_words<BITS>::H;
	constexpr int WAVES = THREADS / 64;
	__shared__ __attribute__((aligned(16))) E s_stage[THREADS * ITEMS];
	__shared__ unsigned s_end[THREADS * PC_END_STRIDE];
	__shared__ unsigned s_wtot[WAVES][H];
	__shared__ __attribute__((aligned(16))) unsigned s_wbase[WAVES][H];

	const unsigned tbase = threadIdx.x * ITEMS;
	E key[ITEMS];
	#pragma unroll
	for (int i = 0; i < ITEMS; ++i) key[i] = (tbase + i < n) ? clo_keyx_fwd<E>(in[tbase + i], kx) : (E) 0;
	for (unsigned done = 0; done < key_bits; done += BITS) {
		const unsigned bits = key_bits - done < (unsigned) BITS ? key_bits - done : (unsigned) BITS;
		pc_local_split<E, BITS, THREADS, ITEMS, H>(key, key_shift + done, (1u << bits) - 1u, n,
			s_stage, s_end, s_wtot, s_wbase);
		// the thread's slice back into registers, as 16-byte LDS reads where the slice
		// is that long (element-wise reads at this lane stride conflict 8-way); slots
		// past n hold nothing anyone looks at
		if constexpr (ITEMS * sizeof(E) >= 16) {
			constexpr int PER = 16 / (int) sizeof(E);
			typedef E vec16 __attribute__((ext_vector_type(PER)));
			#pragma unroll
			for (int k = 0; k < ITEMS / PER; ++k) {
				const vec16 t = *reinterpret_cast<const vec16*>(&s_stage[tbase + k * PER]);
				#pragma unroll
				for (int q = 0; q < PER; ++q) key[k * PER + q] = t[q];
			}
		} else {
			#pragma unroll
			for (int i = 0; i < ITEMS; ++i) if (tbase + i < n) key[i] = s_stage[tbase + i];
		}
	}
	#pragma unroll
	for (int i = 0; i < ITEMS; ++i) if (tbase + i < n) out[tbase + i] = clo_keyx_inv<E>(key[i], kx);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

// Per pass: histogram of the combined digit -> counter scan -> pass kernel.
// A requested digit width b <= 4 pairs two digits (LB = HB = b); a wider digit
// is one pass, split in two halves.
struct rp_layout { size_t thist, toff, partial, tinfo, dig, total, tiles; };   // dig: 0 = no digit stream; tinfo: one word per tile

// (tiles of the shape clo_radix_big_tiles picks for n; `digits`: room for the digit stream
// of a multi-pass sort, n bytes, where clo_radix_digit_stream says so)
rp_layout rp_make_layout(size_t n, int elem_size, int pass_bits, bool digits = true) {
	rp_layout L;
	const size_t R2 = (size_t) 1 << pass_bits;
	const size_t tile = clo_pair_tile_elems(elem_size, clo_radix_big_tiles(n, elem_size));
	L.tiles = (n + tile - 1) / tile;
	if (L.tiles == 0) L.tiles = 1;
	const size_t per = L.tiles * R2 * sizeof(unsigned);
	L.thist = CLO_WS_HEADER_BYTES;
	L.toff = L.thist + per;
	L.partial = L.toff + per;
	L.tinfo = L.partial + ((clo_radixw_partial_rows(L.tiles) * R2 * sizeof(unsigned) + 255) & ~(size_t) 255);
	L.total = L.tinfo + ((L.tiles * sizeof(unsigned) + 255) & ~(size_t) 255);
	L.dig = 0;
	if (digits && clo_radix_digit_stream(n, elem_size)) {
		L.dig = L.total;
		L.total += (n + 255) & ~(size_t) 255;
	}
	return L;
}

// first_dig (optional): the FIRST pass's combined digit of every element, one byte each, in source order, handed
// over by whoever produced the keys (clo_hip_radix_sort_fed): the first histogram then reads n bytes instead of
// the elements — the one re-read of the keys the sort has left.
template <typename E, int LB, int HB>
int rp_sort_impl(const E* src, E* dst, E* tmp, size_t n, int key_shift, int key_bits, clo_keyx kx, const unsigned char* first_dig,
	void* ws, hipStream_t s) {
	constexpr int PB = LB + HB;   // key bits per trip through HBM
	const int passes = (key_bits + PB - 1) / PB;
	const rp_layout L = rp_make_layout(n, (int) sizeof(E), PB);
	unsigned* thist = (unsigned*) ((char*) ws + L.thist);
	unsigned* toff = (unsigned*) ((char*) ws + L.toff);
	unsigned* partial = (unsigned*) ((char*) ws + L.partial);
	unsigned* tinfo = (unsigned*) ((char*) ws + L.tinfo);
	const unsigned tiles = (unsigned) L.tiles;
	const bool big = clo_radix_big_tiles(n, (int) sizeof(E));
	const bool no_dig = clo_hip_env()->no_digits != 0;   // (tests compare both)
	// (the stream exists for the schedules of radix 16 and 256 only, LB = HB = 4: every other digit width
	// would be another set of kernels to compile for sorts nobody times)
	constexpr bool DIG_OK = LB == 4 && HB == 4 && sizeof(E) >= 4;
	unsigned char* dig = (DIG_OK && L.dig != 0 && passes > 1 && !no_dig) ? (unsigned char*) ws + L.dig : nullptr;
	const clo_keyx kx_none = { 0, 0, 0 };

	hipError_t e;   // (no kernel of the sort polls another work-group: the header's status word stays unused)

	const bool inplace_odd = (dst == src) && (passes % 2 == 1);
	const E* cur_in = src;
	for (int p = 0; p < passes; ++p) {
		E* cur_out;
		if (inplace_odd) cur_out = (p % 2 == 0) ? tmp : dst;
		else cur_out = ((passes - 1 - p) % 2 == 0) ? dst : tmp;
		const int rem = key_bits - p * PB;
		const int bits = rem < PB ? rem : PB;
		const int lo_bits = bits < LB ? bits : LB, hi_bits = bits - lo_bits;
		const unsigned shift = (unsigned) (key_shift + p * PB);
		const unsigned mask_lo = (1u << lo_bits) - 1u, mask_hi = (1u << hi_bits) - 1u;
		{
			clo_timing_scope timing("radix_hist", s);
			// (from the second pass on: out of the digit bytes the pass before left behind)
			const unsigned char* hist_bytes = (dig && p > 0) ? dig : ((DIG_OK && big && p == 0 && kx.kind == 0) ? first_dig : nullptr);
			const int st = hist_bytes
				? clo_radixw_launch_tilehist_bytes(hist_bytes, n, (int) sizeof(E), PB, (mask_hi << LB) | mask_lo, thist, tinfo, partial, tiles, big, s)
				: clo_radixw_launch_tilehist(cur_in, n, (int) sizeof(E), PB, shift, (mask_hi << LB) | mask_lo,
					thist, tinfo, partial, tiles, big, p == 0 ? kx : kx_none, s);
			if (st != 0) return st;
		}
		const unsigned* dbase = nullptr;
		{
			clo_timing_scope timing("radix_offsets", s);
			const int st = clo_radixw_launch_offsets(PB, thist, tiles, partial, toff, &dbase, s);
			if (st != 0) return st;
		}
		{
			clo_timing_scope timing("radix_pass", s);
			const int aligned = (int) ((uintptr_t) cur_in % 16 == 0);
			const clo_keyx kin = p == 0 ? kx : kx_none, kout = p + 1 == passes ? kx : kx_none;
			if constexpr (sizeof(E) >= 4) {
				if (big) {
					bool done = false;
					if constexpr (DIG_OK) {
						if (dig && p + 1 < passes) {
							hipLaunchKernelGGL((clo_radix4_pair_kernel<E, LB, HB, true, true>), dim3((tiles + 7u) / 8u * 8u), dim3(pair_shape<E, true>::THREADS), 0, s,
								cur_in, cur_out, n, shift, mask_lo, mask_hi, (const unsigned*) thist, (const unsigned*) toff, dbase, (const unsigned*) tinfo, aligned, kin, kout,
								dig, (unsigned) (key_shift + (p + 1) * PB));
							done = true;
						}
					}
					if (!done)
						hipLaunchKernelGGL((clo_radix4_pair_kernel<E, LB, HB, true, false>), dim3((tiles + 7u) / 8u * 8u), dim3(pair_shape<E, true>::THREADS), 0, s,
							cur_in, cur_out, n, shift, mask_lo, mask_hi, (const unsigned*) thist, (const unsigned*) toff, dbase, (const unsigned*) tinfo, aligned, kin, kout, nullptr, 0u);
					cur_in = cur_out;
					continue;
				}
			}
			hipLaunchKernelGGL((clo_radix4_pair_kernel<E, LB, HB, false, false>), dim3((tiles + 7u) / 8u * 8u), dim3(pair_shape<E, false>::THREADS), 0, s,
				cur_in, cur_out, n, shift, mask_lo, mask_hi, (const unsigned*) thist, (const unsigned*) toff, dbase, (const unsigned*) tinfo, aligned, kin, kout, nullptr, 0u);
		}
		cur_in = cur_out;
	}
	e = hipGetLastError();
	if (e != hipSuccess) return (int) e;
	if (inplace_odd) {
		e = hipMemcpyAsync(dst, tmp, n * sizeof(E), hipMemcpyDeviceToDevice, s);
		if (e != hipSuccess) return (int) e;
	}
	return 0;
}

template <typename E>
int rp_dispatch(const void* src, void* dst, void* tmp, size_t n, int key_shift, int key_bits, int digit_bits,
	clo_keyx kx, const unsigned char* first_dig, void* ws, hipStream_t s) {
	#define CLO_RP(LB, HB) return rp_sort_impl<E, LB, HB>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, kx, first_dig, ws, s)
	switch (digit_bits) {
		case 1: CLO_RP(1, 1);
		case 2: CLO_RP(2, 2);
		case 3: CLO_RP(3, 3);
		case 4: CLO_RP(4, 4);
		case 5: CLO_RP(3, 2);
		case 6: CLO_RP(3, 3);
		case 7: CLO_RP(4, 3);
		case 8: CLO_RP(4, 4);
		default: return CLO_HIP_EUNSUPPORTED;
	}
	#undef CLO_RP
}

// Bucket sizes of a single-pass partition, from the per-tile histograms
// (uint64 because the exchange plan adds them across ranks). R buckets, rows of R2 counters.
template <int R, int R2>
__global__ __launch_bounds__(256)
void clo_radix4_counts_kernel(const unsigned* __restrict__ thist, unsigned tiles, unsigned long long* __restrict__ counts) {
	constexpr int G = 256 / R;
	__shared__ unsigned long long s_part[G][R];
	const unsigned tid = threadIdx.x, d = tid % R, g = tid / R;
	unsigned long long c = 0;
	for (unsigned t = g; t < tiles; t += G) c += thist[(size_t) t * R2 + d];
	s_part[g][d] = c;
	__syncthreads();
	if (tid < (unsigned) R) {
		unsigned long long tot = 0;
		#pragma unroll
		for (int k = 0; k < G; ++k) tot += s_part[k][tid];
		counts[tid] = tot;
	}
}

// The same sizes out of the counter scan's digit bases (an exclusive scan of them over the digits): a difference per
// bucket. (Summing the per-tile histograms with ONE work-group — the kernel above, fine for the 2 .. 64 buckets of
// rounds 1-3, whose threads split the tiles among 256 / R groups — walks every tile in every thread when R = 256:
// 6.5 ms for 2^28 keys, ten times the partition itself.)
template <int R, int R2>
__global__ __launch_bounds__(256)
void clo_radix4_counts_from_bases_kernel(const unsigned* __restrict__ dbase, unsigned n, unsigned long long* __restrict__ counts) {
	const unsigned d = threadIdx.x;
	if (d < (unsigned) R) counts[d] = (unsigned long long) ((d + 1u < (unsigned) R2 ? dbase[d + 1u] : n) - dbase[d]);
}

// One stable pass on `BITS` bits at `shift`: the MSD bucket split of the
// multi-GPU exchange. Up to 3 bits: the pair kernel with no high digit; 4 .. 8 bits
// (the buckets of the ranks times the sub-buckets of a rank, include/clo_shard.h: 8 since
// round 4 — the split then replaces one of the sort's own passes): both local splits, LB + HB = BITS. counts (optional) receives the 1 << BITS bucket sizes.
template <typename E, int BITS, int LB, int HB>
int r4_partition_impl(const E* src, E* dst, size_t n, unsigned shift, unsigned long long* counts, void* ws, hipStream_t s) {
	constexpr unsigned R = 1u << BITS;
	constexpr int PB = LB + HB;   // width of a row of counters (BITS <= 3: LB = HB = BITS, the high half stays empty)
	constexpr bool TWO = BITS > 3;
	static_assert(TWO ? PB == BITS : (LB == BITS && HB == BITS), "see above");
	const rp_layout L = rp_make_layout(n, (int) sizeof(E), PB, false);
	unsigned* thist = (unsigned*) ((char*) ws + L.thist);
	unsigned* toff = (unsigned*) ((char*) ws + L.toff);
	unsigned* partial = (unsigned*) ((char*) ws + L.partial);
	unsigned* tinfo = (unsigned*) ((char*) ws + L.tinfo);
	const unsigned tiles = (unsigned) L.tiles;
	const clo_keyx kx_none = { 0, 0, 0 };
	hipError_t e = hipMemsetAsync(ws, 0, CLO_WS_HEADER_BYTES, s);   // (clo_hip_check_status may be asked about this workspace)
	if (e != hipSuccess) return (int) e;
	clo_timing_scope timing("msd_partition", s);
	const bool big = clo_radix_big_tiles(n, (int) sizeof(E));
	const unsigned mask_lo = TWO ? (1u << LB) - 1u : R - 1u, mask_hi = TWO ? (1u << HB) - 1u : 0u;
	int st = clo_radixw_launch_tilehist(src, n, (int) sizeof(E), PB, shift, R - 1u, thist, tinfo, partial, tiles, big, kx_none, s);
	if (st != 0) return st;
	const unsigned* dbase = nullptr;
	st = clo_radixw_launch_offsets(PB, thist, tiles, partial, toff, &dbase, s);
	if (st != 0) return st;
	if (counts) {
		if (dbase) hipLaunchKernelGGL((clo_radix4_counts_from_bases_kernel<R, (1 << PB)>), dim3(1), dim3(256), 0, s, dbase, (unsigned) n, counts);
		else hipLaunchKernelGGL((clo_radix4_counts_kernel<R, (1 << PB)>), dim3(1), dim3(256), 0, s, (const unsigned*) thist, tiles, counts);   // (one tile)
	}
	if (big)
		hipLaunchKernelGGL((clo_radix4_pair_kernel<E, LB, HB, true, false>), dim3((tiles + 7u) / 8u * 8u), dim3(pair_shape<E, true>::THREADS), 0, s,
			src, dst, n, shift, mask_lo, mask_hi, (const unsigned*) thist, (const unsigned*) toff, dbase, (const unsigned*) tinfo,
			(int) ((uintptr_t) src % 16 == 0), kx_none, kx_none, nullptr, 0u);
	else
		hipLaunchKernelGGL((clo_radix4_pair_kernel<E, LB, HB, false, false>), dim3((tiles + 7u) / 8u * 8u), dim3(pair_shape<E, false>::THREADS), 0, s,
			src, dst, n, shift, mask_lo, mask_hi, (const unsigned*) thist, (const unsigned*) toff, dbase, (const unsigned*) tinfo,
			(int) ((uintptr_t) src % 16 == 0), kx_none, kx_none, nullptr, 0u);
	return (int) hipGetLastError();
}

// ---------------------------------------------------------------------------
// Segmented sort: `nseg` consecutive segments of one array, each sorted on its own by the key bits
// [key_shift, key_shift + key_bits), in SHARED launches — the passes above with one table lookup per tile
// (which segment, where) and per counter-scan chunk. What it is for: the sub-buckets a rank of the sharded sort
// receives (clo_shard.c) share their top key bits, so sorting them one by one on the remaining bits is right
// but launch-bound (32 sorts of 2^23 keys: 12 launches each, every one too small for the chip), while one
// segmented sort of all of them runs launches of the whole slice. The radix-16 / 256 schedule only (LB = HB = 4),
// unsigned keys. The first pass reads `src` (the segments back to back, or their pieces anywhere in it) and writes
// `b`; the later ones go b -> a -> b ...: the result is in `b` after an odd number of passes, in `a` after an even
// one (*result_in_b). `src` may be `a` (it is read by the first pass only) and is otherwise left alone.
// ---------------------------------------------------------------------------
struct rp_seg_layout { size_t thist, toff, partial, tinfo, tdesc, cdesc, dig, total, tile, max_tiles, max_chunks; bool big; };

rp_seg_layout rp_make_seg_layout(size_t n, int nseg, int elem_size) {
	rp_seg_layout L;
	const size_t R2 = 256;
	L.big = clo_radix_big_tiles(n, elem_size);
	L.tile = clo_pair_tile_elems(elem_size, L.big);
	clo_radixw_seg_bounds(n, CLO_SEG_MAX, nseg, L.tile, &L.max_tiles, &L.max_chunks);   // (room for any number of pieces)
	const size_t per = L.max_tiles * R2 * sizeof(unsigned);
	const auto up = [](size_t x) { return (x + 255) & ~(size_t) 255; };
	L.thist = CLO_WS_HEADER_BYTES;
	L.toff = L.thist + per;
	L.partial = L.toff + per;
	L.tinfo = L.partial + up(clo_radixw_partial_rows_seg(L.max_chunks, (size_t) nseg) * R2 * sizeof(unsigned));
	// two pairs of tables: the first pass reads the pieces, the later ones the gathered segments
	L.tdesc = L.tinfo + up(L.max_tiles * sizeof(unsigned));
	L.cdesc = L.tdesc + 2 * up(L.max_tiles * sizeof(clo_seg_tile));
	L.total = L.cdesc + 2 * up(L.max_chunks * sizeof(clo_seg_chunk));
	L.dig = 0;
	if (clo_radix_digit_stream(n, elem_size)) {
		L.dig = L.total;
		L.total += up(n);
	}
	return L;
}

template <typename E>
int rp_sort_seg_impl(const E* src, const E* src2, E* a, E* b, size_t n, const size_t* seg_counts, int nseg,
	const size_t* piece_n, const size_t* piece_base, const int* piece_seg, const int* piece_src, int npieces, int key_shift, int key_bits, void* ws,
	hipStream_t s, int* result_in_b) {
	constexpr int LB = 4, HB = 4, PB = 8;
	const int passes = (key_bits + PB - 1) / PB;
	const rp_seg_layout L = rp_make_seg_layout(n, nseg, (int) sizeof(E));
	unsigned* thist = (unsigned*) ((char*) ws + L.thist);
	unsigned* toff = (unsigned*) ((char*) ws + L.toff);
	unsigned* partial = (unsigned*) ((char*) ws + L.partial);
	unsigned* tinfo = (unsigned*) ((char*) ws + L.tinfo);
	const auto up = [](size_t x) { return (x + 255) & ~(size_t) 255; };
	// the segments as they lie after the first pass: back to back, in order
	size_t seg_base[CLO_SEG_MAX];
	int seg_id[CLO_SEG_MAX];
	{
		size_t at = 0;
		for (int k = 0; k < nseg; ++k) { seg_base[k] = at; seg_id[k] = k; at += seg_counts[k]; }
	}
	clo_seg_tables sg0, sg;   // first pass (pieces) / later passes (segments)
	sg.tiles = (const clo_seg_tile*) ((char*) ws + L.tdesc);
	sg.chunks = (const clo_seg_chunk*) ((char*) ws + L.cdesc);
	sg.nseg = (unsigned) nseg;
	{
		clo_timing_scope timing("radix_seg_tables", s);
		int st = clo_radixw_seg_build(seg_counts, seg_base, seg_id, nullptr, nseg, nseg, L.tile, (clo_seg_tile*) sg.tiles, (clo_seg_chunk*) sg.chunks, &sg.ntiles, &sg.nchunks, s);
		if (st != 0) return st;
		sg0 = sg;
		if (npieces > 0) {
			sg0.tiles = (const clo_seg_tile*) ((char*) ws + L.tdesc + up(L.max_tiles * sizeof(clo_seg_tile)));
			sg0.chunks = (const clo_seg_chunk*) ((char*) ws + L.cdesc + up(L.max_chunks * sizeof(clo_seg_chunk)));
			st = clo_radixw_seg_build(piece_n, piece_base, piece_seg, piece_src, npieces, nseg, L.tile, (clo_seg_tile*) sg0.tiles, (clo_seg_chunk*) sg0.chunks, &sg0.ntiles, &sg0.nchunks, s);
			if (st != 0) return st;
		}
	}
	*result_in_b = passes % 2;
	if (sg.ntiles == 0) return 0;
	if (sg.ntiles > L.max_tiles || sg.nchunks > L.max_chunks || sg0.ntiles > L.max_tiles || sg0.nchunks > L.max_chunks)
		return CLO_HIP_EWORKSPACE;   // (cannot happen: the bounds hold for any split of n)
	unsigned char* dig = (L.dig != 0 && passes > 1) ? (unsigned char*) ws + L.dig : nullptr;
	const clo_keyx kx_none = { 0, 0, 0 };
	const E* cur_in = src;
	for (int p = 0; p < passes; ++p) {
		const clo_seg_tables& sgp = p == 0 ? sg0 : sg;
		const unsigned grid = (sgp.ntiles + 7u) / 8u * 8u;
		E* cur_out = p % 2 == 0 ? b : a;
		const int rem = key_bits - p * PB;
		const int bits = rem < PB ? rem : PB;
		const int lo_bits = bits < LB ? bits : LB, hi_bits = bits - lo_bits;
		const unsigned shift = (unsigned) (key_shift + p * PB);
		const unsigned mask_lo = (1u << lo_bits) - 1u, mask_hi = (1u << hi_bits) - 1u;
		{
			clo_timing_scope timing("radix_hist", s);
			const int st = (dig && p > 0)
				? clo_radixw_launch_tilehist_bytes_seg(dig, sgp, (int) sizeof(E), PB, (mask_hi << LB) | mask_lo, thist, tinfo, partial, L.big, s)
				: clo_radixw_launch_tilehist_seg(cur_in, p == 0 ? src2 : nullptr, sgp, (int) sizeof(E), PB, shift, (mask_hi << LB) | mask_lo, thist, tinfo, partial, L.big, s);
			if (st != 0) return st;
		}
		const unsigned* dbase = nullptr;
		{
			clo_timing_scope timing("radix_offsets", s);
			const int st = clo_radixw_launch_offsets_seg(PB, thist, sgp, partial, toff, &dbase, s);
			if (st != 0) return st;
		}
		{
			clo_timing_scope timing("radix_pass", s);
			const unsigned next_shift = (unsigned) (key_shift + (p + 1) * PB);
			if (L.big) {
				if (dig && p + 1 < passes)
					hipLaunchKernelGGL((clo_radix4_pair_kernel<E, LB, HB, true, true, true>), dim3(grid), dim3(pair_shape<E, true>::THREADS), 0, s,
						(const E*) cur_in, cur_out, (size_t) sgp.ntiles, shift, mask_lo, mask_hi, (const unsigned*) thist, (const unsigned*) toff, dbase, (const unsigned*) tinfo, 0,
						kx_none, kx_none, dig, next_shift, sgp.tiles, p == 0 ? src2 : (const E*) nullptr);
				else
					hipLaunchKernelGGL((clo_radix4_pair_kernel<E, LB, HB, true, false, true>), dim3(grid), dim3(pair_shape<E, true>::THREADS), 0, s,
						(const E*) cur_in, cur_out, (size_t) sgp.ntiles, shift, mask_lo, mask_hi, (const unsigned*) thist, (const unsigned*) toff, dbase, (const unsigned*) tinfo, 0,
						kx_none, kx_none, nullptr, 0u, sgp.tiles, p == 0 ? src2 : (const E*) nullptr);
			} else {
				hipLaunchKernelGGL((clo_radix4_pair_kernel<E, LB, HB, false, false, true>), dim3(grid), dim3(pair_shape<E, false>::THREADS), 0, s,
					(const E*) cur_in, cur_out, (size_t) sgp.ntiles, shift, mask_lo, mask_hi, (const unsigned*) thist, (const unsigned*) toff, dbase, (const unsigned*) tinfo, 0,
					kx_none, kx_none, nullptr, 0u, sgp.tiles, p == 0 ? src2 : (const E*) nullptr);
			}
		}
		cur_in = cur_out;
	}
	return (int) hipGetLastError();
}

template <typename E>
int small_dispatch(const void* src, void* dst, size_t n, int key_shift, int key_bits, int digit_bits, clo_keyx kx, hipStream_t s) {
	clo_timing_scope timing("radix_small", s);
	#define CLO_SMALL(B) case B: \
		if (n <= (size_t) SMALL_TILE) \
			hipLaunchKernelGGL((clo_radix4_small_kernel<E, B, 512, 8>), dim3(1), dim3(512), 0, s, \
				(const E*) src, (E*) dst, (unsigned) n, (unsigned) key_shift, (unsigned) key_bits, kx); \
		else if (n <= (size_t) 1024 * 8) \
			hipLaunchKernelGGL((clo_radix4_small_kernel<E, B, 1024, 8>), dim3(1), dim3(1024), 0, s, \
				(const E*) src, (E*) dst, (unsigned) n, (unsigned) key_shift, (unsigned) key_bits, kx); \
		else \
			hipLaunchKernelGGL((clo_radix4_small_kernel<E, B, small_big<E>::THREADS, small_big<E>::ITEMS>), dim3(1), dim3(small_big<E>::THREADS), 0, s, \
				(const E*) src, (E*) dst, (unsigned) n, (unsigned) key_shift, (unsigned) key_bits, kx); \
		break
	switch (digit_bits) {
		CLO_SMALL(1); CLO_SMALL(2); CLO_SMALL(3); CLO_SMALL(4);
		default: return CLO_HIP_EUNSUPPORTED;
	}
	#undef CLO_SMALL
	return (int) hipGetLastError();
}

}  // namespace

static size_t r4_pair_workspace_bytes(size_t n, int elem_size, int digit_bits) {
	return (rp_make_layout(n, elem_size, digit_bits <= 4 ? 2 * digit_bits : digit_bits).total + 255) & ~(size_t) 255;
}

size_t clo_radix4_workspace_bytes(size_t n, int elem_size, int digit_bits, int key_bits) {
	// [pair passes' counters][single-sweep passes' state, when that path may take the sort]
	size_t b = r4_pair_workspace_bytes(n, elem_size, digit_bits);
	if (clo_radix1_applies(n, elem_size, digit_bits)) b += clo_radix1_workspace_bytes(n, elem_size, key_bits);
	return b;
}

size_t clo_radix4_seg_workspace_bytes(size_t n, int nseg, int elem_size, int digit_bits) {
	if ((digit_bits != 4 && digit_bits != 8) || (elem_size != 4 && elem_size != 8) || nseg < 1 || nseg > CLO_SEG_MAX) return 0;
	return rp_make_seg_layout(n, nseg, elem_size).total;
}

int clo_radix4_sort_segmented(const void* src, const void* src2, void* a, void* b, size_t n, const size_t* seg_counts, int nseg,
	const size_t* piece_n, const size_t* piece_base, const int* piece_seg, const int* piece_src, int npieces, int elem_size, int key_shift,
	int key_bits, int digit_bits, void* ws, hipStream_t s, int* result_in_b) {
	if (digit_bits != 4 && digit_bits != 8) return CLO_HIP_EUNSUPPORTED;
	if (elem_size == 4) return rp_sort_seg_impl<uint32_t>((const uint32_t*) src, (const uint32_t*) src2, (uint32_t*) a, (uint32_t*) b, n, seg_counts, nseg, piece_n, piece_base, piece_seg, piece_src, npieces, key_shift, key_bits, ws, s, result_in_b);
	if (elem_size == 8) return rp_sort_seg_impl<uint64_t>((const uint64_t*) src, (const uint64_t*) src2, (uint64_t*) a, (uint64_t*) b, n, seg_counts, nseg, piece_n, piece_base, piece_seg, piece_src, npieces, key_shift, key_bits, ws, s, result_in_b);
	return CLO_HIP_EUNSUPPORTED;
}

size_t clo_radix4_partition_workspace_bytes(size_t n, int elem_size, int bits) {
	return rp_make_layout(n, elem_size, bits <= 3 ? 2 * bits : bits, false).total;
}

int clo_radix4_partition(const void* src, void* dst, size_t n, int elem_size, unsigned shift, int bits,
	unsigned long long* counts, void* ws, hipStream_t s) {
	#define CLO_R4P(E, B, LB, HB) return r4_partition_impl<E, B, LB, HB>((const E*) src, (E*) dst, n, shift, counts, ws, s)
	if (elem_size == 4) {
		if (bits == 1) CLO_R4P(uint32_t, 1, 1, 1);
		if (bits == 2) CLO_R4P(uint32_t, 2, 2, 2);
		if (bits == 3) CLO_R4P(uint32_t, 3, 3, 3);
		if (bits == 4) CLO_R4P(uint32_t, 4, 2, 2);
		if (bits == 5) CLO_R4P(uint32_t, 5, 3, 2);
		if (bits == 6) CLO_R4P(uint32_t, 6, 3, 3);
		if (bits == 7) CLO_R4P(uint32_t, 7, 4, 3);
		if (bits == 8) CLO_R4P(uint32_t, 8, 4, 4);
	} else if (elem_size == 8) {
		if (bits == 1) CLO_R4P(uint64_t, 1, 1, 1);
		if (bits == 2) CLO_R4P(uint64_t, 2, 2, 2);
		if (bits == 3) CLO_R4P(uint64_t, 3, 3, 3);
		if (bits == 4) CLO_R4P(uint64_t, 4, 2, 2);
		if (bits == 5) CLO_R4P(uint64_t, 5, 3, 2);
		if (bits == 6) CLO_R4P(uint64_t, 6, 3, 3);
		if (bits == 7) CLO_R4P(uint64_t, 7, 4, 3);
		if (bits == 8) CLO_R4P(uint64_t, 8, 4, 4);
	}
	#undef CLO_R4P
	return CLO_HIP_EUNSUPPORTED;
}

// static LDS of the pass kernel (introspection: clo_sort_get_localmem_usage)
size_t clo_radix4_lds_bytes(int elem_size, int digit_bits) {
	const size_t threads = 512;   // (the shape of arrays below 256 MiB; larger ones: 1024 threads, table inside a 64 KiB stage)
	const int half = digit_bits <= 4 ? digit_bits : (digit_bits + 1) / 2;   // the wider of the two local digits
	const size_t hmax = half >= 4 ? 8 : (half == 3 ? 4 : (half == 2 ? 2 : 1));
	const size_t pass_bits = digit_bits <= 4 ? 2 * digit_bits : digit_bits;
	const size_t items = elem_size == 8 ? 8 : 16;
	return threads * items * (size_t) elem_size + threads * PC_END_STRIDE * sizeof(unsigned)
		+ (2 * (threads / 64) * hmax + ((size_t) 1 << pass_bits) + 4) * sizeof(unsigned);
}

// 1: a sort of this shape reads caller-provided first digits (the chain-free passes on big tiles, radix 16 / 256)
int clo_radix4_takes_first_digits(size_t n, int elem_size, int digit_bits) {
	if (elem_size < 4 || (digit_bits != 4 && digit_bits != 8) || clo_hip_env()->no_digits) return 0;
	if (clo_radix1_applies(n, elem_size, digit_bits)) return 0;
	return clo_radix_big_tiles(n, elem_size) ? 1 : 0;
}

int clo_radix4_sort(const void* src, void* dst, void* tmp, size_t n, int elem_size, int key_shift,
	int key_bits, int digit_bits, clo_keyx kx, const unsigned char* first_dig, void* ws, hipStream_t s) {
	const size_t small_max = elem_size == 8 ? (size_t) small_big<uint64_t>::TILE : (size_t) small_big<uint32_t>::TILE;
	if (n <= small_max && digit_bits <= 4) {   // one launch for the whole sort
		switch (elem_size) {
			case 1: return small_dispatch<uint8_t>(src, dst, n, key_shift, key_bits, digit_bits, kx, s);
			case 2: return small_dispatch<uint16_t>(src, dst, n, key_shift, key_bits, digit_bits, kx, s);
			case 4: return small_dispatch<uint32_t>(src, dst, n, key_shift, key_bits, digit_bits, kx, s);
			case 8: return small_dispatch<uint64_t>(src, dst, n, key_shift, key_bits, digit_bits, kx, s);
			default: return CLO_HIP_EUNSUPPORTED;
		}
	}
	if (clo_radix1_applies(n, elem_size, digit_bits))   // every element read once per pass (clo_hip_radix1.hip)
		return clo_radix1_sort(src, dst, tmp, n, elem_size, key_shift, key_bits, kx,
			(char*) ws + r4_pair_workspace_bytes(n, elem_size, digit_bits), (unsigned*) ws, s);
	switch (elem_size) {
		case 1: return rp_dispatch<uint8_t>(src, dst, tmp, n, key_shift, key_bits, digit_bits, kx, first_dig, ws, s);
		case 2: return rp_dispatch<uint16_t>(src, dst, tmp, n, key_shift, key_bits, digit_bits, kx, first_dig, ws, s);
		case 4: return rp_dispatch<uint32_t>(src, dst, tmp, n, key_shift, key_bits, digit_bits, kx, first_dig, ws, s);
		case 8: return rp_dispatch<uint64_t>(src, dst, tmp, n, key_shift, key_bits, digit_bits, kx, first_dig, ws, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}
