// clo_hip_radix_rank.h — ranking with thread-private packed counters and the
// stable local split of a tile built on it: shared by the radix pass kernels
// (clo_hip_radix4.hip: chain-free pair passes; clo_hip_radix1.hip: single-sweep
// passes). Device code only; every function is a template or forced inline.
#ifndef CLO_HIP_RADIX_RANK_H
#define CLO_HIP_RADIX_RANK_H

#include <hip/hip_runtime.h>

#include <cstdlib>

#include "clo_hip_internal.h"

namespace {

// (clo_lds_barrier — the work-group barrier that orders LDS traffic only — lives in clo_hip_internal.h)

// ---------------------------------------------------------------------------
// Ranking with thread-private packed counters.
//
// Ranking by wave-wide match-any costs ~40 VALU instructions per element (one
// ballot and a 64-bit select/and per digit bit); a pass built on it measured
// VALU-bound (SQ_INSTS_VALU = 85 per element). Here each thread owns ITEMS = 8
// CONSECUTIVE elements and counts digits in thread-private packed counters
// (16 digits x 4 bits in one 64-bit register), which also yield each element's
// rank among the thread's own elements. One wave64 DPP scan of the widened
// counters plus a cross-wave step through LDS gives, per thread, the count of
// every digit among all earlier threads of the tile. Thread order = element
// order, so the ranking is stable.
// ---------------------------------------------------------------------------

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_add(unsigned x) {
	return x + (unsigned) __builtin_amdgcn_update_dpp(0, (int) x, CTRL, ROW_MASK, 0xF, true);
}

constexpr int PC_BATCH = 4;        // lookups of ends requested together in a full tile
constexpr int PC_END_STRIDE = 8;   // dwords per thread in the table of ends: 16 digits x 16 bits (digit-major, see pc_local_split)

template <int BITS> struct pc_words { static constexpr int H = (1 << BITS) >= 2 ? (1 << BITS) / 2 : 1; };

template <typename E, int ITEMS>
__device__ __forceinline__ void load_blocked(const E* __restrict__ p, E (&key)[ITEMS], bool aligned) {
	if (aligned) {
		typedef E vecN __attribute__((ext_vector_type(ITEMS)));
		const vecN v = *reinterpret_cast<const vecN*>(p);
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) key[i] = v[i];
	} else {
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) key[i] = p[i];
	}
}

// The same slice at an address that is only element-aligned: 16-byte loads all the same (the hardware takes them;
// the type says so to the compiler).
template <typename E, int ITEMS>
__device__ __forceinline__ void load_blocked_unaligned(const E* __restrict__ p, E (&key)[ITEMS]) {
	constexpr int PER = 16 / (int) sizeof(E);
	typedef E vecU __attribute__((ext_vector_type(PER), aligned(sizeof(E))));
	#pragma unroll
	for (int k = 0; k < ITEMS / PER; ++k) {
		const vecU v = reinterpret_cast<const vecU*>(p)[k];
		#pragma unroll
		for (int q = 0; q < PER; ++q) key[k * PER + q] = v[q];
	}
}

// The thread-private 4-bit counters first widen to 8-bit fields only (4
// digits per VGPR: even digits in one word, odd digits in the next): an
// inclusive scan inside a row of 16 lanes cannot exceed 16 * 15 = 240. Only
// then do they widen to 16-bit fields for the two cross-row steps; v_perm_b32
// builds w[j] = count(2j) | count(2j+1) << 16 from one even and one odd word.
// TWO: the thread counted in two packed counters (up to 16 elements): their
// 8-bit images are added first, and since 16 lanes * 16 could reach 256 the
// last in-row step (row_shr:8) runs on the 16-bit fields.
template <int BITS, bool TWO>
__device__ __forceinline__ void pc2_wave_scan(unsigned long long c, unsigned long long c2, unsigned (&w)[pc_words<BITS>::H]) {
	constexpr int H = pc_words<BITS>::H;
	constexpr int NB = (1 << BITS) == 16 ? 4 : 2;   // words of 8-bit fields
	unsigned b[NB];
	const unsigned lo = (unsigned) c, hi = (unsigned) (c >> 32);
	b[0] = lo & 0x0f0f0f0fu;          // digits 0,2,4,6
	b[1] = (lo >> 4) & 0x0f0f0f0fu;   // digits 1,3,5,7
	if constexpr (NB == 4) {
		b[2] = hi & 0x0f0f0f0fu;
		b[3] = (hi >> 4) & 0x0f0f0f0fu;
	}
	if constexpr (TWO) {
		const unsigned lo2 = (unsigned) c2, hi2 = (unsigned) (c2 >> 32);
		b[0] += lo2 & 0x0f0f0f0fu;
		b[1] += (lo2 >> 4) & 0x0f0f0f0fu;
		if constexpr (NB == 4) {
			b[2] += hi2 & 0x0f0f0f0fu;
			b[3] += (hi2 >> 4) & 0x0f0f0f0fu;
		}
	}
	#pragma unroll
	for (int k = 0; k < NB; ++k) {
		unsigned x = b[k];
		x = dpp_add<0x111, 0xF>(x);
		x = dpp_add<0x112, 0xF>(x);
		x = dpp_add<0x114, 0xF>(x);
		b[k] = TWO ? x : dpp_add<0x118, 0xF>(x);
	}
	#pragma unroll
	for (int j = 0; j < H; ++j) {
		// byte (j & 3) of the even word -> bits 0..15, of the odd word -> bits 16..31
		const unsigned sel = 0x0c040c00u + (unsigned) (j & 3) * 0x00010001u;
		unsigned x = __builtin_amdgcn_perm(b[(j >> 2) * 2 + 1], b[(j >> 2) * 2], sel);
		if constexpr (TWO) x = dpp_add<0x118, 0xF>(x);
		x = dpp_add<0x142, 0xA>(x);
		w[j] = dpp_add<0x143, 0xC>(x);
	}
}

// Byte K of a register: written from the low byte of x / subtracted from a — one SDWA
// instruction each (in C the first is a shift, a mask and an or; the second an
// extract and a subtract). K is a constant after unrolling.
__device__ __forceinline__ void pc_put_byte(unsigned& dst, unsigned x, int K) {
	switch (K) {
		case 0: asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(dst) : "v"(x)); break;
		case 1: asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(dst) : "v"(x)); break;
		case 2: asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(dst) : "v"(x)); break;
		default: asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0" : "+v"(dst) : "v"(x)); break;
	}
}
// (a - byte K of b) mod 2^16, zero-extended: one SDWA subtract whose destination is the low word (the high one padded with zeros).
__device__ __forceinline__ unsigned pc_sub_byte16(unsigned a, unsigned b, int K) {
	unsigned r;
	switch (K) {
		case 0: asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(a), "v"(b)); break;
		case 1: asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(a), "v"(b)); break;
		case 2: asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(a), "v"(b)); break;
		default: asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(a), "v"(b)); break;
	}
	return r;
}
// Both 16-bit halves of a word at once, each mod 2^16 (no carry from the low half into the high one): a * S + b, a * S.
typedef unsigned short pc_u16x2 __attribute__((ext_vector_type(2)));
template <int S>
__device__ __forceinline__ unsigned pc_pk_scale(unsigned a) {
	if constexpr (S == 1) return a;
	const pc_u16x2 r = __builtin_bit_cast(pc_u16x2, a) * (unsigned short) S;
	return __builtin_bit_cast(unsigned, r);
}
template <int S>
__device__ __forceinline__ unsigned pc_pk_scale_add(unsigned a, unsigned b) {
	unsigned r;   // (one v_pk_mad_u16; from C the compiler makes a packed shift and a packed add of it)
	if constexpr (S == 1) asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	else asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"((unsigned) S * 0x00010001u), "v"(b));
	return r;
}
// (x & 0x0f0f0f0f) * S + S * 0x01010101 for S = 1, 2, 4, 8: an and, and one shift-and-add
template <int S>
__device__ __forceinline__ unsigned pc_rank_bytes(unsigned x) {
	x &= 0x0f0f0f0fu;
	if constexpr (S == 1) {
		return x + 0x01010101u;
	} else {
		unsigned r;
		constexpr int SH = S == 2 ? 1 : (S == 4 ? 2 : 3);
		static_assert(S == 2 || S == 4 || S == 8, "element sizes 1, 2, 4, 8");
		asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "n"(SH), "s"((unsigned) S * 0x01010101u));
		return r;
	}
}
// The digit of a key: one bit-field extract for elements of up to 4 bytes
// (nbits = the number of bits of the digit's mask, the same for the whole launch).
template <typename E>
__device__ __forceinline__ unsigned pc_digit(E key, unsigned dshift, unsigned dmask, unsigned nbits) {
	if constexpr (sizeof(E) <= 4) return __builtin_amdgcn_ubfe((unsigned) key, dshift, nbits);
	else return (unsigned) (key >> dshift) & dmask;
}

// One stable local split of the tile by the digit (key >> dshift) & dmask;
// on return (after a barrier) s_stage holds the tile in digit order. The
// thread's elements are ITEMS consecutive positions of the tile.
// `mid`: work placed between the first two barriers, where only wave 0 is busy
// (it turns the wave totals into wave bases) — every thread calls it, it picks
// its own waves; it must not touch s_wtot / s_wbase / s_end.
// `counted`: called by every thread right after it has counted its own elements,
// with the two packed counters (4 bits per digit; c2 only for ITEMS == 16).
struct pc_no_mid { __device__ __forceinline__ void operator()() const {} };
struct pc_no_counted { __device__ __forceinline__ void operator()(unsigned long long, unsigned long long) const {} };

// FULL: every element of the tile exists. The two loops over the thread's elements
// then carry no per-element test: compiled with one (`full || tbase + i < count`),
// every element sat in a basic block of its own — a compare, an exec-mask dance and
// a scalar branch each, and the lookup of its end waited for (`s_waitcnt lgkmcnt(0)`)
// before the next element's was even requested.
template <typename E, int BITS, int THREADS, int ITEMS, int HMAX, bool FULL, bool ALIAS, typename Mid, typename Counted>
__device__ __forceinline__ void pc_local_split_impl(const E (&key)[ITEMS], unsigned dshift, unsigned dmask, unsigned count,
	E* s_stage, unsigned* s_end, unsigned (*s_wtot)[HMAX], unsigned (*s_wbase)[HMAX], Mid mid, Counted counted) {
	constexpr int H = pc_words<BITS>::H;
	constexpr int WAVES = THREADS / 64;
	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const unsigned tbase = tid * ITEMS;

	// Thread-private counts, LAST element first: `before` = the number of LATER
	// elements of the thread with the same digit (<= 15), so that the element's
	// position is (end of the thread's slice of that digit) - before - 1; kept as
	// bytes, four to a register. More than 8 elements per thread: two counters for
	// the wave scan (a 4-bit field holds up to 15): elements 15..8 count in c2; the
	// running counter then starts from c2, so one shift reads an element's `before`
	// (its last increment may carry out of a field: it is never read again), and
	// c = run - c2 (exact: integer addition) is the count of elements 7..0 alone.
	static_assert(ITEMS == 8 || ITEMS == 16, "one or two packed counters");
	unsigned long long run = 0, c2 = 0;
	unsigned rb[ITEMS / 4];
	const unsigned nbits = (unsigned) __builtin_popcount(dmask);
	#pragma unroll
	for (int k = 0; k < ITEMS / 4; ++k) rb[k] = 0;
	#pragma unroll
	for (int i = ITEMS - 1; i >= 0; --i) {
		if (ITEMS == 16 && i == 7) c2 = run;
		if (FULL || tbase + i < count) {
			const unsigned sh = pc_digit<E>(key[i], dshift, dmask, nbits) * 4u;
			pc_put_byte(rb[i >> 2], (unsigned) (run >> sh), i & 3);   // (the byte's high half is the next field: masked below)
			run += 1ull << sh;
		}
	}
	const unsigned long long c = run - c2;
	// Positions are kept as BYTE offsets into the stage, in 16 bits: the table of ends holds end * sizeof(E) mod 2^16
	// (a tile is at most 64 KiB: only an end at the very end of a 64 KiB tile wraps, to 0), a rank byte holds
	// (before + 1) * sizeof(E) <= 128, and one SDWA subtract into the low word gives the element's address —
	// (end - before - 1) * sizeof(E) mod 2^16, always inside the stage — with no decrement, shift or mask per element
	// (round 5: two VALU instructions per element and split fewer).
	constexpr unsigned ES = (unsigned) sizeof(E);
	static_assert((size_t) THREADS * ITEMS * sizeof(E) <= 65536, "byte offsets of the stage fit in 16 bits");
	#pragma unroll
	for (int k = 0; k < ITEMS / 4; ++k) rb[k] = pc_rank_bytes<(int) ES>(rb[k]);
	counted(c, c2);
	unsigned w[H];
	pc2_wave_scan<BITS, (ITEMS > 8)>(c, c2, w);
	if (lane == 63) {
		#pragma unroll
		for (int j = 0; j < H; ++j) s_wtot[wave][j] = w[j];
	}
	clo_lds_barrier();
	if (wave == 0) {
		// Wave 0 turns the wave totals into every wave's base, per digit: lane
		// (q, wv) = (lane / 16, lane % 16) takes word 4r + q of wave wv; a DPP scan
		// inside rows of 16 lanes runs over the waves, lane 15 of a row ends up with
		// the digit totals of its word, and a 16-step serial prefix over those
		// (wave-uniform values) gives the digit starts. One barrier later every
		// thread has its bases.
		static_assert(WAVES <= 16, "one row of 16 lanes spans the waves");

		constexpr int ROUNDS = (H + 3) / 4;
		const unsigned wv = lane & 15u, q = lane >> 4;
		unsigned excl[ROUNDS], tot[ROUNDS];
		#pragma unroll
		for (int r = 0; r < ROUNDS; ++r) {
			const unsigned j = r * 4 + q;
			const unsigned x = (j < (unsigned) H && wv < (unsigned) WAVES) ? s_wtot[wv][j] : 0u;
			unsigned incl = dpp_add<0x111, 0xF>(x);
			incl = dpp_add<0x112, 0xF>(incl);
			incl = dpp_add<0x114, 0xF>(incl);
			incl = dpp_add<0x118, 0xF>(incl);
			excl[r] = incl - x;
			tot[r] = incl;
		}
		unsigned run = 0, dstart16[H];
		#pragma unroll
		for (int j = 0; j < H; ++j) {
			const unsigned t = (unsigned) __builtin_amdgcn_readlane((int) tot[j / 4], (j % 4) * 16 + 15);   // packed totals of digits 2j, 2j+1 (wave-uniform: scalar registers)
			dstart16[j] = run | ((run + (t & 0xffffu)) << 16);
			run += (t & 0xffffu) + (t >> 16);
		}
		#pragma unroll
		for (int r = 0; r < ROUNDS; ++r) {
			unsigned mine = dstart16[r * 4];
			#pragma unroll
			for (int k = 1; k < 4; ++k) if (r * 4 + k < H && q == (unsigned) k) mine = dstart16[r * 4 + k];
			const unsigned j = r * 4 + q;
			if (j < (unsigned) H && wv < (unsigned) WAVES) s_wbase[wv][j] = pc_pk_scale<(int) ES>(mine + excl[r]);   // (bytes, see above)
		}
	}
	mid();
	clo_lds_barrier();
	// The table of ends: 16-bit entries, DIGIT-major, one row of THREADS entries per
	// digit; inside a row a wave's lanes l and l + 32 share a dword. The bank of an
	// entry then depends on the lane alone (dword index = digit * THREADS / 2 + wave * 32
	// + l % 32), so neither the 16 writes nor the lookup of an element's end ever
	// conflict, whatever the digits are. (Thread-major rows of 9 dwords — the first
	// layout — put lanes with different digits on one bank: with the scattered stage
	// writes, half of the kernel's LDS cycles were bank conflicts, SQ_LDS_BANK_CONFLICT.)
	typedef unsigned short __attribute__((may_alias)) pc_u16;   // (the array is declared as dwords)
	pc_u16* const tab = reinterpret_cast<pc_u16*>(s_end) + ((tid & ~63u) + ((tid & 31u) << 1) + ((tid >> 5) & 1u));
	unsigned wb[H];   // the wave's bases (16-byte LDS reads where the row is that long; s_wbase is declared 16-byte aligned)
	if constexpr (H % 4 == 0 && HMAX % 4 == 0) {
		typedef unsigned vec4u __attribute__((ext_vector_type(4)));
		#pragma unroll
		for (int k = 0; k < H / 4; ++k) {
			const vec4u t = reinterpret_cast<const vec4u*>(s_wbase[wave])[k];
			wb[4 * k] = t[0]; wb[4 * k + 1] = t[1]; wb[4 * k + 2] = t[2]; wb[4 * k + 3] = t[3];
		}
	} else {
		#pragma unroll
		for (int j = 0; j < H; ++j) wb[j] = s_wbase[wave][j];
	}
	#pragma unroll
	for (int j = 0; j < H; ++j) {
		const unsigned e2 = pc_pk_scale_add<(int) ES>(w[j], wb[j]);   // ends of digits 2j, 2j + 1 in bytes, each mod 2^16
		tab[(2 * j) * THREADS] = (unsigned short) e2;
		if (2 * j + 1 < (1 << BITS)) tab[(2 * j + 1) * THREADS] = (unsigned short) (e2 >> 16);
	}
	unsigned pos[ALIAS ? ITEMS : 1];
	char* const stage_bytes = reinterpret_cast<char*>(s_stage);
	#pragma unroll
	for (int i = 0; i < ITEMS; ++i) {
		// (lookups in flight: PC_BATCH at a time — all ITEMS at once cost 2 * ITEMS registers)
		if (FULL && i > 0 && i % PC_BATCH == 0) __builtin_amdgcn_sched_barrier(0);
		if (FULL || tbase + i < count) {
			const unsigned d = pc_digit<E>(key[i], dshift, dmask, nbits);
			const unsigned end = tab[d * THREADS];
			const unsigned at = pc_sub_byte16(end, rb[i >> 2], i & 3);   // (end - before - 1) * sizeof(E): counts of these very elements, always inside the tile
			if constexpr (ALIAS) pos[i] = at;   // stored after the barrier below
			else *reinterpret_cast<E*>(stage_bytes + at) = key[i];
		}
	}
	if constexpr (ALIAS) {
		// the table lives in the head of the stage (thread-private, dead from here on)
		clo_lds_barrier();
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i)
			if (FULL || tbase + i < count) *reinterpret_cast<E*>(stage_bytes + pos[i]) = key[i];
	}
	clo_lds_barrier();
}

// ALIAS: the table of ends is built in the head of the stage itself (s_end is then
// ignored; positions are looked up into registers, one more barrier, then the
// stage is written) — for tiles whose stage and table do not both fit in LDS.
// The stage must not be read by anyone between the call's first barrier and its end.
template <typename E, int BITS, int THREADS, int ITEMS, int HMAX, typename Mid = pc_no_mid, typename Counted = pc_no_counted, bool ALIAS = false>
__device__ __forceinline__ void pc_local_split(const E (&key)[ITEMS], unsigned dshift, unsigned dmask, unsigned count,
	E* s_stage, unsigned* s_end, unsigned (*s_wtot)[HMAX], unsigned (*s_wbase)[HMAX], Mid mid = Mid(), Counted counted = Counted()) {
	static_assert(!ALIAS || (size_t) THREADS * PC_END_STRIDE * 4 <= (size_t) THREADS * ITEMS * sizeof(E), "the table fits in the stage");
	unsigned* const tab = ALIAS ? reinterpret_cast<unsigned*>(s_stage) : s_end;
	if (count == (unsigned) (THREADS * ITEMS))   // (the same for the whole work-group)
		pc_local_split_impl<E, BITS, THREADS, ITEMS, HMAX, true, ALIAS>(key, dshift, dmask, count, s_stage, tab, s_wtot, s_wbase, mid, counted);
	else
		pc_local_split_impl<E, BITS, THREADS, ITEMS, HMAX, false, ALIAS>(key, dshift, dmask, count, s_stage, tab, s_wtot, s_wbase, mid, counted);
}

// XCD the wave runs on (HW_REG_XCC_ID, bits 3..0), 0..7. The sweep kernel's eight ticket pools rely
// on it for forward progress, not only for speed, and are restricted accordingly (clo_hip_radix1.hip, head).
__device__ __forceinline__ unsigned clo_xcc_id() {
	return (unsigned) __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;
}

// Shape of a tile of the single-sweep pass kernels: 512 threads x 16 consecutive
// elements (8 for 8-byte elements) = 32 KiB of LDS stage either way.
template <typename E> struct sweep_shape {
	static constexpr int THREADS = 512;
	static constexpr int ITEMS = sizeof(E) == 8 ? 8 : 16;
};

// Shape of a tile of the chain-free pair passes (= the tiles of clo_hip_radixw.hip's
// histogram). Two shapes, chosen per sort by the array's size (clo_radix_big_tiles):
// 512 threads x 16 elements (8 for 8-byte elements) with the table of ends next to
// the 32 KiB stage, 3 work-groups per CU; and, for arrays of 256 MiB and more (32 MiB
// of 8-byte elements), BIG:
// 1024 threads on twice the tile, the table inside the 64 KiB stage (ALIAS), 2
// work-groups per CU — runs of 256 bytes per digit pair and tile instead of 128 in
// the scatter, half the counters: pair kernel 0.475 -> 0.443 ms per pass at 2^28
// uint32, uint64 0.887 -> 0.820 (docs/lab_notebook.md).
template <typename E, bool BIG> struct pair_shape {
	static constexpr int THREADS = (BIG && sizeof(E) >= 4) ? 1024 : 512;
	static constexpr int ITEMS = sizeof(E) == 8 ? 8 : 16;
	static constexpr int TILE = THREADS * ITEMS;
	static constexpr bool ALIAS = BIG && sizeof(E) >= 4;
};
// (8-byte elements gain from 32 MiB on — their digit stream is an eighth of the
// array; 2^22 uint64 0.254 -> 0.234 ms, pairs 0.127 -> 0.119, profiles/r03_big_tile_threshold.txt —
// 4-byte ones from 256 MiB: measured, docs/lab_notebook.md)
inline size_t clo_big_tile_bytes(int elem_size) { return (size_t) (elem_size == 8 ? 32 : 256) << 20; }
inline bool clo_radix_big_tiles(size_t n, int elem_size) { return elem_size >= 4 && n * (size_t) elem_size >= clo_big_tile_bytes(elem_size); }
// The digit stream (one byte per element between two passes) goes with the
// big tiles: on 8 192-element tiles (arrays that sit in the last-level cache) its extra
// writes cost more than the histogram's shorter read saves (2^25 uint32: 0.471 vs 0.421 ms in round 2;
// re-measured in round 3 after the counter scan became one launch: 2^22 0.113 vs 0.108, 2^24 0.224 vs 0.222,
// 2^25 0.440 vs 0.389 — runs of 32 digit bytes per digit pair and tile are sub-line writes).
inline bool clo_radix_digit_stream(size_t n, int elem_size) { return clo_radix_big_tiles(n, elem_size); }
__host__ __device__ inline size_t clo_pair_tile_elems(int elem_size, bool big) {
	return (size_t) ((big && elem_size >= 4) ? 1024 : 512) * (elem_size == 8 ? 8 : 16);
}

}  // namespace

#endif
