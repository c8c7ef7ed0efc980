// clo_hip_radix4.hip — LSD radix sort passes for small radices (digits of 1..4
// bits, radix <= 16): the default "satradix" configuration (radix = 16).
//
// Structure per digit, the reference's own (sort/clo_sort_satradix.c:264-313):
//   per-tile digit histogram -> scan of the counters -> scatter,
// with two changes that remove every redundant pass over the keys:
//   * the tile-local sort + scatter of a digit is ONE kernel that reads each
//     element once and writes it once (clo_radix4_pass_pc_kernel);
//   * the per-tile histogram of the NEXT digit is accumulated by that same
//     kernel while it scatters: an element's destination index, hence its tile
//     in the next pass, is known when it is stored. Elements of one (tile,
//     digit) run land in at most two destination tiles, so a 2 KiB LDS table
//     [digit][2][next digit] collects the counts, flushed with 64-byte
//     contiguous global atomics. Only the first digit needs a histogram kernel.
// Between two passes a tiny two-kernel scan turns counts[tile][digit] into
// global offsets (digit-major order, i.e. upstream's counters_sum). No kernel
// waits on another work-group: no look-back, no tickets, no spinning.
//
// HBM traffic per element and digit: one read + one write (upstream: ~5 element
// streams + 6 counter streams).
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "clo_hip.h"
#include "clo_hip_internal.h"

namespace {

constexpr int R4_THREADS = 512;
constexpr int R4_WAVES = R4_THREADS / 64;

// LT = log2(tile elements) = 12: 8 items per thread (4096-element tiles measured
// faster than 8192: occupancy)
template <int LT> struct r4_shape {
	static constexpr int ITEMS = (1 << LT) / R4_THREADS;
	static constexpr int TILE = 1 << LT;
	static constexpr int LOG_TILE = LT;
};
unsigned long long* g_r4_dbg = nullptr;  // developer stamps buffer, see clo_hip_radix_set_debug_buffer

// ---------------------------------------------------------------------------
// counts[tile][digit] -> offsets[tile][digit] in digit-major order:
//   off[t][d] = sum_{d'<d} total[d'] + sum_{t'<t} cnt[t'][d]
// (exactly upstream's exclusive scan of counters[num_wgs*d + wg]).
// Two small kernels over chunks of 256 tiles.
// ---------------------------------------------------------------------------
constexpr int OFF_CHUNK = 256;

template <int R>
__global__ __launch_bounds__(OFF_CHUNK)
void clo_radix4_chunksum_kernel(const unsigned* __restrict__ thist, unsigned tiles, unsigned* __restrict__ partial) {
	__shared__ unsigned s_w[OFF_CHUNK / 64][R];
	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const unsigned t = blockIdx.x * OFF_CHUNK + tid;
	#pragma unroll
	for (int d = 0; d < R; ++d) {
		const unsigned c = t < tiles ? thist[(size_t) t * R + d] : 0u;
		const unsigned sum = clo_wave_reduce_sum<unsigned>(c);
		if (lane == 0) s_w[wave][d] = sum;
	}
	__syncthreads();
	if (tid < (unsigned) R) {
		unsigned s = 0;
		#pragma unroll
		for (int w = 0; w < OFF_CHUNK / 64; ++w) s += s_w[w][tid];
		partial[(size_t) blockIdx.x * R + tid] = s;
	}
}

template <int R>
__global__ __launch_bounds__(OFF_CHUNK)
void clo_radix4_offsets_kernel(const unsigned* __restrict__ thist, unsigned tiles,
	const unsigned* __restrict__ partial, unsigned chunks, unsigned* __restrict__ toff) {
	__shared__ unsigned s_before[R];   // count of digit d in earlier chunks
	__shared__ unsigned s_total[R];    // count of digit d in all chunks
	__shared__ unsigned s_w[OFF_CHUNK / 64][R];
	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	// every block re-derives its starting point from the chunk sums (<= a few KB):
	// thread (g, d) sums digit d over chunks g, g+G, ... in registers, then the G
	// groups are combined through LDS
	constexpr int G = OFF_CHUNK / R;
	__shared__ unsigned s_pb[G][R], s_pt[G][R];
	if (chunks > 1) {  // (a single chunk derives its totals from its own scan below: one launch fewer)
		const unsigned d = tid % R, g = tid / R;
		unsigned before = 0, total = 0;
		for (unsigned c = g; c < chunks; c += G) {
			const unsigned v = partial[(size_t) c * R + d];
			total += v;
			if (c < blockIdx.x) before += v;
		}
		s_pb[g][d] = before;
		s_pt[g][d] = total;
	}
	__syncthreads();
	if (chunks > 1 && tid < (unsigned) R) {
		unsigned before = 0, total = 0;
		#pragma unroll
		for (int g = 0; g < G; ++g) { before += s_pb[g][tid]; total += s_pt[g][tid]; }
		s_before[tid] = before;
		s_total[tid] = total;
	}
	__syncthreads();
	const unsigned t = blockIdx.x * OFF_CHUNK + tid;
	unsigned cnt[R], excl[R];
	#pragma unroll
	for (int d = 0; d < R; ++d) {
		cnt[d] = t < tiles ? thist[(size_t) t * R + d] : 0u;
		const unsigned incl = clo_wave_scan_inclusive<unsigned>(cnt[d], lane);
		excl[d] = incl - cnt[d];
		if (lane == 63) s_w[wave][d] = incl;
	}
	__syncthreads();
	if (chunks == 1) {
		if (tid < (unsigned) R) {
			unsigned total = 0;
			#pragma unroll
			for (int w = 0; w < OFF_CHUNK / 64; ++w) total += s_w[w][tid];
			s_before[tid] = 0;
			s_total[tid] = total;
		}
		__syncthreads();
	}
	if (t < tiles) {
		unsigned dbase = 0;
		#pragma unroll
		for (int d = 0; d < R; ++d) {
			unsigned add = s_before[d] + dbase;
			#pragma unroll
			for (int w = 0; w < OFF_CHUNK / 64; ++w) if ((unsigned) w < wave) add += s_w[w][d];
			toff[(size_t) t * R + d] = excl[d] + add;
			dbase += s_total[d];
		}
	}
}

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
// order, so the ranking is stable. (45 VALU per element and pass.)
// ---------------------------------------------------------------------------

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_add(unsigned x) {
	return x + (unsigned) __builtin_amdgcn_update_dpp(0, (int) x, CTRL, ROW_MASK, 0xF, true);
}

// Inclusive scan over the 64 lanes with DPP: row_shr 1,2,4,8 inside rows of
// 16, then row_bcast:15 (into rows 1,3) and row_bcast:31 (into rows 2,3).
__device__ __forceinline__ unsigned wave_scan_dpp(unsigned x) {
	x = dpp_add<0x111, 0xF>(x);
	x = dpp_add<0x112, 0xF>(x);
	x = dpp_add<0x114, 0xF>(x);
	x = dpp_add<0x118, 0xF>(x);
	x = dpp_add<0x142, 0xA>(x);
	x = dpp_add<0x143, 0xC>(x);
	return x;
}

// Thread-private digit counters: digit q lives in bits [4q, 4q+4) of a 64-bit
// word. Good for up to 15 elements per thread.
struct packed4 {
	unsigned long long c;
};

// Count one digit; returns how many equal digits this thread counted before.
template <int BITS>
__device__ __forceinline__ unsigned packed4_count(packed4& c, unsigned d) {
	const unsigned sh = d * 4u;
	const unsigned prev = (unsigned) (c.c >> sh) & 15u;
	c.c += 1ull << sh;
	return prev;
}

template <int BITS> struct pc_words { static constexpr int H = (1 << BITS) >= 2 ? (1 << BITS) / 2 : 1; };

// Widen to 16-bit fields: w[j] = count(2j) | count(2j+1) << 16.
template <int BITS>
__device__ __forceinline__ void packed4_widen(const packed4& c, unsigned (&w)[pc_words<BITS>::H]) {
	#pragma unroll
	for (int j = 0; j < pc_words<BITS>::H; ++j) {
		const unsigned x = j < 4 ? (unsigned) c.c : (unsigned) (c.c >> 32);
		const int q = (j & 3) * 8;
		w[j] = ((x >> q) & 15u) | (((x >> (q + 4)) & 15u) << 16);
	}
}

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

template <typename E, int BITS, int LT>
__global__ __launch_bounds__(R4_THREADS)
void clo_radix4_tilehist_pc_kernel(const E* __restrict__ in, size_t n, unsigned shift, unsigned mask,
	unsigned* __restrict__ thist, int aligned, clo_keyx kx) {
	constexpr int R = 1 << BITS;
	constexpr int H = pc_words<BITS>::H;
	constexpr int ITEMS = r4_shape<LT>::ITEMS;
	constexpr int TILE = r4_shape<LT>::TILE;
	static_assert(ITEMS <= 15, "4-bit thread-private counters");
	__shared__ unsigned s_wtot[R4_WAVES][H];
	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const size_t base = (size_t) blockIdx.x * TILE;
	const unsigned count = (n - base) < (size_t) TILE ? (unsigned) (n - base) : (unsigned) TILE;
	const unsigned tbase = tid * ITEMS;
	packed4 c = { 0ull };
	if (count == (unsigned) TILE) {
		E key[ITEMS];
		load_blocked<E, ITEMS>(in + base + tbase, key, aligned != 0);
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) packed4_count<BITS>(c, (unsigned) (clo_keyx_fwd<E>(key[i], kx) >> shift) & mask);
	} else {
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i)
			if (tbase + i < count) packed4_count<BITS>(c, (unsigned) (clo_keyx_fwd<E>(in[base + tbase + i], kx) >> shift) & mask);
	}
	unsigned w[H];
	packed4_widen<BITS>(c, w);
	#pragma unroll
	for (int j = 0; j < H; ++j) {
		const unsigned tot = wave_scan_dpp(w[j]);
		if (lane == 63) s_wtot[wave][j] = tot;
	}
	__syncthreads();
	if (tid < (unsigned) R) {
		unsigned h = 0;
		#pragma unroll
		for (int wv = 0; wv < R4_WAVES; ++wv) h += (s_wtot[wv][tid >> 1] >> ((tid & 1u) * 16u)) & 0xffffu;
		thist[(size_t) blockIdx.x * R + tid] = h;
	}
}

// ---- pass kernel: two-stage packed scan ----
// The thread-private 4-bit counters first widen to 8-bit fields only (4
// digits per VGPR: even digits in one word, odd digits in the next): an
// inclusive scan inside a row of 16 lanes cannot exceed 16 * 15 = 240. Only
// then do they widen to 16-bit fields for the two cross-row steps; v_perm_b32
// builds w[j] = count(2j) | count(2j+1) << 16 from one even and one odd word.
template <int BITS>
__device__ __forceinline__ void pc2_wave_scan(unsigned long long c, unsigned (&w)[pc_words<BITS>::H]) {
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
	#pragma unroll
	for (int k = 0; k < NB; ++k) {
		unsigned x = b[k];
		x = dpp_add<0x111, 0xF>(x);
		x = dpp_add<0x112, 0xF>(x);
		x = dpp_add<0x114, 0xF>(x);
		b[k] = dpp_add<0x118, 0xF>(x);
	}
	#pragma unroll
	for (int j = 0; j < H; ++j) {
		// byte (j & 3) of the even word -> bits 0..15, of the odd word -> bits 16..31
		const unsigned sel = 0x0c040c00u + (unsigned) (j & 3) * 0x00010001u;
		unsigned x = __builtin_amdgcn_perm(b[(j >> 2) * 2 + 1], b[(j >> 2) * 2], sel);
		x = dpp_add<0x142, 0xA>(x);
		w[j] = dpp_add<0x143, 0xC>(x);
	}
}

template <typename E, int BITS, int LT>
__global__ __launch_bounds__(R4_THREADS)
void clo_radix4_pass_pc_kernel(const E* __restrict__ in, E* __restrict__ out, size_t n,
	unsigned shift, unsigned mask, int has_next, unsigned next_shift, unsigned next_mask,
	const unsigned* __restrict__ thist, const unsigned* __restrict__ toff,
	unsigned* __restrict__ thist_next, int aligned, clo_keyx kx_in, clo_keyx kx_out, unsigned long long* dbg) {

	constexpr int R = 1 << BITS;
	constexpr int NW = pc_words<BITS>::H;
	constexpr int ITEMS = r4_shape<LT>::ITEMS;
	constexpr int TILE = r4_shape<LT>::TILE;
	constexpr int LOG_TILE = r4_shape<LT>::LOG_TILE;
	static_assert(ITEMS <= 15, "4-bit thread-private counters, 8-bit row sums");
	static_assert(TILE <= 65536 / 2, "16-bit positions");
	// developer diagnostics: phase stamps of the first 32768 tiles (dbg != NULL only in tools/stamp_probe.py)
	#define CLO_STAMP(k) do { if (dbg && threadIdx.x == 0 && blockIdx.x < 32768u) dbg[(size_t) blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
	CLO_STAMP(0);

	__shared__ E s_stage[TILE];
	__shared__ unsigned s_end[NW][R4_THREADS];        // [digit / 2][thread]: tile-local END of the thread's slice of the two
	                                                  // digits (16-bit fields); a lane only touches its own column: conflict-free
	__shared__ unsigned s_wtot[R4_WAVES][NW];         // wave totals (packed like the counters)
	__shared__ unsigned s_wbase[R4_WAVES][NW];        // digit start + totals of earlier waves
	__shared__ unsigned s_next[R][2][R];              // [digit][destination tile 0/1][next digit]
	__shared__ unsigned s_delta[R];                   // global index = tile-local position + delta[digit]
	__shared__ unsigned s_comb[R];                    // s_next row of (digit, destination tile t) = t*R + comb[digit]
	__shared__ unsigned s_dstart16[NW];               // tile-local digit starts, packed like the counters

	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	// Work-groups are dealt round-robin over the 8 XCDs (observed, not promised —
	// used for speed only: any mapping is correct, tiles are independent). Giving
	// each residue class of blockIdx a contiguous range of tiles puts neighbouring
	// tiles, whose digit runs share their boundary cache lines in the output,
	// behind the same L2.
	const unsigned per_xcd = (unsigned) ((n + (size_t) TILE * 8 - 1) / ((size_t) TILE * 8));
	const unsigned tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
	const size_t base = (size_t) tile * TILE;
	if (base >= n) return;
	const unsigned count = (n - base) < (size_t) TILE ? (unsigned) (n - base) : (unsigned) TILE;
	const bool full = count == (unsigned) TILE;
	const unsigned tbase = tid * ITEMS;

	// tile bookkeeping from the scanned counters (upstream's offsets /
	// counters_sum): requested BEFORE the keys so that their latency hides
	// behind the key loads instead of stalling wave 0 in front of a barrier
	unsigned h = 0, goff = 0;
	if (tid < (unsigned) R) {
		h = thist[(size_t) tile * R + tid];
		goff = toff[(size_t) tile * R + tid];
	}
	for (unsigned i = tid; i < R * 2 * R; i += R4_THREADS) (&s_next[0][0][0])[i] = 0;

	// ---- 1. load: ITEMS consecutive elements per thread (16-byte vector loads) ----
	E key[ITEMS];
	if (full) {
		load_blocked<E, ITEMS>(in + base + tbase, key, aligned != 0);
	} else {
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) key[i] = (tbase + i < count) ? in[base + tbase + i] : (E) 0;
	}
	if (kx_in.kind) {   // first pass of a sort on signed / floating-point keys
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) key[i] = clo_keyx_fwd<E>(key[i], kx_in);
	}

	if (dbg) { asm volatile("" :: "v"((unsigned) key[ITEMS - 1])); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
	CLO_STAMP(1);

	// ---- 2a. thread-private digit counts, LAST element first: rr = 1 + the
	// number of LATER elements of the thread with the same digit, so that the
	// element's position is (end of the thread's slice of that digit) - rr ----
	unsigned long long c = 0;
	unsigned rr = 0;   // 4 bits per element
	#pragma unroll
	for (int i = ITEMS - 1; i >= 0; --i) {
		if (full || tbase + i < count) {
			const unsigned sh = ((unsigned) (key[i] >> shift) & mask) * 4u;
			c += 1ull << sh;
			rr |= ((unsigned) (c >> sh) & 15u) << (4 * i);
		}
	}

	// ---- 2b. inclusive count of every digit over the threads of the wave ----
	unsigned w[NW];
	pc2_wave_scan<BITS>(c, w);
	if (lane == 63) {
		#pragma unroll
		for (int j = 0; j < NW; ++j) s_wtot[wave][j] = w[j];
	}
	if (tid < 64) {
		const unsigned dstart = clo_wave_scan_inclusive<unsigned>(h, lane) - h;
		if (tid < (unsigned) R) {
			s_delta[tid] = goff - dstart;
			s_comb[tid] = tid * 2u * R - ((goff >> LOG_TILE) << BITS);
		}
		const unsigned odd = (unsigned) __shfl((int) dstart, (int) (lane | 1u), 64);
		if (tid < (unsigned) R && (tid & 1u) == 0) s_dstart16[tid >> 1] = dstart | ((R > 1 ? odd : 0u) << 16);
	}
	CLO_STAMP(2);
	__syncthreads();
	CLO_STAMP(3);
	if (tid < R4_WAVES * NW) {
		const unsigned wv = tid / NW, j = tid % NW;
		unsigned run = s_dstart16[j];
		for (unsigned k = 0; k < wv; ++k) run += s_wtot[k][j];
		s_wbase[wv][j] = run;
	}
	__syncthreads();
	CLO_STAMP(4);
	#pragma unroll
	for (int j = 0; j < NW; ++j) s_end[j][tid] = w[j] + s_wbase[wave][j];

	// ---- 4a. scatter into the LDS stage in digit order ----
	#pragma unroll
	for (int i = 0; i < ITEMS; ++i) {
		if (full || tbase + i < count) {
			const unsigned d = (unsigned) (key[i] >> shift) & mask;
			const unsigned end = (s_end[d >> 1][tid] >> ((d & 1u) * 16u)) & 0xffffu;
			const unsigned pos = end - ((rr >> (4 * i)) & 15u);
			s_stage[pos & (TILE - 1)] = key[i];
		}
	}
	if (dbg) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	CLO_STAMP(5);
	__syncthreads();
	CLO_STAMP(6);

	// ---- 4b. contiguous runs to HBM; 5. next digit's per-tile histogram ----
	// A thread takes VEC consecutive positions of the digit-sorted tile (one
	// 16-byte LDS read). Inside a digit run they go to consecutive global
	// indices: one 16-byte store (only element alignment is guaranteed, which
	// global memory accepts); a group that straddles two runs is stored
	// element by element.
	constexpr int VEC = sizeof(E) >= 8 ? 1 : 4;
	typedef E vecE __attribute__((ext_vector_type(VEC)));
	typedef E vecE_u __attribute__((ext_vector_type(VEC), aligned(sizeof(E))));
	unsigned* const next_flat = &s_next[0][0][0];
	auto count_next = [&](E e, unsigned d, unsigned gi) {
		const unsigned row = ((gi >> LOG_TILE) << BITS) + s_comb[d];   // (digit*2 + destination tile 0/1) * R
		atomicAdd(&next_flat[(row + ((unsigned) (e >> next_shift) & next_mask)) & (2u * R * R - 1u)], 1u);
	};
	const unsigned n32 = n > 0xffffffffull ? 0xffffffffu : (unsigned) n;   // global indices are 32-bit here
	#pragma unroll
	for (int j = 0; j < ITEMS / VEC; ++j) {
		const unsigned p = (j * R4_THREADS + tid) * VEC;
		if (full) {
			const vecE v = *reinterpret_cast<const vecE*>(&s_stage[p]);
			const unsigned d0 = (unsigned) (v[0] >> shift) & mask, dl = (unsigned) (v[VEC - 1] >> shift) & mask;
			const unsigned gi0 = p + s_delta[d0];
			if (d0 == dl && gi0 <= n32 - VEC) {
				vecE vo = v;
				if (kx_out.kind) {   // last pass: back to the caller's encoding
					#pragma unroll
					for (int k = 0; k < VEC; ++k) vo[k] = clo_keyx_inv<E>(v[k], kx_out);
				}
				*reinterpret_cast<vecE_u*>(&out[gi0]) = vo;
				if (has_next) {
					if ((gi0 >> LOG_TILE) == ((gi0 + VEC - 1) >> LOG_TILE)) {
						// one destination tile: one row of the table
						unsigned* const rowp = &next_flat[(((gi0 >> LOG_TILE) << BITS) + s_comb[d0]) & (2u * R * R - R)];
						#pragma unroll
						for (int k = 0; k < VEC; ++k) atomicAdd(&rowp[(unsigned) (v[k] >> next_shift) & next_mask & (R - 1u)], 1u);
					} else {
						#pragma unroll
						for (int k = 0; k < VEC; ++k) count_next(v[k], d0, gi0 + k);
					}
				}
			} else {
				#pragma unroll
				for (int k = 0; k < VEC; ++k) {
					const unsigned d = (unsigned) (v[k] >> shift) & mask;
					const unsigned gi = p + k + s_delta[d];
					if (gi < n32) {
						out[gi] = clo_keyx_inv<E>(v[k], kx_out);
						if (has_next) count_next(v[k], d, gi);
					}
				}
			}
		} else {
			#pragma unroll
			for (int k = 0; k < VEC; ++k) {
				if (p + k < count) {
					const E e = s_stage[p + k];
					const unsigned d = (unsigned) (e >> shift) & mask;
					const unsigned gi = p + k + s_delta[d];
					if (gi < n32) {
						out[gi] = clo_keyx_inv<E>(e, kx_out);
						if (has_next) count_next(e, d, gi);
					}
				}
			}
		}
	}
	if (dbg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	CLO_STAMP(7);
	#undef CLO_STAMP
	if (has_next) {
		__syncthreads();
		// 16 consecutive lanes = the 16 counters of one destination tile (64 B)
		for (unsigned i = tid; i < R * 2 * R; i += R4_THREADS) {
			const unsigned v = next_flat[i];
			if (v) {
				const unsigned d = i / (2 * R), half = (i / R) & 1u, dn = i % R;
				const unsigned first = (d * 2u * R - s_comb[d]) >> BITS;
				atomicAdd(&thist_next[(size_t) (first + half) * R + dn], v);
			}
		}
	}
}

// ---------------------------------------------------------------------------
// Digit pairs: TWO digit steps per trip through HBM.
//
// A tile is split by digit `lo` and then by digit `hi` inside the work-group
// (two stable local splits through the LDS stage, each with the packed-counter
// ranking above), which leaves it sorted by the combined digit D = hi:lo. The
// global step — per-tile histogram of D (clo_hip_radixw.hip), digit-major scan
// of the counters, scatter — then runs once for the combined digit. This is the
// pass for digits of 5..8 bits (radix 32..256), split in two halves of <= 4
// bits so that the packed-counter ranking applies. Stability of both local
// splits and of the scatter makes the result the one the reference's passes
// produce.
// (Measured as a replacement for two single-digit passes on 4-bit digits — 12
// element streams per 32-bit key instead of 17 — it LOSES: 0.79 ms per pair
// against 2 x 0.46 + fused histograms, 4.38 ms against 3.94 ms for 2^28 keys.
// The single-digit pass keeps its LDS pipe ~90 % busy (SQ_ACTIVE_INST_LDS), so
// a second local split costs what a second pass costs. Against ranking wide
// digits with one ballot per digit bit it wins: 0.79 ms vs 0.87 ms per 8-bit
// pass, 0.58 vs 0.75 ms per 6-bit pass.)
// ---------------------------------------------------------------------------

// One stable local split of the tile by the digit (key >> dshift) & dmask;
// on return (after a barrier) s_stage holds the tile in digit order. The
// thread's elements are ITEMS consecutive positions of the tile.
template <typename E, int BITS, int THREADS, int ITEMS, int HMAX>
__device__ __forceinline__ void pc_local_split(const E (&key)[ITEMS], unsigned dshift, unsigned dmask, unsigned count,
	E* s_stage, unsigned (*s_end)[THREADS], unsigned (*s_wtot)[HMAX], unsigned (*s_wbase)[HMAX], unsigned* s_dstart16) {
	constexpr int H = pc_words<BITS>::H;
	constexpr int WAVES = THREADS / 64;
	constexpr int TILE = THREADS * ITEMS;
	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const unsigned tbase = tid * ITEMS;
	const bool full = count == (unsigned) TILE;

	// thread-private counts, LAST element first (see the pass kernel above)
	unsigned long long c = 0;
	unsigned rr = 0;
	#pragma unroll
	for (int i = ITEMS - 1; i >= 0; --i) {
		if (full || tbase + i < count) {
			const unsigned sh = ((unsigned) (key[i] >> dshift) & dmask) * 4u;
			c += 1ull << sh;
			rr |= ((unsigned) (c >> sh) & 15u) << (4 * i);
		}
	}
	unsigned w[H];
	pc2_wave_scan<BITS>(c, w);
	if (lane == 63) {
		#pragma unroll
		for (int j = 0; j < H; ++j) s_wtot[wave][j] = w[j];
	}
	__syncthreads();
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
			const unsigned t = (unsigned) __shfl((int) tot[j / 4], (j % 4) * 16 + 15, 64);   // packed totals of digits 2j, 2j+1
			dstart16[j] = run | ((run + (t & 0xffffu)) << 16);
			run += (t & 0xffffu) + (t >> 16);
		}
		#pragma unroll
		for (int r = 0; r < ROUNDS; ++r) {
			unsigned mine = dstart16[r * 4];
			#pragma unroll
			for (int k = 1; k < 4; ++k) if (r * 4 + k < H && q == (unsigned) k) mine = dstart16[r * 4 + k];
			const unsigned j = r * 4 + q;
			if (j < (unsigned) H && wv < (unsigned) WAVES) s_wbase[wv][j] = mine + excl[r];
		}
	}
	__syncthreads();
	#pragma unroll
	for (int j = 0; j < H; ++j) s_end[j][tid] = w[j] + s_wbase[wave][j];
	#pragma unroll
	for (int i = 0; i < ITEMS; ++i) {
		if (full || tbase + i < count) {
			const unsigned d = (unsigned) (key[i] >> dshift) & dmask;
			const unsigned end = (s_end[d >> 1][tid] >> ((d & 1u) * 16u)) & 0xffffu;
			s_stage[(end - ((rr >> (4 * i)) & 15u)) & (TILE - 1)] = key[i];
		}
	}
	__syncthreads();
}

template <typename E> struct pair_shape {
	static constexpr int THREADS = sizeof(E) == 8 ? 512 : 1024;   // 8 items per thread: tiles of 4096 / 8192 elements,
	static constexpr int ITEMS = 8;                               // = the tiles of clo_hip_radixw.hip's histogram
};

template <typename E, int LB, int HB>
__global__ __launch_bounds__(pair_shape<E>::THREADS)
void clo_radix4_pair_kernel(const E* __restrict__ in, E* __restrict__ out, size_t n,
	unsigned shift, unsigned mask_lo, unsigned mask_hi,
	const unsigned* __restrict__ thist, const unsigned* __restrict__ toff, int aligned,
	clo_keyx kx_in, clo_keyx kx_out, unsigned xf) {

	constexpr int THREADS = pair_shape<E>::THREADS;
	constexpr int ITEMS = pair_shape<E>::ITEMS;
	constexpr int TILE = THREADS * ITEMS;
	constexpr int WAVES = THREADS / 64;
	constexpr int R2 = 1 << (LB + HB);
	constexpr int HMAX = pc_words<(LB > HB ? LB : HB)>::H;
	static_assert(R2 <= 256 && R2 <= THREADS, "one thread per combined digit, scanned by the first four waves");

	__shared__ E s_stage[TILE];
	__shared__ unsigned s_end[HMAX][THREADS];
	__shared__ unsigned s_wtot[WAVES][HMAX];
	__shared__ unsigned s_wbase[WAVES][HMAX];
	__shared__ unsigned s_dstart16[HMAX];
	__shared__ unsigned s_delta[R2];   // global index = tile-local position + delta[D]
	__shared__ unsigned s_w4[4];

	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	// neighbouring tiles behind the same L2 (see clo_radix4_pass_pc_kernel)
	const unsigned per_xcd = (unsigned) ((n + (size_t) TILE * 8 - 1) / ((size_t) TILE * 8));
	const unsigned tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
	const size_t base = (size_t) tile * TILE;
	if (base >= n) return;
	const unsigned count = (n - base) < (size_t) TILE ? (unsigned) (n - base) : (unsigned) TILE;
	const bool full = count == (unsigned) TILE;
	const unsigned tbase = tid * ITEMS;
	const unsigned mask2 = (mask_hi << LB) | mask_lo;
	const unsigned n32 = n > 0xffffffffull ? 0xffffffffu : (unsigned) n;   // global indices are 32-bit here

	// the tile's counters (upstream's counters / counters_sum), requested before the keys
	unsigned h2 = 0, goff = 0;
	if (tid < (unsigned) R2) {
		h2 = thist[(size_t) tile * R2 + tid];
		goff = toff[(size_t) tile * R2 + tid];
	}
	E key[ITEMS];
	if (full) {
		load_blocked<E, ITEMS>(in + base + tbase, key, aligned != 0);
	} else {
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) key[i] = (tbase + i < count) ? in[base + tbase + i] : (E) 0;
	}
	if (kx_in.kind) {
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) key[i] = clo_keyx_fwd<E>(key[i], kx_in);
	}
	// tile-local start of every combined digit: exclusive scan of the tile's histogram
	const unsigned incl2 = clo_wave_scan_inclusive<unsigned>(h2, lane);
	if (lane == 63 && wave < 4) s_w4[wave] = incl2;

	if (!(xf & 4u)) pc_local_split<E, LB, THREADS, ITEMS, HMAX>(key, shift, mask_lo, count, s_stage, s_end, s_wtot, s_wbase, s_dstart16);
	else { for (int i = 0; i < ITEMS; ++i) s_stage[tbase + i] = key[i]; __syncthreads(); }

	if (tid < (unsigned) R2) {
		unsigned dstart2 = incl2 - h2;
		#pragma unroll
		for (unsigned w = 0; w < 4; ++w) if (w < wave) dstart2 += s_w4[w];
		s_delta[tid] = goff - dstart2;
	}
	if (mask_hi != 0 && !(xf & 1u)) {
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
		pc_local_split<E, HB, THREADS, ITEMS, HMAX>(key, shift + LB, mask_hi, count, s_stage, s_end, s_wtot, s_wbase, s_dstart16);
	} else {
		__syncthreads();
	}

	// contiguous runs to HBM: a thread takes VEC consecutive positions of the
	// sorted tile; inside a run they go out as one 16-byte store
	constexpr int VEC = sizeof(E) >= 8 ? 1 : 4;
	typedef E vecE __attribute__((ext_vector_type(VEC)));
	typedef E vecE_u __attribute__((ext_vector_type(VEC), aligned(sizeof(E))));
	#pragma unroll
	for (int j = 0; j < ITEMS / VEC; ++j) {
		const unsigned p = (j * THREADS + tid) * VEC;
		if (full) {
			const vecE v = *reinterpret_cast<const vecE*>(&s_stage[p]);
			const unsigned d0 = (unsigned) (v[0] >> shift) & mask2, dl = (unsigned) (v[VEC - 1] >> shift) & mask2;
			const unsigned gi0 = (xf & 2u) ? (unsigned) base + p : p + s_delta[d0];
			if (((xf & 2u) || d0 == dl) && gi0 <= n32 - VEC) {
				vecE vo = v;
				if (kx_out.kind) {
					#pragma unroll
					for (int k = 0; k < VEC; ++k) vo[k] = clo_keyx_inv<E>(v[k], kx_out);
				}
				*reinterpret_cast<vecE_u*>(&out[gi0]) = vo;
			} else {
				#pragma unroll
				for (int k = 0; k < VEC; ++k) {
					const unsigned gi = p + k + s_delta[(unsigned) (v[k] >> shift) & mask2];
					if (gi < n32) out[gi] = clo_keyx_inv<E>(v[k], kx_out);
				}
			}
		} else {
			#pragma unroll
			for (int k = 0; k < VEC; ++k) {
				if (p + k < count) {
					const E e = s_stage[p + k];
					const unsigned gi = p + k + s_delta[(unsigned) (e >> shift) & mask2];
					if (gi < n32) out[gi] = clo_keyx_inv<E>(e, kx_out);
				}
			}
		}
	}
}

// ---------------------------------------------------------------------------
// Arrays of at most one tile: every digit pass inside ONE work-group, one
// launch for the whole sort (upstream's harness sweeps sizes from 2^4 up; a
// multi-kernel sort costs ~25 dependent launches however small the array).
// Same packed-counter ranking as above; the tile goes through the LDS stage
// once per digit.
// ---------------------------------------------------------------------------
template <typename E, int BITS, int LT>
__global__ __launch_bounds__(R4_THREADS)
void clo_radix4_small_kernel(const E* in, E* out, unsigned n,
	unsigned key_shift, unsigned key_bits, clo_keyx kx) {
	constexpr int R = 1 << BITS;
	constexpr int H = pc_words<BITS>::H;
	constexpr int ITEMS = r4_shape<LT>::ITEMS;
	constexpr int TILE = r4_shape<LT>::TILE;
	__shared__ E s_stage[TILE];
	__shared__ unsigned s_scr[H][R4_THREADS];
	__shared__ unsigned s_wtot[R4_WAVES][H];
	__shared__ unsigned s_wbase[R4_WAVES][H];
	__shared__ unsigned s_dstart16[H];

	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const unsigned tbase = tid * ITEMS;
	E key[ITEMS];
	#pragma unroll
	for (int i = 0; i < ITEMS; ++i) key[i] = (tbase + i < n) ? clo_keyx_fwd<E>(in[tbase + i], kx) : (E) 0;

	for (unsigned done = 0; done < key_bits; done += BITS) {
		const unsigned shift = key_shift + done;
		const unsigned bits = key_bits - done < (unsigned) BITS ? key_bits - done : (unsigned) BITS;
		const unsigned mask = (1u << bits) - 1u;
		packed4 c = { 0ull };
		unsigned lrank = 0;
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i)
			if (tbase + i < n) lrank |= packed4_count<BITS>(c, (unsigned) (key[i] >> shift) & mask) << (4 * i);
		unsigned w[H];
		packed4_widen<BITS>(c, w);
		#pragma unroll
		for (int j = 0; j < H; ++j) {
			const unsigned incl = wave_scan_dpp(w[j]);
			if (lane == 63) s_wtot[wave][j] = incl;
			w[j] = incl - w[j];
		}
		__syncthreads();
		if (tid < 64) {
			// digit totals -> tile-local digit starts (exclusive scan over digits)
			unsigned h = 0;
			if (tid < (unsigned) R) {
				#pragma unroll
				for (int wv = 0; wv < R4_WAVES; ++wv) h += (s_wtot[wv][tid >> 1] >> ((tid & 1u) * 16u)) & 0xffffu;
			}
			const unsigned dstart = clo_wave_scan_inclusive<unsigned>(h, lane) - h;
			const unsigned odd = (unsigned) __shfl((int) dstart, (int) (lane | 1u), 64);
			if (tid < (unsigned) R && (tid & 1u) == 0) s_dstart16[tid >> 1] = dstart | ((R > 1 ? odd : 0u) << 16);
		}
		__syncthreads();
		if (tid < R4_WAVES * H) {
			const unsigned wv = tid / H, j = tid % H;
			unsigned run = s_dstart16[j];
			for (unsigned k = 0; k < wv; ++k) run += s_wtot[k][j];
			s_wbase[wv][j] = run;
		}
		__syncthreads();
		#pragma unroll
		for (int j = 0; j < H; ++j) s_scr[j][tid] = w[j] + s_wbase[wave][j];
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) {
			if (tbase + i < n) {
				const unsigned d = (unsigned) (key[i] >> shift) & mask;
				const unsigned start = (s_scr[d >> 1][tid] >> ((d & 1u) * 16u)) & 0xffffu;
				s_stage[start + ((lrank >> (4 * i)) & 15u)] = key[i];
			}
		}
		__syncthreads();
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) if (tbase + i < n) key[i] = s_stage[tbase + i];
		__syncthreads();
	}
	#pragma unroll
	for (int i = 0; i < ITEMS; ++i) if (tbase + i < n) out[tbase + i] = clo_keyx_inv<E>(key[i], kx);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

struct r4_layout { size_t thist, toff, partial, total, tiles, chunks; };

r4_layout r4_make_layout(size_t n, int elem_size, int passes, int digit_bits, int log_tile) {
	r4_layout L;
	const size_t R = (size_t) 1 << digit_bits;
	const size_t tile = (size_t) 1 << log_tile;
	L.tiles = (n + tile - 1) / tile;
	if (L.tiles == 0) L.tiles = 1;
	L.chunks = (L.tiles + OFF_CHUNK - 1) / OFF_CHUNK;
	const size_t per_pass = (L.tiles + 1) * R * sizeof(unsigned);  // +1: a run may touch the tile after the last
	L.thist = CLO_WS_HEADER_BYTES;
	L.toff = L.thist + (size_t) passes * per_pass;
	L.partial = L.toff + per_pass;
	L.total = L.partial + ((L.chunks * R * sizeof(unsigned) + 255) & ~(size_t) 255);
	return L;
}

template <typename E, int BITS, int LT>
int r4_sort_impl(const E* src, E* dst, E* tmp, size_t n, int key_shift, int key_bits, clo_keyx kx, void* ws, hipStream_t s) {
	constexpr unsigned R = 1u << BITS;
	const int passes = (key_bits + BITS - 1) / BITS;
	const r4_layout L = r4_make_layout(n, (int) sizeof(E), passes, BITS, LT);
	const size_t per_pass = (L.tiles + 1) * R;
	unsigned* thist = (unsigned*) ((char*) ws + L.thist);
	unsigned* toff = (unsigned*) ((char*) ws + L.toff);
	unsigned* partial = (unsigned*) ((char*) ws + L.partial);
	const unsigned tiles = (unsigned) L.tiles, chunks = (unsigned) L.chunks;

	const clo_keyx kx_none = { 0, 0, 0 };
	if (n <= ((size_t) 1 << LT)) {
		clo_timing_scope timing("radix_small", s);
		hipLaunchKernelGGL((clo_radix4_small_kernel<E, BITS, LT>), dim3(1), dim3(R4_THREADS), 0, s,
			src, dst, (unsigned) n, (unsigned) key_shift, (unsigned) key_bits, kx);
		return (int) hipGetLastError();
	}

	// zero the header (status word) and every pass's histogram
	hipError_t e = hipMemsetAsync(ws, 0, L.toff, s);
	if (e != hipSuccess) return (int) e;

	const unsigned bits0 = key_bits < BITS ? key_bits : BITS;
	{
		clo_timing_scope timing("radix_hist", s);
		hipLaunchKernelGGL((clo_radix4_tilehist_pc_kernel<E, BITS, LT>), dim3(tiles), dim3(R4_THREADS), 0, s,
			src, n, (unsigned) key_shift, (1u << bits0) - 1u, thist, (int) ((uintptr_t) src % 16 == 0), kx);
	}

	const bool inplace_odd = (dst == src) && (passes % 2 == 1);
	const E* cur_in = src;
	for (int p = 0; p < passes; ++p) {
		E* cur_out;
		if (inplace_odd) cur_out = (p % 2 == 0) ? tmp : dst;
		else cur_out = ((passes - 1 - p) % 2 == 0) ? dst : tmp;
		const int rem = key_bits - p * BITS;
		const unsigned bits = rem < BITS ? rem : BITS;
		const int has_next = p + 1 < passes;
		const int nrem = key_bits - (p + 1) * BITS;
		const unsigned nbits = has_next ? (nrem < BITS ? nrem : BITS) : 1;
		unsigned* th = thist + (size_t) p * per_pass;
		{
			clo_timing_scope timing("radix_offsets", s);
			if (chunks > 1)
				hipLaunchKernelGGL((clo_radix4_chunksum_kernel<R>), dim3(chunks), dim3(OFF_CHUNK), 0, s,
					(const unsigned*) th, tiles, partial);
			hipLaunchKernelGGL((clo_radix4_offsets_kernel<R>), dim3(chunks), dim3(OFF_CHUNK), 0, s,
				(const unsigned*) th, tiles, (const unsigned*) partial, chunks, toff);
		}
		{
			clo_timing_scope timing("radix_pass", s);
			hipLaunchKernelGGL((clo_radix4_pass_pc_kernel<E, BITS, LT>), dim3((tiles + 7u) / 8u * 8u), dim3(R4_THREADS), 0, s,
				cur_in, cur_out, n, (unsigned) (key_shift + p * BITS), (1u << bits) - 1u,
				has_next, (unsigned) (key_shift + (p + 1) * BITS), (1u << nbits) - 1u,
				(const unsigned*) th, (const unsigned*) toff, th + per_pass,
				(int) ((uintptr_t) cur_in % 16 == 0), p == 0 ? kx : kx_none, has_next ? kx_none : kx, g_r4_dbg);
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

// Bucket sizes of a single-pass partition, from the per-tile histograms
// (uint64 because the exchange plan adds them across ranks).
template <int R>
__global__ __launch_bounds__(256)
void clo_radix4_counts_kernel(const unsigned* __restrict__ thist, unsigned tiles, unsigned long long* __restrict__ counts) {
	constexpr int G = 256 / R;
	__shared__ unsigned long long s_part[G][R];
	const unsigned tid = threadIdx.x, d = tid % R, g = tid / R;
	unsigned long long c = 0;
	for (unsigned t = g; t < tiles; t += G) c += thist[(size_t) t * R + d];
	s_part[g][d] = c;
	__syncthreads();
	if (tid < (unsigned) R) {
		unsigned long long tot = 0;
		#pragma unroll
		for (int k = 0; k < G; ++k) tot += s_part[k][tid];
		counts[tid] = tot;
	}
}

// One stable pass on `bits` bits at `shift`: the MSD bucket split of the
// multi-GPU exchange. counts (optional) receives the 1 << bits bucket sizes.
template <typename E, int BITS>
int r4_partition_impl(const E* src, E* dst, size_t n, unsigned shift, unsigned long long* counts, void* ws, hipStream_t s) {
	constexpr unsigned R = 1u << BITS;
	constexpr int LT = 12;
	const r4_layout L = r4_make_layout(n, (int) sizeof(E), 1, BITS, LT);
	unsigned* thist = (unsigned*) ((char*) ws + L.thist);
	unsigned* toff = (unsigned*) ((char*) ws + L.toff);
	unsigned* partial = (unsigned*) ((char*) ws + L.partial);
	const unsigned tiles = (unsigned) L.tiles, chunks = (unsigned) L.chunks;
	hipError_t e = hipMemsetAsync(ws, 0, L.toff, s);
	if (e != hipSuccess) return (int) e;
	clo_timing_scope timing("msd_partition", s);
	const clo_keyx kx_none = { 0, 0, 0 };
	hipLaunchKernelGGL((clo_radix4_tilehist_pc_kernel<E, BITS, LT>), dim3(tiles), dim3(R4_THREADS), 0, s,
		src, n, shift, R - 1u, thist, (int) ((uintptr_t) src % 16 == 0), kx_none);
	if (counts)
		hipLaunchKernelGGL((clo_radix4_counts_kernel<R>), dim3(1), dim3(256), 0, s, (const unsigned*) thist, tiles, counts);
	if (chunks > 1)
		hipLaunchKernelGGL((clo_radix4_chunksum_kernel<R>), dim3(chunks), dim3(OFF_CHUNK), 0, s,
			(const unsigned*) thist, tiles, partial);
	hipLaunchKernelGGL((clo_radix4_offsets_kernel<R>), dim3(chunks), dim3(OFF_CHUNK), 0, s,
		(const unsigned*) thist, tiles, (const unsigned*) partial, chunks, toff);
	hipLaunchKernelGGL((clo_radix4_pass_pc_kernel<E, BITS, LT>), dim3((tiles + 7u) / 8u * 8u), dim3(R4_THREADS), 0, s,
		src, dst, n, shift, R - 1u, 0, 0u, 0u, (const unsigned*) thist, (const unsigned*) toff,
		thist + (L.tiles + 1) * R, (int) ((uintptr_t) src % 16 == 0), kx_none, kx_none, (unsigned long long*) nullptr);
	return (int) hipGetLastError();
}

// Wide digits on the host side: per pass, histogram of the digit -> counter
// scan -> pair kernel (the digit split in two halves of <= 4 bits).

struct rp_layout { size_t thist, toff, partial, total, tiles; };

rp_layout rp_make_layout(size_t n, int elem_size, int pass_bits) {
	rp_layout L;
	const size_t R2 = (size_t) 1 << pass_bits;
	const size_t tile = clo_radixw_tile_elems(elem_size);
	L.tiles = (n + tile - 1) / tile;
	if (L.tiles == 0) L.tiles = 1;
	const size_t per = L.tiles * R2 * sizeof(unsigned);
	L.thist = CLO_WS_HEADER_BYTES;
	L.toff = L.thist + per;
	L.partial = L.toff + per;
	L.total = L.partial + (((L.tiles / 128 + 1) * R2 * sizeof(unsigned) + 255) & ~(size_t) 255);
	return L;
}

template <typename E, int LB, int HB>
int rp_sort_impl(const E* src, E* dst, E* tmp, size_t n, int key_shift, int key_bits, clo_keyx kx, void* ws, hipStream_t s) {
	constexpr int PB = LB + HB;   // key bits per trip through HBM
	const int passes = (key_bits + PB - 1) / PB;
	const rp_layout L = rp_make_layout(n, (int) sizeof(E), PB);
	unsigned* thist = (unsigned*) ((char*) ws + L.thist);
	unsigned* toff = (unsigned*) ((char*) ws + L.toff);
	unsigned* partial = (unsigned*) ((char*) ws + L.partial);
	const unsigned tiles = (unsigned) L.tiles;
	const clo_keyx kx_none = { 0, 0, 0 };

	hipError_t e = hipMemsetAsync(ws, 0, CLO_WS_HEADER_BYTES, s);   // status word
	if (e != hipSuccess) return (int) e;

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
			const int st = clo_radixw_launch_tilehist(cur_in, n, (int) sizeof(E), PB, shift, (mask_hi << LB) | mask_lo,
				thist, tiles, p == 0 ? kx : kx_none, s);
			if (st != 0) return st;
		}
		{
			clo_timing_scope timing("radix_offsets", s);
			const int st = clo_radixw_launch_offsets(PB, thist, tiles, partial, toff, s);
			if (st != 0) return st;
		}
		{
			clo_timing_scope timing("radix_pass", s);
			hipLaunchKernelGGL((clo_radix4_pair_kernel<E, LB, HB>), dim3((tiles + 7u) / 8u * 8u), dim3(pair_shape<E>::THREADS), 0, s,
				cur_in, cur_out, n, shift, mask_lo, mask_hi, (const unsigned*) thist, (const unsigned*) toff,
				(int) ((uintptr_t) cur_in % 16 == 0), p == 0 ? kx : kx_none, p + 1 == passes ? kx : kx_none,
				(unsigned) (getenv("CLO_RP_XF") ? atoi(getenv("CLO_RP_XF")) : 0));
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
	clo_keyx kx, void* ws, hipStream_t s) {
	#define CLO_RP(LB, HB) return rp_sort_impl<E, LB, HB>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, kx, ws, s)
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

template <typename E, int LT>
int r4_dispatch(const void* src, void* dst, void* tmp, size_t n, int key_shift, int key_bits, int digit_bits,
	clo_keyx kx, void* ws, hipStream_t s) {
	switch (digit_bits) {
		case 1: return r4_sort_impl<E, 1, LT>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, kx, ws, s);
		case 2: return r4_sort_impl<E, 2, LT>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, kx, ws, s);
		case 3: return r4_sort_impl<E, 3, LT>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, kx, ws, s);
		case 4: return r4_sort_impl<E, 4, LT>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, kx, ws, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

}  // namespace

size_t clo_radix4_workspace_bytes(size_t n, int elem_size, int key_bits, int digit_bits) {
	const int passes = (key_bits + digit_bits - 1) / digit_bits;
	return r4_make_layout(n, elem_size, passes, digit_bits, 12).total;  // the smaller tile needs more
}

size_t clo_radix4_partition_workspace_bytes(size_t n, int elem_size, int bits) {
	return r4_make_layout(n, elem_size, 2, bits, 12).total;
}

int clo_radix4_partition(const void* src, void* dst, size_t n, int elem_size, unsigned shift, int bits,
	unsigned long long* counts, void* ws, hipStream_t s) {
	#define CLO_R4P(E, B) return r4_partition_impl<E, B>((const E*) src, (E*) dst, n, shift, counts, ws, s)
	if (elem_size == 4) {
		if (bits == 1) CLO_R4P(uint32_t, 1);
		if (bits == 2) CLO_R4P(uint32_t, 2);
		if (bits == 3) CLO_R4P(uint32_t, 3);
	} else if (elem_size == 8) {
		if (bits == 1) CLO_R4P(uint64_t, 1);
		if (bits == 2) CLO_R4P(uint64_t, 2);
		if (bits == 3) CLO_R4P(uint64_t, 3);
	}
	#undef CLO_R4P
	return CLO_HIP_EUNSUPPORTED;
}

void clo_radix4_set_debug_buffer(void* p) { g_r4_dbg = (unsigned long long*) p; }

// static LDS of the two kernels (introspection: clo_sort_get_localmem_usage)
size_t clo_radix4_lds_bytes(const char* kernel, int elem_size, int digit_bits) {
	const size_t R = (size_t) 1 << digit_bits, NW = R >= 2 ? R / 2 : 1;
	if (kernel[0] == 'h') return R4_WAVES * (R >= 2 ? R / 2 : 1) * sizeof(unsigned);
	return ((size_t) 4096 * elem_size) + NW * R4_THREADS * sizeof(unsigned)
		+ (2 * R4_WAVES * NW + 2 * R * R + 2 * R + NW) * sizeof(unsigned);
}

size_t clo_radix4_pair_workspace_bytes(size_t n, int elem_size, int digit_bits) {
	return rp_make_layout(n, elem_size, digit_bits <= 4 ? 2 * digit_bits : digit_bits).total;
}

size_t clo_radix4_pair_lds_bytes(int elem_size, int digit_bits) {
	const size_t threads = elem_size == 8 ? 512 : 1024, hmax = digit_bits >= 7 ? 8 : 4;
	return threads * 8 * (size_t) elem_size + hmax * threads * sizeof(unsigned)
		+ (2 * (threads / 64) * hmax + hmax + ((size_t) 1 << digit_bits) + 4) * sizeof(unsigned);
}

int clo_radix4_pair_sort(const void* src, void* dst, void* tmp, size_t n, int elem_size, int key_shift,
	int key_bits, int digit_bits, clo_keyx kx, void* ws, hipStream_t s) {
	switch (elem_size) {
		case 1: return rp_dispatch<uint8_t>(src, dst, tmp, n, key_shift, key_bits, digit_bits, kx, ws, s);
		case 2: return rp_dispatch<uint16_t>(src, dst, tmp, n, key_shift, key_bits, digit_bits, kx, ws, s);
		case 4: return rp_dispatch<uint32_t>(src, dst, tmp, n, key_shift, key_bits, digit_bits, kx, ws, s);
		case 8: return rp_dispatch<uint64_t>(src, dst, tmp, n, key_shift, key_bits, digit_bits, kx, ws, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

int clo_radix4_sort(const void* src, void* dst, void* tmp, size_t n, int elem_size, int key_shift,
	int key_bits, int digit_bits, clo_keyx kx, void* ws, hipStream_t s) {
	switch (elem_size) {
		case 1: return r4_dispatch<uint8_t, 12>(src, dst, tmp, n, key_shift, key_bits, digit_bits, kx, ws, s);
		case 2: return r4_dispatch<uint16_t, 12>(src, dst, tmp, n, key_shift, key_bits, digit_bits, kx, ws, s);
		case 4: return r4_dispatch<uint32_t, 12>(src, dst, tmp, n, key_shift, key_bits, digit_bits, kx, ws, s);
		case 8: return r4_dispatch<uint64_t, 12>(src, dst, tmp, n, key_shift, key_bits, digit_bits, kx, ws, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}
