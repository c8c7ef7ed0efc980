// clo_hip_radixw.hip — the histogram and counter-scan steps of the LSD radix
// passes on wide digits (5..8 bits, radix 32..256: the "radix=32 … 256"
// options of satradix). The pass kernel itself (two local splits of <= 4 bits,
// clo_radix4_pair_kernel) lives in clo_hip_radix4.hip next to the ranking code.
//
// Same structure per digit as the reference (sort/clo_sort_satradix.c:264-313):
// per-tile digit histogram (upstream's satradix_histogram) -> scan of the
// counters in digit-major order (upstream's counters_sum) -> tile-local stable
// sort + scatter. For R > 16 the next digit's per-tile histogram cannot ride on
// the scatter (the [digit][2][next digit] table of clo_hip_radix4.hip would need
// R*2*R counters), so every pass starts with its own histogram kernel: two
// reads + one write of every element per digit (upstream: ~5 element streams +
// 6 counter streams). No kernel waits on another work-group.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "clo_hip.h"
#include "clo_hip_internal.h"
#include "clo_hip_radix_rank.h"

namespace {

constexpr int RW_CHUNK = 128;   // tiles per chunk of the counter scan (arrays of up to RW_SMALL_TILES tiles: RW_CHUNK_SMALL, below)

// the pair kernel's tile (clo_hip_radix_rank.h), one thread per 16 elements (8 of 8 bytes)
// (Half the threads on the same tile — twice the loads in flight per thread — measured the same: profiles/r05_ab_mid_sizes.txt, lib_h.)
template <typename E, bool BIG> struct rw_shape {
	static constexpr int TILE = pair_shape<E, BIG>::TILE;
	static constexpr int THREADS = pair_shape<E, BIG>::THREADS;
	static constexpr int ITEMS = TILE / THREADS;
};

// The histogram kernels also zero the hand-off words of the counter scan that follows them on the stream
// (clo_radixw_offsets_lb_kernel: chunk sums with a "written" bit, the ticket): word i by the thread that owns
// counter i of the launch — no launch of its own, and the scan of the pass before has long finished with them.
__device__ __forceinline__ void rw_clear(unsigned* __restrict__ clear, unsigned clear_words, unsigned i) {
	if (i < clear_words) clear[i] = 0u;
}


// One word per tile beside its histogram row: 1 when ONE bin holds the whole tile (every element of the
// tile carries the same digit: small keys, a shared prefix, equal keys). The pass kernel reads it with a
// scalar load and sends such a tile past its two local splits (clo_hip_radix4.hip: the split of a
// single-digit tile is its worst case). Decided here because the counts are here: in the pass kernel the
// row sits with 256 threads and asking them costs a barrier per tile (+3 % on uniform keys, measured).
// No thread needs to know more than its own bin: a bin that holds the whole tile says 1, a bin that holds a
// part of it says 0 (several may, all the same value), an empty bin says nothing — exactly one of the first
// two kinds exists in every tile, so the word is always written and never needs clearing or a barrier.
__device__ __forceinline__ void rw_tile_info(unsigned h, unsigned count, unsigned* __restrict__ tinfo, unsigned tile) {
	if (h == count) tinfo[tile] = 1u;
	else if (h != 0u) tinfo[tile] = 0u;
}

// Where a tile of a segmented launch lies (clo_hip_internal.h): one 16-byte load, the same for the whole work-group.
// (bit 31 of count_seg: the tile lies in the launch's SECOND source — a piece that never travelled, clo_hip_radix_sort_segmented2)
__device__ __forceinline__ bool rw_seg_tile(const clo_seg_tile* __restrict__ tdesc, size_t& base, unsigned& count, unsigned tile_elems) {
	const clo_seg_tile td = tdesc[blockIdx.x];
	base = (size_t) td.in_base;
	count = td.count_seg & 0xffffu;
	(void) tile_elems;
	return (td.count_seg >> 31) != 0u;
}

// ---------------------------------------------------------------------------
// per-tile digit histogram (upstream's satradix_histogram job)
// ---------------------------------------------------------------------------
// SEG: a segmented launch — `tdesc` says where the tile lies (n is unused); a segment starts at any element, so the
// vector loads are only element-aligned.
// TPW tiles per work-group (not for segmented launches), one per THREADS / TPW consecutive threads with counters of their own:
// twice the key bytes in flight per thread, half the work-groups. Launched with TPW = 1: with two tiles the histogram over the
// KEYS takes the same time (2^28 uint32: 0.222 ms either way — it reads at the 4.9 TB/s a read-only stream gets on this part;
// profiles/r05_ab_hist_keys_two_tiles.txt), unlike the one over the digit bytes below.
template <typename E, int BITS, bool BIG, bool SEG = false, int TPW = 1>
__global__ __launch_bounds__((rw_shape<E, BIG>::THREADS))
void clo_radixw_tilehist_kernel(const E* __restrict__ in, size_t n, unsigned shift, unsigned mask,
	unsigned* __restrict__ thist, unsigned* __restrict__ tinfo, int aligned, clo_keyx kx,
	unsigned* __restrict__ clear, unsigned clear_words, const clo_seg_tile* __restrict__ tdesc = nullptr, const E* __restrict__ in2 = nullptr) {
	static_assert(!SEG || TPW == 1, "segmented launches: one tile per work-group");
	constexpr int R = 1 << BITS;
	constexpr int TILE = rw_shape<E, BIG>::TILE;
	constexpr int RW_THREADS = rw_shape<E, BIG>::THREADS;
	constexpr int PART = RW_THREADS / TPW;   // threads of one tile
	constexpr int ITEMS = TILE / PART;
	static_assert(PART % 64 == 0 && PART * TPW == RW_THREADS, "whole waves per tile");
	constexpr int VB = ITEMS * (int) sizeof(E) >= 16 ? 16 : ITEMS * (int) sizeof(E);   // bytes per vector load
	constexpr int VECS = ITEMS * (int) sizeof(E) / VB;
	constexpr int PER = VB / (int) sizeof(E);
	// 32 copies of every counter, copy = lane mod 32, bin-major: the 32 lanes an LDS
	// instruction serves together hit 32 different banks and never one address (an
	// LDS add holds its bank for many cycles; one copy per wave, lanes colliding on
	// banks, made the adds a co-bottleneck of this otherwise streaming kernel).
	constexpr int COPIES = 32;
	__shared__ __attribute__((aligned(16))) unsigned s_cnt[TPW * R * COPIES];
	const unsigned tid = threadIdx.x, lane = tid & 63u;
	const unsigned part = tid / PART, ptid = tid % PART;
	const unsigned tile = blockIdx.x * TPW + part;   // this thread's tile (the same for its whole wave)
	size_t base = (size_t) tile * TILE;
	unsigned count = base >= n ? 0u : ((n - base) < (size_t) TILE ? (unsigned) (n - base) : (unsigned) TILE);
	if constexpr (SEG) { if (rw_seg_tile(tdesc, base, count, (unsigned) TILE)) in = in2; }   // (the same for the whole work-group)
	const unsigned tbase = ptid * ITEMS;
	unsigned* const cnt = s_cnt + part * (R * COPIES) + (lane & (COPIES - 1));
	typedef E vecA __attribute__((ext_vector_type(PER)));   // (`aligned`: the source is 16-byte aligned)
	typedef E vecU __attribute__((ext_vector_type(PER), aligned(sizeof(E))));
	typedef typename std::conditional<SEG, vecU, vecA>::type vecE;
	const bool whole = count == (unsigned) TILE && (aligned || SEG);
	// The keys are requested FIRST (round 5): the counters are zeroed and the barrier passed while they are on their way
	// (an LDS-only barrier: a __syncthreads() would wait for the loads). A work-group lives about one load latency; with
	// the loads behind the barrier a CU had its requests in flight less than half of the time.
	vecE v[VECS];
	if (whole) {
		// Lanes read adjacent vectors (a histogram takes its elements in any order; a thread's own 64 bytes, as the pass
		// kernel reads them, are 64 different cache-line halves per instruction): 2^28 uint32 sorts -0.012 ms, round 5.
		const vecE* p = reinterpret_cast<const vecE*>(in + base) + ptid;
		#pragma unroll
		for (int k = 0; k < VECS; ++k) v[k] = p[k * PART];
	}
	{   // (16-byte stores: a quarter of the LDS instructions of a dword loop)
		typedef unsigned vec4u __attribute__((ext_vector_type(4)));
		const vec4u z = { 0u, 0u, 0u, 0u };
		for (unsigned i = tid; i < (unsigned) (TPW * R * COPIES / 4); i += RW_THREADS) reinterpret_cast<vec4u*>(s_cnt)[i] = z;
	}
	clo_lds_barrier();
	if (whole) {
		#pragma unroll
		for (int k = 0; k < VECS; ++k) {
			#pragma unroll
			for (int q = 0; q < PER; ++q)
				atomicAdd(&cnt[((unsigned) (clo_keyx_fwd<E>(v[k][q], kx) >> shift) & mask) << 5], 1u);
		}
	} else {
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i)
			if (tbase + i < count)
				atomicAdd(&cnt[((unsigned) (clo_keyx_fwd<E>(in[base + tbase + i], kx) >> shift) & mask) << 5], 1u);
	}
	__syncthreads();
	for (unsigned i = tid; i < (unsigned) (TPW * R); i += RW_THREADS) {
		typedef unsigned vec4u __attribute__((ext_vector_type(4)));
		const unsigned row_part = i / R, d = i % R, rt = blockIdx.x * TPW + row_part;   // (the row's tile, which may be another part's)
		unsigned rcount = count;
		if constexpr (TPW > 1) {
			const size_t rb = (size_t) rt * TILE;
			if (rb >= n) break;
			rcount = (n - rb) < (size_t) TILE ? (unsigned) (n - rb) : (unsigned) TILE;
		}
		const vec4u* row = reinterpret_cast<const vec4u*>(&s_cnt[row_part * (R * COPIES) + d * COPIES]);
		unsigned h = 0;
		#pragma unroll
		for (int k = 0; k < COPIES / 4; ++k) {   // (rotated: the lanes' rows are 128 bytes apart)
			const vec4u x = row[(k + d) & (COPIES / 4 - 1)];
			h += x[0] + x[1] + x[2] + x[3];
		}
		thist[(size_t) rt * R + d] = h;
		rw_tile_info(h, rcount, tinfo, rt);
		rw_clear(clear, clear_words, rt * R + d);
		if constexpr (SEG) rw_clear(clear, clear_words, (gridDim.x + rt) * R + d);   // (up to one chunk per tile, and the ticket's row)
	}
}

// The same histogram out of the digit bytes the pass before wrote (one byte per
// element, in the order of the elements; clo_radix4_pair_kernel<..., DIG>): a quarter
// (uint32) or an eighth (8-byte elements) of the bytes to read.
// TPW tiles per work-group, one per THREADS / TPW consecutive threads with counters of their own: a tile of
// 8-byte elements is 8 192 digit bytes — one 8-byte load per thread of a 1024-thread group left half as many
// bytes in flight per CU as the 4-byte elements' 16 (150 us per launch against 98 for the same 256 MiB,
// profiles/r04_satradix_u64_kernel_stats.csv); two such tiles share a group, 16 bytes per thread.
template <int BITS, int ITEMS, int THREADS, bool SEG = false, int TPW = 1>   // ITEMS bytes per thread; THREADS / TPW * ITEMS = the tile
__global__ __launch_bounds__(THREADS)
void clo_radixw_tilehist_bytes_kernel(const unsigned char* __restrict__ dig, size_t n, unsigned tiles, unsigned mask, unsigned* __restrict__ thist,
	unsigned* __restrict__ tinfo, unsigned* __restrict__ clear, unsigned clear_words, const clo_seg_tile* __restrict__ tdesc = nullptr) {
	constexpr int R = 1 << BITS;
	constexpr int PART = THREADS / TPW;   // threads of one tile
	constexpr int TILE = PART * ITEMS;
	constexpr int COPIES = 32;
	static_assert(PART % 64 == 0 && PART * TPW == THREADS, "whole waves per tile");
	__shared__ __attribute__((aligned(16))) unsigned s_cnt[TPW * R * COPIES];
	const unsigned tid = threadIdx.x, lane = tid & 63u;
	const unsigned part = tid / PART, ptid = tid % PART;
	const unsigned t = blockIdx.x * TPW + part;   // this thread's tile (the same for its whole wave)
	size_t base = (size_t) t * TILE;
	unsigned count = 0;
	if (t < tiles) {
		count = (n - base) < (size_t) TILE ? (unsigned) (n - base) : (unsigned) TILE;
		if constexpr (SEG) {
			const clo_seg_tile td = tdesc[t];
			base = (size_t) td.in_base;
			count = td.count_seg & 0xffffu;
		}
	}
	const unsigned tbase = ptid * ITEMS;
	unsigned* const cnt = s_cnt + part * (R * COPIES) + (lane & (COPIES - 1));
	typedef unsigned vecA __attribute__((ext_vector_type(ITEMS / 4)));   // (the stream starts 256-byte aligned, a tile is a multiple of 16 bytes)
	typedef unsigned vecU __attribute__((ext_vector_type(ITEMS / 4), aligned(1)));   // (a segment starts at any byte of it)
	typedef typename std::conditional<SEG, vecU, vecA>::type vecw;
	vecw v;
	// (requested before the counters are zeroed: see clo_radixw_tilehist_kernel; lanes read adjacent 16-byte vectors)
	if (count == (unsigned) TILE) {
		typedef unsigned q4A __attribute__((ext_vector_type(4)));
		typedef unsigned q4U __attribute__((ext_vector_type(4), aligned(1)));
		typedef typename std::conditional<SEG, q4U, q4A>::type q4;
		static_assert(ITEMS % 16 == 0, "whole 16-byte vectors per thread");
		const q4* p = reinterpret_cast<const q4*>(dig + base) + ptid;
		#pragma unroll
		for (int k = 0; k < ITEMS / 16; ++k) {
			const q4 x = p[k * PART];
			#pragma unroll
			for (int q = 0; q < 4; ++q) v[k * 4 + q] = x[q];
		}
	}
	{
		typedef unsigned vec4u __attribute__((ext_vector_type(4)));
		const vec4u z = { 0u, 0u, 0u, 0u };
		for (unsigned i = tid; i < (unsigned) (TPW * R * COPIES / 4); i += THREADS) reinterpret_cast<vec4u*>(s_cnt)[i] = z;
	}
	clo_lds_barrier();
	if (count == (unsigned) TILE) {
		#pragma unroll
		for (int k = 0; k < ITEMS / 4; ++k) {
			#pragma unroll
			for (int b = 0; b < 4; ++b)
				atomicAdd(&cnt[(((unsigned) v[k] >> (8 * b)) & mask) << 5], 1u);
		}
	} else {
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i)
			if (tbase + i < count) atomicAdd(&cnt[((unsigned) dig[base + tbase + i] & mask) << 5], 1u);
	}
	__syncthreads();
	for (unsigned i = tid; i < (unsigned) (TPW * R); i += THREADS) {
		typedef unsigned vec4u __attribute__((ext_vector_type(4)));
		const unsigned row_part = i / R, d = i % R, rt = blockIdx.x * TPW + row_part;
		if (rt >= tiles) break;
		const vec4u* row = reinterpret_cast<const vec4u*>(&s_cnt[row_part * (R * COPIES) + d * COPIES]);
		unsigned h = 0;
		#pragma unroll
		for (int k = 0; k < COPIES / 4; ++k) {
			const vec4u x = row[(k + d) & (COPIES / 4 - 1)];
			h += x[0] + x[1] + x[2] + x[3];
		}
		unsigned rcount;   // (the row's tile, which may be another part's)
		if constexpr (SEG) rcount = tdesc[rt].count_seg & 0xffffu;
		else { const size_t rb = (size_t) rt * TILE; rcount = (n - rb) < (size_t) TILE ? (unsigned) (n - rb) : (unsigned) TILE; }
		thist[(size_t) rt * R + d] = h;
		rw_tile_info(h, rcount, tinfo, rt);
		rw_clear(clear, clear_words, rt * R + d);
		if constexpr (SEG) rw_clear(clear, clear_words, (tiles + rt) * R + d);
	}
}

// ---------------------------------------------------------------------------
// counts[tile][digit] -> offsets[tile][digit] in digit-major order:
//   off[t][d] = sum_{d'<d} total[d'] + sum_{t'<t} cnt[t'][d]
// (exactly upstream's exclusive scan of counters[num_wgs*d + wg]). A row of R
// counters is contiguous, thread = digit: every access is coalesced.
//
// This step moves a few MB and is bound by LATENCY. History: round 2 walked the rows (a
// round trip per row: 16-25 us for three launches — sums of chunks of RW_CHUNK tiles, a
// one-work-group scan of the chunk sums, the walk over each chunk's tiles —, a quarter of a
// 2^24-key sort); round 3 first gave every thread group SUB = RW_CHUNK / G consecutive rows,
// requested all at once (13.5 / 16.6 us at 2^24 / 2^28 keys: three launches of a launch + one
// round trip each, profiles/r03_kernel_timeline_2p24.txt), then made it ONE launch:
//
// work-group = chunk, in the order of a ticket. It sums its chunk's rows (all in registers), PUBLISHES the
// chunk's digit sums — one word each, bit 31 = written, agent scope — and reads the sums of every chunk before
// it: those never wait for anything before they publish, and the ticket says they have started, so the wait is
// bounded by their one load and there is nothing to give up on (no chain: chunk c adds up c published rows, thread
// group g the chunks g, g + G, ..., eight requests in flight). What no chunk can know without waiting for LATER
// chunks is the digit bases (the totals of all smaller digits): the chunk with the last ticket has them when it
// is done and leaves them in a row of their own (dbase); the pass kernel adds dbase[digit] to the offset it
// reads — toff holds sum_{t' < t} cnt[t'][d] only. The hand-off words (sums, ticket) are zeroed by the histogram
// kernel that precedes this one on the stream (rw_clear). 2^24 keys 0.240 -> 0.221 ms per sort, 2^25 0.409 ->
// 0.385, 2^28 2.51 -> 2.46 (profiles/r03_counter_scan_one_launch.txt).
// Tried and dropped on the way (profiles/r03_hist_chain_probe.txt, docs/lab_notebook.md): the chunk sums chained onto the
// histogram kernel (write-through rows, arrival counters, the last arrival sums: the histogram kernel, bound by
// its loads in flight, lost 0.07 ms per pass to the arrivals); the chunk scan folded into the offsets kernel
// with every work-group re-deriving its chunk's start from ALL chunk sums, digit bases included (128 KiB of L2
// reads per work-group cost more than the launch they replaced).
//   partial: rows 0 .. chunks-1 the sums, row `chunks` word 0 the ticket, row chunks+1 dbase
// ---------------------------------------------------------------------------
constexpr int RW_CS_THREADS = 1024;
// Arrays of up to RW_SMALL_TILES tiles scan their counters in chunks of RW_CHUNK_SMALL tiles: with 128-tile chunks a 2^24-key
// sort (2 048 tiles) runs this latency-bound kernel on 16 work-groups of a 256-CU chip (round 5, profiles/r05_ab_mid_sizes.txt:
// 2^22 keys 0.101 -> 0.093 ms per sort, 2^24 0.208 -> 0.204; from 4 096 tiles on the larger chunk is the faster one, at 16 384
// tiles by a factor of two — the last chunk adds up the published rows of all earlier ones).
constexpr int RW_CHUNK_SMALL = 32;
constexpr unsigned RW_SMALL_TILES = 2048;
constexpr int rw_chunk_for(unsigned tiles) { return tiles <= RW_SMALL_TILES ? RW_CHUNK_SMALL : RW_CHUNK; }
template <int R, int CHUNK = RW_CHUNK> struct rw_cs {
	static constexpr int G = (RW_CS_THREADS / R) < CHUNK ? (RW_CS_THREADS / R) : CHUNK;   // thread groups per chunk
	static constexpr int SUB = CHUNK / G;                                                  // rows per group
	static_assert(G * SUB == CHUNK && G * R <= RW_CS_THREADS, "the groups tile the chunk");
};

constexpr unsigned RW_WRITTEN = 0x80000000u;
// SEG: the chunks of a segmented launch (cdesc): a chunk's tiles belong to ONE segment, it adds up the published sums of
// the earlier chunks of ITS segment only (they hold earlier tickets, as before), and the last chunk of every segment
// leaves that segment's digit bases in row chunks + 1 + segment.
template <int R, bool SEG = false, int CHUNK = RW_CHUNK>
__global__ __launch_bounds__(RW_CS_THREADS)
void clo_radixw_offsets_lb_kernel(const unsigned* __restrict__ thist, unsigned tiles, unsigned chunks,
	unsigned* __restrict__ partial, unsigned* __restrict__ toff, const clo_seg_chunk* __restrict__ cdesc = nullptr) {
	constexpr int G = rw_cs<R, CHUNK>::G, SUB = rw_cs<R, CHUNK>::SUB;
	constexpr int GA = RW_CS_THREADS / R;   // thread groups of the look-back (all threads)
	constexpr int LB = 8;                   // published rows a thread asks for at once (16: the same times; 32: twice as long — registers)
	__shared__ unsigned s_p[G * R], s_lb[GA * R], s_w[4], s_c;
	const unsigned tid = threadIdx.x, d = tid % R, g = tid / R, lane = tid & 63u, wave = tid >> 6;
	if (tid == 0) s_c = __hip_atomic_fetch_add(&partial[(size_t) chunks * R], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	// (Requesting the rows of chunk blockIdx.x while the ticket is on its way, to load again when the ticket
	// names another chunk, was slower — 7.9 instead of 5.9 us at 2^24 keys: work-groups of different XCDs do not
	// start in the order of their numbers often enough.)
	__syncthreads();
	const unsigned c = s_c;
	unsigned t0 = c * CHUNK;
	unsigned tend = t0 + CHUNK < tiles ? t0 + CHUNK : tiles;
	unsigned c_first = 0u, dbase_row = chunks + 1u;
	bool last = c + 1u == chunks;
	if constexpr (SEG) {
		const clo_seg_chunk cd = cdesc[c];
		t0 = cd.t0; tend = cd.tend; c_first = cd.c_first;
		last = (cd.seg_last >> 31) != 0u;
		dbase_row = chunks + 1u + (cd.seg_last & 0x7fffffffu);
	}
	const bool active = g < (unsigned) G;
	unsigned v[SUB];
	if (active) {
		#pragma unroll
		for (int k = 0; k < SUB; ++k) {
			const unsigned t = t0 + g * SUB + k;
			v[k] = t < tend ? thist[(size_t) t * R + d] : 0u;
		}
		unsigned own = 0;
		#pragma unroll
		for (int k = 0; k < SUB; ++k) own += v[k];
		s_p[g * R + d] = own;
	}
	__syncthreads();
	unsigned tot = 0;
	if (tid < (unsigned) R) {
		#pragma unroll 8
		for (int k = 0; k < G; ++k) tot += s_p[k * R + tid];
		__hip_atomic_store(&partial[(size_t) c * R + tid], tot | RW_WRITTEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	// the chunks before this one
	unsigned acc = 0;
	for (unsigned b = c_first + g; b < c; b += (unsigned) (GA * LB)) {
		unsigned x[LB];
		bool ok;
		do {
			#pragma unroll
			for (int k = 0; k < LB; ++k) {
				const unsigned cc = b + (unsigned) (k * GA);
				x[k] = cc < c ? __hip_atomic_load(&partial[(size_t) cc * R + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : RW_WRITTEN;
			}
			ok = true;
			#pragma unroll
			for (int k = 0; k < LB; ++k) ok = ok && (x[k] & RW_WRITTEN) != 0u;
			if (!ok) __builtin_amdgcn_s_sleep(1);
		} while (!ok);
		#pragma unroll
		for (int k = 0; k < LB; ++k) acc += x[k] & ~RW_WRITTEN;
	}
	s_lb[g * R + d] = acc;
	__syncthreads();
	unsigned before = 0;
	#pragma unroll 8
	for (int k = 0; k < GA; ++k) before += s_lb[k * R + d];
	if (last) {   // (the same for the whole work-group) the digit bases: exclusive scan of the totals over the digits
		const unsigned t = tid < (unsigned) R ? before + tot : 0u;
		const unsigned incl = clo_wave_scan_inclusive<unsigned>(t, lane);
		if (lane == 63 && wave < 4) s_w[wave] = incl;   // R <= 256: the digits sit in the first four waves
		__syncthreads();
		if (tid < (unsigned) R) {
			unsigned dbase = incl - t;
			#pragma unroll
			for (unsigned w = 0; w < 4; ++w) if (w < wave) dbase += s_w[w];
			partial[(size_t) dbase_row * R + tid] = dbase;
		}
	}
	if (!active) return;
	unsigned run = before;
	for (unsigned k = 0; k < g; ++k) run += s_p[k * R + d];
	#pragma unroll
	for (int k = 0; k < SUB; ++k) {
		const unsigned t = t0 + g * SUB + k;
		if (t < tend) toff[(size_t) t * R + d] = run;
		run += v[k];
	}
}

// One tile (nothing to hand over, nobody to zero a ticket): the scan in one work-group, as in round 2.
template <int R>
__global__ __launch_bounds__(256)
void clo_radixw_offsets1_kernel(const unsigned* __restrict__ thist, unsigned tiles, unsigned* __restrict__ toff) {
	constexpr int G = 256 / R;
	constexpr int SUB = RW_CHUNK / G;
	__shared__ unsigned s_a[G][R], s_base[R], s_w[4];
	const unsigned tid = threadIdx.x, d = tid % R, g = tid / R, lane = tid & 63u, wave = tid >> 6;
	const unsigned ts = g * SUB < tiles ? g * SUB : tiles;
	const unsigned te = ts + SUB < tiles ? ts + SUB : tiles;
	unsigned own = 0;
	#pragma unroll 8
	for (unsigned t = ts; t < te; ++t) own += thist[(size_t) t * R + d];
	s_a[g][d] = own;
	__syncthreads();
	unsigned before = 0, tot = 0;
	#pragma unroll
	for (int k = 0; k < G; ++k) {
		const unsigned v = s_a[k][d];
		if ((unsigned) k < g) before += v;
		tot += v;
	}
	const unsigned t0 = tid < (unsigned) R ? tot : 0u;
	const unsigned incl = clo_wave_scan_inclusive<unsigned>(t0, lane);
	if (lane == 63) s_w[wave] = incl;
	__syncthreads();
	if (tid < (unsigned) R) {
		unsigned dbase = incl - t0;
		#pragma unroll
		for (unsigned w = 0; w < 4; ++w) if (w < wave) dbase += s_w[w];
		s_base[tid] = dbase;
	}
	__syncthreads();
	unsigned run = s_base[d] + before;
	#pragma unroll 8
	for (unsigned t = ts; t < te; ++t) {
		const unsigned c = thist[(size_t) t * R + d];
		toff[(size_t) t * R + d] = run;
		run += c;
	}
}

// ---------------------------------------------------------------------------
// Segmented sorts: the two tables of a launch (clo_hip_internal.h), from the segments' lengths. Every work-group
// scans the (at most 256) lengths itself — a few hundred LDS operations — and then describes one tile and one chunk
// per thread: one launch per sort, whatever the number of tiles.
// ---------------------------------------------------------------------------
// (Tiles start wherever their piece starts: 16-byte loads at addresses that are only element-aligned measured no
// slower — giving every misaligned piece a short first tile that ends on a 16-element boundary of memory, so that all
// later tiles load aligned, changed nothing: 2^28 uint32 keys in 256 segments, 0.593 -> 0.608 ms per pass, the partial
// tiles it adds cost more than the alignment buys.)
constexpr int SEG_BUILD_THREADS = 1024;
__global__ __launch_bounds__(SEG_BUILD_THREADS)
void clo_radix_seg_build_kernel(clo_seg_pieces pc, unsigned npieces, unsigned tile, unsigned ntiles, unsigned nchunks,
	clo_seg_tile* __restrict__ tdesc, clo_seg_chunk* __restrict__ cdesc) {
	// per piece: first tile (s_pt); per segment: length (s_sn), first element in the output (s_ob), first tile (s_st), first chunk (s_sc)
	__shared__ unsigned s_pt[CLO_SEG_MAX + 1], s_sn[CLO_SEG_MAX], s_stl[CLO_SEG_MAX], s_ob[CLO_SEG_MAX + 1], s_st[CLO_SEG_MAX + 1], s_sc[CLO_SEG_MAX + 1], s_w[4][4];
	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	if (tid < (unsigned) CLO_SEG_MAX) { s_sn[tid] = 0u; s_stl[tid] = 0u; }
	__syncthreads();
	unsigned pn = 0, ptl = 0;
	if (tid < npieces) {
		pn = pc.n[tid];
		ptl = (pn + tile - 1u) / tile;
		atomicAdd(&s_sn[pc.seg[tid] & 0x7fffu], pn);   // (bit 15: the piece lies in the second source)
		atomicAdd(&s_stl[pc.seg[tid] & 0x7fffu], ptl);
	}
	__syncthreads();
	if (tid < (unsigned) CLO_SEG_MAX) {
		const unsigned sn = s_sn[tid], stl = s_stl[tid], sch = (stl + (unsigned) RW_CHUNK - 1u) / (unsigned) RW_CHUNK;
		const unsigned i0 = clo_wave_scan_inclusive<unsigned>(ptl, lane), i1 = clo_wave_scan_inclusive<unsigned>(sn, lane),
			i2 = clo_wave_scan_inclusive<unsigned>(stl, lane), i3 = clo_wave_scan_inclusive<unsigned>(sch, lane);
		if (lane == 63u) { s_w[0][wave] = i0; s_w[1][wave] = i1; s_w[2][wave] = i2; s_w[3][wave] = i3; }
		s_pt[tid + 1] = i0; s_ob[tid + 1] = i1; s_st[tid + 1] = i2; s_sc[tid + 1] = i3;
	}
	__syncthreads();
	if (tid < (unsigned) CLO_SEG_MAX) {
		unsigned b0 = 0, b1 = 0, b2 = 0, b3 = 0;
		for (unsigned w = 0; w < wave; ++w) { b0 += s_w[0][w]; b1 += s_w[1][w]; b2 += s_w[2][w]; b3 += s_w[3][w]; }
		s_pt[tid + 1] += b0; s_ob[tid + 1] += b1; s_st[tid + 1] += b2; s_sc[tid + 1] += b3;
		if (tid == 0) s_pt[0] = s_ob[0] = s_st[0] = s_sc[0] = 0u;
	}
	__syncthreads();
	const unsigned i = blockIdx.x * SEG_BUILD_THREADS + tid;
	if (i < ntiles) {   // the piece p with s_pt[p] <= i < s_pt[p + 1] (empty pieces have empty ranges)
		unsigned lo = 0, hi = CLO_SEG_MAX;
		while (hi - lo > 1u) { const unsigned mid = (lo + hi) >> 1; if (s_pt[mid] <= i) lo = mid; else hi = mid; }
		const unsigned off = (i - s_pt[lo]) * tile, left = pc.n[lo] - off, sg = pc.seg[lo] & 0x7fffu, from2 = pc.seg[lo] >> 15;
		clo_seg_tile td;
		td.in_base = pc.in_base[lo] + off;
		td.count_seg = (left < tile ? left : tile) | (sg << 16) | (from2 << 31);
		td.out_base = s_ob[sg];
		td.seg_n = s_sn[sg];
		tdesc[i] = td;
	}
	if (i < nchunks) {
		unsigned lo = 0, hi = CLO_SEG_MAX;
		while (hi - lo > 1u) { const unsigned mid = (lo + hi) >> 1; if (s_sc[mid] <= i) lo = mid; else hi = mid; }
		clo_seg_chunk cd;
		cd.t0 = s_st[lo] + (i - s_sc[lo]) * (unsigned) RW_CHUNK;
		cd.tend = cd.t0 + (unsigned) RW_CHUNK < s_st[lo + 1] ? cd.t0 + (unsigned) RW_CHUNK : s_st[lo + 1];
		cd.c_first = s_sc[lo];
		cd.seg_last = lo | (i + 1u == s_sc[lo + 1] ? 0x80000000u : 0u);
		cdesc[i] = cd;
	}
}

}  // namespace

// ---- segmented launches ----
void clo_radixw_seg_bounds(size_t numel, int npieces, int nseg, size_t tile, size_t* max_tiles, size_t* max_chunks) {
	// every piece may end in a partial tile, and a segment's tiles in a partial chunk
	*max_tiles = numel / tile + (size_t) npieces;
	*max_chunks = *max_tiles / RW_CHUNK + (size_t) nseg;
}

int clo_radixw_seg_build(const size_t* piece_n, const size_t* piece_base, const int* piece_seg, const int* piece_src, int npieces, int nseg, size_t tile,
	clo_seg_tile* tiles, clo_seg_chunk* chunks, unsigned* ntiles, unsigned* nchunks, hipStream_t s) {
	if (nseg < 1 || nseg > CLO_SEG_MAX || npieces < 1 || npieces > CLO_SEG_MAX) return CLO_HIP_EARGS;
	clo_seg_pieces c;
	size_t seg_tiles[CLO_SEG_MAX];
	for (int k = 0; k < CLO_SEG_MAX; ++k) seg_tiles[k] = 0;
	size_t nt = 0, nc = 0, total = 0;
	int prev = 0;
	for (int i = 0; i < CLO_SEG_MAX; ++i) {
		const size_t n = i < npieces ? piece_n[i] : 0;
		const int sg = i < npieces ? piece_seg[i] : prev;
		if (n > 0xffffffffull || sg < prev || sg >= nseg) return CLO_HIP_EARGS;   // (pieces come in segment order)
		prev = sg;
		c.n[i] = (unsigned) n;
		c.in_base[i] = i < npieces ? (unsigned) piece_base[i] : 0u;
		c.seg[i] = (unsigned short) (sg | ((i < npieces && piece_src && piece_src[i]) ? 0x8000 : 0));
		const size_t t = (n + tile - 1) / tile;
		seg_tiles[sg] += t;
		nt += t;
		total += n;
	}
	for (int k = 0; k < nseg; ++k) nc += (seg_tiles[k] + RW_CHUNK - 1) / RW_CHUNK;
	if (total > 0xffffffffull) return CLO_HIP_EARGS;
	*ntiles = (unsigned) nt;
	*nchunks = (unsigned) nc;
	if (nt == 0) return 0;
	hipLaunchKernelGGL(clo_radix_seg_build_kernel, dim3((unsigned) ((nt + SEG_BUILD_THREADS - 1) / SEG_BUILD_THREADS)), dim3(SEG_BUILD_THREADS), 0, s,
		c, (unsigned) npieces, (unsigned) tile, (unsigned) nt, (unsigned) nc, tiles, chunks);
	return (int) hipGetLastError();
}

size_t clo_radixw_partial_rows_seg(size_t chunks, size_t nseg) { return chunks + 1 + nseg; }

template <typename E>
static int rw_launch_tilehist_seg(const void* in, const void* in2, const clo_seg_tables& sg, int bits, unsigned shift, unsigned mask, unsigned* thist,
	unsigned* tinfo, unsigned* partial, bool big, hipStream_t s) {
	const unsigned clear_words = (sg.nchunks + 1u) << bits;
	const clo_keyx kx_none = { 0, 0, 0 };
	if (bits != 8) return CLO_HIP_EUNSUPPORTED;   // (the segmented sorts run the radix-16 / 256 schedule only)
	if (big) hipLaunchKernelGGL((clo_radixw_tilehist_kernel<E, 8, true, true>), dim3(sg.ntiles), dim3(rw_shape<E, true>::THREADS), 0, s,
		(const E*) in, (size_t) 0, shift, mask, thist, tinfo, 0, kx_none, partial, clear_words, sg.tiles, (const E*) in2);
	else hipLaunchKernelGGL((clo_radixw_tilehist_kernel<E, 8, false, true>), dim3(sg.ntiles), dim3(rw_shape<E, false>::THREADS), 0, s,
		(const E*) in, (size_t) 0, shift, mask, thist, tinfo, 0, kx_none, partial, clear_words, sg.tiles, (const E*) in2);
	return (int) hipGetLastError();
}

int clo_radixw_launch_tilehist_seg(const void* in, const void* in2, const clo_seg_tables& sg, int elem_size, int bits, unsigned shift, unsigned mask,
	unsigned* thist, unsigned* tinfo, unsigned* partial, bool big, hipStream_t s) {
	switch (elem_size) {
		case 4: return rw_launch_tilehist_seg<uint32_t>(in, in2, sg, bits, shift, mask, thist, tinfo, partial, big, s);
		case 8: return rw_launch_tilehist_seg<uint64_t>(in, in2, sg, bits, shift, mask, thist, tinfo, partial, big, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

int clo_radixw_launch_tilehist_bytes_seg(const unsigned char* dig, const clo_seg_tables& sg, int elem_size, int bits, unsigned mask,
	unsigned* thist, unsigned* tinfo, unsigned* partial, bool big, hipStream_t s) {
	const unsigned clear_words = (sg.nchunks + 1u) << bits;
	if (bits != 8 || !big) return CLO_HIP_EUNSUPPORTED;
	if (elem_size == 8) hipLaunchKernelGGL((clo_radixw_tilehist_bytes_kernel<8, 16, 1024, true, 2>), dim3((sg.ntiles + 1u) / 2u), dim3(1024), 0, s,
		dig, (size_t) 0, sg.ntiles, mask, thist, tinfo, partial, clear_words, sg.tiles);
	else if (elem_size == 4) hipLaunchKernelGGL((clo_radixw_tilehist_bytes_kernel<8, 32, 1024, true, 2>), dim3((sg.ntiles + 1u) / 2u), dim3(1024), 0, s,
		dig, (size_t) 0, sg.ntiles, mask, thist, tinfo, partial, clear_words, sg.tiles);
	else return CLO_HIP_EUNSUPPORTED;
	return (int) hipGetLastError();
}

int clo_radixw_launch_offsets_seg(int bits, const unsigned* thist, const clo_seg_tables& sg, unsigned* partial, unsigned* toff,
	const unsigned** dbase, hipStream_t s) {
	if (bits != 8) return CLO_HIP_EUNSUPPORTED;
	hipLaunchKernelGGL((clo_radixw_offsets_lb_kernel<256, true>), dim3(sg.nchunks), dim3(RW_CS_THREADS), 0, s,
		thist, sg.ntiles, sg.nchunks, partial, toff, sg.chunks);
	*dbase = partial + ((size_t) (sg.nchunks + 1u) << bits);
	return (int) hipGetLastError();
}

// ---- the histogram / counter-scan steps for any digit width 1..8 (used by
// the digit-pair passes of clo_hip_radix4.hip) ----

unsigned clo_radixw_clear_words(int bits, unsigned tiles);

template <typename E>
static int rw_launch_tilehist(const void* in, size_t n, int bits, unsigned shift, unsigned mask, unsigned* thist, unsigned* tinfo,
	unsigned* partial, unsigned tiles, bool big, clo_keyx kx, hipStream_t s) {
	const int aligned = (int) ((uintptr_t) in % 16 == 0);
	const unsigned clear_words = partial ? clo_radixw_clear_words(bits, tiles) : 0u;
	#define CLO_RW_TH(B) case B: \
		if (big && sizeof(E) >= 4) hipLaunchKernelGGL((clo_radixw_tilehist_kernel<E, B, (sizeof(E) >= 4)>), dim3(tiles), dim3(rw_shape<E, (sizeof(E) >= 4)>::THREADS), 0, s, \
			(const E*) in, n, shift, mask, thist, tinfo, aligned, kx, partial, clear_words); \
		else hipLaunchKernelGGL((clo_radixw_tilehist_kernel<E, B, false>), dim3(tiles), dim3(rw_shape<E, false>::THREADS), 0, s, \
			(const E*) in, n, shift, mask, thist, tinfo, aligned, kx, partial, clear_words); \
		break
	switch (bits) {
		CLO_RW_TH(1); CLO_RW_TH(2); CLO_RW_TH(3); CLO_RW_TH(4); CLO_RW_TH(5); CLO_RW_TH(6); CLO_RW_TH(7); CLO_RW_TH(8);
		default: return CLO_HIP_EUNSUPPORTED;
	}
	#undef CLO_RW_TH
	return (int) hipGetLastError();
}

// `big`: the tiles are those of clo_radix_big_tiles(n, elem_size) (clo_hip_radix_rank.h); `partial`: the
// counter scan's workspace (clo_radixw_launch_offsets follows on the same stream), whose hand-off words this launch zeroes
int clo_radixw_launch_tilehist(const void* in, size_t n, int elem_size, int bits, unsigned shift, unsigned mask,
	unsigned* thist, unsigned* tinfo, unsigned* partial, unsigned tiles, bool big, clo_keyx kx, hipStream_t s) {
	switch (elem_size) {
		case 1: return rw_launch_tilehist<uint8_t>(in, n, bits, shift, mask, thist, tinfo, partial, tiles, big, kx, s);
		case 2: return rw_launch_tilehist<uint16_t>(in, n, bits, shift, mask, thist, tinfo, partial, tiles, big, kx, s);
		case 4: return rw_launch_tilehist<uint32_t>(in, n, bits, shift, mask, thist, tinfo, partial, tiles, big, kx, s);
		case 8: return rw_launch_tilehist<uint64_t>(in, n, bits, shift, mask, thist, tinfo, partial, tiles, big, kx, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

// Histograms out of the digit stream (tiles of the shape `big` names).
int clo_radixw_launch_tilehist_bytes(const unsigned char* dig, size_t n, int elem_size, int bits, unsigned mask,
	unsigned* thist, unsigned* tinfo, unsigned* partial, unsigned tiles, bool big, hipStream_t s) {
	const unsigned clear_words = partial ? clo_radixw_clear_words(bits, tiles) : 0u;
	#define CLO_RW_THB1(B, I, T, W) hipLaunchKernelGGL((clo_radixw_tilehist_bytes_kernel<B, I, T, false, W>), dim3((tiles + W - 1u) / W), dim3(T), 0, s, dig, n, tiles, mask, thist, tinfo, partial, clear_words)
	/* 4-byte elements (round 5): two 16 384-byte tiles per work-group as well, 32 bytes per thread in flight (histograms -3 %, 2^26 / 2^27 sorts -0.8 %) */
	#define CLO_RW_THB4(B) CLO_RW_THB1(B, 32, 1024, 2u)
	#define CLO_RW_THB(B) case B: \
		if (!big) return CLO_HIP_EUNSUPPORTED;   /* (the stream goes with the big tiles) */ \
		if (elem_size == 8) CLO_RW_THB1(B, 16, 1024, 2u); else CLO_RW_THB4(B); \
		break
	if (elem_size != 4 && elem_size != 8) return CLO_HIP_EUNSUPPORTED;
	switch (bits) {
		CLO_RW_THB(1); CLO_RW_THB(2); CLO_RW_THB(3); CLO_RW_THB(4); CLO_RW_THB(5); CLO_RW_THB(6); CLO_RW_THB(7); CLO_RW_THB(8);
		default: return CLO_HIP_EUNSUPPORTED;
	}
	#undef CLO_RW_THB
	#undef CLO_RW_THB4
	#undef CLO_RW_THB1
	return (int) hipGetLastError();
}

// Words of `partial` the histogram launch zeroes for the scan: the chunk sums and the ticket's row (0: one tile, nothing to hand over).
unsigned clo_radixw_clear_words(int bits, unsigned tiles) {
	const unsigned chunk = (unsigned) rw_chunk_for(tiles), chunks = (tiles + chunk - 1) / chunk;
	return tiles > 1 ? ((chunks + 1u) << bits) : 0u;   // (chunks + 1 <= tiles: every word has a thread of the launch)
}
// Rows of (1 << bits) words `partial` needs.
size_t clo_radixw_partial_rows(size_t tiles) { const size_t chunk = (size_t) rw_chunk_for((unsigned) (tiles > 0xffffffffull ? 0xffffffffull : tiles)); return (tiles + chunk - 1) / chunk + 2; }

// *dbase: null — toff holds the final offsets (one tile) —, or the row of digit bases the consumer adds to toff[tile][digit].
int clo_radixw_launch_offsets(int bits, const unsigned* thist, unsigned tiles, unsigned* partial, unsigned* toff,
	const unsigned** dbase, hipStream_t s) {
	const unsigned chunk = (unsigned) rw_chunk_for(tiles);
	const unsigned chunks = (tiles + chunk - 1) / chunk;
	*dbase = nullptr;
	#define CLO_RW_OFF1(B) case B: hipLaunchKernelGGL((clo_radixw_offsets1_kernel<(1 << B)>), dim3(1), dim3(256), 0, s, thist, tiles, toff); break
	if (tiles == 1) {
		switch (bits) {
			CLO_RW_OFF1(1); CLO_RW_OFF1(2); CLO_RW_OFF1(3); CLO_RW_OFF1(4); CLO_RW_OFF1(5); CLO_RW_OFF1(6); CLO_RW_OFF1(7); CLO_RW_OFF1(8);
			default: return CLO_HIP_EUNSUPPORTED;
		}
		return (int) hipGetLastError();
	}
	#undef CLO_RW_OFF1
	#define CLO_RW_LB(B) case B: \
		if (chunk == (unsigned) RW_CHUNK) hipLaunchKernelGGL((clo_radixw_offsets_lb_kernel<(1 << B), false, RW_CHUNK>), dim3(chunks), dim3(RW_CS_THREADS), 0, s, thist, tiles, chunks, partial, toff); \
		else hipLaunchKernelGGL((clo_radixw_offsets_lb_kernel<(1 << B), false, RW_CHUNK_SMALL>), dim3(chunks), dim3(RW_CS_THREADS), 0, s, thist, tiles, chunks, partial, toff); \
		break
	switch (bits) {
		CLO_RW_LB(1); CLO_RW_LB(2); CLO_RW_LB(3); CLO_RW_LB(4); CLO_RW_LB(5); CLO_RW_LB(6); CLO_RW_LB(7); CLO_RW_LB(8);
		default: return CLO_HIP_EUNSUPPORTED;
	}
	#undef CLO_RW_LB
	*dbase = partial + ((size_t) (chunks + 1u) << bits);
	return (int) hipGetLastError();
}

// Loads this file's code object (HIP loads one at the first launch that needs it — 2.4 ms inside the first chain-free
// pass of a process, which the sorter's warm-up sorts are too small to reach): asking for a kernel's attributes does it.
int clo_radixw_preload() {
	hipFuncAttributes a;
	return (int) hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&clo_radix_seg_build_kernel));
}

size_t clo_radixw_lds_bytes(int digit_bits) { return 32 * ((size_t) 1 << digit_bits) * sizeof(unsigned); }
