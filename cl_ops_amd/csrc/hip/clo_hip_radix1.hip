// clo_hip_radix1.hip — LSD radix passes that read every element ONCE: the
// per-tile histogram stream of the chain-free passes (clo_hip_radix4.hip: a
// separate kernel re-reads the whole array per pass) is replaced by digit
// counts handed from tile to tile inside the pass kernel.
//
// Per sort: one kernel reads the source once and counts, for every pass, the
// 256 global digit totals (their exclusive scan = where each digit's output
// starts); then one "sweep" kernel per 8 key bits. Per tile it does what
// upstream's localsort + histogram + scan + scatter do per digit
// (sort/clo_sort_satradix.c:264-313), for two 4-bit digits at once:
//   load -> tile histogram of the combined digit (LDS counters) -> PUBLISH it
//   -> two stable local splits (clo_hip_radix_rank.h) -> look back for the
//   count of every digit in all earlier tiles -> scatter.
// Output position of an element = global digit start + count of its digit in
// earlier tiles + its rank inside the tile: upstream's
// counters_sum[num_wgs*d + wg] + lid - offsets[d] (sort/clo_sort_satradix.cl:253),
// so the order is the one upstream produces (stable, ascending).
//
// The look-back is two-level because a dependent memory round trip costs 1-3 us
// on this chip while tiles start every ~15 ns: a tile-to-tile chain (each tile
// walking back over single predecessors) resolves ~1 tile per round trip and
// measured 6.4 ms per sort in round 1. Tiles form chunks of 16:
//   level 1  a tile sums the published histograms of the earlier tiles of its
//            own chunk (<= 15 rows, all requested at once, no chain);
//   level 2  every tile also adds its histogram to the chunk's accumulator with
//            a RETURNING atomic; the arrival that completes a (chunk, digit)
//            looks back over a WINDOW of earlier chunks (their accumulators /
//            inclusive prefixes, all requested at once) and publishes the
//            chunk's inclusive prefix; a tile needs the prefix of the chunk
//            before its own.
// One hop of level 2 covers 16 * 8 = 128 tiles. Tile numbers are tickets (HIP
// promises no dispatch order: a tile may only wait for tiles already handed
// out). Two ways of drawing them (r1_pass::pools):
//   1 pool   ONE counter, tickets = tile numbers: whatever the device looks like,
//            every tile a work-group can be waiting for has been handed out. The
//            library's default (it takes this path for 4 .. 256 tiles).
//   8 pools  one counter per XCD, chunk j of pool x = global chunk 8*j + x, the
//            XCD read from the hardware (XCC_ID): the 16 tiles of a chunk run
//            behind one L2, where the boundary lines of neighbouring runs merge
//            (clo_hip_radix4.hip has the measurements). A tile of pool x waits for
//            the chunk before its own, which belongs to pool x - 1: forward
//            progress then DEPENDS on all eight pools being drawn from side by
//            side, i.e. on work-groups being resident on all eight XCDs. With
//            every work-group reporting one XCC_ID (a partitioned device, CU
//            masking) and more tiles in a pool than resident work-groups, the
//            resident ones would hold pool-x tiles that wait for chunks of pools
//            nobody draws from, until the bounded spins give up. Used only where
//            that cannot happen by construction: a device with all 256 CUs (8
//            XCDs, >= 768 resident work-groups), more than 1024 tiles, i.e. the
//            forced sweeps of the A/B runs (CLO_RADIX_SWEEP=1) — and a work-group
//            whose own pool is used up helps the others.
// Every poll loop is bounded (status word, as the scan).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "clo_hip.h"
#include "clo_hip_internal.h"
#include "clo_hip_radix_rank.h"

namespace {

constexpr int R1_PEEL = 4;                           // groups of equal digits a skewed wave counts by ballot, per element
constexpr int R1_WINDOW = 8;                         // chunks a level-2 hop inspects
constexpr unsigned R1_VALID = 0x80000000u;            // 32-bit entries: bit 31 = written, bits 30..0 = count
constexpr int R1_CNT_SHIFT = 40;                      // 64-bit accumulators: arrivals << 40 | sum
constexpr clo_u64 R1_SUM_MASK = (1ull << R1_CNT_SHIFT) - 1ull;
constexpr int R1_ROW = 256;                           // counters per tile / chunk row
constexpr int R1_POOLS = 8;
constexpr int R1_TICKET_STRIDE = 16;                  // words between the pools' ticket counters (one 64-byte line each)
constexpr int R1_GH_THREADS = 512;

__device__ __forceinline__ unsigned r1_ld32(const unsigned* p) {
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void r1_st32(unsigned* p, unsigned v) {
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------
// Global digit totals of every pass: one read of the source.
// ---------------------------------------------------------------------------
template <typename E, int NP>
__global__ __launch_bounds__(R1_GH_THREADS)
void clo_radix1_ghist_kernel(const E* __restrict__ in, size_t n, unsigned key_shift, unsigned key_bits,
	unsigned* __restrict__ ghist, unsigned* __restrict__ gbase, unsigned* __restrict__ done,
	int aligned, clo_keyx kx, unsigned tiles_per_group) {
	constexpr int ITEMS = sweep_shape<E>::ITEMS;
	constexpr int TILE = R1_GH_THREADS * ITEMS;
	// Counters: 16-bit, two digits per dword, COPIES copies of every dword (copy = lane
	// mod COPIES, dword-major): with 32 copies the 32 lanes an LDS instruction serves
	// together never share a bank (see the sweep kernel). A work-group counts at most
	// 16 tiles before it flushes: <= 8192 keys per copy, so 16 bits hold any count.
	constexpr int COPIES = NP <= 4 ? 32 : 16;   // 16 KiB (8 KiB) of LDS per pass
	__shared__ unsigned s_cnt[NP * (R1_ROW / 2) * COPIES];
	const unsigned tid = threadIdx.x, cp = tid & (COPIES - 1);
	for (unsigned i = tid; i < (unsigned) (NP * (R1_ROW / 2) * COPIES); i += R1_GH_THREADS) s_cnt[i] = 0;
	__syncthreads();
	const unsigned last_bits = key_bits - 8u * (NP - 1);
	const unsigned last_mask = (1u << last_bits) - 1u;
	const size_t tile0 = (size_t) blockIdx.x * tiles_per_group;
	for (unsigned t = 0; t < tiles_per_group; ++t) {
		const size_t base = (tile0 + t) * TILE;
		if (base >= n) break;
		const unsigned count = (n - base) < (size_t) TILE ? (unsigned) (n - base) : (unsigned) TILE;
		const unsigned tbase = tid * ITEMS;
		E key[ITEMS];
		if (count == (unsigned) TILE) {
			load_blocked<E, ITEMS>(in + base + tbase, key, aligned != 0);
		} else {
			#pragma unroll
			for (int i = 0; i < ITEMS; ++i) key[i] = (tbase + i < count) ? in[base + tbase + i] : (E) 0;
		}
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) {
			if (count == (unsigned) TILE || tbase + i < count) {
				const E k = clo_keyx_fwd<E>(key[i], kx) >> key_shift;
				#pragma unroll
				for (int p = 0; p < NP; ++p) {
					const unsigned d = (unsigned) (k >> (8 * p)) & (p == NP - 1 ? last_mask : 255u);
					atomicAdd(&s_cnt[((unsigned) p * (R1_ROW / 2) + (d >> 1)) * COPIES + cp], 1u << ((d & 1u) * 16u));
				}
			}
		}
	}
	__syncthreads();
	for (unsigned i = tid; i < (unsigned) (NP * R1_ROW / 2); i += R1_GH_THREADS) {   // i: (pass, digit pair)
		unsigned even = 0, odd = 0;
		#pragma unroll
		for (int c = 0; c < COPIES; ++c) {
			const unsigned x = s_cnt[i * COPIES + ((c + i) & (COPIES - 1))];   // (rotated: rows are whole bank rounds apart)
			even += x & 0xffffu;
			odd += x >> 16;
		}
		if (even) atomicAdd(&ghist[2 * i], even);
		if (odd) atomicAdd(&ghist[2 * i + 1], odd);
	}
	// The work-group that finishes last turns the totals into digit starts (an
	// exclusive scan over the 256 digits of every pass): no kernel of its own.
	// (The adds are performed at the memory side and counted in vmcnt until they
	// are: waiting for them orders them before the arrival; the totals are read
	// back with agent-scope loads. A release fence here writes the XCD's L2 back once
	// per group: measured 0.4 ms on the 2^28-key sort.)
	__shared__ unsigned s_last, s_w[4];
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (tid == 0) s_last = atomicAdd(done, 1u) == gridDim.x - 1u;
	__syncthreads();
	if (!s_last) return;
	const unsigned lane = tid & 63u, wave = tid >> 6;
	for (int p = 0; p < NP; ++p) {
		unsigned v = 0, incl = 0;
		if (tid < (unsigned) R1_ROW) {
			v = __hip_atomic_load(&ghist[p * R1_ROW + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			incl = clo_wave_scan_inclusive<unsigned>(v, lane);
			if (lane == 63) s_w[wave] = incl;
		}
		__syncthreads();
		if (tid < (unsigned) R1_ROW) {
			unsigned b = incl - v;
			#pragma unroll
			for (unsigned w = 0; w < 4; ++w) if (w < wave) b += s_w[w];
			gbase[p * R1_ROW + tid] = b;
		}
		__syncthreads();
	}
}

// ---------------------------------------------------------------------------
// The sweep kernel: one pass on 8 key bits (two local digits of <= 4 bits).
// ---------------------------------------------------------------------------
struct r1_pass {
	unsigned* agg;           // [tiles][256]   tile histograms: R1_VALID | count
	clo_u64* cacc;           // [chunks][256]  arrivals << 40 | sum of the chunk's tile histograms
	unsigned* cprefix;       // [chunks][256]  R1_VALID | count of the digit in all tiles up to the end of the chunk
	unsigned* ticket;        // [8 pools], R1_TICKET_STRIDE words apart
	const unsigned* gbase;   // [256] global start of every digit
	unsigned* status;        // workspace status word
	clo_u64* stamps;         // optional: 8 s_memtime stamps per tile (diagnostic builds of the probe tools)
	unsigned tiles, max_spins;
	unsigned pools;          // 1 or R1_POOLS ticket counters (see the head of this file)
};

// R1_EARLY: level-1 entries (the previous chunk's prefix + the nearest rows) requested
// before the second split, so that their round trip runs under it.
template <typename E, int LB, int HB, int R1_CHUNK_LOG, int R1_EARLY>
__global__ __launch_bounds__(sweep_shape<E>::THREADS, 6)
void clo_radix1_sweep_kernel(const E* __restrict__ in, E* __restrict__ out, size_t n,
	unsigned shift, unsigned mask_lo, unsigned mask_hi, r1_pass P, int aligned, clo_keyx kx_in, clo_keyx kx_out) {

	constexpr int THREADS = sweep_shape<E>::THREADS;
	constexpr int ITEMS = sweep_shape<E>::ITEMS;
	constexpr int TILE = THREADS * ITEMS;
	constexpr int WAVES = THREADS / 64;
	constexpr int R2 = 1 << (LB + HB);
	constexpr int HMAX = pc_words<(LB > HB ? LB : HB)>::H;
	constexpr unsigned R1_CHUNK = 1u << R1_CHUNK_LOG;    // tiles per chunk
	static_assert(R1_EARLY <= (int) R1_CHUNK, "at most the whole level 1");
	static_assert(R2 <= R1_ROW && R2 <= THREADS, "one thread per combined digit");
	static_assert(THREADS >= 2 * R1_ROW, "the digit threads are the second half of the work-group");

	__shared__ __attribute__((aligned(16))) E s_stage[TILE];
	__shared__ unsigned s_end[THREADS * PC_END_STRIDE];
	__shared__ unsigned s_wtot[WAVES][HMAX];
	__shared__ __attribute__((aligned(16))) unsigned s_wbase[WAVES][HMAX];
	__shared__ unsigned s_delta[R2];   // global index = tile-local position + delta[D]
	__shared__ unsigned s_w4[4];
	__shared__ unsigned s_tile;

	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	clo_u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
	if (P.stamps) t0 = __builtin_amdgcn_s_memtime();

	// The digits' bookkeeping (publishing, look-back) belongs to the SECOND half of
	// the work-group, one thread per combined digit: those waves have nothing to do
	// while wave 0 turns wave totals into wave bases inside a local split.
	const int dg = (int) tid - (THREADS - R1_ROW);
	const bool dthread = dg >= 0 && dg < R2;
	const bool dwaves = wave >= (unsigned) (WAVES - R1_ROW / 64);
	const unsigned dwave = wave - (unsigned) (WAVES - R1_ROW / 64);

	// per-wave counters of the tile histogram: they borrow the head of the stage
	// (read back before the first split writes the stage). 32 bank-private copies of
	// every bin (no two lanes of an LDS instruction on one bank) measured SLOWER
	// (13 470 vs 12 600 ticks for counting + first split): the LDS add unit itself,
	// ~2 lanes per clock, is the limit, not its bank conflicts.
	unsigned* s_hist = reinterpret_cast<unsigned*>(s_stage);
	static_assert(sizeof(E) * TILE >= WAVES * R1_ROW * sizeof(unsigned), "the histogram fits the stage");
	#pragma unroll
	for (int k = 0; k < WAVES * R1_ROW / THREADS; ++k) s_hist[k * THREADS + tid] = 0;

	// ---- ticket: the next tile (of this XCD's pool; another pool's once it is used up) ----
	if (tid == 0) {
		const unsigned nchunks = (P.tiles + R1_CHUNK - 1u) >> R1_CHUNK_LOG;
		const unsigned x = clo_xcc_id();
		unsigned tile = 0xffffffffu;
		for (unsigned t = 0; t < P.pools; ++t) {
			const unsigned pool = (x + t) & (P.pools - 1u);
			// (tiles of pool `pool`: chunks pool, pool + pools, ...; only the very last chunk may be partial)
			const unsigned k = atomicAdd(&P.ticket[pool * R1_TICKET_STRIDE], 1u);
			const unsigned c = (k >> R1_CHUNK_LOG) * P.pools + pool;
			const unsigned cand = (c << R1_CHUNK_LOG) + (k & (R1_CHUNK - 1u));
			if (c < nchunks && cand < P.tiles) { tile = cand; break; }
		}
		s_tile = tile;
	}
	clo_lds_barrier();
	const unsigned tile = s_tile;
	if (tile == 0xffffffffu) return;   // (cannot happen: as many work-groups as tiles, one valid ticket each)
	const unsigned c = tile >> R1_CHUNK_LOG, q = tile & (R1_CHUNK - 1u);
	const unsigned in_chunk = P.tiles - (c << R1_CHUNK_LOG) < R1_CHUNK ? P.tiles - (c << R1_CHUNK_LOG) : R1_CHUNK;
	const size_t base = (size_t) tile * TILE;
	const unsigned count = (n - base) < (size_t) TILE ? (unsigned) (n - base) : (unsigned) TILE;
	const bool full = count == (unsigned) TILE;
	const unsigned tbase = tid * ITEMS;
	const unsigned mask2 = (mask_hi << LB) | mask_lo;
	const unsigned n32 = (unsigned) n;   // n < 2^31 here

	// ---- load; count the combined digit (LDS adds, not waited for) ----
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
	if (P.stamps) t1 = __builtin_amdgcn_s_memtime();
	{
		// Keys that repeat (few distinct values, one hot bin) put many lanes of an LDS add
		// on ONE address, which the LDS serialises lane by lane: 2^28 equal keys sorted
		// in 6.5 ms instead of 3.0. A wave that sees a quarter of its lanes agree on the
		// first element's digit counts by groups instead: up to R1_PEEL times per element
		// the lanes that share the first remaining lane's digit are counted with one
		// ballot and added by that lane alone; whoever is left adds as usual.
		const unsigned d0 = (unsigned) (key[0] >> shift) & mask2;
		const bool skewed = __popcll(__ballot(d0 == (unsigned) __builtin_amdgcn_readfirstlane((int) d0))) >= 16;   // wave-uniform
		unsigned* const row = s_hist + wave * R1_ROW;
		if (!skewed) {
			#pragma unroll
			for (int i = 0; i < ITEMS; ++i)
				if (full || tbase + i < count) atomicAdd(&row[(unsigned) (key[i] >> shift) & mask2], 1u);
		} else {
			#pragma unroll
			for (int i = 0; i < ITEMS; ++i) {
				const unsigned d = (unsigned) (key[i] >> shift) & mask2;
				bool mine = full || tbase + i < count;   // still to be counted
				#pragma unroll 1
				for (int r = 0; r < R1_PEEL; ++r) {
					const unsigned long long rem = __ballot(mine);
					if (rem == 0) break;
					const int l = __ffsll((long long) rem) - 1;
					const unsigned dl = (unsigned) __builtin_amdgcn_readlane((int) d, l);
					const bool grp = mine && d == dl;
					const unsigned cnt = (unsigned) __popcll(__ballot(grp));
					if ((int) lane == l) atomicAdd(&row[dl], cnt);
					mine = mine && !grp;
				}
				if (mine) atomicAdd(&row[d], 1u);
			}
		}
	}

	// ---- first local split; between its first two barriers (all counts are in by
	// then) the digit threads publish the tile's row and its arrival at the chunk ----
	unsigned h2 = 0, incl2 = 0;
	clo_u64 old = 0;
	auto publish = [&]() {
		if (!dwaves) return;
		if (dthread) {
			#pragma unroll
			for (int w = 0; w < WAVES; ++w) h2 += s_hist[w * R1_ROW + dg];
			r1_st32(&P.agg[(size_t) tile * R1_ROW + dg], R1_VALID | h2);
			old = __hip_atomic_fetch_add(&P.cacc[(size_t) c * R1_ROW + dg], (1ull << R1_CNT_SHIFT) | (clo_u64) h2,
				__ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		// tile-local start of every combined digit: exclusive scan of the tile's histogram
		incl2 = clo_wave_scan_inclusive<unsigned>(h2, lane);
		if (lane == 63) s_w4[dwave] = incl2;
	};
	pc_local_split<E, LB, THREADS, ITEMS, HMAX>(key, shift, mask_lo, count, s_stage, s_end, s_wtot, s_wbase, publish);
	if (P.stamps) t2 = __builtin_amdgcn_s_memtime();

	unsigned dstart2 = 0;
	unsigned early[R1_EARLY > 0 ? R1_EARLY : 1];   // early[0]: prefix of the previous chunk; early[k]: row of tile - k (R1_VALID | count)
	// (not valid yet: a pass whose high digit is empty — a key width that is 4 mod 8 — has no second
	// split to request them from, and must fetch them in the loop below like the entries beyond
	// R1_EARLY. Round 2 initialised them as "valid, count 0": such a pass then ignored the counts of
	// up to 7 predecessors and the previous chunk — 28-, 20-, 12-bit keys on this path came out wrong.)
	#pragma unroll
	for (unsigned k = 0; k < (unsigned) R1_EARLY; ++k) early[k] = (k == 0 ? c > 0 : k <= q) ? 0u : R1_VALID;   // (entries this tile does not need count as zero)
	if (dthread) {
		dstart2 = incl2 - h2;
		#pragma unroll
		for (unsigned w = 0; w < 4; ++w) if (w < dwave) dstart2 += s_w4[w];
		// ---- level 2, by whichever tile completed (chunk, digit): look back over the
		// earlier chunks, a window at a time, and publish the chunk's inclusive prefix ----
		if ((unsigned) (old >> R1_CNT_SHIFT) == in_chunk - 1u) {
			const clo_u64 total = (old & R1_SUM_MASK) + h2;
			clo_u64 excl = 0;
			long j = (long) c - 1;
			unsigned spins = 0;
			while (j >= 0) {
				unsigned pv[R1_WINDOW];
				clo_u64 av[R1_WINDOW];
				#pragma unroll
				for (int w = 0; w < R1_WINDOW; ++w) {
					pv[w] = 0; av[w] = 0;
					if (j - w >= 0) {
						pv[w] = r1_ld32(&P.cprefix[(size_t) (j - w) * R1_ROW + dg]);
						av[w] = clo_ld_agent(&P.cacc[(size_t) (j - w) * R1_ROW + dg]);
					}
				}
				bool closed = false, stalled = false;
				int used = 0;
				#pragma unroll
				for (int w = 0; w < R1_WINDOW; ++w) {
					if (closed || stalled || j - w < 0) continue;
					if (pv[w] & R1_VALID) { excl += pv[w] & ~R1_VALID; closed = true; }
					else if ((unsigned) (av[w] >> R1_CNT_SHIFT) == R1_CHUNK) { excl += av[w] & R1_SUM_MASK; used = w + 1; }   // (every earlier chunk is full)
					else stalled = true;
				}
				if (closed) break;
				j -= used;
				if (stalled) {
					if (++spins > P.max_spins) { atomicExch(P.status, 1u); break; }
					__builtin_amdgcn_s_sleep(2);
				}
			}
			r1_st32(&P.cprefix[(size_t) c * R1_ROW + dg], R1_VALID | (unsigned) (excl + total));
		}
	}
	if (P.stamps) t3 = __builtin_amdgcn_s_memtime();

	if (mask_hi != 0) {
		if (full) {   // 16-byte LDS reads (scalar reads at this lane stride would conflict 8-way)
			constexpr int PER = ITEMS * (int) sizeof(E) >= 16 ? 16 / (int) sizeof(E) : ITEMS;
			typedef E vec16 __attribute__((ext_vector_type(PER)));
			#pragma unroll
			for (int k = 0; k < ITEMS / PER; ++k) {
				const vec16 t = *reinterpret_cast<const vec16*>(&s_stage[tbase + k * PER]);
				#pragma unroll
				for (int qq = 0; qq < PER; ++qq) key[k * PER + qq] = t[qq];
			}
		} else {
			#pragma unroll
			for (int i = 0; i < ITEMS; ++i) if (tbase + i < count) key[i] = s_stage[tbase + i];
		}
		// The nearest rows of level 1 and the previous chunk's prefix are requested from
		// INSIDE the second split (its idle slot for the digit waves): late enough for
		// rows published a moment ago to have become visible (a store takes 1-2 us to
		// show, and the tile before this one is only ~0.2 us ahead), early enough for
		// the round trip (~2.5 us) to run under the rest of the split.
		auto request = [&]() {
			if (!dthread) return;
			#pragma unroll
			for (unsigned k = 0; k < (unsigned) R1_EARLY; ++k) {
				const bool want = k == 0 ? c > 0 : k <= q;
				const unsigned* a = !want ? &P.agg[(size_t) tile * R1_ROW + dg]   // (own row: valid, ignored)
					: (k == 0 ? &P.cprefix[(size_t) (c - 1) * R1_ROW + dg] : &P.agg[(size_t) (tile - k) * R1_ROW + dg]);
				early[k] = r1_ld32(a);
			}
		};
		pc_local_split<E, HB, THREADS, ITEMS, HMAX>(key, shift + LB, mask_hi, count, s_stage, s_end, s_wtot, s_wbase, request);
	}
	if (P.stamps) t4 = __builtin_amdgcn_s_memtime();

	// ---- the tile's prefix: level 1 (earlier tiles of the chunk) + the prefix of the chunk before ----
	if (dthread) {
		// Entry k: 0 = the previous chunk's inclusive prefix, k >= 1 = the row of tile - k.
		// Entries this tile does not need read its OWN row instead (published long
		// ago, so valid) and count as zero: every load of a round is then unconditional
		// and the compiler issues the round as one batch — with a branch around each
		// load it waited for each before issuing the next, a round trip apiece.
		const unsigned* addr[R1_CHUNK];
		unsigned v[R1_CHUNK];
		#pragma unroll
		for (unsigned k = 0; k < R1_CHUNK; ++k) {
			const bool want = k == 0 ? c > 0 : k <= q;
			addr[k] = !want ? &P.agg[(size_t) tile * R1_ROW + dg]
				: (k == 0 ? &P.cprefix[(size_t) (c - 1) * R1_ROW + dg] : &P.agg[(size_t) (tile - k) * R1_ROW + dg]);
			v[k] = k < (unsigned) R1_EARLY ? early[k] : (want ? 0u : R1_VALID);
		}
		// Rounds of loads, each issued as batches of four entries; a batch is skipped
		// when the tile needs none of it (q is the same for the whole work-group: a
		// scalar branch, the loads inside stay unconditional).
		unsigned spins = 0;
		for (;;) {
			bool all = true;
			#pragma unroll
			for (unsigned k = 0; k < R1_CHUNK; ++k) all = all && (v[k] & R1_VALID);
			if (all) break;
			if (spins != 0) __builtin_amdgcn_s_sleep(1);
			if (++spins > P.max_spins) { atomicExch(P.status, 1u); break; }
			#pragma unroll
			for (unsigned g = 0; g < R1_CHUNK / 4; ++g) {
				if (g != 0 && q < 4 * g) continue;   // (entries 4g .. 4g+3 are rows of tile - 4g ...: not needed)
				if (spins == 1 && 4 * g + 3 < (unsigned) R1_EARLY) {   // requested already: reload only if something is missing
					const bool have = (v[4 * g] & v[4 * g + 1] & v[4 * g + 2] & v[4 * g + 3] & R1_VALID) != 0;
					if (__builtin_amdgcn_ballot_w64(!have) == 0) continue;
				}
				unsigned nv[4];
				#pragma unroll
				for (unsigned k = 0; k < 4; ++k) nv[k] = r1_ld32(addr[4 * g + k]);
				#pragma unroll
				for (unsigned k = 0; k < 4; ++k) if (!(v[4 * g + k] & R1_VALID)) v[4 * g + k] = nv[k];
			}
		}
		#pragma unroll
		for (unsigned k = 0; k < R1_CHUNK; ++k) {
			const bool want = k == 0 ? c > 0 : k <= q;
			if (!want) v[k] = R1_VALID;
		}
		unsigned excl = 0;
		#pragma unroll
		for (unsigned k = 0; k < R1_CHUNK; ++k) excl += v[k] & ~R1_VALID;
		s_delta[dg] = P.gbase[dg] + excl - dstart2;
	}
	clo_lds_barrier();
	if (P.stamps) t5 = __builtin_amdgcn_s_memtime();

	// ---- contiguous runs to HBM (as the pair kernel of clo_hip_radix4.hip) ----
	constexpr int VEC = sizeof(E) >= 8 ? 1 : 4;
	typedef E vecE __attribute__((ext_vector_type(VEC)));
	typedef E vecE_u __attribute__((ext_vector_type(VEC), aligned(sizeof(E))));
	#pragma unroll
	for (int j = 0; j < ITEMS / VEC; ++j) {
		const unsigned p = (j * THREADS + tid) * VEC;
		if (full) {
			const vecE vv = *reinterpret_cast<const vecE*>(&s_stage[p]);
			const unsigned d0 = (unsigned) (vv[0] >> shift) & mask2, dl = (unsigned) (vv[VEC - 1] >> shift) & mask2;
			const unsigned gi0 = p + s_delta[d0];
			if (d0 == dl && gi0 <= n32 - VEC) {
				vecE vo = vv;
				if (kx_out.kind) {
					#pragma unroll
					for (int k = 0; k < VEC; ++k) vo[k] = clo_keyx_inv<E>(vv[k], kx_out);
				}
				*reinterpret_cast<vecE_u*>(&out[gi0]) = vo;
			} else {
				#pragma unroll
				for (int k = 0; k < VEC; ++k) {
					const unsigned gi = p + k + s_delta[(unsigned) (vv[k] >> shift) & mask2];
					if (gi < n32) out[gi] = clo_keyx_inv<E>(vv[k], kx_out);
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
	if (P.stamps && tid == 0) {
		clo_u64* st = P.stamps + (size_t) tile * 8;
		st[0] = t0; st[1] = t1; st[2] = t2; st[3] = t3; st[4] = t4; st[5] = t5;
		st[6] = __builtin_amdgcn_s_memtime();
		st[7] = ((clo_u64) clo_xcc_id() << 32) | blockIdx.x;
	}
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct r1_layout { size_t ghist, gbase, tickets, done, pass0, per_pass, agg, cacc, cprefix, total, tiles, chunks; int passes; };

r1_layout r1_make_layout(size_t n, int elem_size, int key_bits) {
	r1_layout L;
	const size_t tile = (size_t) 512 * (elem_size == 8 ? 8 : 16);
	L.tiles = (n + tile - 1) / tile;
	if (L.tiles == 0) L.tiles = 1;
	L.chunks = (L.tiles + 7) / 8;   // (sized for the smaller chunk)
	L.passes = (key_bits + 7) / 8;
	size_t off = 0;
	L.ghist = off; off += (size_t) L.passes * R1_ROW * sizeof(unsigned);
	L.gbase = off; off += (size_t) L.passes * R1_ROW * sizeof(unsigned);
	L.tickets = off; off += (size_t) L.passes * R1_POOLS * R1_TICKET_STRIDE * sizeof(unsigned);
	L.done = off; off += 64;
	off = (off + 255) & ~(size_t) 255;
	L.pass0 = off;
	L.agg = 0;
	L.cacc = L.agg + L.tiles * R1_ROW * sizeof(unsigned);
	L.cprefix = L.cacc + L.chunks * R1_ROW * sizeof(clo_u64);
	L.per_pass = (L.cprefix + L.chunks * R1_ROW * sizeof(unsigned) + 255) & ~(size_t) 255;
	L.total = L.pass0 + L.per_pass * (size_t) L.passes;
	return L;
}

int r1_device_cus() {   // compute units of the current device (256 = all eight XCDs of an MI355X)
	int dev = 0, cus = 0;
	if (hipGetDevice(&dev) != hipSuccess) return 0;
	if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
	return cus;
}

clo_u64* g_r1_stamps = nullptr;   // diagnostic (clo_hip_radix_debug_stamps)
size_t g_r1_stamps_tiles = 0;

template <typename E, int NP>
void r1_launch_ghist(const E* src, size_t n, int key_shift, int key_bits, unsigned* ghist, unsigned* gbase, unsigned* done,
	clo_keyx kx, unsigned tiles, hipStream_t s) {
	unsigned per = tiles > 4096u ? (tiles + 2047u) / 2048u : (tiles > 256u ? 2u : 1u);
	if (per > 16u) per = 16u;   // (16-bit counters: at most 16 tiles between flushes)
	const unsigned groups = (tiles + per - 1u) / per;
	hipLaunchKernelGGL((clo_radix1_ghist_kernel<E, NP>), dim3(groups), dim3(R1_GH_THREADS), 0, s,
		src, n, (unsigned) key_shift, (unsigned) key_bits, ghist, gbase, done, (int) ((uintptr_t) src % 16 == 0), kx, per);
}

template <typename E>
int r1_sort_impl(const E* src, E* dst, E* tmp, size_t n, int key_shift, int key_bits, clo_keyx kx, void* ws, unsigned* status, hipStream_t s) {
	const r1_layout L = r1_make_layout(n, (int) sizeof(E), key_bits);
	char* w = (char*) ws;
	unsigned* ghist = (unsigned*) (w + L.ghist);
	unsigned* gbase = (unsigned*) (w + L.gbase);
	unsigned* tickets = (unsigned*) (w + L.tickets);
	const unsigned tiles = (unsigned) L.tiles;
	const int passes = L.passes;
	const clo_keyx kx_none = { 0, 0, 0 };
	const clo_hip_env_t* env = clo_hip_env();
	const unsigned max_spins = env->max_spins;

	// ticket pools: one counter unless the eight pools are safe (head of this file); CLO_R1_POOLS: tests of both
	unsigned pools = (tiles > 1024u && r1_device_cus() >= 256) ? (unsigned) R1_POOLS : 1u;
	if (env->r1_pools != 0) pools = env->r1_pools == R1_POOLS ? (unsigned) R1_POOLS : 1u;

	// everything the passes publish or count in starts from zero
	hipError_t e = hipMemsetAsync(ws, 0, L.total, s);
	if (e != hipSuccess) return (int) e;
	{
		clo_timing_scope timing("radix_ghist", s);
		switch (passes) {
			#define CLO_R1_GH(NP) case NP: r1_launch_ghist<E, NP>(src, n, key_shift, key_bits, ghist, gbase, (unsigned*) (w + L.done), kx, tiles, s); break
			CLO_R1_GH(1); CLO_R1_GH(2); CLO_R1_GH(3); CLO_R1_GH(4); CLO_R1_GH(5); CLO_R1_GH(6); CLO_R1_GH(7); CLO_R1_GH(8);
			#undef CLO_R1_GH
			default: return CLO_HIP_EUNSUPPORTED;
		}
	}
	const bool inplace_odd = (dst == src) && (passes % 2 == 1);
	const E* cur_in = src;
	for (int p = 0; p < passes; ++p) {
		E* cur_out;
		if (inplace_odd) cur_out = (p % 2 == 0) ? tmp : dst;
		else cur_out = ((passes - 1 - p) % 2 == 0) ? dst : tmp;
		const int rem = key_bits - p * 8;
		const int bits = rem < 8 ? rem : 8;
		const int lo_bits = bits < 4 ? bits : 4, hi_bits = bits - lo_bits;
		char* pw = w + L.pass0 + (size_t) p * L.per_pass;
		r1_pass P;
		P.agg = (unsigned*) (pw + L.agg);
		P.cacc = (clo_u64*) (pw + L.cacc);
		P.cprefix = (unsigned*) (pw + L.cprefix);
		P.ticket = tickets + (size_t) p * R1_POOLS * R1_TICKET_STRIDE;
		P.gbase = gbase + (size_t) p * R1_ROW;
		P.status = status;
		P.stamps = (g_r1_stamps && g_r1_stamps_tiles >= L.tiles && p == passes - 1) ? g_r1_stamps : nullptr;
		P.tiles = tiles;
		P.max_spins = max_spins;
		P.pools = pools;
		clo_timing_scope timing("radix_sweep", s);
		#define CLO_R1_SWEEP(CL, EARLY) hipLaunchKernelGGL((clo_radix1_sweep_kernel<E, 4, 4, CL, EARLY>), dim3(tiles), dim3(sweep_shape<E>::THREADS), 0, s, \
			cur_in, cur_out, n, (unsigned) (key_shift + p * 8), (1u << lo_bits) - 1u, (1u << hi_bits) - 1u, P, \
			(int) ((uintptr_t) cur_in % 16 == 0), p == 0 ? kx : kx_none, p + 1 == passes ? kx : kx_none)
		CLO_R1_SWEEP(4, 8);   // (16 tiles per chunk, 8 rows requested early: measured against 8 tiles / no early rows in round 2)
		#undef CLO_R1_SWEEP
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

}  // namespace

// 1: the single-sweep passes handle this sort; 0: the chain-free pair passes do
// (clo_hip_radix4.hip). Digits of 4 or 8 bits only (radix 16: two digits per
// sweep; radix 256: one), 4- and 8-byte elements, fewer than 2^31 of them.
int clo_radix1_applies(size_t n, int elem_size, int digit_bits) {
	// CLO_RADIX_SWEEP: 0 never, 1 whenever possible, unset: the library's choice
	const int mode = clo_hip_env()->radix_sweep;
	if (mode == 0) return 0;
	if ((digit_bits != 4 && digit_bits != 8) || (elem_size != 4 && elem_size != 8) || n >= 0x80000000ull) return 0;
	const size_t tiles = (n + (size_t) 512 * (elem_size == 8 ? 8 : 16) - 1) / ((size_t) 512 * (elem_size == 8 ? 8 : 16));
	if (mode == 1) return tiles > 1;
	// The library's choice (measured, docs/lab_notebook.md; profiles/r03_sweep_sizes*.txt, r03_sweep_edge*.txt):
	// 2 .. 256 tiles (up to 2^21 4-byte elements, 2^20 8-byte ones): the sort is launch-bound and
	// the sweeps need 6 launches instead of 12: 5-25 % less time (and half the events on a profiling
	// queue). Above that the chain-free passes are ahead, and no work-group ever waits for another's
	// result there. The switch moved down as their counter scan got cheaper: round 2 switched at 1024 tiles,
	// round 3 at 512 when the scan's three kernels lost 10 us per pass, then at 256 when they became
	// one launch (2^22 uint32: 0.104 vs 0.119 ms, 2^23: 0.144 vs 0.190; 2^21 uint64: 0.187 vs 0.199;
	// 2^21 pairs: 0.093 vs 0.107 — at 2^21 uint32 the sweeps still win, 0.090 vs 0.095).
	return tiles >= 2 && tiles <= 256;   // (round 4: from 2 tiles, not 4 — 8 193 uint64 keys 0.109 -> 0.083 ms, 16 385 uint32 keys 0.065 -> 0.055)
}

size_t clo_radix1_workspace_bytes(size_t n, int elem_size, int key_bits) {
	return r1_make_layout(n, elem_size, key_bits).total;
}

int clo_radix1_sort(const void* src, void* dst, void* tmp, size_t n, int elem_size, int key_shift, int key_bits,
	clo_keyx kx, void* ws, unsigned* status, hipStream_t s) {
	if (elem_size == 4) return r1_sort_impl<uint32_t>((const uint32_t*) src, (uint32_t*) dst, (uint32_t*) tmp, n, key_shift, key_bits, kx, ws, status, s);
	if (elem_size == 8) return r1_sort_impl<uint64_t>((const uint64_t*) src, (uint64_t*) dst, (uint64_t*) tmp, n, key_shift, key_bits, kx, ws, status, s);
	return CLO_HIP_EUNSUPPORTED;
}

extern "C" int clo_hip_radix_debug_stamps(void* buffer, size_t tiles) {
	g_r1_stamps = (clo_u64*) buffer;
	g_r1_stamps_tiles = buffer ? tiles : 0;
	return 0;
}
