// clo_hip_scan.hip — exclusive prefix sum for gfx950.
//
// Replaces the three launches of the reference Blelloch scan
// (scan/clo_scan_blelloch.c:146-195 -> workgroupScan / workgroupSumsScan /
// addWorkgroupSums, scan/clo_scan_blelloch.cl:49-211), which move 16 B per
// 4-byte element through HBM and run on at most lws work-groups. Here: ONE
// launch, each element read once and written once (8 B per u32->u32 element):
//
//   * a tile = 256 threads x 4 consecutive elements x ROWS rows, loaded with
//     one 16-byte (u32) coalesced vector load per lane and row, kept in VGPRs;
//   * in-lane prefix of the 4 elements, wave64 shuffle scan of the lane sums,
//     cross-wave/row offsets through 1 KiB of LDS;
//   * as many work-groups as the chip holds at once; each draws tiles by ticket
//     (atomicAdd) until none is left, so tile ids follow the order in which the
//     tiles are started; the running prefix travels tile to tile by decoupled look-back on
//     8-byte {tag,value} granules with agent-scope relaxed loads/stores
//     (per-XCD L2s are not coherent: clo_hip_internal.h).
//
// Arithmetic is modular in the sum type, exactly as the reference's
// CLO_SCAN_SUM_TYPE additions (blelloch.cl:79-124).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

#include "clo_hip.h"
#include "clo_hip_internal.h"

namespace {

constexpr int SCAN_VEC = 4;

// 4 consecutive elements as one aligned vector access where possible.
template <typename T>
__device__ __forceinline__ void load4(const T* p, T (&v)[4]) {
	typedef T vec4 __attribute__((ext_vector_type(4)));
	vec4 x = *reinterpret_cast<const vec4*>(p);
	v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
}
// (experiments: non-temporal variants, CLO_SCAN_XFLAGS bits 2 / 4)
template <typename T>
__device__ __forceinline__ void load4_nt(const T* p, T (&v)[4]) {
	typedef T vec4 __attribute__((ext_vector_type(4)));
	vec4 x = __builtin_nontemporal_load(reinterpret_cast<const vec4*>(p));
	v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
}
template <typename T>
__device__ __forceinline__ void store4_nt(T* p, const T (&v)[4]) {
	typedef T vec4 __attribute__((ext_vector_type(4)));
	vec4 x; x.x = v[0]; x.y = v[1]; x.z = v[2]; x.w = v[3];
	__builtin_nontemporal_store(x, reinterpret_cast<vec4*>(p));
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const T (&v)[4]) {
	typedef T vec4 __attribute__((ext_vector_type(4)));
	vec4 x; x.x = v[0]; x.y = v[1]; x.z = v[2]; x.w = v[3];
	*reinterpret_cast<vec4*>(p) = x;
}

// What the look-back needs to know about this call: granules carry the call's
// epoch in their tag (clo_hip_internal.h), so nothing written by an earlier call
// on the same workspace can be mistaken for this call's — the workspace is
// never cleared between calls.
struct scan_ctl {
	unsigned epoch;       // 1 .. CLO_LB_EPOCH_MAX
	unsigned max_spins;   // bound of every poll loop
	unsigned* status;     // workspace status word
};

// One look-back window, run by wave 0 (all 64 lanes): lane l inspects entry
// (idx - l) of `state` as long as it is >= lo. Waits until every inspected
// entry nearer than the nearest inclusive prefix is published, then returns
// their sum; *closed tells whether a prefix ended the window. Entries below
// `lo` end the window: as a prefix of 0 if `floor_is_prefix`, else silently.
// NG granules per entry carry the sum in 32-bit pieces.
template <typename TSum, int NG>
__device__ __forceinline__ void scan_read_entry(const clo_u64* state, long j, unsigned& tag, TSum& val) {
	const clo_u64 g0 = clo_ld_agent(&state[(size_t) j * 2]);
	tag = clo_lb_tag(g0);
	val = (TSum) clo_lb_val(g0);
	if (NG == 2) {
		const clo_u64 g1 = clo_ld_agent(&state[(size_t) j * 2 + 1]);
		if (clo_lb_tag(g1) != tag) tag = 0;  // mid-update: poll again
		val = (TSum) (((clo_u64) clo_lb_val(g1) << 32) | clo_lb_val(g0));
	}
}

// `alt` (may be NULL): a second array consulted for entries whose `state`
// granule is not published yet (super-tiles keep aggregates and prefixes in
// two single-writer arrays).
template <typename TSum, int NG>
__device__ TSum scan_window(const clo_u64* state, const clo_u64* alt, long idx, long lo, bool floor_is_prefix,
	unsigned lane, bool* closed, const scan_ctl& ctl) {
	unsigned spins = 0;
	while (true) {
		const long j = idx - (long) lane;
		unsigned tag = (ctl.epoch << 2) | (floor_is_prefix ? CLO_LB_PREFIX : CLO_LB_AGG);  // below lo: value 0
		TSum val = 0;
		if (j >= lo) {
			scan_read_entry<TSum, NG>(state, j, tag, val);
			if (alt != nullptr && !((tag >> 2) == ctl.epoch && (tag & 3u) != 0u)) scan_read_entry<TSum, NG>(alt, j, tag, val);
		}
		const unsigned st = tag & 3u;
		const bool valid = (tag >> 2) == ctl.epoch && st != 0u;
		const clo_u64 pmask = __ballot(valid && st == CLO_LB_PREFIX);
		const clo_u64 imask = __ballot(!valid);
		if (pmask) {
			const unsigned first = (unsigned) __ffsll((long long) pmask) - 1u;
			const clo_u64 below = (first == 0) ? 0ull : ((~0ull) >> (64 - first));
			if ((imask & below) == 0) {
				*closed = true;
				return clo_wave_reduce_sum<TSum>(lane <= first ? val : (TSum) 0);
			}
		} else if (imask == 0) {
			*closed = false;
			return clo_wave_reduce_sum<TSum>(val);
		}
		if (++spins > ctl.max_spins) {
			if (lane == 0) atomicExch(ctl.status, 1u);
			*closed = true;
			return 0;
		}
		__builtin_amdgcn_s_sleep(4);
	}
}

// Two-level look-back. A running prefix handed tile to tile advances at most
// one window (64 tiles) per poll round trip, and a round trip to another
// XCD's granule costs microseconds under streaming load: measured, that chain
// alone capped the scan at 0.14 ms for 4096 tiles (0.11 ms with the look-back
// stubbed out). So tiles are grouped in super-tiles of 64: a tile sums the
// AGGREGATES of the earlier tiles of its own super-tile (published as soon as
// those tiles have loaded, no chain) and takes the prefix of its super-tile
// from a second granule array that carries super-tile aggregates/prefixes
// (64 super-tiles = 4096 tiles per window).
constexpr int SCAN_SUPER_LOG = 6;

template <typename TSum, int NG>
__device__ __forceinline__ void scan_publish(clo_u64* state, unsigned tile, unsigned epoch, unsigned st, TSum v);

// Super-tile state lives in two arrays with one writer each (two writers on a
// two-granule entry could leave a mixed pair): `sagg` gets the super-tile's
// aggregate from whichever tile arrives 64th at the super-tile's accumulator,
// `sprefix` gets its inclusive prefix from the super-tile's last tile.
template <typename TSum, int NG>
__device__ TSum scan_lookback2(clo_u64* tstate, clo_u64* sprefix, const clo_u64* sagg, unsigned tile, TSum aggregate,
	unsigned lane, const scan_ctl& ctl) {
	const unsigned q = tile & ((1u << SCAN_SUPER_LOG) - 1u);
	const long sup = (long) (tile >> SCAN_SUPER_LOG);
	bool closed = false;
	TSum excl = 0;
	if (q != 0)
		excl = scan_window<TSum, NG>(tstate, nullptr, (long) tile - 1, sup << SCAN_SUPER_LOG, false, lane, &closed, ctl);
	long idx = sup - 1;
	while (!closed) {
		excl += scan_window<TSum, NG>(sprefix, sagg, idx, 0, true, lane, &closed, ctl);
		idx -= 64;
	}
	if (q == (1u << SCAN_SUPER_LOG) - 1u && lane == 0)
		scan_publish<TSum, NG>(sprefix, (unsigned) sup, ctl.epoch, CLO_LB_PREFIX, (TSum) (excl + aggregate));
	return excl;
}

template <typename TSum, int NG>
__device__ __forceinline__ void scan_publish(clo_u64* state, unsigned tile, unsigned epoch, unsigned st, TSum v) {
	const clo_u64 x = (clo_u64) v;
	clo_st_agent(&state[(size_t) tile * 2], clo_lb_pack(epoch, st, (unsigned) x));
	if (NG == 2) clo_st_agent(&state[(size_t) tile * 2 + 1], clo_lb_pack(epoch, st, (unsigned) (x >> 32)));
}

// TIn/TOut: memory types; TSum: 32- or 64-bit accumulator (sums narrower than
// 32 bits are computed mod 2^32 and truncated on store, which is the same
// residue).
#ifndef CLO_SCAN_BIG_WAVES_PER_EU
#define CLO_SCAN_BIG_WAVES_PER_EU 4
#endif
template <typename TIn, typename TOut, typename TSum, int ROWS, int SCAN_THREADS>
__global__ __launch_bounds__(SCAN_THREADS, (SCAN_THREADS == 1024 ? CLO_SCAN_BIG_WAVES_PER_EU : 1))
void clo_scan_kernel(const TIn* __restrict__ in, TOut* __restrict__ out, size_t n,
	unsigned* hdr, clo_u64* state, clo_u64* sstate, clo_u64* sagg, clo_u64* sacc, int aligned, unsigned xflags,
	const clo_u64* __restrict__ carry_in, clo_u64* __restrict__ carry_out, unsigned last_tile,
	unsigned max_spins, size_t ws_granules) {

	constexpr int SCAN_WAVES = SCAN_THREADS / 64;
	constexpr int ROW_ELEMS = SCAN_THREADS * SCAN_VEC;
	constexpr int TILE = ROW_ELEMS * ROWS;
	constexpr int NG = sizeof(TSum) > 4 ? 2 : 1;

	__shared__ unsigned s_tile;
	__shared__ TSum s_part[ROWS][SCAN_WAVES];
	__shared__ TSum s_off[ROWS][SCAN_WAVES];
	__shared__ TSum s_excl;

	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

	// This call's epoch: one more than the value the previous call on this
	// workspace left in the header (its last work-group to leave wrote it, below;
	// clo_hip_scan_workspace_init wrote 0). Every work-group of the launch reads it
	// before that happens again: the header only changes when all have left.
	// (requested here, waited for only where the first tile's loads have been issued: the ticket below and this load are two
	// round trips to memory that would otherwise stand one behind the other in front of every work-group's first load)
	const unsigned epoch_raw = __hip_atomic_load(&hdr[CLO_WS_EPOCH_WORD], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	scan_ctl ctl;
	ctl.epoch = 0;
	ctl.max_spins = max_spins;
	ctl.status = &hdr[0];
	const unsigned tiles = last_tile + 1u;

	// value carried into this call (a chunk of a longer array): added to every output
	const TSum carry = carry_in ? (TSum) *carry_in : (TSum) 0;
	// A work-group draws tiles until none is left (it holds one ticket at a time). Measured and NOT done (round 5,
	// profiles/r05_ab_scan_barriers.txt): drawing the next tile's ticket early — before the look-back: tickets run ahead of the
	// tiles' starts and the groups behind wait for tiles nobody has begun, 2^28 0.36 -> 0.54 ms; right behind the look-back,
	// with barriers that order LDS only so that no wave waits for its stores' acknowledgements: still 0.366 -> 0.381 (2^26
	// 0.1027 -> 0.1056). The full barriers stay.
	for (;;) {
	if (tid == 0) s_tile = atomicAdd(&hdr[CLO_WS_TICKET_WORD], 1u);
	__syncthreads();
	// (read back from LDS the compiler takes the tile for a per-lane value: every row's address then becomes 64-bit vector
	// arithmetic in registers of its own — 16 VGPRs for the loads, 16 for the stores; as a scalar it is a base in SGPRs)
	const unsigned tile = (unsigned) __builtin_amdgcn_readfirstlane((int) s_tile);
	const size_t base = (size_t) tile * TILE;
	if (base >= n) {
		// Leaving. The LAST work-group to leave (every group draws exactly one
		// ticket past the end) hands the workspace to the next call: ticket and
		// leave counters back to 0, the epoch advanced — so no call ever clears
		// the workspace. Once in CLO_LB_EPOCH_MAX calls the epoch wraps: then this
		// group zeroes every granule first (nobody else is left to read them).
		__syncthreads();   // (s_tile is rewritten)
		if (tid == 0) s_tile = atomicAdd(&hdr[CLO_WS_DONE_WORD], 1u);
		__syncthreads();
		if (s_tile != gridDim.x - 1u) return;
		ctl.epoch = (unsigned) __builtin_amdgcn_readfirstlane((int) epoch_raw) + 1u;
		const bool wrap = ctl.epoch >= CLO_LB_EPOCH_MAX;
		if (wrap) {
			clo_u64* g = reinterpret_cast<clo_u64*>(reinterpret_cast<char*>(hdr) + CLO_WS_HEADER_BYTES);
			for (size_t i = tid; i < ws_granules; i += SCAN_THREADS) clo_st_agent(&g[i], 0ull);
			__syncthreads();
		}
		if (tid == 0) {
			__hip_atomic_store(&hdr[CLO_WS_TICKET_WORD], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(&hdr[CLO_WS_DONE_WORD], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(&hdr[CLO_WS_EPOCH_WORD], wrap ? 0u : ctl.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		return;
	}
	const bool full = (base + TILE <= n) && aligned;

	// ---- load (elem -> sum type conversion on load, blelloch.cl:79-80) ----
	TSum v[ROWS][SCAN_VEC];
	if (full) {
		#pragma unroll
		for (int r = 0; r < ROWS; ++r) {
			TIn t[4];
			const TIn* const rowp = in + base + (size_t) r * ROW_ELEMS;   // (wave-uniform: a scalar base + a 32-bit lane offset)
			if (xflags & 2u) load4_nt<TIn>(rowp + tid * SCAN_VEC, t);
			else load4<TIn>(rowp + tid * SCAN_VEC, t);
			#pragma unroll
			for (int c = 0; c < 4; ++c) v[r][c] = (TSum) t[c];
		}
	} else {
		#pragma unroll
		for (int r = 0; r < ROWS; ++r) {
			#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const size_t i = base + (size_t) r * ROW_ELEMS + tid * SCAN_VEC + c;
				v[r][c] = i < n ? (TSum) in[i] : (TSum) 0;
			}
		}
	}

	ctl.epoch = (unsigned) __builtin_amdgcn_readfirstlane((int) epoch_raw) + 1u;   // (wave-uniform: scalar from here on)

	// ---- in-lane inclusive prefix, wave scan of lane totals per row ----
	TSum lane_excl[ROWS];
	#pragma unroll
	for (int r = 0; r < ROWS; ++r) {
		v[r][1] += v[r][0]; v[r][2] += v[r][1]; v[r][3] += v[r][2];
		const TSum incl = clo_wave_scan_inclusive<TSum>(v[r][3], lane);
		lane_excl[r] = incl - v[r][3];
		if (lane == 63) s_part[r][wave] = incl;
	}
	__syncthreads();

	// ---- offsets of the (row, wave) pieces inside the tile; tile aggregate:
	// one wave-wide scan of the ROWS * SCAN_WAVES piece totals (row-major =
	// element order) by wave 0, which needs the aggregate for the look-back
	// anyway; the other waves pick their offsets up after the look-back barrier.
	// (Every thread summing the pieces itself kept 64 more values live: 171
	// VGPRs, two work-groups per CU; now 108, four.) ----
	constexpr int PIECES = ROWS * SCAN_WAVES;
	constexpr int PPL = (PIECES + 63) / 64;   // pieces per lane of wave 0
	static_assert(PIECES % PPL == 0, "whole pieces per lane");
	TSum aggregate = 0;
	if (wave == 0) {
		TSum piece[PPL], mine = 0;
		#pragma unroll
		for (int k = 0; k < PPL; ++k) {
			piece[k] = lane * PPL + k < (unsigned) PIECES ? (&s_part[0][0])[lane * PPL + k] : (TSum) 0;
			mine += piece[k];
		}
		const TSum incl = clo_wave_scan_inclusive<TSum>(mine, lane);
		TSum run = incl - mine;
		#pragma unroll
		for (int k = 0; k < PPL; ++k) {
			if (lane * PPL + k < (unsigned) PIECES) (&s_off[0][0])[lane * PPL + k] = run;
			run += piece[k];
		}
		aggregate = clo_wave_reduce_sum<TSum>(lane == 63 ? incl : (TSum) 0);   // (lane 63's value, wave-uniform)
	}

	// ---- prefix of the tile (wave 0): two-level decoupled look-back ----
	if (wave == 0) {
		TSum excl = 0;
		const bool first = tile == 0 || (xflags & 1u);
		const unsigned sup = tile >> SCAN_SUPER_LOG;
		if (lane == 0) {
			scan_publish<TSum, NG>(state, tile, ctl.epoch, first ? CLO_LB_PREFIX : CLO_LB_AGG, aggregate);
			// Add the aggregate to the super-tile's accumulator; the arrival that
			// completes the super-tile publishes the total and puts the accumulator
			// back to 0 for the next call (it is the last one to touch it). The final
			// super-tile of the array may hold fewer than 64 tiles.
			const unsigned in_super = tiles - (sup << SCAN_SUPER_LOG) < (1u << SCAN_SUPER_LOG)
				? tiles - (sup << SCAN_SUPER_LOG) : (1u << SCAN_SUPER_LOG);
			bool last;
			TSum total;
			if (NG == 1) {
				const clo_u64 old = __hip_atomic_fetch_add(&sacc[(size_t) sup * 2],
					((clo_u64) aggregate << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				last = (unsigned) old == in_super - 1u;
				total = (TSum) ((unsigned) (old >> 32)) + aggregate;
			} else {
				// RETURNING add: its result coming back means the add has been performed at
				// the memory side, so the sum is in before the arrival is counted
				const clo_u64 before = __hip_atomic_fetch_add(&sacc[(size_t) sup * 2], (clo_u64) aggregate,
					__ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				asm volatile("s_waitcnt vmcnt(0)" :: "v"((unsigned) before) : "memory");
				const clo_u64 cnt = __hip_atomic_fetch_add(&sacc[(size_t) sup * 2 + 1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				last = cnt == (clo_u64) in_super - 1ull;
				total = last ? (TSum) clo_ld_agent(&sacc[(size_t) sup * 2]) : (TSum) 0;
			}
			if (last) {
				scan_publish<TSum, NG>(sagg, sup, ctl.epoch, CLO_LB_AGG, total);
				clo_st_agent(&sacc[(size_t) sup * 2], 0ull);
				if (NG == 2) clo_st_agent(&sacc[(size_t) sup * 2 + 1], 0ull);
			}
		}
		if (!first) {
			excl = scan_lookback2<TSum, NG>(state, sstate, sagg, tile, aggregate, lane, ctl);
			if (lane == 0) scan_publish<TSum, NG>(state, tile, ctl.epoch, CLO_LB_PREFIX, (TSum) (excl + aggregate));
		} else if (tile == 0 && lane == 0 && (1u << SCAN_SUPER_LOG) == 1u) {
			scan_publish<TSum, NG>(sstate, 0, ctl.epoch, CLO_LB_PREFIX, aggregate);
		}
		if (lane == 0) {
			s_excl = excl;
			if (carry_out && tile == last_tile) *carry_out = (clo_u64) (TSum) (carry + excl + aggregate);
		}
	}
	__syncthreads();
	const TSum tile_excl = s_excl + carry;
	#pragma unroll
	for (int r = 0; r < ROWS; ++r) lane_excl[r] += s_off[r][wave];

	// ---- store: exclusive value of element c = offset + inclusive(c-1) ----
	if (full) {
		#pragma unroll
		for (int r = 0; r < ROWS; ++r) {
			const TSum o = tile_excl + lane_excl[r];
			TOut t[4] = { (TOut) o, (TOut) (o + v[r][0]), (TOut) (o + v[r][1]), (TOut) (o + v[r][2]) };
			TOut* const rowp = out + base + (size_t) r * ROW_ELEMS;
			if (xflags & 4u) store4_nt<TOut>(rowp + tid * SCAN_VEC, t);
			else store4<TOut>(rowp + tid * SCAN_VEC, t);
		}
	} else {
		#pragma unroll
		for (int r = 0; r < ROWS; ++r) {
			const TSum o = tile_excl + lane_excl[r];
			#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const size_t i = base + (size_t) r * ROW_ELEMS + tid * SCAN_VEC + c;
				if (i < n) out[i] = (TOut) (c == 0 ? o : (TSum) (o + v[r][c - 1]));
			}
		}
	}
	__syncthreads();   // s_tile, s_part, s_off, s_excl are reused
	}
}

constexpr unsigned g_scan_xflags = 0;  // (the kernel's experiment switches — non-temporal loads / stores: no difference, docs/lab_notebook.md — stay off)
unsigned g_scan_max_spins = CLO_MAX_SPINS;   // CLO_MAX_SPINS in the environment overrides (tests force a give-up with it)

// Work-group shape by array size. Every work-group draws a ticket from one
// counter (HIP promises no dispatch order, so tile ids cannot come from
// blockIdx: a tile may only wait for tiles already handed out), and same-address
// atomics complete one per ~12 ns: with 256-thread groups a 2^26 scan spends 49
// us of its 120 in that queue. Large arrays therefore use 1024-thread groups —
// four times fewer tickets and look-back entries (2^26 uint -> ulong: 0.207 ->
// 0.182 ms) — and small ones 256-thread groups, which spread over more CUs.
// The large shape keeps 8 rows per thread whatever the sum type: 2^26 uint
// measured 0.121 ms with 16 rows and 0.1135 with 8 (512 x 16 and 512 x 8 within
// 2 % of that). Either way one 16-wave group is resident per CU (96 VGPRs with
// 8 rows, 108 with 16; capping at 64 for two groups spills 82 registers): the
// shorter tile is what helps — its load, look-back and store phases alternate
// twice as often.
#ifndef CLO_SCAN_BIG_NUMEL32
#define CLO_SCAN_BIG_NUMEL32 ((size_t) 1 << 21)
#endif
constexpr size_t SCAN_BIG_NUMEL = CLO_SCAN_BIG_NUMEL32;
#ifndef CLO_SCAN_BIG_ROWS
#define CLO_SCAN_BIG_ROWS 8
#endif
#ifndef CLO_SCAN_BIG_GROUPS
#define CLO_SCAN_BIG_GROUPS 256
#endif
constexpr int SCAN_BIG_ROWS = CLO_SCAN_BIG_ROWS;
// Where the large shape starts (round 4, tools/scan_sizes_probe.py; it was 2^24 for every sum type): the small shape
// has every tile in flight at once and hands the super-tiles' prefixes on one after the other, 3.5 us a hop — 1024 tiles
// of 8 192 elements (64-bit sums) are sixteen hops: 2^23 uint -> ulong took 0.083 ms BETWEEN 0.030 at 2^22 and 0.055 at
// 2^24, found as a dip in the harness's sweep. From 2^21 elements on the large shape (64 and more tiles of 32 768) is
// the faster one for both sum widths: uint -> ulong 2^23 0.083 -> 0.033 ms, 2^22 0.030 -> 0.025, 2^21 0.023 -> 0.021;
// uint -> uint 2^23 0.026 -> 0.021, 2^22 0.020 -> 0.016, 2^21 0.0153 -> 0.0136 (2^20: slower, stays small).
#ifndef CLO_SCAN_BIG_NUMEL64
#define CLO_SCAN_BIG_NUMEL64 ((size_t) 1 << 21)
#endif
constexpr bool scan_big(size_t numel, int sum_size) { return numel >= (sum_size > 4 ? CLO_SCAN_BIG_NUMEL64 : SCAN_BIG_NUMEL); }
constexpr int scan_threads(size_t numel, int sum_size) { return scan_big(numel, sum_size) ? 1024 : 256; }
constexpr int scan_small_rows(int sum_size) { return sum_size > 4 ? 8 : 16; }
constexpr size_t scan_tile_elems(size_t numel, int sum_size) {
	return (size_t) scan_threads(numel, sum_size) * SCAN_VEC * (scan_big(numel, sum_size) ? SCAN_BIG_ROWS : scan_small_rows(sum_size));
}

// Granules a workspace for `tiles` tiles holds, and how many of them (at its
// start) are the super-tile accumulators: two per super-tile the workspace could
// ever serve (a workspace of G granules serves at most G / 2 tiles).
constexpr size_t scan_ws_granules(size_t tiles) { return tiles * 2 + 8 * ((tiles >> SCAN_SUPER_LOG) + 2); }
constexpr size_t scan_sacc_granules(size_t ws_granules) { return 2 * (ws_granules / 128 + 2); }

template <typename TIn, typename TOut>
int launch_scan(const void* in, void* out, size_t n, const clo_u64* carry_in, clo_u64* carry_out, void* ws, size_t ws_bytes, hipStream_t s) {
	if constexpr (sizeof(TIn) > sizeof(TOut)) {
		return CLO_HIP_EUNSUPPORTED;
	} else {
		typedef typename std::conditional<(sizeof(TOut) > 4), uint64_t, uint32_t>::type TSum;
		constexpr int ROWS = scan_small_rows((int) sizeof(TOut));
		const size_t tile = scan_tile_elems(n, (int) sizeof(TOut));
		const size_t tiles = (n + tile - 1) / tile;
		// Workspace: [header][accumulators][look-back entries of THIS call's tiles].
		// The accumulators are plain counters that must start from zero — the
		// arrival that completes one zeroes it again — so they live in a region
		// whose place depends on the workspace's size only: the epoch-tagged entries
		// of another call (another tile count, another layout) never land on them.
		unsigned* hdr = (unsigned*) ws;
		const size_t ws_granules = (ws_bytes - CLO_WS_HEADER_BYTES) / sizeof(clo_u64);
		clo_u64* sacc = (clo_u64*) ((char*) ws + CLO_WS_HEADER_BYTES);
		clo_u64* state = sacc + scan_sacc_granules(ws_granules);
		const size_t supers = (tiles >> SCAN_SUPER_LOG) + 1;
		clo_u64* sstate = state + tiles * 2;
		clo_u64* sagg = sstate + supers * 2;
		const int aligned = ((uintptr_t) in % (4 * sizeof(TIn)) == 0) && ((uintptr_t) out % (4 * sizeof(TOut)) == 0);
		// No clearing of the workspace: granules carry the call's epoch, accumulators
		// and counters are put back by the kernel itself (clo_hip_scan_workspace_init
		// zeroed everything once).
		const unsigned max_spins = g_scan_max_spins;
		clo_timing_scope timing("scan", s);
		// as many work-groups as fit the chip at once; each draws tiles until none is left
		const unsigned groups_big = (unsigned) (tiles < CLO_SCAN_BIG_GROUPS ? tiles : CLO_SCAN_BIG_GROUPS), groups_small = (unsigned) (tiles < 2048 ? tiles : 2048);
		if (scan_threads(n, (int) sizeof(TOut)) == 1024)
			hipLaunchKernelGGL((clo_scan_kernel<TIn, TOut, TSum, SCAN_BIG_ROWS, 1024>), dim3(groups_big), dim3(1024), 0, s,
				(const TIn*) in, (TOut*) out, n, hdr, state, sstate, sagg, sacc, aligned, g_scan_xflags,
				carry_in, carry_out, (unsigned) (tiles - 1), max_spins, ws_granules);
		else
			hipLaunchKernelGGL((clo_scan_kernel<TIn, TOut, TSum, ROWS, 256>), dim3(groups_small), dim3(256), 0, s,
				(const TIn*) in, (TOut*) out, n, hdr, state, sstate, sagg, sacc, aligned, g_scan_xflags,
				carry_in, carry_out, (unsigned) (tiles - 1), max_spins, ws_granules);
		return (int) hipGetLastError();
	}
}

template <typename TOut>
int dispatch_in(const void* in, void* out, size_t n, int es, int sgn, const clo_u64* ci, clo_u64* co, void* ws, size_t wsb, hipStream_t s) {
	switch (es) {
		case 1: return sgn ? launch_scan<int8_t, TOut>(in, out, n, ci, co, ws, wsb, s) : launch_scan<uint8_t, TOut>(in, out, n, ci, co, ws, wsb, s);
		case 2: return sgn ? launch_scan<int16_t, TOut>(in, out, n, ci, co, ws, wsb, s) : launch_scan<uint16_t, TOut>(in, out, n, ci, co, ws, wsb, s);
		case 4: return sgn ? launch_scan<int32_t, TOut>(in, out, n, ci, co, ws, wsb, s) : launch_scan<uint32_t, TOut>(in, out, n, ci, co, ws, wsb, s);
		case 8: return launch_scan<uint64_t, TOut>(in, out, n, ci, co, ws, wsb, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

// Sum of all elements mod 2^64 (sign-extended when TIn is signed): the total a
// sharded scan hands to the later shards.
template <typename TIn>
__global__ __launch_bounds__(256)
void clo_reduce_kernel(const TIn* __restrict__ in, size_t n, unsigned long long* __restrict__ total, int aligned) {
	__shared__ unsigned long long s_w[4];
	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	unsigned long long acc = 0;
	// A work-group sums one CONTIGUOUS segment, four 16-byte loads per thread in
	// flight. (Grid-stride chunks made 4096 groups walk 4096 streams 16 MiB apart:
	// every iteration touched every page of the array, 2.2 TB/s at 2^28 uint.)
	constexpr int UNROLL = 4;
	const size_t chunk = 256u * 4u;
	const size_t chunks = (n + chunk - 1) / chunk;
	const size_t per = (chunks + gridDim.x - 1) / gridDim.x;
	const size_t first = (size_t) blockIdx.x * per * chunk;
	const size_t last = first + per * chunk < n ? first + per * chunk : n;
	for (size_t base = first; base < last; base += chunk * UNROLL) {
		TIn t[UNROLL][4];
		#pragma unroll
		for (int u = 0; u < UNROLL; ++u) {
			const size_t i = base + (size_t) u * chunk + (size_t) tid * 4;
			if (aligned && i + 4 <= last) {
				load4<TIn>(in + i, t[u]);
			} else {
				#pragma unroll
				for (int c = 0; c < 4; ++c) t[u][c] = i + c < last ? in[i + c] : (TIn) 0;
			}
		}
		#pragma unroll
		for (int u = 0; u < UNROLL; ++u) {
			#pragma unroll
			for (int c = 0; c < 4; ++c) acc += (unsigned long long) (long long) t[u][c];
		}
	}
	const unsigned long long w = clo_wave_reduce_sum<unsigned long long>(acc);
	if (lane == 0) s_w[wave] = w;
	__syncthreads();
	if (tid == 0) atomicAdd(total, s_w[0] + s_w[1] + s_w[2] + s_w[3]);
}

template <typename TIn>
int launch_reduce(const void* in, size_t n, uint64_t* total, hipStream_t s) {
	size_t blocks = (n + 1023) / 1024;
	if (blocks > 4096) blocks = 4096;
	clo_timing_scope timing("reduce", s);
	hipLaunchKernelGGL((clo_reduce_kernel<TIn>), dim3((unsigned) blocks), dim3(256), 0, s,
		(const TIn*) in, n, (unsigned long long*) total, (int) ((uintptr_t) in % (4 * sizeof(TIn)) == 0));
	return (int) hipGetLastError();
}

// Workspaces clo_hip_scan_workspace_init has prepared, with the byte count it was given
// (host side, this process). A scan keeps its workspace consistent itself and clears
// nothing per call, and where its accumulators live depends on the workspace's SIZE: a
// range that was never initialised, or a call that passes another byte count than the
// one the range was initialised with (one workspace for the largest n and
// clo_hip_scan_workspace_bytes(n) per call — the natural usage while every call cleared
// its workspace), would read garbage as counters and return wrong sums with no
// status raised. Such calls are refused (CLO_HIP_EWORKSPACE) instead.
struct scan_ws_entry { void* ptr; size_t bytes; };
std::mutex g_scan_ws_mutex;
std::vector<scan_ws_entry> g_scan_ws;
constexpr size_t SCAN_WS_MAX_ENTRIES = 1024;

void scan_ws_remember(void* ptr, size_t bytes) {   // bytes == 0: forget whatever starts inside [ptr, ptr + 1)
	std::lock_guard<std::mutex> lock(g_scan_ws_mutex);
	const char* lo = (const char*) ptr;
	const char* hi = lo + (bytes ? bytes : 1);
	for (size_t i = 0; i < g_scan_ws.size(); ) {   // anything that overlaps the range is gone
		const char* a = (const char*) g_scan_ws[i].ptr;
		if (a < hi && lo < a + g_scan_ws[i].bytes) { g_scan_ws[i] = g_scan_ws.back(); g_scan_ws.pop_back(); }
		else ++i;
	}
	if (bytes == 0) return;
	if (g_scan_ws.size() >= SCAN_WS_MAX_ENTRIES) g_scan_ws.erase(g_scan_ws.begin());
	g_scan_ws.push_back({ ptr, bytes });
}

bool scan_ws_known(void* ptr, size_t bytes) {
	std::lock_guard<std::mutex> lock(g_scan_ws_mutex);
	for (const scan_ws_entry& e : g_scan_ws) if (e.ptr == ptr) return e.bytes == bytes;
	return false;
}

}  // namespace

extern "C" {

size_t clo_hip_scan_workspace_bytes(size_t numel, int elem_size, int sum_size) {
	(void) elem_size;
	// sized for the small work-group shape, so that the size grows with numel
	const size_t tile = scan_tile_elems(0, sum_size);
	const size_t tiles = (numel + tile - 1) / tile;
	const size_t t = tiles ? tiles : 1;
	return CLO_WS_HEADER_BYTES + scan_ws_granules(t) * sizeof(clo_u64);
}

int clo_hip_scan_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
	if (!workspace || workspace_bytes < CLO_WS_HEADER_BYTES) return CLO_HIP_EARGS;
	const hipError_t e = hipMemsetAsync(workspace, 0, workspace_bytes, (hipStream_t) stream);
	if (e != hipSuccess) return (int) e;
	scan_ws_remember(workspace, workspace_bytes);
	return 0;
}

int clo_hip_scan_workspace_forget(void* workspace) {
	if (!workspace) return CLO_HIP_EARGS;
	scan_ws_remember(workspace, 0);
	return 0;
}

int clo_hip_scan_workspace_set_epoch(void* workspace, unsigned epoch, void* stream) {
	if (!workspace) return CLO_HIP_EARGS;
	return (int) hipMemsetD32Async((hipDeviceptr_t) ((char*) workspace + CLO_WS_EPOCH_WORD * sizeof(unsigned)), (int) epoch, 1,
		(hipStream_t) stream);
}

int clo_hip_scan_exclusive_carry(const void* data_in, void* data_out, size_t numel,
	int elem_size, int elem_signed, int sum_size,
	const uint64_t* carry_in_dev, uint64_t* carry_out_dev,
	void* workspace, size_t workspace_bytes, void* stream) {

	hipStream_t s = (hipStream_t) stream;
	if (numel == 0) {   // nothing to scan: the carry passes through
		if (!carry_out_dev) return 0;
		if (carry_in_dev) return (int) hipMemcpyAsync(carry_out_dev, carry_in_dev, sizeof(uint64_t), hipMemcpyDeviceToDevice, s);
		return (int) hipMemsetAsync(carry_out_dev, 0, sizeof(uint64_t), s);
	}
	if (!data_in || !data_out || !workspace) return CLO_HIP_EARGS;
	if (sum_size < elem_size) return CLO_HIP_EUNSUPPORTED;
	g_scan_max_spins = clo_hip_env()->max_spins;
	if (workspace_bytes < clo_hip_scan_workspace_bytes(numel, elem_size, sum_size)) return CLO_HIP_EWORKSPACE;
	// the place of the accumulators follows from the workspace's size, and nothing is cleared per call:
	// this must be the range clo_hip_scan_workspace_init prepared, under the byte count it was given
	if (!scan_ws_known(workspace, workspace_bytes)) return CLO_HIP_EWORKSPACE;
	if (numel / scan_tile_elems(numel, sum_size) >= 0x7fffffffull) return CLO_HIP_EARGS;
	const clo_u64* ci = (const clo_u64*) carry_in_dev;
	clo_u64* co = (clo_u64*) carry_out_dev;
	// The sum type only matters by width: two's complement addition is the
	// same for signed and unsigned sums.
	switch (sum_size) {
		case 1: return dispatch_in<uint8_t>(data_in, data_out, numel, elem_size, elem_signed, ci, co, workspace, workspace_bytes, s);
		case 2: return dispatch_in<uint16_t>(data_in, data_out, numel, elem_size, elem_signed, ci, co, workspace, workspace_bytes, s);
		case 4: return dispatch_in<uint32_t>(data_in, data_out, numel, elem_size, elem_signed, ci, co, workspace, workspace_bytes, s);
		case 8: return dispatch_in<uint64_t>(data_in, data_out, numel, elem_size, elem_signed, ci, co, workspace, workspace_bytes, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

int clo_hip_scan_exclusive(const void* data_in, void* data_out, size_t numel,
	int elem_size, int elem_signed, int sum_size,
	void* workspace, size_t workspace_bytes, void* stream) {
	return clo_hip_scan_exclusive_carry(data_in, data_out, numel, elem_size, elem_signed, sum_size,
		nullptr, nullptr, workspace, workspace_bytes, stream);
}

int clo_hip_reduce_sum(const void* data_in, size_t numel, int elem_size, int elem_signed,
	uint64_t* total_dev, void* stream) {
	if (!total_dev) return CLO_HIP_EARGS;
	hipStream_t s = (hipStream_t) stream;
	hipError_t e = hipMemsetAsync(total_dev, 0, sizeof(uint64_t), s);
	if (e != hipSuccess) return (int) e;
	if (numel == 0) return 0;
	if (!data_in) return CLO_HIP_EARGS;
	switch (elem_size) {
		case 1: return elem_signed ? launch_reduce<int8_t>(data_in, numel, total_dev, s) : launch_reduce<uint8_t>(data_in, numel, total_dev, s);
		case 2: return elem_signed ? launch_reduce<int16_t>(data_in, numel, total_dev, s) : launch_reduce<uint16_t>(data_in, numel, total_dev, s);
		case 4: return elem_signed ? launch_reduce<int32_t>(data_in, numel, total_dev, s) : launch_reduce<uint32_t>(data_in, numel, total_dev, s);
		case 8: return launch_reduce<uint64_t>(data_in, numel, total_dev, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

}  // extern "C"
