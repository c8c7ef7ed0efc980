// clo_hip_radix.hip — stable LSD radix sort for gfx950 ("satradix" replacement).
//
// Upstream, every digit pass is four launches over four buffers
// (sort/clo_sort_satradix.c:264-313): satradix_localsort (b one-bit splits, each a
// full LDS Blelloch scan over the tile), satradix_histogram, a 3-kernel scan of
// num_wgs*radix counters, satradix_scatter (sort/clo_sort_satradix.cl:34-258) —
// about 5 element streams plus 6 counter streams through HBM per digit.
//
// Here a digit pass is ONE kernel that reads every element once and writes it
// once (chained-scan "onesweep" structure):
//   1. a work-group draws a ticket = tile id, loads its tile wave-striped
//      (lane l of wave w holds element w*64*ITEMS + i*64 + l of the tile);
//   2. ranks each element among equal digits of the tile, stably: per item one
//      __ballot per digit bit gives the lanes holding the same digit (match-any),
//      v_mbcnt gives the rank inside the wave item, a per-wave LDS counter row
//      carries the count across items; a small cross-wave pass makes the tile
//      histogram and per-wave offsets (this is the job of upstream's localsort +
//      histogram kernels, with no one-bit split loop and no barriers inside it);
//   3. publishes the tile histogram and resolves its global offsets per digit
//      by decoupled look-back over earlier tiles (8-byte {tag,count} granules,
//      agent-scope relaxed accesses) — upstream's counters scan;
//   4. scatters the tile into digit order through an LDS stage so that HBM
//      writes are contiguous runs per digit (upstream's scatter kernel);
//   5. while the keys are in registers, counts the NEXT pass's digit so no
//      separate histogram read is needed except before the first pass.
// Stability per pass + LSD order give exactly the order the reference produces
// (stable ascending by key), for any digit width.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "clo_hip.h"
#include "clo_hip_internal.h"

namespace {

// ---------------------------------------------------------------------------
// first-pass digit histogram
// ---------------------------------------------------------------------------

constexpr int HIST_THREADS = 256;
constexpr int HIST_ITEMS = 16;  // elements per thread per block-iteration
constexpr unsigned GH_COPIES = 256;  // partial global histograms per pass

template <typename E>
__global__ __launch_bounds__(HIST_THREADS)
void clo_radix_hist_kernel(const E* __restrict__ in, size_t n, unsigned shift, unsigned mask,
	unsigned* __restrict__ ghist32, unsigned stride, unsigned long long* __restrict__ ghist64) {

	__shared__ unsigned h[HIST_THREADS / 64][256];
	const unsigned tid = threadIdx.x, wave = tid >> 6;
	for (unsigned i = tid; i < (HIST_THREADS / 64) * 256; i += HIST_THREADS) (&h[0][0])[i] = 0;
	__syncthreads();

	const size_t chunk = (size_t) HIST_THREADS * HIST_ITEMS;
	for (size_t base = (size_t) blockIdx.x * chunk; base < n; base += (size_t) gridDim.x * chunk) {
		#pragma unroll
		for (int i = 0; i < HIST_ITEMS; ++i) {
			const size_t idx = base + (size_t) i * HIST_THREADS + tid;
			if (idx < n) {
				const unsigned d = (unsigned) (in[idx] >> shift) & mask;
				atomicAdd(&h[wave][d], 1u);
			}
		}
	}
	__syncthreads();
	if (tid <= mask) {
		unsigned s = 0;
		#pragma unroll
		for (int w = 0; w < HIST_THREADS / 64; ++w) s += h[w][tid];
		if (s) {
			// ghist32 is GH_COPIES partial histograms of `stride` counters: spreading
			// the adds keeps same-address atomics (one per ~12 ns) off the critical path
			if (ghist32) atomicAdd(&ghist32[(size_t) (blockIdx.x % GH_COPIES) * stride + tid], s);
			if (ghist64) atomicAdd(&ghist64[tid], (unsigned long long) s);
		}
	}
}

// Sum the partial histograms of one pass and turn them into global digit
// bases (exclusive scan over digits). One work-group of 256 threads.
__global__ __launch_bounds__(256)
void clo_radix_bases_kernel(const unsigned* __restrict__ parts, unsigned stride, unsigned R, unsigned* __restrict__ gbase) {
	__shared__ unsigned s_tmp[4];
	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	unsigned c = 0;
	if (tid < R)
		for (unsigned k = 0; k < GH_COPIES; ++k) c += parts[(size_t) k * stride + tid];
	const unsigned incl = clo_wave_scan_inclusive<unsigned>(c, lane);
	if (lane == 63) s_tmp[wave] = incl;
	__syncthreads();
	unsigned add = 0;
	for (unsigned w = 0; w < wave; ++w) add += s_tmp[w];
	if (tid < R) gbase[tid] = incl - c + add;
}

// ---------------------------------------------------------------------------
// one digit pass
// ---------------------------------------------------------------------------

// Exclusive scan over the R (<= 256) values held by threads 0..R-1 (others pass
// 0). Every thread of the block must call it.
template <int THREADS>
__device__ __forceinline__ unsigned block_excl_scan(unsigned x, unsigned tid, unsigned* s_tmp) {
	const unsigned lane = tid & 63u, wave = tid >> 6;
	const unsigned incl = clo_wave_scan_inclusive<unsigned>(x, lane);
	if (lane == 63 && wave < 4) s_tmp[wave] = incl;
	__syncthreads();
	unsigned add = 0;
	#pragma unroll
	for (unsigned w = 0; w < 4; ++w) if (w < wave) add += s_tmp[w];
	return incl - x + add;
}

// Serial tail of the look-back for one digit: polls predecessors j, j-1, ...
// (4 granules in flight per poll) until one carries an inclusive prefix.
__device__ __forceinline__ unsigned radix_lookback_serial(const clo_u64* state, unsigned R, unsigned d,
	long j, unsigned excl, unsigned epoch, unsigned* status) {
	unsigned spins = 0;
	bool done = j < 0;
	while (!done) {
		clo_u64 g[4];
		#pragma unroll
		for (int k = 0; k < 4; ++k)
			g[k] = (j - k >= 0) ? clo_ld_agent(state + (size_t) (j - k) * R + d) : 0ull;
		bool stalled = false;
		#pragma unroll
		for (int k = 0; k < 4; ++k) {
			if (done || stalled) continue;
			if (j - k < 0) { done = true; continue; }
			const unsigned tag = clo_lb_tag(g[k]);
			const unsigned st = tag & 3u;
			if ((tag >> 2) != epoch || st == 0u) {
				j -= k;
				stalled = true;
			} else {
				excl += clo_lb_val(g[k]);
				if (st == CLO_LB_PREFIX) done = true;
			}
		}
		if (!done && !stalled) j -= 4;
		if (stalled) {
			if (++spins > CLO_MAX_SPINS) {
				atomicExch(status, 1u);
				done = true;
			}
			__builtin_amdgcn_s_sleep(1);
		}
	}
	return excl;
}

template <typename E, int BITS, int THREADS, int ITEMS, int ROUNDS>
struct radix_smem {
	static constexpr int R = 1 << BITS;
	static constexpr int WAVES = THREADS / 64;
	static constexpr int STAGE = THREADS * ITEMS / ROUNDS;
	E stage[STAGE];
	unsigned wcnt[WAVES][R];   // per-wave digit counts, then running tile-local position of (wave, digit)
	unsigned next[WAVES][R];   // next pass's digit counts
	clo_u64 lb[THREADS];       // look-back window, [THREADS / R][R]
	unsigned delta[R];         // global index = tile-local position + delta[digit]
	unsigned tmp[4];
	unsigned tile;
};

template <bool FULL, typename E, int BITS, int THREADS, int ITEMS, int ROUNDS>
__device__ __forceinline__ void radix_pass_body(radix_smem<E, BITS, THREADS, ITEMS, ROUNDS>& sm,
	const E* __restrict__ in, E* __restrict__ out, size_t n, size_t base, unsigned count, unsigned tile,
	unsigned shift, unsigned mask, int has_next, unsigned next_shift, unsigned next_mask,
	unsigned* hdr, const unsigned* __restrict__ gbase_cur, unsigned* __restrict__ ghist_next,
	clo_u64* state, unsigned epoch, unsigned xflags, unsigned long long* dbg) {

	constexpr int R = 1 << BITS;
	constexpr int WAVES = THREADS / 64;
	constexpr int STAGE = THREADS * ITEMS / ROUNDS;
	constexpr int LBW = THREADS / R;                 // threads per digit in the look-back window
	constexpr int LBK = LBW >= 4 ? 1 : 4 / LBW;      // predecessors per thread (window = LBW * LBK >= 4)
	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const unsigned wbase = wave * 64u * ITEMS + lane;
	#define CLO_STAMP(k) do { if (dbg && tid == 0 && tile < 32768u) dbg[(size_t) tile * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
	CLO_STAMP(0);

	// ---- 1. load, wave-striped: lane l of wave w holds tile element w*64*ITEMS + i*64 + l ----
	E key[ITEMS];
	#pragma unroll
	for (int i = 0; i < ITEMS; ++i) {
		if (FULL) key[i] = in[base + wbase + i * 64];
		else key[i] = (wbase + i * 64 < count) ? in[base + wbase + i * 64] : (E) 0;
	}
	if (dbg) { asm volatile("" :: "v"((unsigned) key[ITEMS - 1])); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
	CLO_STAMP(1);

	// ---- 5. next pass's digit counts (order-independent, so done early) ----
	if (has_next && !(xflags & 2u)) {
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) {
			if (FULL || wbase + i * 64 < count)
				atomicAdd(&sm.next[wave][(unsigned) (key[i] >> next_shift) & next_mask], 1u);
		}
	}

	// ---- 2a. match: per item, the lanes of my wave holding my digit ----
	// One ballot per digit bit gives the group (match-any); v_mbcnt my rank in
	// it. ONE lane per distinct digit (the group's first) adds the group size
	// to the wave's digit count: distinct LDS addresses within the instruction,
	// so no atomic conflicts. (rank, size, leader lane) stay packed in a VGPR.
	unsigned grp[ITEMS];
	#pragma unroll
	for (int i = 0; i < ITEMS; ++i) {
		const bool valid = FULL || (wbase + i * 64 < count);
		const unsigned d = (unsigned) (key[i] >> shift) & mask;
		clo_u64 peers = FULL ? ~0ull : __ballot(valid);
		#pragma unroll
		for (int k = 0; k < BITS; ++k) {
			const bool bit = (d >> k) & 1u;
			const clo_u64 b = __ballot(bit);
			peers &= bit ? b : ~b;
		}
		const unsigned r = clo_mbcnt(peers);
		const unsigned c = (unsigned) __popcll(peers);
		const unsigned leader = valid ? (unsigned) (__ffsll((long long) peers) - 1) : lane;
		if (valid && r == 0) atomicAdd(&sm.wcnt[wave][d], c);
		grp[i] = r | (c << 8) | (leader << 16);
	}
	if (dbg) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	CLO_STAMP(2);
	__syncthreads();
	CLO_STAMP(3);

	// ---- tile histogram; publish it; start of every (wave, digit) run ----
	unsigned hist = 0, cw[WAVES];
	if (tid < (unsigned) R) {
		#pragma unroll
		for (int w = 0; w < WAVES; ++w) { cw[w] = sm.wcnt[w][tid]; hist += cw[w]; }
		clo_st_agent(state + (size_t) tile * R + tid,
			clo_lb_pack(epoch, tile == 0 ? CLO_LB_PREFIX : CLO_LB_AGG, hist));
	}
	const unsigned dstart = block_excl_scan<THREADS>(hist, tid, sm.tmp);
	if (tid < (unsigned) R) {
		unsigned run = dstart;
		#pragma unroll
		for (int w = 0; w < WAVES; ++w) { sm.wcnt[w][tid] = run; run += cw[w]; }
	}

	// ---- 3a. look-back window: every thread fetches the granules of LBK
	// predecessors of one digit; they are in flight during step 2b/4a ----
	const unsigned lb_d = tid % R, lb_q = tid / R;
	clo_u64 g[LBK];
	#pragma unroll
	for (int k = 0; k < LBK; ++k) {
		const long j = (long) tile - 1 - (long) (lb_q * LBK + k);
		g[k] = (j >= 0) ? clo_ld_agent(state + (size_t) j * R + lb_d) : 0ull;
	}
	__syncthreads();
	CLO_STAMP(4);

	// ---- 2b. rank: the group's first lane takes the group's slice of the
	// (wave, digit) run with a returning LDS atomic (a wave's LDS atomics run in
	// issue order, so slices follow item order = stable); ds_bpermute hands the
	// slice start to the group. 4a. scatter into the LDS stage. ----
	#pragma unroll
	for (int round = 0; round < ROUNDS; ++round) {
		const unsigned lo = (unsigned) round * STAGE;
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) {
			const bool valid = FULL || (wbase + i * 64 < count);
			if (round == 0) {
				const unsigned d = (unsigned) (key[i] >> shift) & mask;
				const unsigned r = grp[i] & 0xffu, c = (grp[i] >> 8) & 0xffu, leader = grp[i] >> 16;
				unsigned start = 0;
				if (valid && r == 0) start = atomicAdd(&sm.wcnt[wave][d], c);
				start = (unsigned) __shfl((int) start, (int) leader, 64);
				grp[i] = start + r;  // tile-local position
			}
			const unsigned p = grp[i] - lo;
			if (ROUNDS == 1) { if (valid) sm.stage[p] = key[i]; }
			else if (valid && p < (unsigned) STAGE) sm.stage[p] = key[i];
		}
		if (round == 0) {
			CLO_STAMP(5);
			// ---- 3b. reduce the window: nearest predecessor first ----
			{
				unsigned sum = 0, st = 3u, idx = 0;  // st: 3 = all aggregates, 2 = prefix found, 0 = stalled at idx
				#pragma unroll
				for (int k = 0; k < LBK; ++k) {
					if (st != 3u) continue;
					const long j = (long) tile - 1 - (long) (lb_q * LBK + k);
					if (j < 0) { st = CLO_LB_PREFIX; continue; }
					const unsigned tag = clo_lb_tag(g[k]);
					if ((tag >> 2) != epoch || (tag & 3u) == 0u) { st = 0u; idx = (unsigned) k; continue; }
					sum += clo_lb_val(g[k]);
					if ((tag & 3u) == CLO_LB_PREFIX) st = CLO_LB_PREFIX;
				}
				sm.lb[lb_q * R + lb_d] = ((clo_u64) ((idx << 2) | st) << 32) | sum;
			}
			__syncthreads();
			if (tid < (unsigned) R) {
				unsigned excl = 0;
				if (tile != 0) {
					long resume = -1;  // predecessor to resume serial polling at, if the window did not close
					bool closed = false;
					for (int q = 0; q < LBW && !closed && resume < 0; ++q) {
						const clo_u64 w = sm.lb[q * R + tid];
						const unsigned st = (unsigned) (w >> 32) & 3u, idx = (unsigned) (w >> 34);
						excl += (unsigned) w;
						if (st == CLO_LB_PREFIX) closed = true;
						else if (st == 0u) resume = (long) tile - 1 - (long) (q * LBK + idx);
					}
					if (!closed) {
						if (resume < 0) resume = (long) tile - 1 - (long) (LBW * LBK);
						excl = radix_lookback_serial(state, R, tid, resume, excl, epoch, &hdr[0]);
					}
					clo_st_agent(state + (size_t) tile * R + tid, clo_lb_pack(epoch, CLO_LB_PREFIX, excl + hist));
				}
				sm.delta[tid] = gbase_cur[tid] + excl - dstart;
			}
			CLO_STAMP(6);
		}
		__syncthreads();
		// ---- 4b. contiguous runs to HBM ----
		#pragma unroll
		for (int j = 0; j < STAGE / THREADS; ++j) {
			const unsigned p = lo + j * THREADS + tid;
			if (FULL || p < count) {
				const E e = sm.stage[p - lo];
				const unsigned d = (unsigned) (e >> shift) & mask;
				// (bounded even if a look-back gave up and delta is garbage)
				const size_t gi = (size_t) (unsigned) (p + sm.delta[d]);
				if (gi < n) out[gi] = e;
			}
		}
		if (round + 1 < ROUNDS) __syncthreads();
	}
	if (dbg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	CLO_STAMP(7);
	#undef CLO_STAMP

	// ---- 5b. hand the next pass its (partial) global histogram ----
	if (has_next && tid <= next_mask) {
		unsigned s = 0;
		#pragma unroll
		for (int w = 0; w < WAVES; ++w) s += sm.next[w][tid];
		if (s) atomicAdd(&ghist_next[(size_t) (tile % GH_COPIES) * R + tid], s);
	}
}

template <typename E, int BITS, int THREADS, int ITEMS, int ROUNDS, int MINW>
__global__ __launch_bounds__(THREADS, MINW)
void clo_radix_pass_kernel(const E* __restrict__ in, E* __restrict__ out, size_t n,
	unsigned shift, unsigned mask,
	int has_next, unsigned next_shift, unsigned next_mask,
	unsigned* hdr, unsigned ticket_word,
	const unsigned* __restrict__ gbase_cur, unsigned* __restrict__ ghist_next,
	clo_u64* state, unsigned epoch, unsigned xflags, unsigned long long* dbg) {

	constexpr int R = 1 << BITS;
	constexpr int WAVES = THREADS / 64;
	constexpr int TILE = THREADS * ITEMS;
	static_assert(R <= THREADS, "one thread per digit");
	static_assert((TILE / ROUNDS) % THREADS == 0, "stage is read back in whole rows");

	__shared__ radix_smem<E, BITS, THREADS, ITEMS, ROUNDS> sm;
	const unsigned tid = threadIdx.x;

	// tile id = ticket: ids follow dispatch order, so every predecessor a tile
	// waits for in the look-back has already been handed to a running group
	if (tid == 0) sm.tile = (xflags & 4u) ? blockIdx.x : atomicAdd(&hdr[ticket_word], 1u);
	for (unsigned i = tid; i < WAVES * R; i += THREADS) {
		(&sm.wcnt[0][0])[i] = 0;
		(&sm.next[0][0])[i] = 0;
	}
	__syncthreads();
	const unsigned tile = sm.tile;
	const size_t base = (size_t) tile * TILE;
	if (base >= n) return;
	const unsigned count = (n - base) < (size_t) TILE ? (unsigned) (n - base) : (unsigned) TILE;

	if (count == (unsigned) TILE)
		radix_pass_body<true, E, BITS, THREADS, ITEMS, ROUNDS>(sm, in, out, n, base, count, tile, shift, mask,
			has_next, next_shift, next_mask, hdr, gbase_cur, ghist_next, state, epoch, xflags, dbg);
	else
		radix_pass_body<false, E, BITS, THREADS, ITEMS, ROUNDS>(sm, in, out, n, base, count, tile, shift, mask,
			has_next, next_shift, next_mask, hdr, gbase_cur, ghist_next, state, epoch, xflags, dbg);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

int g_variant = 0;
unsigned long long* g_dbg = nullptr;  // developer stamps buffer (8 u64 per tile), diagnostic runs only
unsigned g_xflags = 0;  // developer experiments (CLO_RADIX_XFLAGS), never set in production

// Tile shapes (threads, items per thread, LDS stage rounds). The LDS stage is
// 32 KiB for 4/8-byte elements in every shape.
template <typename E> struct shape0 { static constexpr int T = 512, I = (sizeof(E) == 8 ? 8 : 16), RD = 1, W = 1; };
template <typename E> struct shape1 { static constexpr int T = 256, I = (sizeof(E) == 8 ? 8 : 16), RD = 1, W = 1; };
template <typename E> struct shape2 { static constexpr int T = 1024, I = (sizeof(E) == 8 ? 8 : 16), RD = 2, W = 1; };

size_t tile_elems(int elem_size, int variant) {
	const bool wide = elem_size == 8;
	switch (variant) {
		case 1: return 256u * (wide ? 8 : 16);
		case 2: return 1024u * (wide ? 8 : 16);
		default: return 512u * (wide ? 8 : 16);
	}
}

struct ws_layout { size_t ghist, gbase, state, total, tiles; };

ws_layout radix_layout(size_t n, int elem_size, int passes, int digit_bits, int variant) {
	ws_layout L;
	const size_t R = (size_t) 1 << digit_bits;
	const size_t tile = tile_elems(elem_size, variant);
	L.tiles = (n + tile - 1) / tile;
	if (L.tiles == 0) L.tiles = 1;
	L.ghist = CLO_WS_HEADER_BYTES;
	const size_t gh = (size_t) (passes + 1) * GH_COPIES * R * sizeof(unsigned);
	L.gbase = L.ghist + gh;
	const size_t gb = ((size_t) passes * R * sizeof(unsigned) + 255) & ~(size_t) 255;
	L.state = L.gbase + gb;
	L.total = L.state + L.tiles * R * sizeof(clo_u64);
	return L;
}

template <typename E, int BITS, typename S>
void launch_pass(const E* in, E* out, size_t n, unsigned shift, unsigned mask,
	int has_next, unsigned nshift, unsigned nmask, unsigned* hdr, unsigned ticket_word,
	const unsigned* gh_cur, unsigned* gh_next, clo_u64* state, unsigned epoch,
	size_t tiles, hipStream_t s, const char* label = "radix_pass") {
	clo_timing_scope timing(label, s);
	hipLaunchKernelGGL((clo_radix_pass_kernel<E, BITS, S::T, S::I, S::RD, S::W>),
		dim3((unsigned) tiles), dim3(S::T), 0, s,
		in, out, n, shift, mask, has_next, nshift, nmask, hdr, ticket_word,
		gh_cur, gh_next, state, epoch, g_xflags, g_dbg);
}

template <typename E, int BITS>
void launch_pass_variant(int variant, const E* in, E* out, size_t n, unsigned shift, unsigned mask,
	int has_next, unsigned nshift, unsigned nmask, unsigned* hdr, unsigned ticket_word,
	const unsigned* gh_cur, unsigned* gh_next, clo_u64* state, unsigned epoch,
	size_t tiles, hipStream_t s) {
	if constexpr (BITS == 4 && sizeof(E) >= 4) {
		if (variant == 1) {
			launch_pass<E, BITS, shape1<E>>(in, out, n, shift, mask, has_next, nshift, nmask, hdr, ticket_word, gh_cur, gh_next, state, epoch, tiles, s);
			return;
		}
		if (variant == 2) {
			launch_pass<E, BITS, shape2<E>>(in, out, n, shift, mask, has_next, nshift, nmask, hdr, ticket_word, gh_cur, gh_next, state, epoch, tiles, s);
			return;
		}
	}
	launch_pass<E, BITS, shape0<E>>(in, out, n, shift, mask, has_next, nshift, nmask, hdr, ticket_word, gh_cur, gh_next, state, epoch, tiles, s);
}

// Variants other than 0 exist only for 4-bit digits on 4/8-byte elements.
int effective_variant(int elem_size, int digit_bits) {
	return (digit_bits == 4 && elem_size >= 4 && g_variant < 3) ? g_variant : 0;
}

template <typename E, int BITS>
int radix_sort_impl(const E* src, E* dst, E* tmp, size_t n, int key_shift, int key_bits,
	void* ws, hipStream_t s) {

	const int passes = (key_bits + BITS - 1) / BITS;
	const int variant = effective_variant((int) sizeof(E), BITS);
	const ws_layout L = radix_layout(n, (int) sizeof(E), passes, BITS, variant);
	unsigned* hdr = (unsigned*) ws;
	unsigned* ghist = (unsigned*) ((char*) ws + L.ghist);
	unsigned* gbase = (unsigned*) ((char*) ws + L.gbase);
	clo_u64* state = (clo_u64*) ((char*) ws + L.state);
	constexpr unsigned R = 1u << BITS;
	constexpr size_t PART = (size_t) GH_COPIES * R;  // counters per pass

	hipError_t e = hipMemsetAsync(ws, 0, L.total, s);
	if (e != hipSuccess) return (int) e;

	const unsigned bits0 = key_bits < BITS ? key_bits : BITS;
	unsigned hist_blocks = (unsigned) ((n + HIST_THREADS * HIST_ITEMS - 1) / (HIST_THREADS * HIST_ITEMS));
	if (hist_blocks > 2048) hist_blocks = 2048;
	{
		clo_timing_scope timing("radix_hist", s);
		hipLaunchKernelGGL((clo_radix_hist_kernel<E>), dim3(hist_blocks), dim3(HIST_THREADS), 0, s,
			src, n, (unsigned) key_shift, (1u << bits0) - 1u, ghist, R, (unsigned long long*) nullptr);
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
		{
			clo_timing_scope timing("radix_bases", s);
			hipLaunchKernelGGL(clo_radix_bases_kernel, dim3(1), dim3(256), 0, s,
				(const unsigned*) (ghist + (size_t) p * PART), R, R, gbase + (size_t) p * R);
		}
		launch_pass_variant<E, BITS>(variant, cur_in, cur_out, n,
			(unsigned) (key_shift + p * BITS), (1u << bits) - 1u,
			has_next, (unsigned) (key_shift + (p + 1) * BITS), (1u << nbits) - 1u,
			hdr, (unsigned) (CLO_WS_TICKET_WORD + p),
			gbase + (size_t) p * R, ghist + (size_t) (p + 1) * PART, state, (unsigned) (p + 1),
			L.tiles, s);
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
int radix_dispatch_bits(const void* src, void* dst, void* tmp, size_t n, int key_shift, int key_bits,
	int digit_bits, void* ws, hipStream_t s) {
	switch (digit_bits) {
		case 1: return radix_sort_impl<E, 1>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, ws, s);
		case 2: return radix_sort_impl<E, 2>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, ws, s);
		case 3: return radix_sort_impl<E, 3>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, ws, s);
		case 4: return radix_sort_impl<E, 4>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, ws, s);
		case 5: return radix_sort_impl<E, 5>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, ws, s);
		case 6: return radix_sort_impl<E, 6>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, ws, s);
		case 7: return radix_sort_impl<E, 7>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, ws, s);
		case 8: return radix_sort_impl<E, 8>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, ws, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

// ---- MSD bucket split = one histogram + one stable pass on the top bits ----

template <typename E>
int msd_hist_impl(const void* src, size_t n, unsigned shift, unsigned mask, uint64_t* counts, hipStream_t s) {
	unsigned blocks = (unsigned) ((n + HIST_THREADS * HIST_ITEMS - 1) / (HIST_THREADS * HIST_ITEMS));
	if (blocks > 2048) blocks = 2048;
	if (blocks == 0) blocks = 1;
	hipLaunchKernelGGL((clo_radix_hist_kernel<E>), dim3(blocks), dim3(HIST_THREADS), 0, s,
		(const E*) src, n, shift, mask, (unsigned*) nullptr, 0u, (unsigned long long*) counts);
	return (int) hipGetLastError();
}

}  // namespace

extern "C" {

int clo_hip_radix_set_debug_buffer(void* dptr) {
	g_dbg = (unsigned long long*) dptr;
	clo_radix4_set_debug_buffer(dptr);
	return 0;
}

int clo_hip_radix_set_variant(int variant) {
	// 0: default (chain-free path for digits <= 4 bits, look-back path above);
	// 1, 2: other tile shapes of the look-back path; 3: look-back path always
	// 4: default path with 8192-element tiles for 4-byte elements (default 4096);
	// 5: default path with the match-any kernels instead of the packed-counter ones
	if (variant < 0 || variant > 5) return CLO_HIP_EARGS;
	clo_radix4_set_log_tile(variant == 4 ? 13 : 12);
	clo_radix4_set_match(variant == 4 || variant == 5);
	if (variant >= 4) variant = 0;
	const char* x = getenv("CLO_RADIX_XFLAGS");
	g_xflags = x ? (unsigned) atoi(x) : 0u;
	g_variant = variant;
	return 0;
}

size_t clo_hip_radix_workspace_bytes(size_t numel, int elem_size, int key_bits, int digit_bits) {
	if (digit_bits < 1 || digit_bits > 8 || key_bits < 1) return 0;
	const int passes = (key_bits + digit_bits - 1) / digit_bits;
	// sized for the smallest tile of any variant so the knob can change later
	size_t worst = 0;
	for (int v = 0; v < 3; ++v) {
		const size_t t = radix_layout(numel, elem_size, passes, digit_bits, v).total;
		if (t > worst) worst = t;
	}
	if (digit_bits <= 4) {
		const size_t t = clo_radix4_workspace_bytes(numel, elem_size, key_bits, digit_bits);
		if (t > worst) worst = t;
	}
	return worst;
}

int clo_hip_radix_sort(const void* src, void* dst, void* tmp, size_t numel,
	int elem_size, int key_shift, int key_bits, int digit_bits,
	void* workspace, size_t workspace_bytes, void* stream) {

	if (numel == 0) return 0;
	if (!src || !dst || !tmp || !workspace || tmp == src || tmp == dst) return CLO_HIP_EARGS;
	if (key_bits < 1 || key_shift < 0 || key_shift + key_bits > 8 * elem_size) return CLO_HIP_EARGS;
	if (digit_bits < 1 || digit_bits > 8) return CLO_HIP_EUNSUPPORTED;
	if (numel > 0xffffffffull) return CLO_HIP_EARGS;  // 32-bit positions, as upstream's uint indices
	if ((key_bits + digit_bits - 1) / digit_bits > CLO_WS_MAX_PASSES) return CLO_HIP_EARGS;
	if (workspace_bytes < clo_hip_radix_workspace_bytes(numel, elem_size, key_bits, digit_bits)) return CLO_HIP_EWORKSPACE;
	hipStream_t s = (hipStream_t) stream;
	if (digit_bits <= 4 && g_variant == 0)
		return clo_radix4_sort(src, dst, tmp, numel, elem_size, key_shift, key_bits, digit_bits, workspace, s);
	switch (elem_size) {
		case 1: return radix_dispatch_bits<uint8_t>(src, dst, tmp, numel, key_shift, key_bits, digit_bits, workspace, s);
		case 2: return radix_dispatch_bits<uint16_t>(src, dst, tmp, numel, key_shift, key_bits, digit_bits, workspace, s);
		case 4: return radix_dispatch_bits<uint32_t>(src, dst, tmp, numel, key_shift, key_bits, digit_bits, workspace, s);
		case 8: return radix_dispatch_bits<uint64_t>(src, dst, tmp, numel, key_shift, key_bits, digit_bits, workspace, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

size_t clo_hip_msd_workspace_bytes(size_t numel, int elem_size, int bucket_bits) {
	if (bucket_bits < 1 || bucket_bits > 3) return 0;
	return clo_radix4_partition_workspace_bytes(numel, elem_size, bucket_bits);
}

int clo_hip_msd_histogram(const void* src, size_t numel, int elem_size,
	int key_shift, int key_bits, int bucket_bits, uint64_t* counts_dev, void* stream) {
	if (!counts_dev || bucket_bits < 1 || bucket_bits > 3 || bucket_bits > key_bits) return CLO_HIP_EARGS;
	hipStream_t s = (hipStream_t) stream;
	hipError_t e = hipMemsetAsync(counts_dev, 0, sizeof(uint64_t) << bucket_bits, s);
	if (e != hipSuccess) return (int) e;
	if (numel == 0) return 0;
	if (!src) return CLO_HIP_EARGS;
	const unsigned shift = (unsigned) (key_shift + key_bits - bucket_bits), mask = (1u << bucket_bits) - 1u;
	switch (elem_size) {
		case 4: return msd_hist_impl<uint32_t>(src, numel, shift, mask, counts_dev, s);
		case 8: return msd_hist_impl<uint64_t>(src, numel, shift, mask, counts_dev, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

int clo_hip_msd_partition(const void* src, void* dst, size_t numel, int elem_size,
	int key_shift, int key_bits, int bucket_bits, uint64_t* counts_dev,
	void* workspace, size_t workspace_bytes, void* stream) {
	hipStream_t s = (hipStream_t) stream;
	if (bucket_bits < 1 || bucket_bits > 3 || bucket_bits > key_bits) return CLO_HIP_EARGS;
	if (numel == 0) {
		if (counts_dev) return (int) hipMemsetAsync(counts_dev, 0, sizeof(uint64_t) << bucket_bits, s);
		return 0;
	}
	if (!src || !dst || src == dst || !workspace) return CLO_HIP_EARGS;
	if (numel > 0xffffffffull) return CLO_HIP_EARGS;
	if (elem_size != 4 && elem_size != 8) return CLO_HIP_EUNSUPPORTED;
	if (workspace_bytes < clo_hip_msd_workspace_bytes(numel, elem_size, bucket_bits)) return CLO_HIP_EWORKSPACE;
	const unsigned shift = (unsigned) (key_shift + key_bits - bucket_bits);
	// one chain-free radix pass on the top bits (clo_hip_radix4.hip)
	return clo_radix4_partition(src, dst, numel, elem_size, shift, bucket_bits,
		(unsigned long long*) counts_dev, workspace, s);
}

}  // extern "C"
