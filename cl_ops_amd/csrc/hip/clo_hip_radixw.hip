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

#include "clo_hip.h"
#include "clo_hip_internal.h"
#include "clo_hip_radix_rank.h"

namespace {

constexpr int RW_CHUNK = 128;   // tiles per chunk of the counter scan

// the pair kernel's tile (clo_hip_radix_rank.h), one thread per 16 elements (8 of 8 bytes)
template <typename E, bool BIG> struct rw_shape {
	static constexpr int TILE = pair_shape<E, BIG>::TILE;
	static constexpr int THREADS = pair_shape<E, BIG>::THREADS;
	static constexpr int ITEMS = TILE / THREADS;
};

// One word per tile beside its histogram row: 1 when ONE bin holds the whole tile (every element of the
// tile carries the same digit: small keys, a shared prefix, equal keys). The pass kernel reads it with a
// scalar load and sends such a tile past its two local splits (clo_hip_radix4.hip: the split of a
// single-digit tile is its worst case). Decided here because the counts are here: in the pass kernel the
// row sits with 256 threads and asking them costs a barrier per tile (+3 % on uniform keys, measured).
// No thread needs to know more than its own bin: a bin that holds the whole tile says 1, a bin that holds a
// part of it says 0 (several may, all the same value), an empty bin says nothing — exactly one of the first
// two kinds exists in every tile, so the word is always written and never needs clearing or a barrier.
__device__ __forceinline__ void rw_tile_info(unsigned h, unsigned count, unsigned* __restrict__ tinfo) {
	if (h == count) tinfo[blockIdx.x] = 1u;
	else if (h != 0u) tinfo[blockIdx.x] = 0u;
}

// ---------------------------------------------------------------------------
// per-tile digit histogram (upstream's satradix_histogram job)
// ---------------------------------------------------------------------------
template <typename E, int BITS, bool BIG>
__global__ __launch_bounds__((rw_shape<E, BIG>::THREADS))
void clo_radixw_tilehist_kernel(const E* __restrict__ in, size_t n, unsigned shift, unsigned mask,
	unsigned* __restrict__ thist, unsigned* __restrict__ tinfo, int aligned, clo_keyx kx) {
	constexpr int R = 1 << BITS;
	constexpr int ITEMS = rw_shape<E, BIG>::ITEMS;
	constexpr int TILE = rw_shape<E, BIG>::TILE;
	constexpr int RW_THREADS = rw_shape<E, BIG>::THREADS;
	constexpr int VB = ITEMS * (int) sizeof(E) >= 16 ? 16 : ITEMS * (int) sizeof(E);   // bytes per vector load
	constexpr int VECS = ITEMS * (int) sizeof(E) / VB;
	constexpr int PER = VB / (int) sizeof(E);
	// 32 copies of every counter, copy = lane mod 32, bin-major: the 32 lanes an LDS
	// instruction serves together hit 32 different banks and never one address (an
	// LDS add holds its bank for many cycles; one copy per wave, lanes colliding on
	// banks, made the adds a co-bottleneck of this otherwise streaming kernel).
	constexpr int COPIES = 32;
	__shared__ __attribute__((aligned(16))) unsigned s_cnt[R * COPIES];
	const unsigned tid = threadIdx.x, lane = tid & 63u;
	const size_t base = (size_t) blockIdx.x * TILE;
	const unsigned count = (n - base) < (size_t) TILE ? (unsigned) (n - base) : (unsigned) TILE;
	{   // (16-byte stores: a quarter of the LDS instructions of a dword loop)
		typedef unsigned vec4u __attribute__((ext_vector_type(4)));
		const vec4u z = { 0u, 0u, 0u, 0u };
		for (unsigned i = tid; i < (unsigned) (R * COPIES / 4); i += RW_THREADS) reinterpret_cast<vec4u*>(s_cnt)[i] = z;
	}
	__syncthreads();
	const unsigned tbase = tid * ITEMS;
	const unsigned cp = lane & (COPIES - 1);
	if (count == (unsigned) TILE && aligned) {
		typedef E vecE __attribute__((ext_vector_type(PER)));   // (`aligned`: the source is 16-byte aligned)
		const vecE* p = reinterpret_cast<const vecE*>(in + base + tbase);
		vecE v[VECS];
		#pragma unroll
		for (int k = 0; k < VECS; ++k) v[k] = p[k];
		#pragma unroll
		for (int k = 0; k < VECS; ++k) {
			#pragma unroll
			for (int q = 0; q < PER; ++q)
				atomicAdd(&s_cnt[(((unsigned) (clo_keyx_fwd<E>(v[k][q], kx) >> shift) & mask) << 5) + cp], 1u);
		}
	} else {
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i)
			if (tbase + i < count)
				atomicAdd(&s_cnt[(((unsigned) (clo_keyx_fwd<E>(in[base + tbase + i], kx) >> shift) & mask) << 5) + cp], 1u);
	}
	__syncthreads();
	for (unsigned d = tid; d < (unsigned) R; d += RW_THREADS) {
		typedef unsigned vec4u __attribute__((ext_vector_type(4)));
		const vec4u* row = reinterpret_cast<const vec4u*>(&s_cnt[d * COPIES]);
		unsigned h = 0;
		#pragma unroll
		for (int k = 0; k < COPIES / 4; ++k) {   // (rotated: the lanes' rows are 128 bytes apart)
			const vec4u x = row[(k + d) & (COPIES / 4 - 1)];
			h += x[0] + x[1] + x[2] + x[3];
		}
		thist[(size_t) blockIdx.x * R + d] = h;
		rw_tile_info(h, count, tinfo);
	}
}

// The same histogram out of the digit bytes the pass before wrote (one byte per
// element, in the order of the elements; clo_radix4_pair_kernel<..., DIG>): a quarter
// (uint32) or an eighth (8-byte elements) of the bytes to read.
template <int BITS, int ITEMS, int THREADS>   // ITEMS bytes per thread: 16 (4-byte elements) or 8; THREADS of the tile's shape
__global__ __launch_bounds__(THREADS)
void clo_radixw_tilehist_bytes_kernel(const unsigned char* __restrict__ dig, size_t n, unsigned mask, unsigned* __restrict__ thist,
	unsigned* __restrict__ tinfo) {
	constexpr int R = 1 << BITS;
	constexpr int TILE = THREADS * ITEMS;
	constexpr int COPIES = 32;
	__shared__ __attribute__((aligned(16))) unsigned s_cnt[R * COPIES];
	const unsigned tid = threadIdx.x, lane = tid & 63u;
	const size_t base = (size_t) blockIdx.x * TILE;
	const unsigned count = (n - base) < (size_t) TILE ? (unsigned) (n - base) : (unsigned) TILE;
	{
		typedef unsigned vec4u __attribute__((ext_vector_type(4)));
		const vec4u z = { 0u, 0u, 0u, 0u };
		for (unsigned i = tid; i < (unsigned) (R * COPIES / 4); i += THREADS) reinterpret_cast<vec4u*>(s_cnt)[i] = z;
	}
	__syncthreads();
	const unsigned tbase = tid * ITEMS;
	const unsigned cp = lane & (COPIES - 1);
	if (count == (unsigned) TILE) {
		typedef unsigned vecw __attribute__((ext_vector_type(ITEMS / 4)));   // (the stream starts 256-byte aligned, a tile is a multiple of 16 bytes)
		const vecw v = *reinterpret_cast<const vecw*>(dig + base + tbase);
		#pragma unroll
		for (int k = 0; k < ITEMS / 4; ++k) {
			#pragma unroll
			for (int b = 0; b < 4; ++b)
				atomicAdd(&s_cnt[((((unsigned) v[k] >> (8 * b)) & mask) << 5) + cp], 1u);
		}
	} else {
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i)
			if (tbase + i < count) atomicAdd(&s_cnt[(((unsigned) dig[base + tbase + i] & mask) << 5) + cp], 1u);
	}
	__syncthreads();
	for (unsigned d = tid; d < (unsigned) R; d += THREADS) {
		typedef unsigned vec4u __attribute__((ext_vector_type(4)));
		const vec4u* row = reinterpret_cast<const vec4u*>(&s_cnt[d * COPIES]);
		unsigned h = 0;
		#pragma unroll
		for (int k = 0; k < COPIES / 4; ++k) {
			const vec4u x = row[(k + d) & (COPIES / 4 - 1)];
			h += x[0] + x[1] + x[2] + x[3];
		}
		thist[(size_t) blockIdx.x * R + d] = h;
		rw_tile_info(h, count, tinfo);
	}
}

// ---------------------------------------------------------------------------
// counts[tile][digit] -> offsets[tile][digit] in digit-major order:
//   off[t][d] = sum_{d'<d} total[d'] + sum_{t'<t} cnt[t'][d]
// (exactly upstream's exclusive scan of counters[num_wgs*d + wg]). A row of R
// counters is contiguous, thread = digit: every access is coalesced. Three
// small kernels: sums of chunks of RW_CHUNK tiles, a one-work-group scan of the
// chunk sums, the walk over each chunk's tiles.
//
// These kernels move a few MB and are bound by LATENCY: a thread that walks its
// rows one after the other pays a round trip per row (round 2: 256 threads, one
// thread per digit walking 128 rows, 8 loads in flight: 16-25 us for the three
// launches, a quarter of a 2^24-key sort). Round 3: 1024 threads = G groups of R
// threads, group g owns SUB = RW_CHUNK / G consecutive rows of the chunk and requests
// ALL of them at once (SUB registers, fully unrolled); the groups meet in LDS. One
// round trip per kernel instead of SUB: 15 / 19 us per pass at 2^24 / 2^28 keys. What is
// left is three launches' worth of launch + one round trip each. Tried and dropped
// (profiles/r03_hist_chain_probe.txt): the first two steps chained onto the
// histogram kernel (write-through rows, arrival counters, the last arrival sums: the
// histogram kernel, bound by its loads in flight, lost 0.07 ms per pass to the arrivals);
// the chunk scan folded into the offsets kernel, every work-group re-deriving its chunk's
// start from all chunk sums (25.7 instead of 19.3 us at 2^28, 19.3 instead of 15.1 at 2^24:
// 128 KiB of L2 reads per work-group cost more than the launch they replace).
// ---------------------------------------------------------------------------
constexpr int RW_CS_THREADS = 1024;
template <int R> struct rw_cs {
	static constexpr int G = (RW_CS_THREADS / R) < RW_CHUNK ? (RW_CS_THREADS / R) : RW_CHUNK;   // thread groups per chunk
	static constexpr int SUB = RW_CHUNK / G;                                                      // rows per group
	static_assert(G * SUB == RW_CHUNK && G * R <= RW_CS_THREADS, "the groups tile the chunk");
};

template <int R>
__global__ __launch_bounds__(RW_CS_THREADS)
void clo_radixw_chunksum_kernel(const unsigned* __restrict__ thist, unsigned tiles, unsigned* __restrict__ partial) {
	constexpr int G = rw_cs<R>::G, SUB = rw_cs<R>::SUB;
	__shared__ unsigned s_p[G * R];
	const unsigned tid = threadIdx.x, d = tid % R, g = tid / R;
	const unsigned t0 = blockIdx.x * RW_CHUNK;
	const unsigned t1 = t0 + RW_CHUNK < tiles ? t0 + RW_CHUNK : tiles;
	if (g < (unsigned) G) {
		unsigned v[SUB];
		#pragma unroll
		for (int k = 0; k < SUB; ++k) {
			const unsigned t = t0 + g * SUB + k;
			v[k] = t < t1 ? thist[(size_t) t * R + d] : 0u;
		}
		unsigned sum = 0;
		#pragma unroll
		for (int k = 0; k < SUB; ++k) sum += v[k];
		s_p[g * R + d] = sum;
	}
	__syncthreads();
	if (tid < (unsigned) R) {
		unsigned tot = 0;
		#pragma unroll 8
		for (int k = 0; k < G; ++k) tot += s_p[k * R + tid];
		partial[(size_t) blockIdx.x * R + tid] = tot;
	}
}

// Chunk sums -> for every (chunk, digit) the offset of the chunk's first tile:
// digit base (exclusive scan of the digit totals over the digits) + count of
// the digit in earlier chunks. ONE work-group: thread (g, d) owns a contiguous
// range of chunks — all of them in registers at once when there are at most
// RW_CS_REGS per thread (up to 128 chunks = 16 384 tiles with R = 256), else walked
// twice —, the groups are combined through LDS. In place: partial[c][d] becomes that offset.
constexpr int RW_CS_REGS = 32;
template <int R>
__global__ __launch_bounds__(RW_CS_THREADS)
void clo_radixw_chunkscan_kernel(unsigned* __restrict__ partial, unsigned chunks) {
	constexpr int G = RW_CS_THREADS / R;   // thread groups: each owns chunks / G chunks
	__shared__ unsigned s_g[G * R], s_w[4];
	const unsigned tid = threadIdx.x, d = tid % R, g = tid / R, lane = tid & 63u, wave = tid >> 6;
	const unsigned per = (chunks + G - 1) / G;
	const unsigned c0 = g * per < chunks ? g * per : chunks, c1 = c0 + per < chunks ? c0 + per : chunks;
	const bool in_regs = per <= (unsigned) RW_CS_REGS;   // (the same for every thread)
	unsigned v[RW_CS_REGS];
	unsigned sum = 0;
	if (in_regs) {
		#pragma unroll
		for (int k = 0; k < RW_CS_REGS; ++k) v[k] = c0 + k < c1 ? partial[(size_t) (c0 + k) * R + d] : 0u;
		#pragma unroll
		for (int k = 0; k < RW_CS_REGS; ++k) sum += v[k];
	} else {
		#pragma unroll 8
		for (unsigned c = c0; c < c1; ++c) sum += partial[(size_t) c * R + d];
	}
	s_g[g * R + d] = sum;
	__syncthreads();
	unsigned before = 0, tot = 0;
	#pragma unroll 4
	for (int k = 0; k < G; ++k) {
		const unsigned x = s_g[k * R + d];
		if ((unsigned) k < g) before += x;
		tot += x;
	}
	// exclusive scan of the digit totals over the digits (threads 0..R-1 carry them)
	const unsigned t = tid < (unsigned) R ? tot : 0u;
	const unsigned incl = clo_wave_scan_inclusive<unsigned>(t, lane);
	if (lane == 63 && wave < 4) s_w[wave] = incl;   // R <= 256: the digits sit in the first four waves
	__syncthreads();
	unsigned dbase = incl - t;
	#pragma unroll
	for (unsigned w = 0; w < 4; ++w) if (w < wave) dbase += s_w[w];
	__syncthreads();
	if (tid < (unsigned) R) s_g[tid] = dbase;   // (row 0 of s_g is free again: every thread has read it)
	__syncthreads();
	unsigned run = s_g[d] + before;
	if (in_regs) {
		#pragma unroll
		for (int k = 0; k < RW_CS_REGS; ++k) {
			if (c0 + k < c1) partial[(size_t) (c0 + k) * R + d] = run;
			run += v[k];
		}
	} else {
		#pragma unroll 8
		for (unsigned c = c0; c < c1; ++c) {
			const unsigned x = partial[(size_t) c * R + d];
			partial[(size_t) c * R + d] = run;
			run += x;
		}
	}
}

// Offsets of the tiles of one chunk: thread (g, d) owns SUB consecutive tiles, all requested at once.
template <int R>
__global__ __launch_bounds__(RW_CS_THREADS)
void clo_radixw_offsets_kernel(const unsigned* __restrict__ thist, unsigned tiles,
	const unsigned* __restrict__ cbase, unsigned* __restrict__ toff) {
	constexpr int G = rw_cs<R>::G, SUB = rw_cs<R>::SUB;
	__shared__ unsigned s_a[G * R];
	const unsigned tid = threadIdx.x, d = tid % R, g = tid / R;
	const unsigned t0 = blockIdx.x * RW_CHUNK;
	const unsigned tend = t0 + RW_CHUNK < tiles ? t0 + RW_CHUNK : tiles;
	const bool active = g < (unsigned) G;
	unsigned v[SUB];
	unsigned run = 0;
	if (active) {
		run = cbase[(size_t) blockIdx.x * R + d];
		#pragma unroll
		for (int k = 0; k < SUB; ++k) {
			const unsigned t = t0 + g * SUB + k;
			v[k] = t < tend ? thist[(size_t) t * R + d] : 0u;
		}
		unsigned own = 0;
		#pragma unroll
		for (int k = 0; k < SUB; ++k) own += v[k];
		s_a[g * R + d] = own;
	}
	__syncthreads();
	if (!active) return;
	for (unsigned k = 0; k < g; ++k) run += s_a[k * R + d];
	#pragma unroll
	for (int k = 0; k < SUB; ++k) {
		const unsigned t = t0 + g * SUB + k;
		if (t < tend) toff[(size_t) t * R + d] = run;
		run += v[k];
	}
}

// Up to RW_CHUNK tiles (one chunk): the three steps above in one launch of one
// work-group — arrays of 2^13 .. 2^20 elements are launch-bound.
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

}  // namespace

// ---- the histogram / counter-scan steps for any digit width 1..8 (used by
// the digit-pair passes of clo_hip_radix4.hip) ----

template <typename E>
static int rw_launch_tilehist(const void* in, size_t n, int bits, unsigned shift, unsigned mask, unsigned* thist, unsigned* tinfo,
	unsigned tiles, bool big, clo_keyx kx, hipStream_t s) {
	const int aligned = (int) ((uintptr_t) in % 16 == 0);
	#define CLO_RW_TH(B) case B: \
		if (big && sizeof(E) >= 4) hipLaunchKernelGGL((clo_radixw_tilehist_kernel<E, B, (sizeof(E) >= 4)>), dim3(tiles), dim3(rw_shape<E, (sizeof(E) >= 4)>::THREADS), 0, s, \
			(const E*) in, n, shift, mask, thist, tinfo, aligned, kx); \
		else hipLaunchKernelGGL((clo_radixw_tilehist_kernel<E, B, false>), dim3(tiles), dim3(rw_shape<E, false>::THREADS), 0, s, \
			(const E*) in, n, shift, mask, thist, tinfo, aligned, kx); \
		break
	switch (bits) {
		CLO_RW_TH(1); CLO_RW_TH(2); CLO_RW_TH(3); CLO_RW_TH(4); CLO_RW_TH(5); CLO_RW_TH(6); CLO_RW_TH(7); CLO_RW_TH(8);
		default: return CLO_HIP_EUNSUPPORTED;
	}
	#undef CLO_RW_TH
	return (int) hipGetLastError();
}

// `big`: the tiles are those of clo_radix_big_tiles(n, elem_size) (clo_hip_radix_rank.h)
int clo_radixw_launch_tilehist(const void* in, size_t n, int elem_size, int bits, unsigned shift, unsigned mask,
	unsigned* thist, unsigned* tinfo, unsigned tiles, bool big, clo_keyx kx, hipStream_t s) {
	switch (elem_size) {
		case 1: return rw_launch_tilehist<uint8_t>(in, n, bits, shift, mask, thist, tinfo, tiles, big, kx, s);
		case 2: return rw_launch_tilehist<uint16_t>(in, n, bits, shift, mask, thist, tinfo, tiles, big, kx, s);
		case 4: return rw_launch_tilehist<uint32_t>(in, n, bits, shift, mask, thist, tinfo, tiles, big, kx, s);
		case 8: return rw_launch_tilehist<uint64_t>(in, n, bits, shift, mask, thist, tinfo, tiles, big, kx, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

// Histograms out of the digit stream (tiles of the shape `big` names).
int clo_radixw_launch_tilehist_bytes(const unsigned char* dig, size_t n, int elem_size, int bits, unsigned mask,
	unsigned* thist, unsigned* tinfo, unsigned tiles, bool big, hipStream_t s) {
	#define CLO_RW_THB1(B, I, T) hipLaunchKernelGGL((clo_radixw_tilehist_bytes_kernel<B, I, T>), dim3(tiles), dim3(T), 0, s, dig, n, mask, thist, tinfo)
	#define CLO_RW_THB(B) case B: \
		if (!big) return CLO_HIP_EUNSUPPORTED;   /* (the stream goes with the big tiles) */ \
		if (elem_size == 8) CLO_RW_THB1(B, 8, 1024); else CLO_RW_THB1(B, 16, 1024); \
		break
	if (elem_size != 4 && elem_size != 8) return CLO_HIP_EUNSUPPORTED;
	switch (bits) {
		CLO_RW_THB(1); CLO_RW_THB(2); CLO_RW_THB(3); CLO_RW_THB(4); CLO_RW_THB(5); CLO_RW_THB(6); CLO_RW_THB(7); CLO_RW_THB(8);
		default: return CLO_HIP_EUNSUPPORTED;
	}
	#undef CLO_RW_THB
	#undef CLO_RW_THB1
	return (int) hipGetLastError();
}

int clo_radixw_launch_offsets(int bits, const unsigned* thist, unsigned tiles, unsigned* partial, unsigned* toff, hipStream_t s) {
	const unsigned chunks = (tiles + RW_CHUNK - 1) / RW_CHUNK;
	#define CLO_RW_OFF1(B) case B: hipLaunchKernelGGL((clo_radixw_offsets1_kernel<(1 << B)>), dim3(1), dim3(256), 0, s, thist, tiles, toff); break
	if (chunks == 1) {
		switch (bits) {
			CLO_RW_OFF1(1); CLO_RW_OFF1(2); CLO_RW_OFF1(3); CLO_RW_OFF1(4); CLO_RW_OFF1(5); CLO_RW_OFF1(6); CLO_RW_OFF1(7); CLO_RW_OFF1(8);
			default: return CLO_HIP_EUNSUPPORTED;
		}
		return (int) hipGetLastError();
	}
	#undef CLO_RW_OFF1
	#define CLO_RW_OFF(B) case B: \
		hipLaunchKernelGGL((clo_radixw_chunksum_kernel<(1 << B)>), dim3(chunks), dim3(RW_CS_THREADS), 0, s, thist, tiles, partial); \
		hipLaunchKernelGGL((clo_radixw_chunkscan_kernel<(1 << B)>), dim3(1), dim3(RW_CS_THREADS), 0, s, partial, chunks); \
		hipLaunchKernelGGL((clo_radixw_offsets_kernel<(1 << B)>), dim3(chunks), dim3(RW_CS_THREADS), 0, s, thist, tiles, (const unsigned*) partial, toff); break
	switch (bits) {
		CLO_RW_OFF(1); CLO_RW_OFF(2); CLO_RW_OFF(3); CLO_RW_OFF(4); CLO_RW_OFF(5); CLO_RW_OFF(6); CLO_RW_OFF(7); CLO_RW_OFF(8);
		default: return CLO_HIP_EUNSUPPORTED;
	}
	#undef CLO_RW_OFF
	return (int) hipGetLastError();
}

size_t clo_radixw_lds_bytes(int digit_bits) { return 32 * ((size_t) 1 << digit_bits) * sizeof(unsigned); }
