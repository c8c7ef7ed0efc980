// clo_hip_radixw.hip — LSD radix sort passes for wide digits (5..8 bits, radix
// 32..256): the "radix=32 … 256" options of satradix.
//
// Same structure per digit as the reference (sort/clo_sort_satradix.c:264-313)
// and as clo_hip_radix4.hip: per-tile digit histogram -> scan of the counters
// (digit-major, upstream's counters_sum) -> tile-local stable sort + scatter.
// Differences from the small-radix path:
//   * ranking: R counters per thread do not fit in registers, so elements are
//     ranked wave-wide — one __ballot per digit bit gives the lanes holding
//     the same digit (match-any), v_mbcnt the rank inside the wave item, and a
//     per-wave LDS counter row carries the count from item to item;
//   * the next digit's per-tile histogram cannot ride on the scatter (the
//     [digit][2][next digit] table would need R*2*R counters), so every pass
//     starts with its own histogram kernel: two reads + one write of every
//     element per digit (upstream: ~5 element streams + 6 counter streams).
// No kernel waits on another work-group.
#include <hip/hip_runtime.h>

#include "clo_hip.h"
#include "clo_hip_internal.h"

namespace {

constexpr int RW_THREADS = 512;
constexpr int RW_WAVES = RW_THREADS / 64;
constexpr int RW_CHUNK = 128;   // tiles per chunk of the counter scan

// 32 KiB LDS stage for 4- and 8-byte elements
template <typename E> struct rw_shape {
	static constexpr int ITEMS = sizeof(E) == 8 ? 8 : 16;
	static constexpr int TILE = RW_THREADS * ITEMS;
};

// ---------------------------------------------------------------------------
// per-tile digit histogram (upstream's satradix_histogram job)
// ---------------------------------------------------------------------------
template <typename E, int BITS>
__global__ __launch_bounds__(RW_THREADS)
void clo_radixw_tilehist_kernel(const E* __restrict__ in, size_t n, unsigned shift, unsigned mask,
	unsigned* __restrict__ thist, int aligned, clo_keyx kx) {
	constexpr int R = 1 << BITS;
	constexpr int ITEMS = rw_shape<E>::ITEMS;
	constexpr int TILE = rw_shape<E>::TILE;
	constexpr int VECS = ITEMS * (int) sizeof(E) / 16;   // 16-byte loads per thread
	constexpr int PER = 16 / (int) sizeof(E);
	static_assert(VECS >= 1, "a thread's slice is at least one 16-byte vector");
	__shared__ unsigned s_cnt[RW_WAVES][R];
	const unsigned tid = threadIdx.x, wave = tid >> 6;
	const size_t base = (size_t) blockIdx.x * TILE;
	const unsigned count = (n - base) < (size_t) TILE ? (unsigned) (n - base) : (unsigned) TILE;
	for (unsigned i = tid; i < RW_WAVES * R; i += RW_THREADS) (&s_cnt[0][0])[i] = 0;
	__syncthreads();
	const unsigned tbase = tid * ITEMS;
	if (count == (unsigned) TILE && aligned) {
		typedef E vecE __attribute__((ext_vector_type(PER)));
		const vecE* p = reinterpret_cast<const vecE*>(in + base + tbase);
		vecE v[VECS];
		#pragma unroll
		for (int k = 0; k < VECS; ++k) v[k] = p[k];
		#pragma unroll
		for (int k = 0; k < VECS; ++k) {
			#pragma unroll
			for (int q = 0; q < PER; ++q)
				atomicAdd(&s_cnt[wave][(unsigned) (clo_keyx_fwd<E>(v[k][q], kx) >> shift) & mask], 1u);
		}
	} else {
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i)
			if (tbase + i < count)
				atomicAdd(&s_cnt[wave][(unsigned) (clo_keyx_fwd<E>(in[base + tbase + i], kx) >> shift) & mask], 1u);
	}
	__syncthreads();
	for (unsigned d = tid; d < (unsigned) R; d += RW_THREADS) {
		unsigned h = 0;
		#pragma unroll
		for (int w = 0; w < RW_WAVES; ++w) h += s_cnt[w][d];
		thist[(size_t) blockIdx.x * R + d] = h;
	}
}

// ---------------------------------------------------------------------------
// counts[tile][digit] -> offsets[tile][digit] in digit-major order:
//   off[t][d] = sum_{d'<d} total[d'] + sum_{t'<t} cnt[t'][d]
// (exactly upstream's exclusive scan of counters[num_wgs*d + wg]). A row of R
// counters is contiguous, thread = digit: every access is coalesced. Two
// kernels over chunks of RW_CHUNK tiles; G = 256 / R thread groups share a
// chunk when R < 256.
// ---------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256)
void clo_radixw_chunksum_kernel(const unsigned* __restrict__ thist, unsigned tiles, unsigned* __restrict__ partial) {
	constexpr int G = 256 / R;
	__shared__ unsigned s_p[G][R];
	const unsigned tid = threadIdx.x, d = tid % R, g = tid / R;
	const unsigned t0 = blockIdx.x * RW_CHUNK;
	const unsigned t1 = t0 + RW_CHUNK < tiles ? t0 + RW_CHUNK : tiles;
	unsigned sum = 0;
	#pragma unroll 8
	for (unsigned t = t0 + g; t < t1; t += G) sum += thist[(size_t) t * R + d];
	s_p[g][d] = sum;
	__syncthreads();
	if (tid < (unsigned) R) {
		unsigned tot = 0;
		#pragma unroll
		for (int k = 0; k < G; ++k) tot += s_p[k][tid];
		partial[(size_t) blockIdx.x * R + tid] = tot;
	}
}

template <int R>
__global__ __launch_bounds__(256)
void clo_radixw_offsets_kernel(const unsigned* __restrict__ thist, unsigned tiles,
	const unsigned* __restrict__ partial, unsigned chunks, unsigned* __restrict__ toff) {
	constexpr int G = 256 / R;
	constexpr int SUB = RW_CHUNK / G;   // tiles per thread group
	__shared__ unsigned s_a[G][R], s_b[G][R], s_base[R], s_w[4];
	const unsigned tid = threadIdx.x, d = tid % R, g = tid / R, lane = tid & 63u, wave = tid >> 6;

	// 1. digit d: count in earlier chunks and in all chunks (every block
	// re-derives its starting point from the chunk sums: no chain)
	{
		unsigned before = 0, total = 0;
		for (unsigned c = g; c < chunks; c += G) {
			const unsigned v = partial[(size_t) c * R + d];
			total += v;
			if (c < blockIdx.x) before += v;
		}
		s_a[g][d] = before;
		s_b[g][d] = total;
	}
	__syncthreads();
	unsigned bef = 0, tot = 0;
	if (tid < (unsigned) R) {
		#pragma unroll
		for (int k = 0; k < G; ++k) { bef += s_a[k][tid]; tot += s_b[k][tid]; }
	}
	// 2. exclusive scan of the digit totals over the digits
	const unsigned incl = clo_wave_scan_inclusive<unsigned>(tot, lane);
	if (lane == 63) s_w[wave] = incl;
	__syncthreads();
	if (tid < (unsigned) R) {
		unsigned add = 0;
		#pragma unroll
		for (unsigned w = 0; w < 4; ++w) if (w < wave) add += s_w[w];
		s_base[tid] = incl - tot + add + bef;
	}
	__syncthreads();

	// 3. walk the chunk: group g takes SUB consecutive tiles
	const unsigned t0 = blockIdx.x * RW_CHUNK;
	const unsigned tend = t0 + RW_CHUNK < tiles ? t0 + RW_CHUNK : tiles;
	const unsigned ts = t0 + g * SUB < tend ? t0 + g * SUB : tend;
	const unsigned te = ts + SUB < tend ? ts + SUB : tend;
	unsigned run = s_base[d];
	if (G > 1) {
		unsigned own = 0;
		#pragma unroll 8
		for (unsigned t = ts; t < te; ++t) own += thist[(size_t) t * R + d];
		s_a[g][d] = own;
		__syncthreads();
		for (unsigned k = 0; k < g; ++k) run += s_a[k][d];
	}
	#pragma unroll 8
	for (unsigned t = ts; t < te; ++t) {
		const unsigned c = thist[(size_t) t * R + d];
		toff[(size_t) t * R + d] = run;
		run += c;
	}
}

// ---------------------------------------------------------------------------
// one digit: tile-local stable sort + scatter
// ---------------------------------------------------------------------------
template <typename E, int BITS>
__global__ __launch_bounds__(RW_THREADS)
void clo_radixw_pass_kernel(const E* __restrict__ in, E* __restrict__ out, size_t n,
	unsigned shift, unsigned mask, const unsigned* __restrict__ toff, clo_keyx kx_in, clo_keyx kx_out) {

	constexpr int R = 1 << BITS;
	constexpr int ITEMS = rw_shape<E>::ITEMS;
	constexpr int TILE = rw_shape<E>::TILE;
	static_assert(R <= RW_THREADS, "one thread per digit");

	__shared__ E s_stage[TILE];
	__shared__ unsigned s_wcnt[RW_WAVES][R];   // per-wave digit counts, then running tile-local position of (wave, digit)
	__shared__ unsigned s_delta[R];            // global index = tile-local position + delta[digit]
	__shared__ unsigned s_w[RW_WAVES];

	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const unsigned tile = blockIdx.x;
	const size_t base = (size_t) tile * TILE;
	const unsigned count = (n - base) < (size_t) TILE ? (unsigned) (n - base) : (unsigned) TILE;
	const bool full = count == (unsigned) TILE;
	const unsigned wbase = wave * 64u * ITEMS + lane;
	const unsigned n32 = n > 0xffffffffull ? 0xffffffffu : (unsigned) n;   // global indices are 32-bit here

	// the tile's global offsets (upstream's counters_sum), requested before the keys
	unsigned goff = 0;
	if (tid < (unsigned) R) goff = toff[(size_t) tile * R + tid];
	for (unsigned i = tid; i < RW_WAVES * R; i += RW_THREADS) (&s_wcnt[0][0])[i] = 0;

	// ---- 1. load, wave-striped: lane l of wave w holds tile element w*64*ITEMS + i*64 + l ----
	E key[ITEMS];
	#pragma unroll
	for (int i = 0; i < ITEMS; ++i)
		key[i] = (full || wbase + i * 64 < count) ? in[base + wbase + i * 64] : (E) 0;
	if (kx_in.kind) {   // first pass of a sort on signed / floating-point keys
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) key[i] = clo_keyx_fwd<E>(key[i], kx_in);
	}
	__syncthreads();

	// ---- 2a. match: per item, the lanes of my wave holding my digit ----
	// One ballot per digit bit gives the group (match-any); v_mbcnt my rank in
	// it. ONE lane per distinct digit (the group's first) adds the group size
	// to the wave's digit count: distinct LDS addresses within the instruction.
	// (rank, size, leader lane) stay packed in a VGPR.
	unsigned grp[ITEMS];
	#pragma unroll
	for (int i = 0; i < ITEMS; ++i) {
		const bool valid = full || (wbase + i * 64 < count);
		const unsigned d = (unsigned) (key[i] >> shift) & mask;
		clo_u64 peers = full ? ~0ull : __ballot(valid);
		#pragma unroll
		for (int k = 0; k < BITS; ++k) {
			const bool bit = (d >> k) & 1u;
			const clo_u64 b = __ballot(bit);
			peers &= bit ? b : ~b;
		}
		const unsigned r = clo_mbcnt(peers);
		const unsigned c = (unsigned) __popcll(peers);
		const unsigned leader = valid ? (unsigned) (__ffsll((long long) peers) - 1) : lane;
		if (valid && r == 0) atomicAdd(&s_wcnt[wave][d], c);
		grp[i] = r | (c << 8) | (leader << 16);
	}
	__syncthreads();

	// ---- tile histogram -> start of every (wave, digit) run ----
	unsigned hist = 0, cw[RW_WAVES];
	if (tid < (unsigned) R) {
		#pragma unroll
		for (int w = 0; w < RW_WAVES; ++w) { cw[w] = s_wcnt[w][tid]; hist += cw[w]; }
	}
	const unsigned incl = clo_wave_scan_inclusive<unsigned>(hist, lane);
	if (lane == 63) s_w[wave] = incl;
	__syncthreads();
	if (tid < (unsigned) R) {
		unsigned dstart = incl - hist;
		#pragma unroll
		for (unsigned w = 0; w < (unsigned) RW_WAVES; ++w) if (w < wave) dstart += s_w[w];
		s_delta[tid] = goff - dstart;
		unsigned run = dstart;
		#pragma unroll
		for (int w = 0; w < RW_WAVES; ++w) { s_wcnt[w][tid] = run; run += cw[w]; }
	}
	__syncthreads();

	// ---- 2b. rank: the group's first lane takes the group's slice of the
	// (wave, digit) run with a returning LDS atomic (a wave's LDS atomics run in
	// issue order, so slices follow item order = stable); ds_bpermute hands the
	// slice start to the group. 4a. scatter into the LDS stage. ----
	#pragma unroll
	for (int i = 0; i < ITEMS; ++i) {
		const bool valid = full || (wbase + i * 64 < count);
		const unsigned d = (unsigned) (key[i] >> shift) & mask;
		const unsigned r = grp[i] & 0xffu, c = (grp[i] >> 8) & 0xffu, leader = grp[i] >> 16;
		unsigned start = 0;
		if (valid && r == 0) start = atomicAdd(&s_wcnt[wave][d], c);
		start = (unsigned) __shfl((int) start, (int) leader, 64);
		if (valid) s_stage[(start + r) & (TILE - 1)] = key[i];
	}
	__syncthreads();

	// ---- 4b. contiguous runs to HBM (bounded even if the counters were garbage) ----
	#pragma unroll
	for (int j = 0; j < ITEMS; ++j) {
		const unsigned p = j * RW_THREADS + tid;
		if (full || p < count) {
			const E e = s_stage[p];
			const unsigned d = (unsigned) (e >> shift) & mask;
			const unsigned gi = p + s_delta[d];
			if (gi < n32) out[gi] = clo_keyx_inv<E>(e, kx_out);
		}
	}
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct rw_layout { size_t thist, toff, partial, total, tiles, chunks; };

rw_layout rw_make_layout(size_t n, int elem_size, int digit_bits) {
	rw_layout L;
	const size_t R = (size_t) 1 << digit_bits;
	const size_t tile = (size_t) RW_THREADS * (elem_size == 8 ? 8 : 16);
	L.tiles = (n + tile - 1) / tile;
	if (L.tiles == 0) L.tiles = 1;
	L.chunks = (L.tiles + RW_CHUNK - 1) / RW_CHUNK;
	const size_t per = L.tiles * R * sizeof(unsigned);
	L.thist = CLO_WS_HEADER_BYTES;
	L.toff = L.thist + per;
	L.partial = L.toff + per;
	L.total = L.partial + ((L.chunks * R * sizeof(unsigned) + 255) & ~(size_t) 255);
	return L;
}

template <typename E, int BITS>
int rw_sort_impl(const E* src, E* dst, E* tmp, size_t n, int key_shift, int key_bits, clo_keyx kx, void* ws, hipStream_t s) {
	constexpr unsigned R = 1u << BITS;
	const int passes = (key_bits + BITS - 1) / BITS;
	const rw_layout L = rw_make_layout(n, (int) sizeof(E), BITS);
	unsigned* thist = (unsigned*) ((char*) ws + L.thist);
	unsigned* toff = (unsigned*) ((char*) ws + L.toff);
	unsigned* partial = (unsigned*) ((char*) ws + L.partial);
	const unsigned tiles = (unsigned) L.tiles, chunks = (unsigned) L.chunks;
	const clo_keyx kx_none = { 0, 0, 0 };

	hipError_t e = hipMemsetAsync(ws, 0, CLO_WS_HEADER_BYTES, s);   // status word
	if (e != hipSuccess) return (int) e;

	const bool inplace_odd = (dst == src) && (passes % 2 == 1);
	const E* cur_in = src;
	for (int p = 0; p < passes; ++p) {
		E* cur_out;
		if (inplace_odd) cur_out = (p % 2 == 0) ? tmp : dst;
		else cur_out = ((passes - 1 - p) % 2 == 0) ? dst : tmp;
		const int rem = key_bits - p * BITS;
		const unsigned bits = rem < BITS ? rem : BITS;
		const unsigned shift = (unsigned) (key_shift + p * BITS), mask = (1u << bits) - 1u;
		{
			clo_timing_scope timing("radix_hist", s);
			hipLaunchKernelGGL((clo_radixw_tilehist_kernel<E, BITS>), dim3(tiles), dim3(RW_THREADS), 0, s,
				cur_in, n, shift, mask, thist, (int) ((uintptr_t) cur_in % 16 == 0), p == 0 ? kx : kx_none);
		}
		{
			clo_timing_scope timing("radix_offsets", s);
			hipLaunchKernelGGL((clo_radixw_chunksum_kernel<R>), dim3(chunks), dim3(256), 0, s,
				(const unsigned*) thist, tiles, partial);
			hipLaunchKernelGGL((clo_radixw_offsets_kernel<R>), dim3(chunks), dim3(256), 0, s,
				(const unsigned*) thist, tiles, (const unsigned*) partial, chunks, toff);
		}
		{
			clo_timing_scope timing("radix_pass", s);
			hipLaunchKernelGGL((clo_radixw_pass_kernel<E, BITS>), dim3(tiles), dim3(RW_THREADS), 0, s,
				cur_in, cur_out, n, shift, mask, (const unsigned*) toff,
				p == 0 ? kx : kx_none, p + 1 == passes ? kx : kx_none);
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
int rw_dispatch(const void* src, void* dst, void* tmp, size_t n, int key_shift, int key_bits, int digit_bits,
	clo_keyx kx, void* ws, hipStream_t s) {
	switch (digit_bits) {
		case 5: return rw_sort_impl<E, 5>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, kx, ws, s);
		case 6: return rw_sort_impl<E, 6>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, kx, ws, s);
		case 7: return rw_sort_impl<E, 7>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, kx, ws, s);
		case 8: return rw_sort_impl<E, 8>((const E*) src, (E*) dst, (E*) tmp, n, key_shift, key_bits, kx, ws, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

}  // namespace

size_t clo_radixw_workspace_bytes(size_t n, int elem_size, int digit_bits) {
	return rw_make_layout(n, elem_size, digit_bits).total;
}

size_t clo_radixw_lds_bytes(const char* kernel, int elem_size, int digit_bits) {
	const size_t R = (size_t) 1 << digit_bits;
	if (kernel[0] == 'h') return RW_WAVES * R * sizeof(unsigned);
	const size_t tile = (size_t) RW_THREADS * (elem_size == 8 ? 8 : 16);
	return tile * (size_t) elem_size + (RW_WAVES * R + R + RW_WAVES) * sizeof(unsigned);
}

int clo_radixw_sort(const void* src, void* dst, void* tmp, size_t n, int elem_size, int key_shift,
	int key_bits, int digit_bits, clo_keyx kx, void* ws, hipStream_t s) {
	switch (elem_size) {
		case 1: return rw_dispatch<uint8_t>(src, dst, tmp, n, key_shift, key_bits, digit_bits, kx, ws, s);
		case 2: return rw_dispatch<uint16_t>(src, dst, tmp, n, key_shift, key_bits, digit_bits, kx, ws, s);
		case 4: return rw_dispatch<uint32_t>(src, dst, tmp, n, key_shift, key_bits, digit_bits, kx, ws, s);
		case 8: return rw_dispatch<uint64_t>(src, dst, tmp, n, key_shift, key_bits, digit_bits, kx, ws, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}
