// Developer probe (not part of the product): what the LAST stage of an "MSD, MSD, then sort every sub-bucket on-chip"
// radix sort would cost: 2^28 uint32 keys already partitioned by their top 16 bits into 65 536
// sub-buckets of about 4 096 keys; one work-group per sub-bucket loads it, sorts it by the low 16 bits with four
// local splits of 4 bits (the product's own split, clo_hip_radix_rank.h), and stores it back in place.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Icl_ops_amd/csrc -Icl_ops_amd/csrc/hip tools/local_sort_probe.hip -o build_probe/local_sort_probe
//   ./local_sort_probe [log2 n = 28] [top bits = 16] [key bytes = 4 | 8]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "clo_hip_internal.h"
#include "clo_hip_radix_rank.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// (the product's timing hooks, referenced by the header's launch helpers)
void clo_timing_begin(const char*, hipStream_t) {}
void clo_timing_end(hipStream_t) {}

// keys: bucket b holds positions [off[b], off[b + 1]); key = b << low_bits | hash(position)
template <typename E>
__global__ void fill_kernel(E* keys, const unsigned* off, unsigned low_bits) {
	const unsigned b = blockIdx.x;
	const unsigned lo = off[b], hi = off[b + 1];
	for (unsigned i = lo + threadIdx.x; i < hi; i += blockDim.x) {
		unsigned long long x = (unsigned long long) i * 0x9e3779b97f4a7c15ull + 12345u;
		x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32;
		keys[i] = (E) (((E) b << low_bits) | ((E) x & (((E) 1 << low_bits) - (E) 1)));
	}
}

// EXPERIMENT (probe only): the product's split, every element present, with the stage laid out so that the reload of a
// thread's ITEMS consecutive elements — 16-byte reads 64 bytes apart between lanes: four lanes of every group of 16 on
// one bank quad, 16 LDS cycles instead of 4 — is conflict-free. SWZ 1: the 16-byte chunk index XORed with two bits of the
// thread index (no LDS growth, three more VALU per element); SWZ 2: four dwords of padding after every 64 (LDS + 6 %,
// two more VALU per element).
template <int SWZ, int ITEMS, int PER> __device__ __forceinline__ unsigned swz_index(unsigned p) {   // ITEMS elements per thread, PER per 16 bytes
	constexpr unsigned LI = ITEMS == 16 ? 4 : 3, LP = PER == 4 ? 2 : 1;
	if (SWZ == 1) return p ^ (((p >> (LI + 2)) & 3u) << LP);
	if (SWZ == 2) return p + ((p >> (LI + 2)) << LP);
	return p;
}
template <typename E, int THREADS, int ITEMS, int HMAX, int SWZ>
__device__ __forceinline__ void swz_split(const E (&key)[ITEMS], unsigned dshift, unsigned dmask,
	E* s_stage, unsigned* s_end, unsigned (*s_wtot)[HMAX], unsigned (*s_wbase)[HMAX]) {
	constexpr int BITS = 4;
	constexpr int H = pc_words<BITS>::H;
	constexpr int WAVES = THREADS / 64;
	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	unsigned long long run = 0, c2 = 0;
	unsigned rb[ITEMS / 4];
	const unsigned nbits = (unsigned) __builtin_popcount(dmask);
	#pragma unroll
	for (int k = 0; k < ITEMS / 4; ++k) rb[k] = 0;
	#pragma unroll
	for (int i = ITEMS - 1; i >= 0; --i) {
		if (ITEMS == 16 && i == 7) c2 = run;
		const unsigned sh = pc_digit<E>(key[i], dshift, dmask, nbits) * 4u;
		pc_put_byte(rb[i >> 2], (unsigned) (run >> sh), i & 3);
		run += 1ull << sh;
	}
	const unsigned long long c = run - c2;
	#pragma unroll
	for (int k = 0; k < ITEMS / 4; ++k) rb[k] &= 0x0f0f0f0fu;
	unsigned w[H];
	pc2_wave_scan<BITS, (ITEMS > 8)>(c, c2, w);
	if (lane == 63) {
		#pragma unroll
		for (int j = 0; j < H; ++j) s_wtot[wave][j] = w[j];
	}
	clo_lds_barrier();
	if (wave == 0) {
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
		unsigned acc = 0, dstart16[H];
		#pragma unroll
		for (int j = 0; j < H; ++j) {
			const unsigned t = (unsigned) __builtin_amdgcn_readlane((int) tot[j / 4], (j % 4) * 16 + 15);
			dstart16[j] = acc | ((acc + (t & 0xffffu)) << 16);
			acc += (t & 0xffffu) + (t >> 16);
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
	clo_lds_barrier();
	typedef unsigned short __attribute__((may_alias)) pc_u16;
	pc_u16* const tab = reinterpret_cast<pc_u16*>(s_end) + ((tid & ~63u) + ((tid & 31u) << 1) + ((tid >> 5) & 1u));
	#pragma unroll
	for (int j = 0; j < H; ++j) {
		const unsigned e2 = w[j] + s_wbase[wave][j];
		tab[(2 * j) * THREADS] = (unsigned short) e2;
		tab[(2 * j + 1) * THREADS] = (unsigned short) (e2 >> 16);
	}
	#pragma unroll
	for (int i = 0; i < ITEMS; ++i) {
		if (i > 0 && i % PC_BATCH == 0) __builtin_amdgcn_sched_barrier(0);
		const unsigned d = pc_digit<E>(key[i], dshift, dmask, nbits);
		const unsigned end = tab[d * THREADS];
		s_stage[swz_index<SWZ, ITEMS, 16 / (int) sizeof(E)>(pc_sub_byte(end, rb[i >> 2], i & 3) - 1u)] = key[i];
	}
	clo_lds_barrier();
}

// One work-group per sub-bucket: THREADS x ITEMS slots, the sub-bucket fills the first `count` of them.
// COAL: the sub-bucket goes through the stage on its way in and out (whole-wave runs of consecutive elements instead of
// one run of ITEMS elements per thread at an address that is only element-aligned).
template <typename E, int THREADS, int ITEMS, int MODE, bool PAD, bool COAL, int SWZ = 0>   // SWZ: the experimental stage layouts above (PAD, splits alone or with the per-thread loads / stores); PAD: empty slots hold the largest key and are sorted along (the split's branch-free path); MODE 0: load + splits + store; 1: load + store only (the copy floor); 2: splits on garbage, no global traffic but the offsets
__global__ __launch_bounds__(THREADS)
void local_sort_kernel(E* __restrict__ keys, const unsigned* __restrict__ off, unsigned low_bits, unsigned* __restrict__ oversize) {
	constexpr int H = pc_words<4>::H;
	constexpr int WAVES = THREADS / 64;
	constexpr int CAP = THREADS * ITEMS;
	__shared__ __attribute__((aligned(16))) E s_stage[SWZ == 2 ? CAP + CAP / 16 : CAP];
	__shared__ unsigned s_end[THREADS * PC_END_STRIDE];
	__shared__ unsigned s_wtot[WAVES][H];
	__shared__ unsigned s_wbase[WAVES][H];
	const unsigned b = blockIdx.x;
	const unsigned lo = off[b];
	const unsigned n = off[b + 1] - lo;
	if (n == 0) return;
	if (n > (unsigned) CAP) { if (threadIdx.x == 0) atomicAdd(oversize, 1u); return; }
	const unsigned tbase = threadIdx.x * ITEMS;
	E key[ITEMS];
	constexpr int PER = 16 / (int) sizeof(E);
	typedef E vec16 __attribute__((ext_vector_type(PER)));
	auto reload = [&]() {
		#pragma unroll
		for (int k = 0; k < ITEMS / PER; ++k) {
			const unsigned at = SWZ == 1 ? tbase + ((unsigned) k ^ ((threadIdx.x >> 2) & 3u)) * PER
				: SWZ == 2 ? tbase + (threadIdx.x >> 2) * PER + k * PER : tbase + k * PER;
			const vec16 t = *reinterpret_cast<const vec16*>(&s_stage[at]);
			#pragma unroll
			for (int q = 0; q < PER; ++q) key[k * PER + q] = t[q];
		}
	};
	if (MODE != 2 && COAL) {
		for (unsigned i = threadIdx.x; i < (unsigned) CAP; i += THREADS) s_stage[i] = i < n ? keys[lo + i] : (E) ~(E) 0;
		__syncthreads();
		reload();
		__syncthreads();
	} else if (MODE != 2) {
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) key[i] = (tbase + i < n) ? keys[lo + tbase + i] : (E) ~(E) 0;
	} else {
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) key[i] = (E) ((tbase + i) * 2654435761u + b) * (E) 0x9e3779b1u;
	}
	if (MODE != 1) {
		for (unsigned done = 0; done < low_bits; done += 4) {
			const unsigned bits = low_bits - done < 4u ? low_bits - done : 4u;
			if constexpr (SWZ != 0) swz_split<E, THREADS, ITEMS, H, SWZ>(key, done, (1u << bits) - 1u, s_stage, s_end, s_wtot, s_wbase);
			else pc_local_split<E, 4, THREADS, ITEMS, H>(key, done, (1u << bits) - 1u, PAD ? (unsigned) CAP : n, s_stage, s_end, s_wtot, s_wbase);
			if (!(COAL && MODE == 0) || done + 4 < low_bits) reload();
		}
	}
	if (MODE == 0 && COAL) {
		for (unsigned i = threadIdx.x; i < n; i += THREADS) keys[lo + i] = s_stage[i];
	} else if (MODE == 1 && COAL) {
		#pragma unroll
		for (int k = 0; k < ITEMS / PER; ++k) {
			vec16 t;
			#pragma unroll
			for (int q = 0; q < PER; ++q) t[q] = key[k * PER + q];
			*reinterpret_cast<vec16*>(&s_stage[tbase + k * PER]) = t;
		}
		__syncthreads();
		for (unsigned i = threadIdx.x; i < n; i += THREADS) keys[lo + i] = s_stage[i];
	} else if (MODE != 2) {
		#pragma unroll
		for (int i = 0; i < ITEMS; ++i) if (tbase + i < n) keys[lo + tbase + i] = key[i];
	} else if (key[0] == (E) 0x12345u) keys[lo] = key[1];
}

template <typename E, int THREADS, int ITEMS, int MODE, bool PAD, bool COAL, int SWZ = 0>
static float run(E* keys, const unsigned* off, unsigned buckets, unsigned low_bits, unsigned* oversize, int reps, const E* pristine, size_t n) {
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	float best = 1e9f;
	for (int r = 0; r < reps; ++r) {
		CK(hipMemcpy(keys, pristine, n * sizeof(E), hipMemcpyDeviceToDevice));   // (also: the keys arrive as the pass before would leave them, freshly written)
		CK(hipMemset(oversize, 0, 4));
		CK(hipEventRecord(e0, 0));
		hipLaunchKernelGGL((local_sort_kernel<E, THREADS, ITEMS, MODE, PAD, COAL, SWZ>), dim3(buckets), dim3(THREADS), 0, 0, keys, off, low_bits, oversize);
		CK(hipEventRecord(e1, 0));
		CK(hipEventSynchronize(e1));
		float ms = 0;
		CK(hipEventElapsedTime(&ms, e0, e1));
		best = std::min(best, ms);
	}
	return best;
}

template <typename E>
static int probe(int logn, unsigned top_bits) {
	const unsigned low_bits = (unsigned) sizeof(E) * 8u - top_bits;
	const size_t n = (size_t) 1 << logn;
	const unsigned buckets = 1u << top_bits;
	// sub-bucket sizes of uniform random keys: a multinomial, here by the normal approximation, made to add up
	std::vector<unsigned> off(buckets + 1);
	{
		std::mt19937_64 rng(7);
		const double mean = (double) n / buckets;
		std::normal_distribution<double> nd(mean, std::sqrt(mean));
		std::vector<long> sz(buckets);
		long sum = 0;
		for (unsigned b = 0; b < buckets; ++b) { sz[b] = std::max(0l, (long) std::lround(nd(rng))); sum += sz[b]; }
		long diff = (long) n - sum;
		for (unsigned b = 0; diff != 0; b = (b + 1) % buckets) { if (diff > 0) { ++sz[b]; --diff; } else if (sz[b] > 0) { --sz[b]; ++diff; } }
		off[0] = 0;
		for (unsigned b = 0; b < buckets; ++b) off[b + 1] = off[b] + (unsigned) sz[b];
	}
	E *d_keys, *d_pristine;
	unsigned *d_off, *d_over;
	CK(hipMalloc(&d_keys, n * sizeof(E))); CK(hipMalloc(&d_pristine, n * sizeof(E))); CK(hipMalloc(&d_off, (buckets + 1) * 4)); CK(hipMalloc(&d_over, 4));
	CK(hipMemcpy(d_off, off.data(), (buckets + 1) * 4, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(fill_kernel<E>, dim3(buckets), dim3(256), 0, 0, d_pristine, d_off, low_bits);
	CK(hipDeviceSynchronize());
	printf("2^%d keys of %d bytes in %u sub-buckets (mean %.0f keys), %u low bits to sort on-chip = %u local splits of 4 bits\n", logn, (int) sizeof(E), buckets, (double) n / buckets, low_bits, (low_bits + 3) / 4);
	#define RUN(T, I, P, C) do { \
		const float full = run<E, T, I, 0, P, C>(d_keys, d_off, buckets, low_bits, d_over, 5, d_pristine, n); \
		unsigned over = 0; CK(hipMemcpy(&over, d_over, 4, hipMemcpyDeviceToHost)); \
		/* check: every sub-bucket sorted (the whole array then is) */ \
		std::vector<E> h(1 << 22); CK(hipMemcpy(h.data(), d_keys + (n / 2), h.size() * sizeof(E), hipMemcpyDeviceToHost)); \
		const bool ok = over != 0 || std::is_sorted(h.begin(), h.end()); \
		const float copy = run<E, T, I, 1, P, C>(d_keys, d_off, buckets, low_bits, d_over, 5, d_pristine, n); \
		const float alu = run<E, T, I, 2, P, C>(d_keys, d_off, buckets, low_bits, d_over, 5, d_pristine, n); \
		printf("  %4d threads x %2d slots (capacity %5d)%s%s: sort %.3f ms%s, load + store alone %.3f, splits alone %.3f; oversize sub-buckets %u\n", T, I, T * I, \
			P ? " padded" : "       ", C ? " through the stage" : "                  ", full, ok ? "" : " NOT SORTED", copy, alu, over); \
	} while (0)
	#define RUNS(T, I, Z) do { \
		const float full = run<E, T, I, 0, true, false, Z>(d_keys, d_off, buckets, low_bits, d_over, 5, d_pristine, n); \
		std::vector<E> h(1 << 22); CK(hipMemcpy(h.data(), d_keys + (n / 2), h.size() * sizeof(E), hipMemcpyDeviceToHost)); \
		const bool ok = std::is_sorted(h.begin(), h.end()); \
		const float alu = run<E, T, I, 2, true, false, Z>(d_keys, d_off, buckets, low_bits, d_over, 5, d_pristine, n); \
		printf("  %4d threads x %2d slots, stage layout %d (0 linear as shipped, 1 chunk index XOR thread bits, 2 padded rows): sort %.3f ms%s, splits alone %.3f\n", T, I, Z, full, ok ? "" : " NOT SORTED", alu); \
	} while (0)
	if (sizeof(E) == 4) { RUNS(320, 16, 0); RUNS(320, 16, 1); RUNS(320, 16, 2); RUNS(512, 16, 0); RUNS(512, 16, 1); RUNS(512, 16, 2); }
	else { RUNS(768, 8, 0); RUNS(768, 8, 1); RUNS(768, 8, 2); }
	#undef RUNS
	if (sizeof(E) == 4) {
		RUN(320, 16, true, false); RUN(320, 16, true, true); RUN(320, 16, false, true); RUN(384, 16, true, true); RUN(512, 16, false, true);
	} else {
		RUN(640, 8, true, false); RUN(768, 8, true, true); RUN(320, 16, true, false); RUN(320, 16, true, true); RUN(384, 16, true, true); RUN(384, 16, false, true); RUN(512, 16, true, true);
	}
	#undef RUN
	CK(hipFree(d_keys)); CK(hipFree(d_pristine)); CK(hipFree(d_off)); CK(hipFree(d_over));
	return 0;
}

int main(int argc, char** argv) {
	const int logn = argc > 1 ? atoi(argv[1]) : 28;
	const unsigned top_bits = argc > 2 ? (unsigned) atoi(argv[2]) : 16u;
	const int bytes = argc > 3 ? atoi(argv[3]) : 4;
	return bytes == 8 ? probe<unsigned long long>(logn, top_bits) : probe<unsigned>(logn, top_bits);
}
