// clo_hip_bitonic_impl.h — the bitonic kernels and their host drivers, as
// templates in an anonymous namespace: included by clo_hip_bitonic.hip (C-ABI,
// sbitonic, gselect) and by one translation unit per element size
// (clo_hip_bitonic_e{1,2,4,8}.hip: the tiled schedule of that size), so that the
// build compiles the four sizes side by side. See clo_hip_bitonic.hip for the
// description of the schedules.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>
#include <type_traits>

#include "clo_hip.h"
#include "clo_hip_internal.h"

namespace {

struct key_desc {
	unsigned shift;
	unsigned kind;        // 0 unsigned, 1 signed, 2 float
	unsigned descending;  // CLO_SORT_COMPARE "((a) < (b))"
	unsigned long long mask, signbit;
};

// Key as an unsigned integer whose order is the typed order of the key.
template <typename E>
__device__ __forceinline__ unsigned long long okey(E e, const key_desc& kd) {
	unsigned long long k = ((unsigned long long) e >> kd.shift) & kd.mask;
	if (kd.kind == 1) k ^= kd.signbit;
	else if (kd.kind == 2) k = (k & kd.signbit) ? (~k & kd.mask) : (k | kd.signbit);
	return k;
}

// MODE 3 (the key is the whole element, IEEE floating point): a kernel maps
// every element to the unsigned integer with the same order when it loads it and
// back when it stores it, and runs the unsigned min/max networks in between (the
// general compare recomputes that image for both elements at every step: 2^26
// floats took 29.6 ms against 2.8 ms for 2^26 uints). Equal images are equal
// bit patterns, so equal keys are equal elements here too.
template <typename E, int MODE>
__device__ __forceinline__ E bt_in(E x) {
	if (MODE != 3) return x;
	typedef typename std::make_signed<E>::type S;
	constexpr int W = 8 * (int) sizeof(E);
	const E m = (E) ((E) ((S) x >> (W - 1)) | (E) ((E) 1 << (W - 1)));   // negative: all ones; else the sign bit
	return (E) (x ^ m);
}
template <typename E, int MODE>
__device__ __forceinline__ E bt_out(E y) {
	if (MODE != 3) return y;
	typedef typename std::make_signed<E>::type S;
	constexpr int W = 8 * (int) sizeof(E);
	const E m = (E) ((E) ~(E) ((S) y >> (W - 1)) | (E) ((E) 1 << (W - 1)));
	return (E) (y ^ m);
}

// Compare-exchange with the reference's rule (abitonic.cl:31-38).
// MODE 0: any key (shift/mask/typed compare). MODE 1 / 2: the key is the whole
// element, unsigned / signed integer: equal keys are equal elements, so the
// exchange is min/max (2 VALU + 2 selects instead of ~20); a descending
// compare is the ascending one with the direction bit flipped.
template <typename E, int MODE>
__device__ __forceinline__ void cmpxch(E& a, E& b, unsigned dir, const key_desc& kd) {
	if (MODE == 0) {
		const unsigned long long ka = okey<E>(a, kd), kb = okey<E>(b, kd);
		const bool cmp = kd.descending ? (ka < kb) : (ka > kb);
		if (cmp != (bool) dir) { const E t = a; a = b; b = t; }
	} else {
		typedef typename std::make_signed<E>::type S;
		E lo, hi;
		if (MODE != 2) { lo = a < b ? a : b; hi = a < b ? b : a; }
		else { lo = (S) a < (S) b ? a : b; hi = (S) a < (S) b ? b : a; }
		const bool up = (dir ^ kd.descending) == 0;
		a = up ? lo : hi;
		b = up ? hi : lo;
	}
}

// General keys (MODE 0): the ordered keys are computed ONCE per call and carried
// with their elements through the steps of the call (cmpxch<E, 0> recomputes
// both keys at every exchange: ~19 VALU per pair; here a compare, the selects
// of key and element, and the keys' ~5 VALU per element and call — abitonic of
// 2^26 (uint key, uint value) pairs: 25.5 ms before). The swap rule is
// cmpxch's, bit for bit. K: 32-bit image when the key has at most 32 bits.
template <typename E, typename K, int V, typename DIRFN>
__device__ __forceinline__ void reg_network_keyed(E (&v)[V], int nsteps, const key_desc& kd, DIRFN dirfn, int min_half = 1) {
	K k[V];
	#pragma unroll
	for (int j = 0; j < V; ++j) k[j] = (K) okey<E>(v[j], kd);
	const bool desc = kd.descending != 0;
	#pragma unroll
	for (int half = V / 2; half >= 1; half /= 2) {
		if (half < (1 << nsteps) && half >= min_half) {
			#pragma unroll
			for (int j = 0; j < V; ++j)
				if ((j & half) == 0) {
					const K ka = k[j], kb = k[j + half];
					const bool cmp = desc ? (ka < kb) : (ka > kb);
					const bool sw = cmp != (bool) dirfn(j);
					k[j] = sw ? kb : ka;
					k[j + half] = sw ? ka : kb;
					const E a = v[j], b = v[j + half];
					v[j] = sw ? b : a;
					v[j + half] = sw ? a : b;
				}
		}
	}
}
template <typename E, int V, typename DIRFN>
__device__ __forceinline__ void reg_network_general(E (&v)[V], int nsteps, const key_desc& kd, DIRFN dirfn, int min_half = 1) {
	if (sizeof(E) <= 4 || (kd.mask >> 32) == 0ull) reg_network_keyed<E, unsigned, V>(v, nsteps, kd, dirfn, min_half);
	else reg_network_keyed<E, unsigned long long, V>(v, nsteps, kd, dirfn, min_half);
}

// Up to log2(V) steps on the V values of one thread: strides 2^(nsteps-1) .. 1
// (the register networks of abitonic.cl:163-224, any size). Value j sits at
// element index idx0 | (j << b0); its direction bit is bit S of that index.
template <typename E, int V, int MODE>
__device__ __forceinline__ void reg_network(E (&v)[V], int nsteps, size_t idx0, unsigned b0, unsigned S,
	const key_desc& kd) {
	const unsigned dbase = (unsigned) ((idx0 >> S) & 1);
	// non-zero iff the direction bit is one of this thread's register bits
	const unsigned dsel = (S >= b0 && S - b0 < 31u) ? ((1u << (S - b0)) & (unsigned) (V - 1)) : 0u;
	if (MODE == 0) {
		reg_network_general<E, V>(v, nsteps, kd, [&](int j) { return dsel ? (((unsigned) j & dsel) ? 1u : 0u) : dbase; });
		return;
	}
	#pragma unroll
	for (int half = V / 2; half >= 1; half /= 2) {
		if (half < (1 << nsteps)) {
			#pragma unroll
			for (int j = 0; j < V; ++j)
				if ((j & half) == 0) {
					const unsigned dir = dsel ? (((unsigned) j & dsel) ? 1u : 0u) : dbase;
					cmpxch<E, MODE>(v[j], v[j + half], dir, kd);
				}
		}
	}
}

// The same network when the direction bit is above every index bit the wave
// (or work-group) spans: `dir` is then one scalar, and for whole-element integer
// keys the exchange is a bare min/max pair under a scalar branch.
template <typename E, int V, int MODE, bool UP>
__device__ __forceinline__ void reg_network_minmax(E (&v)[V], int nsteps, int min_half = 1) {
	typedef typename std::make_signed<E>::type S;
	#pragma unroll
	for (int half = V / 2; half >= 1; half /= 2) {
		if (half < (1 << nsteps) && half >= min_half) {
			#pragma unroll
			for (int j = 0; j < V; ++j)
				if ((j & half) == 0) {
					E &a = v[j], &b = v[j + half];
					const bool lt = MODE != 2 ? (a < b) : ((S) a < (S) b);
					const E lo = lt ? a : b, hi = lt ? b : a;
					a = UP ? lo : hi;
					b = UP ? hi : lo;
				}
		}
	}
}
template <typename E, int V, int MODE>
// (min_half: only the strides >= min_half, i.e. the TOP register bits — for a
// group whose low register bits carry no step)
__device__ __forceinline__ void reg_network_uniform(E (&v)[V], int nsteps, unsigned dir, const key_desc& kd, int min_half = 1) {
	if (MODE == 0) {
		reg_network_general<E, V>(v, nsteps, kd, [&](int) { return dir; }, min_half);
	} else if ((dir ^ kd.descending) == 0) {
		reg_network_minmax<E, V, MODE, true>(v, nsteps, min_half);
	} else {
		reg_network_minmax<E, V, MODE, false>(v, nsteps, min_half);
	}
}

// ---- one launch per step (sbitonic.cl:38-69 / abit_any) ----
template <typename E>
__global__ __launch_bounds__(256)
void clo_bitonic_step_kernel(E* __restrict__ data, size_t npairs, unsigned stage, unsigned step, key_desc kd) {
	const size_t gid = (size_t) blockIdx.x * 256 + threadIdx.x;
	if (gid >= npairs) return;
	const unsigned sh = step - 1;
	const size_t i1 = ((gid >> sh) << (sh + 1)) | (gid & (((size_t) 1 << sh) - 1));
	const size_t i2 = i1 + ((size_t) 1 << sh);
	E a = data[i1], b = data[i2];
	const E a0 = a, b0 = b;
	cmpxch<E, 0>(a, b, (unsigned) ((i1 >> stage) & 1), kd);
	if (a != a0 || b != b0) { data[i1] = a; data[i2] = b; }
}

// ---- strided register kernel: steps p .. p-NS+1 of stage S, p-NS >= 6 ----
template <typename E, int NS, int MODE>
__global__ __launch_bounds__(256)
void clo_bitonic_strided_kernel(E* __restrict__ data, size_t n, unsigned stage, unsigned p, key_desc kd) {
	constexpr int V = 1 << NS;
	const size_t t = (size_t) blockIdx.x * 256 + threadIdx.x;
	if (t >= (n >> NS)) return;
	const unsigned b0 = p - NS;  // lowest index bit handled in registers
	const size_t base = ((t >> b0) << (b0 + NS)) | (t & (((size_t) 1 << b0) - 1));
	E v[V];
	#pragma unroll
	for (int j = 0; j < V; ++j) v[j] = bt_in<E, MODE>(data[base + ((size_t) j << b0)]);
	// bit `stage` of base is bit stage-NS >= b0 >= 6 of t: the same for the 64 lanes
	reg_network_uniform<E, V, MODE>(v, NS, __builtin_amdgcn_readfirstlane((unsigned) ((base >> stage) & 1)), kd);
	#pragma unroll
	for (int j = 0; j < V; ++j) data[base + ((size_t) j << b0)] = bt_out<E, MODE>(v[j]);
}

// ---- LDS tile kernel ----
// Q register bits, 256 threads, tile = 2^(8+Q) elements max; kl = log2 of the
// tile actually used (Q <= kl <= 8+Q). mode 0: run steps p_hi..1 of `stage`;
// mode 1: run all of stages 1..stage (stage <= kl).
template <typename E, int Q, int MODE>
__global__ __launch_bounds__(256)
void clo_bitonic_tile_kernel(E* __restrict__ data, unsigned kl, unsigned stage, unsigned p_hi, int mode, key_desc kd) {
	constexpr int V = 1 << Q;
	constexpr int TILE_MAX = 256 * V;
	__shared__ E s[TILE_MAX + TILE_MAX / 32];

	const unsigned tid = threadIdx.x;
	const unsigned tile = 1u << kl;
	const unsigned nthr = tile >> Q;  // active threads
	const size_t gbase = (size_t) blockIdx.x << kl;
	auto phys = [](unsigned i) __attribute__((always_inline)) { return i + (i >> 5); };

	// The thread's V values stay in VGPRs across consecutive step groups that
	// use the same register bits (all of stages 1..Q, for one). The first group
	// is loaded straight from global memory and the last one stored straight
	// back (its register bits are the lowest Q: V consecutive elements, 16-byte
	// vectors); LDS only carries the exchanges between groups. (In an exchange a
	// thread overwrites exactly the LDS slots it last read — each layout
	// partitions the tile among the threads — so one barrier per exchange.)
	typedef E vec16 __attribute__((ext_vector_type(16 / sizeof(E)), aligned(sizeof(E))));   // element alignment is all a caller guarantees
	constexpr int PER = 16 / (int) sizeof(E);
	static_assert(V % PER == 0, "a thread's consecutive run is whole 16-byte vectors");
	// Whole-element integer keys, all stages of a tile (mode 1): an element whose
	// stage direction is "down" is held COMPLEMENTED for that stage (the direction
	// is a bit of the element's own index, so every thread agrees, also across the
	// LDS exchanges), which makes every exchange of the stage an ascending min/max.
	// The complement state moves at each stage's first group and is undone at the
	// end. Any network gives the same result for these keys: equal keys are equal
	// elements.
	constexpr bool CPL = MODE != 0;
	E v[V];
	int cur_b0 = -1;
	unsigned base = 0;
	const unsigned s_first = mode ? 1u : stage;
	for (unsigned S = s_first; S <= stage; ++S) {
		unsigned p = mode ? S : p_hi;
		while (p >= 1) {
			// register bits [b0, b0+Q) of the tile index; steps p .. b0+1
			const unsigned b0 = p > (unsigned) Q ? p - Q : 0u;
			const int nsteps = (int) (p - b0);
			if (cur_b0 != (int) b0) {
				if (cur_b0 >= 0) {
					if (tid < nthr) {
						#pragma unroll
						for (int j = 0; j < V; ++j) s[phys(base + ((unsigned) j << cur_b0))] = v[j];
					}
					__syncthreads();
				}
				base = ((tid >> b0) << (b0 + Q)) | (tid & ((1u << b0) - 1u));
				if (tid < nthr) {
					if (cur_b0 < 0) {
						// first group: from global memory (b0 == 0: V consecutive elements;
						// b0 >= 6 or a single wave: lanes read adjacent elements)
						if (b0 == 0) {
							const vec16* src = reinterpret_cast<const vec16*>(data + gbase + base);
							#pragma unroll
							for (int k = 0; k < V / PER; ++k) {
								const vec16 t = src[k];
								#pragma unroll
								for (int q = 0; q < PER; ++q) v[k * PER + q] = bt_in<E, MODE>(t[q]);
							}
						} else {
							#pragma unroll
							for (int j = 0; j < V; ++j) v[j] = bt_in<E, MODE>(data[gbase + base + ((unsigned) j << b0)]);
						}
					} else {
						#pragma unroll
						for (int j = 0; j < V; ++j) v[j] = s[phys(base + ((unsigned) j << b0))];
					}
				}
				cur_b0 = (int) b0;
			}
			if (tid < nthr) {
				if (CPL && mode) {
					// first group of a stage: move every element to the stage's complement state
					if (p == S) {
						#pragma unroll
						for (int j = 0; j < V; ++j) {
							const unsigned idx = (unsigned) gbase + base + ((unsigned) j << b0);
							const unsigned f = (idx >> S) ^ (S == 1 ? kd.descending : (idx >> (S - 1)));
							v[j] ^= (E) ((E) 0 - (E) (f & 1u));
						}
					}
					reg_network_minmax<E, V, MODE, true>(v, nsteps);
				}
				// from stage kl up the direction bit is a bit of the tile number
				else if (S >= kl) reg_network_uniform<E, V, MODE>(v, nsteps, (unsigned) ((gbase >> S) & 1), kd);
				else reg_network<E, V, MODE>(v, nsteps, gbase + base, b0, S, kd);
			}
			p = b0;
		}
	}
	// every schedule ends on steps Q..1, i.e. with b0 == 0: V consecutive elements per thread
	if (tid < nthr) {
		if (CPL && mode) {
			#pragma unroll
			for (int j = 0; j < V; ++j) {
				const unsigned f = (((unsigned) gbase + base + (unsigned) j) >> stage) ^ kd.descending;
				v[j] ^= (E) ((E) 0 - (E) (f & 1u));
			}
		}
		vec16* dst = reinterpret_cast<vec16*>(data + gbase + base);
		#pragma unroll
		for (int k = 0; k < V / PER; ++k) {
			vec16 t;
			#pragma unroll
			for (int q = 0; q < PER; ++q) t[q] = bt_out<E, MODE>(v[k * PER + q]);
			dst[k] = t;
		}
	}
}

// ---- full tiles of 2^(TB+Q) elements, 2^TB threads, schedule fixed at compile time ----
// (all of stages 1..TB+Q: the presort; the merge passes follow)
// Every n >= 2^(TB+Q) runs these: with the stage / group loops unrolled the
// layouts are constants, so an LDS exchange is one thread base plus immediate
// offsets (the run-time schedule spends 4 VALU per LDS access on addresses:
// 422 VALU per element in the 91-step presort, SQ_INSTS_VALU) and the
// complement-state masks are one XOR per element and stage.
template <int A, int B, typename F>
__device__ __forceinline__ void static_for(F&& f) {
	if constexpr (A < B) {
		f(std::integral_constant<int, A>());
		static_for<A + 1, B>(f);
	}
}

template <typename E, int Q, int TB, int MODE>
__global__ __launch_bounds__(1 << TB)
void clo_bitonic_tile_presort_kernel(E* __restrict__ data, key_desc kd) {
	constexpr int V = 1 << Q;
	constexpr int KL = TB + Q;
	constexpr int TILE = V << TB;
	constexpr bool CPL = MODE != 0;
	__shared__ E s[TILE + TILE / 32];
	typedef E vec16 __attribute__((ext_vector_type(16 / sizeof(E)), aligned(sizeof(E))));
	constexpr int PER = 16 / (int) sizeof(E);
	static_assert(V % PER == 0, "a thread's consecutive run is whole 16-byte vectors");

	const unsigned tid = threadIdx.x;
	const size_t gbase = (size_t) blockIdx.x << KL;
	// layout b0: value j of thread t is tile element tbase(b0) | (j << b0); the
	// two parts share no bits, so the padded LDS slot is phys(tbase) + a constant
	auto tbase = [&](int b0) __attribute__((always_inline)) { return ((tid >> b0) << (b0 + Q)) | (tid & ((1u << b0) - 1u)); };
	auto phys = [](unsigned i) __attribute__((always_inline)) { return i + (i >> 5); };
	E v[V];

	auto exchange = [&](int from, int to) __attribute__((always_inline)) {
		const unsigned pf = phys(tbase(from)), pt = phys(tbase(to));
		#pragma unroll
		for (int j = 0; j < V; ++j) s[pf + phys((unsigned) j << from)] = v[j];
		__syncthreads();
		#pragma unroll
		for (int j = 0; j < V; ++j) v[j] = s[pt + phys((unsigned) j << to)];
	};
	auto group = [&](int S, int b0, int nsteps) __attribute__((always_inline)) {
		if (CPL) reg_network_minmax<E, V, MODE, true>(v, nsteps);
		else reg_network<E, V, MODE>(v, nsteps, gbase + tbase(b0), (unsigned) b0, (unsigned) S, kd);
	};

	// In and out through LDS: the first and last layouts give a thread V
	// consecutive elements, and 16-byte accesses at a lane stride of V elements
	// are 64 partial cache lines per instruction; transposed, a wave moves 1 KiB
	// contiguous (the merge passes: 111 -> 83 us per 2^26 uint32).
	{
		const vec16* src = reinterpret_cast<const vec16*>(data + gbase) + tid;
		#pragma unroll
		for (int k = 0; k < V / PER; ++k) {
			const vec16 t = src[(unsigned) k << TB];
			const unsigned pe = phys(((unsigned) k << TB) * PER + tid * PER);   // PER consecutive slots: no multiple of 32 inside
			#pragma unroll
			for (int q = 0; q < PER; ++q) s[pe + q] = bt_in<E, MODE>(t[q]);
		}
		__syncthreads();
		const unsigned pt = phys(tbase(0));
		#pragma unroll
		for (int j = 0; j < V; ++j) v[j] = s[pt + (unsigned) j];
	}
	{
		static_for<1, KL + 1>([&](auto Sc) __attribute__((always_inline)) {
			constexpr int S = decltype(Sc)::value;
			static_for<0, (S + Q - 1) / Q>([&](auto gc) __attribute__((always_inline)) {
				constexpr int g = decltype(gc)::value;
				constexpr int p = S - g * Q;
				constexpr int b0 = p > Q ? p - Q : 0;
				constexpr int prev = g > 0 ? p : 0;   // the layout the values are in now
				if (prev != b0) exchange(prev, b0);
				if (CPL && g == 0) {
					// move to stage S's complement state: bit S of the index (^ descending),
					// coming from stage S-1's (nothing complemented before stage 1)
					const unsigned tb = (unsigned) gbase + tbase(b0);
					const unsigned ft = S == 1 ? ((tb >> 1) ^ kd.descending) : ((tb >> S) ^ (tb >> (S - 1)));
					const E m0 = (E) ((E) 0 - (E) (ft & 1u)), m1 = (E) ~m0;
					#pragma unroll
					for (int j = 0; j < V; ++j) {
						const unsigned ij = (unsigned) j << b0;
						const unsigned fj = S == 1 ? (ij >> 1) : ((ij >> S) ^ (ij >> (S - 1)));
						v[j] ^= (fj & 1u) ? m1 : m0;
					}
				}
				group(S, b0, p - b0);
			});
		});
		if (CPL && ((unsigned) (gbase >> KL) & 1u) != kd.descending) {
			#pragma unroll
			for (int j = 0; j < V; ++j) v[j] = (E) ~v[j];
		}
	}
	{
		const unsigned pf = phys(tbase(0));
		#pragma unroll
		for (int j = 0; j < V; ++j) s[pf + (unsigned) j] = v[j];
		__syncthreads();
		vec16* dst = reinterpret_cast<vec16*>(data + gbase) + tid;
		#pragma unroll
		for (int k = 0; k < V / PER; ++k) {
			const unsigned pe = phys(((unsigned) k << TB) * PER + tid * PER);
			vec16 t;
			#pragma unroll
			for (int q = 0; q < PER; ++q) t[q] = bt_out<E, MODE>(s[pe + q]);
			dst[(unsigned) k << TB] = t;
		}
	}
}

// ---- steps KL..1 of one stage > KL on full tiles: the merge passes ----
// The direction of a whole tile is one bit of its number: a scalar branch
// around a bare min/max network. The last layout gives a thread V consecutive
// elements, and 16-byte stores at that lane stride are 64 partial cache lines
// per instruction: the tile leaves through one more LDS transpose, a wave
// storing 1 KiB contiguous (111 -> 83 us per pass of 2^26 uint32). (A group
// walking over several tiles with the next tile's loads in flight measured the
// same for 4-byte elements and 35 % slower for 8-byte ones: 32 more registers.)
template <typename E, int Q, int TB, int MODE>
__global__ __launch_bounds__(1 << TB)
void clo_bitonic_tile_merge_kernel(E* __restrict__ data, unsigned stage, key_desc kd) {
	constexpr int V = 1 << Q;
	constexpr int KL = TB + Q;
	constexpr int TILE = V << TB;
	__shared__ E s[TILE + TILE / 32];
	typedef E vec16 __attribute__((ext_vector_type(16 / sizeof(E)), aligned(sizeof(E))));
	constexpr int PER = 16 / (int) sizeof(E);
	static_assert(V % PER == 0, "a thread's consecutive run is whole 16-byte vectors");

	const unsigned tid = threadIdx.x;
	auto tbase = [&](int b0) __attribute__((always_inline)) { return ((tid >> b0) << (b0 + Q)) | (tid & ((1u << b0) - 1u)); };
	auto phys = [](unsigned i) __attribute__((always_inline)) { return i + (i >> 5); };
	E v[V];
	auto exchange = [&](int from, int to) __attribute__((always_inline)) {
		const unsigned pf = phys(tbase(from)), pt = phys(tbase(to));
		#pragma unroll
		for (int j = 0; j < V; ++j) s[pf + phys((unsigned) j << from)] = v[j];
		__syncthreads();
		#pragma unroll
		for (int j = 0; j < V; ++j) v[j] = s[pt + phys((unsigned) j << to)];
	};
	const size_t gbase = (size_t) blockIdx.x << KL;
	const unsigned dir = (unsigned) ((gbase >> stage) & 1);
	{
		// first group: register bits [KL-Q, KL), lanes read adjacent elements
		const E* src = data + gbase + tbase(KL - Q);
		#pragma unroll
		for (int j = 0; j < V; ++j) v[j] = bt_in<E, MODE>(src[(unsigned) j << (KL - Q)]);
	}
	static_for<0, (KL + Q - 1) / Q>([&](auto gc) __attribute__((always_inline)) {
		constexpr int g = decltype(gc)::value;
		constexpr int p = KL - g * Q;
		constexpr int b0 = p > Q ? p - Q : 0;
		if (g > 0) exchange(p, b0);
		reg_network_uniform<E, V, MODE>(v, p - b0, dir, kd);
	});
	{
		const unsigned pf = phys(tbase(0));
		#pragma unroll
		for (int j = 0; j < V; ++j) s[pf + (unsigned) j] = v[j];
		__syncthreads();
		vec16* dst = reinterpret_cast<vec16*>(data + gbase) + tid;
		#pragma unroll
		for (int k = 0; k < V / PER; ++k) {
			const unsigned pe = phys(((unsigned) k << TB) * PER + tid * PER);   // PER consecutive slots: no multiple of 32 inside
			vec16 t;
			#pragma unroll
			for (int q = 0; q < PER; ++q) t[q] = bt_out<E, MODE>(s[pe + q]);
			dst[(unsigned) k << TB] = t;
		}
	}
}

// ---- steps KL+1 .. 1 of one stage > KL on PAIRS of full tiles: a merge pass that takes the lowest strided step along ----
// A work-group owns two neighbouring tiles (2^(KL+1) contiguous elements): a thread loads its V values of the first
// tile and the V at the same places of the second (step KL+1 pairs exactly those: V compare-exchanges in registers), then
// runs the merge pass above on the first tile and, through the same LDS array, on the second, whose values wait in
// registers meanwhile. Where it pays: the stages whose strided part thereby loses a whole pass (stage KL+1: no strided pass
// at all; the stage whose strided steps otherwise need one pass more) — twice the registers, half the waves per CU.
template <typename E, int Q, int TB, int MODE>
// (One group per CU: 130-138 VGPRs. Capped at 128 for two groups the compiler spills 8-16 registers, and a sort with
// scratch-using launches in it measured SLOWER than without the two-tile merge: 2.52 against 2.48 ms, 2^26 uint32.)
__global__ __launch_bounds__(1 << TB)
void clo_bitonic_tile_merge2_kernel(E* __restrict__ data, unsigned stage, key_desc kd) {
	constexpr int V = 1 << Q;
	constexpr int KL = TB + Q;
	constexpr int TILE = V << TB;
	__shared__ E s[TILE + TILE / 32];
	typedef E vec16 __attribute__((ext_vector_type(16 / sizeof(E)), aligned(sizeof(E))));
	constexpr int PER = 16 / (int) sizeof(E);
	static_assert(V % PER == 0, "a thread's consecutive run is whole 16-byte vectors");

	const unsigned tid = threadIdx.x;
	auto tbase = [&](int b0) __attribute__((always_inline)) { return ((tid >> b0) << (b0 + Q)) | (tid & ((1u << b0) - 1u)); };
	auto phys = [](unsigned i) __attribute__((always_inline)) { return i + (i >> 5); };
	const size_t gbase = (size_t) blockIdx.x << (KL + 1);
	const unsigned dir = (unsigned) ((gbase >> stage) & 1);   // stage > KL: a bit of the pair's number
	E a[V], b[V];
	{
		// first group of the merge: register bits [KL-Q, KL), lanes read adjacent elements — of both tiles
		const E* src = data + gbase + tbase(KL - Q);
		#pragma unroll
		for (int j = 0; j < V; ++j) a[j] = bt_in<E, MODE>(src[(unsigned) j << (KL - Q)]);
		#pragma unroll
		for (int j = 0; j < V; ++j) b[j] = bt_in<E, MODE>(src[(size_t) TILE + ((unsigned) j << (KL - Q))]);
	}
	// step KL+1: element i of the first tile against element i of the second (the direction is the pair's: a scalar
	// branch around bare min / max — selects would keep both results of all V pairs alive)
	if (MODE == 0) {
		#pragma unroll
		for (int j = 0; j < V; ++j) cmpxch<E, 0>(a[j], b[j], dir, kd);
	} else {
		typedef typename std::make_signed<E>::type S;
		if ((dir ^ kd.descending) == 0) {
			#pragma unroll
			for (int j = 0; j < V; ++j) {
				const bool lt = MODE != 2 ? (a[j] < b[j]) : ((S) a[j] < (S) b[j]);
				const E lo = lt ? a[j] : b[j], hi = lt ? b[j] : a[j];
				a[j] = lo; b[j] = hi;
			}
		} else {
			#pragma unroll
			for (int j = 0; j < V; ++j) {
				const bool lt = MODE != 2 ? (a[j] < b[j]) : ((S) a[j] < (S) b[j]);
				const E lo = lt ? a[j] : b[j], hi = lt ? b[j] : a[j];
				a[j] = hi; b[j] = lo;
			}
		}
	}
	auto merge_one = [&](E (&v)[V], size_t tile_base, bool again) __attribute__((always_inline)) {
		auto exchange = [&](int from, int to) __attribute__((always_inline)) {
			const unsigned pf = phys(tbase(from)), pt = phys(tbase(to));
			#pragma unroll
			for (int j = 0; j < V; ++j) s[pf + phys((unsigned) j << from)] = v[j];
			__syncthreads();
			#pragma unroll
			for (int j = 0; j < V; ++j) v[j] = s[pt + phys((unsigned) j << to)];
		};
		static_for<0, (KL + Q - 1) / Q>([&](auto gc) __attribute__((always_inline)) {
			constexpr int g = decltype(gc)::value;
			constexpr int p = KL - g * Q;
			constexpr int b0 = p > Q ? p - Q : 0;
			if (g > 0) exchange(p, b0);
			reg_network_uniform<E, V, MODE>(v, p - b0, dir, kd);
		});
		{
			const unsigned pf = phys(tbase(0));
			#pragma unroll
			for (int j = 0; j < V; ++j) s[pf + (unsigned) j] = v[j];
			__syncthreads();
			vec16* dst = reinterpret_cast<vec16*>(data + tile_base) + tid;
			#pragma unroll
			for (int k = 0; k < V / PER; ++k) {
				const unsigned pe = phys(((unsigned) k << TB) * PER + tid * PER);   // PER consecutive slots: no multiple of 32 inside
				vec16 t;
				#pragma unroll
				for (int q = 0; q < PER; ++q) t[q] = bt_out<E, MODE>(s[pe + q]);
				dst[(unsigned) k << TB] = t;
			}
		}
		if (again) __syncthreads();   // (the array is read out: the second tile may go in)
	};
	__builtin_amdgcn_sched_barrier(0);
	merge_one(a, gbase, true);
	__builtin_amdgcn_sched_barrier(0);
	merge_one(b, gbase + (size_t) TILE, false);
}

// ---- two-level strided pass: steps p .. p-ns+1 (Q < ns <= 2Q) of a stage in ONE pass ----
// A work-group owns 2^ns rows (the index bits [p-ns, p)) of 2^(KL-ns) contiguous
// elements each, as a tile of 2^KL elements in LDS order t = row * 2^(KL-ns) +
// column. Threads first hold the top Q row bits in registers (Q steps), exchange
// once through LDS, then hold tile bits [KL-2Q, KL-Q), whose top ns-Q bits are the
// remaining row bits (ns-Q steps). Rows are >= 64 bytes long, lanes run along
// them. One pass where the plain strided kernel needs two.
template <typename E, int Q, int TB, int MODE>
__global__ __launch_bounds__(1 << TB)
void clo_bitonic_strided2_kernel(E* __restrict__ data, unsigned stage, unsigned p, unsigned ns, key_desc kd) {
	constexpr int V = 1 << Q;
	constexpr int KL = TB + Q;
	constexpr int TILE = V << TB;
	constexpr int B1 = KL - Q, B2 = KL - 2 * Q;   // lowest tile bit held in registers, first / second group
	static_assert(B2 >= 0, "two register groups inside the tile");
	__shared__ E s[TILE + TILE / 32];

	const unsigned tid = threadIdx.x;
	auto tbase = [&](int b0) __attribute__((always_inline)) { return ((tid >> b0) << (b0 + Q)) | (tid & ((1u << b0) - 1u)); };
	auto phys = [](unsigned i) __attribute__((always_inline)) { return i + (i >> 5); };
	const unsigned C = (unsigned) KL - ns;        // column bits of the tile
	const unsigned midbits = p - ns - C;          // index bits between the columns and the rows
	const size_t w = blockIdx.x;
	const size_t wbase = ((w >> midbits) << p) | ((w & (((size_t) 1 << midbits) - 1)) << C);
	const unsigned cmask = (1u << C) - 1u;
	auto gaddr = [&](unsigned t) __attribute__((always_inline)) { return wbase + ((size_t) (t >> C) << (p - ns)) + (t & cmask); };
	const unsigned dir = (unsigned) ((wbase >> stage) & 1);   // stage >= p: a bit of the group's own part of the index
	E v[V];
	{
		// register bits = the top Q row bits = index bits [p-Q, p)
		const E* src = data + gaddr(tbase(B1));
		#pragma unroll
		for (int j = 0; j < V; ++j) v[j] = bt_in<E, MODE>(src[(size_t) j << (p - Q)]);
	}
	reg_network_uniform<E, V, MODE>(v, Q, dir, kd);
	{
		const unsigned pf = phys(tbase(B1)), pt = phys(tbase(B2));
		#pragma unroll
		for (int j = 0; j < V; ++j) s[pf + phys((unsigned) j << B1)] = v[j];
		__syncthreads();
		#pragma unroll
		for (int j = 0; j < V; ++j) v[j] = s[pt + phys((unsigned) j << B2)];
	}
	reg_network_uniform<E, V, MODE>(v, Q, dir, kd, V >> (ns - Q));
	{
		const unsigned t0 = tbase(B2);
		#pragma unroll
		for (int j = 0; j < V; ++j) data[gaddr(t0 | ((unsigned) j << B2))] = bt_out<E, MODE>(v[j]);
	}
}

// ---- pad the tail [numel, padded) with elements that sort last ----
template <typename E>
__global__ void clo_bitonic_pad_kernel(E* data, size_t numel, size_t padded, E pad) {
	const size_t i = numel + (size_t) blockIdx.x * 256 + threadIdx.x;
	if (i < padded) data[i] = pad;
}

size_t nlpo2(size_t x) {
	size_t p = 1;
	while (p < x) p <<= 1;
	return p;
}

unsigned log2u(size_t x) {
	unsigned l = 0;
	while (((size_t) 1 << l) < x) ++l;
	return l;
}

template <typename E>
int make_desc(int key_shift, int key_bits, int key_size, int key_kind, int descending, key_desc* kd, E* pad) {
	if (key_size != 1 && key_size != 2 && key_size != 4 && key_size != 8) return CLO_HIP_EARGS;
	if (key_bits < 1 || key_bits > 8 * key_size) return CLO_HIP_EARGS;
	if (key_shift < 0 || key_shift + key_bits > 8 * (int) sizeof(E)) return CLO_HIP_EARGS;
	if (key_kind < 0 || key_kind > 2) return CLO_HIP_EARGS;
	if (key_kind == 2 && (key_size < 2 || key_bits != 8 * key_size)) return CLO_HIP_EUNSUPPORTED;
	kd->shift = (unsigned) key_shift;
	kd->kind = (unsigned) key_kind;
	kd->descending = descending ? 1u : 0u;
	kd->mask = key_bits == 64 ? ~0ull : ((1ull << key_bits) - 1ull);
	kd->signbit = 1ull << (8 * key_size - 1);
	if (key_kind == 1 && key_bits < 8 * key_size) kd->kind = 0;  // sign bit masked off: plain unsigned order
	// Element whose key is the last one in the requested order.
	unsigned long long last_ordered = descending ? 0ull : kd->mask;  // in okey space
	unsigned long long k = last_ordered;
	if (key_kind == 1) k ^= kd->signbit;
	else if (key_kind == 2) k = (k & kd->signbit) ? (k & ~kd->signbit) : (~k & kd->mask);
	// all other bits of the element: ones (any value would do)
	unsigned long long e = ~0ull;
	e &= ~(kd->mask << key_shift);
	e |= (k & kd->mask) << key_shift;
	*pad = (E) e;
	return 0;
}

template <typename E>
int pad_tail(E* data, size_t numel, size_t padded, E pad, hipStream_t s) {
	if (padded > numel) {
		const size_t cnt = padded - numel;
		hipLaunchKernelGGL((clo_bitonic_pad_kernel<E>), dim3((unsigned) ((cnt + 255) / 256)), dim3(256), 0, s,
			data, numel, padded, pad);
	}
	return 0;
}

template <typename E>
int simple_impl(void* vdata, size_t numel, int key_shift, int key_bits, int key_size, int key_kind, int descending,
	int* launches, hipStream_t s) {
	E* data = (E*) vdata;
	key_desc kd; E pad;
	int st = make_desc<E>(key_shift, key_bits, key_size, key_kind, descending, &kd, &pad);
	if (st) return st;
	const size_t n = nlpo2(numel);
	const unsigned T = log2u(n);
	pad_tail<E>(data, numel, n, pad, s);
	int count = 0;
	const size_t npairs = n / 2;
	const unsigned blocks = (unsigned) ((npairs + 255) / 256);
	for (unsigned stage = 1; stage <= T; ++stage)
		for (unsigned step = stage; step >= 1; --step) {
			clo_timing_scope timing("bitonic_step", s);
			hipLaunchKernelGGL((clo_bitonic_step_kernel<E>), dim3(blocks), dim3(256), 0, s, data, npairs, stage, step, kd);
			++count;
		}
	if (launches) *launches = count;
	return (int) hipGetLastError();
}

// Any numel, any key (round 3): the network in its "flip" form, in place, no padding. The first step of a stage
// compares position o of a block's first half with position B - 1 - o of the block, every other step is a
// half-cleaner, and EVERY comparator puts the element that compares first at the lower index: elements past
// numel then behave like +infinity that never moves, and a comparator whose upper index is >= numel is skipped.
// For keys that are part of the element a sentinel would show in the tie order, which is why the padded
// direction-bit network (simple_impl / tiled_run) is used for whole-element keys only; upstream itself sorts
// powers of two only (its kernels have no bounds), so there is no reference tie order to keep here.
template <typename E>
__global__ __launch_bounds__(256)
void clo_bitonic_step_any_kernel(E* __restrict__ data, size_t n, size_t npairs, unsigned stage, unsigned step, key_desc kd) {
	const size_t gid = (size_t) blockIdx.x * 256 + threadIdx.x;
	if (gid >= npairs) return;
	const unsigned sh = step - 1;
	const size_t i1 = ((gid >> sh) << (sh + 1)) | (gid & (((size_t) 1 << sh) - 1));
	const size_t i2 = step == stage ? (i1 | (((size_t) 1 << stage) - 1)) - (i1 & (((size_t) 1 << sh) - 1)) : i1 + ((size_t) 1 << sh);
	if (i2 >= n) return;
	E a = data[i1], b = data[i2];
	const E a0 = a, b0 = b;
	cmpxch<E, 0>(a, b, 0u, kd);
	if (a != a0 || b != b0) { data[i1] = a; data[i2] = b; }
}

template <typename E>
int any_impl(void* vdata, size_t numel, int key_shift, int key_bits, int key_size, int key_kind, int descending,
	int* launches, hipStream_t s) {
	E* data = (E*) vdata;
	key_desc kd; E pad;
	int st = make_desc<E>(key_shift, key_bits, key_size, key_kind, descending, &kd, &pad);
	if (st) return st;
	const size_t n = nlpo2(numel);
	const unsigned T = log2u(n);
	int count = 0;
	const size_t npairs = n / 2;
	const unsigned blocks = (unsigned) ((npairs + 255) / 256);
	for (unsigned stage = 1; stage <= T; ++stage)
		for (unsigned step = stage; step >= 1; --step) {
			clo_timing_scope timing("bitonic_step", s);
			hipLaunchKernelGGL((clo_bitonic_step_any_kernel<E>), dim3(blocks), dim3(256), 0, s, data, numel, npairs, stage, step, kd);
			++count;
		}
	if (launches) *launches = count;
	return (int) hipGetLastError();
}

template <typename E, int NS, int MODE>
void launch_strided(E* data, size_t n, unsigned stage, unsigned p, const key_desc& kd, hipStream_t s) {
	const size_t threads = n >> NS;
	clo_timing_scope timing("bitonic_strided", s);
	hipLaunchKernelGGL((clo_bitonic_strided_kernel<E, NS, MODE>), dim3((unsigned) ((threads + 255) / 256)), dim3(256), 0, s,
		data, n, stage, p, kd);
}

#ifndef CLO_MERGE2_EXTRA
#define CLO_MERGE2_EXTRA 20u
#endif
template <typename E, int MODE, int TBF = 9>
int tiled_run(void* vdata, size_t numel, int key_shift, int key_bits, int key_size, int key_kind, int descending,
	int* launches, hipStream_t s) {
	// register bits per thread: 32 values of <= 4 bytes, 16 values of 8 bytes
	constexpr int Q = sizeof(E) == 8 ? 4 : 5;
	// Arrays of at least 2^KLF elements: 512-thread groups on tiles of 2^KLF
	// (67 KiB of LDS, two groups per CU) with the compile-time schedule. Smaller
	// ones are one tile, sorted by one launch of the run-time-schedule kernel.
	constexpr unsigned KLF = TBF + Q;
	static_assert(TBF > 9 || KLF - 1 <= 8 + Q, "the run-time kernel covers every smaller array");
	// strided passes need p - NS >= 6 so that a wave's 64 lanes read one
	// contiguous row; KLF >= 13 guarantees it for every p > KLF.
	E* data = (E*) vdata;
	key_desc kd; E pad;
	int st = make_desc<E>(key_shift, key_bits, key_size, key_kind, descending, &kd, &pad);
	if (st) return st;
	const size_t n = nlpo2(numel);
	const unsigned T = log2u(n);
	if (T < (unsigned) Q) return simple_impl<E>(vdata, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
	pad_tail<E>(data, numel, n, pad, s);
	const unsigned kl = T < KLF ? T : KLF;
	const unsigned tiles = (unsigned) (n >> kl);
	int count = 0;
	// stages 1..kl inside the tiles
	{
		clo_timing_scope timing("bitonic_presort", s);
		if (kl == KLF)
			hipLaunchKernelGGL((clo_bitonic_tile_presort_kernel<E, Q, TBF, MODE>), dim3(tiles), dim3(1 << TBF), 0, s, data, kd);
		else
			hipLaunchKernelGGL((clo_bitonic_tile_kernel<E, Q, MODE>), dim3(tiles), dim3(256), 0, s, data, kl, kl, kl, 1, kd);
	}
	++count;
	// Cost of the strided passes for h steps above the tile (us per 2^26 uint32, see below), and where the two-tile merge
	// (which takes the lowest of them along) saves a whole pass.
	const int merge2_mode = clo_hip_env()->bitonic_merge2;   // 0: never, 2: at every stage (tests, A/B runs); 1: where it saves a pass
	// register bits of a strided pass: 64 values per thread for identity keys,
	// one wave per SIMD — these passes only stream (76 us per 2^26 uint32). The
	// general compare needs more temporaries: 32 (16 of 8 bytes). 128 values
	// were measured too: 100-117 us per pass (412 VGPRs and up, spills), which
	// costs more over a sort than the one pass it saves.
	constexpr int QS = MODE == 0 ? Q + 1 : 6;
	// The passes' measured times (2^26 uint32, us per pass: plain 72 + k for k <= 5 steps, 81 for 6; two-level 77 / 78 / 82
	// for 7 / 8 / 9 steps and 124 for 10, whose rows are 64 bytes, half a cache line: profiles/r04_abitonic_strided_by_position.txt)
	auto plain_cost = [&](unsigned k) { unsigned c = 0; while (k > 0) { const unsigned t = k > (unsigned) QS ? (k % QS ? k % QS : QS) : k; c += t >= 6 ? 81u : 72u + t; k -= t; } return c; };
	// (a two-level pass of b2 steps reads rows of 2^(KLF - b2) elements: 77 us and 2 more per halving down to one
	// cache line, 124 for half a line)
	auto s2_cost = [&](unsigned b2) { const size_t row = ((size_t) 1 << (KLF - b2)) * sizeof(E); return row >= 128 ? 75u + 2u * (b2 - (unsigned) Q) : (row >= 64 ? 124u : 250u); };
	// cheapest cut of h strided steps: plain passes first (the top steps), then at most ONE two-level pass of Q + 2 .. 2Q steps
	auto strided_cost = [&](unsigned h) { unsigned best = plain_cost(h); for (unsigned b2 = Q + 2; b2 <= 2u * Q && b2 <= h; ++b2) { const unsigned c = plain_cost(h - b2) + s2_cost(b2); if (c < best) best = c; } return best; };
	for (unsigned stage = kl + 1; stage <= T; ++stage) {
		unsigned p = stage;
		// The two-tile merge where it saves more than it costs (CLO_MERGE2_EXTRA us over a merge pass: 0.099 against 0.082 ms):
		// 2^26 uint32 — stages 15, 24, 25 — 2.48 -> 2.39 ms, 2^24 0.556 -> 0.535; at every stage: 2.55 (profiles/r05_abitonic_merge2.txt).
		// From 64 tiles on (fewer: its groups, one per CU, are too few — 2^16 elements 0.040 -> 0.042 ms). 4-byte elements only: on
		// 8-byte ones the two-tile pass costs 0.167 ms against 0.108 (2^25 ulong 3.19 -> 3.21 ms with it); 1- and 2-byte: not measured.
		const bool use_m2 = MODE != 0 && merge2_mode != 0 && kl == KLF && tiles >= (merge2_mode == 2 ? 2u : 64u) && (merge2_mode == 2 || (sizeof(E) == 4 && strided_cost(stage - kl) > strided_cost(stage - kl - 1u) + CLO_MERGE2_EXTRA));
		const unsigned stop = use_m2 ? kl + 1u : kl;   // the strided passes end above this step
		while (p > stop) {
			unsigned ns = p - stop;
			// How the h = p - stop steps above the tile (above the two-tile merge's step) are cut into passes: plain passes of up
			// to QS steps first (the top steps), then ONE two-level pass of Q + 2 .. 2Q steps on the steps right above the tile —
			// whichever cut costs least by the passes' measured times. Round 3 took the two-level pass first and as long as it
			// could be: 11 steps = 10 + 1, 12 = 6 + 6; now 11 = 4 + 7, 12 = 5 + 7.
			{
				const unsigned h = ns;
				unsigned best = plain_cost(h), best_b = 0;
				for (unsigned b2 = Q + 2; b2 <= 2u * Q && b2 <= h; ++b2) {
					const unsigned c = plain_cost(h - b2) + s2_cost(b2);
					if (c < best) { best = c; best_b = b2; }
				}
				if (best_b == h) {   // this pass IS the two-level one
					clo_timing_scope timing("bitonic_strided2", s);
					hipLaunchKernelGGL((clo_bitonic_strided2_kernel<E, Q, TBF, MODE>), dim3(tiles), dim3(1 << TBF), 0, s, data, stage, p, best_b, kd);
					++count;
					p -= best_b;
					continue;
				}
				ns = h - best_b;   // plain passes down to where the two-level pass (if any) starts
			}
			// Several plain passes: the SHORT one first (the order decides the strides they run at; 64 values per thread
			// 2 MiB apart — stage 25 of 2^26 uint32 as 6 + 5 steps — measured 143 us against 72 .. 86 for every other).
			if (ns > (unsigned) QS) ns = ns % QS ? ns % QS : QS;
			switch (ns) {
				case 1: launch_strided<E, 1, MODE>(data, n, stage, p, kd, s); break;
				case 2: launch_strided<E, 2, MODE>(data, n, stage, p, kd, s); break;
				case 3: launch_strided<E, 3, MODE>(data, n, stage, p, kd, s); break;
				case 4: launch_strided<E, 4, MODE>(data, n, stage, p, kd, s); break;
				case 5: launch_strided<E, 5, MODE>(data, n, stage, p, kd, s); break;
				default:
					if constexpr (QS >= 6) launch_strided<E, 6, MODE>(data, n, stage, p, kd, s);
					break;
			}
			++count;
			p -= ns;
		}
		{
			clo_timing_scope timing("bitonic_tile", s);
			// stage > kl only happens with full tiles (kl == KLF)
			if constexpr (MODE != 0) {   // (general keys: 256 registers and spills — their merge passes stay one tile wide)
				if (use_m2) hipLaunchKernelGGL((clo_bitonic_tile_merge2_kernel<E, Q, TBF, MODE>), dim3(tiles / 2u), dim3(1 << TBF), 0, s, data, stage, kd);
			}
			if (!use_m2) hipLaunchKernelGGL((clo_bitonic_tile_merge_kernel<E, Q, TBF, MODE>), dim3(tiles), dim3(1 << TBF), 0, s, data, stage, kd);
		}
		++count;
	}
	if (launches) *launches = count;
	return (int) hipGetLastError();
}

// Static LDS per work-group of the kernels tiled_run launches for `numel` elements (introspection:
// clo_sort_get_localmem_usage of sbitonic / abitonic) — the same three cases as above: below 2^Q elements the
// one-launch-per-step kernels (registers only); up to one tile the run-time-schedule kernel, whose array is sized for
// 256 threads x 2^Q values whatever the tile it is given; from 2^KLF elements on the compile-time-schedule kernels
// (presort, merge, two-level strided), all on tiles of 2^KLF elements. One padding slot per 32.
template <typename E>
size_t tiled_lds_bytes(size_t numel) {
	constexpr int Q = sizeof(E) == 8 ? 4 : 5;
	constexpr unsigned TBF = 9, KLF = TBF + Q;
	const size_t n = nlpo2(numel ? numel : 1);
	const unsigned T = log2u(n);
	if (T < (unsigned) Q) return 0;
	const size_t tile = T < KLF ? ((size_t) 256 << Q) : ((size_t) 1 << KLF);
	return (tile + tile / 32) * sizeof(E);
}

template <typename E>
int tiled_impl(void* vdata, size_t numel, int key_shift, int key_bits, int key_size, int key_kind, int descending,
	int* launches, hipStream_t s) {
	const bool identity = key_shift == 0 && key_bits == 8 * (int) sizeof(E) && key_size == (int) sizeof(E);
	// (TBF = 10 — 1024-thread groups on 2^15-element tiles, 132 KiB of LDS, one group per CU — was measured in round 4:
	// 24 launches instead of 27 at 2^26 uint32, but the pre-sort takes 0.400 instead of 0.292 ms and a merge pass 0.102
	// instead of 0.082: 2.562 against 2.557 ms per sort. One group per CU has nobody to run beside its barriers.)
	if (identity && key_kind == 0)
		return tiled_run<E, 1>(vdata, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
	if (identity && key_kind == 1)
		return tiled_run<E, 2>(vdata, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
	if constexpr (sizeof(E) >= 2) {
		if (identity && key_kind == 2)
			return tiled_run<E, 3>(vdata, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
	}
	return tiled_run<E, 0>(vdata, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
}

// ---------------------------------------------------------------------------
// gselect: global-memory selection (rank) sort, upstream's O(n^2) baseline
// sorter (sort/clo_sort_gselect.cl:38-58): element gid goes to position
//   #{ i : COMPARE(key_gid, key_i)  or  (key_i == key_gid and i < gid) }.
// Upstream's work-item reads all n keys from global memory; here a work-group
// stages 2048 ordered keys at a time in LDS and every thread walks the stage
// (all lanes read the same LDS words: a broadcast, no bank conflicts).
// Round 3: the ordered keys are 32-bit words whenever the key has at most 32 bits
// (K = unsigned), four of them per LDS read; and the tie rule is split by
// position — a stage that lies wholly before the work-group's elements counts
// `key_i <= key_gid`, one wholly after them `key_i < key_gid`, the same for every
// thread, so the inner loop is one compare and one add-with-carry per pair
// (v_cmp + v_addc) on four independent counters; only the stage that contains the
// work-group's own elements needs the per-thread index test, for those elements.
// The harness's table (profiles/r03_harness_sweep_gselect.txt): 2^16 uint keys 35 -> 131 Mkeys/s.
// ---------------------------------------------------------------------------
constexpr int GSEL_THREADS = 256;
constexpr int GSEL_STAGE = 2048;   // a multiple of GSEL_THREADS: a work-group's elements lie in ONE stage
static_assert(GSEL_STAGE % GSEL_THREADS == 0 && GSEL_THREADS % 4 == 0, "a work-group's elements lie in one stage, at a 16-byte boundary of its keys");

template <typename E>
__device__ __forceinline__ unsigned long long gsel_key(E e, const key_desc& kd) {
	unsigned long long raw = ((unsigned long long) e >> kd.shift) & kd.mask;
	if (kd.kind == 2 && raw == kd.signbit) e = (E) ((unsigned long long) e & ~(kd.signbit << kd.shift));  // -0 == +0, as upstream's float compare
	return okey<E>(e, kd);
}

// #{ i in [0, cnt) : s_key[i] before km } where "before" is <, <= (EQ), > or >= (DESC) — the same for all i
template <typename K, bool DESC, bool EQ>
__device__ __forceinline__ unsigned gsel_count_uniform(const K* s_key, unsigned cnt, K km) {
	constexpr int PER = 16 / (int) sizeof(K);
	typedef K vecK __attribute__((ext_vector_type(PER)));
	auto hit = [&](K k) __attribute__((always_inline)) -> unsigned {
		return DESC ? (EQ ? (unsigned) (k >= km) : (unsigned) (k > km)) : (EQ ? (unsigned) (k <= km) : (unsigned) (k < km));
	};
	unsigned c[4] = { 0u, 0u, 0u, 0u };
	const unsigned whole = cnt / (4 * PER) * (4 * PER);
	for (unsigned i = 0; i < whole; i += 4 * PER) {
		vecK x[4];
		#pragma unroll
		for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const vecK*>(s_key + i + q * PER);
		#pragma unroll
		for (int q = 0; q < 4; ++q) {
			#pragma unroll
			for (int e = 0; e < PER; ++e) c[q] += hit(x[q][e]);
		}
	}
	for (unsigned i = whole; i < cnt; ++i) c[0] += hit(s_key[i]);
	return c[0] + c[1] + c[2] + c[3];
}

template <typename E, typename K>
__global__ __launch_bounds__(GSEL_THREADS)
void clo_gselect_kernel(const E* __restrict__ in, E* __restrict__ out, size_t n, key_desc kd) {
	__shared__ __attribute__((aligned(16))) K s_key[GSEL_STAGE];
	const size_t wg0 = (size_t) blockIdx.x * GSEL_THREADS;   // the work-group's first element
	const size_t gid = wg0 + threadIdx.x;
	const E mine = gid < n ? in[gid] : (E) 0;
	const K km = (K) gsel_key<E>(mine, kd);
	size_t pos = 0;
	// the next stage's elements are requested before the current stage is counted (one wave per SIMD at 2^16
	// elements: nothing else would hide the loads)
	constexpr int MINE = GSEL_STAGE / GSEL_THREADS;
	E nxt[MINE];
	#pragma unroll
	for (int q = 0; q < MINE; ++q) { const size_t i = (size_t) q * GSEL_THREADS + threadIdx.x; nxt[q] = i < n ? in[i] : (E) 0; }
	for (size_t base = 0; base < n; base += GSEL_STAGE) {
		const unsigned cnt = n - base < (size_t) GSEL_STAGE ? (unsigned) (n - base) : (unsigned) GSEL_STAGE;
		__syncthreads();
		#pragma unroll
		for (int q = 0; q < MINE; ++q) s_key[q * GSEL_THREADS + threadIdx.x] = (K) gsel_key<E>(nxt[q], kd);   // (past cnt: never read)
		#pragma unroll
		for (int q = 0; q < MINE; ++q) { const size_t i = base + GSEL_STAGE + (size_t) q * GSEL_THREADS + threadIdx.x; nxt[q] = i < n ? in[i] : (E) 0; }
		__syncthreads();
		unsigned c;
		if (base + cnt <= wg0) {            // wholly before every element of the work-group: equal keys count
			c = kd.descending ? gsel_count_uniform<K, true, true>(s_key, cnt, km) : gsel_count_uniform<K, false, true>(s_key, cnt, km);
		} else if (base >= wg0 + GSEL_THREADS) {   // wholly after: equal keys do not
			c = kd.descending ? gsel_count_uniform<K, true, false>(s_key, cnt, km) : gsel_count_uniform<K, false, false>(s_key, cnt, km);
		} else {
			// the stage holds the work-group's own GSEL_THREADS elements (at a multiple of GSEL_THREADS): what lies
			// before them and after them as above, the index test for them alone
			const unsigned own = (unsigned) (wg0 - base);
			const unsigned own_end = own + GSEL_THREADS < cnt ? own + GSEL_THREADS : cnt;
			const unsigned before = threadIdx.x;   // (gid - base - own)
			c = 0;
			if (kd.descending) {
				c += gsel_count_uniform<K, true, true>(s_key, own, km);
				for (unsigned i = own; i < own_end; ++i) { const K k = s_key[i]; c += (k > km) | ((k == km) & (i - own < before)); }
				c += gsel_count_uniform<K, true, false>(s_key + own_end, cnt - own_end, km);
			} else {
				c += gsel_count_uniform<K, false, true>(s_key, own, km);
				for (unsigned i = own; i < own_end; ++i) { const K k = s_key[i]; c += (k < km) | ((k == km) & (i - own < before)); }
				c += gsel_count_uniform<K, false, false>(s_key + own_end, cnt - own_end, km);
			}
		}
		pos += c;
	}
	if (gid < n && pos < n) out[pos] = mine;
}

template <typename E>
int gselect_impl(const void* src, void* dst, size_t n, int key_shift, int key_bits, int key_size, int key_kind,
	int descending, hipStream_t s) {
	key_desc kd;
	E pad;
	const int st = make_desc<E>(key_shift, key_bits, key_size, key_kind, descending, &kd, &pad);
	if (st != 0) return st;
	clo_timing_scope timing("gselect", s);
	const dim3 grid((unsigned) ((n + GSEL_THREADS - 1) / GSEL_THREADS));
	if (key_bits <= 32)   // (the ordered key has as many bits as the key)
		hipLaunchKernelGGL((clo_gselect_kernel<E, unsigned>), grid, dim3(GSEL_THREADS), 0, s, (const E*) src, (E*) dst, n, kd);
	else
		hipLaunchKernelGGL((clo_gselect_kernel<E, unsigned long long>), grid, dim3(GSEL_THREADS), 0, s, (const E*) src, (E*) dst, n, kd);
	return (int) hipGetLastError();
}

}  // namespace
