/*
 * clo_oracle.c — CPU restatement of the cl_ops sort/scan hot path (see
 * clo_oracle.h for scope, citations and pin status: parity unpinned — the reference
 * holds no vectors for this path and cannot be run here). TEST INFRASTRUCTURE ONLY.
 *
 * Conventions: elements are handled as raw little-endian unsigned integers of
 * elem_size bytes (held in uint64_t). "ref:" comments name the upstream
 * file:line a block follows.
 */
#include "clo_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* raw element access                                                  */
/* ------------------------------------------------------------------ */

static inline uint64_t ld(const void* base, size_t i, int es) {
	const unsigned char* p = (const unsigned char*) base + i * (size_t) es;
	switch (es) {
		case 1: return *p;
		case 2: { uint16_t v; memcpy(&v, p, 2); return v; }
		case 4: { uint32_t v; memcpy(&v, p, 4); return v; }
		default: { uint64_t v; memcpy(&v, p, 8); return v; }
	}
}

static inline void st(void* base, size_t i, int es, uint64_t v) {
	unsigned char* p = (unsigned char*) base + i * (size_t) es;
	switch (es) {
		case 1: *p = (unsigned char) v; break;
		case 2: { uint16_t x = (uint16_t) v; memcpy(p, &x, 2); break; }
		case 4: { uint32_t x = (uint32_t) v; memcpy(p, &x, 4); break; }
		default: memcpy(p, &v, 8);
	}
}

static inline uint64_t size_mask(int bytes) {
	return bytes >= 8 ? ~0ULL : ((1ULL << (8 * bytes)) - 1ULL);
}

/* CLO_SORT_KEY_GET for the supported family: (KEY_TYPE)(elem >> key_shift). */
static inline uint64_t key_get(uint64_t elem, const clo_oracle_desc* d) {
	return (elem >> d->key_shift) & size_mask(d->key_size);
}

/* Typed "a > b" on raw key bits. */
static inline int key_gt(uint64_t a, uint64_t b, int key_size, int kind) {
	if (kind == CLO_ORACLE_KEY_SIGNED) {
		int sh = 64 - 8 * key_size;
		return ((int64_t) (a << sh) >> sh) > ((int64_t) (b << sh) >> sh);
	} else if (kind == CLO_ORACLE_KEY_FLOAT) {
		if (key_size == 4) {
			float x, y; uint32_t ua = (uint32_t) a, ub = (uint32_t) b;
			memcpy(&x, &ua, 4); memcpy(&y, &ub, 4);
			return x > y;
		} else {
			double x, y;
			memcpy(&x, &a, 8); memcpy(&y, &b, 8);
			return x > y;
		}
	}
	return a > b;
}

/* CLO_SORT_COMPARE(a,b): default "((a) > (b))"; descending variant "((a) < (b))". */
static inline int key_compare(uint64_t a, uint64_t b, const clo_oracle_desc* d) {
	return d->descending
		? key_gt(b, a, d->key_size, d->key_kind)
		: key_gt(a, b, d->key_size, d->key_kind);
}

/* ------------------------------------------------------------------ */
/* clo_common.c:141-199                                                */
/* ------------------------------------------------------------------ */

unsigned int clo_oracle_nlpo2(unsigned int x) {
	/* ref: clo_common.c:141-152 — smear the top bit down, add one. */
	if ((x & (x - 1)) == 0) return x;
	x |= x >> 1; x |= x >> 2; x |= x >> 4; x |= x >> 8; x |= x >> 16;
	return x + 1;
}

unsigned int clo_oracle_ones32(unsigned int x) {
	/* ref: clo_common.c:162-173 — population count. */
	unsigned int c = 0;
	while (x) { c += x & 1u; x >>= 1; }
	return c;
}

unsigned int clo_oracle_tzc(int x) {
	/* ref: clo_common.c:183-186 — ones32((x & -x) - 1). */
	return clo_oracle_ones32((unsigned int) ((x & -x) - 1));
}

void clo_oracle_suggest_worksizes(size_t real_ws, size_t dev_max_lws,
	size_t* gws, size_t* lws) {
	/* ref: SURVEY.md §8b (cf4ocl2 is not in the tree): lws <= min(user max,
	 * device max), halved until <= real_ws; without gws it must divide
	 * real_ws, with gws the latter is rounded up to a multiple of lws. */
	size_t l = (*lws != 0 && *lws < dev_max_lws) ? *lws : dev_max_lws;
	while (l > 1 && l > real_ws) l >>= 1;
	if (gws == NULL) {
		while (l > 1 && (real_ws % l) != 0) l >>= 1;
	} else {
		*gws = ((real_ws + l - 1) / l) * l;
	}
	*lws = l;
}

/* ------------------------------------------------------------------ */
/* bitonic building blocks                                             */
/* ------------------------------------------------------------------ */

/* ref: clo_sort_abitonic.cl:31-38 (ABIT_CMPXCH), clo_sort_sbitonic.cl:51-67. */
static inline void cmpxch(void* data, size_t i1, size_t i2, int desc,
	const clo_oracle_desc* d) {
	uint64_t e1 = ld(data, i1, d->elem_size), e2 = ld(data, i2, d->elem_size);
	int swap = key_compare(key_get(e1, d), key_get(e2, d), d) ^ desc;
	if (swap) { st(data, i1, d->elem_size, e2); st(data, i2, d->elem_size, e1); }
}

/* One (stage, step) layer over the whole array: sbitonic.cl:38-69 == abit_any
 * (abitonic.cl:573-603). */
/* `threads`: the work-items of a launch are independent (disjoint pairs / tiles), so the
 * CPU baseline runs them on all host cores; 1 = the serial loop, same result. */
static void layer_any(void* data, size_t n, unsigned stage, unsigned step,
	const clo_oracle_desc* d, int threads) {
	size_t stride = (size_t) 1 << (step - 1);
	#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
	for (size_t gid = 0; gid < n / 2; ++gid) {
		size_t i1 = (gid / stride) * stride * 2 + (gid % stride);
		int desc = (int) ((gid >> (stage - 1)) & 1);
		cmpxch(data, i1, i1 + stride, desc, d);
	}
}

void clo_oracle_sbitonic_mt(void* data, size_t numel, const clo_oracle_desc* d, int threads) {
	/* ref: clo_sort_sbitonic.c:73-80 (gws = nlpo2/2), :83 (stages = tzc(2*gws)),
	 * :102-118 (stage 1..T, step stage..1, one launch each). */
	size_t n = clo_oracle_nlpo2((unsigned int) numel);
	unsigned T = clo_oracle_tzc((int) n);
	if (threads <= 0) threads = omp_get_max_threads();
	for (unsigned stage = 1; stage <= T; ++stage)
		for (unsigned step = stage; step >= 1; --step)
			layer_any(data, n, stage, step, d, threads);
}

void clo_oracle_sbitonic(void* data, size_t numel, const clo_oracle_desc* d) {
	clo_oracle_sbitonic_mt(data, numel, d, 1);
}

/* Typed "a == b" on raw key bits (the kernel's key_i == key_gid). */
static inline int key_eq(uint64_t a, uint64_t b, int key_size, int kind) {
	if (kind == CLO_ORACLE_KEY_FLOAT) {
		if (key_size == 4) {
			float x, y; uint32_t ua = (uint32_t) a, ub = (uint32_t) b;
			memcpy(&x, &ua, 4); memcpy(&y, &ub, 4);
			return x == y;
		} else if (key_size == 8) {
			double x, y;
			memcpy(&x, &a, 8); memcpy(&y, &b, 8);
			return x == y;
		}
	}
	return a == b;
}

void clo_oracle_gselect(const void* data_in, void* data_out, size_t numel, const clo_oracle_desc* d) {
	/* ref: clo_sort_gselect.cl:38-58 — one work-item per element (gws = numel
	 * rounded up, the kernel guards gid < size): position = number of elements
	 * that compare before it, ties broken by index. clo_sort_gselect.c:105-116
	 * launches it once. */
	for (size_t gid = 0; gid < numel; ++gid) {
		uint64_t e = ld(data_in, gid, d->elem_size);
		uint64_t kg = key_get(e, d);
		size_t pos = 0;
		for (size_t i = 0; i < numel; ++i) {
			uint64_t ki = key_get(ld(data_in, i, d->elem_size), d);
			if (key_compare(kg, ki, d) || (key_eq(ki, kg, d->key_size, d->key_kind) && i < gid)) ++pos;
		}
		st(data_out, pos, d->elem_size, e);
	}
}

/* Register network of the priv/hyb kernels: V = 2^S values, strides V/2 .. 1.
 * ref: clo_sort_abitonic.cl:163-224 (ABIT_SORT_4S16V / 3S8V / 2S4V). The values
 * sit at data[base + j*inc]. */
static void priv_network(void* data, size_t base, size_t inc, unsigned S, int desc,
	const clo_oracle_desc* d) {
	unsigned V = 1u << S;
	for (unsigned half = V / 2; half >= 1; half /= 2)
		for (unsigned j = 0; j < V; ++j)
			if ((j & half) == 0)
				cmpxch(data, base + (size_t) j * inc, base + (size_t) (j + half) * inc, desc, d);
}

/* abit_priv_{S}s{V}v launched with "step" p. ref: abitonic.cl:147-161. */
static void kernel_priv(void* data, size_t n, unsigned stage, unsigned p, unsigned S,
	const clo_oracle_desc* d, int threads) {
	size_t V = (size_t) 1 << S, block = (size_t) 1 << p, inc = block / V;
	#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
	for (size_t gid = 0; gid < n / V; ++gid) {
		int desc = (int) (((gid * V) >> stage) & 1);
		size_t base = ((gid * V) / block) * block + (gid % inc);
		priv_network(data, base, inc, S, desc, d);
	}
}

/* abit_local_sK: steps K..1 inside tiles of 2L. ref: abitonic.cl:40-47,118-145. */
static void kernel_local(void* data, size_t n, unsigned stage, unsigned K, size_t L,
	const clo_oracle_desc* d, int threads) {
	size_t tile = 2 * L;
	#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
	for (size_t w = 0; w < n / tile; ++w) {
		unsigned char* t = (unsigned char*) data + w * tile * (size_t) d->elem_size;
		for (unsigned q = K; q >= 1; --q) {
			size_t stride = (size_t) 1 << (q - 1);
			for (size_t lid = 0; lid < L; ++lid) {
				size_t gid = w * L + lid;
				int desc = (int) ((gid >> (stage - 1)) & 1);
				size_t i1 = (lid / stride) * stride * 2 + (lid % stride);
				cmpxch(t, i1, i1 + stride, desc, d);
			}
		}
	}
}

/* abit_hyb_sK_{S}s{V}v: tile of V*L, register networks for q = K, K-S, .., S.
 * ref: abitonic.cl:683-721 (2s4v), :824-870 (3s8v), :965-1028 (4s16v). */
static void kernel_hyb(void* data, size_t n, unsigned stage, unsigned K, unsigned S,
	size_t L, const clo_oracle_desc* d, int threads) {
	size_t V = (size_t) 1 << S, tile = V * L;
	#pragma omp parallel for num_threads(threads) schedule(static) if (threads > 1)
	for (size_t w = 0; w < n / tile; ++w) {
		unsigned char* t = (unsigned char*) data + w * tile * (size_t) d->elem_size;
		for (unsigned q = K; q >= S; q -= S) {
			size_t block = (size_t) 1 << q, inc = block / V;
			for (size_t lid = 0; lid < L; ++lid) {
				size_t gid = w * L + lid;
				int desc = (int) ((gid >> (stage - S)) & 1);
				size_t laddr = ((lid * V) / block) * block + (lid % inc);
				priv_network(t, laddr, inc, S, desc, d);
			}
			if (q < 2 * S) break; /* unsigned guard */
		}
	}
}

typedef struct {
	int kind;        /* 0 any, 1 local, 2 priv, 3 hyb */
	unsigned K;      /* local/hyb: first step handled */
	unsigned S;      /* priv/hyb: steps per register network */
	size_t gws, lws;
	int set_step;
	unsigned num_steps;
} abit_step;

int clo_oracle_abitonic_mt(void* data, size_t numel, const clo_oracle_desc* d,
	size_t lws_max, size_t dev_max_lws,
	unsigned minps, unsigned maxps, unsigned maxsfs, int threads) {
	if (threads <= 0) threads = omp_get_max_threads();

	/* ref: clo_sort_abitonic.c:66-133 — candidate kernels per "stage finish"
	 * step 2..12, in preference order, as (kind, S). kind 1 = local (S=1). */
	static const struct { int kind; unsigned S; } lookup[11][4] = {
		/* 2 */ {{1,1},{0,0},{0,0},{0,0}},
		/* 3 */ {{3,3},{1,1},{0,0},{0,0}},
		/* 4 */ {{3,4},{3,2},{1,1},{0,0}},
		/* 5 */ {{1,1},{0,0},{0,0},{0,0}},
		/* 6 */ {{3,3},{3,2},{1,1},{0,0}},
		/* 7 */ {{1,1},{0,0},{0,0},{0,0}},
		/* 8 */ {{3,4},{3,2},{1,1},{0,0}},
		/* 9 */ {{3,3},{1,1},{0,0},{0,0}},
		/* 10 */ {{3,2},{1,1},{0,0},{0,0}},
		/* 11 */ {{1,1},{0,0},{0,0},{0,0}},
		/* 12 */ {{3,4},{3,3},{3,2},{0,0}},
	};

	size_t n = clo_oracle_nlpo2((unsigned int) numel);
	unsigned T = clo_oracle_tzc((int) n);
	if (T == 0) return 0;
	abit_step* steps = (abit_step*) calloc(T, sizeof(abit_step));

	/* ref: abitonic.c:145-154 — lws for "private" kernels, effective sfs. */
	size_t big_gws = (size_t) 1 << 20, lws_max_sfs = lws_max;
	clo_oracle_suggest_worksizes(big_gws, dev_max_lws, NULL, &lws_max_sfs);
	unsigned sfs = maxsfs < 12 ? maxsfs : 12;
	unsigned lim = clo_oracle_tzc((int) lws_max_sfs) + maxps;
	if (lim < sfs) sfs = lim;

	for (unsigned step = 1; step <= T; ++step) {
		abit_step* s = &steps[step - 1];
		if (step == 1) {
			/* ref: abitonic.c:158-174 */
			s->kind = 0; s->gws = n / 2; s->lws = lws_max;
			clo_oracle_suggest_worksizes(s->gws, dev_max_lws, NULL, &s->lws);
			s->set_step = 1; s->num_steps = 1;
		} else if (step > sfs) {
			/* ref: abitonic.c:175-232 — priv kernel advancing min(step,maxps). */
			unsigned m = step < maxps ? step : maxps;
			s->kind = (m == 1) ? 0 : 2; s->S = m;
			s->gws = n >> m;
			s->lws = lws_max_sfs < s->gws ? lws_max_sfs : s->gws;
			s->set_step = 1; s->num_steps = m;
		} else {
			/* ref: abitonic.c:233-300 — first candidate whose private-step count
			 * is within [minps,maxps] and whose lws covers 2^(step-S). */
			int found = 0;
			for (unsigned i = 0; i < 4 && lookup[step - 2][i].kind != 0; ++i) {
				unsigned S = lookup[step - 2][i].S;
				size_t gws = n >> S, lws = lws_max;
				clo_oracle_suggest_worksizes(gws, dev_max_lws, NULL, &lws);
				if (S <= maxps && S >= minps && lws >= ((size_t) 1 << (step - S))) {
					s->kind = lookup[step - 2][i].kind; s->S = S; s->K = step;
					s->gws = gws; s->lws = lws;
					s->set_step = 0; s->num_steps = step;
					found = 1;
					break;
				}
			}
			if (!found) {
				s->kind = 0; s->gws = n / 2; s->lws = lws_max;
				clo_oracle_suggest_worksizes(s->gws, dev_max_lws, NULL, &s->lws);
				s->set_step = 1; s->num_steps = 1;
			}
		}
	}

	/* ref: abitonic.c:401-432 — stage loop; step decreases by num_steps. */
	int launches = 0;
	for (unsigned stage = 1; stage <= T; ++stage) {
		for (unsigned step = stage; step >= 1; ) {
			const abit_step* s = &steps[step - 1];
			switch (s->kind) {
				case 0: layer_any(data, n, stage, step, d, threads); break;
				case 1: kernel_local(data, n, stage, s->K, s->lws, d, threads); break;
				case 2: kernel_priv(data, n, stage, step, s->S, d, threads); break;
				default: kernel_hyb(data, n, stage, s->K, s->S, s->lws, d, threads); break;
			}
			++launches;
			if (s->num_steps >= step) break;
			step -= s->num_steps;
		}
	}
	free(steps);
	return launches;
}

int clo_oracle_abitonic(void* data, size_t numel, const clo_oracle_desc* d,
	size_t lws_max, size_t dev_max_lws,
	unsigned minps, unsigned maxps, unsigned maxsfs) {
	return clo_oracle_abitonic_mt(data, numel, d, lws_max, dev_max_lws, minps, maxps, maxsfs, 1);
}

/* ------------------------------------------------------------------ */
/* Blelloch scan                                                       */
/* ------------------------------------------------------------------ */

void clo_oracle_serial_scan(const void* data_in, void* data_out, size_t numel,
	int elem_size, int sum_size) {
	/* ref: clo_scan_bench.c:252-271 — running sum compared element-wise. */
	uint64_t acc = 0, m = size_mask(sum_size);
	for (size_t i = 0; i < numel; ++i) {
		st(data_out, i, sum_size, acc & m);
		acc = (acc + ld(data_in, i, elem_size)) & m;
	}
}

/* Exclusive scan of a block of `len` sums held in `aux` (LDS stand-in).
 * ref: clo_scan_blelloch.cl:82-117 — up-sweep, clear last, down-sweep. The
 * tree order is kept so that wrap-around behaves exactly as upstream (it is
 * associative anyway). Returns the block total. */
static uint64_t block_tree_scan(uint64_t* aux, size_t len, uint64_t m) {
	size_t offset = 1;
	for (size_t dd = len >> 1; dd > 0; dd >>= 1) {
		for (size_t lid = 0; lid < dd; ++lid) {
			size_t ai = offset * (2 * lid + 1) - 1, bi = offset * (2 * lid + 2) - 1;
			aux[bi] = (aux[bi] + aux[ai]) & m;
		}
		offset *= 2;
	}
	uint64_t total = aux[len - 1];
	aux[len - 1] = 0;
	for (size_t dd = 1; dd < len; dd *= 2) {
		offset >>= 1;
		for (size_t lid = 0; lid < dd; ++lid) {
			size_t ai = offset * (2 * lid + 1) - 1, bi = offset * (2 * lid + 2) - 1;
			uint64_t t = aux[ai];
			aux[ai] = aux[bi];
			aux[bi] = (aux[bi] + t) & m;
		}
	}
	return total;
}

static int blelloch_impl(const void* data_in, void* data_out, size_t numel,
	int elem_size, int sum_size, size_t lws_max, size_t dev_max_lws, int threads) {

	uint64_t m = size_mask(sum_size);
	/* ref: clo_scan_blelloch.c:129-141 */
	size_t lws = lws_max, realws = numel / 2, gws;
	if (realws == 0) return 0;
	clo_oracle_suggest_worksizes(realws, dev_max_lws, &gws, &lws);
	if (gws > lws * lws) gws = lws * lws;
	size_t num_wgs = gws / lws;
	size_t block = 2 * lws;
	size_t bpw = (realws + gws - 1) / gws;
	size_t nblocks = numel / block;
	uint64_t* wgsum = (uint64_t*) calloc(num_wgs ? num_wgs : 1, sizeof(uint64_t));
	int launches = 1;
	(void) threads;

	/* Kernel 1 — workgroupScan. ref: clo_scan_blelloch.cl:49-126. */
	#pragma omp parallel for num_threads(threads) schedule(static)
	for (size_t wg = 0; wg < num_wgs; ++wg) {
		uint64_t* aux = (uint64_t*) malloc(block * sizeof(uint64_t));
		uint64_t in_sum = 0;
		for (size_t b = 0; b < bpw && (wg * bpw + b) < nblocks; ++b) {
			size_t g0 = (bpw * wg + b) * block;
			for (size_t i = 0; i < block; ++i) aux[i] = ld(data_in, g0 + i, elem_size) & m;
			uint64_t prev = in_sum;
			in_sum = (in_sum + block_tree_scan(aux, block, m)) & m;
			for (size_t i = 0; i < block; ++i) st(data_out, g0 + i, sum_size, (aux[i] + prev) & m);
		}
		wgsum[wg] = in_sum;
		free(aux);
	}

	if (gws > lws) {
		/* Kernel 2 — workgroupSumsScan over the num_wgs totals, one WG.
		 * ref: clo_scan_blelloch.cl:134-182, launch at blelloch.c:176-181. */
		block_tree_scan(wgsum, num_wgs, m);
		/* Kernel 3 — addWorkgroupSums. ref: clo_scan_blelloch.cl:193-211; the
		 * upstream kernel has no bound on gid, here gid < numel is enforced. */
		#pragma omp parallel for num_threads(threads) schedule(static)
		for (size_t grp = 0; grp < (numel + lws - 1) / lws; ++grp) {
			uint64_t add = wgsum[grp / (2 * bpw)];
			for (size_t lid = 0; lid < lws; ++lid) {
				size_t gid = grp * lws + lid;
				if (gid < numel) st(data_out, gid, sum_size, (ld(data_out, gid, sum_size) + add) & m);
			}
		}
		launches = 3;
	}
	free(wgsum);
	return launches;
}

int clo_oracle_blelloch(const void* data_in, void* data_out, size_t numel,
	int elem_size, int sum_size, size_t lws_max, size_t dev_max_lws) {
	return blelloch_impl(data_in, data_out, numel, elem_size, sum_size, lws_max, dev_max_lws, 1);
}

int clo_oracle_blelloch_mt(const void* data_in, void* data_out, size_t numel,
	int elem_size, int sum_size, size_t lws, int threads) {
#ifdef _OPENMP
	if (threads <= 0) threads = omp_get_max_threads();
#else
	threads = 1;
#endif
	blelloch_impl(data_in, data_out, numel, elem_size, sum_size, lws, lws, threads);
	return threads;
}

/* ------------------------------------------------------------------ */
/* SatRadix                                                            */
/* ------------------------------------------------------------------ */

/* OpenCL C "key >> b": the count is taken modulo the width of the promoted
 * left operand (32 for <=4-byte types, 64 for 8-byte); narrow signed keys are
 * sign-extended by the integer promotion. */
static inline uint64_t ocl_shr(uint64_t key, unsigned b, const clo_oracle_desc* d) {
	if (d->key_size == 8) {
		b &= 63u;
		return (d->key_kind == CLO_ORACLE_KEY_SIGNED)
			? (uint64_t) ((int64_t) key >> b) : key >> b;
	} else {
		uint32_t k32;
		if (d->key_kind == CLO_ORACLE_KEY_SIGNED && d->key_size < 4) {
			int sh = 32 - 8 * d->key_size;
			k32 = (uint32_t) ((int32_t) ((uint32_t) key << sh) >> sh);
		} else {
			k32 = (uint32_t) key;
		}
		b &= 31u;
		return (d->key_kind == CLO_ORACLE_KEY_SIGNED)
			? (uint64_t) (uint32_t) ((int32_t) k32 >> b) : (uint64_t) (k32 >> b);
	}
}

static int satradix_impl(void* data, size_t numel, const clo_oracle_desc* d,
	unsigned radix, size_t lws_max, size_t dev_max_lws, int threads,
	uint32_t* dbg_offsets, uint32_t* dbg_counters, uint32_t* dbg_counters_sum) {

	if (radix < 2 || clo_oracle_ones32(radix) != 1) return -1; /* ref: satradix.c:385-392 */
	const int es = d->elem_size;
	/* ref: clo_sort_satradix.c:166-169 */
	unsigned bits = clo_oracle_tzc((int) radix);
	unsigned total_digits = (unsigned) (es * 8) / bits;
	/* ref: :184-197 */
	size_t n = clo_oracle_nlpo2((unsigned int) numel);
	size_t L = lws_max;
	clo_oracle_suggest_worksizes(n, dev_max_lws, NULL, &L);
	if (L < radix) L = radix;
	size_t num_wgs = n / L + n % L;
	size_t array_len = n / num_wgs;
	if (array_len != L || n % L) return -2; /* launch shape upstream assumes */
	size_t naux = num_wgs * radix;

	unsigned char* aux = (unsigned char*) malloc(n * (size_t) es);
	uint32_t* offsets = (uint32_t*) malloc(naux * sizeof(uint32_t));
	uint32_t* counters = (uint32_t*) malloc(naux * sizeof(uint32_t));
	uint32_t* counters_sum = (uint32_t*) malloc(naux * sizeof(uint32_t));
	(void) threads;

	for (unsigned pass = 0; pass < total_digits; ++pass) {
		unsigned start_bit = pass * bits;

		#pragma omp parallel num_threads(threads)
		{
			uint64_t* tile = (uint64_t*) malloc(L * sizeof(uint64_t));
			uint64_t* tmp = (uint64_t*) malloc(L * sizeof(uint64_t));
			uint32_t* dig = (uint32_t*) malloc(L * sizeof(uint32_t));
			uint32_t* off = (uint32_t*) malloc(radix * sizeof(uint32_t));

			#pragma omp for schedule(static)
			for (size_t wg = 0; wg < num_wgs; ++wg) {
				/* ---- satradix_localsort, ref: satradix.cl:34-123 ---- */
				for (size_t j = 0; j < L; ++j) tile[j] = ld(data, wg * L + j, es);
				for (unsigned b = start_bit; b < start_bit + bits; ++b) {
					/* f = !bit; e = exclusive scan(f); Z = total zeros;
					 * pos = bit ? j - e + Z : e   (satradix.cl:58-118) */
					size_t Z = 0;
					for (size_t j = 0; j < L; ++j)
						Z += !(ocl_shr(key_get(tile[j], d), b, d) & 1u);
					size_t e = 0;
					for (size_t j = 0; j < L; ++j) {
						unsigned bit = (unsigned) (ocl_shr(key_get(tile[j], d), b, d) & 1u);
						size_t pos = bit ? (j - e + Z) : e;
						tmp[pos] = tile[j];
						e += !bit;
					}
					uint64_t* sw = tile; tile = tmp; tmp = sw;
				}
				for (size_t j = 0; j < L; ++j) st(aux, wg * L + j, es, tile[j]);

				/* ---- satradix_histogram, ref: satradix.cl:125-222 ---- */
				for (size_t j = 0; j < array_len; ++j)
					dig[j] = (uint32_t) (ocl_shr(key_get(tile[j], d), start_bit, d) & (radix - 1));
				for (unsigned r = 0; r < radix; ++r) off[r] = UINT_MAX;
				/* region starts (:152-160); lid 0 takes the else branch. */
				off[dig[0]] = 0;
				for (size_t j = 1; j < array_len; ++j)
					if (dig[j] != dig[j - 1]) off[dig[j]] = (uint32_t) j;
				/* last offset defaults to array_len (:165-170) */
				if (off[radix - 1] == UINT_MAX) off[radix - 1] = (uint32_t) array_len;
				/* leading unset offsets become 0 (:172-180) */
				for (unsigned r = 0; r < radix && off[r] == UINT_MAX; ++r) off[r] = 0;
				/* each set offset > 0 back-fills the unset run below it (:185-201) */
				for (unsigned r = 1; r < radix; ++r) {
					if (off[r] > 0 && off[r] != UINT_MAX && off[r - 1] == UINT_MAX) {
						uint32_t cur = off[r];
						for (unsigned i = r - 1; i > 0 && off[i] == UINT_MAX; --i) off[i] = cur;
					}
				}
				/* counts (:207-213) and digit-major store (:217-220) */
				for (unsigned r = 0; r < radix; ++r) {
					uint32_t c = (r < radix - 1) ? off[r + 1] - off[r]
						: (uint32_t) array_len - off[r];
					offsets[(size_t) radix * wg + r] = off[r];
					counters[num_wgs * r + wg] = c;
				}
			}
			free(tile); free(tmp); free(dig); free(off);
		}

		/* ---- scan of the counters, ref: satradix.c:298-299 ---- */
		blelloch_impl(counters, counters_sum, naux, 4, 4, lws_max, dev_max_lws, threads);

		if (pass == 0) {
			if (dbg_offsets) memcpy(dbg_offsets, offsets, naux * sizeof(uint32_t));
			if (dbg_counters) memcpy(dbg_counters, counters, naux * sizeof(uint32_t));
			if (dbg_counters_sum) memcpy(dbg_counters_sum, counters_sum, naux * sizeof(uint32_t));
		}

		/* ---- satradix_scatter, ref: satradix.cl:224-258 ---- */
		#pragma omp parallel for num_threads(threads) schedule(static)
		for (size_t wg = 0; wg < num_wgs; ++wg) {
			for (size_t lid = 0; lid < L; ++lid) {
				uint64_t e = ld(aux, wg * L + lid, es);
				uint32_t digit = (uint32_t) (ocl_shr(key_get(e, d), start_bit, d) & (radix - 1));
				size_t out = (size_t) counters_sum[num_wgs * digit + wg] + lid
					- offsets[(size_t) radix * wg + digit];
				st(data, out, es, e);
			}
		}
	}

	free(aux); free(offsets); free(counters); free(counters_sum);
	return (int) total_digits;
}

int clo_oracle_satradix(void* data, size_t numel, const clo_oracle_desc* d,
	unsigned radix, size_t lws_max, size_t dev_max_lws,
	uint32_t* dbg_offsets, uint32_t* dbg_counters, uint32_t* dbg_counters_sum) {
	return satradix_impl(data, numel, d, radix, lws_max, dev_max_lws, 1,
		dbg_offsets, dbg_counters, dbg_counters_sum);
}

int clo_oracle_satradix_mt(void* data, size_t numel, const clo_oracle_desc* d,
	unsigned radix, size_t lws, int threads) {
#ifdef _OPENMP
	if (threads <= 0) threads = omp_get_max_threads();
#else
	threads = 1;
#endif
	int r = satradix_impl(data, numel, d, radix, lws, lws, threads, NULL, NULL, NULL);
	return r < 0 ? r : threads;
}

/* ------------------------------------------------------------------ */
/* the reference's own checks + independent references                 */
/* ------------------------------------------------------------------ */

long clo_oracle_check_sorted(const void* data, size_t numel, int elem_size, int kind) {
	/* ref: clo_sort_bench.c:216-226 with clo_bench.c:26-65 (typed a > b). */
	for (size_t i = 0; i + 1 < numel; ++i)
		if (key_gt(ld(data, i, elem_size), ld(data, i + 1, elem_size), elem_size, kind))
			return (long) i;
	return -1;
}

static void merge_sort(uint64_t* a, uint64_t* tmp, size_t n, const clo_oracle_desc* d) {
	if (n < 2) return;
	size_t h = n / 2;
	merge_sort(a, tmp, h, d);
	merge_sort(a + h, tmp, n - h, d);
	size_t i = 0, j = h, k = 0;
	while (i < h && j < n) {
		/* take right only if strictly "before" left: stable */
		if (key_compare(key_get(a[i], d), key_get(a[j], d), d)) tmp[k++] = a[j++];
		else tmp[k++] = a[i++];
	}
	while (i < h) tmp[k++] = a[i++];
	while (j < n) tmp[k++] = a[j++];
	memcpy(a, tmp, n * sizeof(uint64_t));
}

void clo_oracle_stable_sort(void* data, size_t numel, const clo_oracle_desc* d) {
	uint64_t* a = (uint64_t*) malloc(numel * sizeof(uint64_t));
	uint64_t* t = (uint64_t*) malloc(numel * sizeof(uint64_t));
	for (size_t i = 0; i < numel; ++i) a[i] = ld(data, i, d->elem_size);
	merge_sort(a, t, numel, d);
	for (size_t i = 0; i < numel; ++i) st(data, i, d->elem_size, a[i]);
	free(a); free(t);
}

/* ------------------------------------------------------------------ */
/* benchmark input distributions (GRand = MT19937)                     */
/* ------------------------------------------------------------------ */

typedef struct { uint32_t mt[624]; int idx; } grand;

static void grand_seed(grand* g, uint32_t seed) {
	/* GLib g_rand_set_seed (MT19937 init_genrand, 2002 version). */
	g->mt[0] = seed;
	for (int i = 1; i < 624; ++i)
		g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t) i;
	g->idx = 624;
}

static uint32_t grand_int(grand* g) {
	if (g->idx >= 624) {
		for (int k = 0; k < 624; ++k) {
			uint32_t y = (g->mt[k] & 0x80000000u) | (g->mt[(k + 1) % 624] & 0x7fffffffu);
			g->mt[k] = g->mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
		}
		g->idx = 0;
	}
	uint32_t y = g->mt[g->idx++];
	y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
	return y;
}

static double grand_double(grand* g) {
	/* GLib g_rand_double: two draws, [0,1). */
	const double t = 2.3283064365386962890625e-10;
	double r = grand_int(g) * t;
	r = (r + grand_int(g)) * t;
	if (r >= 1.0) return grand_double(g);
	return r;
}

static int32_t grand_int_range(grand* g, int32_t begin, int32_t end) {
	/* GLib g_rand_int_range (2.x "new" algorithm): rejection on dist. */
	uint32_t dist = (uint32_t) end - (uint32_t) begin, random = 0;
	if (dist == 0) return begin;
	uint32_t maxvalue;
	if (dist <= 0x80000000u) {
		uint32_t leftover = (0x80000000u % dist) * 2;
		if (leftover >= dist) leftover -= dist;
		maxvalue = 0xffffffffu - leftover;
	} else {
		maxvalue = dist - 1;
	}
	do random = grand_int(g); while (random > maxvalue);
	random %= dist;
	return begin + (int32_t) random;
}

void clo_oracle_bench_rand(uint32_t seed, int clo_type, void* out, size_t numel) {
	/* ref: clo_bench.c:67-142. Types: CloType numbering. */
	grand g; grand_seed(&g, seed);
	for (size_t i = 0; i < numel; ++i) {
		switch (clo_type) {
			case 0: st(out, i, 1, (uint64_t) (int8_t) grand_int_range(&g, -128, 127)); break;
			case 1: st(out, i, 1, (uint64_t) grand_int_range(&g, 0, 255)); break;
			case 2: st(out, i, 2, (uint64_t) (int16_t) grand_int_range(&g, -32768, 32767)); break;
			case 3: st(out, i, 2, (uint64_t) grand_int_range(&g, 0, 65535)); break;
			case 4: st(out, i, 4, (uint64_t) (uint32_t) grand_int_range(&g, INT32_MIN, INT32_MAX)); break;
			case 5: st(out, i, 4, (uint64_t) (uint32_t) (grand_double(&g) * 4294967295.0)); break;
			case 6: {
				double u = grand_double(&g);
				int neg = (grand_int(&g) & (1u << 15)) != 0; /* g_rand_boolean */
				int64_t v = (int64_t) (u * (neg ? (double) INT64_MIN : (double) INT64_MAX));
				st(out, i, 8, (uint64_t) v);
				break;
			}
			case 7: {
				double v = grand_double(&g) * 18446744073709551615.0;
				st(out, i, 8, v >= 18446744073709551615.0 ? ~0ULL : (uint64_t) v);
				break;
			}
			default: st(out, i, 4, 0); break; /* half/float/double: not on the path */
		}
	}
}

void clo_oracle_scan_bench_rand(uint32_t seed, int elem_size, void* out, size_t numel) {
	/* ref: clo_scan_bench.c:219-223 — (gulong)(g_rand_double * 128), low bytes kept. */
	grand g; grand_seed(&g, seed);
	for (size_t i = 0; i < numel; ++i)
		st(out, i, elem_size, (uint64_t) (grand_double(&g) * 128));
}
