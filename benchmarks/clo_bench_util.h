/*
 * clo_bench_util.h — helpers of the two command-line harnesses: a GLib-GRand
 * compatible generator (MT19937; GLib is not available in this image) and the
 * value distributions of the reference harness (benchmarks/clo_bench.c:67-142,
 * benchmarks/clo_scan_bench.c:219-223 upstream), typed element comparison
 * (clo_bench.c:26-65).
 */
#ifndef CLO_BENCH_UTIL_H
#define CLO_BENCH_UTIL_H

#include <stdint.h>
#include <cl_ops.h>

typedef struct {
	uint32_t mt[624];
	int idx;
} CloBenchRand;

void clo_bench_rand_seed(CloBenchRand* r, uint32_t seed);   /* g_rand_new_with_seed */
uint32_t clo_bench_rand_int(CloBenchRand* r);               /* g_rand_int */
double clo_bench_rand_double(CloBenchRand* r);              /* g_rand_double, [0,1) */
int32_t clo_bench_rand_int_range(CloBenchRand* r, int32_t begin, int32_t end); /* g_rand_int_range */

/* One random element of `type` at `location` (clo_bench.c:67-142 distributions). */
void clo_bench_rand(CloBenchRand* r, CloType type, void* location);

/* <0, 0, >0 like the typed comparison of clo_bench.c:26-65. */
int clo_bench_compare(CloType type, const void* a, const void* b);

#endif
