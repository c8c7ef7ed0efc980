/*
 * clo_oracle.h — CPU restatement of the cl_ops sort/scan hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE. Only tests/, the smoke test in
 * __graft_entry__.py and bench.py's cpu_baseline leg may load it; the shipped
 * library (cl_ops_amd/lib/libcl_ops_hip.so) never links or calls anything here.
 *
 * Every function restates, in plain C, the algorithm of the reference file:line
 * it cites (paths relative to the upstream cl_ops tree, src/cl_ops/...). Nothing
 * is copied: the reference kernels are OpenCL C executed one work-item per
 * element; here each kernel is restated as loops over work-groups/work-items
 * with the barriers turned into loop boundaries.
 *
 * Pin status: PARITY UNPINNED. The upstream tree holds NO golden vectors or unit tests for
 * sort/scan (only src/tests/test_rng.c). Its only known-answer checks for this
 * path are the benchmark self-checks (clo_sort_bench.c:211-226 adjacent-pair
 * order, clo_scan_bench.c:252-271 serial exclusive scan), restated below as
 * clo_oracle_check_sorted / clo_oracle_serial_scan; tests/test_oracle.py runs
 * every oracle algorithm through them and through independent stable-sort /
 * serial-scan references. The reference itself is unbuildable here (needs
 * cf4ocl2, GLib dev files and an OpenCL CPU device; none present) so the oracle
 * is NOT pinned against executed reference output.
 */
#ifndef CLO_ORACLE_H
#define CLO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Key interpretation for comparison sorts (CLO_SORT_COMPARE works on typed keys). */
enum { CLO_ORACLE_KEY_UNSIGNED = 0, CLO_ORACLE_KEY_SIGNED = 1, CLO_ORACLE_KEY_FLOAT = 2 };

/* Describes CLO_SORT_ELEM_TYPE / CLO_SORT_KEY_TYPE / CLO_SORT_KEY_GET / CLO_SORT_COMPARE
 * (clo_sort_abstract.c:144-168) for the fixed family this build supports:
 * key = (KEY_TYPE)(elem >> key_shift), compare = a > b (ascending) or a < b. */
typedef struct {
	int elem_size;   /* 1,2,4,8 bytes */
	int key_size;    /* 1,2,4,8 bytes */
	int key_shift;   /* bits */
	int key_kind;    /* CLO_ORACLE_KEY_* */
	int descending;  /* 0: "((a) > (b))" default; 1: "((a) < (b))" */
} clo_oracle_desc;

/* clo_common.c:141-199 */
unsigned int clo_oracle_nlpo2(unsigned int x);
unsigned int clo_oracle_ones32(unsigned int x);
unsigned int clo_oracle_tzc(int x);

/* cf4ocl2 ccl_kernel_suggest_worksizes contract as restated in SURVEY.md §8b.
 * gws may be NULL (then lws must divide real_ws). dev_max_lws: device limit. */
void clo_oracle_suggest_worksizes(size_t real_ws, size_t dev_max_lws,
	size_t* gws, size_t* lws /* in: user max (0 = none), out: chosen */);

/* clo_sort_sbitonic.c:73-118 + clo_sort_sbitonic.cl:38-69. In place; numel must
 * be a power of two (reference kernels have no bounds). */
void clo_oracle_sbitonic(void* data, size_t numel, const clo_oracle_desc* d);

/* clo_sort_gselect.c:60-118 + clo_sort_gselect.cl:38-58: O(n^2) rank sort, out of
 * place (data_out != data_in), any numel. */
void clo_oracle_gselect(const void* data_in, void* data_out, size_t numel, const clo_oracle_desc* d);

/* clo_sort_abitonic.c:58-313 (strategy), :401-432 (stage/step loop) and
 * clo_sort_abitonic.cl (any / local_sK / priv_SsVv / hyb_sK_SsVv index rules).
 * Returns the number of kernel launches (global round trips) the strategy
 * makes, i.e. SURVEY §8d's G. numel power of two. */
int clo_oracle_abitonic(void* data, size_t numel, const clo_oracle_desc* d,
	size_t lws_max, size_t dev_max_lws,
	unsigned minps, unsigned maxps, unsigned maxsfs);

/* clo_sort_satradix.c:166-197,264-313 + clo_sort_satradix.cl:34-258, with the
 * scan step done by clo_oracle_blelloch (clo_sort_satradix.c:298). In place.
 * numel power of two. If dbg_offsets/dbg_counters/dbg_counters_sum are non-NULL
 * they receive the three aux arrays of the FIRST digit pass (num_wgs*radix
 * uints each). Returns number of digit passes, <0 on bad arguments. */
int clo_oracle_satradix(void* data, size_t numel, const clo_oracle_desc* d,
	unsigned radix, size_t lws_max, size_t dev_max_lws,
	uint32_t* dbg_offsets, uint32_t* dbg_counters, uint32_t* dbg_counters_sum);

/* clo_scan_blelloch.c:129-195 + clo_scan_blelloch.cl:49-211. Exclusive scan,
 * elem_size -> sum_size widening, wrap-around in the sum type. Reference
 * contract kept: the tail numel % (2*lws) is never scanned (blelloch.cl:70)
 * and data_out beyond the scanned blocks is left untouched. Returns launches. */
int clo_oracle_blelloch(const void* data_in, void* data_out, size_t numel,
	int elem_size, int sum_size, size_t lws_max, size_t dev_max_lws);

/* The reference's own known-answer checks. */
/* clo_scan_bench.c:252-271: serial exclusive scan (sum type arithmetic). */
void clo_oracle_serial_scan(const void* data_in, void* data_out, size_t numel,
	int elem_size, int sum_size);
/* clo_sort_bench.c:211-226 + clo_bench.c:26-65: returns index of first
 * adjacent pair out of order (typed compare of whole elements), or -1. */
long clo_oracle_check_sorted(const void* data, size_t numel, int elem_size, int kind);

/* Independent references (not restatements): stable merge sort by key, used to
 * pin the restatements above. */
void clo_oracle_stable_sort(void* data, size_t numel, const clo_oracle_desc* d);

/* clo_bench.c:67-142 value distributions on top of GLib's GRand (MT19937,
 * g_rand_double = two 32-bit draws). type uses the CloType numbering
 * (clo_common.in.h:108-120). Fills numel elements. */
void clo_oracle_bench_rand(uint32_t seed, int clo_type, void* out, size_t numel);
/* clo_scan_bench.c:219-223: (gulong)(g_rand_double(rng) * 128) stored as elem type. */
void clo_oracle_scan_bench_rand(uint32_t seed, int elem_size, void* out, size_t numel);

/* CPU baseline ("port"): the satradix / blelloch decomposition above with the
 * per-work-group loops spread over OpenMP threads. Same results as the serial
 * versions. Returns threads used. */
int clo_oracle_satradix_mt(void* data, size_t numel, const clo_oracle_desc* d,
	unsigned radix, size_t lws, int threads);
int clo_oracle_blelloch_mt(const void* data_in, void* data_out, size_t numel,
	int elem_size, int sum_size, size_t lws, int threads);
/* The bitonic networks with the work-items of every launch (disjoint pairs, tiles) spread over
 * OpenMP threads: same bits as the serial versions. threads <= 0: all host cores. */
void clo_oracle_sbitonic_mt(void* data, size_t numel, const clo_oracle_desc* d, int threads);
int clo_oracle_abitonic_mt(void* data, size_t numel, const clo_oracle_desc* d,
	size_t lws_max, size_t dev_max_lws,
	unsigned minps, unsigned maxps, unsigned maxsfs, int threads);

#ifdef __cplusplus
}
#endif
#endif
