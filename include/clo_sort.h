/*
 * clo_sort.h — the CloSort plugin API, as exported by the reference's
 * src/cl_ops/sort/clo_sort_abstract.in.h:43-170 (same names, argument order,
 * ownership and error behaviour), implemented over HIP.
 *
 * Algorithms registered: "sbitonic", "abitonic", "gselect", "satradix"
 * (clo_sort_sbitonic.in.h:36, clo_sort_abitonic.in.h:116, clo_sort_gselect.in.h:36,
 * clo_sort_satradix.in.h:55).
 *
 * Divergences from upstream, all deliberate:
 *  - `compare` / `get_key` are OpenCL C macro bodies upstream (JIT). Here they
 *    are parsed into a fixed family: get_key = x, (x) >> N, ((x) >> N) & MASK,
 *    optional casts; compare = ((a) > (b)) or ((a) < (b)) select ahead-of-time
 *    kernels. Any other expression is compiled at run time with hiprtc (as
 *    upstream does with the OpenCL JIT): into the bitonic and gselect kernels
 *    themselves, or, for satradix, into a kernel that materialises the keys
 *    (any key type; 8-byte keys take two rounds).
 *  - data_out != NULL works (upstream sorts data_in regardless,
 *    clo_sort_satradix.c:276,305 / clo_sort_abitonic.c:388) and leaves data_in
 *    untouched; numel need not be a power of two.
 *  - `compiler_opts` reaches the run-time compiler whenever kernels are compiled at
 *    run time, as upstream hands it to the OpenCL JIT (clo_sort_abstract.c:177-178):
 *    -DNAME[=value], -UNAME, -Idir (OpenCL's -cl-... switches are dropped, anything
 *    else is the compiler's to refuse). An expression that uses a name the options
 *    define is always compiled, never parsed: compiler_opts = "-DSHIFT=12" with
 *    get_key = "((x) >> SHIFT) & 0xfff" works as upstream. Sorters built from the
 *    ahead-of-time kernels have nothing to compile and ignore the options.
 *  - `lws_max` is accepted and ignored: a tile is a work-group's registers and LDS,
 *    fixed at compile time (upstream caps the work-group size it asks for,
 *    clo_sort_satradix.c:184-190); results never depend on it.
 */
#ifndef CLO_SORT_H
#define CLO_SORT_H

#include "clo_common.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CLO_SORT_IMPLS "sbitonic, abitonic, gselect, satradix"

/* clo_sort_abstract.in.h:43-110 */
typedef struct clo_sort_impl_def {
	const char* name;
	cl_bool in_place;
	/* Returns a non-NULL token on success (upstream: the kernel source text). */
	const char* (*init)(CloSort* sorter, const char* options, GError** err);
	void (*finalize)(CloSort* sorter);
	CCLEvent* (*sort_with_device_data)(CloSort* sorter, CCLQueue* cq_exec, CCLQueue* cq_comm,
		CCLBuffer* data_in, CCLBuffer* data_out, size_t numel, size_t lws_max, GError** err);
	cl_uint (*get_num_kernels)(CloSort* sorter, GError** err);
	const char* (*get_kernel_name)(CloSort* sorter, cl_uint i, GError** err);
	size_t (*get_localmem_usage)(CloSort* sorter, cl_uint i, size_t lws_max, size_t numel, GError** err);
} CloSortImplDef;

/* clo_sort_abstract.in.h:116-170.
 * lws_max (every call that takes it): upstream caps the OpenCL work-group size it asks
 * cf4ocl2 to suggest (sort/clo_sort_satradix.c:184-190, clo_sort_abitonic.c:340-349). The HIP
 * kernels' shapes are fixed at compile time — a tile IS a work-group's registers and LDS —
 * and are chosen by array size, not by the caller: the argument is accepted, recorded in the
 * CLO_DEBUG trace and otherwise ignored in every sorter and in the scan; results never depend
 * on it (upstream's do not either). clo_*_get_localmem_usage reports the LDS of the shape
 * that `numel` selects.
 * clo_sort_with_host_data: upstream's blocking path (copy in, sort, copy out), except that
 * satradix on unsigned keys in arrays of 128 MiB and more overlaps the sort with both copies
 * (same result bit for bit, 2-10 % less wall time; the exec queue then runs a split pass and
 * sixteen segmented sorts instead of one sort: clo_sort_satradix.c) unless cq_exec is a profiling
 * queue; CLO_SORT_HOST_PIPELINE=0 / 1 (read at clo_sort_new) decides for all queues from 2^24 elements. */
CloSort* clo_sort_new(const char* type, const char* options, CCLContext* ctx,
	CloType* elem_type, CloType* key_type, const char* compare, const char* get_key,
	const char* compiler_opts, GError** err);
void clo_sort_destroy(CloSort* sorter);
CCLEvent* clo_sort_with_device_data(CloSort* sorter, CCLQueue* cq_exec, CCLQueue* cq_comm,
	CCLBuffer* data_in, CCLBuffer* data_out, size_t numel, size_t lws_max, GError** err);
cl_bool clo_sort_with_host_data(CloSort* sorter, CCLQueue* cq_exec, CCLQueue* cq_comm,
	void* data_in, void* data_out, size_t numel, size_t lws_max, GError** err);
CCLContext* clo_sort_get_context(CloSort* sorter);
CCLProgram* clo_sort_get_program(CloSort* sorter);
CloType clo_sort_get_element_type(CloSort* sorter);
size_t clo_sort_get_element_size(CloSort* sorter);
CloType clo_sort_get_key_type(CloSort* sorter);
size_t clo_sort_get_key_size(CloSort* sorter);
void* clo_sort_get_data(CloSort* sorter);
void clo_sort_set_data(CloSort* sorter, void* data);
cl_uint clo_sort_get_num_kernels(CloSort* sorter, GError** err);
const char* clo_sort_get_kernel_name(CloSort* sorter, cl_uint i, GError** err);
size_t clo_sort_get_localmem_usage(CloSort* sorter, cl_uint i, size_t lws_max, size_t numel, GError** err);

extern const CloSortImplDef clo_sort_sbitonic_def;  /* clo_sort_sbitonic.in.h:36 */
extern const CloSortImplDef clo_sort_gselect_def;   /* clo_sort_gselect.in.h:36 */
extern const CloSortImplDef clo_sort_abitonic_def;  /* clo_sort_abitonic.in.h:116 */
extern const CloSortImplDef clo_sort_satradix_def;  /* clo_sort_satradix.in.h:55 */

/* Kernel-name strings reported by get_kernel_name — part of the observable API
 * (clo_sort_sbitonic.in.h:33, clo_sort_satradix.in.h:42-52, clo_sort_abitonic.in.h:64-106). */
#define CLO_SORT_SBITONIC_KNAME "sbitonic"
#define CLO_SORT_GSELECT_KNAME "gselect"   /* clo_sort_gselect.in.h:33 */
#define CLO_SORT_SATRADIX_NUM_KERNELS 3
#define CLO_SORT_SATRADIX_KNAME_LOCALSORT "satradix_localsort"
#define CLO_SORT_SATRADIX_KNAME_HISTOGRAM "satradix_histogram"
#define CLO_SORT_SATRADIX_KNAME_SCATTER "satradix_scatter"
/* The 26 abitonic names (abit_any, abit_local_s2 .. s11, abit_priv_*, abit_hyb_*) are UPSTREAM's kernel list
 * (clo_sort_abitonic.in.h:64-106), reported for API compatibility: the HIP schedule launches four kernels of its own
 * (a tile presort, a tile merge and two strided kernels). clo_sort_get_localmem_usage answers for the
 * kernel that `numel` selects: 0 for the names that stand for register-only kernels ("any", "priv"), the tile kernels'
 * static LDS for the others; sbitonic's one name answers for the tiled schedule it runs (0 only below 32 elements or
 * with CLO_SBITONIC_STEPS=1). */
#define CLO_SORT_ABITONIC_NUM_KERNELS 26

/* Not upstream — parsed form of (elem_type, key_type, compare, get_key) shared
 * by the three drivers. */
typedef struct {
	int elem_size;
	int key_size;
	int key_shift;     /* key = (elem >> key_shift) & key_mask */
	int key_bits;      /* significant bits after masking */
	int key_kind;      /* 0 unsigned, 1 signed, 2 float */
	int descending;
} CloSortKeySpec;
const CloSortKeySpec* clo_sort_get_key_spec(CloSort* sorter);
/* Not upstream — non-NULL when the sorter was specialised at run time (hiprtc)
 * because compare/get_key fall outside the ahead-of-time family. */
void* clo_sort_get_jit(CloSort* sorter);

#ifdef __cplusplus
}
#endif
#endif
