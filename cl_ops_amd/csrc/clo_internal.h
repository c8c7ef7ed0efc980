/*
 * clo_internal.h — helpers shared by the host drivers (not installed).
 */
#ifndef CLO_INTERNAL_H
#define CLO_INTERNAL_H

#include <stdio.h>
#include <stdlib.h>

#include "clo_common.h"
#include "clo_hip.h"

#ifdef CLO_USE_GLIB
#define clo_return_if_fail g_return_if_fail
#define clo_return_val_if_fail g_return_val_if_fail
#define clo_gerror_set g_set_error
#define clo_gerror_propagate g_propagate_error
#else
/* g_return_*_if_fail: a soft assertion — complain on stderr, return a default. */
#define clo_return_if_fail(expr) \
	do { if (!(expr)) { fprintf(stderr, "cl_ops-CRITICAL: %s: assertion '%s' failed\n", __func__, #expr); return; } } while (0)
#define clo_return_val_if_fail(expr, val) \
	do { if (!(expr)) { fprintf(stderr, "cl_ops-CRITICAL: %s: assertion '%s' failed\n", __func__, #expr); return (val); } } while (0)
#endif

/* CLO_DEBUG=1 in the environment turns on the g_debug-style trace upstream
 * emits under log domain "cl_ops" (e.g. clo_sort_satradix.c:171,192). */
/* Helpers shared by the host drivers: not part of the library's interface (the headers under include/ are), so not exported either. */
#define CLO_INTERNAL __attribute__((visibility("hidden")))

CLO_INTERNAL int clo_debug_enabled(void);
#define clo_debug(...) do { if (clo_debug_enabled()) { fprintf(stderr, "cl_ops-DEBUG: " __VA_ARGS__); fputc('\n', stderr); } } while (0)

/* Environment switches of the host drivers: read when an object is made, never per call. CLO_NO_WARMUP (no
 * dummy sorts / scans at the first object of a kind), and the 0 / 1 switches CLO_SORT_HOST_PIPELINE (satradix's
 * clo_sort_with_host_data: pipelined or blocking whatever the queue) and CLO_SBITONIC_STEPS (sbitonic with one
 * launch per step). */
CLO_INTERNAL int clo_env_no_warmup(void);
CLO_INTERNAL int clo_env_flag(const char* name);

/* A grow-only device allocation cached inside a sorter/scanner object, so the
 * hot path never calls hipMalloc after the first use (upstream allocates and
 * frees its aux buffers on every call: clo_sort_satradix.c:242-257,327-330). */
typedef struct {
	void* ptr;
	size_t bytes;
} clo_devbuf;

/* Returns 0 or a clo_hip status. Contents are NOT preserved when it grows. */
CLO_INTERNAL int clo_devbuf_reserve(clo_devbuf* b, size_t bytes);
CLO_INTERNAL void clo_devbuf_release(clo_devbuf* b);

/* Buffers cached in a sorter / scanner belong to one queue at a time. When a
 * call arrives on another queue than the previous one, its work must come after
 * the previous call's. The guard remembers the QUEUE of the last call (holding
 * its struct, clo_queue_hold), not a stream handle: the queue may have been
 * destroyed meanwhile (clo_*_with_host_data makes and destroys a queue of its own
 * when the caller passes none; round 1 kept the raw handle and called
 * hipStreamSynchronize on it later: undefined behaviour, observed as an abort
 * inside the HIP runtime) — a destroyed queue was synchronised when it went, so
 * nothing is left to wait for; a live one gets an event recorded on it now, and
 * the new queue's stream waits for that. Calls that stay on one queue pay
 * nothing (an event per call cost 3.5 us of device time between two kernels). */
typedef struct {
	CCLQueue* cq;    /* the queue of the last call (held), or NULL */
	void* evt;       /* created the first time two live queues alternate */
} clo_stream_guard;
CLO_INTERNAL int clo_stream_guard_enter(clo_stream_guard* g, CCLQueue* cq);   /* before the call's first enqueue; 0 or a clo_hip status */
CLO_INTERNAL void clo_stream_guard_release(clo_stream_guard* g);

/* A launch sequence that depends only on its arguments (buffers, size,
 * stream), cached as an executable hipGraph: the FIRST call with a given key
 * launches normally, a second consecutive call with the same key captures the
 * sequence and replays it, later ones replay. Any other key drops the graph.
 * Never used while per-kernel timing is on (its event pairs need real
 * launches). `enqueue` puts the whole sequence on `stream` and returns a
 * clo_hip status. */
typedef struct {
	void* exec;
	const void* k0;
	const void* k1;
	void* stream;
	size_t n;
	int variant, seen;
} clo_graph_cache;
typedef int (*clo_enqueue_fn)(void* user, void* stream);
CLO_INTERNAL int clo_graph_cache_run(clo_graph_cache* gc, int allowed, const void* k0, const void* k1, size_t n, int variant,
	void* stream, clo_enqueue_fn enqueue, void* user);
CLO_INTERNAL void clo_graph_cache_release(clo_graph_cache* gc);

/* Extensions of a scan implementation that are NOT part of the public
 * CloScanImplDef (whose layout is upstream's, clo_scan_abstract.in.h:41-103: a
 * plugin compiled against upstream's header must stay valid). Looked up by
 * implementation name; an implementation without an entry simply has none.
 *  scan_chunk: scans `numel` elements at raw device pointers as one chunk of a
 *    longer array: *carry_in_dev (device uint64, NULL = 0) is added to every
 *    output, *carry_out_dev receives the next chunk's carry. Enqueued on cq_exec
 *    as one command (a profiling queue sees it like any scan). Used by the
 *    pipelined clo_scan_with_host_data.
 *  check_status: after cq has been synchronised, fails with CLO_ERROR_LIBRARY if
 *    a kernel of this scanner gave up a bounded spin (wrong output). */
struct clo_scan;
typedef struct {
	const char* name;
	cl_bool (*scan_chunk)(struct clo_scan* scanner, CCLQueue* cq_exec, const void* in_dev, void* out_dev, size_t numel,
		const void* carry_in_dev, void* carry_out_dev, GError** err);
	cl_bool (*check_status)(struct clo_scan* scanner, CCLQueue* cq, GError** err);
} clo_scan_impl_ext;
CLO_INTERNAL const clo_scan_impl_ext* clo_scan_impl_ext_find(const char* name);
CLO_INTERNAL extern const clo_scan_impl_ext clo_scan_blelloch_ext;

/* The same for sort implementations (CloSortImplDef keeps upstream's layout,
 * clo_sort_abstract.in.h:43-110).
 *  check_status: after the sort's result has been waited for, fails with
 *    CLO_ERROR_LIBRARY if a kernel of this sorter gave up a bounded spin (the
 *    single-sweep radix passes poll other work-groups). clo_sort_with_host_data
 *    calls it whatever queues the caller passed. */
struct clo_sort;
typedef struct {
	const char* name;
	cl_bool (*check_status)(struct clo_sort* sorter, CCLQueue* cq, GError** err);
	/* host_pipeline: clo_sort_with_host_data with the transfers overlapped with the sort (SURVEY.md
	 * §8f-2). Sets *handled = 0 and touches nothing when this sort is not one it pipelines (the caller
	 * then takes upstream's blocking path, sort/clo_sort_abstract.c:348-395); otherwise *handled = 1
	 * and the return value is the call's. */
	cl_bool (*host_pipeline)(struct clo_sort* sorter, CCLQueue* cq_exec, CCLQueue* cq_comm, const void* data_in, void* data_out,
		size_t numel, int* handled, GError** err);
	/* reserve: grows the sorter's cached buffers for a sort of numel elements on cq_exec NOW. A caller about to
	 * enqueue several sorts of different sizes back to back (the slices of the sharded sort) asks for every size
	 * first: growing a buffer between two of them would free memory the earlier one is still using — hipFree
	 * waits for the device, and the overlap the slices exist for is gone. */
	cl_bool (*reserve)(struct clo_sort* sorter, CCLQueue* cq_exec, size_t numel, GError** err);
	/* sort_segments: `nseg` segments of one device range of `numel` elements, each sorted on its own by the key bits
	 * [key_shift, key_shift + key_bits), in shared launches (clo_hip_radix_sort_segmented, include/clo_hip.h: where the
	 * segments and the optional pieces lie; the result is in b_dev when *result_in_b, else in a_dev; a_dev is overwritten).
	 * *handled = 0 and nothing enqueued when this sorter cannot (another radix than 16 / 256, element size, key kind) —
	 * the caller then sorts the range some other way. reserve_segments grows the buffers it needs for that size NOW. */
	/* (src2_dev / piece_source: pieces that lie in a second device range, clo_hip_radix_sort_segmented2; NULL / NULL: none) */
	CCLEvent* (*sort_segments)(struct clo_sort* sorter, CCLQueue* cq_exec, void* a_dev, void* b_dev, size_t numel,
		const size_t* seg_counts, int nseg, const size_t* piece_counts, const size_t* piece_offsets, const int* piece_segment, int npieces,
		const void* src2_dev, const int* piece_source,
		int key_shift, int key_bits, int* result_in_b, int* handled, GError** err);
	cl_bool (*reserve_segments)(struct clo_sort* sorter, CCLQueue* cq_exec, size_t numel, int nseg, int* handled, GError** err);
} clo_sort_impl_ext;
CLO_INTERNAL const clo_sort_impl_ext* clo_sort_impl_ext_find(const char* name);
CLO_INTERNAL extern const clo_sort_impl_ext clo_sort_satradix_ext;

/* Per-kernel events on a profiling queue. Upstream enqueues every kernel itself
 * and names its event (clo_sort_satradix.c:282,295,312; clo_scan_blelloch.c:158,
 * 183,193; clo_sort_sbitonic.c:115), and CCLProf consumers see those names. Here
 * one C-ABI call launches all the kernels of a sort, so while such a call runs
 * the driver has the C-ABI report every launch (clo_hip_set_launch_observer) and
 * opens / closes one CCLEvent per kernel, named through `map` (kernel family ->
 * upstream's event name; families not listed get `other`). Only used when the
 * queue profiles: event pairs between back-to-back kernels cost a few percent. */
typedef struct { const char* label; const char* name; } clo_kname;
typedef struct {
	CCLQueue* cq;
	const clo_kname* map;
	size_t nmap;
	const char* other;
	CCLEvent* open;
	CCLEvent* last;
	int failed;
} clo_kernel_events;
CLO_INTERNAL void clo_kernel_events_install(clo_kernel_events* ke, CCLQueue* cq, const clo_kname* map, size_t nmap, const char* other);
/* Removes the observer; returns the last kernel's event (NULL if there was no
 * launch) and reports a failed event operation through err. */
CLO_INTERNAL CCLEvent* clo_kernel_events_remove(clo_kernel_events* ke, GError** err);

/* Set *err from a clo_hip_* status (domain CCL_HIP_ERROR); returns 1 if st != 0. */
CLO_INTERNAL int clo_hip_failed(int st, GError** err, const char* what);

/* Parse one "key=value" option list the way upstream does
 * (g_strsplit_set on "," then "="; clo_sort_abitonic.c:486-543,
 * clo_sort_satradix.c:366-421). Calls cb(key, value, token, user) per non-empty
 * token; a token without exactly one '=' makes it return 0 and set *bad to a
 * malloc'd copy of the token. cb returns 0 to abort. */
typedef int (*clo_option_cb)(const char* key, const char* value, const char* token, void* user, GError** err);
CLO_INTERNAL int clo_parse_options(const char* options, clo_option_cb cb, void* user, const char* algo, GError** err);

#endif
