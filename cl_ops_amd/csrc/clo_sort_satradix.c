/*
 * clo_sort_satradix.c — host driver of the "satradix" sorter over HIP.
 *
 * Mirrors src/cl_ops/sort/clo_sort_satradix.c of the reference: options
 * (`radix=`, `scan=`, `scan<opt>=`; :366-421), lazily created scanner (:62-111),
 * introspection (:478-675) and the in-place contract (:680). The per-digit
 * launch loop (:264-313: localsort, histogram, scan, scatter) is replaced by
 * one call into the C-ABI, clo_hip_radix_sort (include/clo_hip.h), which runs
 * histogram -> counter scan -> one sort-and-scatter kernel per PAIR of digits.
 *
 * Differences from upstream, on purpose:
 *  - passes cover the KEY bits only: upstream loops over elem_size*8/bits
 *    digits (:167-169); the extra ones re-sort by digits already sorted
 *    (OpenCL shifts wrap modulo the key width) and change nothing;
 *  - when radix's bit count does not divide the key width (e.g. radix=8),
 *    upstream drops the top bits (integer division at :168-169) and returns a
 *    partially sorted array; here the last digit is simply narrower;
 *  - aux buffers live in the sorter and are reused (upstream @todo at :239);
 *  - data_out != NULL gives the sorted array in data_out and leaves data_in
 *    untouched; numel need not be a power of two;
 *  - COMPARE is ignored exactly as upstream (always ascending); signed and
 *    floating-point keys come out in numeric order (upstream: raw-bit order,
 *    which its own typed check rejects for negative keys).
 */
#include "clo_sort.h"
#include "clo_scan.h"
#include "clo_internal.h"

#include <string.h>

#define CLO_SORT_SATRADIX_SCAN_DEFAULT "blelloch"

typedef struct {
	cl_uint radix;
	char* scan_type;
	char* scan_opts;
	CloScan* scanner;
	clo_devbuf tmp;       /* ping-pong partner of the array being sorted */
	clo_devbuf workspace; /* per-tile histograms, offsets, chunk sums */
	clo_devbuf pairs;     /* (ordered key, index) pairs of a run-time compiled get_key */
	clo_devbuf seg_ws;    /* workspace of the segmented sorts (sort_segments: the sharded sort's local step) */
	int host_pipeline;    /* CLO_SORT_HOST_PIPELINE as read when the sorter was made: -1 unset, 0, 1 */
	clo_status_cell* status;  /* the workspace's status word, for the sorts whose kernels poll (clo_hip_radix_polls) */
	void* ws_ready;           /* the allocation whose header (status word) has been cleared */
	size_t ws_ready_bytes;
	clo_stream_guard guard;   /* the cached buffers are used by one stream at a time */
	struct sat_pipe_res_s* pipe;   /* resources of the pipelined host-data path (created at its first use) */
} clo_sort_satradix_data;

static const char* clo_sort_satradix_knames[] = {
	CLO_SORT_SATRADIX_KNAME_LOCALSORT, CLO_SORT_SATRADIX_KNAME_HISTOGRAM, CLO_SORT_SATRADIX_KNAME_SCATTER
};

/* ref: clo_sort_satradix.c:62-111 */
static CloScan* clo_sort_satradix_get_scanner(CloSort* sorter, GError** err) {
	clo_sort_satradix_data* data = (clo_sort_satradix_data*) clo_sort_get_data(sorter);
	if (data->scanner == NULL) {
		data->scanner = clo_scan_new(data->scan_type, data->scan_opts, clo_sort_get_context(sorter),
			CLO_UINT, CLO_UINT, ccl_program_get_build_options(clo_sort_get_program(sorter)), err);
	}
	return data->scanner;
}

/* The cached buffers for a sort of `numel` elements on cq_exec: the ping-pong partner, the
 * workspace (its header cleared when the allocation is fresh), the status word watched by the
 * queue when the sort's kernels poll. Before the command's start event, so that (re)allocation
 * is not timed as device work. The cached buffers belong to one queue at a time. */
static int satradix_reserve(CloSort* sorter, CCLQueue* cq_exec, size_t numel, GError** err) {
	clo_sort_satradix_data* data = (clo_sort_satradix_data*) clo_sort_get_data(sorter);
	const CloSortKeySpec* ks = clo_sort_get_key_spec(sorter);
	const int bits_in_digit = (int) clo_tzc((int) data->radix);
	void* stream = ccl_queue_get_stream(cq_exec);
	if (clo_hip_failed(clo_stream_guard_enter(&data->guard, cq_exec), err, "hipStreamWaitEvent")) return 0;
	const int jit = clo_sort_get_jit(sorter) != NULL;   /* then (key, index) pairs of 8 bytes are what gets sorted */
	const size_t ws_bytes = clo_hip_radix_workspace_bytes(numel, jit ? 8 : ks->elem_size, jit ? 32 : ks->key_bits, bits_in_digit);
	if (clo_hip_failed(clo_devbuf_reserve(&data->tmp, jit ? numel * 8 : numel * (size_t) ks->elem_size), err, "hipMalloc(satradix aux)")) return 0;
	if (clo_hip_failed(clo_devbuf_reserve(&data->workspace, ws_bytes), err, "hipMalloc(satradix workspace)")) return 0;
	if (jit && clo_hip_failed(clo_devbuf_reserve(&data->pairs, numel * 8), err, "hipMalloc(satradix key pairs)")) return 0;
	if (data->ws_ready != data->workspace.ptr || data->ws_ready_bytes != data->workspace.bytes) {   /* a fresh allocation: its status word is garbage */
		if (clo_hip_failed(clo_hip_memset_async(data->workspace.ptr, 0, 512, stream), err, "hipMemsetAsync")) return 0;
		data->ws_ready = data->workspace.ptr;
		data->ws_ready_bytes = data->workspace.bytes;
	}
	/* the queues that watch this sorter's status word must never be left with the
	 * address of a workspace that has been reallocated since */
	clo_status_cell_set_word(data->status, data->workspace.ptr);
	if (clo_hip_radix_polls(numel, jit ? 8 : ks->elem_size, bits_in_digit)) {
		/* tile-to-tile look-back inside the passes: a give-up must not pass as success */
		if (clo_status_cell_take_tripped(data->status)) clo_debug("SATRADIX: the previous sort on this sorter gave up a spin");
		ccl_queue_watch_status(cq_exec, data->status);
	}
	return 1;
}

static CCLEvent* clo_sort_satradix_sort_with_device_data(CloSort* sorter, CCLQueue* cq_exec,
	CCLQueue* cq_comm, CCLBuffer* data_in, CCLBuffer* data_out, size_t numel, size_t lws_max,
	GError** err) {

	clo_return_val_if_fail(err == NULL || *err == NULL, NULL);
	clo_return_val_if_fail(cq_exec != NULL, NULL);
	clo_return_val_if_fail(data_in != NULL, NULL);
	(void) cq_comm;  /* no pre-copy is needed: the first pass reads data_in directly */
	(void) lws_max;  /* launch shapes are fixed by the HIP kernels */

	clo_sort_satradix_data* data = (clo_sort_satradix_data*) clo_sort_get_data(sorter);
	const CloSortKeySpec* ks = clo_sort_get_key_spec(sorter);
	const int bits_in_digit = (int) clo_tzc((int) data->radix);
	const size_t bytes = numel * (size_t) ks->elem_size;
	void* stream = ccl_queue_get_stream(cq_exec);
	CCLEvent* evt = NULL;

	/* Signed and floating-point keys: upstream's kernels order every type by
	 * its raw bits (ref: clo_sort_satradix.cl:34-258 never look at the type), so
	 * negative keys end up after the positive ones and upstream's own check
	 * (benchmarks/clo_bench.c:26-65) rejects the result. Here such keys go through
	 * an order-preserving transform inside the first and last pass; non-negative
	 * inputs give the same output as upstream. A float key must be the whole
	 * key type (no sub-field of a float). */
	if (ks->key_kind == 2 && ks->key_bits != 8 * ks->key_size) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "satradix: a floating-point key must span its whole type");
		return NULL;
	}
	const int key_kind = (ks->key_kind == 1 && ks->key_bits < 8 * ks->key_size) ? 0 : ks->key_kind;  /* sign bit masked off */
	if (bytes > ccl_buffer_get_size(data_in) || (data_out && bytes > ccl_buffer_get_size(data_out))) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "numel (%zu) exceeds the size of the device buffers", numel);
		return NULL;
	}
	if (numel > 0xffffffffull) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "numel must be below 2^32");
		return NULL;
	}

	clo_debug("SATRADIX: radix=%u (bits_in_digit=%d), numel=%zu, key bits [%d,%d)",
		data->radix, bits_in_digit, numel, ks->key_shift, ks->key_shift + ks->key_bits);

	if (numel > 0 && !satradix_reserve(sorter, cq_exec, numel, err)) return NULL;

	/* A profiling queue gets one event per kernel under upstream's names
	 * (clo_sort_satradix.c:282,295,312; the counter scan's launches under the
	 * scanner's, clo_scan_blelloch.c:158); any other queue one event per sort. */
	static const clo_kname knames[] = {
		{ "radix_hist", CLO_SORT_SATRADIX_KNAME_HISTOGRAM }, { "radix_ghist", CLO_SORT_SATRADIX_KNAME_HISTOGRAM },
		{ "radix_offsets", "clo_scan_blelloch_wgscan" },
		{ "radix_pass", CLO_SORT_SATRADIX_KNAME_SCATTER }, { "radix_sweep", CLO_SORT_SATRADIX_KNAME_SCATTER },
		{ "radix_small", CLO_SORT_SATRADIX_KNAME_LOCALSORT },
		{ "radix_extract", "satradix_keys" }, { "radix_gather", "satradix_gather" }
	};
	const int per_kernel = ccl_queue_is_profiling(cq_exec) && numel > 0;
	clo_kernel_events ke;
	if (per_kernel) {
		clo_kernel_events_install(&ke, cq_exec, knames, sizeof(knames) / sizeof(knames[0]), CLO_SORT_SATRADIX_KNAME_SCATTER);
	} else {
		evt = ccl_queue_begin_command(cq_exec, CLO_SORT_SATRADIX_KNAME_SCATTER, err);
		if (!evt) return NULL;
	}

	if (numel > 0) {
		void* src = ccl_buffer_get_device_ptr(data_in);
		void* dst = data_out ? ccl_buffer_get_device_ptr(data_out) : src;
		int st;
		if (clo_sort_get_jit(sorter) != NULL)
			st = clo_hip_radix_jit_sort(clo_sort_get_jit(sorter), src, dst, data->pairs.ptr, data->tmp.ptr, numel,
				bits_in_digit, data->workspace.ptr, data->workspace.bytes, stream);
		else
			st = clo_hip_radix_sort(src, dst, data->tmp.ptr, numel, ks->elem_size, ks->key_shift,
				ks->key_bits, key_kind, bits_in_digit, data->workspace.ptr, data->workspace.bytes, stream);
		if (clo_hip_failed(st, err, "clo_hip_radix_sort")) {
			if (per_kernel) clo_kernel_events_remove(&ke, NULL); else ccl_queue_abort_command(cq_exec, evt);
			return NULL;
		}
	}

	if (per_kernel) {
		GError* e2 = NULL;
		CCLEvent* last = clo_kernel_events_remove(&ke, &e2);
		if (e2) { clo_gerror_propagate(err, e2); return NULL; }
		if (last) return last;
		evt = ccl_queue_begin_command(cq_exec, CLO_SORT_SATRADIX_KNAME_SCATTER, err);   /* (no launch was observed) */
		if (!evt) return NULL;
	}
	if (!ccl_queue_end_command(cq_exec, evt, err)) { ccl_queue_abort_command(cq_exec, evt); return NULL; }
	return evt;
}

/* After the sort's result has been waited for: did one of its kernels give up a
 * bounded look-back spin (single-sweep passes)? clo_internal.h: clo_sort_impl_ext. */
static cl_bool clo_sort_satradix_check_status(CloSort* sorter, CCLQueue* cq, GError** err) {
	clo_sort_satradix_data* data = (clo_sort_satradix_data*) clo_sort_get_data(sorter);
	if (!data || !data->workspace.ptr || data->ws_ready != data->workspace.ptr) return CL_TRUE;
	if (clo_status_cell_take_tripped(data->status)) {   /* a queue's own check found it first (and has cleared the word) */
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY,
			"satradix: a work-group gave up waiting for its predecessors' counts (bounded look-back spin); the output is not valid");
		return CL_FALSE;
	}
	const int st = clo_hip_check_status(data->workspace.ptr, ccl_queue_get_stream(cq));
	if (st == 0) return CL_TRUE;
	if (st == CLO_HIP_ETIMEOUT) {
		data->ws_ready = NULL;   /* the header is cleared again before the next sort */
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY,
			"satradix: a work-group gave up waiting for its predecessors' counts (bounded look-back spin); the output is not valid");
		return CL_FALSE;
	}
	clo_hip_failed(st, err, "clo_hip_check_status");
	return CL_FALSE;
}

/* ---------------------------------------------------------------------------
 * clo_sort_with_host_data with the transfers overlapped.
 *
 * Upstream's path is blocking: copy in, sort, copy out (sort/clo_sort_abstract.c:348-395).
 * A sort's first output element depends on its last input element, so the two copies can
 * never overlap each other (docs/lab_notebook.md, "Measurement"); what CAN disappear behind them is the sort:
 *   copy in   the array arrives in SAT_PIPE_CHUNKS chunks on the transfer stream; as soon
 *             as chunk c is there, the exec stream splits it — stably — into 16 buckets by
 *             the top 4 key bits (clo_hip_msd_partition: the sort's own pass kernel), under
 *             the copy of chunk c + 1;
 *   then      the 16 x chunks bucket sizes reach the host (the one synchronisation), and for
 *             bucket b = 0 .. 15: its pieces are copied together in chunk order (stable) to
 *             their place in the result, sorted there by the remaining passes, and
 *   copy out  a helper thread copies bucket b to the caller's array while bucket b + 1 is
 *             being gathered and sorted (a copy to pageable memory holds its thread).
 * Exposed beside the two copies: the split of the last chunk and the sort of the first
 * bucket, a sixteenth of the array. The result is the blocking path's, bit for bit: a stable
 * split by the top key bits followed by stable sorts of the buckets IS a stable sort.
 * For unsigned keys (whole elements or fields) of 4- and 8-byte elements from 2^24 elements on.
 *
 * It buys wall time with device time. Host to host (profiles/r03_hostsort_pipeline.txt; copy in +
 * copy out alone in brackets): 2^24 uint32 3.4 ms instead of 4.7 (2.4), 2^28 uint32 39.3 instead of
 * 41.1 (38.1), 2^28 uint64 78.1 instead of 86.2 (76.2) — 0.7 .. 1.9 ms beside the bare copies
 * where the blocking path has 0.9 .. 10. But the exec queue — what upstream's harness times
 * (benchmarks/clo_sort_bench.c:201-207) — does a split pass plus sixteen sorts of a sixteenth of
 * the array instead of one sort: 1.9 ms of device time instead of 0.25 at 2^24 keys, about twice
 * the time at 2^28. Hence the default: pipelined, UNLESS cq_exec was created with
 * CL_QUEUE_PROFILING_ENABLE — a caller who profiles the exec queue is timing "the sort" and gets
 * the one sort upstream would run. CLO_SORT_HOST_PIPELINE=0 / 1 in the environment overrides.
 * --------------------------------------------------------------------------- */
#include <pthread.h>

#ifndef SAT_PIPE_MIN_NUMEL   /* (the sanitizer build of tests/hoststub shrinks it) */
#define SAT_PIPE_MIN_NUMEL ((size_t) 1 << 24)
#endif
#define SAT_PIPE_DEFAULT_BYTES ((size_t) 128 << 20)   /* the default takes the pipeline from this array size on */
#define SAT_PIPE_CHUNKS 8
#define SAT_PIPE_BITS 4
#define SAT_PIPE_BUCKETS (1 << SAT_PIPE_BITS)
/* Round 4: where the sorter runs segmented sorts (radix 16 / 256) the split takes 8 key bits instead of 4 — 256
 * sub-buckets, 16 to a bucket — and a bucket is ONE segmented sort of its 16 sub-buckets on the remaining bits
 * (clo_hip_radix_sort_segmented), whose first pass gathers the sub-buckets' pieces out of the split chunks by itself:
 * split + 3 passes for 32-bit keys, where the 4-bit split needed a gather copy and 4 passes per bucket (6 trips). */
#define SAT_PIPE_SEG_BITS 8
#define SAT_PIPE_SUBS (1 << SAT_PIPE_SEG_BITS)
/* The two array-sized buffers of the pipeline stay cached in the sorter up to this size each; larger ones are freed
 * when the call ends (the sorter would otherwise keep three times the array resident until it is destroyed). Not lower:
 * freeing and allocating 2 x 1 GiB per call cost 19 ms of a 39 ms sort of 2^28 keys (measured); at 8 GiB the array's own
 * transfers take 300 ms. */
#define SAT_PIPE_KEEP_BYTES ((size_t) 8 << 30)

typedef struct sat_pipe_res_s {
	clo_devbuf in, part, counts, msd_ws;   /* the array as it arrives / split chunk by chunk; bucket sizes; workspace of the splits */
	void* s_in;                            /* own transfer stream when the caller gave one queue for everything */
	void* s_out;
	void* in_done[SAT_PIPE_CHUNKS];
	void* sorted[SAT_PIPE_BUCKETS];
} sat_pipe_res;

typedef struct {
	int device;
	sat_pipe_res* r;
	char* out_host;
	size_t es;
	size_t off[SAT_PIPE_BUCKETS + 1];      /* where bucket b starts in the result (elements) */
	pthread_mutex_t mtx;
	pthread_cond_t cv;
	int posted, completed, abort, status;
} sat_pipe;

static int satradix_segments_apply(CloSort* sorter);
static cl_bool clo_sort_satradix_reserve_segments(CloSort* sorter, CCLQueue* cq_exec, size_t numel, int nseg, int* handled, GError** err);

static void sat_pipe_res_free(sat_pipe_res* r) {
	if (!r) return;
	if (r->s_in) { clo_hip_stream_synchronize(r->s_in); clo_hip_stream_destroy(r->s_in); }
	if (r->s_out) { clo_hip_stream_synchronize(r->s_out); clo_hip_stream_destroy(r->s_out); }
	for (int i = 0; i < SAT_PIPE_CHUNKS; ++i) clo_hip_event_destroy(r->in_done[i]);
	for (int i = 0; i < SAT_PIPE_BUCKETS; ++i) clo_hip_event_destroy(r->sorted[i]);
	clo_devbuf_release(&r->in);
	clo_devbuf_release(&r->part);
	clo_devbuf_release(&r->counts);
	clo_devbuf_release(&r->msd_ws);
	free(r);
}

static void* sat_pipe_copy_out(void* arg) {
	sat_pipe* p = (sat_pipe*) arg;
	clo_hip_set_device(p->device);
	for (int b = 0; b < SAT_PIPE_BUCKETS; ++b) {
		pthread_mutex_lock(&p->mtx);
		while (p->posted <= b && !p->abort) pthread_cond_wait(&p->cv, &p->mtx);
		const int stop = p->posted <= b;
		pthread_mutex_unlock(&p->mtx);
		if (stop) break;
		const size_t cnt = p->off[b + 1] - p->off[b];
		int st = clo_hip_event_synchronize(p->r->sorted[b]);
		if (st == 0 && cnt) st = clo_hip_memcpy_d2h_async(p->out_host + p->off[b] * p->es, (const char*) p->r->in.ptr + p->off[b] * p->es, cnt * p->es, p->r->s_out);
		if (st == 0) st = clo_hip_stream_synchronize(p->r->s_out);
		pthread_mutex_lock(&p->mtx);
		if (st != 0 && p->status == 0) p->status = st;
		p->completed = b + 1;
		pthread_cond_broadcast(&p->cv);
		pthread_mutex_unlock(&p->mtx);
	}
	return NULL;
}

static cl_bool clo_sort_satradix_host_pipeline(CloSort* sorter, CCLQueue* cq_exec, CCLQueue* cq_comm, const void* data_in,
	void* data_out, size_t numel, int* handled, GError** err) {
	clo_sort_satradix_data* data = (clo_sort_satradix_data*) clo_sort_get_data(sorter);
	const CloSortKeySpec* ks = clo_sort_get_key_spec(sorter);
	const int es = ks->elem_size;
	const int key_kind = (ks->key_kind == 1 && ks->key_bits < 8 * ks->key_size) ? 0 : ks->key_kind;
	*handled = 0;
	const int seg = satradix_segments_apply(sorter) && ks->key_bits > SAT_PIPE_SEG_BITS;
	const int part_bits = seg ? SAT_PIPE_SEG_BITS : SAT_PIPE_BITS;
	const int subs = 1 << part_bits, per = subs / SAT_PIPE_BUCKETS;   /* sub-buckets of the split; per bucket */
	if (clo_sort_get_jit(sorter) != NULL || key_kind != 0 || (es != 4 && es != 8) || ks->key_bits <= part_bits) return CL_FALSE;
	if (numel < SAT_PIPE_MIN_NUMEL || numel > 0xffffffffull) return CL_FALSE;
	{   /* default: on, unless the caller profiles the exec queue (see above); CLO_SORT_HOST_PIPELINE=0 / 1 decides for both */
		const int flag = ((clo_sort_satradix_data*) clo_sort_get_data(sorter))->host_pipeline;
		/* (unset: from 128 MiB on — at 64 MiB, 2^24 uint32, the blocking path is the shorter: 2.96 against 3.20 ms host to
		 * host, profiles/r04_hostsort_pipeline.txt; at 128 MiB the pipeline wins, 5.54 against 5.74) */
		const int want = flag >= 0 ? flag : (!ccl_queue_is_profiling(cq_exec) && numel * (size_t) es >= SAT_PIPE_DEFAULT_BYTES);
		if (!want) return CL_FALSE;
	}
	*handled = 1;

	const int bits_in_digit = (int) clo_tzc((int) data->radix);
	void* s_exec = ccl_queue_get_stream(cq_exec);
	size_t chunk = (numel + SAT_PIPE_CHUNKS - 1) / SAT_PIPE_CHUNKS;
	chunk = (chunk + 4095) & ~(size_t) 4095;   /* every chunk starts 16-byte aligned */
	const int nchunks = (int) ((numel + chunk - 1) / chunk);
	uint64_t counts[SAT_PIPE_CHUNKS][SAT_PIPE_SUBS];   /* [chunk][sub-bucket]: `subs` entries of a row are used */
	sat_pipe p;
	pthread_t helper;
	int helper_started = 0, st = 0;
	cl_bool ok = CL_FALSE;
	CCLEvent* evt = NULL;
	const char* what = "pipeline resources (hipMalloc / hipStreamCreate)";

	memset(&p, 0, sizeof(p));
	pthread_mutex_init(&p.mtx, NULL);
	pthread_cond_init(&p.cv, NULL);
	if (clo_hip_get_device(&p.device) != 0) p.device = 0;
	sat_pipe_res* r = data->pipe;
	if (!r) {
		r = data->pipe = (sat_pipe_res*) calloc(1, sizeof(*r));
		if (!r) { st = CLO_HIP_EARGS; goto finish; }
		st = clo_hip_stream_create(&r->s_out);
		if (st == 0) st = clo_hip_stream_create(&r->s_in);
		for (int i = 0; i < SAT_PIPE_CHUNKS && st == 0; ++i) st = clo_hip_event_create(&r->in_done[i]);
		for (int i = 0; i < SAT_PIPE_BUCKETS && st == 0; ++i) st = clo_hip_event_create(&r->sorted[i]);
		if (st != 0) goto finish;
	}
	if ((st = clo_stream_guard_enter(&data->guard, cq_exec)) != 0) { what = "hipStreamWaitEvent"; goto finish; }
	st = clo_devbuf_reserve(&r->in, numel * (size_t) es);
	if (st == 0) st = clo_devbuf_reserve(&r->part, numel * (size_t) es);
	if (st == 0) st = clo_devbuf_reserve(&r->counts, sizeof(counts));
	if (st == 0) {   /* (not monotone in the size — the tile shape changes with it: the last chunk may need more than a full one) */
		const size_t last = numel - (size_t) (nchunks - 1) * chunk;
		const size_t w_full = clo_hip_msd_workspace_bytes(chunk, es, part_bits), w_last = clo_hip_msd_workspace_bytes(last, es, part_bits);
		st = clo_devbuf_reserve(&r->msd_ws, w_full > w_last ? w_full : w_last);
	}
	if (st != 0) goto finish;
	p.r = r; p.out_host = (char*) data_out; p.es = (size_t) es;
	void* s_in = ccl_queue_get_stream(cq_comm);
	if (s_in == s_exec) s_in = r->s_in;

	/* ---- copy in, every chunk split as soon as it is there ---- */
	for (int c = 0; c < nchunks; ++c) {
		const size_t off = (size_t) c * chunk;
		const size_t cnt = numel - off < chunk ? numel - off : chunk;
		what = "hipMemcpyAsync(h2d)";
		st = clo_hip_memcpy_h2d_async((char*) r->in.ptr + off * es, (const char*) data_in + off * es, cnt * es, s_in);
		if (st == 0) st = clo_hip_event_record(r->in_done[c], s_in);
		if (st == 0) st = clo_hip_stream_wait_event(s_exec, r->in_done[c]);
		if (st != 0) goto finish;
		evt = ccl_queue_begin_command(cq_exec, CLO_SORT_SATRADIX_KNAME_HISTOGRAM, err);
		if (!evt) goto finish;
		what = "clo_hip_msd_partition";
		st = clo_hip_msd_partition((char*) r->in.ptr + off * es, (char*) r->part.ptr + off * es, cnt, es, ks->key_shift, ks->key_bits,
			part_bits, (uint64_t*) r->counts.ptr + (size_t) c * SAT_PIPE_SUBS, r->msd_ws.ptr, r->msd_ws.bytes, s_exec);
		if (st != 0) { ccl_queue_abort_command(cq_exec, evt); goto finish; }
		if (!ccl_queue_end_command(cq_exec, evt, err)) { ccl_queue_abort_command(cq_exec, evt); goto finish; }
	}
	/* ---- the bucket sizes (the one synchronisation of the call) ---- */
	what = "bucket sizes";
	st = clo_hip_memcpy_d2h_async(counts, r->counts.ptr, (size_t) nchunks * SAT_PIPE_SUBS * sizeof(uint64_t), s_exec);
	if (st == 0) st = clo_hip_stream_synchronize(s_exec);
	if (st != 0) goto finish;
	size_t largest = 0, sum = 0;
	for (int b = 0; b < SAT_PIPE_BUCKETS; ++b) {
		size_t tot = 0;
		for (int c = 0; c < nchunks; ++c)
			for (int k = b * per; k < (b + 1) * per; ++k) tot += (size_t) counts[c][k];
		p.off[b] = sum;
		sum += tot;
		if (tot > largest) largest = tot;
	}
	p.off[SAT_PIPE_BUCKETS] = sum;
	if (sum != numel) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY, "bucket sizes (%zu) do not add up to numel (%zu)", sum, numel);
		goto finish;
	}
	int polls = 0;
	if (!seg) {   /* the workspace a sort needs is not monotone in its size (the single-sweep passes of mid-sized buckets keep more state): the largest any bucket asks for */
		size_t ws_max = 0;
		for (int b = 0; b < SAT_PIPE_BUCKETS; ++b) {
			const size_t tot = p.off[b + 1] - p.off[b];
			const size_t w = tot ? clo_hip_radix_workspace_bytes(tot, es, ks->key_bits - SAT_PIPE_BITS, bits_in_digit) : 0;
			if (w > ws_max) ws_max = w;
			if (tot && clo_hip_radix_polls(tot, es, bits_in_digit)) polls = 1;
		}
		what = "hipMalloc(satradix workspace)";
		if ((st = clo_devbuf_reserve(&data->workspace, ws_max)) != 0) goto finish;
	}
	if (largest > 0 && !satradix_reserve(sorter, cq_exec, largest, err)) goto finish;
	/* (satradix_reserve arms the give-up watch when the LARGEST bucket's sort polls; the mid-sized ones are the
	 * ones that take the polling passes: a caller's ccl_queue_finish must report their give-up too) */
	if (polls) ccl_queue_watch_status(cq_exec, data->status);
	if (seg && largest > 0) {
		int h = 0;
		if (!clo_sort_satradix_reserve_segments(sorter, cq_exec, largest, per, &h, err)) goto finish;
	}
	if (pthread_create(&helper, NULL, sat_pipe_copy_out, &p) != 0) { st = CLO_HIP_EARGS; what = "pthread_create"; goto finish; }
	helper_started = 1;

	/* ---- bucket by bucket: gather its pieces in chunk order, sort it where it belongs, hand it to the copy out ---- */
	for (int b = 0; b < SAT_PIPE_BUCKETS; ++b) {
		const size_t tot = p.off[b + 1] - p.off[b];
		evt = ccl_queue_begin_command(cq_exec, CLO_SORT_SATRADIX_KNAME_SCATTER, err);
		if (!evt) goto finish;
		size_t pos = p.off[b];
		if (seg && tot > 0) {
			/* the bucket's `per` sub-buckets = segments, each in one piece per chunk (in chunk order: stable); the first
			 * pass gathers them. The passes go part -> X0 -> X1 ..., X alternating between the bucket's place in the
			 * result and the scratch buffer such that the LAST one writes the result. */
			size_t seg_counts[SAT_PIPE_SUBS / SAT_PIPE_BUCKETS], pn[SAT_PIPE_CHUNKS * SAT_PIPE_SUBS / SAT_PIPE_BUCKETS], po[SAT_PIPE_CHUNKS * SAT_PIPE_SUBS / SAT_PIPE_BUCKETS];
			int ps[SAT_PIPE_CHUNKS * SAT_PIPE_SUBS / SAT_PIPE_BUCKETS], np = 0, in_b = 0;
			for (int kk = 0; kk < per; ++kk) {
				const int sub = b * per + kk;
				seg_counts[kk] = 0;
				for (int c = 0; c < nchunks; ++c) {
					size_t before = 0;
					for (int k = 0; k < sub; ++k) before += (size_t) counts[c][k];
					pn[np] = (size_t) counts[c][sub];
					po[np] = (size_t) c * chunk + before;
					ps[np] = kk;
					seg_counts[kk] += pn[np];
					++np;
				}
			}
			const int rest = ks->key_bits - SAT_PIPE_SEG_BITS, passes = (rest + 7) / 8;
			void* final = (char*) r->in.ptr + p.off[b] * es;
			what = "clo_hip_radix_sort_segmented";
			st = clo_hip_radix_sort_segmented(r->part.ptr, passes % 2 ? data->tmp.ptr : final, passes % 2 ? final : data->tmp.ptr, tot, seg_counts, per,
				pn, po, ps, np, es, ks->key_shift, rest, bits_in_digit, data->seg_ws.ptr, data->seg_ws.bytes, s_exec, &in_b);
			/* (which buffer ends up holding the result follows from the pass count; the two were chosen above from THIS
			 * file's count — should the sort's pass schedule ever differ, fail loudly rather than copy out the wrong one) */
			if (st == 0 && in_b != passes % 2) { what = "clo_hip_radix_sort_segmented (result buffer parity)"; st = CLO_HIP_EARGS; }
		}
		what = seg ? what : "hipMemcpyAsync(d2d)";
		for (int c = 0; c < nchunks && st == 0 && !seg; ++c) {
			size_t before = 0;
			for (int k = 0; k < b; ++k) before += (size_t) counts[c][k];
			const size_t cnt = (size_t) counts[c][b];
			if (cnt) st = clo_hip_memcpy_d2d_async((char*) r->in.ptr + pos * es, (const char*) r->part.ptr + ((size_t) c * chunk + before) * es, cnt * es, s_exec);
			pos += cnt;
		}
		if (st == 0 && tot > 0 && !seg) {
			what = "clo_hip_radix_sort";
			void* at = (char*) r->in.ptr + p.off[b] * es;
			/* (the top SAT_PIPE_BITS key bits are equal inside a bucket: the passes cover the rest) */
			st = clo_hip_radix_sort(at, at, data->tmp.ptr, tot, es, ks->key_shift, ks->key_bits - SAT_PIPE_BITS, 0, bits_in_digit,
				data->workspace.ptr, data->workspace.bytes, s_exec);
		}
		if (st != 0) { ccl_queue_abort_command(cq_exec, evt); goto finish; }
		if (!ccl_queue_end_command(cq_exec, evt, err)) { ccl_queue_abort_command(cq_exec, evt); goto finish; }
		what = "hipEventRecord";
		if ((st = clo_hip_event_record(r->sorted[b], s_exec)) != 0) goto finish;
		pthread_mutex_lock(&p.mtx);
		p.posted = b + 1;
		pthread_cond_broadcast(&p.cv);
		pthread_mutex_unlock(&p.mtx);
	}
	pthread_mutex_lock(&p.mtx);
	while (p.completed < SAT_PIPE_BUCKETS && p.status == 0) pthread_cond_wait(&p.cv, &p.mtx);
	st = p.status;
	pthread_mutex_unlock(&p.mtx);
	what = "copy out";
	ok = st == 0;
	/* every bucket is back on the host, so every sort has completed: did one give up a look-back spin? */
	if (ok && !clo_sort_satradix_check_status(sorter, cq_exec, err)) ok = CL_FALSE;

finish:
	if (helper_started) {
		pthread_mutex_lock(&p.mtx);
		p.abort = 1;
		pthread_cond_broadcast(&p.cv);
		pthread_mutex_unlock(&p.mtx);
		pthread_join(helper, NULL);
	}
	if (st != 0 && (err == NULL || *err == NULL)) clo_hip_failed(st, err, what);
	if (!ok) {   /* leave nothing of this call in flight */
		if (r && r->s_in) clo_hip_stream_synchronize(r->s_in);
		clo_hip_stream_synchronize(ccl_queue_get_stream(cq_comm));
		clo_hip_stream_synchronize(s_exec);
	}
	pthread_mutex_destroy(&p.mtx);
	pthread_cond_destroy(&p.cv);
	if (r && r->in.bytes > SAT_PIPE_KEEP_BYTES) {   /* large arrays: do not leave two more copies of them resident in the sorter */
		clo_hip_stream_synchronize(s_exec);
		clo_devbuf_release(&r->in);
		clo_devbuf_release(&r->part);
	}
	return ok && (err == NULL || *err == NULL);
}

static cl_bool clo_sort_satradix_reserve(CloSort* sorter, CCLQueue* cq_exec, size_t numel, GError** err) {
	return (numel == 0 || satradix_reserve(sorter, cq_exec, numel, err)) ? CL_TRUE : CL_FALSE;
}

/* ---- segmented sorts (clo_internal.h: clo_sort_impl_ext.sort_segments) ---- */
static int satradix_segments_apply(CloSort* sorter) {
	clo_sort_satradix_data* data = (clo_sort_satradix_data*) clo_sort_get_data(sorter);
	const CloSortKeySpec* ks = clo_sort_get_key_spec(sorter);
	return (data->radix == 16 || data->radix == 256) && (ks->elem_size == 4 || ks->elem_size == 8) && ks->key_kind == 0
		&& clo_sort_get_jit(sorter) == NULL;
}

static cl_bool clo_sort_satradix_reserve_segments(CloSort* sorter, CCLQueue* cq_exec, size_t numel, int nseg, int* handled, GError** err) {
	clo_sort_satradix_data* data = (clo_sort_satradix_data*) clo_sort_get_data(sorter);
	const CloSortKeySpec* ks = clo_sort_get_key_spec(sorter);
	*handled = satradix_segments_apply(sorter);
	if (!*handled || numel == 0) return CL_TRUE;
	if (clo_hip_failed(clo_stream_guard_enter(&data->guard, cq_exec), err, "hipStreamWaitEvent")) return CL_FALSE;
	const size_t need = clo_hip_radix_seg_workspace_bytes(numel, nseg, ks->elem_size, (int) clo_tzc((int) data->radix));
	if (need == 0) { *handled = 0; return CL_TRUE; }
	if (clo_hip_failed(clo_devbuf_reserve(&data->seg_ws, need), err, "hipMalloc(satradix segmented workspace)")) return CL_FALSE;
	return CL_TRUE;
}

static CCLEvent* clo_sort_satradix_sort_segments(CloSort* sorter, CCLQueue* cq_exec, void* a_dev, void* b_dev, size_t numel,
	const size_t* seg_counts, int nseg, const size_t* piece_counts, const size_t* piece_offsets, const int* piece_segment, int npieces,
	const void* src2_dev, const int* piece_source,
	int key_shift, int key_bits, int* result_in_b, int* handled, GError** err) {
	clo_sort_satradix_data* data = (clo_sort_satradix_data*) clo_sort_get_data(sorter);
	const CloSortKeySpec* ks = clo_sort_get_key_spec(sorter);
	*result_in_b = 0;
	if (!clo_sort_satradix_reserve_segments(sorter, cq_exec, numel, nseg, handled, err)) return NULL;
	if (!*handled) return NULL;
	static const clo_kname knames[] = {
		{ "radix_hist", CLO_SORT_SATRADIX_KNAME_HISTOGRAM }, { "radix_offsets", "clo_scan_blelloch_wgscan" },
		{ "radix_pass", CLO_SORT_SATRADIX_KNAME_SCATTER }, { "radix_seg_tables", CLO_SORT_SATRADIX_KNAME_LOCALSORT }
	};
	const int per_kernel = ccl_queue_is_profiling(cq_exec) && numel > 0;
	clo_kernel_events ke;
	CCLEvent* evt = NULL;
	if (per_kernel) {
		clo_kernel_events_install(&ke, cq_exec, knames, sizeof(knames) / sizeof(knames[0]), CLO_SORT_SATRADIX_KNAME_SCATTER);
	} else {
		evt = ccl_queue_begin_command(cq_exec, CLO_SORT_SATRADIX_KNAME_SCATTER, err);
		if (!evt) return NULL;
	}
	if (numel > 0) {
		const int st = clo_hip_radix_sort_segmented2(a_dev, src2_dev, a_dev, b_dev, numel, seg_counts, nseg, piece_counts, piece_offsets, piece_segment, piece_source, npieces,
			ks->elem_size, key_shift, key_bits, (int) clo_tzc((int) data->radix), data->seg_ws.ptr, data->seg_ws.bytes,
			ccl_queue_get_stream(cq_exec), result_in_b);
		if (clo_hip_failed(st, err, "clo_hip_radix_sort_segmented")) {
			if (per_kernel) clo_kernel_events_remove(&ke, NULL); else ccl_queue_abort_command(cq_exec, evt);
			return NULL;
		}
	} else {
		*result_in_b = ((key_bits + 7) / 8) % 2;   /* (include/clo_hip.h: where a sort of these key bits ends; an empty one says the same as its neighbours) */
	}
	if (per_kernel) {
		GError* e2 = NULL;
		CCLEvent* last = clo_kernel_events_remove(&ke, &e2);
		if (e2) { clo_gerror_propagate(err, e2); return NULL; }
		if (last) return last;
		evt = ccl_queue_begin_command(cq_exec, CLO_SORT_SATRADIX_KNAME_SCATTER, err);
		if (!evt) return NULL;
	}
	if (!ccl_queue_end_command(cq_exec, evt, err)) { ccl_queue_abort_command(cq_exec, evt); return NULL; }
	return evt;
}

const clo_sort_impl_ext clo_sort_satradix_ext = { "satradix", clo_sort_satradix_check_status, clo_sort_satradix_host_pipeline,
	clo_sort_satradix_reserve, clo_sort_satradix_sort_segments, clo_sort_satradix_reserve_segments };

typedef struct {
	clo_sort_satradix_data* data;
	char* scan_opts;
	size_t scan_opts_len;
} satradix_opt_ctx;

/* ref: clo_sort_satradix.c:381-414 */
static int satradix_option(const char* key, const char* value, const char* token, void* user, GError** err) {
	satradix_opt_ctx* c = (satradix_opt_ctx*) user;
	if (strcmp(key, "radix") == 0) {
		c->data->radix = (cl_uint) atoi(value);
		if (clo_ones32(c->data->radix) != 1) {
			clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "Radix must be a power of 2.");
			return 0;
		}
		if (c->data->radix < 2 || c->data->radix > 256) {
			clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "Radix must be between 2 and 256 in the HIP build.");
			return 0;
		}
	} else if (strncasecmp(key, "scan", 4) == 0) {
		if (strlen(key) == 4) {
			free(c->data->scan_type);
			c->data->scan_type = strdup(value);
		} else {
			/* "scanfoo=bar" is forwarded to the scanner as "foo=bar," */
			const char* fwd = token + 4;
			size_t n = strlen(fwd);
			char* p = (char*) realloc(c->scan_opts, c->scan_opts_len + n + 2);
			if (!p) return 0;
			c->scan_opts = p;
			memcpy(p + c->scan_opts_len, fwd, n);
			p[c->scan_opts_len + n] = ',';
			p[c->scan_opts_len + n + 1] = '\0';
			c->scan_opts_len += n + 1;
		}
	} else {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "Invalid option key '%s' for satradix sort.", key);
		return 0;
	}
	return 1;
}

static void sat_pipe_res_free(struct sat_pipe_res_s* r);

static void satradix_free(clo_sort_satradix_data* data) {
	sat_pipe_res_free(data->pipe);
	free(data->scan_type);
	free(data->scan_opts);
	if (data->scanner) clo_scan_destroy(data->scanner);
	clo_status_cell_set_word(data->status, NULL);
	clo_status_cell_unref(data->status);
	clo_devbuf_release(&data->tmp);
	clo_devbuf_release(&data->workspace);
	clo_devbuf_release(&data->pairs);
	clo_devbuf_release(&data->seg_ws);
	clo_stream_guard_release(&data->guard);
	free(data);
}

/* ref: clo_sort_satradix.c:342-452 */
static const char* clo_sort_satradix_init(CloSort* sorter, const char* options, GError** err) {
	clo_return_val_if_fail(err == NULL || *err == NULL, NULL);
	clo_sort_satradix_data* data = (clo_sort_satradix_data*) calloc(1, sizeof(*data));
	if (!data) return NULL;
	data->radix = 16;
	data->host_pipeline = clo_env_flag("CLO_SORT_HOST_PIPELINE");
	data->status = clo_status_cell_new(NULL);
	satradix_opt_ctx c = { data, NULL, 0 };
	if (!clo_parse_options(options, satradix_option, &c, "satradix", err)) {
		free(c.scan_opts);
		satradix_free(data);
		return NULL;
	}
	if (data->scan_type == NULL) data->scan_type = strdup(CLO_SORT_SATRADIX_SCAN_DEFAULT);
	data->scan_opts = c.scan_opts ? c.scan_opts : strdup("");
	if (!data->scan_type || !data->scan_opts) {   /* (out of host memory) */
		satradix_free(data);
		return NULL;
	}
	/* Upstream hands the per-digit counters to a CloScan of this type
	 * (clo_sort_satradix.c:94,298) and so fails at the first sort when the type or
	 * its options are not valid. Here the counter scan is fused into the radix
	 * kernels (clo_hip_radixw.hip: a scan in digit-major order, upstream's
	 * counters_sum); the scanner object serves the introspection calls only.
	 * Type and options are therefore checked now, with the scan API's own
	 * messages, instead of being accepted and never looked at. */
	if (strcmp(data->scan_type, "blelloch") != 0) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_IMPL_NOT_FOUND,
			"The requested scan implementation, '%s', was not found.", data->scan_type);
		satradix_free(data);
		return NULL;
	}
	if (data->scan_opts[0] != '\0') {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "Invalid options for blelloch scan.");
		satradix_free(data);
		return NULL;
	}
	clo_sort_set_data(sorter, data);
	return "satradix:hip";
}

static void clo_sort_satradix_finalize(CloSort* sorter) {
	clo_sort_satradix_data* data = (clo_sort_satradix_data*) clo_sort_get_data(sorter);
	if (data) satradix_free(data);
	clo_sort_set_data(sorter, NULL);
}

/* ref: clo_sort_satradix.c:478-510 — own kernels + the scanner's. */
static cl_uint clo_sort_satradix_get_num_kernels(CloSort* sorter, GError** err) {
	GError* err_internal = NULL;
	CloScan* scanner = clo_sort_satradix_get_scanner(sorter, &err_internal);
	if (err_internal) { clo_gerror_propagate(err, err_internal); return 0; }
	cl_uint n = clo_scan_get_num_kernels(scanner, &err_internal);
	if (err_internal) { clo_gerror_propagate(err, err_internal); return 0; }
	return CLO_SORT_SATRADIX_NUM_KERNELS + n;
}

/* ref: clo_sort_satradix.c:516-566 */
static const char* clo_sort_satradix_get_kernel_name(CloSort* sorter, cl_uint i, GError** err) {
	GError* err_internal = NULL;
	cl_uint num_kernels = clo_sort_satradix_get_num_kernels(sorter, &err_internal);
	if (err_internal) { clo_gerror_propagate(err, err_internal); return NULL; }
	clo_return_val_if_fail(i < num_kernels, NULL);
	if (i < CLO_SORT_SATRADIX_NUM_KERNELS) return clo_sort_satradix_knames[i];
	return clo_scan_get_kernel_name(clo_sort_satradix_get_scanner(sorter, err),
		i - CLO_SORT_SATRADIX_NUM_KERNELS, err);
}

/* ref: clo_sort_satradix.c:573-675. The values are those of the HIP kernels
 * that do each job: "localsort" is fused into the scatter kernel (its LDS is
 * reported there), "histogram" is the first-pass digit count. */
static size_t clo_sort_satradix_get_localmem_usage(CloSort* sorter, cl_uint i, size_t lws_max,
	size_t numel, GError** err) {
	GError* err_internal = NULL;
	cl_uint num_kernels = clo_sort_satradix_get_num_kernels(sorter, &err_internal);
	if (err_internal) { clo_gerror_propagate(err, err_internal); return 0; }
	clo_return_val_if_fail(i < num_kernels, 0);
	clo_sort_satradix_data* data = (clo_sort_satradix_data*) clo_sort_get_data(sorter);
	const int es = (int) clo_sort_get_element_size(sorter);
	const int bits = (int) clo_tzc((int) data->radix);
	switch (i) {
		case 0: return 0;
		case 1: return clo_hip_kernel_lds_bytes("radix_hist", es, bits);
		case 2: return clo_hip_kernel_lds_bytes("radix_pass", es, bits);
		default:
			return clo_scan_get_localmem_usage(clo_sort_satradix_get_scanner(sorter, err),
				i - CLO_SORT_SATRADIX_NUM_KERNELS, lws_max, numel, err);
	}
}

/* ref: clo_sort_satradix.c:678-687 */
const CloSortImplDef clo_sort_satradix_def = {
	"satradix",
	CL_TRUE,
	clo_sort_satradix_init,
	clo_sort_satradix_finalize,
	clo_sort_satradix_sort_with_device_data,
	clo_sort_satradix_get_num_kernels,
	clo_sort_satradix_get_kernel_name,
	clo_sort_satradix_get_localmem_usage
};
