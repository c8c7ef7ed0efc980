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
	clo_status_cell* status;  /* the workspace's status word, for the sorts whose kernels poll (clo_hip_radix_polls) */
	void* ws_ready;           /* the allocation whose header (status word) has been cleared */
	size_t ws_ready_bytes;
	clo_stream_guard guard;   /* the cached buffers are used by one stream at a time */
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

	if (numel > 0) {
		/* The cached buffers belong to one queue at a time. Reserved before the
		 * command's start event so that (re)allocation is not timed as device work. */
		if (clo_hip_failed(clo_stream_guard_enter(&data->guard, cq_exec), err, "hipStreamWaitEvent")) return NULL;
		const int jit = clo_sort_get_jit(sorter) != NULL;   /* then (key, index) pairs of 8 bytes are what gets sorted */
		const size_t ws_bytes = clo_hip_radix_workspace_bytes(numel, jit ? 8 : ks->elem_size, jit ? 32 : ks->key_bits, bits_in_digit);
		if (clo_hip_failed(clo_devbuf_reserve(&data->tmp, jit ? numel * 8 : bytes), err, "hipMalloc(satradix aux)")) return NULL;
		if (clo_hip_failed(clo_devbuf_reserve(&data->workspace, ws_bytes), err, "hipMalloc(satradix workspace)")) return NULL;
		if (jit && clo_hip_failed(clo_devbuf_reserve(&data->pairs, numel * 8), err, "hipMalloc(satradix key pairs)")) return NULL;
		if (data->ws_ready != data->workspace.ptr || data->ws_ready_bytes != data->workspace.bytes) {   /* a fresh allocation: its status word is garbage */
			if (clo_hip_failed(clo_hip_memset_async(data->workspace.ptr, 0, 512, stream), err, "hipMemsetAsync")) return NULL;
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
	}

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

const clo_sort_impl_ext clo_sort_satradix_ext = { "satradix", clo_sort_satradix_check_status };

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

static void satradix_free(clo_sort_satradix_data* data) {
	free(data->scan_type);
	free(data->scan_opts);
	if (data->scanner) clo_scan_destroy(data->scanner);
	clo_status_cell_set_word(data->status, NULL);
	clo_status_cell_unref(data->status);
	clo_devbuf_release(&data->tmp);
	clo_devbuf_release(&data->workspace);
	clo_devbuf_release(&data->pairs);
	clo_stream_guard_release(&data->guard);
	free(data);
}

/* ref: clo_sort_satradix.c:342-452 */
static const char* clo_sort_satradix_init(CloSort* sorter, const char* options, GError** err) {
	clo_return_val_if_fail(err == NULL || *err == NULL, NULL);
	clo_sort_satradix_data* data = (clo_sort_satradix_data*) calloc(1, sizeof(*data));
	if (!data) return NULL;
	data->radix = 16;
	data->status = clo_status_cell_new(NULL);
	satradix_opt_ctx c = { data, NULL, 0 };
	if (!clo_parse_options(options, satradix_option, &c, "satradix", err)) {
		free(c.scan_opts);
		satradix_free(data);
		return NULL;
	}
	if (data->scan_type == NULL) data->scan_type = strdup(CLO_SORT_SATRADIX_SCAN_DEFAULT);
	data->scan_opts = c.scan_opts ? c.scan_opts : strdup("");
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
