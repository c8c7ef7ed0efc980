/*
 * clo_scan_blelloch.c — host driver of the "blelloch" scanner over HIP.
 *
 * Mirrors src/cl_ops/scan/clo_scan_blelloch.c of the reference: any option is
 * an error (:43-45), three kernel names (:251-279), exclusive scan
 * data_in[elem] -> data_out[sum]. The three launches of :146-195 (workgroupScan,
 * workgroupSumsScan, addWorkgroupSums) are replaced by one call into the C-ABI,
 * clo_hip_scan_exclusive: a single-pass chained scan that reads and writes
 * every element once. The whole array is scanned (upstream leaves the tail
 * numel % (2*lws) unscanned, clo_scan_blelloch.cl:70).
 */
#include "clo_scan.h"
#include "clo_internal.h"

#include <string.h>

typedef struct {
	clo_devbuf fp_workspace;   /* tile sums of the floating-point scans */
	clo_devbuf workspace;
	void* ws_ready;            /* the allocation clo_hip_scan_workspace_init has prepared (NULL: none) */
	size_t ws_ready_bytes;
	clo_status_cell* status;   /* the workspace's status word, watched by the queues this scanner has used */
	clo_stream_guard guard;    /* the workspace is used by one stream at a time */
} clo_scan_blelloch_data;

/* The one kernel of this scanner does the jobs of upstream's three; its event
 * carries the first one's name (clo_scan_blelloch.c:158). */
#define CLO_SCAN_BLELLOCH_EVENT "clo_scan_blelloch_wgscan"

/* The workspace for `numel` elements, ready for a scan on `stream`: (re)allocated
 * when it has to grow, zeroed once per allocation (and again after a call that
 * gave up a spin), its status word watched by the queue. */
static int blelloch_workspace(CloScan* scanner, CCLQueue* cq_exec, size_t numel, GError** err) {
	clo_scan_blelloch_data* data = (clo_scan_blelloch_data*) clo_scan_get_data(scanner);
	void* stream = ccl_queue_get_stream(cq_exec);
	const int es = (int) clo_scan_get_element_size(scanner);
	const int ss = (int) clo_scan_get_sum_size(scanner);
	if (clo_hip_failed(clo_stream_guard_enter(&data->guard, cq_exec), err, "hipStreamWaitEvent")) return 0;
	const size_t ws = clo_hip_scan_workspace_bytes(numel, es, ss);
	if (data->workspace.ptr && data->workspace.bytes < ws) {
		/* about to grow: the old range is freed, and the registry of prepared workspaces (clo_hip_scan_workspace_init)
		 * must not go on vouching for whatever is allocated at that address next */
		clo_hip_scan_workspace_forget(data->workspace.ptr);
		data->ws_ready = NULL;
	}
	if (clo_hip_failed(clo_devbuf_reserve(&data->workspace, ws), err, "hipMalloc(scan workspace)")) return 0;
	const int tripped = clo_status_cell_take_tripped(data->status);
	if (data->ws_ready != data->workspace.ptr || data->ws_ready_bytes != data->workspace.bytes || tripped) {
		if (clo_hip_failed(clo_hip_scan_workspace_init(data->workspace.ptr, data->workspace.bytes, stream), err,
			"clo_hip_scan_workspace_init")) return 0;
		data->ws_ready = data->workspace.ptr;
		data->ws_ready_bytes = data->workspace.bytes;
		clo_status_cell_set_word(data->status, data->workspace.ptr);
	}
	ccl_queue_watch_status(cq_exec, data->status);
	clo_debug("BLELLOCH: N=%zu elem=%dB sum=%dB workspace=%zuB", numel, es, ss, ws);
	return 1;
}

static CCLEvent* clo_scan_blelloch_scan_with_device_data(CloScan* scanner, CCLQueue* cq_exec,
	CCLQueue* cq_comm, CCLBuffer* data_in, CCLBuffer* data_out, size_t numel, size_t lws_max,
	GError** err) {

	clo_return_val_if_fail(err == NULL || *err == NULL, NULL);
	clo_return_val_if_fail(cq_exec != NULL, NULL);
	clo_return_val_if_fail(data_in != NULL && data_out != NULL, NULL);
	(void) cq_comm;
	(void) lws_max;

	clo_scan_blelloch_data* data = (clo_scan_blelloch_data*) clo_scan_get_data(scanner);
	const int es = (int) clo_scan_get_element_size(scanner);
	const int ss = (int) clo_scan_get_sum_size(scanner);
	void* stream = ccl_queue_get_stream(cq_exec);

	if (numel * (size_t) es > ccl_buffer_get_size(data_in) || numel * (size_t) ss > ccl_buffer_get_size(data_out)) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "numel (%zu) exceeds the size of the device buffers", numel);
		return NULL;
	}

	if (clo_hip_scan_is_typed((int) clo_scan_get_elem_type(scanner), (int) clo_scan_get_sum_type(scanner))) {
		/* half / float / double sums, floating-point elements into integer sums, sums narrower than
		 * the elements: reduce, scan the tile sums, apply (clo_hip_fscan.hip) */
		const size_t wsb = clo_hip_scan_typed_workspace_bytes(numel, (int) clo_scan_get_sum_type(scanner));
		if (numel > 0) {
			if (clo_hip_failed(clo_stream_guard_enter(&data->guard, cq_exec), err, "hipStreamWaitEvent")) return NULL;
			if (clo_hip_failed(clo_devbuf_reserve(&data->fp_workspace, wsb), err, "hipMalloc(scan workspace)")) return NULL;
		}
		CCLEvent* fevt = ccl_queue_begin_command(cq_exec, CLO_SCAN_BLELLOCH_EVENT, err);
		if (!fevt) return NULL;
		if (numel > 0) {
			const int st = clo_hip_scan_exclusive_typed(ccl_buffer_get_device_ptr(data_in), ccl_buffer_get_device_ptr(data_out), numel,
				(int) clo_scan_get_elem_type(scanner), (int) clo_scan_get_sum_type(scanner), data->fp_workspace.ptr, data->fp_workspace.bytes, stream);
			if (clo_hip_failed(st, err, "clo_hip_scan_exclusive_typed")) { ccl_queue_abort_command(cq_exec, fevt); return NULL; }
		}
		if (!ccl_queue_end_command(cq_exec, fevt, err)) { ccl_queue_abort_command(cq_exec, fevt); return NULL; }
		return fevt;
	}

	if (numel > 0 && !blelloch_workspace(scanner, cq_exec, numel, err)) return NULL;

	CCLEvent* evt = ccl_queue_begin_command(cq_exec, CLO_SCAN_BLELLOCH_EVENT, err);
	if (!evt) return NULL;

	if (numel > 0) {
		int st = clo_hip_scan_exclusive(ccl_buffer_get_device_ptr(data_in), ccl_buffer_get_device_ptr(data_out),
			numel, es, clo_type_is_signed(clo_scan_get_elem_type(scanner)), ss,
			data->workspace.ptr, data->workspace.bytes, stream);
		if (clo_hip_failed(st, err, "clo_hip_scan_exclusive")) { ccl_queue_abort_command(cq_exec, evt); return NULL; }
	}

	if (!ccl_queue_end_command(cq_exec, evt, err)) { ccl_queue_abort_command(cq_exec, evt); return NULL; }
	return evt;
}

/* One chunk of a longer scan, on raw device pointers (clo_internal.h: clo_scan_impl_ext). */
static cl_bool clo_scan_blelloch_scan_chunk(CloScan* scanner, CCLQueue* cq_exec, const void* in_dev, void* out_dev,
	size_t numel, const void* carry_in_dev, void* carry_out_dev, GError** err) {
	clo_scan_blelloch_data* data = (clo_scan_blelloch_data*) clo_scan_get_data(scanner);
	void* stream = ccl_queue_get_stream(cq_exec);
	const int es = (int) clo_scan_get_element_size(scanner);
	const int ss = (int) clo_scan_get_sum_size(scanner);
	if (!blelloch_workspace(scanner, cq_exec, numel, err)) return CL_FALSE;
	CCLEvent* evt = ccl_queue_begin_command(cq_exec, CLO_SCAN_BLELLOCH_EVENT, err);
	if (!evt) return CL_FALSE;
	int st = clo_hip_scan_exclusive_carry(in_dev, out_dev, numel, es,
		clo_type_is_signed(clo_scan_get_elem_type(scanner)), ss,
		(const uint64_t*) carry_in_dev, (uint64_t*) carry_out_dev, data->workspace.ptr, data->workspace.bytes, stream);
	if (clo_hip_failed(st, err, "clo_hip_scan_exclusive_carry")) { ccl_queue_abort_command(cq_exec, evt); return CL_FALSE; }
	if (!ccl_queue_end_command(cq_exec, evt, err)) { ccl_queue_abort_command(cq_exec, evt); return CL_FALSE; }
	return CL_TRUE;
}

/* After cq has been synchronised: did a scan of this scanner give up a spin? */
static cl_bool clo_scan_blelloch_check_status(CloScan* scanner, CCLQueue* cq, GError** err) {
	clo_scan_blelloch_data* data = (clo_scan_blelloch_data*) clo_scan_get_data(scanner);
	if (!data || !data->workspace.ptr) return CL_TRUE;
	const int st = clo_hip_check_status(data->workspace.ptr, ccl_queue_get_stream(cq));
	if (st == 0) return CL_TRUE;
	if (st == CLO_HIP_ETIMEOUT) {
		data->ws_ready = NULL;   /* prepare the workspace again before the next scan */
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY,
			"blelloch scan: a work-group gave up waiting for its predecessors' prefix (bounded look-back spin); the output is not valid");
		return CL_FALSE;
	}
	clo_hip_failed(st, err, "clo_hip_check_status");
	return CL_FALSE;
}

/* ref: clo_scan_blelloch.c:219-249 — options must be empty. */
static const char* clo_scan_blelloch_init(CloScan* scanner, const char* options, GError** err) {
	if (options != NULL && strlen(options) > 0) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "Invalid options for blelloch scan.");
		return NULL;
	}
	clo_scan_blelloch_data* data = (clo_scan_blelloch_data*) calloc(1, sizeof(*data));
	if (!data) return NULL;
	data->status = clo_status_cell_new(NULL);
	clo_scan_set_data(scanner, data);
	return "blelloch:hip";
}

static void clo_scan_blelloch_finalize(CloScan* scan) {
	clo_scan_blelloch_data* data = (clo_scan_blelloch_data*) clo_scan_get_data(scan);
	if (data) {
		clo_status_cell_set_word(data->status, NULL);   /* queues still watching must not read freed memory */
		if (data->workspace.ptr) clo_hip_scan_workspace_forget(data->workspace.ptr);
		clo_status_cell_unref(data->status);
		clo_devbuf_release(&data->workspace);
		clo_devbuf_release(&data->fp_workspace);
		clo_stream_guard_release(&data->guard);
		free(data);
	}
	clo_scan_set_data(scan, NULL);
}

static cl_uint clo_scan_blelloch_get_num_kernels(CloScan* scanner, GError** err) {
	(void) scanner; (void) err;
	return CLO_SCAN_BLELLOCH_NUM_KERNELS;
}

/* ref: clo_scan_blelloch.c:251-279 */
static const char* clo_scan_blelloch_get_kernel_name(CloScan* scanner, cl_uint i, GError** err) {
	clo_return_val_if_fail(i < CLO_SCAN_BLELLOCH_NUM_KERNELS, NULL);
	(void) scanner; (void) err;
	switch (i) {
		case 0: return CLO_SCAN_BLELLOCH_KNAME_WGSCAN;
		case 1: return CLO_SCAN_BLELLOCH_KNAME_WGSUMSSCAN;
		default: return CLO_SCAN_BLELLOCH_KNAME_ADDWGSUMS;
	}
}

/* ref: clo_scan_blelloch.c:286-331. Only "workgroupScan" exists as a HIP
 * kernel (the other two jobs are done inside it by look-back). */
static size_t clo_scan_blelloch_get_localmem_usage(CloScan* scanner, cl_uint i, size_t lws_max,
	size_t numel, GError** err) {
	clo_return_val_if_fail(i < CLO_SCAN_BLELLOCH_NUM_KERNELS, 0);
	(void) lws_max; (void) numel; (void) err;
	if (i != 0) return 0;
	return clo_hip_kernel_lds_bytes("scan", (int) clo_scan_get_element_size(scanner), (int) clo_scan_get_sum_size(scanner));
}

/* ref: clo_scan_blelloch.c:335-343 */
const CloScanImplDef clo_scan_blelloch_def = {
	"blelloch",
	clo_scan_blelloch_init,
	clo_scan_blelloch_finalize,
	clo_scan_blelloch_scan_with_device_data,
	clo_scan_blelloch_get_num_kernels,
	clo_scan_blelloch_get_kernel_name,
	clo_scan_blelloch_get_localmem_usage
};

/* not part of the public struct (clo_internal.h) */
const clo_scan_impl_ext clo_scan_blelloch_ext = {
	"blelloch",
	clo_scan_blelloch_scan_chunk,
	clo_scan_blelloch_check_status
};
