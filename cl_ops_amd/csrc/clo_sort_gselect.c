/*
 * clo_sort_gselect.c — host driver of the "gselect" sorter over HIP.
 * Mirrors src/cl_ops/sort/clo_sort_gselect.c:30-243 of the reference: no
 * options, one kernel ("gselect"), no local memory reported, NOT in place —
 * without data_out a temporary is sorted into and copied back (:83-103,
 * :118-127, events "gselect_ndrange" / "gselect_copy"). The launch itself
 * (:105-116, clo_sort_gselect.cl:38-58) is clo_hip_gselect.
 */
#include "clo_sort.h"
#include "clo_internal.h"

typedef struct {
	clo_devbuf tmp;   /* stands in for upstream's per-call data_out buffer */
} clo_sort_gselect_data;

static CCLEvent* clo_sort_gselect_sort_with_device_data(CloSort* sorter, CCLQueue* cq_exec,
	CCLQueue* cq_comm, CCLBuffer* data_in, CCLBuffer* data_out, size_t numel, size_t lws_max,
	GError** err) {

	clo_return_val_if_fail(err == NULL || *err == NULL, NULL);
	clo_return_val_if_fail(cq_exec != NULL, NULL);
	clo_return_val_if_fail(data_in != NULL, NULL);
	(void) lws_max;
	if (cq_comm == NULL) cq_comm = cq_exec;

	clo_sort_gselect_data* data = (clo_sort_gselect_data*) clo_sort_get_data(sorter);
	const CloSortKeySpec* ks = clo_sort_get_key_spec(sorter);
	const size_t bytes = numel * (size_t) ks->elem_size;
	void* stream = ccl_queue_get_stream(cq_exec);
	const int copy_back = data_out == NULL;

	if (bytes > ccl_buffer_get_size(data_in) || (data_out && bytes > ccl_buffer_get_size(data_out))) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "numel (%zu) exceeds the size of the device buffers", numel);
		return NULL;
	}
	if (data_out && ccl_buffer_get_device_ptr(data_out) == ccl_buffer_get_device_ptr(data_in)) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "gselect is not an in-place sort: data_out must differ from data_in");
		return NULL;
	}
	if (copy_back && numel > 0
		&& clo_hip_failed(clo_devbuf_reserve(&data->tmp, bytes), err, "hipMalloc(gselect output)")) return NULL;

	CCLEvent* evt = ccl_queue_begin_command(cq_exec, "gselect_ndrange", err);
	if (!evt) return NULL;
	void* src = ccl_buffer_get_device_ptr(data_in);
	void* dst = copy_back ? data->tmp.ptr : ccl_buffer_get_device_ptr(data_out);
	if (numel > 0) {
		/* compare / get_key outside the ahead-of-time family: the kernel compiled at clo_sort_new with
		 * the two macro bodies pasted in, as upstream does (sort/clo_sort_gselect.cl:46-51) */
		void* jit = clo_sort_get_jit(sorter);
		int st = jit ? clo_hip_bitonic_jit_gselect(jit, src, dst, numel, stream)
			: clo_hip_gselect(src, dst, numel, ks->elem_size, ks->key_shift, ks->key_bits, ks->key_size,
				ks->key_kind, ks->descending, stream);
		if (clo_hip_failed(st, err, "clo_hip_gselect")) { ccl_queue_abort_command(cq_exec, evt); return NULL; }
	}
	if (!ccl_queue_end_command(cq_exec, evt, err)) return NULL;

	if (copy_back && numel > 0) {
		/* ref: :118-127 — copy on the comm queue after the sort */
		CCLEventWaitList ewl = NULL;
		CCLBuffer* tmp = ccl_buffer_new_from_device_ptr(ccl_queue_get_context(cq_comm, NULL), data->tmp.ptr, bytes, err);
		if (!tmp) return NULL;
		evt = ccl_buffer_enqueue_copy(tmp, data_in, cq_comm, 0, 0, bytes, ccl_ewl(&ewl, evt, NULL), err);
		ccl_event_wait_list_clear(&ewl);
		ccl_buffer_destroy(tmp);
		if (!evt) return NULL;
		ccl_event_set_name(evt, "gselect_copy");
		/* the cached temporary is reused by the next call on the exec queue */
		if (cq_comm != cq_exec && !ccl_queue_finish(cq_comm, err)) return NULL;
	}
	return evt;
}

/* ref: clo_sort_gselect.c:140-155 — options are ignored. */
static const char* clo_sort_gselect_init(CloSort* sorter, const char* options, GError** err) {
	clo_return_val_if_fail(err == NULL || *err == NULL, NULL);
	(void) options;
	clo_sort_gselect_data* data = (clo_sort_gselect_data*) calloc(1, sizeof(*data));
	if (!data) return NULL;
	clo_sort_set_data(sorter, data);
	return "gselect:hip";
}

static void clo_sort_gselect_finalize(CloSort* sorter) {
	clo_sort_gselect_data* data = (clo_sort_gselect_data*) clo_sort_get_data(sorter);
	if (data) {
		clo_devbuf_release(&data->tmp);
		free(data);
	}
	clo_sort_set_data(sorter, NULL);
}

static cl_uint clo_sort_gselect_get_num_kernels(CloSort* sorter, GError** err) {
	(void) sorter; (void) err;
	return 1;
}

static const char* clo_sort_gselect_get_kernel_name(CloSort* sorter, cl_uint i, GError** err) {
	clo_return_val_if_fail(i == 0, NULL);
	(void) sorter; (void) err;
	return CLO_SORT_GSELECT_KNAME;
}

/* ref: clo_sort_gselect.c:212-231 — upstream's kernel uses no local memory and
 * reports 0; the HIP kernel's LDS stage is an implementation detail of a
 * fixed size, reported here because that is what the getter is for. */
static size_t clo_sort_gselect_get_localmem_usage(CloSort* sorter, cl_uint i, size_t lws_max,
	size_t numel, GError** err) {
	clo_return_val_if_fail(i == 0, 0);
	(void) sorter; (void) lws_max; (void) numel; (void) err;
	return clo_hip_kernel_lds_bytes("gselect", 0, 0);
}

/* ref: clo_sort_gselect.c:234-243 */
const CloSortImplDef clo_sort_gselect_def = {
	"gselect",
	CL_FALSE,
	clo_sort_gselect_init,
	clo_sort_gselect_finalize,
	clo_sort_gselect_sort_with_device_data,
	clo_sort_gselect_get_num_kernels,
	clo_sort_gselect_get_kernel_name,
	clo_sort_gselect_get_localmem_usage
};
