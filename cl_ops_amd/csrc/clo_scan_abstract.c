/*
 * clo_scan_abstract.c — the CloScan object and its dispatch. Follows the
 * behaviour of src/cl_ops/scan/clo_scan_abstract.c:74-569 of the reference
 * (constructor with by-value types, destructor, device-data / host-data entry
 * points, getters); the -DCLO_SCAN_ELEM_TYPE/-DCLO_SCAN_SUM_TYPE JIT options
 * (:122-125) become the (elem_size, signedness, sum_size) arguments of the
 * ahead-of-time HIP kernel.
 */
#include "clo_scan.h"
#include "clo_internal.h"

#include <string.h>

struct clo_scan {
	CloScanImplDef impl_def;
	CCLContext* ctx;
	CCLProgram* prg;
	CloType elem_type;
	CloType sum_type;
	void* data;
};

CloScan* clo_scan_new(const char* type, const char* options, CCLContext* ctx,
	CloType elem_type, CloType sum_type, const char* compiler_opts, GError** err) {

	clo_return_val_if_fail(type != NULL, NULL);
	clo_return_val_if_fail(ctx != NULL, NULL);
	clo_return_val_if_fail(err == NULL || *err == NULL, NULL);

	/* ref: clo_scan_abstract.c:86-89 */
	const CloScanImplDef* impls[] = { &clo_scan_blelloch_def, NULL };
	CloScan* scanner = NULL;
	GError* err_internal = NULL;

	for (unsigned i = 0; impls[i] != NULL; ++i) {
		if (strcmp(type, impls[i]->name) != 0) continue;
		scanner = (CloScan*) calloc(1, sizeof(CloScan));
		if (!scanner) break;
		scanner->impl_def = *impls[i];
		ccl_context_ref(ctx);
		scanner->ctx = ctx;
		scanner->elem_type = elem_type;
		scanner->sum_type = sum_type;

		if (clo_type_sizeof(elem_type) == 0 || clo_type_sizeof(sum_type) == 0) {
			clo_gerror_set(err, CLO_ERROR, CLO_ERROR_UNKNOWN_TYPE, "Unknown element or sum type");
			goto error_handler;
		}
		if (clo_type_is_float(elem_type) || clo_type_is_float(sum_type)) {
			clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "Floating point scans are not part of the HIP build");
			goto error_handler;
		}
		if (clo_type_sizeof(sum_type) < clo_type_sizeof(elem_type)) {
			clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "The sum type must be at least as wide as the element type");
			goto error_handler;
		}

		const char* token = scanner->impl_def.init(scanner, options, &err_internal);
		if (err_internal) { clo_gerror_propagate(err, err_internal); goto error_handler; }
		if (!token) {
			clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY, "Scan implementation '%s' failed to initialise", type);
			goto error_handler;
		}
		scanner->prg = ccl_program_new_token(ctx, token, compiler_opts);
		break;
	}

	if (scanner == NULL) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_IMPL_NOT_FOUND,
			"The requested scan implementation, '%s', was not found.", type);
	}
	return scanner;

error_handler:
	if (scanner) {
		if (scanner->data) scanner->impl_def.finalize(scanner);
		ccl_context_unref(scanner->ctx);
		ccl_program_destroy(scanner->prg);
		free(scanner);
	}
	return NULL;
}

void clo_scan_destroy(CloScan* scan) {
	clo_return_if_fail(scan != NULL);
	scan->impl_def.finalize(scan);
	if (scan->ctx) ccl_context_unref(scan->ctx);
	if (scan->prg) ccl_program_destroy(scan->prg);
	free(scan);
}

CCLEvent* clo_scan_with_device_data(CloScan* scanner, CCLQueue* cq_exec, CCLQueue* cq_comm,
	CCLBuffer* data_in, CCLBuffer* data_out, size_t numel, size_t lws_max, GError** err) {
	clo_return_val_if_fail(scanner != NULL, NULL);
	clo_return_val_if_fail(err == NULL || *err == NULL, NULL);
	clo_return_val_if_fail(cq_exec != NULL, NULL);
	return scanner->impl_def.scan_with_device_data(scanner, cq_exec, cq_comm, data_in, data_out, numel, lws_max, err);
}

/* ref: clo_scan_abstract.c:255-362 */
cl_bool clo_scan_with_host_data(CloScan* scanner, CCLQueue* cq_exec, CCLQueue* cq_comm,
	void* data_in, void* data_out, size_t numel, size_t lws_max, GError** err) {

	clo_return_val_if_fail(scanner != NULL, CL_FALSE);
	clo_return_val_if_fail(err == NULL || *err == NULL, CL_FALSE);

	cl_bool status = CL_FALSE;
	CCLEvent* evt = NULL;
	CCLBuffer* data_in_dev = NULL;
	CCLBuffer* data_out_dev = NULL;
	CCLQueue* intern_queue = NULL;
	CCLEventWaitList ewl = NULL;
	GError* err_internal = NULL;
	const size_t data_in_size = numel * clo_type_sizeof(scanner->elem_type);
	const size_t data_out_size = numel * clo_type_sizeof(scanner->sum_type);

	if (cq_exec == NULL) {
		CCLDevice* dev = ccl_context_get_device(scanner->ctx, 0, &err_internal);
		if (err_internal) goto error_handler;
		intern_queue = ccl_queue_new(scanner->ctx, dev, 0, &err_internal);
		if (err_internal) goto error_handler;
		cq_exec = intern_queue;
	}
	if (cq_comm == NULL) cq_comm = cq_exec;

	data_in_dev = ccl_buffer_new(scanner->ctx, CL_MEM_READ_ONLY, data_in_size, NULL, &err_internal);
	if (err_internal) goto error_handler;
	data_out_dev = ccl_buffer_new(scanner->ctx, CL_MEM_READ_WRITE, data_out_size, NULL, &err_internal);
	if (err_internal) goto error_handler;

	evt = ccl_buffer_enqueue_write(data_in_dev, cq_comm, CL_FALSE, 0, data_in_size, data_in, NULL, &err_internal);
	if (err_internal) goto error_handler;
	ccl_event_set_name(evt, "clo_scan_write");
	ccl_event_wait(ccl_ewl(&ewl, evt, NULL), &err_internal);
	if (err_internal) goto error_handler;

	evt = scanner->impl_def.scan_with_device_data(scanner, cq_exec, cq_comm, data_in_dev, data_out_dev,
		numel, lws_max, &err_internal);
	if (err_internal) goto error_handler;

	evt = ccl_buffer_enqueue_read(data_out_dev, cq_comm, CL_FALSE, 0, data_out_size, data_out,
		evt ? ccl_ewl(&ewl, evt, NULL) : NULL, &err_internal);
	if (err_internal) goto error_handler;
	ccl_event_set_name(evt, "clo_scan_read");
	ccl_event_wait(ccl_ewl(&ewl, evt, NULL), &err_internal);
	if (err_internal) goto error_handler;

	status = CL_TRUE;
	goto finish;

error_handler:
	clo_gerror_propagate(err, err_internal);
	status = CL_FALSE;

finish:
	ccl_event_wait_list_clear(&ewl);
	if (data_in_dev) ccl_buffer_destroy(data_in_dev);
	if (data_out_dev) ccl_buffer_destroy(data_out_dev);
	if (intern_queue) ccl_queue_destroy(intern_queue);
	return status;
}

/* ---- getters, ref: clo_scan_abstract.c:372-569 ---- */

CCLContext* clo_scan_get_context(CloScan* scanner) {
	clo_return_val_if_fail(scanner != NULL, NULL);
	return scanner->ctx;
}

CCLProgram* clo_scan_get_program(CloScan* scanner) {
	clo_return_val_if_fail(scanner != NULL, NULL);
	return scanner->prg;
}

CloType clo_scan_get_elem_type(CloScan* scanner) {
	clo_return_val_if_fail(scanner != NULL, (CloType) -1);
	return scanner->elem_type;
}

size_t clo_scan_get_element_size(CloScan* scanner) {
	clo_return_val_if_fail(scanner != NULL, 0);
	return clo_type_sizeof(scanner->elem_type);
}

CloType clo_scan_get_sum_type(CloScan* scanner) {
	clo_return_val_if_fail(scanner != NULL, (CloType) -1);
	return scanner->sum_type;
}

size_t clo_scan_get_sum_size(CloScan* scanner) {
	clo_return_val_if_fail(scanner != NULL, 0);
	return clo_type_sizeof(scanner->sum_type);
}

void* clo_scan_get_data(CloScan* scanner) {
	clo_return_val_if_fail(scanner != NULL, NULL);
	return scanner->data;
}

void clo_scan_set_data(CloScan* scanner, void* data) {
	clo_return_if_fail(scanner != NULL);
	scanner->data = data;
}

cl_uint clo_scan_get_num_kernels(CloScan* scanner, GError** err) {
	clo_return_val_if_fail(scanner != NULL, 0);
	return scanner->impl_def.get_num_kernels(scanner, err);
}

const char* clo_scan_get_kernel_name(CloScan* scanner, cl_uint i, GError** err) {
	clo_return_val_if_fail(scanner != NULL, NULL);
	return scanner->impl_def.get_kernel_name(scanner, i, err);
}

size_t clo_scan_get_localmem_usage(CloScan* scanner, cl_uint i, size_t lws_max, size_t numel, GError** err) {
	clo_return_val_if_fail(scanner != NULL, 0);
	return scanner->impl_def.get_localmem_usage(scanner, i, lws_max, numel, err);
}
