/*
 * clo_sort_abitonic.c — host driver of the "abitonic" sorter over HIP.
 *
 * Mirrors src/cl_ops/sort/clo_sort_abitonic.c of the reference: options
 * `minps`, `maxps`, `maxsfs` with the same validation and messages (:486-543),
 * the 26 kernel names (:clo_sort_abitonic.in.h:64-106) and the in-place
 * contract (:708-717). The per-step strategy table and launch loop
 * (:58-313,401-432) are replaced by clo_hip_bitonic_tiled, which applies the
 * same network with two tile kernels; the three options only steer upstream's
 * choice among its 26 kernels, which changes no result, so they are validated,
 * stored and otherwise unused.
 */
#include "clo_sort_bitonic_common.h"

#include <limits.h>
#include <string.h>

typedef struct {
	clo_bitonic_state state;  /* must stay first */
	cl_uint max_inkrnl_stps;
	cl_uint min_inkrnl_stps;
	cl_uint max_inkrnl_sfs;
} clo_sort_abitonic_data;

/* ref: clo_sort_abitonic.in.h:64-113 */
static const char* clo_sort_abitonic_knames[CLO_SORT_ABITONIC_NUM_KERNELS] = {
	"abit_any",
	"abit_local_s2", "abit_local_s3", "abit_local_s4", "abit_local_s5", "abit_local_s6",
	"abit_local_s7", "abit_local_s8", "abit_local_s9", "abit_local_s10", "abit_local_s11",
	"abit_priv_2s4v", "abit_priv_3s8v", "abit_priv_4s16v",
	"abit_hyb_s4_2s4v", "abit_hyb_s6_2s4v", "abit_hyb_s8_2s4v", "abit_hyb_s10_2s4v", "abit_hyb_s12_2s4v",
	"abit_hyb_s3_3s8v", "abit_hyb_s6_3s8v", "abit_hyb_s9_3s8v", "abit_hyb_s12_3s8v",
	"abit_hyb_s4_4s16v", "abit_hyb_s8_4s16v", "abit_hyb_s12_4s16v"
};

static CCLEvent* clo_sort_abitonic_sort_with_device_data(CloSort* sorter, CCLQueue* cq_exec,
	CCLQueue* cq_comm, CCLBuffer* data_in, CCLBuffer* data_out, size_t numel, size_t lws_max,
	GError** err) {
	(void) lws_max;
	clo_sort_abitonic_data* data = (clo_sort_abitonic_data*) clo_sort_get_data(sorter);
	return clo_bitonic_run(sorter, &data->state, 1, 0, "abit_tile", "abit_copy",
		cq_exec, cq_comm, data_in, data_out, numel, err);
}

/* ref: clo_sort_abitonic.c:507-536 */
static int abitonic_option(const char* key, const char* value, const char* token, void* user, GError** err) {
	clo_sort_abitonic_data* data = (clo_sort_abitonic_data*) user;
	(void) token;
	cl_uint v = (cl_uint) atoi(value);
	if (strcmp(key, "minps") == 0) {
		if (v > 4 || v < 1) {
			clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "Option 'minps' must be between 1 and 4.");
			return 0;
		}
		data->min_inkrnl_stps = v;
	} else if (strcmp(key, "maxps") == 0) {
		if (v > 4 || v < 1) {
			clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "Option 'maxps' must be between 1 and 4.");
			return 0;
		}
		data->max_inkrnl_stps = v;
	} else if (strcmp(key, "maxsfs") == 0) {
		data->max_inkrnl_sfs = v;
	} else {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "Invalid option key '%s' for abitonic sort.", key);
		return 0;
	}
	return 1;
}

/* ref: clo_sort_abitonic.c:458-567 */
static const char* clo_sort_abitonic_init(CloSort* sorter, const char* options, GError** err) {
	clo_return_val_if_fail(err == NULL || *err == NULL, NULL);
	clo_sort_abitonic_data* data = (clo_sort_abitonic_data*) calloc(1, sizeof(*data));
	if (!data) return NULL;
	data->max_inkrnl_stps = 4;
	data->min_inkrnl_stps = 1;
	data->max_inkrnl_sfs = UINT_MAX;
	if (!clo_parse_options(options, abitonic_option, data, "abitonic", err)) {
		free(data);
		return NULL;
	}
	if (data->max_inkrnl_stps < data->min_inkrnl_stps) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "'minps' (%d) must be less or equal than 'maxps' (%d).",
			(int) data->min_inkrnl_stps, (int) data->max_inkrnl_stps);
		free(data);
		return NULL;
	}
	clo_sort_set_data(sorter, data);
	return "abitonic:hip";
}

static void clo_sort_abitonic_finalize(CloSort* sorter) {
	clo_sort_abitonic_data* data = (clo_sort_abitonic_data*) clo_sort_get_data(sorter);
	if (data) {
		clo_bitonic_state_release(&data->state);
		free(data);
	}
	clo_sort_set_data(sorter, NULL);
}

static cl_uint clo_sort_abitonic_get_num_kernels(CloSort* sorter, GError** err) {
	(void) sorter; (void) err;
	return CLO_SORT_ABITONIC_NUM_KERNELS;
}

static const char* clo_sort_abitonic_get_kernel_name(CloSort* sorter, cl_uint i, GError** err) {
	clo_return_val_if_fail(i < CLO_SORT_ABITONIC_NUM_KERNELS, NULL);
	(void) sorter; (void) err;
	return clo_sort_abitonic_knames[i];
}

/* ref: clo_sort_abitonic.c:617-704. The 26 names are upstream's list, kept so that code which walks
 * clo_sort_get_num_kernels / _get_kernel_name keeps working; the HIP schedule launches kernels of its own (their names
 * appear on a profiling queue: abit_presort, abit_merge, abit_strided, abit_strided2). "any" and "priv" names stand
 * for the register-only kernels (no LDS); "local" and "hyb" names for the LDS tile kernels: the static LDS of the one
 * that `numel` selects (clo_hip_bitonic_lds_bytes: nothing below 32 elements, the run-time-schedule tile kernel up
 * to one tile, the compile-time-schedule kernels above). lws_max plays no part (include/clo_sort.h). */
static size_t clo_sort_abitonic_get_localmem_usage(CloSort* sorter, cl_uint i, size_t lws_max,
	size_t numel, GError** err) {
	clo_return_val_if_fail(i < CLO_SORT_ABITONIC_NUM_KERNELS, 0);
	(void) lws_max; (void) err;
	const char* name = clo_sort_abitonic_knames[i];
	if (strcmp(name, "abit_any") == 0 || strstr(name, "priv")) return 0;
	void* jit = clo_sort_get_jit(sorter);
	if (jit) return clo_hip_bitonic_jit_lds_bytes(jit, numel, 1);
	return clo_hip_bitonic_lds_bytes(numel, (int) clo_sort_get_element_size(sorter), 1);
}

/* ref: clo_sort_abitonic.c:708-717 */
const CloSortImplDef clo_sort_abitonic_def = {
	"abitonic",
	CL_TRUE,
	clo_sort_abitonic_init,
	clo_sort_abitonic_finalize,
	clo_sort_abitonic_sort_with_device_data,
	clo_sort_abitonic_get_num_kernels,
	clo_sort_abitonic_get_kernel_name,
	clo_sort_abitonic_get_localmem_usage
};
