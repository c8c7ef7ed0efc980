/*
 * clo_sort_sbitonic.c — host driver of the "sbitonic" sorter over HIP.
 * Mirrors the INTERFACE of src/cl_ops/sort/clo_sort_sbitonic.c:31-233 of the reference: no
 * options, one kernel name ("sbitonic"), in place. Upstream runs the
 * bitonic network one launch per (stage, step) (:102-118); the network — the sequence of
 * compare-exchanges, and with it the result, ties included — does not depend on how its steps
 * are grouped into launches, so the steps run in the tiled schedule abitonic uses (registers and
 * LDS, clo_hip_bitonic_tiled: 2^16 keys in 5 launches instead of 136, 0.24 -> 0.05 ms). The
 * one-launch-per-step schedule (clo_hip_bitonic_simple, replayed from a hipGraph) stays behind
 * CLO_SBITONIC_STEPS=1 (tests; A/B runs).
 */
#include "clo_sort_bitonic_common.h"

static CCLEvent* clo_sort_sbitonic_sort_with_device_data(CloSort* sorter, CCLQueue* cq_exec,
	CCLQueue* cq_comm, CCLBuffer* data_in, CCLBuffer* data_out, size_t numel, size_t lws_max,
	GError** err) {
	(void) lws_max;
	clo_bitonic_state* state = (clo_bitonic_state*) clo_sort_get_data(sorter);
	return clo_bitonic_run(sorter, state, !state->steps, 1,
		"sbitonic_ndrange", "sbitonic_copy", cq_exec, cq_comm, data_in, data_out, numel, err);
}

/* ref: clo_sort_sbitonic.c:140-155 — options are ignored. */
static const char* clo_sort_sbitonic_init(CloSort* sorter, const char* options, GError** err) {
	clo_return_val_if_fail(err == NULL || *err == NULL, NULL);
	(void) options;
	clo_bitonic_state* state = (clo_bitonic_state*) calloc(1, sizeof(*state));
	if (!state) return NULL;
	state->steps = clo_env_flag("CLO_SBITONIC_STEPS") == 1;
	clo_sort_set_data(sorter, state);
	return "sbitonic:hip";
}

static void clo_sort_sbitonic_finalize(CloSort* sorter) {
	clo_bitonic_state* state = (clo_bitonic_state*) clo_sort_get_data(sorter);
	if (state) {
		clo_bitonic_state_release(state);
		free(state);
	}
	clo_sort_set_data(sorter, NULL);
}

static cl_uint clo_sort_sbitonic_get_num_kernels(CloSort* sorter, GError** err) {
	(void) sorter; (void) err;
	return 1;
}

static const char* clo_sort_sbitonic_get_kernel_name(CloSort* sorter, cl_uint i, GError** err) {
	clo_return_val_if_fail(i == 0, NULL);
	(void) sorter; (void) err;
	return CLO_SORT_SBITONIC_KNAME;
}

/* ref: clo_sort_sbitonic.c:206-222 — upstream's one kernel uses no local memory and says so. Here the steps run in the
 * tiled schedule (see the head of this file), whose kernels stage their tile in LDS: the answer is the static LDS of
 * the kernel that `numel` selects (nothing below 32 elements or on the one-launch-per-step schedule, the tile kernel
 * above: include/clo_hip.h, clo_hip_bitonic_lds_bytes) — what the launches really hold, not upstream's 0. */
static size_t clo_sort_sbitonic_get_localmem_usage(CloSort* sorter, cl_uint i, size_t lws_max,
	size_t numel, GError** err) {
	clo_return_val_if_fail(i == 0, 0);
	(void) lws_max; (void) err;
	clo_bitonic_state* state = (clo_bitonic_state*) clo_sort_get_data(sorter);
	const int tiled = !(state && state->steps);
	void* jit = clo_sort_get_jit(sorter);
	if (jit) return clo_hip_bitonic_jit_lds_bytes(jit, numel, tiled);
	return clo_hip_bitonic_lds_bytes(numel, (int) clo_sort_get_element_size(sorter), tiled);
}

/* ref: clo_sort_sbitonic.c:224-233 */
const CloSortImplDef clo_sort_sbitonic_def = {
	"sbitonic",
	CL_TRUE,
	clo_sort_sbitonic_init,
	clo_sort_sbitonic_finalize,
	clo_sort_sbitonic_sort_with_device_data,
	clo_sort_sbitonic_get_num_kernels,
	clo_sort_sbitonic_get_kernel_name,
	clo_sort_sbitonic_get_localmem_usage
};
