/*
 * clo_sort_bitonic_common.c — what the two bitonic drivers share: buffer
 * choice (in place / copy first, as src/cl_ops/sort/clo_sort_sbitonic.c:83-97
 * and clo_sort_abitonic.c:359-375 upstream), power-of-two padding, and the call
 * into the C-ABI (clo_hip_bitonic_simple / clo_hip_bitonic_tiled).
 */
#include "clo_sort_bitonic_common.h"

void clo_bitonic_state_release(clo_bitonic_state* state) {
	if (!state) return;
	clo_graph_cache_release(&state->graph);
	clo_devbuf_release(&state->padded);
	clo_stream_guard_release(&state->guard);
}

typedef struct {
	const CloSortKeySpec* ks;
	void* work;
	size_t numel;
	int tiled;
	int* launches;
} bitonic_call;

static int bitonic_enqueue(void* user, void* stream) {
	bitonic_call* c = (bitonic_call*) user;
	const CloSortKeySpec* ks = c->ks;
	return c->tiled
		? clo_hip_bitonic_tiled(c->work, c->numel, ks->elem_size, ks->key_shift, ks->key_bits, ks->key_size, ks->key_kind, ks->descending, c->launches, stream)
		: clo_hip_bitonic_simple(c->work, c->numel, ks->elem_size, ks->key_shift, ks->key_bits, ks->key_size, ks->key_kind, ks->descending, c->launches, stream);
}

CCLEvent* clo_bitonic_run(CloSort* sorter, clo_bitonic_state* state, int tiled, int one_name, const char* evt_name,
	const char* copy_evt_name, CCLQueue* cq_exec, CCLQueue* cq_comm, CCLBuffer* data_in,
	CCLBuffer* data_out, size_t numel, GError** err) {

	clo_return_val_if_fail(err == NULL || *err == NULL, NULL);
	clo_return_val_if_fail(cq_exec != NULL, NULL);
	clo_return_val_if_fail(data_in != NULL, NULL);

	const CloSortKeySpec* ks = clo_sort_get_key_spec(sorter);
	const size_t bytes = numel * (size_t) ks->elem_size;
	void* stream = ccl_queue_get_stream(cq_exec);
	CCLEventWaitList ewl = NULL;
	CCLEvent* evt = NULL;

	if (cq_comm == NULL) cq_comm = cq_exec;
	if (bytes > ccl_buffer_get_size(data_in) || (data_out && bytes > ccl_buffer_get_size(data_out))) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "numel (%zu) exceeds the size of the device buffers", numel);
		return NULL;
	}
	if (numel > 0x80000000ull) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "numel must not exceed 2^31");
		return NULL;
	}

	/* Sort in data_out after copying data_in into it, or directly in data_in
	 * (ref: clo_sort_sbitonic.c:83-97). The copy travels on cq_comm and the
	 * kernels wait for it. */
	CCLBuffer* target = data_in;
	if (data_out != NULL && data_out != data_in) {
		evt = ccl_buffer_enqueue_copy(data_in, data_out, cq_comm, 0, 0, bytes, NULL, err);
		if (!evt) return NULL;
		ccl_event_set_name(evt, copy_evt_name);
		ccl_event_wait_list_add(&ewl, evt, NULL);
		target = data_out;
	}
	if (!ccl_queue_wait_for(cq_exec, &ewl, err)) { ccl_event_wait_list_clear(&ewl); return NULL; }
	ccl_event_wait_list_clear(&ewl);

	/* a profiling queue gets one event per launch (upstream: clo_sort_sbitonic.c:115
	 * "sbitonic_ndrange", clo_sort_abitonic.c:426 the step's kernel name) */
	static const clo_kname knames[] = {
		{ "bitonic_step", "sbitonic_ndrange" }, { "bitonic_presort", "abit_presort" }, { "bitonic_tile", "abit_merge" },
		{ "bitonic_strided", "abit_strided" }, { "bitonic_strided2", "abit_strided2" }
	};
	/* (sbitonic: every launch under the one name upstream gives its events, whatever schedule runs) */
	static const clo_kname knames_one[] = {
		{ "bitonic_step", "sbitonic_ndrange" }, { "bitonic_presort", "sbitonic_ndrange" }, { "bitonic_tile", "sbitonic_ndrange" },
		{ "bitonic_strided", "sbitonic_ndrange" }, { "bitonic_strided2", "sbitonic_ndrange" }
	};
	const int per_kernel = ccl_queue_is_profiling(cq_exec) && numel > 1;
	clo_kernel_events ke;
	if (per_kernel) {
		clo_kernel_events_install(&ke, cq_exec, one_name ? knames_one : knames, sizeof(knames) / sizeof(knames[0]), evt_name);
	} else {
		evt = ccl_queue_begin_command(cq_exec, evt_name, err);
		if (!evt) return NULL;
	}
#define BITONIC_FAIL() do { if (per_kernel) clo_kernel_events_remove(&ke, NULL); else ccl_queue_abort_command(cq_exec, evt); return NULL; } while (0)

	void* jit = clo_sort_get_jit(sorter);
	if (numel > 1 && jit != NULL) {
		/* kernels specialised at run time for the user's compare / get_key */
		int launches = 0;
		int st = clo_hip_bitonic_jit_sort(jit, ccl_buffer_get_device_ptr(target), numel, tiled, &launches, stream);
		if (clo_hip_failed(st, err, "clo_hip_bitonic_jit_sort")) BITONIC_FAIL();
		clo_debug("%s (jit): numel=%zu launches=%d", evt_name, numel, launches);
	} else if (numel > 1) {
		const size_t padded = clo_hip_bitonic_padded_numel(numel);
		void* work = ccl_buffer_get_device_ptr(target);
		int use_pad = 0;
		if (padded != numel && (ks->key_shift != 0 || ks->key_bits != 8 * ks->elem_size)) {
			/* Upstream handles powers of two only (its kernels have no bounds). Padding with a
			 * sentinel is safe when ties are invisible, i.e. whole-element keys (below); a key
			 * that is part of the element takes the network's flip form, in place, with the
			 * comparators that reach past numel skipped (clo_hip_bitonic_any: one launch per step). */
			int launches = 0;
			int st = clo_hip_bitonic_any(work, numel, ks->elem_size, ks->key_shift, ks->key_bits, ks->key_size, ks->key_kind,
				ks->descending, &launches, stream);
			if (clo_hip_failed(st, err, "clo_hip_bitonic_any")) BITONIC_FAIL();
			clo_debug("%s (any numel): numel=%zu launches=%d", evt_name, numel, launches);
		} else {
		if (padded != numel) {
			if (clo_hip_failed(clo_stream_guard_enter(&state->guard, cq_exec), err, "hipStreamWaitEvent")) BITONIC_FAIL();
			if (clo_hip_failed(clo_devbuf_reserve(&state->padded, padded * (size_t) ks->elem_size), err, "hipMalloc(bitonic pad)")) BITONIC_FAIL();
			if (clo_hip_failed(clo_hip_memcpy_d2d_async(state->padded.ptr, work, bytes, stream), err, "hipMemcpyAsync")) BITONIC_FAIL();
			work = state->padded.ptr;
			use_pad = 1;
		}
		/* graph replay only for the one-launch-per-step schedule (abitonic's 33
		 * launches of ~0.1 ms measured no gain) and only in place of the caller's buffer */
		int launches = state->launches;
		bitonic_call call = { ks, work, numel, tiled, &launches };
		int st = clo_graph_cache_run(&state->graph, !tiled && !use_pad && !per_kernel, work, NULL, numel, tiled, stream, bitonic_enqueue, &call);
		if (clo_hip_failed(st, err, tiled ? "clo_hip_bitonic_tiled" : "clo_hip_bitonic_simple")) BITONIC_FAIL();
		state->launches = launches;
		clo_debug("%s: numel=%zu padded=%zu launches=%d", evt_name, numel, padded, launches);
		if (use_pad) {
			if (clo_hip_failed(clo_hip_memcpy_d2d_async(ccl_buffer_get_device_ptr(target), work, bytes, stream), err, "hipMemcpyAsync")) BITONIC_FAIL();
		}
		}
	}
#undef BITONIC_FAIL

	if (per_kernel) {
		GError* e2 = NULL;
		CCLEvent* last = clo_kernel_events_remove(&ke, &e2);
		if (e2) { clo_gerror_propagate(err, e2); return NULL; }
		if (last) return last;
		evt = ccl_queue_begin_command(cq_exec, evt_name, err);   /* (no launch was made) */
		if (!evt) return NULL;
	}
	if (!ccl_queue_end_command(cq_exec, evt, err)) { ccl_queue_abort_command(cq_exec, evt); return NULL; }
	return evt;
}
