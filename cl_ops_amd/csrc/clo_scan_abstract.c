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

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

struct clo_scan {
	CloScanImplDef impl_def;
	CCLContext* ctx;
	CCLProgram* prg;
	CloType elem_type;
	CloType sum_type;
	void* data;
	const clo_scan_impl_ext* ext;  /* private extensions of the implementation (clo_internal.h), or NULL */
	struct scan_pipe_res* pipe;   /* streams / buffers of the pipelined host-data path, created on first use */
};

const clo_scan_impl_ext* clo_scan_impl_ext_find(const char* name) {
	static const clo_scan_impl_ext* const table[] = { &clo_scan_blelloch_ext, NULL };
	for (unsigned i = 0; table[i] != NULL; ++i)
		if (strcmp(table[i]->name, name) == 0) return table[i];
	return NULL;
}

/* Device-side resources of the pipelined clo_scan_with_host_data: creating a
 * stream costs milliseconds, so they are made once per scanner. */
typedef struct scan_pipe_res {
	void* s_in_own;              /* copies in, when the caller gave one queue for both */
	void* s_out;                 /* copies out */
	size_t chunk_cap;            /* elements the chunk buffers hold */
	void* in_dev[2];
	void* out_dev[2];
	void* carry;                 /* two device uint64: carry of even / odd chunks */
	void* in_done[2];
	void* scan_done[2];
} scan_pipe_res;

static void scan_pipe_res_free(scan_pipe_res* r);

/* The first scanner of each pair of types in a process scans two dummy arrays — one for each work-group shape of the
 * kernel (clo_hip_scan.hip: 256 threads below 2^21 elements, 1024 from there on) — so that the code object is loaded and
 * both kernels have had their first launch here, where upstream builds its program, and not inside the first timed scan
 * (see sort_warmup in clo_sort_abstract.c; the first launch of the large shape with 64-bit sums cost 0.13 ms on top of
 * a 0.02 ms scan: a dip at 2^21 in the harness's sweeps). Best effort. */
static void scan_warmup(CloScan* scanner) {
	static unsigned char done[16][16];
	const unsigned e = (unsigned) scanner->elem_type & 15u, t = (unsigned) scanner->sum_type & 15u;
	if (clo_env_no_warmup() || __atomic_exchange_n(&done[e][t], 1, __ATOMIC_RELAXED)) return;   /* (scanners may be made on several threads at once) */
	const size_t sizes[2] = { 20000, (size_t) 1 << 21 };
	for (int i = 0; i < 2; ++i) {
		void* in = calloc(sizes[i], clo_type_sizeof(scanner->elem_type));
		void* out = malloc(sizes[i] * clo_type_sizeof(scanner->sum_type));
		GError* err = NULL;
		if (in && out) clo_scan_with_host_data(scanner, NULL, NULL, in, out, sizes[i], 0, &err);
		if (err) clo_gerror_free(err);
		free(in);
		free(out);
	}
}

CloScan* clo_scan_new(const char* type, const char* options, CCLContext* ctx,
	CloType elem_type, CloType sum_type, const char* compiler_opts, GError** err) {

	clo_return_val_if_fail(type != NULL, NULL);
	clo_return_val_if_fail(ctx != NULL, NULL);
	clo_return_val_if_fail(err == NULL || *err == NULL, NULL);

	/* ref: clo_scan_abstract.c:86-89 */
	const CloScanImplDef* impls[] = { &clo_scan_blelloch_def, NULL };
	CloScan* scanner = NULL;
	GError* err_internal = NULL;
	clo_hip_env_refresh();   /* the environment switches are read when an object is made, never per call */

	for (unsigned i = 0; impls[i] != NULL; ++i) {
		if (strcmp(type, impls[i]->name) != 0) continue;
		scanner = (CloScan*) calloc(1, sizeof(CloScan));
		if (!scanner) break;
		scanner->impl_def = *impls[i];
		scanner->ext = clo_scan_impl_ext_find(impls[i]->name);
		ccl_context_ref(ctx);
		scanner->ctx = ctx;
		scanner->elem_type = elem_type;
		scanner->sum_type = sum_type;

		if (clo_type_sizeof(elem_type) == 0 || clo_type_sizeof(sum_type) == 0) {
			clo_gerror_set(err, CLO_ERROR, CLO_ERROR_UNKNOWN_TYPE, "Unknown element or sum type");
			goto error_handler;
		}
		/* Every pair of types is scanned, as upstream's generic kernel does (clo_scan_abstract.c:122-125):
		 * integer sums at least as wide as integer elements by the single-pass kernel, everything else —
		 * half / float / double sums, floating-point elements into integer sums (every element truncated by
		 * the cast), sums narrower than the elements — by the deterministic reduce-scan-apply kernels
		 * (clo_hip_fscan.hip; clo_hip_scan_is_typed decides). */

		const char* token = scanner->impl_def.init(scanner, options, &err_internal);
		if (err_internal) { clo_gerror_propagate(err, err_internal); goto error_handler; }
		if (!token) {
			clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY, "Scan implementation '%s' failed to initialise", type);
			goto error_handler;
		}
		scanner->prg = ccl_program_new_token(ctx, token, compiler_opts);
		scan_warmup(scanner);
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
	scan_pipe_res_free(scan->pipe);
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

/* ------------------------------------------------------------------ */
/* Pipelined host-data scan. Upstream copies the whole
 * array in, scans, copies the whole array out (clo_scan_abstract.c:290-339):
 * the two PCIe directions never overlap. Here the array goes through in
 * chunks: while chunk k is scanned (a device-resident carry links the
 * chunks), chunk k+1 is on its way in and chunk k-1 on its way out. A copy
 * from/to pageable host memory blocks the calling thread for its duration
 * (measured: hipMemcpyAsync of 512 MiB returns after 9.5 ms), so the copies
 * out are issued by a helper thread; pinning the caller's memory instead
 * costs ~10 ms per GiB, more than the overlap gains.                     */
/* ------------------------------------------------------------------ */
#ifndef CLO_SCAN_PIPE_MIN_NUMEL   /* (the sanitizer build of tests/hoststub shrinks it) */
#define CLO_SCAN_PIPE_MIN_NUMEL ((size_t) 1 << 25)   /* below this one copy in, one scan, one copy out */
#endif
#ifndef CLO_SCAN_PIPE_CHUNK_MAX   /* (the sanitizer build of tests/hoststub shrinks it) */
#define CLO_SCAN_PIPE_CHUNK_MAX ((size_t) 1 << 24)
#endif

/* Chunks of the pipeline: chunks of 2^24 elements (the last one shorter), from 2^25
 * elements on. Never smaller (round 3): a chunk below 2^24 elements is scanned by the
 * small work-group shape (clo_hip_scan.hip) at two thirds of the rate, and with the halves
 * and quarters of rounds 1-2 the harness's table fell from 293 GValues/s at 2^23 elements
 * to 253 at 2^24 and 266 at 2^25 before jumping to 428 at 2^26
 * (profiles/r03_harness_sweep_scan.txt, first collection).
 * Every chunk scan starts on an idle stream (it waits for its copy), which
 * costs ~20 us of launch latency inside its event pair whatever its size, and the
 * reference harness prints the SUM of those event times (benchmarks/
 * clo_scan_bench.c:240-278): with the 2^22-element chunks of round 1 the printed
 * rate fell from 195 GValues/s at 2^23 elements to 172 at 2^24 and 2^25
 * (profiles/r01_harness_sweep_scan.txt). Few, large chunks keep that figure
 * monotone, and the wall time (measured, uint -> uint, host to host: 2^26 elements
 * 6.2 ms with chunks of 2^22 or 2^23, 6.7 with 2^24; 2^28: 23.2-23.8 ms with 2^22
 * .. 2^24) is flat in the chunk size. */
static size_t scan_pipe_chunk(size_t numel) {
	(void) numel;
	return CLO_SCAN_PIPE_CHUNK_MAX;   /* (a multiple of 16-KiB blocks: every chunk starts 16-byte aligned) */
}

static void scan_pipe_res_free(scan_pipe_res* r) {
	if (!r) return;
	if (r->s_in_own) { clo_hip_stream_synchronize(r->s_in_own); clo_hip_stream_destroy(r->s_in_own); }
	if (r->s_out) { clo_hip_stream_synchronize(r->s_out); clo_hip_stream_destroy(r->s_out); }
	for (int i = 0; i < 2; ++i) {
		clo_hip_free(r->in_dev[i]);
		clo_hip_free(r->out_dev[i]);
		clo_hip_event_destroy(r->in_done[i]);
		clo_hip_event_destroy(r->scan_done[i]);
	}
	clo_hip_free(r->carry);
	free(r);
}

static scan_pipe_res* scan_pipe_res_get(CloScan* scanner, size_t chunk, size_t es, size_t ss, int* status) {
	if (scanner->pipe && scanner->pipe->chunk_cap >= chunk) return scanner->pipe;
	if (scanner->pipe) { scan_pipe_res_free(scanner->pipe); scanner->pipe = NULL; }
	scan_pipe_res* r = (scan_pipe_res*) calloc(1, sizeof(*r));
	int st = r ? 0 : CLO_HIP_EARGS;
	if (r) r->chunk_cap = chunk;
	for (int i = 0; i < 2 && st == 0; ++i) {
		st = clo_hip_malloc(&r->in_dev[i], chunk * es);
		if (st == 0) st = clo_hip_malloc(&r->out_dev[i], chunk * ss);
		if (st == 0) st = clo_hip_event_create(&r->in_done[i]);
		if (st == 0) st = clo_hip_event_create(&r->scan_done[i]);
	}
	if (st == 0) st = clo_hip_malloc(&r->carry, 2 * sizeof(uint64_t));
	if (st == 0) st = clo_hip_stream_create(&r->s_out);
	if (st == 0) st = clo_hip_stream_create(&r->s_in_own);
	if (st != 0) { scan_pipe_res_free(r); *status = st; return NULL; }
	scanner->pipe = r;
	return r;
}

typedef struct {
	int device;
	scan_pipe_res* r;
	char* out_host;
	size_t sum_size, chunk, numel;
	pthread_mutex_t mtx;
	pthread_cond_t cv;
	size_t posted, completed;    /* chunks handed to / finished by the helper */
	int abort, status;
} scan_pipe;

static void* scan_pipe_copy_out(void* arg) {
	scan_pipe* p = (scan_pipe*) arg;
	clo_hip_set_device(p->device);
	for (size_t k = 0; ; ++k) {
		pthread_mutex_lock(&p->mtx);
		while (p->posted <= k && !p->abort) pthread_cond_wait(&p->cv, &p->mtx);
		const int stop = p->posted <= k;   /* aborted, or nothing more will come */
		pthread_mutex_unlock(&p->mtx);
		if (stop) break;
		const size_t off = k * p->chunk;
		const size_t cnt = p->numel - off < p->chunk ? p->numel - off : p->chunk;
		int st = clo_hip_event_synchronize(p->r->scan_done[k & 1]);
		if (st == 0) st = clo_hip_memcpy_d2h_async(p->out_host + off * p->sum_size, p->r->out_dev[k & 1], cnt * p->sum_size, p->r->s_out);
		if (st == 0) st = clo_hip_stream_synchronize(p->r->s_out);
		pthread_mutex_lock(&p->mtx);
		if (st != 0 && p->status == 0) p->status = st;
		p->completed = k + 1;
		pthread_cond_broadcast(&p->cv);
		pthread_mutex_unlock(&p->mtx);
		if (off + cnt >= p->numel) break;
	}
	return NULL;
}

static cl_bool scan_with_host_data_pipelined(CloScan* scanner, CCLQueue* cq_exec, CCLQueue* cq_comm,
	const void* data_in, void* data_out, size_t numel, GError** err) {

	const size_t es = clo_type_sizeof(scanner->elem_type), ss = clo_type_sizeof(scanner->sum_type);
	const size_t chunk = scan_pipe_chunk(numel);
	const size_t nchunks = (numel + chunk - 1) / chunk;
	void* s_in = ccl_queue_get_stream(cq_comm);
	void* s_exec = ccl_queue_get_stream(cq_exec);
	scan_pipe p;
	pthread_t helper;
	int helper_started = 0, st = 0;
	cl_bool ok = CL_FALSE;
	const char* what = "pipeline resources (hipMalloc / hipStreamCreate)";

	memset(&p, 0, sizeof(p));
	pthread_mutex_init(&p.mtx, NULL);
	pthread_cond_init(&p.cv, NULL);
	p.sum_size = ss; p.chunk = chunk; p.numel = numel; p.out_host = (char*) data_out;
	if (clo_hip_get_device(&p.device) != 0) p.device = 0;
	scan_pipe_res* r = scan_pipe_res_get(scanner, chunk, es, ss, &st);
	if (!r) goto finish;
	p.r = r;
	if (s_in == s_exec) s_in = r->s_in_own;
	what = "hipMemsetAsync";
	st = clo_hip_memset_async(r->carry, 0, 2 * sizeof(uint64_t), s_exec);
	if (st != 0) goto finish;
	if (pthread_create(&helper, NULL, scan_pipe_copy_out, &p) != 0) { st = CLO_HIP_EARGS; what = "pthread_create"; goto finish; }
	helper_started = 1;

	for (size_t k = 0; k < nchunks; ++k) {
		const int slot = (int) (k & 1);
		const size_t off = k * chunk;
		const size_t cnt = numel - off < chunk ? numel - off : chunk;
		if (k >= 2) {   /* the slot's buffers are free once chunk k-2 is back on the host */
			pthread_mutex_lock(&p.mtx);
			while (p.completed < k - 1 && p.status == 0) pthread_cond_wait(&p.cv, &p.mtx);
			st = p.status;
			pthread_mutex_unlock(&p.mtx);
			if (st != 0) { what = "copy out"; goto finish; }
		}
		what = "hipMemcpyAsync(h2d)";
		st = clo_hip_memcpy_h2d_async(r->in_dev[slot], (const char*) data_in + off * es, cnt * es, s_in);
		if (st == 0) st = clo_hip_event_record(r->in_done[slot], s_in);
		if (st == 0) st = clo_hip_stream_wait_event(s_exec, r->in_done[slot]);
		if (st != 0) goto finish;
		if (!scanner->ext->scan_chunk(scanner, cq_exec, r->in_dev[slot], r->out_dev[slot], cnt,
			(char*) r->carry + slot * sizeof(uint64_t), (char*) r->carry + (slot ^ 1) * sizeof(uint64_t), err)) goto finish;
		what = "hipEventRecord";
		st = clo_hip_event_record(r->scan_done[slot], s_exec);
		if (st != 0) goto finish;
		pthread_mutex_lock(&p.mtx);
		p.posted = k + 1;
		pthread_cond_broadcast(&p.cv);
		pthread_mutex_unlock(&p.mtx);
	}
	pthread_mutex_lock(&p.mtx);
	while (p.completed < nchunks && p.status == 0) pthread_cond_wait(&p.cv, &p.mtx);
	st = p.status;
	pthread_mutex_unlock(&p.mtx);
	what = "copy out";
	ok = st == 0;
	/* every chunk is back on the host, so every scan has completed: did one of
	 * them give up a look-back spin? */
	if (ok && scanner->ext->check_status && !scanner->ext->check_status(scanner, cq_exec, err)) ok = CL_FALSE;

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
		if (s_in) clo_hip_stream_synchronize(s_in);
		clo_hip_stream_synchronize(s_exec);
	}
	pthread_mutex_destroy(&p.mtx);
	pthread_cond_destroy(&p.cv);
	return ok && (err == NULL || *err == NULL);
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

	if (scanner->ext != NULL && scanner->ext->scan_chunk != NULL && !clo_hip_scan_is_typed((int) scanner->elem_type, (int) scanner->sum_type) && numel >= CLO_SCAN_PIPE_MIN_NUMEL && ccl_queue_get_stream(cq_exec) != NULL) {
		status = scan_with_host_data_pipelined(scanner, cq_exec, cq_comm, data_in, data_out, numel, &err_internal);
		if (err_internal) goto error_handler;
		goto finish;
	}

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
	/* the result is on the host: it is only good if no look-back spin gave up */
	if (scanner->ext && scanner->ext->check_status && !scanner->ext->check_status(scanner, cq_exec, &err_internal)) goto error_handler;

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
