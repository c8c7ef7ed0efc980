/*
 * clo_ccl.c — the cf4ocl2 object subset of include/clo_ccl.h over the thin HIP
 * C-ABI (clo_hip.h). Plain C; no OpenCL anywhere.
 */
#include "clo_ccl.h"
#include "clo_common.h"
#include "clo_hip.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef CLO_USE_GLIB
#define clo_gerror_set g_set_error
#define CLO_QUARK(s) g_quark_from_static_string(s)
#else
#define CLO_QUARK(s) clo_quark_from_string(s)
#endif

struct ccl_device {
	int index;
	clo_hip_device_props props;
};

struct ccl_context {
	int refcount;
	struct ccl_device dev;
};

struct ccl_event {
	struct ccl_event* next;  /* queue-owned list */
	struct ccl_queue* cq;    /* the queue that owns it */
	char name[48];
	void* start;             /* hip events; start only when profiling */
	void* end;               /* may be NULL until somebody needs it (lazy, below) */
	int start_borrowed;      /* start is the `end` of the command before (a chain of kernels): not ours to destroy */
	int recorded;            /* `end` has been recorded on the queue's stream */
};

/* A device status word some kernel of this queue's commands may raise (the
 * bounded look-back spins), shared between its owner (a scanner) and the
 * queues that have run its commands. */
struct clo_status_cell {
	int refs;
	void* dev_word;   /* NULL once the owner has released the memory */
	int tripped;      /* set when a check found the word raised; the owner re-initialises its workspace */
};

#define CCL_QUEUE_MAX_CELLS 8
/* Events a queue WITHOUT profiling keeps distinct (the most recent ones; a queue
 * with profiling keeps all of them until ccl_prof_calc / ccl_queue_gc). Older
 * ones are RETIRED, not freed: the struct and its hip event go to a spare list
 * and serve a later command of the same queue. A CCLEvent* the caller still
 * holds therefore stays valid memory until ccl_queue_gc / ccl_queue_destroy, as
 * in cf4ocl2; waiting on such a stale handle waits for a LATER command of the
 * same in-order queue, which implies the one it was handed out for. */
#define CCL_QUEUE_KEEP_EVENTS 64

struct ccl_queue {
	CCLContext* ctx;
	void* stream;
	int owns_stream;
	int profiling;
	struct ccl_event* events;  /* most recent first */
	struct ccl_event* first;   /* oldest since last gc */
	struct ccl_event* spare;   /* retired events of a queue without profiling, reused by later commands */
	size_t nevents;
	struct clo_status_cell* cells[CCL_QUEUE_MAX_CELLS];
	unsigned char armed[CCL_QUEUE_MAX_CELLS];   /* a command that may raise the word has been enqueued (or waited for) since the last check */
	int ncells;
	int refs;     /* the caller's + one per sorter / scanner whose last call ran here (clo_queue_hold) */
	int closed;   /* ccl_queue_destroy has run: synchronised, stream gone; only the struct is still held */
};

struct ccl_buffer {
	CCLContext* ctx;
	void* dptr;
	size_t size;
	int owns;
};

struct ccl_program {
	char* what;
	char* options;
};

struct ccl_event_wait_list {
	size_t n, cap;
	CCLEvent** evts;
};

#define CCL_PROF_MAX_QUEUES 8
struct ccl_prof {
	CCLQueue* queues[CCL_PROF_MAX_QUEUES];
	int nqueues;
	cl_ulong duration_ns;
	CCLProfAgg* aggs;     /* per event name, filled by ccl_prof_calc */
	char (*agg_names)[48];
	size_t naggs, cap, iter;
};

GQuark ccl_hip_error_quark(void) { return CLO_QUARK("ccl-hip-error-quark"); }

/* Returns 1 (and sets err) when a clo_hip_* status is a failure. */
static int hip_failed(int st, GError** err, const char* what) {
	if (st == 0) return 0;
	clo_gerror_set(err, CCL_HIP_ERROR, st, "%s: %s", what, clo_hip_error_string(st));
	return 1;
}

static int use_device(CCLContext* ctx, GError** err) {
	if (ctx->dev.index < 0) {
		clo_gerror_set(err, CCL_HIP_ERROR, 100, "This context is offline (no device): nothing can be enqueued on it");
		return 0;
	}
	return !hip_failed(clo_hip_set_device(ctx->dev.index), err, "hipSetDevice");
}

/* ---------------- context / device ---------------- */

CCLContext* ccl_context_new_from_device_index(int device_index, GError** err) {
	int count = 0;
	int st = clo_hip_device_count(&count);
	if (st != 0 || count <= 0) {
		clo_gerror_set(err, CCL_HIP_ERROR, st ? st : 100,
			"No HIP device available (%s); cl_ops_amd has no CPU path", clo_hip_error_string(st ? st : 100));
		return NULL;
	}
	if (device_index < 0) device_index = 0;
	if (device_index >= count) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "Device index %d out of range (%d devices)", device_index, count);
		return NULL;
	}
	CCLContext* ctx = (CCLContext*) calloc(1, sizeof(*ctx));
	if (!ctx) return NULL;
	ctx->refcount = 1;
	ctx->dev.index = device_index;
	if (hip_failed(clo_hip_get_device_props(device_index, &ctx->dev.props), err, "hipGetDeviceProperties")) {
		free(ctx);
		return NULL;
	}
	if (!use_device(ctx, err)) { free(ctx); return NULL; }
	return ctx;
}

CCLContext* ccl_context_new_offline(GError** err) {
	(void) err;
	CCLContext* ctx = (CCLContext*) calloc(1, sizeof(*ctx));
	if (!ctx) return NULL;
	ctx->refcount = 1;
	ctx->dev.index = -1;
	ctx->dev.props.max_threads_per_block = 1024;
	ctx->dev.props.wavefront_size = 64;
	strcpy(ctx->dev.props.name, "offline (no device)");
	return ctx;
}

CCLContext* ccl_context_new_gpu(GError** err) {
	return ccl_context_new_from_device_index(0, err);
}

CCLContext* ccl_context_new_from_menu_full(void* dev_idx_ptr, GError** err) {
	int idx = dev_idx_ptr ? *(int*) dev_idx_ptr : -1;
	return ccl_context_new_from_device_index(idx, err);
}

void ccl_context_ref(CCLContext* ctx) { if (ctx) ++ctx->refcount; }

void ccl_context_unref(CCLContext* ctx) {
	if (ctx && --ctx->refcount == 0) free(ctx);
}

void ccl_context_destroy(CCLContext* ctx) { ccl_context_unref(ctx); }

CCLDevice* ccl_context_get_device(CCLContext* ctx, cl_uint index, GError** err) {
	if (!ctx || index != 0) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "Context holds exactly one device (index 0)");
		return NULL;
	}
	return &ctx->dev;
}

int ccl_device_get_index(CCLDevice* dev) { return dev ? dev->index : -1; }
size_t ccl_device_get_max_work_group_size(CCLDevice* dev) { return dev ? (size_t) dev->props.max_threads_per_block : 0; }
const char* ccl_device_get_name(CCLDevice* dev) { return dev ? dev->props.name : NULL; }

/* ---------------- queue ---------------- */

static CCLQueue* queue_alloc(CCLContext* ctx, void* stream, int owns, cl_ulong properties) {
	CCLQueue* cq = (CCLQueue*) calloc(1, sizeof(*cq));
	if (!cq) return NULL;
	ccl_context_ref(ctx);
	cq->ctx = ctx;
	cq->stream = stream;
	cq->owns_stream = owns;
	cq->profiling = (properties & CL_QUEUE_PROFILING_ENABLE) != 0;
	cq->refs = 1;
	return cq;
}

CCLQueue* ccl_queue_new(CCLContext* ctx, CCLDevice* dev, cl_ulong properties, GError** err) {
	(void) dev;
	if (!ctx) { clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "NULL context"); return NULL; }
	if (!use_device(ctx, err)) return NULL;
	void* stream = NULL;
	if (hip_failed(clo_hip_stream_create(&stream), err, "hipStreamCreate")) return NULL;
	return queue_alloc(ctx, stream, 1, properties);
}

CCLQueue* ccl_queue_new_from_stream(CCLContext* ctx, void* hip_stream, cl_ulong properties, GError** err) {
	if (!ctx) { clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "NULL context"); return NULL; }
	if (!use_device(ctx, err)) return NULL;
	return queue_alloc(ctx, hip_stream, 0, properties);
}

void ccl_queue_gc(CCLQueue* cq) {
	if (!cq) return;
	struct ccl_event* e = cq->events;
	while (e) {
		struct ccl_event* next = e->next;
		if (!e->start_borrowed) clo_hip_event_destroy(e->start);
		clo_hip_event_destroy(e->end);
		free(e);
		e = next;
	}
	for (e = cq->spare; e; ) {
		struct ccl_event* next = e->next;
		clo_hip_event_destroy(e->end);
		free(e);
		e = next;
	}
	cq->spare = NULL;
	cq->events = NULL;
	cq->first = NULL;
	cq->nevents = 0;
}

/* ---- status cells (clo_ccl.h, internal part) ---- */

clo_status_cell* clo_status_cell_new(void* dev_word) {
	clo_status_cell* c = (clo_status_cell*) calloc(1, sizeof(*c));
	if (c) { c->refs = 1; c->dev_word = dev_word; }
	return c;
}

void clo_status_cell_set_word(clo_status_cell* c, void* dev_word) { if (c) c->dev_word = dev_word; }
int clo_status_cell_take_tripped(clo_status_cell* c) { if (!c || !c->tripped) return 0; c->tripped = 0; return 1; }

void clo_status_cell_unref(clo_status_cell* c) {
	if (c && --c->refs == 0) free(c);
}

static void queue_drop_cell(CCLQueue* cq, int i) {
	clo_status_cell_unref(cq->cells[i]);
	memmove(&cq->cells[i], &cq->cells[i + 1], (size_t) (cq->ncells - 1 - i) * sizeof(cq->cells[0]));
	memmove(&cq->armed[i], &cq->armed[i + 1], (size_t) (cq->ncells - 1 - i) * sizeof(cq->armed[0]));
	--cq->ncells;
}

/* Called by a sorter / scanner for every command that polls other work-groups:
 * the queue checks the word at its next synchronisation point (and only then:
 * the check is a small blocking read, so a queue whose watched commands have all
 * been checked pays nothing more). */
void ccl_queue_watch_status(CCLQueue* cq, clo_status_cell* cell) {
	if (!cq || !cell) return;
	for (int i = 0; i < cq->ncells; ++i) if (cq->cells[i] == cell) { cq->armed[i] = 1; return; }
	for (int i = 0; i < cq->ncells; ++i)   /* owners that are gone */
		if (!cq->cells[i]->dev_word) { queue_drop_cell(cq, i); --i; }
	if (cq->ncells == CCL_QUEUE_MAX_CELLS) queue_drop_cell(cq, 0);   /* drop the oldest watch */
	++cell->refs;
	cq->armed[cq->ncells] = 1;
	cq->cells[cq->ncells++] = cell;
}

/* `cq` is made to wait for a command of another queue: whatever that queue
 * watches (and has not checked yet) can invalidate what `cq` produces from now
 * on, and the caller may well synchronise with `cq` alone — upstream's own
 * harness reads the sorted array back on a separate transfer queue
 * (benchmarks/clo_sort_bench.c:160-162,196). The watch travels with the wait. */
static void queue_inherit_watches(CCLQueue* cq, CCLQueue* from) {
	if (!cq || !from || cq == from) return;
	for (int i = 0; i < from->ncells; ++i)
		if (from->armed[i] && from->cells[i]->dev_word) ccl_queue_watch_status(cq, from->cells[i]);
}

/* After the queue's stream (final) or one of its events has been synchronised:
 * did any watched kernel give up a bounded spin? Returns 0 (and sets err) if so.
 * Only a FINAL check disarms a watch: behind a single event later commands of
 * the queue may still be running. */
static int queue_check_status(CCLQueue* cq, int final, GError** err) {
	int ok = 1;
	for (int i = 0; i < cq->ncells; ++i) {
		clo_status_cell* c = cq->cells[i];
		if (!c->dev_word) { queue_drop_cell(cq, i); --i; continue; }   /* the owner is gone */
		if (!cq->armed[i]) continue;
		if (final) cq->armed[i] = 0;
		const int st = clo_hip_check_status(c->dev_word, cq->stream);
		if (st == CLO_HIP_ETIMEOUT) {
			c->tripped = 1;
			clo_hip_memset_async(c->dev_word, 0, sizeof(unsigned), cq->stream);
			if (ok) clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY,
				"A kernel gave up waiting for another work-group's counts (bounded look-back spin): "
				"the data this queue produced since its last synchronisation is not valid");
			ok = 0;
		} else if (st != 0 && ok) {
			hip_failed(st, err, "clo_hip_check_status");
			ok = 0;
		}
	}
	return ok;
}

/* Synchronises and closes the queue NOW (the stream of a queue made by
 * ccl_queue_new is destroyed, a wrapped one is the caller's again); the struct
 * itself lives until the last holder has let go of it. */
void ccl_queue_destroy(CCLQueue* cq) {
	if (!cq) return;
	if (!cq->closed) {
		clo_hip_set_device(cq->ctx->dev.index);
		clo_hip_stream_synchronize(cq->stream);
		for (int i = 0; i < cq->ncells; ++i) clo_status_cell_unref(cq->cells[i]);
		cq->ncells = 0;
		ccl_queue_gc(cq);
		if (cq->owns_stream) clo_hip_stream_destroy(cq->stream);
		cq->stream = NULL;
		ccl_context_unref(cq->ctx);
		cq->ctx = NULL;
		cq->closed = 1;
	}
	clo_queue_drop(cq);
}

void clo_queue_hold(CCLQueue* cq) { if (cq) ++cq->refs; }
void clo_queue_drop(CCLQueue* cq) { if (cq && --cq->refs == 0) free(cq); }
int clo_queue_is_closed(CCLQueue* cq) { return !cq || cq->closed; }

CCLDevice* ccl_queue_get_device(CCLQueue* cq, GError** err) {
	if (!cq) { clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "NULL queue"); return NULL; }
	return &cq->ctx->dev;
}

CCLContext* ccl_queue_get_context(CCLQueue* cq, GError** err) {
	if (!cq) { clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "NULL queue"); return NULL; }
	return cq->ctx;
}

cl_bool ccl_queue_finish(CCLQueue* cq, GError** err) {
	if (!cq) return CL_FALSE;
	if (hip_failed(clo_hip_stream_synchronize(cq->stream), err, "hipStreamSynchronize")) return CL_FALSE;
	return queue_check_status(cq, 1, err) ? CL_TRUE : CL_FALSE;
}

void* ccl_queue_get_stream(CCLQueue* cq) { return cq ? cq->stream : NULL; }
int ccl_queue_is_profiling(CCLQueue* cq) { return cq ? cq->profiling : 0; }

static void event_free(struct ccl_event* e) {
	if (!e->start_borrowed) clo_hip_event_destroy(e->start);
	clo_hip_event_destroy(e->end);
	free(e);
}

CCLEvent* ccl_queue_begin_command(CCLQueue* cq, const char* name, GError** err) {
	return ccl_queue_begin_command_after(cq, name, NULL, err);
}

/* `after` (may be NULL): the command enqueued right before this one on the same
 * queue, of the same group of launches. On a profiling queue the new command's
 * time then runs from `after`'s end — one marker between two kernels instead of
 * two, and the durations of the group add up to its span exactly. */
CCLEvent* ccl_queue_begin_command_after(CCLQueue* cq, const char* name, CCLEvent* after, GError** err) {
	if (!cq) { clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "NULL queue"); return NULL; }
	if (!use_device(cq->ctx, err)) return NULL;
	struct ccl_event* e = NULL;
	if (!cq->profiling && cq->spare) {   /* a retired event: struct and hip event are reused */
		e = cq->spare;
		cq->spare = e->next;
		void* end = e->end;
		memset(e, 0, sizeof(*e));
		e->end = end;
	} else {
		e = (struct ccl_event*) calloc(1, sizeof(*e));
		if (!e) return NULL;
		/* a profiling queue stamps every command; any other queue creates and records the event of a
		 * command only if somebody asks for it (event_ensure_recorded) */
		if (cq->profiling && hip_failed(clo_hip_event_create(&e->end), err, "hipEventCreate")) { free(e); return NULL; }
	}
	ccl_event_set_name(e, name);
	if (cq->profiling && after != NULL && after->cq == cq && after->end != NULL) {
		e->start = after->end;
		e->start_borrowed = 1;
	} else if (cq->profiling) {
		if (hip_failed(clo_hip_event_create(&e->start), err, "hipEventCreate")
			|| hip_failed(clo_hip_event_record(e->start, cq->stream), err, "hipEventRecord")) {
			clo_hip_event_destroy(e->start);
			clo_hip_event_destroy(e->end);
			free(e);
			return NULL;
		}
	}
	e->cq = cq;
	e->next = cq->events;
	cq->events = e;
	if (!cq->first) cq->first = e;
	if (++cq->nevents > 2 * CCL_QUEUE_KEEP_EVENTS && !cq->profiling) {
		/* Nobody profiles this queue: only the most recent events stay distinct (a
		 * loop of sorts would otherwise grow the list without bound); the older
		 * ones are retired to the spare list (see CCL_QUEUE_KEEP_EVENTS). */
		struct ccl_event* keep = cq->events;
		for (size_t i = 1; i < CCL_QUEUE_KEEP_EVENTS && keep->next; ++i) keep = keep->next;
		struct ccl_event* old = keep->next;
		keep->next = NULL;
		cq->first = keep;
		cq->nevents = CCL_QUEUE_KEEP_EVENTS;
		while (old) { struct ccl_event* next = old->next; old->next = cq->spare; cq->spare = old; old = next; }
	}
	return e;
}

void ccl_queue_abort_command(CCLQueue* cq, CCLEvent* evt) {
	if (!cq || !evt) return;
	struct ccl_event** link = &cq->events;
	while (*link && *link != evt) link = &(*link)->next;
	if (!*link) return;
	*link = evt->next;
	if (cq->first == evt) cq->first = NULL;
	for (struct ccl_event* e = cq->events; e; e = e->next) if (!e->next) cq->first = e;
	if (cq->nevents) --cq->nevents;
	event_free(evt);
}

/* On a queue WITHOUT profiling the end of a command is not recorded when the command is enqueued:
 * a hipEventRecord is a marker packet between two kernels (a few microseconds of pipeline bubble per
 * command: 6 % of a 2^26-element scan), and most events are never looked at. Whoever does need the
 * event — a wait list, a blocking copy, ccl_event_wait — has it recorded THEN, on the queue's stream:
 * the queue is in order, so that marker completes after the command it stands for (and after whatever
 * was enqueued behind it meanwhile: a wait that is at worst a little longer, never too short). A
 * profiling queue records every command where it ends, as cf4ocl2's events do. */
static int event_ensure_recorded(struct ccl_event* e, GError** err) {
	if (e->recorded) return 1;
	if (!e->cq || e->cq->closed) return 1;   /* the queue was synchronised when it went: nothing left to wait for */
	if (!e->end && hip_failed(clo_hip_event_create(&e->end), err, "hipEventCreate")) return 0;
	if (hip_failed(clo_hip_event_record(e->end, e->cq->stream), err, "hipEventRecord")) return 0;
	e->recorded = 1;
	return 1;
}

cl_bool ccl_queue_end_command(CCLQueue* cq, CCLEvent* evt, GError** err) {
	if (!cq || !evt) return CL_FALSE;
	if (!cq->profiling) { evt->recorded = 0; return CL_TRUE; }   /* lazily (above) */
	if (hip_failed(clo_hip_event_record(evt->end, cq->stream), err, "hipEventRecord")) return CL_FALSE;
	evt->recorded = 1;
	return CL_TRUE;
}

cl_bool ccl_queue_wait_for(CCLQueue* cq, CCLEventWaitList* ewl, GError** err) {
	if (!cq) return CL_FALSE;
	if (!ewl || !*ewl) return CL_TRUE;
	for (size_t i = 0; i < (*ewl)->n; ++i) {
		if ((*ewl)->evts[i]->cq == cq && !cq->profiling) continue;   /* the same in-order queue: already behind it */
		if (!event_ensure_recorded((*ewl)->evts[i], err)) return CL_FALSE;
		if ((*ewl)->evts[i]->end == NULL) continue;                    /* (its queue is gone, and was synchronised) */
		if (hip_failed(clo_hip_stream_wait_event(cq->stream, (*ewl)->evts[i]->end), err, "hipStreamWaitEvent"))
			return CL_FALSE;
		queue_inherit_watches(cq, (*ewl)->evts[i]->cq);
	}
	return CL_TRUE;
}

/* ---------------- buffer ---------------- */

CCLBuffer* ccl_buffer_new(CCLContext* ctx, cl_ulong flags, size_t size, void* host_ptr, GError** err) {
	(void) flags;
	if (!ctx) { clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "NULL context"); return NULL; }
	if (host_ptr) { clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "host_ptr buffers are not supported"); return NULL; }
	if (!use_device(ctx, err)) return NULL;
	CCLBuffer* b = (CCLBuffer*) calloc(1, sizeof(*b));
	if (!b) return NULL;
	if (hip_failed(clo_hip_malloc(&b->dptr, size), err, "hipMalloc")) { free(b); return NULL; }
	ccl_context_ref(ctx);
	b->ctx = ctx;
	b->size = size;
	b->owns = 1;
	return b;
}

CCLBuffer* ccl_buffer_new_from_device_ptr(CCLContext* ctx, void* device_ptr, size_t size, GError** err) {
	if (!ctx || !device_ptr) { clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "NULL context or pointer"); return NULL; }
	if (!use_device(ctx, err)) return NULL;
	CCLBuffer* b = (CCLBuffer*) calloc(1, sizeof(*b));
	if (!b) return NULL;
	ccl_context_ref(ctx);
	b->ctx = ctx;
	b->dptr = device_ptr;
	b->size = size;
	b->owns = 0;
	return b;
}

void ccl_buffer_destroy(CCLBuffer* buf) {
	if (!buf) return;
	if (buf->owns) {
		clo_hip_set_device(buf->ctx->dev.index);
		clo_hip_free(buf->dptr);  /* hipFree waits for work using the memory */
	}
	ccl_context_unref(buf->ctx);
	free(buf);
}

size_t ccl_buffer_get_size(CCLBuffer* buf) { return buf ? buf->size : 0; }
void* ccl_buffer_get_device_ptr(CCLBuffer* buf) { return buf ? buf->dptr : NULL; }

typedef int (*copy_fn)(void*, const void*, size_t, void*);

static CCLEvent* enqueue_copy(CCLQueue* cq, const char* name, copy_fn fn, void* dst, const void* src,
	size_t size, cl_bool blocking, CCLEventWaitList* ewl, GError** err) {
	if (fn != clo_hip_memcpy_d2d_async && ewl && *ewl) {
		/* A copy from/to (pageable) host memory holds the calling thread until it
		 * is done, so nothing is lost by waiting for its dependencies here — and a
		 * wait pending in the copy's stream pushes the runtime onto a slower copy
		 * path (measured: reading back 256 MiB took 12.9 ms instead of 4.6 ms). */
		for (size_t i = 0; i < (*ewl)->n; ++i) {
			if (!event_ensure_recorded((*ewl)->evts[i], err)) return NULL;
			if ((*ewl)->evts[i]->end && hip_failed(clo_hip_event_synchronize((*ewl)->evts[i]->end), err, "hipEventSynchronize")) return NULL;
			queue_inherit_watches(cq, (*ewl)->evts[i]->cq);
		}
	} else if (!ccl_queue_wait_for(cq, ewl, err)) return NULL;
	if (ewl) ccl_event_wait_list_clear(ewl);
	CCLEvent* e = ccl_queue_begin_command(cq, name, err);
	if (!e) return NULL;
	if (hip_failed(fn(dst, src, size, cq->stream), err, name)) { ccl_queue_abort_command(cq, e); return NULL; }
	if (!ccl_queue_end_command(cq, e, err)) { ccl_queue_abort_command(cq, e); return NULL; }
	if (blocking && (!event_ensure_recorded(e, err) || (e->end && hip_failed(clo_hip_event_synchronize(e->end), err, "hipEventSynchronize")))) return NULL;
	return e;
}

static int range_ok(CCLBuffer* b, size_t off, size_t size, GError** err) {
	if (!b || off > b->size || size > b->size - off) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "Buffer range out of bounds");
		return 0;
	}
	return 1;
}

CCLEvent* ccl_buffer_enqueue_write(CCLBuffer* buf, CCLQueue* cq, cl_bool blocking, size_t offset,
	size_t size, void* ptr, CCLEventWaitList* ewl, GError** err) {
	if (!cq || !range_ok(buf, offset, size, err)) return NULL;
	return enqueue_copy(cq, "write_buffer", clo_hip_memcpy_h2d_async, (char*) buf->dptr + offset, ptr, size, blocking, ewl, err);
}

CCLEvent* ccl_buffer_enqueue_read(CCLBuffer* buf, CCLQueue* cq, cl_bool blocking, size_t offset,
	size_t size, void* ptr, CCLEventWaitList* ewl, GError** err) {
	if (!cq || !range_ok(buf, offset, size, err)) return NULL;
	return enqueue_copy(cq, "read_buffer", clo_hip_memcpy_d2h_async, ptr, (char*) buf->dptr + offset, size, blocking, ewl, err);
}

CCLEvent* ccl_buffer_enqueue_copy(CCLBuffer* src, CCLBuffer* dst, CCLQueue* cq, size_t src_offset,
	size_t dst_offset, size_t size, CCLEventWaitList* ewl, GError** err) {
	if (!cq || !range_ok(src, src_offset, size, err) || !range_ok(dst, dst_offset, size, err)) return NULL;
	return enqueue_copy(cq, "copy_buffer", clo_hip_memcpy_d2d_async, (char*) dst->dptr + dst_offset,
		(char*) src->dptr + src_offset, size, CL_FALSE, ewl, err);
}

/* ---------------- events ---------------- */

void ccl_event_set_name(CCLEvent* evt, const char* name) {
	if (!evt) return;
	strncpy(evt->name, name ? name : "", sizeof(evt->name) - 1);
	evt->name[sizeof(evt->name) - 1] = '\0';
}

const char* ccl_event_get_name(CCLEvent* evt) { return evt ? evt->name : NULL; }

static void ewl_push(CCLEventWaitList* ewl, CCLEvent* e) {
	if (!ewl || !e) return;
	if (!*ewl) *ewl = (CCLEventWaitList) calloc(1, sizeof(**ewl));
	if (!*ewl) return;
	if ((*ewl)->n == (*ewl)->cap) {
		size_t cap = (*ewl)->cap ? (*ewl)->cap * 2 : 4;
		CCLEvent** p = (CCLEvent**) realloc((*ewl)->evts, cap * sizeof(CCLEvent*));
		if (!p) return;
		(*ewl)->evts = p;
		(*ewl)->cap = cap;
	}
	(*ewl)->evts[(*ewl)->n++] = e;
}

CCLEventWaitList* ccl_ewl(CCLEventWaitList* ewl, ...) {
	va_list ap;
	va_start(ap, ewl);
	for (CCLEvent* e = va_arg(ap, CCLEvent*); e != NULL; e = va_arg(ap, CCLEvent*)) ewl_push(ewl, e);
	va_end(ap);
	return ewl;
}

void ccl_event_wait_list_add(CCLEventWaitList* ewl, ...) {
	va_list ap;
	va_start(ap, ewl);
	for (CCLEvent* e = va_arg(ap, CCLEvent*); e != NULL; e = va_arg(ap, CCLEvent*)) ewl_push(ewl, e);
	va_end(ap);
}

void ccl_event_wait_list_clear(CCLEventWaitList* ewl) {
	if (ewl && *ewl) {
		free((*ewl)->evts);
		free(*ewl);
		*ewl = NULL;
	}
}

cl_bool ccl_event_wait(CCLEventWaitList* ewl, GError** err) {
	cl_bool ok = CL_TRUE;
	if (ewl && *ewl) {
		for (size_t i = 0; i < (*ewl)->n && ok; ++i) {
			if (!event_ensure_recorded((*ewl)->evts[i], err)) ok = CL_FALSE;
			else if ((*ewl)->evts[i]->end && hip_failed(clo_hip_event_synchronize((*ewl)->evts[i]->end), err, "hipEventSynchronize")) ok = CL_FALSE;
		}
		/* a command that polls other work-groups may have given up: its queue
		 * watches the status word (only queues that ran such commands pay this) */
		for (size_t i = 0; i < (*ewl)->n && ok; ++i) {
			CCLQueue* cq = (*ewl)->evts[i]->cq;
			if (cq && cq->ncells > 0 && !queue_check_status(cq, 0, err)) ok = CL_FALSE;
		}
		ccl_event_wait_list_clear(ewl);
	}
	return ok;
}

/* ---------------- program token ---------------- */

CCLProgram* ccl_program_new_token(CCLContext* ctx, const char* what, const char* build_options) {
	(void) ctx;
	CCLProgram* p = (CCLProgram*) calloc(1, sizeof(*p));
	if (!p) return NULL;
	p->what = strdup(what ? what : "");
	p->options = strdup(build_options ? build_options : "");
	return p;
}

void ccl_program_destroy(CCLProgram* prg) {
	if (!prg) return;
	free(prg->what);
	free(prg->options);
	free(prg);
}

const char* ccl_program_get_build_options(CCLProgram* prg) { return prg ? prg->options : NULL; }

/* ---------------- profiling ---------------- */

CCLProf* ccl_prof_new(void) { return (CCLProf*) calloc(1, sizeof(CCLProf)); }
void ccl_prof_destroy(CCLProf* prof) {
	if (!prof) return;
	free(prof->aggs);
	free(prof->agg_names);
	free(prof);
}

static void prof_agg_add(CCLProf* prof, const char* name, cl_ulong ns) {
	for (size_t i = 0; i < prof->naggs; ++i)
		if (strcmp(prof->agg_names[i], name) == 0) { prof->aggs[i].absolute_time += ns; return; }
	if (prof->naggs == prof->cap) {
		const size_t cap = prof->cap ? prof->cap * 2 : 8;
		CCLProfAgg* a = (CCLProfAgg*) realloc(prof->aggs, cap * sizeof(*a));
		if (!a) return;
		prof->aggs = a;
		char (*n)[48] = (char (*)[48]) realloc(prof->agg_names, cap * sizeof(*n));
		if (!n) return;
		prof->agg_names = n;
		prof->cap = cap;
	}
	snprintf(prof->agg_names[prof->naggs], sizeof(prof->agg_names[0]), "%s", name);
	prof->aggs[prof->naggs].absolute_time = ns;
	prof->aggs[prof->naggs].relative_time = 0.0;
	++prof->naggs;
}

const CCLProfAgg* ccl_prof_get_agg(CCLProf* prof, const char* event_name) {
	if (!prof || !event_name) return NULL;
	for (size_t i = 0; i < prof->naggs; ++i)
		if (strcmp(prof->agg_names[i], event_name) == 0) return &prof->aggs[i];
	return NULL;
}

void ccl_prof_iter_agg_init(CCLProf* prof, int sort) { (void) sort; if (prof) prof->iter = 0; }
const CCLProfAgg* ccl_prof_iter_agg_next(CCLProf* prof) {
	if (!prof || prof->iter >= prof->naggs) return NULL;
	return &prof->aggs[prof->iter++];
}

void ccl_prof_add_queue(CCLProf* prof, const char* name, CCLQueue* cq) {
	(void) name;
	if (prof && cq && prof->nqueues < CCL_PROF_MAX_QUEUES) prof->queues[prof->nqueues++] = cq;
}

cl_bool ccl_prof_calc(CCLProf* prof, GError** err) {
	if (!prof) return CL_FALSE;
	prof->duration_ns = 0;
	prof->naggs = 0;
	for (int q = 0; q < prof->nqueues; ++q) {
		CCLQueue* cq = prof->queues[q];
		if (!cq->profiling) {
			clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "Queue was not created with CL_QUEUE_PROFILING_ENABLE");
			return CL_FALSE;
		}
		if (!cq->events) continue;
		if (hip_failed(clo_hip_stream_synchronize(cq->stream), err, "hipStreamSynchronize")) return CL_FALSE;
		if (!queue_check_status(cq, 1, err)) return CL_FALSE;
		/* Sum of the commands' own durations: on an in-order queue that is the
		 * time the device spent on them, idle gaps between commands (a chunked
		 * pipeline waiting for its next copy) excluded — cf4ocl2's aggregate
		 * event time. */
		for (struct ccl_event* e = cq->events; e != NULL; e = e->next) {
			float ms = 0.f;
			if (!e->start) continue;
			if (hip_failed(clo_hip_event_elapsed_ms(e->start, e->end, &ms), err, "hipEventElapsedTime")) return CL_FALSE;
			prof->duration_ns += (cl_ulong) ((double) ms * 1e6);
			prof_agg_add(prof, e->name, (cl_ulong) ((double) ms * 1e6));
		}
		/* cf4ocl2 releases the queue's events once profiled. */
		ccl_queue_gc(cq);
	}
	for (size_t i = 0; i < prof->naggs; ++i) {   /* (the names array may have moved while growing) */
		prof->aggs[i].event_name = prof->agg_names[i];
		prof->aggs[i].relative_time = prof->duration_ns ? (double) prof->aggs[i].absolute_time / (double) prof->duration_ns : 0.0;
	}
	return CL_TRUE;
}

cl_ulong ccl_prof_get_duration(CCLProf* prof) { return prof ? prof->duration_ns : 0; }
