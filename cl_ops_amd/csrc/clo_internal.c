/* clo_internal.c — see clo_internal.h. */
#include "clo_internal.h"

#include <string.h>

int clo_debug_enabled(void) {
	static int cached = -1;
	int c = __atomic_load_n(&cached, __ATOMIC_RELAXED);
	if (c < 0) {
		const char* e = getenv("CLO_DEBUG");
		c = (e && *e && strcmp(e, "0") != 0) ? 1 : 0;
		__atomic_store_n(&cached, c, __ATOMIC_RELAXED);
	}
	return c;
}

/* The host drivers' environment switches, each read in one place and only when an object is made
 * (INTEGRATION.md has the table): */
int clo_env_no_warmup(void) { return getenv("CLO_NO_WARMUP") != NULL; }
int clo_env_flag(const char* name) {   /* -1 unset, else 0 / 1 */
	const char* x = getenv(name);
	return x ? (atoi(x) != 0) : -1;
}

int clo_devbuf_reserve(clo_devbuf* b, size_t bytes) {
	if (b->ptr && b->bytes >= bytes) return 0;
	if (b->ptr) {
		int st = clo_hip_free(b->ptr);
		b->ptr = NULL;
		b->bytes = 0;
		if (st) return st;
	}
	int st = clo_hip_malloc(&b->ptr, bytes);
	if (st) { b->ptr = NULL; return st; }
	b->bytes = bytes;
	return 0;
}

void clo_devbuf_release(clo_devbuf* b) {
	if (b->ptr) clo_hip_free(b->ptr);
	b->ptr = NULL;
	b->bytes = 0;
}

int clo_hip_failed(int st, GError** err, const char* what) {
	if (st == 0) return 0;
	clo_gerror_set(err, CCL_HIP_ERROR, st, "%s: %s", what, clo_hip_error_string(st));
	return 1;
}

int clo_parse_options(const char* options, clo_option_cb cb, void* user, const char* algo, GError** err) {
	if (!options) return 1;
	char* copy = strdup(options);
	if (!copy) return 0;
	int ok = 1;
	char* save = NULL;
	for (char* tok = strtok_r(copy, ",", &save); tok && ok; tok = strtok_r(NULL, ",", &save)) {
		if (tok[0] == '\0') continue;
		char* eq = strchr(tok, '=');
		/* g_strsplit_set(tok, "=", 2) yields exactly two tokens iff there is
		 * at least one '=' */
		if (!eq) {
			clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "Invalid option '%s' for %s sort.", tok, algo);
			ok = 0;
			break;
		}
		char* token_copy = strdup(tok);
		*eq = '\0';
		ok = cb(tok, eq + 1, token_copy ? token_copy : "", user, err);
		free(token_copy);
	}
	free(copy);
	return ok;
}

/* ---- stream guard (clo_internal.h) ---- */

int clo_stream_guard_enter(clo_stream_guard* g, CCLQueue* cq) {
	if (g->cq == cq) return 0;
	if (g->cq) {
		if (!clo_queue_is_closed(g->cq) && ccl_queue_get_stream(g->cq) != ccl_queue_get_stream(cq)) {
			if (!g->evt) {
				const int st = clo_hip_event_create(&g->evt);
				if (st != 0) return st;
			}
			int st = clo_hip_event_record(g->evt, ccl_queue_get_stream(g->cq));
			if (st == 0) st = clo_hip_stream_wait_event(ccl_queue_get_stream(cq), g->evt);
			if (st != 0) return st;
		}
		clo_queue_drop(g->cq);
	}
	clo_queue_hold(cq);
	g->cq = cq;
	return 0;
}

void clo_stream_guard_release(clo_stream_guard* g) {
	if (g->evt) clo_hip_event_destroy(g->evt);
	g->evt = NULL;
	clo_queue_drop(g->cq);
	g->cq = NULL;
}

/* ---- per-kernel events (clo_internal.h) ---- */

static void kernel_events_observer(void* user, const char* label, int phase, void* stream) {
	clo_kernel_events* ke = (clo_kernel_events*) user;
	(void) stream;
	if (ke->failed) return;
	if (phase == 0) {
		const char* name = ke->other;
		for (size_t i = 0; i < ke->nmap; ++i)
			if (strcmp(ke->map[i].label, label) == 0) { name = ke->map[i].name; break; }
		ke->open = ccl_queue_begin_command_after(ke->cq, name, ke->last, NULL);
		if (!ke->open) ke->failed = 1;
	} else if (ke->open) {
		if (!ccl_queue_end_command(ke->cq, ke->open, NULL)) {
			ccl_queue_abort_command(ke->cq, ke->open);
			ke->failed = 1;
		} else {
			ke->last = ke->open;
		}
		ke->open = NULL;
	}
}

void clo_kernel_events_install(clo_kernel_events* ke, CCLQueue* cq, const clo_kname* map, size_t nmap, const char* other) {
	memset(ke, 0, sizeof(*ke));
	ke->cq = cq;
	ke->map = map;
	ke->nmap = nmap;
	ke->other = other;
	clo_hip_set_launch_observer(kernel_events_observer, ke);
}

CCLEvent* clo_kernel_events_remove(clo_kernel_events* ke, GError** err) {
	clo_hip_set_launch_observer(NULL, NULL);
	if (ke->open) { ccl_queue_abort_command(ke->cq, ke->open); ke->open = NULL; }   /* a launch that never reported its end */
	if (ke->failed) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY, "Could not record the profiling events of a kernel launch");
		return NULL;
	}
	return ke->last;
}

/* ---- cached launch sequences (clo_internal.h) ---- */

void clo_graph_cache_release(clo_graph_cache* gc) {
	if (!gc) return;
	clo_hip_graph_destroy(gc->exec);
	gc->exec = NULL;
	gc->seen = 0;
}

int clo_graph_cache_run(clo_graph_cache* gc, int allowed, const void* k0, const void* k1, size_t n, int variant,
	void* stream, clo_enqueue_fn enqueue, void* user) {
	const int same = gc->k0 == k0 && gc->k1 == k1 && gc->n == n && gc->variant == variant && gc->stream == stream;
	allowed = allowed && !clo_hip_timing_enabled();
	if (!same) {
		clo_graph_cache_release(gc);
		gc->k0 = k0; gc->k1 = k1; gc->n = n; gc->variant = variant; gc->stream = stream;
	}
	if (allowed && same && gc->exec != NULL) return clo_hip_graph_launch(gc->exec, stream);
	/* Replay is an optimisation, never a requirement: the legacy NULL stream
	 * (a queue adopted from torch's default stream) cannot be captured, nor can
	 * a stream that is being captured already; and when capturing or
	 * instantiating fails the sequence is simply enqueued. */
	if (allowed && same && gc->seen >= 1 && stream != NULL && !clo_hip_stream_is_capturing(stream)) {
		if (clo_hip_graph_capture_begin(stream) == 0) {
			const int st = enqueue(user, stream);
			void* exec = NULL;
			const int st2 = clo_hip_graph_capture_end(stream, &exec);
			if (st == 0 && st2 == 0 && exec != NULL) {
				gc->exec = exec;
				return clo_hip_graph_launch(exec, stream);   /* capturing recorded the launches, it did not run them */
			}
			clo_hip_graph_destroy(exec);   /* (nothing was enqueued: capturing only records) */
		}
		gc->seen = -1;   /* never try to capture this key again */
		return enqueue(user, stream);
	}
	if (gc->seen >= 0) gc->seen += 1;
	return enqueue(user, stream);
}
