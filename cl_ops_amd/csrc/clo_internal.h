/*
 * clo_internal.h — helpers shared by the host drivers (not installed).
 */
#ifndef CLO_INTERNAL_H
#define CLO_INTERNAL_H

#include <stdio.h>
#include <stdlib.h>

#include "clo_common.h"
#include "clo_hip.h"

#ifdef CLO_USE_GLIB
#define clo_return_if_fail g_return_if_fail
#define clo_return_val_if_fail g_return_val_if_fail
#define clo_gerror_set g_set_error
#define clo_gerror_propagate g_propagate_error
#else
/* g_return_*_if_fail: a soft assertion — complain on stderr, return a default. */
#define clo_return_if_fail(expr) \
	do { if (!(expr)) { fprintf(stderr, "cl_ops-CRITICAL: %s: assertion '%s' failed\n", __func__, #expr); return; } } while (0)
#define clo_return_val_if_fail(expr, val) \
	do { if (!(expr)) { fprintf(stderr, "cl_ops-CRITICAL: %s: assertion '%s' failed\n", __func__, #expr); return (val); } } while (0)
#endif

/* CLO_DEBUG=1 in the environment turns on the g_debug-style trace upstream
 * emits under log domain "cl_ops" (e.g. clo_sort_satradix.c:171,192). */
int clo_debug_enabled(void);
#define clo_debug(...) do { if (clo_debug_enabled()) { fprintf(stderr, "cl_ops-DEBUG: " __VA_ARGS__); fputc('\n', stderr); } } while (0)

/* A grow-only device allocation cached inside a sorter/scanner object, so the
 * hot path never calls hipMalloc after the first use (upstream allocates and
 * frees its aux buffers on every call: clo_sort_satradix.c:242-257,327-330). */
typedef struct {
	void* ptr;
	size_t bytes;
} clo_devbuf;

/* Returns 0 or a clo_hip status. Contents are NOT preserved when it grows. */
int clo_devbuf_reserve(clo_devbuf* b, size_t bytes);
void clo_devbuf_release(clo_devbuf* b);

/* A launch sequence that depends only on its arguments (buffers, size,
 * stream), cached as an executable hipGraph: the FIRST call with a given key
 * launches normally, a second consecutive call with the same key captures the
 * sequence and replays it, later ones replay. Any other key drops the graph.
 * Never used while per-kernel timing is on (its event pairs need real
 * launches). `enqueue` puts the whole sequence on `stream` and returns a
 * clo_hip status. */
typedef struct {
	void* exec;
	const void* k0;
	const void* k1;
	void* stream;
	size_t n;
	int variant, seen;
} clo_graph_cache;
typedef int (*clo_enqueue_fn)(void* user, void* stream);
int clo_graph_cache_run(clo_graph_cache* gc, int allowed, const void* k0, const void* k1, size_t n, int variant,
	void* stream, clo_enqueue_fn enqueue, void* user);
void clo_graph_cache_release(clo_graph_cache* gc);

/* Set *err from a clo_hip_* status (domain CCL_HIP_ERROR); returns 1 if st != 0. */
int clo_hip_failed(int st, GError** err, const char* what);

/* Parse one "key=value" option list the way upstream does
 * (g_strsplit_set on "," then "="; clo_sort_abitonic.c:486-543,
 * clo_sort_satradix.c:366-421). Calls cb(key, value, token, user) per non-empty
 * token; a token without exactly one '=' makes it return 0 and set *bad to a
 * malloc'd copy of the token. cb returns 0 to abort. */
typedef int (*clo_option_cb)(const char* key, const char* value, const char* token, void* user, GError** err);
int clo_parse_options(const char* options, clo_option_cb cb, void* user, const char* algo, GError** err);

#endif
