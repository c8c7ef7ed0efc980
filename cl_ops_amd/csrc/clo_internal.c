/* clo_internal.c — see clo_internal.h. */
#include "clo_internal.h"

#include <string.h>

int clo_debug_enabled(void) {
	static int cached = -1;
	if (cached < 0) {
		const char* e = getenv("CLO_DEBUG");
		cached = (e && *e && strcmp(e, "0") != 0) ? 1 : 0;
	}
	return cached;
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
