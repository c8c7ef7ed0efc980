/*
 * clo_common.c — CloType table, bit utilities, error quark and the GLib-shaped
 * error helpers. Behaviour follows src/cl_ops/common/clo_common.c:54-221 of the
 * reference (restated, not copied).
 */
#include "clo_common.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- GError helpers (clo_glib_compat.h) ---- */
#ifndef CLO_USE_GLIB

void clo_gerror_free(GError* err) {
	if (!err) return;
	free(err->message);
	free(err);
}

void clo_gerror_clear(GError** err) {
	if (err && *err) {
		clo_gerror_free(*err);
		*err = NULL;
	}
}

void clo_gerror_set(GError** err, GQuark domain, gint code, const char* fmt, ...) {
	if (!err) return;
	if (*err) return; /* GLib warns and keeps the first error */
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	GError* e = (GError*) malloc(sizeof(GError));
	if (!e) return;
	e->domain = domain;
	e->code = code;
	e->message = strdup(buf);
	*err = e;
}

void clo_gerror_propagate(GError** dest, GError* src) {
	if (!src) return;
	if (dest && !*dest) *dest = src;
	else clo_gerror_free(src);
}

/* A fixed quark table: ids are stable within the process, 0 means "none". */
static const char* const clo_quarks[] = { NULL, "clo-error-quark", "ccl-hip-error-quark" };

GQuark clo_quark_from_string(const char* s) {
	for (GQuark i = 1; i < sizeof(clo_quarks) / sizeof(clo_quarks[0]); ++i)
		if (s && strcmp(s, clo_quarks[i]) == 0) return i;
	return 0;
}

const char* clo_quark_to_string(GQuark q) {
	return q < sizeof(clo_quarks) / sizeof(clo_quarks[0]) ? clo_quarks[q] : NULL;
}

#define CLO_QUARK(s) clo_quark_from_string(s)
#else
#define CLO_QUARK(s) g_quark_from_static_string(s)
#define clo_gerror_set g_set_error
#endif

/* ---- types (ref: clo_common.c:54-124) ---- */

static const struct { const char* name; size_t size; } clo_types[] = {
	{"char", 1}, {"uchar", 1}, {"short", 2}, {"ushort", 2}, {"int", 4}, {"uint", 4},
	{"long", 8}, {"ulong", 8}, {"half", 2}, {"float", 4}, {"double", 8}
};
#define CLO_NUM_TYPES ((int) (sizeof(clo_types) / sizeof(clo_types[0])))

const char* clo_type_get_name(CloType type) {
	if ((int) type < 0 || (int) type >= CLO_NUM_TYPES) return NULL;
	return clo_types[type].name;
}

size_t clo_type_sizeof(CloType type) {
	if ((int) type < 0 || (int) type >= CLO_NUM_TYPES) return 0;
	return clo_types[type].size;
}

CloType clo_type_by_name(const char* name, GError** err) {
	for (int i = 0; i < CLO_NUM_TYPES; ++i)
		if (name && strcmp(name, clo_types[i].name) == 0) return (CloType) i;
	clo_gerror_set(err, CLO_ERROR, CLO_ERROR_UNKNOWN_TYPE, "Unknown type '%s'", name ? name : "(null)");
	return (CloType) -1;
}

int clo_type_is_signed(CloType type) {
	return type == CLO_CHAR || type == CLO_SHORT || type == CLO_INT || type == CLO_LONG;
}

int clo_type_is_float(CloType type) {
	return type == CLO_HALF || type == CLO_FLOAT || type == CLO_DOUBLE;
}

/* ---- bit utilities (ref: clo_common.c:141-199) ---- */

unsigned int clo_nlpo2(unsigned int x) {
	if ((x & (x - 1)) == 0) return x;
	unsigned int p = 1;
	while (p < x && p != 0) p <<= 1;
	return p; /* wraps to 0 above 2^31 like upstream's (x|x>>1|...)+1 */
}

unsigned int clo_ones32(unsigned int x) {
	return (unsigned int) __builtin_popcount(x);
}

unsigned int clo_tzc(int x) {
	return clo_ones32((unsigned int) ((x & -x) - 1));
}

unsigned int clo_sum(unsigned int x) {
	/* 0 + 1 + ... + x, modulo 2^32 like the recursive upstream form. */
	return (unsigned int) (((unsigned long long) x * ((unsigned long long) x + 1ull)) / 2ull);
}

void clo_print_to_null(const gchar* string) {
	(void) string;
}

GQuark clo_error_quark(void) {
	return CLO_QUARK("clo-error-quark");
}
