/*
 * clo_glib_compat.h — the few GLib names the cl_ops public API is written in
 * (GError**, gboolean, guint, GQuark ...), for builds without GLib.
 *
 * Upstream cl_ops reports errors GLib-style (src/cl_ops/common/clo_common.c:221,
 * _g_err_macros.h:61-96): `GError** err` out-parameters holding {domain quark,
 * code, message}. This image has no GLib development files, so the library is
 * built against this header, which declares a layout-compatible GError and
 * clo_-prefixed helpers (no g_* symbol is exported, so linking the real GLib
 * next to this library never clashes). A maintainer wiring this library into
 * the real cl_ops tree compiles with -DCLO_USE_GLIB to get <glib.h> instead
 * (see INTEGRATION.md).
 */
#ifndef CLO_GLIB_COMPAT_H
#define CLO_GLIB_COMPAT_H

#ifdef CLO_USE_GLIB
#include <glib.h>
#define clo_gerror_free g_error_free
#define clo_gerror_clear g_clear_error
#else

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef char gchar;
typedef int gint;
typedef int gboolean;
typedef unsigned int guint;
typedef uint32_t guint32;
typedef uint32_t GQuark;
typedef void* gpointer;

/* Same field order and types as GLib's struct _GError. */
typedef struct _GError {
	GQuark domain;
	gint code;
	gchar* message;
} GError;

#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif

/* Free an error returned through a GError** (g_error_free). NULL is ignored. */
void clo_gerror_free(GError* err);
/* g_clear_error: free *err and set it to NULL. */
void clo_gerror_clear(GError** err);
/* g_set_error with printf formatting; does nothing if err is NULL. */
void clo_gerror_set(GError** err, GQuark domain, gint code, const char* fmt, ...)
#if defined(__GNUC__)
	__attribute__((format(printf, 4, 5)))
#endif
	;
/* g_propagate_error: move src into *dest (or free it if dest is NULL). */
void clo_gerror_propagate(GError** dest, GError* src);
/* g_quark_from_static_string for the handful of domains this library uses. */
GQuark clo_quark_from_string(const char* s);
const char* clo_quark_to_string(GQuark q);

#ifdef __cplusplus
}
#endif
#endif /* CLO_USE_GLIB */

/* OpenCL scalar typedefs the API mentions (CL/cl.h is not needed otherwise). */
typedef uint32_t cl_uint;
typedef int32_t cl_int;
typedef uint64_t cl_ulong;
typedef cl_uint cl_bool;
#ifndef CL_TRUE
#define CL_TRUE 1
#define CL_FALSE 0
#endif

#endif
