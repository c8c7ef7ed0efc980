/*
 * clo_common.h — types, utilities and error codes shared by the sort/scan API.
 * Mirrors the reference's src/cl_ops/common/clo_common.in.h:53-165 (names,
 * values and meaning); implementation in cl_ops_amd/csrc/clo_common.c.
 */
#ifndef CLO_COMMON_H
#define CLO_COMMON_H

#include "clo_glib_compat.h"
#include "clo_ccl.h"

#ifdef __cplusplus
extern "C" {
#endif

/* clo_common.in.h:32 */
#define CLO_DEFAULT_SEED 0
/* clo_common.in.h:36 */
#define CLO_ERROR clo_error_quark()

/* clo_common.in.h:53,63,70 */
#define CLO_DIV_CEIL(a, b) (((a) + (b) - 1) / (b))
#define CLO_GWS_MULT(gws, lws) ((lws) * CLO_DIV_CEIL(gws, lws))
#define CLO_IS_PO2(x) (((x) & ((x) - 1)) == 0)

/* clo_common.in.h:80-95 */
enum clo_error_codes {
	CLO_SUCCESS = 0,
	CLO_ERROR_OPENFILE = 1,
	CLO_ERROR_ARGS = 2,
	CLO_ERROR_STREAM_WRITE = 3,
	CLO_ERROR_IMPL_NOT_FOUND = 5,
	CLO_ERROR_UNKNOWN_TYPE = 6,
	CLO_ERROR_LIBRARY = 7
};

/* clo_common.in.h:108-120 */
typedef enum {
	CLO_CHAR = 0,
	CLO_UCHAR = 1,
	CLO_SHORT = 2,
	CLO_USHORT = 3,
	CLO_INT = 4,
	CLO_UINT = 5,
	CLO_LONG = 6,
	CLO_ULONG = 7,
	CLO_HALF = 8,
	CLO_FLOAT = 9,
	CLO_DOUBLE = 10
} CloType;

typedef struct clo_sort CloSort;
typedef struct clo_scan CloScan;

/* clo_common.in.h:139-165 */
const char* clo_type_get_name(CloType type);
size_t clo_type_sizeof(CloType type);
CloType clo_type_by_name(const char* name, GError** err);
unsigned int clo_nlpo2(unsigned int x);
unsigned int clo_ones32(unsigned int x);
unsigned int clo_tzc(int x);
unsigned int clo_sum(unsigned int x);
void clo_print_to_null(const gchar* string);
GQuark clo_error_quark(void);

/* Not upstream: type classification used by the HIP drivers. */
int clo_type_is_signed(CloType type);   /* char, short, int, long */
int clo_type_is_float(CloType type);    /* half, float, double */

#ifdef __cplusplus
}
#endif
#endif
