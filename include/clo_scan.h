/*
 * clo_scan.h — the CloScan plugin API, as exported by the reference's
 * src/cl_ops/scan/clo_scan_abstract.in.h:41-162, implemented over HIP.
 * Algorithm registered: "blelloch" (clo_scan_blelloch.in.h:46).
 *
 * Divergence (deliberate): the whole array is scanned for any numel; upstream
 * never scans the tail numel % (2*lws) (clo_scan_blelloch.cl:70).
 */
#ifndef CLO_SCAN_H
#define CLO_SCAN_H

#include "clo_common.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CLO_SCAN_IMPLS "blelloch"

/* clo_scan_abstract.in.h:41-103 */
typedef struct clo_scan_impl_def {
	const char* name;
	const char* (*init)(CloScan* scanner, const char* options, GError** err);
	void (*finalize)(CloScan* scan);
	CCLEvent* (*scan_with_device_data)(CloScan* scanner, CCLQueue* cq_exec, CCLQueue* cq_comm,
		CCLBuffer* data_in, CCLBuffer* data_out, size_t numel, size_t lws_max, GError** err);
	cl_uint (*get_num_kernels)(CloScan* scanner, GError** err);
	const char* (*get_kernel_name)(CloScan* scanner, cl_uint i, GError** err);
	size_t (*get_localmem_usage)(CloScan* scanner, cl_uint i, size_t lws_max, size_t numel, GError** err);
} CloScanImplDef;

/* clo_scan_abstract.in.h:109-162 — note elem_type/sum_type BY VALUE. */
CloScan* clo_scan_new(const char* type, const char* options, CCLContext* ctx,
	CloType elem_type, CloType sum_type, const char* compiler_opts, GError** err);
void clo_scan_destroy(CloScan* scan);
CCLEvent* clo_scan_with_device_data(CloScan* scanner, CCLQueue* cq_exec, CCLQueue* cq_comm,
	CCLBuffer* data_in, CCLBuffer* data_out, size_t numel, size_t lws_max, GError** err);
cl_bool clo_scan_with_host_data(CloScan* scanner, CCLQueue* cq_exec, CCLQueue* cq_comm,
	void* data_in, void* data_out, size_t numel, size_t lws_max, GError** err);
CCLContext* clo_scan_get_context(CloScan* scanner);
CCLProgram* clo_scan_get_program(CloScan* scanner);
CloType clo_scan_get_elem_type(CloScan* scanner);
size_t clo_scan_get_element_size(CloScan* scanner);
CloType clo_scan_get_sum_type(CloScan* scanner);
size_t clo_scan_get_sum_size(CloScan* scanner);
void* clo_scan_get_data(CloScan* scanner);
void clo_scan_set_data(CloScan* scanner, void* data);
cl_uint clo_scan_get_num_kernels(CloScan* scanner, GError** err);
const char* clo_scan_get_kernel_name(CloScan* scanner, cl_uint i, GError** err);
size_t clo_scan_get_localmem_usage(CloScan* scanner, cl_uint i, size_t lws_max, size_t numel, GError** err);

extern const CloScanImplDef clo_scan_blelloch_def;  /* clo_scan_blelloch.in.h:46 */

/* clo_scan_blelloch.in.h:33-43 */
#define CLO_SCAN_BLELLOCH_NUM_KERNELS 3
#define CLO_SCAN_BLELLOCH_KNAME_WGSCAN "workgroupScan"
#define CLO_SCAN_BLELLOCH_KNAME_WGSUMSSCAN "workgroupSumsScan"
#define CLO_SCAN_BLELLOCH_KNAME_ADDWGSUMS "addWorkgroupSums"

#ifdef __cplusplus
}
#endif
#endif
