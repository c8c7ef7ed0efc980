/*
 * clo_ccl.h — the part of the cf4ocl2 object model that the cl_ops sort/scan
 * API is expressed in (CCLContext, CCLQueue, CCLBuffer, CCLEvent, CCLProf ...),
 * re-implemented over HIP.
 *
 * Upstream these come from <cf4ocl2.h> (src/cl_ops/common/clo_common.in.h:29)
 * and wrap OpenCL objects. Here each is a small C struct over the thin C-ABI of
 * clo_hip.h: a context is one HIP device, a queue is one HIP stream, a buffer
 * is one device allocation, an event is a pair of HIP events. Only the calls
 * the hot path and its callers use are provided (census in SURVEY.md §8b):
 *   sort/clo_sort_abstract.c:130,226,335-395; scan/clo_scan_abstract.c:106,290-339;
 *   sort/clo_sort_sbitonic.c:60-95; sort/clo_sort_satradix.c:176-257;
 *   benchmarks/clo_sort_bench.c:148-208.
 * Names and argument order follow cf4ocl2 so that reference-side code
 * compiles against this header unchanged for that subset.
 */
#ifndef CLO_CCL_H
#define CLO_CCL_H

#include "clo_glib_compat.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ccl_context CCLContext;
typedef struct ccl_device CCLDevice;
typedef struct ccl_queue CCLQueue;
typedef struct ccl_buffer CCLBuffer;
typedef struct ccl_event CCLEvent;
typedef struct ccl_program CCLProgram;
typedef struct ccl_prof CCLProf;
/* cf4ocl2: an event wait list is an opaque growable array. */
typedef struct ccl_event_wait_list* CCLEventWaitList;

/* Error domain of runtime (HIP) failures; upstream this is CCL_OCL_ERROR. */
#define CCL_HIP_ERROR ccl_hip_error_quark()
GQuark ccl_hip_error_quark(void);

/* cl_command_queue_properties / cl_mem_flags bits that callers pass. */
#define CL_QUEUE_PROFILING_ENABLE (1u << 1)
#define CL_MEM_READ_WRITE (1u << 0)
#define CL_MEM_WRITE_ONLY (1u << 1)
#define CL_MEM_READ_ONLY (1u << 2)

/* ---- context / device ---- */
/* ccl_context_new_gpu: first GPU. */
CCLContext* ccl_context_new_gpu(GError** err);
/* ccl_context_new_from_menu_full(&dev_idx): device chosen by index, -1 = 0
 * (benchmarks/clo_sort_bench.c:148; no interactive menu here). */
CCLContext* ccl_context_new_from_menu_full(void* dev_idx_ptr, GError** err);
/* Not in cf4ocl2: context on HIP device `device_index`. */
CCLContext* ccl_context_new_from_device_index(int device_index, GError** err);
/* Not in cf4ocl2: a context with no device behind it. Sorters and scanners can
 * be constructed on it and introspected (option validation, kernel names, key
 * parsing) — nothing that touches a GPU; creating a queue or a buffer on it
 * fails with a CCL_HIP_ERROR. Lets host-side logic be tested on machines
 * without a GPU; it is never a compute path. */
CCLContext* ccl_context_new_offline(GError** err);
void ccl_context_ref(CCLContext* ctx);
void ccl_context_unref(CCLContext* ctx);
void ccl_context_destroy(CCLContext* ctx);
CCLDevice* ccl_context_get_device(CCLContext* ctx, cl_uint index, GError** err);
/* Not in cf4ocl2: HIP device ordinal and limits. */
int ccl_device_get_index(CCLDevice* dev);
size_t ccl_device_get_max_work_group_size(CCLDevice* dev);
const char* ccl_device_get_name(CCLDevice* dev);

/* ---- queue ---- */
CCLQueue* ccl_queue_new(CCLContext* ctx, CCLDevice* dev, cl_ulong properties, GError** err);
/* Not in cf4ocl2: adopt an existing HIP stream (e.g. torch's current stream);
 * the queue does not own it. */
CCLQueue* ccl_queue_new_from_stream(CCLContext* ctx, void* hip_stream, cl_ulong properties, GError** err);
void ccl_queue_destroy(CCLQueue* cq);
CCLDevice* ccl_queue_get_device(CCLQueue* cq, GError** err);
CCLContext* ccl_queue_get_context(CCLQueue* cq, GError** err);
cl_bool ccl_queue_finish(CCLQueue* cq, GError** err);
/* Release the events the queue has handed out (cf4ocl2: ccl_queue_gc). A queue
 * created without CL_QUEUE_PROFILING_ENABLE also drops events by itself once
 * more than 128 have piled up, keeping the 64 most recent: an event pointer is
 * good for the next 64 commands on its queue. */
void ccl_queue_gc(CCLQueue* cq);
/* Not in cf4ocl2: the underlying hipStream_t. */
void* ccl_queue_get_stream(CCLQueue* cq);

/* ---- buffer ---- */
CCLBuffer* ccl_buffer_new(CCLContext* ctx, cl_ulong flags, size_t size, void* host_ptr, GError** err);
/* Not in cf4ocl2: wrap device memory owned by someone else (a torch tensor). */
CCLBuffer* ccl_buffer_new_from_device_ptr(CCLContext* ctx, void* device_ptr, size_t size, GError** err);
void ccl_buffer_destroy(CCLBuffer* buf);
size_t ccl_buffer_get_size(CCLBuffer* buf);
void* ccl_buffer_get_device_ptr(CCLBuffer* buf);
CCLEvent* ccl_buffer_enqueue_write(CCLBuffer* buf, CCLQueue* cq, cl_bool blocking, size_t offset,
	size_t size, void* ptr, CCLEventWaitList* evt_wait_lst, GError** err);
CCLEvent* ccl_buffer_enqueue_read(CCLBuffer* buf, CCLQueue* cq, cl_bool blocking, size_t offset,
	size_t size, void* ptr, CCLEventWaitList* evt_wait_lst, GError** err);
CCLEvent* ccl_buffer_enqueue_copy(CCLBuffer* src, CCLBuffer* dst, CCLQueue* cq, size_t src_offset,
	size_t dst_offset, size_t size, CCLEventWaitList* evt_wait_lst, GError** err);

/* ---- events ---- */
void ccl_event_set_name(CCLEvent* evt, const char* name);
const char* ccl_event_get_name(CCLEvent* evt);
/* Wait for every event in the list, then clear the list. */
cl_bool ccl_event_wait(CCLEventWaitList* evt_wait_lst, GError** err);
/* ccl_ewl(&ewl, evt1, ..., NULL): append and return the list. */
CCLEventWaitList* ccl_ewl(CCLEventWaitList* ewl, ...);
void ccl_event_wait_list_add(CCLEventWaitList* ewl, ...);
void ccl_event_wait_list_clear(CCLEventWaitList* ewl);

/* ---- program (no JIT here: a token naming the ahead-of-time kernel set) ---- */
CCLProgram* ccl_program_new_token(CCLContext* ctx, const char* what, const char* build_options);
void ccl_program_destroy(CCLProgram* prg);
const char* ccl_program_get_build_options(CCLProgram* prg);

/* ---- profiling of a queue created with CL_QUEUE_PROFILING_ENABLE ---- */
CCLProf* ccl_prof_new(void);
void ccl_prof_destroy(CCLProf* prof);
void ccl_prof_add_queue(CCLProf* prof, const char* name, CCLQueue* cq);
cl_bool ccl_prof_calc(CCLProf* prof, GError** err);
/* Device time from the start of the first to the end of the last command of
 * the added queues, in nanoseconds (benchmarks/clo_sort_bench.c:201-208). */
cl_ulong ccl_prof_get_duration(CCLProf* prof);
/* Aggregate of the events carrying one name (cf4ocl2: CCLProfAgg): total device
 * time of the kernels upstream enqueues under that name, e.g. "satradix_scatter"
 * (sort/clo_sort_satradix.c:312) or "clo_scan_blelloch_wgscan"
 * (scan/clo_scan_blelloch.c:158). Valid until the next ccl_prof_calc. */
typedef struct ccl_prof_agg {
	const char* event_name;
	cl_ulong absolute_time;   /* ns */
	double relative_time;     /* share of ccl_prof_get_duration */
} CCLProfAgg;
const CCLProfAgg* ccl_prof_get_agg(CCLProf* prof, const char* event_name);
void ccl_prof_iter_agg_init(CCLProf* prof, int sort);
const CCLProfAgg* ccl_prof_iter_agg_next(CCLProf* prof);

/* Internal to the library's own drivers (clo_sort_*.c / clo_scan_*.c): open and
 * close an event around a group of launches on the queue's stream. An event
 * whose command could not be enqueued is taken back with ccl_queue_abort_command
 * (it would otherwise sit in the queue with an `end` that was never recorded). */
CCLEvent* ccl_queue_begin_command(CCLQueue* cq, const char* name, GError** err);
/* The same for a command that follows `after` directly in a group of launches:
 * its profiled time starts where `after` ended (one marker between two kernels). */
CCLEvent* ccl_queue_begin_command_after(CCLQueue* cq, const char* name, CCLEvent* after, GError** err);
cl_bool ccl_queue_end_command(CCLQueue* cq, CCLEvent* evt, GError** err);
void ccl_queue_abort_command(CCLQueue* cq, CCLEvent* evt);
int ccl_queue_is_profiling(CCLQueue* cq);
/* For the library's own objects that remember the queue of their last call
 * (clo_stream_guard): keep the queue STRUCT alive past ccl_queue_destroy, which
 * still synchronises and closes the queue at once. */
void clo_queue_hold(CCLQueue* cq);
void clo_queue_drop(CCLQueue* cq);
int clo_queue_is_closed(CCLQueue* cq);
/* A kernel that waits for other work-groups bounds its spins and raises a
 * status word in device memory when it gives up (clo_hip.h: clo_hip_check_status).
 * The owner of that word wraps it in a cell and has every queue it enqueues on
 * watch the cell; ccl_queue_finish, ccl_event_wait and ccl_prof_calc then read
 * the word after synchronising and fail with CLO_ERROR_LIBRARY if it is raised. */
typedef struct clo_status_cell clo_status_cell;
clo_status_cell* clo_status_cell_new(void* dev_word);
void clo_status_cell_set_word(clo_status_cell* cell, void* dev_word);   /* NULL: the memory is gone */
int clo_status_cell_take_tripped(clo_status_cell* cell);               /* 1 once after a check found the word raised */
void clo_status_cell_unref(clo_status_cell* cell);
void ccl_queue_watch_status(CCLQueue* cq, clo_status_cell* cell);
/* Make the queue's stream wait for every event of the list (does not clear). */
cl_bool ccl_queue_wait_for(CCLQueue* cq, CCLEventWaitList* ewl, GError** err);

#ifdef __cplusplus
}
#endif
#endif
