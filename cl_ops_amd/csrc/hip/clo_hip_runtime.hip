// clo_hip_runtime.hip — device/stream/memory/event part of the thin C-ABI over
// HIP (include/clo_hip.h). Stands where cf4ocl2's context/queue/buffer/event
// wrappers stood upstream; no kernels here.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "clo_hip.h"
#include "clo_hip_internal.h"

namespace {
struct timing_rec { std::string label; hipEvent_t e0, e1; };
std::mutex g_timing_mu;
std::vector<timing_rec> g_timing;
bool g_timing_on = false;
thread_local clo_hip_launch_observer g_observer = nullptr;
thread_local void* g_observer_user = nullptr;
}  // namespace

void clo_timing_begin(const char* label, hipStream_t s) {
	if (g_observer) g_observer(g_observer_user, label, 0, (void*) s);
	if (!g_timing_on) return;
	std::lock_guard<std::mutex> lk(g_timing_mu);
	timing_rec r;
	r.label = label;
	if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
	(void) hipEventRecord(r.e0, s);
	g_timing.push_back(r);
}

void clo_timing_end(hipStream_t s) {
	if (g_timing_on) {
		std::lock_guard<std::mutex> lk(g_timing_mu);
		if (!g_timing.empty()) (void) hipEventRecord(g_timing.back().e1, s);
	}
	if (g_observer) g_observer(g_observer_user, nullptr, 1, (void*) s);
}

extern "C" {

int clo_hip_set_launch_observer(clo_hip_launch_observer fn, void* user) {
	g_observer = fn;
	g_observer_user = fn ? user : nullptr;
	return 0;
}

int clo_hip_timing_enabled(void) { return g_timing_on ? 1 : 0; }

int clo_hip_timing_enable(int on) {
	std::lock_guard<std::mutex> lk(g_timing_mu);
	g_timing_on = on != 0;
	return 0;
}

int clo_hip_timing_reset(void) {
	std::lock_guard<std::mutex> lk(g_timing_mu);
	for (auto& r : g_timing) { (void) hipEventDestroy(r.e0); (void) hipEventDestroy(r.e1); }
	g_timing.clear();
	return 0;
}

int clo_hip_timing_read(const char* label, unsigned* count, float* total_ms) {
	if (!label || !count || !total_ms) return CLO_HIP_EARGS;
	std::lock_guard<std::mutex> lk(g_timing_mu);
	*count = 0;
	*total_ms = 0.f;
	for (auto& r : g_timing) {
		if (r.label != label) continue;
		hipError_t e = hipEventSynchronize(r.e1);
		if (e != hipSuccess) return (int) e;
		float ms = 0.f;
		e = hipEventElapsedTime(&ms, r.e0, r.e1);
		if (e != hipSuccess) return (int) e;
		*total_ms += ms;
		++*count;
	}
	return 0;
}

int clo_hip_device_count(int* count) {
	if (!count) return CLO_HIP_EARGS;
	hipError_t e = hipGetDeviceCount(count);
	if (e != hipSuccess) { *count = 0; }
	return (int) e;
}

int clo_hip_set_device(int device) { return (int) hipSetDevice(device); }
int clo_hip_get_device(int* device) { return device ? (int) hipGetDevice(device) : CLO_HIP_EARGS; }

int clo_hip_get_device_props(int device, clo_hip_device_props* props) {
	if (!props) return CLO_HIP_EARGS;
	hipDeviceProp_t p;
	hipError_t e = hipGetDeviceProperties(&p, device);
	if (e != hipSuccess) return (int) e;
	memset(props, 0, sizeof(*props));
	snprintf(props->name, sizeof(props->name), "%s", p.name);
	snprintf(props->gcn_arch, sizeof(props->gcn_arch), "%s", p.gcnArchName);
	props->compute_units = p.multiProcessorCount;
	props->max_threads_per_block = p.maxThreadsPerBlock;
	props->wavefront_size = p.warpSize;
	props->lds_bytes_per_block = p.sharedMemPerBlock;
	props->global_mem_bytes = p.totalGlobalMem;
	return 0;
}

int clo_hip_stream_create(void** stream) {
	if (!stream) return CLO_HIP_EARGS;
	hipStream_t s;
	hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	*stream = (e == hipSuccess) ? (void*) s : nullptr;
	return (int) e;
}
int clo_hip_stream_create_high_priority(void** stream) {
	if (!stream) return CLO_HIP_EARGS;
	int least = 0, greatest = 0;   // (numerically lower = higher priority)
	hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
	if (e != hipSuccess) return (int) e;
	hipStream_t s;
	e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest);
	*stream = (e == hipSuccess) ? (void*) s : nullptr;
	return (int) e;
}
int clo_hip_stream_destroy(void* stream) { return (int) hipStreamDestroy((hipStream_t) stream); }
int clo_hip_stream_synchronize(void* stream) { return (int) hipStreamSynchronize((hipStream_t) stream); }
int clo_hip_stream_query(void* stream) {
	const hipError_t e = hipStreamQuery((hipStream_t) stream);
	if (e == hipErrorNotReady) { (void) hipGetLastError(); return CLO_HIP_ENOTREADY; }   // (not an error: nothing to leave behind)
	return (int) e;
}

int clo_hip_malloc(void** dptr, size_t bytes) {
	if (!dptr) return CLO_HIP_EARGS;
	*dptr = nullptr;
	if (bytes == 0) bytes = 4;
	return (int) hipMalloc(dptr, bytes);
}
int clo_hip_free(void* dptr) { return dptr ? (int) hipFree(dptr) : 0; }

int clo_hip_memcpy_h2d_async(void* dst, const void* src, size_t bytes, void* stream) {
	return (int) hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t) stream);
}
int clo_hip_memcpy_d2h_async(void* dst, const void* src, size_t bytes, void* stream) {
	return (int) hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t) stream);
}
int clo_hip_memcpy_d2d_async(void* dst, const void* src, size_t bytes, void* stream) {
	return (int) hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t) stream);
}
int clo_hip_memset_async(void* dst, int value, size_t bytes, void* stream) {
	return (int) hipMemsetAsync(dst, value, bytes, (hipStream_t) stream);
}

int clo_hip_host_register(void* host_ptr, size_t bytes) {
	if (!host_ptr || bytes == 0) return CLO_HIP_EARGS;
	return (int) hipHostRegister(host_ptr, bytes, hipHostRegisterDefault);
}
int clo_hip_host_unregister(void* host_ptr) { return host_ptr ? (int) hipHostUnregister(host_ptr) : 0; }

int clo_hip_event_create(void** event) {
	if (!event) return CLO_HIP_EARGS;
	hipEvent_t ev;
	hipError_t e = hipEventCreate(&ev);
	*event = (e == hipSuccess) ? (void*) ev : nullptr;
	return (int) e;
}
int clo_hip_event_destroy(void* event) { return event ? (int) hipEventDestroy((hipEvent_t) event) : 0; }
int clo_hip_event_record(void* event, void* stream) {
	return (int) hipEventRecord((hipEvent_t) event, (hipStream_t) stream);
}
// ---- environment switches: one reader (clo_hip_internal.h has the list) ----
static clo_hip_env_t g_env = { CLO_MAX_SPINS, 2, 0, 0, 1 };
static int g_env_read = 0;
void clo_hip_env_refresh(void) {
	clo_hip_env_t e = { CLO_MAX_SPINS, 2, 0, 0, 1 };
	if (const char* m = getenv("CLO_MAX_SPINS")) e.max_spins = (unsigned) strtoul(m, nullptr, 10);
	if (const char* m = getenv("CLO_RADIX_SWEEP")) e.radix_sweep = atoi(m) != 0 ? 1 : 0;
	if (const char* m = getenv("CLO_R1_POOLS")) e.r1_pools = atoi(m);
	e.no_digits = getenv("CLO_RADIX_NO_DIGITS") != nullptr;
	if (const char* m = getenv("CLO_BITONIC_MERGE2")) e.bitonic_merge2 = atoi(m);
	// (objects may be created on several threads while others sort: field by field, never a torn struct — the values
	// only differ from the ones already there when the environment changed between two object creations)
	__atomic_store_n(&g_env.max_spins, e.max_spins, __ATOMIC_RELAXED);
	__atomic_store_n(&g_env.radix_sweep, e.radix_sweep, __ATOMIC_RELAXED);
	__atomic_store_n(&g_env.r1_pools, e.r1_pools, __ATOMIC_RELAXED);
	__atomic_store_n(&g_env.no_digits, e.no_digits, __ATOMIC_RELAXED);
	__atomic_store_n(&g_env.bitonic_merge2, e.bitonic_merge2, __ATOMIC_RELAXED);
	__atomic_store_n(&g_env_read, 1, __ATOMIC_RELEASE);
}
int clo_hip_event_synchronize(void* event) { return (int) hipEventSynchronize((hipEvent_t) event); }
int clo_hip_event_query(void* event) { return event ? (int) hipEventQuery((hipEvent_t) event) : CLO_HIP_EARGS; }
int clo_hip_event_elapsed_ms(void* start, void* stop, float* ms) {
	return ms ? (int) hipEventElapsedTime(ms, (hipEvent_t) start, (hipEvent_t) stop) : CLO_HIP_EARGS;
}
int clo_hip_stream_wait_event(void* stream, void* event) {
	return (int) hipStreamWaitEvent((hipStream_t) stream, (hipEvent_t) event, 0);
}

int clo_hip_graph_capture_begin(void* stream) {
	return (int) hipStreamBeginCapture((hipStream_t) stream, hipStreamCaptureModeThreadLocal);
}

int clo_hip_stream_is_capturing(void* stream) {
	hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
	if (hipStreamIsCapturing((hipStream_t) stream, &st) != hipSuccess) { (void) hipGetLastError(); return 1; }   // unknown: do not capture
	return st != hipStreamCaptureStatusNone ? 1 : 0;
}

int clo_hip_graph_capture_end(void* stream, void** graph_exec) {
	if (!graph_exec) return CLO_HIP_EARGS;
	*graph_exec = nullptr;
	hipGraph_t graph = nullptr;
	hipError_t e = hipStreamEndCapture((hipStream_t) stream, &graph);
	if (e != hipSuccess) return (int) e;
	hipGraphExec_t exec = nullptr;
	e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
	(void) hipGraphDestroy(graph);
	if (e != hipSuccess) return (int) e;
	*graph_exec = exec;
	return 0;
}

int clo_hip_graph_launch(void* graph_exec, void* stream) {
	return graph_exec ? (int) hipGraphLaunch((hipGraphExec_t) graph_exec, (hipStream_t) stream) : CLO_HIP_EARGS;
}

int clo_hip_graph_destroy(void* graph_exec) {
	return graph_exec ? (int) hipGraphExecDestroy((hipGraphExec_t) graph_exec) : 0;
}

const char* clo_hip_error_string(int status) {
	switch (status) {
		case 0: return "success";
		case CLO_HIP_EARGS: return "clo_hip: invalid arguments";
		case CLO_HIP_EUNSUPPORTED: return "clo_hip: unsupported type or option";
		case CLO_HIP_EWORKSPACE: return "clo_hip: workspace too small";
		case CLO_HIP_ETIMEOUT: return "clo_hip: in-kernel look-back spin timed out";
		case CLO_HIP_ENOTREADY: return "clo_hip: work on the stream is still running";
		default:
			if (status <= CLO_HIP_ERCCL) return "clo_hip: RCCL reported an error (status = CLO_HIP_ERCCL - ncclResult_t)";
			return status > 0 ? hipGetErrorString((hipError_t) status) : "clo_hip: unknown error";
	}
}

int clo_hip_check_status(void* workspace, void* stream) {
	if (!workspace) return CLO_HIP_EARGS;
	unsigned int word = 0;
	hipError_t e = hipMemcpyAsync(&word, (char*) workspace + CLO_WS_STATUS_OFFSET, sizeof(word),
		hipMemcpyDeviceToHost, (hipStream_t) stream);
	if (e != hipSuccess) return (int) e;
	e = hipStreamSynchronize((hipStream_t) stream);
	if (e != hipSuccess) return (int) e;
	return word ? CLO_HIP_ETIMEOUT : 0;
}

}  // extern "C"

const clo_hip_env_t* clo_hip_env() {
	if (!g_env_read) clo_hip_env_refresh();
	return &g_env;
}

