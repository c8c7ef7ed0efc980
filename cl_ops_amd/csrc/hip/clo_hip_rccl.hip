// clo_hip_rccl.hip — the RCCL part of the thin C-ABI (include/clo_hip.h): what the
// sharded sort (include/clo_shard.h) exchanges between the GPUs of a node over
// xGMI. New functionality — the reference is single-device
// (sort/clo_sort_abstract.c:335 creates its queue on device 0).
//
// Two collectives, both stream-ordered on the caller's stream:
//   * all-gather of a few uint64 per rank (the bucket counts);
//   * all-to-all(v) as ONE group of ncclSend / ncclRecv pairs: xGMI is point to
//     point, every pair of GPUs has its own link, so the world-1 transfers of a
//     rank run side by side; peers are visited in ring order (rank + k, rank - k)
//     so that all ranks post matching operations in a compatible order.
//
// RCCL is NOT a link-time dependency: it is looked up when the first of these
// calls is made. A process that also runs PyTorch already holds a copy of RCCL
// (torch ships its own librccl.so); loading ROCm's next to it put two RCCLs in one
// process, and their exit handlers freed the same things twice ("double free or
// corruption" when the interpreter shut down). So: the copy already in the
// process if there is one, else ROCm's librccl.so.1.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "clo_hip.h"

static_assert(sizeof(ncclUniqueId) == CLO_HIP_RCCL_ID_BYTES, "clo_hip.h carries the id as 128 opaque bytes");

namespace {
// RCCL statuses travel as negative numbers below the CLO_HIP_E* range
inline int rccl_status(ncclResult_t r) { return r == ncclSuccess ? 0 : CLO_HIP_ERCCL - (int) r; }

struct rccl_api {
	decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&::ncclCommInitRank) CommInitRank = nullptr;
	decltype(&::ncclCommDestroy) CommDestroy = nullptr;
	decltype(&::ncclCommAbort) CommAbort = nullptr;
	decltype(&::ncclCommGetAsyncError) CommGetAsyncError = nullptr;
	decltype(&::ncclAllGather) AllGather = nullptr;
	decltype(&::ncclSend) Send = nullptr;
	decltype(&::ncclRecv) Recv = nullptr;
	decltype(&::ncclGroupStart) GroupStart = nullptr;
	decltype(&::ncclGroupEnd) GroupEnd = nullptr;
	bool ok = false;
};

const rccl_api& rccl() {
	static rccl_api api;
	static std::once_flag once;
	std::call_once(once, [] {
		void* h = nullptr;
		for (const char* name : { "librccl.so.1", "librccl.so" }) if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);   // already here (torch's)?
		for (const char* name : { "librccl.so.1", "/opt/rocm/lib/librccl.so.1" }) if (!h) h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
		if (!h) return;
		#define CLO_RCCL_SYM(n) api.n = (decltype(api.n)) dlsym(h, "nccl" #n)
		CLO_RCCL_SYM(GetUniqueId); CLO_RCCL_SYM(CommInitRank); CLO_RCCL_SYM(CommDestroy); CLO_RCCL_SYM(CommAbort); CLO_RCCL_SYM(CommGetAsyncError); CLO_RCCL_SYM(AllGather);
		CLO_RCCL_SYM(Send); CLO_RCCL_SYM(Recv); CLO_RCCL_SYM(GroupStart); CLO_RCCL_SYM(GroupEnd);
		#undef CLO_RCCL_SYM
		api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.Send && api.Recv && api.GroupStart && api.GroupEnd;
	});
	return api;
}
}

extern "C" {

int clo_hip_rccl_unique_id(void* id_out) {
	if (!id_out) return CLO_HIP_EARGS;
	if (!rccl().ok) return CLO_HIP_EUNSUPPORTED;
	ncclUniqueId id;
	const ncclResult_t r = rccl().GetUniqueId(&id);
	if (r == ncclSuccess) memcpy(id_out, &id, sizeof(id));
	return rccl_status(r);
}

int clo_hip_rccl_comm_create(void** comm, const void* id_in, int rank, int world) {
	if (!comm || !id_in || world < 1 || rank < 0 || rank >= world) return CLO_HIP_EARGS;
	if (!rccl().ok) return CLO_HIP_EUNSUPPORTED;
	ncclUniqueId id;
	memcpy(&id, id_in, sizeof(id));
	ncclComm_t c = nullptr;
	const ncclResult_t r = rccl().CommInitRank(&c, world, id, rank);   // on the calling thread's current device
	*comm = r == ncclSuccess ? (void*) c : nullptr;
	return rccl_status(r);
}

int clo_hip_rccl_comm_destroy(void* comm) {
	return comm ? rccl_status(rccl().CommDestroy((ncclComm_t) comm)) : 0;
}

int clo_hip_rccl_comm_abort(void* comm) {
	if (!comm) return 0;
	if (!rccl().CommAbort) return CLO_HIP_EUNSUPPORTED;
	return rccl_status(rccl().CommAbort((ncclComm_t) comm));
}

int clo_hip_rccl_comm_async_error(void* comm) {
	if (!comm) return CLO_HIP_EARGS;
	if (!rccl().CommGetAsyncError) return 0;   // (an RCCL without it: nothing to report)
	ncclResult_t async = ncclSuccess;
	const ncclResult_t r = rccl().CommGetAsyncError((ncclComm_t) comm, &async);
	return rccl_status(r != ncclSuccess ? r : async);
}

int clo_hip_rccl_all_gather_u64(void* comm, const uint64_t* send_dev, uint64_t* recv_dev, size_t count, void* stream) {
	if (!comm || !send_dev || !recv_dev) return CLO_HIP_EARGS;
	return rccl_status(rccl().AllGather(send_dev, recv_dev, count, ncclUint64, (ncclComm_t) comm, (hipStream_t) stream));
}

int clo_hip_rccl_all_to_all_v(void* comm, int rank, int world,
	const void* send_dev, const size_t* send_bytes, const size_t* send_offset_bytes,
	void* recv_dev, const size_t* recv_bytes, const size_t* recv_offset_bytes, void* stream) {
	if (!comm || !send_bytes || !send_offset_bytes || !recv_bytes || !recv_offset_bytes) return CLO_HIP_EARGS;
	ncclComm_t c = (ncclComm_t) comm;
	hipStream_t s = (hipStream_t) stream;
	// this rank's own bucket never leaves the GPU
	if (send_bytes[rank] != recv_bytes[rank]) return CLO_HIP_EARGS;
	if (send_bytes[rank]) {
		const hipError_t e = hipMemcpyAsync((char*) recv_dev + recv_offset_bytes[rank], (const char*) send_dev + send_offset_bytes[rank],
			send_bytes[rank], hipMemcpyDeviceToDevice, s);
		if (e != hipSuccess) return (int) e;
	}
	ncclResult_t r = rccl().GroupStart();
	for (int k = 1; k < world && r == ncclSuccess; ++k) {
		const int dst = (rank + k) % world, src = (rank - k + world) % world;
		if (send_bytes[dst]) r = rccl().Send((const char*) send_dev + send_offset_bytes[dst], send_bytes[dst], ncclUint8, dst, c, s);
		if (r == ncclSuccess && recv_bytes[src]) r = rccl().Recv((char*) recv_dev + recv_offset_bytes[src], recv_bytes[src], ncclUint8, src, c, s);
	}
	const ncclResult_t r2 = rccl().GroupEnd();
	return rccl_status(r != ncclSuccess ? r : r2);
}

}  // extern "C"
