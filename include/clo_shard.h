/*
 * clo_shard.h — sorting an array whose shards live on the GPUs of one node, behind
 * the C API: most-significant-digit bucket exchange + local satradix sorts
 * (SURVEY.md §8e). NEW functionality: the reference is single-device — its
 * clo_sort_with_host_data creates its queue on device 0 of the context
 * (sort/clo_sort_abstract.c:335) — so nothing here replaces a reference interface;
 * the objects follow the conventions of clo_sort.h (constructor with GError**,
 * destroy, *_with_device_data returning the queue-owned event).
 *
 * One process per GPU. With G = world size (a power of two), rank r ends up
 * holding bucket r — the keys whose top log2(G) bits equal r — sorted; the ranks'
 * results in rank order are the globally sorted array. Per sort:
 *   1. stable local partition into G buckets by the top log2(G) key bits; the
 *      bucket sizes are a by-product              (clo_hip_msd_partition)
 *   2. all-gather of the G counts                  (transport: all_gather_u64)
 *   3. all-to-all(v) of the buckets                (transport: all_to_all_v — over
 *      RCCL one group of ncclSend/ncclRecv pairs, every pair on its own xGMI link)
 *   4. local satradix of what arrived              (clo_sort_with_device_data)
 * The two exchanges go through a small table of functions (CloShardTransport):
 * RCCL in production (clo_shard_transport_new_rccl), anything else that moves the
 * same bytes in tests (two ranks on one GPU cannot use RCCL).
 *
 * Slices (round 3). The local sort of step 4 cannot start before the LAST key has
 * arrived, so with one exchange the links idle while the GPU sorts and the GPU idles
 * while the keys travel. With S = 2, 4 or 8 slices the partition of step 1 splits by
 * log2(G) + log2(S) top bits instead: every rank's bucket comes as S sub-buckets (key
 * ranges in ascending order). Sub-bucket j of every rank travels as one all-to-all(v)
 * on a stream of its own, and the local sort of sub-bucket j (in place, at its place in
 * the result) runs on cq_exec while sub-bucket j + 1 travels. Result and interface are
 * unchanged; `slices=1` in the options gives the single exchange.
 *
 * Failing together. A rank that fails before a collective its peers enter would leave
 * them waiting inside RCCL for ever. So nothing that can fail on one rank alone sits
 * between "decided to communicate" and the collective: every rank ALWAYS joins the
 * count exchange and contributes a status word with its counts (0 = fine); after it
 * every rank knows of every failure and all return an error without touching the key
 * exchange; whatever is decided from the count matrix (bucket too large for one GPU,
 * who has to grow its receive buffer) is decided identically everywhere, and growing a
 * buffer — the one local step left — is followed by a second agreement round. A failure
 * that still strikes in between (the runtime refusing to enqueue the exchange) aborts the
 * transport (ncclCommAbort), which fails the peers' pending operations.
 */
#ifndef CLO_SHARD_H
#define CLO_SHARD_H

#include "clo_common.h"
#include "clo_sort.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct clo_shard_transport {
	void* user;
	int rank, world;
	/* Every rank contributes `count` uint64 (device memory) and receives
	 * world * count of them, in rank order. Ordered on `stream`. 0 or a clo_hip status. */
	int (*all_gather_u64)(void* user, const uint64_t* send_dev, uint64_t* recv_dev, size_t count, void* stream);
	/* Bytes [send_offset[p], +send_bytes[p]) of send_dev go to rank p and land at
	 * its [recv_offset[me], +recv_bytes[me]) (arrays of `world` entries, host
	 * memory; the entry of this rank itself is a device-to-device copy). Ordered on `stream`. */
	int (*all_to_all_v)(void* user, const void* send_dev, const size_t* send_bytes, const size_t* send_offset,
		void* recv_dev, const size_t* recv_bytes, const size_t* recv_offset, void* stream);
	void (*destroy)(void* user);   /* may be NULL */
	/* Ends the transport at once so that operations of it pending on ANY rank fail instead of
	 * waiting for this rank (RCCL: ncclCommAbort). May be NULL. After it only destroy is called. */
	void (*abort)(void* user);
} CloShardTransport;

/* The 128-byte id one rank creates and every rank passes to
 * clo_shard_transport_new_rccl (carried to the others by the launcher's side
 * channel: a file, an environment variable, torch.distributed ...). */
#define CLO_SHARD_RCCL_ID_BYTES 128
cl_bool clo_shard_rccl_unique_id(void* id_out, GError** err);
/* RCCL over xGMI, on the calling thread's current HIP device. Collective: every rank calls it. */
CloShardTransport* clo_shard_transport_new_rccl(const void* id, int rank, int world, GError** err);
void clo_shard_transport_destroy(CloShardTransport* t);

typedef struct clo_shard_sort CloShardSort;

/* elem_type: CLO_UINT or CLO_ULONG (the key is the whole element); options: the
 * satradix options of the local sort ("radix=16" ...) and `slices=S` (1, 2, 4 or 8
 * sub-buckets per rank, log2(world) + log2(S) <= 6; default 4 above 2^22 elements per
 * rank, else 1). The transport stays the caller's. */
CloShardSort* clo_shard_sort_new(CCLContext* ctx, CloShardTransport* transport, CloType elem_type,
	const char* options, GError** err);
void clo_shard_sort_destroy(CloShardSort* ss);

/* Sorts the global array whose shard on this rank is data_in[0 .. numel) (left
 * untouched). On return *data_out is a buffer OWNED BY THE OBJECT (valid until the
 * next call or the destroy) whose first *numel_out elements are this rank's bucket,
 * sorted. Synchronises the host once (the bucket sizes must reach it to size the
 * exchange); the local sort is asynchronous on cq_exec like clo_sort_with_device_data.
 * COLLECTIVE: every rank of the transport calls it, and all of them either succeed or
 * return NULL with an error set (see "Failing together" above). */
CCLEvent* clo_shard_sort_with_device_data(CloShardSort* ss, CCLQueue* cq_exec, CCLBuffer* data_in, size_t numel,
	CCLBuffer** data_out, size_t* numel_out, GError** err);

/* Host milliseconds the phases of the last call took up to their enqueue/sync
 * points and, when cq_exec profiles, device milliseconds: [0] partition, [1] count
 * exchange, [2] key exchange, [3] local sort. */
void clo_shard_sort_get_phase_ms(CloShardSort* ss, double device_ms[4]);

/* The key exchange of the last call, for rooflines: bytes this rank sent to / received
 * from OTHER ranks and the device milliseconds from the start of the first all-to-all(v)
 * to the end of the last (0 when the world is one rank). Synchronises like get_phase_ms. */
void clo_shard_sort_get_exchange(CloShardSort* ss, size_t* bytes_out, size_t* bytes_in, double* device_ms, int* slices);

/* Pure host logic, exposed for tests: from the G x G matrix counts[src * G + bucket]
 * the four arrays of rank `rank` (elements): send counts / offsets, receive counts / offsets. */
void clo_shard_plan(const uint64_t* counts, int world, int rank,
	size_t* send_counts, size_t* send_offsets, size_t* recv_counts, size_t* recv_offsets);
/* The same for a sliced sort: counts[src * row + bucket * slices + j] (row = the stride
 * of a rank's row, >= world * slices); the four arrays of slice j (`world` entries each,
 * elements; offsets into the partitioned shard and into the result). *slice_offset (may
 * be NULL) = where sub-bucket j starts in the result, *slice_total its size. Returns the
 * size of this rank's whole bucket. */
size_t clo_shard_plan_slice(const uint64_t* counts, size_t row, int world, int slices, int rank, int j,
	size_t* send_counts, size_t* send_offsets, size_t* recv_counts, size_t* recv_offsets,
	size_t* slice_offset, size_t* slice_total);

#ifdef __cplusplus
}
#endif
#endif
