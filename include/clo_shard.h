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
 * satradix options of the local sort ("radix=16" ...). The transport stays the caller's. */
CloShardSort* clo_shard_sort_new(CCLContext* ctx, CloShardTransport* transport, CloType elem_type,
	const char* options, GError** err);
void clo_shard_sort_destroy(CloShardSort* ss);

/* Sorts the global array whose shard on this rank is data_in[0 .. numel) (left
 * untouched). On return *data_out is a buffer OWNED BY THE OBJECT (valid until the
 * next call or the destroy) whose first *numel_out elements are this rank's bucket,
 * sorted. Synchronises the host once (the bucket sizes must reach it to size the
 * exchange); the local sort is asynchronous on cq_exec like clo_sort_with_device_data. */
CCLEvent* clo_shard_sort_with_device_data(CloShardSort* ss, CCLQueue* cq_exec, CCLBuffer* data_in, size_t numel,
	CCLBuffer** data_out, size_t* numel_out, GError** err);

/* Host milliseconds the phases of the last call took up to their enqueue/sync
 * points and, when cq_exec profiles, device milliseconds: [0] partition, [1] count
 * exchange, [2] key exchange, [3] local sort. */
void clo_shard_sort_get_phase_ms(CloShardSort* ss, double device_ms[4]);

/* Pure host logic, exposed for tests: from the G x G matrix counts[src * G + bucket]
 * the four arrays of rank `rank` (elements): send counts / offsets, receive counts / offsets. */
void clo_shard_plan(const uint64_t* counts, int world, int rank,
	size_t* send_counts, size_t* send_offsets, size_t* recv_counts, size_t* recv_offsets);

#ifdef __cplusplus
}
#endif
#endif
