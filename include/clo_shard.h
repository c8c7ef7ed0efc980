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
 * results in rank order are the globally sorted array. Per sort (round 4):
 *   1. stable local partition by the top 8 key bits into 256 sub-buckets, 256 / G
 *      per destination rank (ascending key ranges); their sizes are a by-product
 *                                                  (clo_hip_msd_partition: ONE radix pass)
 *   2. all-gather of the 256 counts + a status word + the receive capacity per rank
 *                                                  (transport: all_gather_u64)
 *   3. all-to-all(v) of the sub-buckets in S slices (S = 1, 2, 4 or 8; slice j = the
 *      sub-buckets [j * 256 / (G S), (j + 1) * 256 / (G S)) of every rank: one contiguous
 *      range per pair of ranks) on a stream of its own   (transport: all_to_all_v — over
 *      RCCL one group of ncclSend/ncclRecv pairs per slice, every pair on its own xGMI link)
 *   4. per slice, ONE segmented sort of its sub-buckets on the remaining key bits
 *      (clo_hip_radix_sort_segmented: every sub-buckets sorted on its own, in launches
 *      shared by all of them; its first pass gathers a sub-bucket's G pieces — one per
 *      source rank — as a by-product), on cq_exec while slice j + 1 travels.
 * The partition consumes 8 key bits, so the local sorts run one 8-bit pass fewer than a
 * plain sort: partition + local sort together make exactly the passes a single-GPU sort
 * of the same keys makes (4 for uint, 8 for ulong) — the exchange is what a sharded sort
 * adds, not a pass. (Round 3 partitioned on log2(G S) <= 6 bits and sorted every slice
 * on the full key width: one pass more, and S mid-size sorts instead of shared launches.)
 * Arrays below 256 MiB of keys per rank (global mean: where a segmented sort of 256
 * sub-buckets starts to beat a plain sort of the whole bucket, measured; the option
 * `slice_min=<bytes>` moves it) and sorters whose radix is neither 16 nor 256 use one
 * exchange and one plain sort of the whole bucket.
 * The two exchanges go through a small table of functions (CloShardTransport):
 * RCCL in production (clo_shard_transport_new_rccl), anything else that moves the
 * same bytes in tests (two ranks on one GPU cannot use RCCL).
 *
 * Slices. The local sort of a bucket cannot start before its LAST key has arrived, so
 * with one exchange the links idle while the GPU sorts and the GPU idles while the keys
 * travel. With S slices the sort of slice j runs while slice j + 1 travels. How many
 * slices pay depends on the links' rate against the sort's, which only a run shows:
 * `slices=auto` (the default) times every sliced call on the device, shares the times
 * through the count exchange (so that every rank holds the same table and decides the
 * same), tries 4, 2, 1 and 8 slices twice each for a size class and then keeps the
 * fastest; `slices=1|2|4|8` fixes the number.
 *
 * Failing together. A rank that fails before a collective its peers enter would leave
 * them waiting inside RCCL for ever. So nothing that can fail on one rank alone sits
 * between "decided to communicate" and the collective: buffers (for the rank's receive
 * capacity, the sorter's included), events and the transfer stream are made BEFORE the
 * count exchange, every rank ALWAYS joins it and contributes a status word with its counts
 * (0 = fine); after it every rank knows of every failure and all return an error without
 * touching the key exchange; whatever is decided from the count matrix (bucket too large
 * for one GPU, who has to grow its receive buffer, how many slices) is decided identically
 * everywhere, and growing buffers — the one local step left, only when a bucket exceeds
 * the capacity — is followed by a second agreement round. After that nothing is allocated.
 * What can still fail is the runtime refusing to enqueue: while the exchange is being
 * enqueued that aborts the transport (ncclCommAbort: this rank's part of the collective
 * can no longer happen); once all of this rank's transfers are enqueued, a later failure
 * (a sort that cannot be launched) is this rank's alone and is only RETURNED — the
 * exchange completes for the peers. An abort is local: it does not wake peers that are
 * already blocked inside a transfer with the aborted rank — and a peer that never enters
 * the collective at all (a crashed process) leaves the others waiting for it. The protocol
 * above guarantees that no LOCAL failure — arguments, memory — ever reaches that state; for
 * the rest the waits can be BOUNDED (round 5): the option `timeout_ms=N` bounds every host
 * wait inside clo_shard_sort_with_device_data (the count exchange, the agreement round after
 * growing buffers), and clo_shard_sort_finish(ss, cq_exec, timeout_ms, err) waits for what
 * the call left running (the key exchange, the local sorts) for at most that long. Both poll
 * the streams and the transport's async_error (RCCL: ncclCommGetAsyncError) instead of
 * blocking; when the time is up or the transport reports a failure they abort this rank's
 * side of the transport and return an error (CLO_ERROR_LIBRARY). The object is then only good
 * for clo_shard_sort_destroy; the process itself stays usable (nothing is restarted).
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
	/* Ends this rank's side of the transport at once (RCCL: ncclCommAbort): its pending operations fail instead
	 * of waiting. May be NULL. After it only destroy is called. */
	void (*abort)(void* user);
	/* Optional (both or neither): device memory for the receive buffers, for transports whose transfers want
	 * memory of their own kind (registered with the communicator, say). NULL from recv_alloc = out of memory:
	 * reported like any allocation failure (the ranks fail together). */
	void* (*recv_alloc)(void* user, size_t bytes);
	void (*recv_free)(void* user, void* ptr);
	/* Optional, never blocks: 0 while the transport is healthy, a clo_hip status once it has failed asynchronously (RCCL:
	 * ncclCommGetAsyncError — a peer that died or aborted, a link error). Polled by every bounded wait of the sort. */
	int (*async_error)(void* user);
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
 * satradix options of the local sort ("radix=16" ...), `slices=S` (1, 2, 4, 8 or auto —
 * the default, see above), `slice_min=<bytes per rank>` (see above), `timeout_ms=N` (bounded
 * waits, see "Failing together"; default 0 = wait for the peers without bound) and `loopback=1`: with a world of ONE rank, run the whole
 * protocol all the same, the rank sending every slice to itself through the transport
 * (a rehearsal of config 5's code path on a one-GPU box; without it one rank takes a
 * shortcut: copy + local sort). The transport stays the caller's. */
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

/* Waits, for at most timeout_ms milliseconds (0: the object's `timeout_ms` option; that being 0 too: without bound), until
 * everything the last clo_shard_sort_with_device_data enqueued — the key exchange on the transfer stream, the local sorts on
 * cq_exec — has completed, polling the streams and the transport's async_error. CL_TRUE: done, the result may be read.
 * CL_FALSE with an error: the time was up (a peer left the exchange hanging) or the transport or a stream reported a
 * failure; this rank's side of the transport has been aborted (see "Failing together"). Not collective. */
cl_bool clo_shard_sort_finish(CloShardSort* ss, CCLQueue* cq_exec, unsigned timeout_ms, GError** err);

/* DEVICE milliseconds between the time stamps the last call left on cq_exec (zeros for one rank without `loopback=1`,
 * which takes the shortcut): [0] partition, [1] count exchange (includes the host's wait for the counts), [2] from the
 * plan to the arrival of the first slice (with one exchange: the whole key exchange), [3] local sort — for a sliced call
 * this includes the stream's waits for the later slices' arrival, so it is NOT pure sorting time. Synchronises on the
 * last stamp. */
void clo_shard_sort_get_phase_ms(CloShardSort* ss, double device_ms[4]);

/* The key exchange of the last call, for rooflines: bytes this rank sent to / received
 * from OTHER ranks and the device milliseconds from the start of the first all-to-all(v)
 * to the end of the last (0 when the world is one rank). Synchronises like get_phase_ms. */
void clo_shard_sort_get_exchange(CloShardSort* ss, size_t* bytes_out, size_t* bytes_in, double* device_ms, int* slices);

/* Pure host logic, exposed for tests: from the G x G matrix counts[src * G + bucket]
 * the four arrays of rank `rank` (elements): send counts / offsets, receive counts / offsets. */
void clo_shard_plan(const uint64_t* counts, int world, int rank,
	size_t* send_counts, size_t* send_offsets, size_t* recv_counts, size_t* recv_offsets);
/* The same for a sliced sort: counts[src * row + dest * subs + k] (row = the stride of a
 * rank's row, >= world * subs; subs sub-buckets per destination, slices divides subs);
 * the four arrays of slice j (`world` entries each, elements; offsets into the partitioned
 * shard and into the receive buffer, where the slices lie one after the other and inside a
 * slice one block per source rank). *slice_offset (may be NULL) = where slice j starts in
 * the receive buffer, *slice_total its size. Returns the size of this rank's whole bucket. */
size_t clo_shard_plan_slice(const uint64_t* counts, size_t row, int world, int subs, int slices, int rank, int j,
	size_t* send_counts, size_t* send_offsets, size_t* recv_counts, size_t* recv_offsets,
	size_t* slice_offset, size_t* slice_total);

#ifdef __cplusplus
}
#endif
#endif
