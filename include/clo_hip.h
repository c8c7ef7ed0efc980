/*
 * clo_hip.h — the thin C-ABI layer over HIP that replaces the cf4ocl2/OpenCL
 * dispatch of cl_ops for the sort/scan hot path.
 *
 * Plain C, plain pointers and sizes; no C++/torch/HIP types in any signature
 * (streams and events travel as void*). The host drivers in
 * cl_ops_amd/csrc (clo_sort_*.c, clo_scan_*.c; plain C, built with gcc) call ONLY this header; the
 * implementations live in the .hip files of cl_ops_amd/csrc/hip (built with hipcc for
 * gfx950). Everything returns 0 on success or a non-zero status that
 * clo_hip_error_string() decodes (positive = hipError_t, negative = CLO_HIP_E*).
 *
 * What each entry point stands in for upstream is cited as file:line relative
 * to the reference tree's src/cl_ops/.
 */
#ifndef CLO_HIP_H
#define CLO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLO_HIP_EARGS      (-1)  /* invalid argument combination */
#define CLO_HIP_EUNSUPPORTED (-2)  /* type/option not built into this library */
#define CLO_HIP_EWORKSPACE (-3)  /* workspace too small */
#define CLO_HIP_ETIMEOUT   (-4)  /* an in-kernel bounded spin gave up (see clo_hip_check_status) */
#define CLO_HIP_ENOTREADY  (-5)  /* clo_hip_stream_query: work enqueued on the stream is still running */
#define CLO_HIP_ERCCL      (-100) /* RCCL failures: CLO_HIP_ERCCL - ncclResult_t */

/* ---- device / runtime (replaces ccl_context_*, ccl_queue_*, ccl_buffer_*,
 *      ccl_event_*, ccl_prof_* as used at sort/clo_sort_abstract.c:335-395,
 *      scan/clo_scan_abstract.c:290-339, benchmarks/clo_sort_bench.c:148-208) ---- */

typedef struct {
	char name[128];
	int compute_units;
	int max_threads_per_block;  /* CL_DEVICE_MAX_WORK_GROUP_SIZE stand-in */
	int wavefront_size;
	size_t lds_bytes_per_block;
	size_t global_mem_bytes;
	char gcn_arch[64];
} clo_hip_device_props;

int clo_hip_device_count(int* count);
int clo_hip_set_device(int device);
int clo_hip_get_device(int* device);
int clo_hip_get_device_props(int device, clo_hip_device_props* props);

int clo_hip_stream_create(void** stream);
/* The same with the device's highest stream priority: for the transfer stream of the sharded sort, whose (few)
 * work-groups should be dispatched ahead of the sort kernels' that fill the device beside them. */
int clo_hip_stream_create_high_priority(void** stream);
int clo_hip_stream_destroy(void* stream);
int clo_hip_stream_synchronize(void* stream);
int clo_hip_stream_query(void* stream);   /* 0 = everything enqueued so far has completed; CLO_HIP_ENOTREADY = not yet; else an error */

int clo_hip_malloc(void** dptr, size_t bytes);
int clo_hip_free(void* dptr);
int clo_hip_memcpy_h2d_async(void* dst, const void* src, size_t bytes, void* stream);
int clo_hip_memcpy_d2h_async(void* dst, const void* src, size_t bytes, void* stream);
int clo_hip_memcpy_d2d_async(void* dst, const void* src, size_t bytes, void* stream);
int clo_hip_memset_async(void* dst, int value, size_t bytes, void* stream);
/* Pin / unpin a caller's host range so that copies to and from it are true
 * asynchronous DMA (the *_with_host_data pipelines, sort/clo_sort_abstract.c:348-395,
 * scan/clo_scan_abstract.c:290-339). Failure is not fatal: pageable copies still work. */
int clo_hip_host_register(void* host_ptr, size_t bytes);
int clo_hip_host_unregister(void* host_ptr);

/* Re-reads the library's environment switches (CLO_MAX_SPINS, CLO_RADIX_SWEEP, CLO_R1_POOLS, CLO_RADIX_NO_DIGITS;
 * INTEGRATION.md lists them all). The C API calls it whenever a sorter / scanner is made; calls in between use what
 * was read then — nothing is read per call. */
void clo_hip_env_refresh(void);
int clo_hip_event_create(void** event);
int clo_hip_event_destroy(void* event);
int clo_hip_event_record(void* event, void* stream);
int clo_hip_event_synchronize(void* event);
int clo_hip_event_query(void* event);   /* 0 = the event has completed; anything else: not yet (or an error) */
int clo_hip_event_elapsed_ms(void* start, void* stop, float* ms);
/* Make `stream` wait for `event` (cq_exec waiting on a cq_comm copy,
 * sort/clo_sort_sbitonic.c:86-95). */
int clo_hip_stream_wait_event(void* stream, void* event);

/* Stream capture into an executable graph: a launch-bound sequence (sbitonic
 * makes one launch per (stage, step): 136 for 2^16 elements) is recorded once
 * and replayed with one call. begin/end bracket the launches on `stream`;
 * end instantiates the graph. */
int clo_hip_stream_is_capturing(void* stream);   /* 1: a capture is active (or the state is unknown), 0: none */
int clo_hip_graph_capture_begin(void* stream);
int clo_hip_graph_capture_end(void* stream, void** graph_exec);
int clo_hip_graph_launch(void* graph_exec, void* stream);
int clo_hip_graph_destroy(void* graph_exec);

const char* clo_hip_error_string(int status);

/* ---- exclusive prefix sum (replaces the three launches of
 *      scan/clo_scan_blelloch.c:146-195: workgroupScan, workgroupSumsScan,
 *      addWorkgroupSums — scan/clo_scan_blelloch.cl:49-211) ----
 * data_out[i] = sum_{j<i} (sum_t) data_in[j], wrap-around in the sum type.
 * elem_size, sum_size in {1,2,4,8} bytes with sum_size >= elem_size (unsigned
 * widening; signed inputs: pass elem_signed=1 for sign extension).
 * workspace: clo_hip_scan_workspace_bytes() bytes of device memory that
 * clo_hip_scan_workspace_init() has zeroed ONCE (after allocating it, and again
 * after a call that ended in CLO_HIP_ETIMEOUT); from then on the scans keep it
 * consistent themselves — look-back entries carry the epoch of the call that
 * wrote them, counters are put back by the last work-group to leave — so a
 * call clears nothing. One scan at a time per workspace. Asynchronous on
 * `stream`. */
size_t clo_hip_scan_workspace_bytes(size_t numel, int elem_size, int sum_size);
int clo_hip_scan_workspace_init(void* workspace, size_t workspace_bytes, void* stream);
/* RULE: every scan on a workspace passes the SAME workspace_bytes the range was
 * initialised with (where the scan keeps its accumulators follows from that
 * size; a workspace sized for the largest array serves every smaller one under
 * its full byte count). The library remembers what clo_hip_scan_workspace_init
 * prepared (pointer and byte count, host side) and answers CLO_HIP_EWORKSPACE
 * to a scan on a range it did not prepare or under another byte count — round 1's
 * "contents don't care, pass clo_hip_scan_workspace_bytes(n) per call" would
 * otherwise return wrong sums silently. clo_hip_scan_workspace_forget: the
 * caller is about to free the range (optional; initialising a range that
 * overlaps a remembered one replaces it). */
int clo_hip_scan_workspace_forget(void* workspace);
/* Test hook: sets the epoch the next scan on this workspace continues from
 * (epochs run 1 .. 2^30-1, then the workspace is zeroed in-kernel and they start
 * over); lets a test cross the wrap without 2^30 calls. */
int clo_hip_scan_workspace_set_epoch(void* workspace, unsigned epoch, void* stream);
int clo_hip_scan_exclusive(const void* data_in, void* data_out, size_t numel,
	int elem_size, int elem_signed, int sum_size,
	void* workspace, size_t workspace_bytes, void* stream);
/* The same scan as one chunk of a longer array (new: the chunked host-data
 * pipeline of clo_scan_with_host_data and the sharded multi-GPU scan):
 * *carry_in_dev (device uint64, low sum_size bytes used; NULL = 0) is added to
 * every output, *carry_out_dev (NULL = not wanted) receives carry_in + the sum
 * of the chunk, i.e. the next chunk's carry_in. */
int clo_hip_scan_exclusive_carry(const void* data_in, void* data_out, size_t numel,
	int elem_size, int elem_signed, int sum_size,
	const uint64_t* carry_in_dev, uint64_t* carry_out_dev,
	void* workspace, size_t workspace_bytes, void* stream);
/* The same scan with a FLOATING-POINT sum type (float: sum_size 4, double: 8);
 * elem_type is a CloType number (any of the eleven), elements are converted to the
 * sum type on load as upstream's kernels do (scan/clo_scan_blelloch.cl:79-80).
 * Reduce / scan the tile sums / apply, every addition in an order fixed by the
 * layout alone: deterministic, equal to upstream's result to rounding (its order
 * is the Blelloch tree of its own work-group size). No look-back, no polling. */
/* Every other pair of CloTypes upstream's generic kernel accepts (scan/clo_scan_abstract.c:122-125 pastes any two
 * type names): a half sum type, floating-point elements summed in an integer type (each element truncated by the
 * cast `(sum type) x`, as upstream's kernel does), integer sums narrower than the elements (the cast keeps the low
 * bits). The same deterministic reduce / scan / apply kernels, arithmetic in the sum type. clo_hip_scan_is_typed
 * says which pairs go here (1) and which to clo_hip_scan_exclusive (0). elem_type / sum_type: CloType numbers. */
int clo_hip_scan_is_typed(int elem_type, int sum_type);
size_t clo_hip_scan_typed_workspace_bytes(size_t numel, int sum_type);
int clo_hip_scan_exclusive_typed(const void* data_in, void* data_out, size_t numel, int elem_type, int sum_type,
	void* workspace, size_t workspace_bytes, void* stream);
size_t clo_hip_scan_fp_workspace_bytes(size_t numel, int sum_size);
int clo_hip_scan_exclusive_fp(const void* data_in, void* data_out, size_t numel, int elem_type, int sum_size,
	void* workspace, size_t workspace_bytes, void* stream);
/* Sum of the elements mod 2^64 into *total_dev (device): what a shard hands
 * to the later shards of a multi-GPU scan. */
int clo_hip_reduce_sum(const void* data_in, size_t numel, int elem_size, int elem_signed,
	uint64_t* total_dev, void* stream);

/* ---- LSD radix sort (replaces the per-digit loop of
 *      sort/clo_sort_satradix.c:264-313: satradix_localsort, satradix_histogram,
 *      clo_scan_with_device_data, satradix_scatter — sort/clo_sort_satradix.cl:34-258) ----
 * Stable ascending sort of `numel` elements of elem_size bytes by the key
 * field  key = (elem >> key_shift) & ((1<<key_bits)-1), digits of
 * `digit_bits` bits (1..8; radix = 1<<digit_bits), least significant first.
 * key_kind: 0 = unsigned (raw bit order, what upstream's kernels do for every
 * type), 1 = two's complement, 2 = IEEE-754 (key_bits 16/32/64; -0 < +0, NaNs
 * at the ends by sign) — kinds 1 and 2 go through an order-preserving
 * transform on the first read and back on the last write, so negative keys
 * land where upstream's own check (benchmarks/clo_bench.c:26-65) wants them.
 * src is read, the sorted result is written to dst; tmp is scratch of the same
 * size. dst may equal src (in place); tmp must be distinct from both. src is
 * left untouched when dst != src. Asynchronous on `stream`. */
/* Loads the code objects of the radix passes that a small first sort does not reach (HIP loads one at the first launch
 * that needs it; upstream builds and loads every kernel in clo_sort_new, sort/clo_sort_abstract.c:144-179). */
int clo_hip_radix_preload(void);
size_t clo_hip_radix_workspace_bytes(size_t numel, int elem_size, int key_bits, int digit_bits);
/* 1 if the kernels of such a sort wait for other work-groups (the single-sweep
 * passes hand digit counts from tile to tile; every spin is bounded and a give-up
 * raises the workspace's status word: clo_hip_check_status), 0 if none does (the
 * chain-free passes). */
int clo_hip_radix_polls(size_t numel, int elem_size, int digit_bits);
int clo_hip_radix_sort(const void* src, void* dst, void* tmp, size_t numel,
	int elem_size, int key_shift, int key_bits, int key_kind, int digit_bits,
	void* workspace, size_t workspace_bytes, void* stream);
/* The same sort FED by whoever produced the keys: first_digits[i] = the low 8 bits of element i's key field,
 * (elem[i] >> key_shift) & 0xff (device memory, numel bytes, read before anything is written — it may live in
 * `tmp`). The one re-read of the keys the sort has left is its first histogram; a producer that touches every
 * key anyway (a key extractor, a generator, the pass before in a pipeline) writes those bytes for nearly nothing
 * and the histogram reads numel bytes instead of numel elements. clo_hip_radix_takes_first_digits says whether a
 * sort of this shape reads them (1: the chain-free passes on 16 384-element tiles, radix 16 or 256, unsigned
 * keys) — otherwise they are ignored, NULL is always allowed. */
int clo_hip_radix_takes_first_digits(size_t numel, int elem_size, int key_kind, int digit_bits);
int clo_hip_radix_sort_fed(const void* src, void* dst, void* tmp, size_t numel,
	int elem_size, int key_shift, int key_bits, int key_kind, int digit_bits, const unsigned char* first_digits,
	void* workspace, size_t workspace_bytes, void* stream);
/* Diagnostic: a device buffer of 8 x tiles uint64 that the LAST single-sweep pass of a sort fills with per-tile
 * s_memtime stamps (tile drawn ... stored) until it is taken away again with NULL. Process-wide, off by default. */
int clo_hip_radix_debug_stamps(void* buffer, size_t tiles);

/* Segmented sort (new functionality, the local step of the sharded sort): `nseg` (1..256) segments of ONE array —
 * seg_counts[i] elements each, host memory, summing to numel — are sorted independently, each stably by the key
 * field [key_shift, key_shift + key_bits), in SHARED launches: a launch's tiles and counter-scan chunks are
 * numbered through all segments and one small table lookup tells a work-group which segment it works for.
 * Sub-buckets that share their top key bits (what an MSD partition leaves) are thereby sorted on the REMAINING
 * bits at the rate of one large sort, where sorting them one by one would be launch-bound.
 * Where the segments lie: the RESULT holds them back to back in order (segment k at the sum of the counts before
 * it). The SOURCE `src` either holds them the same way (npieces = 0) or in up to 256 PIECES anywhere in it
 * (offsets below 2^32 elements): piece i = piece_counts[i] elements from piece_offsets[i], belonging to segment
 * piece_segment[i], pieces listed in segment order (what a rank of the sharded sort receives: one piece per
 * source rank and sub-bucket); the first pass then gathers a segment's pieces, in the order listed, as a
 * by-product. The first pass reads `src` and writes `b` (numel elements), the later ones go b -> a -> b ...: the
 * result is in `b` when *result_in_b comes back 1 (an odd number of 8-bit passes, ceil(key_bits / 8)), else in
 * `a`. `src` may be `a` itself (the first pass is the only one to read it); otherwise it is left untouched.
 * digit_bits 4 or 8 (radix 16 / 256), elem_size 4 or 8, unsigned keys. Asynchronous; the host arrays are read
 * before the call returns. */
size_t clo_hip_radix_seg_workspace_bytes(size_t numel, int nseg, int elem_size, int digit_bits);
int clo_hip_radix_sort_segmented(const void* src, void* a, void* b, size_t numel, const size_t* seg_counts, int nseg,
	const size_t* piece_counts, const size_t* piece_offsets, const int* piece_segment, int npieces,
	int elem_size, int key_shift, int key_bits, int digit_bits, void* workspace, size_t workspace_bytes, void* stream,
	int* result_in_b);
/* The same with the pieces in TWO sources: piece i lies in `src2` when piece_source[i] is 1 (piece_source NULL: all in
 * `src`), its offset counted from that source's start. What it is for: the piece of a sub-bucket that a rank of the
 * sharded sort keeps for itself never needs to move — it is gathered straight out of the partitioned shard, and the
 * rank's own share of every exchange (1 / G of the bytes; all of them on one rank) is not copied at all. `src2` is
 * only read, by the first pass; it must not be `b`. */
int clo_hip_radix_sort_segmented2(const void* src, const void* src2, void* a, void* b, size_t numel, const size_t* seg_counts, int nseg,
	const size_t* piece_counts, const size_t* piece_offsets, const int* piece_segment, const int* piece_source, int npieces,
	int elem_size, int key_shift, int key_bits, int digit_bits, void* workspace, size_t workspace_bytes, void* stream,
	int* result_in_b);

/* MSD bucket partition used by the multi-GPU exchange (new
 * functionality): stable split of src into 1<<bucket_bits buckets by the top
 * bucket_bits of the key field; dst gets the buckets back to back in bucket
 * order. counts_dev (device, 1<<bucket_bits uint64; may be NULL for the
 * partition) receives the bucket sizes — the partition produces them as a
 * by-product, clo_hip_msd_histogram computes them alone. bucket_bits: 1..8 for
 * the partition (the sharded sort splits on 8: ranks x sub-buckets per rank = 256,
 * include/clo_shard.h), 1..3 for the histogram alone. */
int clo_hip_msd_histogram(const void* src, size_t numel, int elem_size,
	int key_shift, int key_bits, int bucket_bits,
	uint64_t* counts_dev, void* stream);
int clo_hip_msd_partition(const void* src, void* dst, size_t numel, int elem_size,
	int key_shift, int key_bits, int bucket_bits, uint64_t* counts_dev,
	void* workspace, size_t workspace_bytes, void* stream);
size_t clo_hip_msd_workspace_bytes(size_t numel, int elem_size, int bucket_bits);

/* ---- RCCL over xGMI: the two collectives of the sharded sort (include/clo_shard.h;
 *      new functionality, the reference is single-device). One communicator per
 *      process = per GPU; the 128-byte id is made by one rank and handed to the
 *      others by whatever side channel the launcher has. Both calls are ordered on
 *      `stream`. all_to_all_v: rank r's bytes [send_offset[p], +send_bytes[p]) go to
 *      peer p and land at its [recv_offset[r], +recv_bytes[r]); ONE group of
 *      ncclSend / ncclRecv pairs, every pair on its own xGMI link. ---- */
#define CLO_HIP_RCCL_ID_BYTES 128
int clo_hip_rccl_unique_id(void* id_out);
int clo_hip_rccl_comm_create(void** comm, const void* id_in, int rank, int world);
int clo_hip_rccl_comm_destroy(void* comm);
/* ncclCommAbort: ends the communicator NOW; operations of it that are pending on any
 * rank's streams fail instead of waiting for this rank for ever. What a rank that
 * cannot take part in a collective its peers have already entered calls before it
 * returns its error. The handle is gone afterwards (no clo_hip_rccl_comm_destroy). */
int clo_hip_rccl_comm_abort(void* comm);
/* ncclCommGetAsyncError: 0 while the communicator is healthy, the RCCL status of an asynchronous failure (a peer that
 * died, a network error) once there is one. Never blocks. */
int clo_hip_rccl_comm_async_error(void* comm);
int clo_hip_rccl_all_gather_u64(void* comm, const uint64_t* send_dev, uint64_t* recv_dev, size_t count, void* stream);
int clo_hip_rccl_all_to_all_v(void* comm, int rank, int world,
	const void* send_dev, const size_t* send_bytes, const size_t* send_offset_bytes,
	void* recv_dev, const size_t* recv_bytes, const size_t* recv_offset_bytes, void* stream);

/* ---- bitonic sorts (replace the launch loops of
 *      sort/clo_sort_sbitonic.c:102-118 / sort/clo_sort_sbitonic.cl:38-69 and
 *      sort/clo_sort_abitonic.c:401-432 / sort/clo_sort_abitonic.cl:234-1067) ----
 * In-place bitonic network over nlpo2(numel) slots (numel need not be a power
 * of two: the caller provides a buffer of clo_hip_bitonic_padded_numel(numel)
 * elements and the call pads the tail so that it sorts to the end, i.e.
 * data[0..numel) comes out sorted). The key is the low key_bits bits of
 * (KEY_TYPE)(elem >> key_shift), KEY_TYPE being key_size bytes wide;
 * key_kind: 0 unsigned, 1 signed, 2 float (then key_bits = 8*key_size).
 * descending: CLO_SORT_COMPARE "((a) < (b))".
 * clo_hip_bitonic_simple: one launch per (stage, step) — the sbitonic schedule.
 * clo_hip_bitonic_tiled: LDS/register-tiled schedule — the abitonic replacement.
 * *launches (may be NULL) receives the number of kernel launches made. */
size_t clo_hip_bitonic_padded_numel(size_t numel);
int clo_hip_bitonic_simple(void* data, size_t numel, int elem_size,
	int key_shift, int key_bits, int key_size, int key_kind, int descending,
	int* launches, void* stream);
/* In place, ANY numel and any key: the flip form of the network (every comparator ascending, comparators that reach
 * past numel skipped), one launch per step. What the drivers use for a numel that is not a power of two when the key
 * is only part of the element (for whole-element keys they pad and run the fast schedules). */
int clo_hip_bitonic_any(void* data, size_t numel, int elem_size,
	int key_shift, int key_bits, int key_size, int key_kind, int descending, int* launches, void* stream);
int clo_hip_bitonic_tiled(void* data, size_t numel, int elem_size,
	int key_shift, int key_bits, int key_size, int key_kind, int descending,
	int* launches, void* stream);

/* ---- gselect (replaces the launch of sort/clo_sort_gselect.c:60-118 /
 *      sort/clo_sort_gselect.cl:38-58): O(n^2) rank sort, stable, out of place
 *      (dst != src); key description as for the bitonic sorts. ---- */
int clo_hip_gselect(const void* src, void* dst, size_t numel, int elem_size,
	int key_shift, int key_bits, int key_size, int key_kind, int descending, void* stream);

/* ---- bitonic sorts specialised at run time (hiprtc) for arbitrary compare /
 *      get_key expressions — what upstream does for every sorter by OpenCL JIT
 *      (sort/clo_sort_abstract.c:144-179). elem_type / key_type: CloType numbers.
 *      compare: body of CLO_SORT_COMPARE(a, b) (NULL: "((a) > (b))"); get_key:
 *      body of CLO_SORT_KEY_GET(x) (NULL: "(x)"). compiler_opts (may be NULL): the
 *      caller's options for the compiler, white-space separated, as upstream hands
 *      them to the OpenCL JIT (sort/clo_sort_abstract.c:177-178) — -DNAME[=value] /
 *      -UNAME / -Idir reach hiprtc (also spelled "-D NAME"), OpenCL's own -cl-...
 *      switches are dropped, anything else is the compiler's to accept or refuse. On a
 *      compile error returns CLO_HIP_EARGS and, if log != NULL, a malloc'd build log
 *      (caller frees). ----
 */
int clo_hip_bitonic_jit_create(int elem_type, int key_type, const char* compare, const char* get_key, const char* compiler_opts,
	void** handle, char** log);
void clo_hip_bitonic_jit_destroy(void* handle);
/* In place, numel a power of two. tiled: 0 = sbitonic schedule, 1 = abitonic. */
/* gselect (upstream's O(n^2) rank sort, sort/clo_sort_gselect.cl:38-58) through the same compiled module:
 * src -> dst (distinct), ties by index. */
int clo_hip_bitonic_jit_gselect(void* handle, const void* src, void* dst, size_t numel, void* stream);
int clo_hip_bitonic_jit_sort(void* handle, void* data, size_t numel, int tiled, int* launches, void* stream);
/* Static LDS bytes per work-group of the kernels that call would launch (0: one launch per step, registers only). */
size_t clo_hip_bitonic_jit_lds_bytes(void* handle, size_t numel, int tiled);

/* ---- satradix specialised at run time (hiprtc) for a get_key expression
 *      outside the ahead-of-time family: the key is materialised as
 *      (ordered key bits << 32 | index) pairs by a compiled kernel, the pairs are
 *      radix-sorted, the elements gathered. 8-byte keys take two such rounds
 *      (low half of the key, then the high half in the first round's order).
 *      pairs / pairs_tmp: numel * 8 bytes each; workspace as for
 *      clo_hip_radix_workspace_bytes(numel, 8, 32, digit_bits). ---- */
int clo_hip_radix_jit_create(int elem_type, int key_type, const char* get_key, const char* compiler_opts, void** handle, char** log);
void clo_hip_radix_jit_destroy(void* handle);
int clo_hip_radix_jit_sort(void* handle, const void* src, void* dst, void* pairs, void* pairs_tmp, size_t numel,
	int digit_bits, void* workspace, size_t workspace_bytes, void* stream);

/* ---- status word of the bounded spins ----
 * The scan is the one kernel that polls other work-groups' state (decoupled
 * look-back); it bounds every spin (CLO_MAX_SPINS polls; the environment
 * variable of that name overrides the bound) and on give-up sets a word in its
 * workspace and finishes with wrong output. This reads the word back
 * (synchronises `stream`). Returns 0, CLO_HIP_ETIMEOUT or a hip error; after
 * CLO_HIP_ETIMEOUT the workspace must be initialised again. The host drivers
 * check it wherever they synchronise anyway (clo_scan_with_host_data,
 * ccl_queue_finish, ccl_event_wait). Meaningful for the workspaces of the scan
 * and of clo_hip_msd_partition (always 0 there: no kernel of the sorts waits on
 * another work-group) and of clo_hip_radix_sort when clo_hip_radix_polls() says
 * its kernels poll — the caller clears the first 512 bytes of that workspace once
 * after allocating it, the sorts never clear the word themselves. */
int clo_hip_check_status(void* workspace, void* stream);

/* ---- launch observer: per-kernel events for profiling queues ----
 * Upstream names every kernel launch it enqueues (ccl_event_set_name at
 * sort/clo_sort_satradix.c:282,295,312, scan/clo_scan_blelloch.c:158,183,193) and
 * CCLProf reports per name. The C-ABI calls below launch several kernels each;
 * while an observer is installed (per calling thread), it is told about every
 * launch: phase 0 right before (label = the kernel family: "radix_hist",
 * "radix_offsets", "radix_pass", "radix_small", "scan", "bitonic_step", ...),
 * phase 1 right after (label NULL), with the stream the launch went to. The
 * host drivers use it to record one CCLEvent per kernel when the queue was
 * created with CL_QUEUE_PROFILING_ENABLE. NULL removes it. */
typedef void (*clo_hip_launch_observer)(void* user, const char* label, int phase, void* stream);
int clo_hip_set_launch_observer(clo_hip_launch_observer fn, void* user);

/* ---- per-kernel device timing (measurement only; bench.py's roofline leg) ----
 * While enabled, every kernel launch made by this library is bracketed by a
 * pair of HIP events recorded on the stream the kernel runs on.
 * clo_hip_timing_read sums the elapsed time of the launches recorded under
 * `label` since the last reset ("radix_pass", "radix_hist", "radix_offsets",
 * "radix_small", "msd_partition", "scan", "reduce", "bitonic_presort",
 * "bitonic_tile", "bitonic_strided", "bitonic_step", "gselect"). */
int clo_hip_timing_enable(int on);
int clo_hip_timing_enabled(void);
int clo_hip_timing_reset(void);
int clo_hip_timing_read(const char* label, unsigned* count, float* total_ms);

/* Static LDS bytes per work-group of the kernel families, for the
 * get_localmem_usage introspection calls (sort/clo_sort_satradix.c:626-658,
 * scan/clo_scan_blelloch.c:307-319). family: "radix_hist", "radix_pass",
 * "scan", "bitonic_tile", "bitonic_strided", "bitonic_step". */
size_t clo_hip_kernel_lds_bytes(const char* family, int elem_size, int param);
/* Static LDS bytes per work-group of the kernels the bitonic schedules launch for `numel` elements: tiled = 1 the tiled
 * schedule (clo_hip_bitonic_tiled: 0 below 32 elements, the run-time-schedule tile kernel up to one tile, the
 * compile-time-schedule kernels on 2^14-element tiles — 2^13 of 8 bytes — above), tiled = 0 one launch per step (0). */
size_t clo_hip_bitonic_lds_bytes(size_t numel, int elem_size, int tiled);

#ifdef __cplusplus
}
#endif
#endif
