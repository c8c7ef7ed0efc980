/*
 * clo_hip_stub.c — TEST INFRASTRUCTURE, never part of the product: a host-memory implementation of the thin C-ABI
 * (include/clo_hip.h) so that the C host drivers above it (cl_ops_amd/csrc/*.c: the threaded clo_*_with_host_data
 * pipelines, the sharded sort's plan / slices / pieces logic, the queue / event layer) run on the CPU under
 * AddressSanitizer, UBSan and ThreadSanitizer (tests/test_host_sanitizers.py). "Device memory" is malloc'd host
 * memory, streams execute every command at once when it is enqueued, events are time stamps; the compute entry
 * points are plain serial C with the SAME contract as the HIP ones (stable radix sort by a key field, segmented sort
 * with pieces, MSD partition with bucket counts, exclusive scan with carry) and check their arguments the way the HIP
 * layer does, so that a wrong offset or a short buffer in a driver shows up as a sanitizer report or a wrong
 * result. Nothing here is linked into libcl_ops_hip.so, and nothing of it runs on the GPU box.
 */
#define _GNU_SOURCE
#include "clo_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- device / runtime ---- */
static __thread int t_device;
int clo_hip_device_count(int* count) { if (count) *count = 1; return 0; }
int clo_hip_set_device(int device) { t_device = device; return device == 0 ? 0 : CLO_HIP_EARGS; }
int clo_hip_get_device(int* device) { if (device) *device = t_device; return 0; }
int clo_hip_get_device_props(int device, clo_hip_device_props* props) {
	if (!props || device != 0) return CLO_HIP_EARGS;
	memset(props, 0, sizeof(*props));
	strcpy(props->name, "host stub");
	strcpy(props->gcn_arch, "none");
	props->compute_units = 1; props->max_threads_per_block = 1024; props->wavefront_size = 64;
	props->lds_bytes_per_block = 65536; props->global_mem_bytes = (size_t) 1 << 34;
	return 0;
}
int clo_hip_stream_create(void** stream) { *stream = malloc(8); return *stream ? 0 : 2; }
int clo_hip_stream_create_high_priority(void** stream) { return clo_hip_stream_create(stream); }
int clo_hip_stream_destroy(void* stream) { free(stream); return 0; }
/* A stream is idle at once — unless the test says otherwise: its hook (tests/hoststub/host_paths_test.c: a transport whose
 * collective is still waiting for a rank) makes clo_hip_stream_query / _synchronize see pending work. */
int (*clo_hip_stub_stream_hook)(void* stream);
int clo_hip_stream_query(void* stream) { return clo_hip_stub_stream_hook ? clo_hip_stub_stream_hook(stream) : 0; }
int clo_hip_stream_synchronize(void* stream) {
	while (clo_hip_stub_stream_hook && clo_hip_stub_stream_hook(stream) == CLO_HIP_ENOTREADY) { struct timespec ts = { 0, 100000 }; nanosleep(&ts, NULL); }
	return 0;
}
int clo_hip_malloc(void** dptr, size_t bytes) { *dptr = malloc(bytes ? bytes : 1); return *dptr ? 0 : 2; }
int clo_hip_free(void* dptr) { free(dptr); return 0; }
int clo_hip_memcpy_h2d_async(void* dst, const void* src, size_t bytes, void* stream) { (void) stream; memcpy(dst, src, bytes); return 0; }
int clo_hip_memcpy_d2h_async(void* dst, const void* src, size_t bytes, void* stream) { (void) stream; memcpy(dst, src, bytes); return 0; }
int clo_hip_memcpy_d2d_async(void* dst, const void* src, size_t bytes, void* stream) { (void) stream; memmove(dst, src, bytes); return 0; }
int clo_hip_memset_async(void* dst, int value, size_t bytes, void* stream) { (void) stream; memset(dst, value, bytes); return 0; }
int clo_hip_host_register(void* host_ptr, size_t bytes) { (void) host_ptr; (void) bytes; return 0; }
int clo_hip_host_unregister(void* host_ptr) { (void) host_ptr; return 0; }
void clo_hip_env_refresh(void) {}

typedef struct { double t; } stub_event;
static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
int clo_hip_event_create(void** event) { *event = calloc(1, sizeof(stub_event)); return *event ? 0 : 2; }
int clo_hip_event_destroy(void* event) { free(event); return 0; }
int clo_hip_event_record(void* event, void* stream) { (void) stream; if (!event) return CLO_HIP_EARGS; ((stub_event*) event)->t = now_ms(); return 0; }
int clo_hip_event_synchronize(void* event) { return event ? 0 : CLO_HIP_EARGS; }
int clo_hip_event_query(void* event) { return event ? 0 : CLO_HIP_EARGS; }
int clo_hip_event_elapsed_ms(void* start, void* stop, float* ms) {
	if (!start || !stop || !ms) return CLO_HIP_EARGS;
	*ms = (float) (((stub_event*) stop)->t - ((stub_event*) start)->t);
	return 0;
}
int clo_hip_stream_wait_event(void* stream, void* event) { (void) stream; return event ? 0 : CLO_HIP_EARGS; }
int clo_hip_stream_is_capturing(void* stream) { (void) stream; return 0; }
int clo_hip_graph_capture_begin(void* stream) { (void) stream; return CLO_HIP_EUNSUPPORTED; }
int clo_hip_graph_capture_end(void* stream, void** graph_exec) { (void) stream; if (graph_exec) *graph_exec = NULL; return CLO_HIP_EUNSUPPORTED; }
int clo_hip_graph_launch(void* graph_exec, void* stream) { (void) graph_exec; (void) stream; return CLO_HIP_EUNSUPPORTED; }
int clo_hip_graph_destroy(void* graph_exec) { (void) graph_exec; return 0; }
const char* clo_hip_error_string(int status) {
	switch (status) {
		case 0: return "success";
		case CLO_HIP_EARGS: return "invalid arguments";
		case CLO_HIP_EUNSUPPORTED: return "not supported (host stub)";
		case CLO_HIP_EWORKSPACE: return "workspace too small";
		case CLO_HIP_ETIMEOUT: return "a bounded spin gave up";
		case CLO_HIP_ENOTREADY: return "not ready";
		default: return "stub error";
	}
}
int clo_hip_check_status(void* workspace, void* stream) { (void) stream; return workspace && *(unsigned*) workspace ? CLO_HIP_ETIMEOUT : 0; }
int clo_hip_set_launch_observer(clo_hip_launch_observer fn, void* user) { (void) fn; (void) user; return 0; }
int clo_hip_timing_enable(int on) { (void) on; return 0; }
int clo_hip_timing_enabled(void) { return 0; }
int clo_hip_timing_reset(void) { return 0; }
int clo_hip_timing_read(const char* label, unsigned* count, float* total_ms) { (void) label; if (count) *count = 0; if (total_ms) *total_ms = 0; return 0; }
size_t clo_hip_kernel_lds_bytes(const char* family, int elem_size, int param) { (void) family; (void) elem_size; (void) param; return 0; }
size_t clo_hip_bitonic_lds_bytes(size_t numel, int elem_size, int tiled) { (void) numel; (void) elem_size; (void) tiled; return 0; }

/* ---- keys ---- */
static uint64_t load_elem(const void* p, int es) {
	switch (es) {
		case 1: return *(const uint8_t*) p;
		case 2: { uint16_t v; memcpy(&v, p, 2); return v; }
		case 4: { uint32_t v; memcpy(&v, p, 4); return v; }
		default: { uint64_t v; memcpy(&v, p, 8); return v; }
	}
}
/* the ordered unsigned image of the key field [shift, shift + bits): kind 0 unsigned, 1 two's complement, 2 IEEE */
static uint64_t ordered_key(uint64_t e, int shift, int bits, int kind) {
	const uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
	uint64_t k = (e >> shift) & mask;
	const uint64_t sign = 1ull << (bits - 1);
	if (kind == 1) k ^= sign;
	else if (kind == 2) k = (k & sign) ? (~k & mask) : (k | sign);
	return k;
}

/* stable LSD byte-wise counting sort of n elements by the ordered key; src -> dst, tmp scratch (may be NULL: allocated) */
static int stable_sort(const void* src, void* dst, size_t n, int es, int shift, int bits, int kind, int descending) {
	if (n == 0) return 0;
	char* a = (char*) malloc(n * (size_t) es);
	char* b = (char*) malloc(n * (size_t) es);
	uint64_t* ka = (uint64_t*) malloc(n * sizeof(uint64_t));
	uint64_t* kb = (uint64_t*) malloc(n * sizeof(uint64_t));
	if (!a || !b || !ka || !kb) { free(a); free(b); free(ka); free(kb); return 2; }
	memcpy(a, src, n * (size_t) es);
	const uint64_t flip = descending ? (bits >= 64 ? ~0ull : ((1ull << bits) - 1ull)) : 0ull;
	for (size_t i = 0; i < n; ++i) ka[i] = ordered_key(load_elem(a + i * (size_t) es, es), shift, bits, kind) ^ flip;
	for (int d = 0; d * 8 < bits; ++d) {
		size_t cnt[257];
		memset(cnt, 0, sizeof(cnt));
		for (size_t i = 0; i < n; ++i) cnt[((ka[i] >> (8 * d)) & 255u) + 1]++;
		for (int k = 0; k < 256; ++k) cnt[k + 1] += cnt[k];
		for (size_t i = 0; i < n; ++i) {
			const size_t to = cnt[(ka[i] >> (8 * d)) & 255u]++;
			kb[to] = ka[i];
			memcpy(b + to * (size_t) es, a + i * (size_t) es, (size_t) es);
		}
		char* t = a; a = b; b = t;
		uint64_t* kt = ka; ka = kb; kb = kt;
	}
	memcpy(dst, a, n * (size_t) es);
	free(a); free(b); free(ka); free(kb);
	return 0;
}

/* ---- radix sort ---- */
int clo_hip_radix_preload(void) { return 0; }
size_t clo_hip_radix_workspace_bytes(size_t numel, int elem_size, int key_bits, int digit_bits) {
	(void) numel; (void) elem_size;
	return (digit_bits < 1 || digit_bits > 8 || key_bits < 1) ? 0 : 1024;
}
int clo_hip_radix_polls(size_t numel, int elem_size, int digit_bits) { (void) numel; (void) elem_size; (void) digit_bits; return 0; }
int clo_hip_radix_takes_first_digits(size_t numel, int elem_size, int key_kind, int digit_bits) { (void) numel; (void) elem_size; (void) key_kind; (void) digit_bits; return 0; }
int clo_hip_radix_sort(const void* src, void* dst, void* tmp, size_t numel, int elem_size, int key_shift, int key_bits, int key_kind,
	int digit_bits, void* workspace, size_t workspace_bytes, void* stream) {
	(void) stream;
	if (numel == 0) return 0;
	if (!src || !dst || !tmp || !workspace || tmp == src || tmp == dst) return CLO_HIP_EARGS;
	if (key_bits < 1 || key_shift < 0 || key_shift + key_bits > 8 * elem_size || key_kind < 0 || key_kind > 2) return CLO_HIP_EARGS;
	if (digit_bits < 1 || digit_bits > 8) return CLO_HIP_EUNSUPPORTED;
	if (workspace_bytes < clo_hip_radix_workspace_bytes(numel, elem_size, key_bits, digit_bits)) return CLO_HIP_EWORKSPACE;
	memset(tmp, 0xA5, numel * (size_t) elem_size);   /* the scratch really is scratch: whoever expected data there finds out */
	memset(workspace, 0, 512);
	return stable_sort(src, dst, numel, elem_size, key_shift, key_bits, key_kind, 0);
}
int clo_hip_radix_sort_fed(const void* src, void* dst, void* tmp, size_t numel, int elem_size, int key_shift, int key_bits, int key_kind,
	int digit_bits, const unsigned char* first_digits, void* workspace, size_t workspace_bytes, void* stream) {
	(void) first_digits;
	return clo_hip_radix_sort(src, dst, tmp, numel, elem_size, key_shift, key_bits, key_kind, digit_bits, workspace, workspace_bytes, stream);
}
size_t clo_hip_radix_seg_workspace_bytes(size_t numel, int nseg, int elem_size, int digit_bits) {
	(void) numel;
	if ((digit_bits != 4 && digit_bits != 8) || (elem_size != 4 && elem_size != 8) || nseg < 1 || nseg > 256) return 0;
	return 2048;
}
int clo_hip_radix_sort_segmented(const void* src, void* a, void* b, size_t numel, const size_t* seg_counts, int nseg,
	const size_t* piece_counts, const size_t* piece_offsets, const int* piece_segment, int npieces,
	int elem_size, int key_shift, int key_bits, int digit_bits, void* workspace, size_t workspace_bytes, void* stream, int* result_in_b) {
	return clo_hip_radix_sort_segmented2(src, NULL, a, b, numel, seg_counts, nseg, piece_counts, piece_offsets, piece_segment, NULL, npieces,
		elem_size, key_shift, key_bits, digit_bits, workspace, workspace_bytes, stream, result_in_b);
}
int clo_hip_radix_sort_segmented2(const void* src, const void* src2, void* a, void* b, size_t numel, const size_t* seg_counts, int nseg,
	const size_t* piece_counts, const size_t* piece_offsets, const int* piece_segment, const int* piece_source, int npieces,
	int elem_size, int key_shift, int key_bits, int digit_bits, void* workspace, size_t workspace_bytes, void* stream, int* result_in_b) {
	(void) stream;
	if (!result_in_b) return CLO_HIP_EARGS;
	*result_in_b = 0;
	if (numel == 0) return 0;
	if (!src || !a || !b || a == b || src == b || !workspace || !seg_counts || nseg < 1 || nseg > 256) return CLO_HIP_EARGS;
	if (npieces < 0 || npieces > 256 || (npieces > 0 && (!piece_counts || !piece_offsets || !piece_segment))) return CLO_HIP_EARGS;
	if (elem_size != 4 && elem_size != 8) return CLO_HIP_EUNSUPPORTED;
	if (digit_bits != 4 && digit_bits != 8) return CLO_HIP_EUNSUPPORTED;
	if (key_bits < 1 || key_shift < 0 || key_shift + key_bits > 8 * elem_size) return CLO_HIP_EARGS;
	if (workspace_bytes < clo_hip_radix_seg_workspace_bytes(numel, nseg, elem_size, digit_bits)) return CLO_HIP_EWORKSPACE;
	size_t total = 0;
	for (int k = 0; k < nseg; ++k) total += seg_counts[k];
	if (total != numel) return CLO_HIP_EARGS;
	const size_t es = (size_t) elem_size;
	char* gathered = (char*) malloc(numel * es);
	if (!gathered) return 2;
	if (npieces > 0) {   /* segment k = its pieces in the order listed; pieces come in segment order */
		size_t per[256], at = 0;
		int prev = 0;
		memset(per, 0, sizeof(per));
		for (int i = 0; i < npieces; ++i) {
			if (piece_segment[i] < prev || piece_segment[i] >= nseg) { free(gathered); return CLO_HIP_EARGS; }
			prev = piece_segment[i];
			per[piece_segment[i]] += piece_counts[i];
			const int from2 = piece_source && piece_source[i];
			if (from2 && (piece_source[i] != 1 || !src2 || src2 == (const void*) b)) { free(gathered); return CLO_HIP_EARGS; }
			memcpy(gathered + at * es, (const char*) (from2 ? src2 : src) + piece_offsets[i] * es, piece_counts[i] * es);   /* (a piece outside its source: ASan says so) */
			at += piece_counts[i];
		}
		for (int k = 0; k < nseg; ++k) if (per[k] != seg_counts[k]) { free(gathered); return CLO_HIP_EARGS; }
	} else {
		memcpy(gathered, src, numel * es);
	}
	const int passes = (key_bits + 7) / 8;
	*result_in_b = passes % 2;
	char* out = (char*) (*result_in_b ? b : a);
	char* other = (char*) (*result_in_b ? a : b);
	size_t at = 0;
	int st = 0;
	for (int k = 0; k < nseg && st == 0; ++k) {
		st = stable_sort(gathered + at * es, out + at * es, seg_counts[k], elem_size, key_shift, key_bits, 0, 0);
		at += seg_counts[k];
	}
	/* the buffer that does not hold the result is scratch (when there is more than one pass, or when it is the source) */
	if (passes > 1 || (const void*) other == src) memset(other, 0x5A, numel * es);
	free(gathered);
	return st;
}

/* ---- MSD partition ---- */
size_t clo_hip_msd_workspace_bytes(size_t numel, int elem_size, int bucket_bits) {
	(void) numel; (void) elem_size;
	return (bucket_bits < 1 || bucket_bits > 8) ? 0 : 1024;
}
int clo_hip_msd_histogram(const void* src, size_t numel, int elem_size, int key_shift, int key_bits, int bucket_bits, uint64_t* counts_dev, void* stream) {
	(void) stream;
	if (!counts_dev || bucket_bits < 1 || bucket_bits > 3 || bucket_bits > key_bits) return CLO_HIP_EARGS;
	memset(counts_dev, 0, sizeof(uint64_t) << bucket_bits);
	for (size_t i = 0; i < numel; ++i)
		counts_dev[(load_elem((const char*) src + i * (size_t) elem_size, elem_size) >> (key_shift + key_bits - bucket_bits)) & ((1u << bucket_bits) - 1u)]++;
	return 0;
}
int clo_hip_msd_partition(const void* src, void* dst, size_t numel, int elem_size, int key_shift, int key_bits, int bucket_bits,
	uint64_t* counts_dev, void* workspace, size_t workspace_bytes, void* stream) {
	(void) stream;
	if (bucket_bits < 1 || bucket_bits > 8 || bucket_bits > key_bits) return CLO_HIP_EARGS;
	if (numel == 0) { if (counts_dev) memset(counts_dev, 0, sizeof(uint64_t) << bucket_bits); return 0; }
	if (!src || !dst || src == dst || !workspace) return CLO_HIP_EARGS;
	if (elem_size != 4 && elem_size != 8) return CLO_HIP_EUNSUPPORTED;
	if (workspace_bytes < clo_hip_msd_workspace_bytes(numel, elem_size, bucket_bits)) return CLO_HIP_EWORKSPACE;
	const int shift = key_shift + key_bits - bucket_bits;
	const int st = stable_sort(src, dst, numel, elem_size, shift, bucket_bits, 0, 0);
	if (st == 0 && counts_dev) {
		memset(counts_dev, 0, sizeof(uint64_t) << bucket_bits);
		for (size_t i = 0; i < numel; ++i) counts_dev[(load_elem((const char*) src + i * (size_t) elem_size, elem_size) >> shift) & ((1u << bucket_bits) - 1u)]++;
	}
	return st;
}

/* ---- scans ---- */
size_t clo_hip_scan_workspace_bytes(size_t numel, int elem_size, int sum_size) { (void) numel; (void) elem_size; (void) sum_size; return 1024; }
int clo_hip_scan_workspace_init(void* workspace, size_t workspace_bytes, void* stream) { (void) stream; if (!workspace) return CLO_HIP_EARGS; memset(workspace, 0, workspace_bytes); return 0; }
int clo_hip_scan_workspace_forget(void* workspace) { (void) workspace; return 0; }
int clo_hip_scan_workspace_set_epoch(void* workspace, unsigned epoch, void* stream) { (void) workspace; (void) epoch; (void) stream; return 0; }
int clo_hip_scan_exclusive_carry(const void* data_in, void* data_out, size_t numel, int elem_size, int elem_signed, int sum_size,
	const uint64_t* carry_in_dev, uint64_t* carry_out_dev, void* workspace, size_t workspace_bytes, void* stream) {
	(void) stream;
	uint64_t acc = carry_in_dev ? *carry_in_dev : 0;
	if (numel > 0) {
		if (!data_in || !data_out || !workspace) return CLO_HIP_EARGS;
		if (sum_size < elem_size) return CLO_HIP_EUNSUPPORTED;
		if (workspace_bytes < 1024) return CLO_HIP_EWORKSPACE;
		for (size_t i = 0; i < numel; ++i) {
			uint64_t v = load_elem((const char*) data_in + i * (size_t) elem_size, elem_size);
			if (elem_signed && elem_size < 8 && (v >> (8 * elem_size - 1))) v |= ~0ull << (8 * elem_size);
			memcpy((char*) data_out + i * (size_t) sum_size, &acc, (size_t) sum_size);   /* (little endian: the low sum_size bytes) */
			acc += v;
		}
	}
	if (carry_out_dev) *carry_out_dev = acc;
	return 0;
}
int clo_hip_scan_exclusive(const void* data_in, void* data_out, size_t numel, int elem_size, int elem_signed, int sum_size,
	void* workspace, size_t workspace_bytes, void* stream) {
	return clo_hip_scan_exclusive_carry(data_in, data_out, numel, elem_size, elem_signed, sum_size, NULL, NULL, workspace, workspace_bytes, stream);
}
int clo_hip_reduce_sum(const void* data_in, size_t numel, int elem_size, int elem_signed, uint64_t* total_dev, void* stream) {
	(void) stream;
	uint64_t acc = 0;
	for (size_t i = 0; i < numel; ++i) {
		uint64_t v = load_elem((const char*) data_in + i * (size_t) elem_size, elem_size);
		if (elem_signed && elem_size < 8 && (v >> (8 * elem_size - 1))) v |= ~0ull << (8 * elem_size);
		acc += v;
	}
	*total_dev = acc;
	return 0;
}
int clo_hip_scan_is_typed(int elem_type, int sum_type) {   /* CloType numbers: 8 half, 9 float, 10 double; sizes by number / 2 */
	static const int size_of[11] = { 1, 1, 2, 2, 4, 4, 8, 8, 2, 4, 8 };
	if (elem_type < 0 || elem_type > 10 || sum_type < 0 || sum_type > 10) return 0;
	return elem_type >= 8 || sum_type >= 8 || size_of[sum_type] < size_of[elem_type];
}
size_t clo_hip_scan_typed_workspace_bytes(size_t numel, int sum_type) { (void) numel; (void) sum_type; return 1024; }
size_t clo_hip_scan_fp_workspace_bytes(size_t numel, int sum_size) { (void) numel; (void) sum_size; return 1024; }
int clo_hip_scan_exclusive_typed(const void* data_in, void* data_out, size_t numel, int elem_type, int sum_type, void* workspace, size_t workspace_bytes, void* stream) {
	(void) data_in; (void) data_out; (void) numel; (void) elem_type; (void) sum_type; (void) workspace; (void) workspace_bytes; (void) stream;
	return CLO_HIP_EUNSUPPORTED;
}
int clo_hip_scan_exclusive_fp(const void* data_in, void* data_out, size_t numel, int elem_type, int sum_size, void* workspace, size_t workspace_bytes, void* stream) {
	(void) data_in; (void) data_out; (void) numel; (void) elem_type; (void) sum_size; (void) workspace; (void) workspace_bytes; (void) stream;
	return CLO_HIP_EUNSUPPORTED;
}

/* ---- bitonic / gselect: any correct order will do for what the stub is for (the networks' tie order is the GPU tests' business) ---- */
size_t clo_hip_bitonic_padded_numel(size_t numel) { size_t p = 1; while (p < numel) p <<= 1; return p; }
static int sort_in_place(void* data, size_t numel, int es, int key_shift, int key_bits, int key_kind, int descending, int* launches) {
	if (launches) *launches = 1;
	if (numel == 0) return 0;
	void* t = malloc(numel * (size_t) es);
	if (!t) return 2;
	const int st = stable_sort(data, t, numel, es, key_shift, key_bits, key_kind, descending);
	if (st == 0) memcpy(data, t, numel * (size_t) es);
	free(t);
	return st;
}
int clo_hip_bitonic_simple(void* data, size_t numel, int elem_size, int key_shift, int key_bits, int key_size, int key_kind, int descending, int* launches, void* stream) {
	(void) key_size; (void) stream; return sort_in_place(data, numel, elem_size, key_shift, key_bits, key_kind, descending, launches);
}
int clo_hip_bitonic_any(void* data, size_t numel, int elem_size, int key_shift, int key_bits, int key_size, int key_kind, int descending, int* launches, void* stream) {
	(void) key_size; (void) stream; return sort_in_place(data, numel, elem_size, key_shift, key_bits, key_kind, descending, launches);
}
int clo_hip_bitonic_tiled(void* data, size_t numel, int elem_size, int key_shift, int key_bits, int key_size, int key_kind, int descending, int* launches, void* stream) {
	(void) key_size; (void) stream; return sort_in_place(data, numel, elem_size, key_shift, key_bits, key_kind, descending, launches);
}
int clo_hip_gselect(const void* src, void* dst, size_t numel, int elem_size, int key_shift, int key_bits, int key_size, int key_kind, int descending, void* stream) {
	(void) key_size; (void) stream;
	return numel ? stable_sort(src, dst, numel, elem_size, key_shift, key_bits, key_kind, descending) : 0;
}

/* ---- what needs a GPU: refused ---- */
int clo_hip_bitonic_jit_create(int elem_type, int key_type, const char* compare, const char* get_key, const char* compiler_opts, void** handle, char** log) {
	(void) elem_type; (void) key_type; (void) compare; (void) get_key; (void) compiler_opts;
	if (handle) *handle = NULL;
	if (log) *log = NULL;
	return CLO_HIP_EUNSUPPORTED;
}
void clo_hip_bitonic_jit_destroy(void* handle) { (void) handle; }
int clo_hip_bitonic_jit_gselect(void* handle, const void* src, void* dst, size_t numel, void* stream) { (void) handle; (void) src; (void) dst; (void) numel; (void) stream; return CLO_HIP_EUNSUPPORTED; }
int clo_hip_bitonic_jit_sort(void* handle, void* data, size_t numel, int tiled, int* launches, void* stream) { (void) handle; (void) data; (void) numel; (void) tiled; (void) launches; (void) stream; return CLO_HIP_EUNSUPPORTED; }
size_t clo_hip_bitonic_jit_lds_bytes(void* handle, size_t numel, int tiled) { (void) handle; (void) numel; (void) tiled; return 0; }
int clo_hip_radix_jit_create(int elem_type, int key_type, const char* get_key, const char* compiler_opts, void** handle, char** log) {
	(void) elem_type; (void) key_type; (void) get_key; (void) compiler_opts;
	if (handle) *handle = NULL;
	if (log) *log = NULL;
	return CLO_HIP_EUNSUPPORTED;
}
void clo_hip_radix_jit_destroy(void* handle) { (void) handle; }
int clo_hip_radix_jit_sort(void* handle, const void* src, void* dst, void* pairs, void* pairs_tmp, size_t numel, int digit_bits, void* workspace, size_t workspace_bytes, void* stream) {
	(void) handle; (void) src; (void) dst; (void) pairs; (void) pairs_tmp; (void) numel; (void) digit_bits; (void) workspace; (void) workspace_bytes; (void) stream;
	return CLO_HIP_EUNSUPPORTED;
}
int clo_hip_rccl_unique_id(void* id_out) { (void) id_out; return CLO_HIP_EUNSUPPORTED; }
int clo_hip_rccl_comm_create(void** comm, const void* id_in, int rank, int world) { (void) comm; (void) id_in; (void) rank; (void) world; return CLO_HIP_EUNSUPPORTED; }
int clo_hip_rccl_comm_destroy(void* comm) { (void) comm; return 0; }
int clo_hip_rccl_comm_abort(void* comm) { (void) comm; return 0; }
int clo_hip_rccl_comm_async_error(void* comm) { (void) comm; return 0; }
int clo_hip_rccl_all_gather_u64(void* comm, const uint64_t* send_dev, uint64_t* recv_dev, size_t count, void* stream) { (void) comm; (void) send_dev; (void) recv_dev; (void) count; (void) stream; return CLO_HIP_EUNSUPPORTED; }
int clo_hip_rccl_all_to_all_v(void* comm, int rank, int world, const void* send_dev, const size_t* send_bytes, const size_t* send_offset_bytes,
	void* recv_dev, const size_t* recv_bytes, const size_t* recv_offset_bytes, void* stream) {
	(void) comm; (void) rank; (void) world; (void) send_dev; (void) send_bytes; (void) send_offset_bytes; (void) recv_dev; (void) recv_bytes; (void) recv_offset_bytes; (void) stream;
	return CLO_HIP_EUNSUPPORTED;
}
