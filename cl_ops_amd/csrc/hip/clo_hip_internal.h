// clo_hip_internal.h — shared device-side helpers for the HIP kernels
// (not part of the C-ABI). gfx950 / wave64 only.
#ifndef CLO_HIP_INTERNAL_H
#define CLO_HIP_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

// ---- workspace header (first CLO_WS_HEADER_BYTES of every workspace) ----
// word 0: status (non-zero = a bounded spin gave up)
// word 2: epoch of the last completed call (scan), word 3: work-groups that have left
// words 16..79: work-queue tickets
#define CLO_WS_STATUS_OFFSET 0
#define CLO_WS_EPOCH_WORD    2
#define CLO_WS_DONE_WORD     3
#define CLO_WS_TICKET_WORD   16
#define CLO_WS_MAX_PASSES    64
#define CLO_WS_HEADER_BYTES  512

// The environment switches of the HIP layer, read in ONE place (clo_hip_runtime.hip: clo_hip_env_refresh) and only
// when an object is made (clo_sort_new / clo_scan_new / clo_shard_sort_new call it) or a test asks — never per call:
//   CLO_MAX_SPINS        bound of every look-back poll loop (default CLO_MAX_SPINS below; tests force a give-up with 0)
//   CLO_RADIX_SWEEP      0: never take the single-sweep radix passes, 1: whenever possible; unset: the library's choice
//   CLO_R1_POOLS         8: the single-sweep passes draw tiles from one ticket pool per XCD (256-CU devices only)
//   CLO_RADIX_NO_DIGITS  set: no digit stream between the chain-free passes (tests compare both)
//   CLO_BITONIC_MERGE2   0: no two-tile merge passes in the tiled bitonic schedule, 2: at every stage; unset / 1: where they save a pass
struct clo_hip_env_t { unsigned max_spins; int radix_sweep; int r1_pools; int no_digits; int bitonic_merge2; };
const clo_hip_env_t* clo_hip_env();
int clo_radixw_preload();   // clo_hip_radixw.hip

#ifdef __HIPCC__
#include <hip/hip_runtime.h>

// Optional per-kernel timing (clo_hip_timing_* in clo_hip.h): when enabled,
// launch sites bracket a kernel with a pair of HIP events on its own stream.
void clo_timing_begin(const char* label, hipStream_t s);
void clo_timing_end(hipStream_t s);
struct clo_timing_scope {
	hipStream_t s;
	clo_timing_scope(const char* label, hipStream_t stream) : s(stream) { clo_timing_begin(label, s); }
	~clo_timing_scope() { clo_timing_end(s); }
};

// Order-preserving key transform for signed / floating-point radix keys,
// applied to the key field [shift, shift+bits) of an element when the first
// pass reads the source and undone when the last pass stores the result:
//   kind 1 (two's complement): flip the sign bit;
//   kind 2 (IEEE-754): negative -> flip every key bit, else flip the sign bit.
// kind 0 = unsigned keys, nothing to do (a wave-uniform branch).
struct clo_keyx {
	unsigned long long sign, field;
	int kind;
};
template <typename E>
__device__ __forceinline__ E clo_keyx_fwd(E x, const clo_keyx& k) {
	if (k.kind == 1) return (E) (x ^ (E) k.sign);
	if (k.kind == 2) return (E) (x ^ ((x & (E) k.sign) ? (E) k.field : (E) k.sign));
	return x;
}
template <typename E>
__device__ __forceinline__ E clo_keyx_inv(E x, const clo_keyx& k) {
	if (k.kind == 1) return (E) (x ^ (E) k.sign);
	if (k.kind == 2) return (E) (x ^ ((x & (E) k.sign) ? (E) k.sign : (E) k.field));
	return x;
}
inline clo_keyx clo_keyx_make(int kind, int key_shift, int key_bits) {
	clo_keyx k = { 0, 0, 0 };
	if (kind == 1 || kind == 2) {
		k.kind = kind;
		k.sign = 1ull << (key_shift + key_bits - 1);
		k.field = (key_bits >= 64 ? ~0ull : ((1ull << key_bits) - 1ull)) << key_shift;
	}
	return k;
}

// Radix passes (clo_hip_radix4.hip) and their histogram / counter-scan steps
// (clo_hip_radixw.hip).
size_t clo_radix4_workspace_bytes(size_t n, int elem_size, int digit_bits, int key_bits);
size_t clo_radix4_partition_workspace_bytes(size_t n, int elem_size, int bits);
int clo_radix4_partition(const void* src, void* dst, size_t n, int elem_size, unsigned shift, int bits,
	unsigned long long* counts, void* ws, hipStream_t s);
int clo_radix4_sort(const void* src, void* dst, void* tmp, size_t n, int elem_size, int key_shift,
	int key_bits, int digit_bits, clo_keyx kx, const unsigned char* first_dig, void* ws, hipStream_t s);
int clo_radix4_takes_first_digits(size_t n, int elem_size, int digit_bits);
size_t clo_radix4_lds_bytes(int elem_size, int digit_bits);
// single-sweep passes (clo_hip_radix1.hip); ws: its own region, status: the workspace's status word
int clo_radix1_applies(size_t n, int elem_size, int digit_bits);
size_t clo_radix1_workspace_bytes(size_t n, int elem_size, int key_bits);
int clo_radix1_sort(const void* src, void* dst, void* tmp, size_t n, int elem_size, int key_shift, int key_bits,
	clo_keyx kx, void* ws, unsigned* status, hipStream_t s);
// the tiled bitonic schedule, one translation unit per element size
int clo_bitonic_tiled_e1(void* data, size_t numel, int key_shift, int key_bits, int key_size, int key_kind, int descending, int* launches, hipStream_t s);
int clo_bitonic_tiled_e2(void* data, size_t numel, int key_shift, int key_bits, int key_size, int key_kind, int descending, int* launches, hipStream_t s);
int clo_bitonic_tiled_e4(void* data, size_t numel, int key_shift, int key_bits, int key_size, int key_kind, int descending, int* launches, hipStream_t s);
int clo_bitonic_tiled_e8(void* data, size_t numel, int key_shift, int key_bits, int key_size, int key_kind, int descending, int* launches, hipStream_t s);
size_t clo_radixw_lds_bytes(int digit_bits);
// tinfo: one word per tile, 1 = one bin holds the whole tile (read by the pass kernel)
// partial: the counter scan's workspace (clo_radixw_partial_rows(tiles) rows of 1 << bits words); the histogram launch zeroes
// the hand-off words of the scan that follows it on the stream (clo_radixw_launch_offsets), which returns in *dbase the row
// of digit bases the pass kernel adds to toff[tile][digit] (null: toff is final)
int clo_radixw_launch_tilehist(const void* in, size_t n, int elem_size, int bits, unsigned shift, unsigned mask,
	unsigned* thist, unsigned* tinfo, unsigned* partial, unsigned tiles, bool big, clo_keyx kx, hipStream_t s);
int clo_radixw_launch_tilehist_bytes(const unsigned char* dig, size_t n, int elem_size, int bits, unsigned mask,
	unsigned* thist, unsigned* tinfo, unsigned* partial, unsigned tiles, bool big, hipStream_t s);
int clo_radixw_launch_offsets(int bits, const unsigned* thist, unsigned tiles, unsigned* partial, unsigned* toff,
	const unsigned** dbase, hipStream_t s);
size_t clo_radixw_partial_rows(size_t tiles);

// ---- segmented sorts (clo_hip_radix_sort_segmented): several segments of one array, each sorted on its own, in
// SHARED launches. A launch's tiles and counter-scan chunks are numbered through all segments; two small tables,
// built on the device by one launch (clo_radix_seg_build_kernel), say what a tile / a chunk is. The INPUT of a
// segment may lie in several pieces anywhere in the source (the sub-bucket a rank of the sharded sort receives
// comes as one piece per source rank): a tile never straddles pieces, its output goes to the segment's contiguous
// place, so the first pass gathers the pieces as a by-product. Chunks never straddle segments. ----
#define CLO_SEG_MAX 256
struct clo_seg_tile { unsigned in_base, count_seg, out_base, seg_n; };  // the tile's first element in the source; count | segment << 16 | (second source) << 31; where the segment starts in the output; its length
struct clo_seg_chunk { unsigned t0, tend, c_first, seg_last; };        // tiles [t0, tend) of the launch; first chunk of the same segment; segment | (last chunk of it) << 31
struct clo_seg_pieces {                                                // (a kernel argument: no host buffer has to outlive the call) pieces in segment order
	unsigned n[CLO_SEG_MAX], in_base[CLO_SEG_MAX];
	unsigned short seg[CLO_SEG_MAX];                                   // (bit 15: the piece lies in the call's second source)
};
struct clo_seg_tables { const clo_seg_tile* tiles; const clo_seg_chunk* chunks; unsigned ntiles, nchunks, nseg; };
int clo_radixw_launch_tilehist_seg(const void* in, const void* in2, const clo_seg_tables& sg, int elem_size, int bits, unsigned shift, unsigned mask,
	unsigned* thist, unsigned* tinfo, unsigned* partial, bool big, hipStream_t s);
int clo_radixw_launch_tilehist_bytes_seg(const unsigned char* dig, const clo_seg_tables& sg, int elem_size, int bits, unsigned mask,
	unsigned* thist, unsigned* tinfo, unsigned* partial, bool big, hipStream_t s);
// *dbase: nseg rows of digit bases (row = segment)
int clo_radixw_launch_offsets_seg(int bits, const unsigned* thist, const clo_seg_tables& sg, unsigned* partial, unsigned* toff,
	const unsigned** dbase, hipStream_t s);
size_t clo_radixw_partial_rows_seg(size_t chunks, size_t nseg);
// pieces (npieces <= CLO_SEG_MAX, ordered by segment): lengths, first elements, segments, source (0 / 1; may be null: all 0)
int clo_radixw_seg_build(const size_t* piece_n, const size_t* piece_base, const int* piece_seg, const int* piece_src, int npieces, int nseg, size_t tile,
	clo_seg_tile* tiles, clo_seg_chunk* chunks, unsigned* ntiles, unsigned* nchunks, hipStream_t s);
void clo_radixw_seg_bounds(size_t numel, int npieces, int nseg, size_t tile, size_t* max_tiles, size_t* max_chunks);
size_t clo_radix4_seg_workspace_bytes(size_t n, int nseg, int elem_size, int digit_bits);
int clo_radix4_sort_segmented(const void* src, const void* src2, void* a, void* b, size_t n, const size_t* seg_counts, int nseg,
	const size_t* piece_n, const size_t* piece_base, const int* piece_seg, const int* piece_src, int npieces, int elem_size, int key_shift,
	int key_bits, int digit_bits, void* ws, hipStream_t s, int* result_in_b);

typedef unsigned long long clo_u64;

// Look-back granule: one naturally aligned 8-byte word written by ONE store:
//   bits 63..34 epoch of the call that wrote it (1 .. CLO_LB_EPOCH_MAX; 0 = never written)
//   bits 33..32 state (1 = tile aggregate, 2 = inclusive prefix)
//   bits 31..0  value
// The data is the flag (cdna_hip_programming.md G16 R2): relaxed agent-scope
// (sc1) store on the producer, relaxed agent-scope load polls on the consumer,
// no fences, no separate flag word.
#define CLO_LB_EPOCH_MAX ((1u << 30) - 1u)
#define CLO_LB_AGG    1u
#define CLO_LB_PREFIX 2u

__device__ __forceinline__ clo_u64 clo_lb_pack(unsigned epoch, unsigned state, unsigned value) {
	return ((clo_u64) ((epoch << 2) | state) << 32) | value;
}
__device__ __forceinline__ unsigned clo_lb_tag(clo_u64 g) { return (unsigned) (g >> 32); }
__device__ __forceinline__ unsigned clo_lb_val(clo_u64 g) { return (unsigned) g; }

__device__ __forceinline__ clo_u64 clo_ld_agent(const clo_u64* p) {
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void clo_st_agent(clo_u64* p, clo_u64 v) {
	__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every poll loop is bounded: after this many failed polls the wave records
// the failure in the workspace status word and carries on (wrong output, but
// the grid drains). ~1 us per poll under load => seconds before giving up.
#define CLO_MAX_SPINS (1u << 22)

__device__ __forceinline__ unsigned clo_lane_id() {
	return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Number of set bits of `mask` strictly below the calling lane.
__device__ __forceinline__ unsigned clo_mbcnt(clo_u64 mask) {
	return __builtin_amdgcn_mbcnt_hi((unsigned) (mask >> 32),
		__builtin_amdgcn_mbcnt_lo((unsigned) mask, 0u));
}

// Work-group barrier that orders LDS traffic only: global loads and stores requested before it stay in flight across it
// (a __syncthreads() makes every wave wait for ALL its outstanding memory operations first). For barriers that hand over
// nothing but LDS contents: the radix splits, the scan kernel's tile loop.
__device__ __forceinline__ void clo_lds_barrier() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
	__builtin_amdgcn_s_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Wave64 prefix sums of INTEGER values (32 or 64 bits) on the DPP network — four shifts inside the rows of 16 lanes,
// two row broadcasts — with no LDS traffic and nothing to wait for. (Rounds 1-4 went through __shfl_up, i.e. one
// ds_bpermute_b32 per step and `s_waitcnt lgkmcnt(0)` behind each: six dependent LDS round trips per scan, 48 of them per
// tile in the scan kernel. Integer addition is associative: the sums are the same bits.)
template <int CTRL, int ROW_MASK, typename T>
__device__ __forceinline__ T clo_dpp_add(T x) {
	static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- or 64-bit integers");
	if constexpr (sizeof(T) == 4) {
		return (T) ((unsigned) x + (unsigned) __builtin_amdgcn_update_dpp(0, (int) x, CTRL, ROW_MASK, 0xF, true));
	} else {
		const unsigned long long v = (unsigned long long) x;
		const unsigned lo = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) v, CTRL, ROW_MASK, 0xF, true);
		const unsigned hi = (unsigned) __builtin_amdgcn_update_dpp(0, (int) (unsigned) (v >> 32), CTRL, ROW_MASK, 0xF, true);
		return (T) (v + (((unsigned long long) hi << 32) | lo));
	}
}

// Inclusive scan across the 64 lanes of a wave.
template <typename T>
__device__ __forceinline__ T clo_wave_scan_inclusive(T x, unsigned lane) {
	(void) lane;
	x = clo_dpp_add<0x111, 0xF>(x);   // row_shr:1
	x = clo_dpp_add<0x112, 0xF>(x);   // row_shr:2
	x = clo_dpp_add<0x114, 0xF>(x);   // row_shr:4
	x = clo_dpp_add<0x118, 0xF>(x);   // row_shr:8
	x = clo_dpp_add<0x142, 0xA>(x);   // row_bcast:15 into rows 1 and 3
	x = clo_dpp_add<0x143, 0xC>(x);   // row_bcast:31 into rows 2 and 3
	return x;
}

// Sum over the 64 lanes, in every lane (wave-uniform: it comes back through a scalar register).
template <typename T>
__device__ __forceinline__ T clo_wave_reduce_sum(T x) {
	const T incl = clo_wave_scan_inclusive<T>(x, 0u);
	if constexpr (sizeof(T) == 4) {
		return (T) (unsigned) __builtin_amdgcn_readlane((int) incl, 63);
	} else {
		const unsigned long long v = (unsigned long long) incl;
		const unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) v, 63);
		const unsigned hi = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (v >> 32), 63);
		return (T) (((unsigned long long) hi << 32) | lo);
	}
}

#endif  // __HIPCC__
#endif
