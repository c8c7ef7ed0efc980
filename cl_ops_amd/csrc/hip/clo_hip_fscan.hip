// clo_hip_fscan.hip — exclusive prefix sums in FLOATING POINT (sum type half, float or
// double; elements of any CloType, converted to the sum type on load as upstream's
// kernels do: scan/clo_scan_blelloch.cl:79-80 with CLO_SCAN_SUM_TYPE float/double),
// and — round 3 — every other pair of types upstream's generic kernel accepts and the
// single-pass integer kernel does not: floating-point elements summed in an integer
// type (every element truncated by the cast, as upstream's `(CLO_SCAN_SUM_TYPE) x`
// does) and integer sums NARROWER than the elements (the cast keeps the low bits).
//
// Not the single-pass kernel of clo_hip_scan.hip: its look-back adds up whichever
// predecessors have published when it polls, which is harmless in modular integer
// arithmetic and would make a floating-point result depend on timing. Here the
// order of every addition is fixed by the data layout alone — reduce, scan the
// tile sums (recursively), apply — so two runs give the same bits:
//   tile = 256 threads x 16 consecutive elements; a thread adds its 16 left to
//   right, a wave scans its 64 thread sums (shuffle tree), the 4 wave sums are
//   added left to right; tiles are combined by the same procedure one level up
//   (4096 tile sums per upper tile; three levels cover any array).
// Upstream's order is the Blelloch tree of its own work-group shape, so results
// agree with it to rounding, not bit for bit (no two work-group sizes of upstream
// agree bit for bit either); like upstream's down-sweep, no step subtracts. 3 element
// streams instead of 2.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "clo_hip.h"
#include "clo_hip_internal.h"

namespace {

constexpr int FS_THREADS = 256;
constexpr int FS_ITEMS = 16;
constexpr int FS_TILE = FS_THREADS * FS_ITEMS;   // 4096

// the cast upstream writes: (CLO_SCAN_SUM_TYPE) element
template <typename TSum, typename TIn>
__device__ __forceinline__ TSum fs_cvt(TIn x) { return (TSum) x; }

// wave shuffle of any 1..8-byte value by its bits
template <typename T>
__device__ __forceinline__ T fs_shfl_up(T v, int off) {
	if constexpr (sizeof(T) == 8) {
		long long b;
		__builtin_memcpy(&b, &v, 8);
		b = __shfl_up(b, off, 64);
		T r;
		__builtin_memcpy(&r, &b, 8);
		return r;
	} else {
		unsigned b = 0;
		__builtin_memcpy(&b, &v, sizeof(T));
		b = (unsigned) __shfl_up((int) b, off, 64);
		T r;
		__builtin_memcpy(&r, &b, sizeof(T));
		return r;
	}
}

// thread sums -> exclusive offset of every thread inside the tile, and the tile total
template <typename TSum>
__device__ __forceinline__ TSum fs_block_exclusive(TSum mine, TSum* total, TSum* s_w) {
	const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	TSum incl = mine;
	#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const TSum y = fs_shfl_up<TSum>(incl, off);
		if (lane >= (unsigned) off) incl += y;
	}
	if (lane == 63) s_w[wave] = incl;
	__syncthreads();
	TSum base = 0, tot = 0;
	#pragma unroll
	for (unsigned w = 0; w < FS_THREADS / 64; ++w) {
		if (w < wave) base += s_w[w];
		tot += s_w[w];
	}
	*total = tot;
	// The lanes before this one: the inclusive value of the lane below, never `incl - mine` —
	// in floating point (a + b) - b is not a: a thread sum much larger than the prefix in
	// front of it would swallow that prefix (thread sums 16, 1e9: the second thread's
	// offset must be 16, upstream's down-sweep never subtracts either).
	TSum excl = fs_shfl_up<TSum>(incl, 1);
	if (lane == 0) excl = 0;
	return base + excl;
}

template <typename TIn, typename TSum>
__global__ __launch_bounds__(FS_THREADS)
void clo_fscan_reduce_kernel(const TIn* __restrict__ in, size_t n, TSum* __restrict__ sums) {
	__shared__ TSum s_w[FS_THREADS / 64];
	const size_t base = (size_t) blockIdx.x * FS_TILE + (size_t) threadIdx.x * FS_ITEMS;
	TSum mine = 0;
	#pragma unroll
	for (int i = 0; i < FS_ITEMS; ++i) if (base + i < n) mine += fs_cvt<TSum, TIn>(in[base + i]);
	TSum total;
	(void) fs_block_exclusive<TSum>(mine, &total, s_w);
	if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// out[i] = offset of the tile (tile_excl[tile], or 0) + exclusive scan inside the tile.
// in == out is allowed when TIn == TOut (the upper levels scan their sums in place).
template <typename TIn, typename TOut, typename TSum>
__global__ __launch_bounds__(FS_THREADS)
void clo_fscan_apply_kernel(const TIn* in, TOut* out, size_t n, const TSum* __restrict__ tile_excl) {
	__shared__ TSum s_w[FS_THREADS / 64];
	const size_t base = (size_t) blockIdx.x * FS_TILE + (size_t) threadIdx.x * FS_ITEMS;
	TSum v[FS_ITEMS];
	TSum mine = 0;
	#pragma unroll
	for (int i = 0; i < FS_ITEMS; ++i) {
		v[i] = base + i < n ? fs_cvt<TSum, TIn>(in[base + i]) : (TSum) 0;
		mine += v[i];
	}
	TSum total;
	TSum run = fs_block_exclusive<TSum>(mine, &total, s_w) + (tile_excl ? tile_excl[blockIdx.x] : (TSum) 0);
	#pragma unroll
	for (int i = 0; i < FS_ITEMS; ++i) {
		if (base + i < n) out[base + i] = (TOut) run;
		run += v[i];
	}
}

template <typename TIn, typename TOut, typename TSum>
int fs_scan(const TIn* in, TOut* out, size_t n, TSum* ws, hipStream_t s) {
	const size_t t1 = (n + FS_TILE - 1) / FS_TILE;
	clo_timing_scope timing("scan", s);
	if (t1 <= 1) {
		hipLaunchKernelGGL((clo_fscan_apply_kernel<TIn, TOut, TSum>), dim3(1), dim3(FS_THREADS), 0, s, in, out, n, (const TSum*) nullptr);
		return (int) hipGetLastError();
	}
	TSum* sums1 = ws;
	hipLaunchKernelGGL((clo_fscan_reduce_kernel<TIn, TSum>), dim3((unsigned) t1), dim3(FS_THREADS), 0, s, in, n, sums1);
	const size_t t2 = (t1 + FS_TILE - 1) / FS_TILE;
	if (t2 <= 1) {
		hipLaunchKernelGGL((clo_fscan_apply_kernel<TSum, TSum, TSum>), dim3(1), dim3(FS_THREADS), 0, s, sums1, sums1, t1, (const TSum*) nullptr);
	} else {
		TSum* sums2 = sums1 + t1;
		hipLaunchKernelGGL((clo_fscan_reduce_kernel<TSum, TSum>), dim3((unsigned) t2), dim3(FS_THREADS), 0, s, sums1, t1, sums2);
		const size_t t3 = (t2 + FS_TILE - 1) / FS_TILE;
		if (t3 > 1) return CLO_HIP_EUNSUPPORTED;   // (> 2^36 elements)
		hipLaunchKernelGGL((clo_fscan_apply_kernel<TSum, TSum, TSum>), dim3(1), dim3(FS_THREADS), 0, s, sums2, sums2, t2, (const TSum*) nullptr);
		hipLaunchKernelGGL((clo_fscan_apply_kernel<TSum, TSum, TSum>), dim3((unsigned) t2), dim3(FS_THREADS), 0, s, sums1, sums1, t1, (const TSum*) sums2);
	}
	hipLaunchKernelGGL((clo_fscan_apply_kernel<TIn, TOut, TSum>), dim3((unsigned) t1), dim3(FS_THREADS), 0, s, in, out, n, (const TSum*) sums1);
	return (int) hipGetLastError();
}

template <typename TSum>
int fs_dispatch(const void* in, void* out, size_t n, int elem_type, void* ws, hipStream_t s) {
	TSum* w = (TSum*) ws;
	TSum* o = (TSum*) out;
	switch (elem_type) {   // CloType numbers (clo_common.h)
		case 0: return fs_scan<int8_t, TSum, TSum>((const int8_t*) in, o, n, w, s);
		case 1: return fs_scan<uint8_t, TSum, TSum>((const uint8_t*) in, o, n, w, s);
		case 2: return fs_scan<int16_t, TSum, TSum>((const int16_t*) in, o, n, w, s);
		case 3: return fs_scan<uint16_t, TSum, TSum>((const uint16_t*) in, o, n, w, s);
		case 4: return fs_scan<int32_t, TSum, TSum>((const int32_t*) in, o, n, w, s);
		case 5: return fs_scan<uint32_t, TSum, TSum>((const uint32_t*) in, o, n, w, s);
		case 6: return fs_scan<int64_t, TSum, TSum>((const int64_t*) in, o, n, w, s);
		case 7: return fs_scan<uint64_t, TSum, TSum>((const uint64_t*) in, o, n, w, s);
		case 8: return fs_scan<_Float16, TSum, TSum>((const _Float16*) in, o, n, w, s);
		case 9: return fs_scan<float, TSum, TSum>((const float*) in, o, n, w, s);
		case 10: return fs_scan<double, TSum, TSum>((const double*) in, o, n, w, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

// Integer sums of the pairs the single-pass kernel does not take: floating-point elements, and elements
// wider than the sum (only these are compiled: the others go to clo_hip_scan_exclusive).
template <typename TSum>
int fs_dispatch_int(const void* in, void* out, size_t n, int elem_type, void* ws, hipStream_t s) {
	TSum* w = (TSum*) ws;
	TSum* o = (TSum*) out;
	switch (elem_type) {
		case 8: return fs_scan<_Float16, TSum, TSum>((const _Float16*) in, o, n, w, s);
		case 9: return fs_scan<float, TSum, TSum>((const float*) in, o, n, w, s);
		case 10: return fs_scan<double, TSum, TSum>((const double*) in, o, n, w, s);
		default: break;
	}
	if constexpr (sizeof(TSum) < 2) {
		if (elem_type == 2) return fs_scan<int16_t, TSum, TSum>((const int16_t*) in, o, n, w, s);
		if (elem_type == 3) return fs_scan<uint16_t, TSum, TSum>((const uint16_t*) in, o, n, w, s);
	}
	if constexpr (sizeof(TSum) < 4) {
		if (elem_type == 4) return fs_scan<int32_t, TSum, TSum>((const int32_t*) in, o, n, w, s);
		if (elem_type == 5) return fs_scan<uint32_t, TSum, TSum>((const uint32_t*) in, o, n, w, s);
	}
	if constexpr (sizeof(TSum) < 8) {
		if (elem_type == 6) return fs_scan<int64_t, TSum, TSum>((const int64_t*) in, o, n, w, s);
		if (elem_type == 7) return fs_scan<uint64_t, TSum, TSum>((const uint64_t*) in, o, n, w, s);
	}
	return CLO_HIP_EUNSUPPORTED;
}

const int k_type_size[11] = { 1, 1, 2, 2, 4, 4, 8, 8, 2, 4, 8 };   // CloType numbering (clo_common.h)

}  // namespace

extern "C" {

// 1: this pair of CloTypes is scanned here (clo_hip_scan_exclusive_typed), 0: by the single-pass integer kernel
int clo_hip_scan_is_typed(int elem_type, int sum_type) {
	if (elem_type < 0 || elem_type > 10 || sum_type < 0 || sum_type > 10) return 0;
	if (sum_type >= 8) return 1;                                  // half / float / double sums
	if (elem_type >= 8) return 1;                                 // floating-point elements into integer sums
	return k_type_size[sum_type] < k_type_size[elem_type];       // integer sums narrower than the elements
}

size_t clo_hip_scan_typed_workspace_bytes(size_t numel, int sum_type) {
	if (sum_type < 0 || sum_type > 10) return 0;
	const size_t t1 = (numel + FS_TILE - 1) / FS_TILE;
	const size_t t2 = (t1 + FS_TILE - 1) / FS_TILE;
	return (t1 + t2 + 8) * (size_t) k_type_size[sum_type];
}

int clo_hip_scan_exclusive_typed(const void* data_in, void* data_out, size_t numel, int elem_type, int sum_type,
	void* workspace, size_t workspace_bytes, void* stream) {
	if (numel == 0) return 0;
	if (!data_in || !data_out || !workspace) return CLO_HIP_EARGS;
	if (!clo_hip_scan_is_typed(elem_type, sum_type)) return CLO_HIP_EUNSUPPORTED;
	if (workspace_bytes < clo_hip_scan_typed_workspace_bytes(numel, sum_type)) return CLO_HIP_EWORKSPACE;
	hipStream_t s = (hipStream_t) stream;
	switch (sum_type) {
		case 0: return fs_dispatch_int<int8_t>(data_in, data_out, numel, elem_type, workspace, s);
		case 1: return fs_dispatch_int<uint8_t>(data_in, data_out, numel, elem_type, workspace, s);
		case 2: return fs_dispatch_int<int16_t>(data_in, data_out, numel, elem_type, workspace, s);
		case 3: return fs_dispatch_int<uint16_t>(data_in, data_out, numel, elem_type, workspace, s);
		case 4: return fs_dispatch_int<int32_t>(data_in, data_out, numel, elem_type, workspace, s);
		case 5: return fs_dispatch_int<uint32_t>(data_in, data_out, numel, elem_type, workspace, s);
		case 6: return fs_dispatch_int<int64_t>(data_in, data_out, numel, elem_type, workspace, s);
		case 7: return fs_dispatch_int<uint64_t>(data_in, data_out, numel, elem_type, workspace, s);
		case 8: return fs_dispatch<_Float16>(data_in, data_out, numel, elem_type, workspace, s);
		case 9: return fs_dispatch<float>(data_in, data_out, numel, elem_type, workspace, s);
		case 10: return fs_dispatch<double>(data_in, data_out, numel, elem_type, workspace, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

size_t clo_hip_scan_fp_workspace_bytes(size_t numel, int sum_size) {
	const size_t t1 = (numel + FS_TILE - 1) / FS_TILE;
	const size_t t2 = (t1 + FS_TILE - 1) / FS_TILE;
	return (t1 + t2 + 8) * (size_t) (sum_size == 8 ? 8 : 4);
}

int clo_hip_scan_exclusive_fp(const void* data_in, void* data_out, size_t numel, int elem_type, int sum_size,
	void* workspace, size_t workspace_bytes, void* stream) {
	if (numel == 0) return 0;
	if (!data_in || !data_out || !workspace) return CLO_HIP_EARGS;
	if (sum_size != 4 && sum_size != 8) return CLO_HIP_EUNSUPPORTED;
	if (workspace_bytes < clo_hip_scan_fp_workspace_bytes(numel, sum_size)) return CLO_HIP_EWORKSPACE;
	hipStream_t s = (hipStream_t) stream;
	return sum_size == 8 ? fs_dispatch<double>(data_in, data_out, numel, elem_type, workspace, s)
	                     : fs_dispatch<float>(data_in, data_out, numel, elem_type, workspace, s);
}

}  // extern "C"
