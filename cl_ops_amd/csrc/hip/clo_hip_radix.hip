// clo_hip_radix.hip — C-ABI entry points of the stable LSD radix sort for
// gfx950 (the "satradix" replacement) and of the MSD bucket split.
//
// Upstream, every digit pass is four launches over four buffers
// (sort/clo_sort_satradix.c:264-313): satradix_localsort (b one-bit splits, each a
// full LDS Blelloch scan over the tile), satradix_histogram, a 3-kernel scan of
// num_wgs*radix counters, satradix_scatter (sort/clo_sort_satradix.cl:34-258) —
// about 5 element streams plus 6 counter streams through HBM per digit.
//
// Here a pass keeps that decomposition (per-tile histogram -> counters scan ->
// stable tile sort + scatter) but handles two digits of <= 4 bits per trip
// through HBM (two local splits inside the work-group), reads every element
// twice and writes it once per pass, and no kernel waits on another
// work-group: clo_hip_radix4.hip (ranking, pass kernel, small arrays, host
// side), clo_hip_radixw.hip (histogram and counter scan). Stability per pass +
// LSD order give exactly the order the reference produces (stable ascending by
// key), for any digit width.
#include <hip/hip_runtime.h>

#include "clo_hip.h"
#include "clo_hip_internal.h"

namespace {

// ---- bucket sizes of the MSD split alone (the partition yields them too) ----
constexpr int HIST_THREADS = 256;
constexpr int HIST_ITEMS = 16;  // elements per thread per block-iteration

template <typename E>
__global__ __launch_bounds__(HIST_THREADS)
void clo_msd_hist_kernel(const E* __restrict__ in, size_t n, unsigned shift, unsigned mask,
	unsigned long long* __restrict__ counts) {
	__shared__ unsigned h[HIST_THREADS / 64][8];
	const unsigned tid = threadIdx.x, wave = tid >> 6;
	if (tid < (HIST_THREADS / 64) * 8) (&h[0][0])[tid] = 0;
	__syncthreads();
	const size_t chunk = (size_t) HIST_THREADS * HIST_ITEMS;
	for (size_t base = (size_t) blockIdx.x * chunk; base < n; base += (size_t) gridDim.x * chunk) {
		#pragma unroll
		for (int i = 0; i < HIST_ITEMS; ++i) {
			const size_t idx = base + (size_t) i * HIST_THREADS + tid;
			if (idx < n) atomicAdd(&h[wave][(unsigned) (in[idx] >> shift) & mask & 7u], 1u);
		}
	}
	__syncthreads();
	if (tid <= mask && tid < 8) {
		unsigned s = 0;
		#pragma unroll
		for (int w = 0; w < HIST_THREADS / 64; ++w) s += h[w][tid];
		if (s) atomicAdd(&counts[tid], (unsigned long long) s);
	}
}

template <typename E>
int msd_hist_impl(const void* src, size_t n, unsigned shift, unsigned mask, uint64_t* counts, hipStream_t s) {
	unsigned blocks = (unsigned) ((n + HIST_THREADS * HIST_ITEMS - 1) / (HIST_THREADS * HIST_ITEMS));
	if (blocks > 2048) blocks = 2048;
	if (blocks == 0) blocks = 1;
	hipLaunchKernelGGL((clo_msd_hist_kernel<E>), dim3(blocks), dim3(HIST_THREADS), 0, s,
		(const E*) src, n, shift, mask, (unsigned long long*) counts);
	return (int) hipGetLastError();
}

}  // namespace

extern "C" {

size_t clo_hip_radix_workspace_bytes(size_t numel, int elem_size, int key_bits, int digit_bits) {
	if (digit_bits < 1 || digit_bits > 8 || key_bits < 1) return 0;
	return clo_radix4_workspace_bytes(numel, elem_size, digit_bits, key_bits);
}

int clo_hip_radix_polls(size_t numel, int elem_size, int digit_bits) {
	return clo_radix1_applies(numel, elem_size, digit_bits) ? 1 : 0;
}

int clo_hip_radix_takes_first_digits(size_t numel, int elem_size, int key_kind, int digit_bits) {
	return key_kind == 0 ? clo_radix4_takes_first_digits(numel, elem_size, digit_bits) : 0;
}

int clo_hip_radix_sort(const void* src, void* dst, void* tmp, size_t numel,
	int elem_size, int key_shift, int key_bits, int key_kind, int digit_bits,
	void* workspace, size_t workspace_bytes, void* stream) {
	return clo_hip_radix_sort_fed(src, dst, tmp, numel, elem_size, key_shift, key_bits, key_kind, digit_bits, nullptr,
		workspace, workspace_bytes, stream);
}

int clo_hip_radix_sort_fed(const void* src, void* dst, void* tmp, size_t numel,
	int elem_size, int key_shift, int key_bits, int key_kind, int digit_bits, const unsigned char* first_digits,
	void* workspace, size_t workspace_bytes, void* stream) {

	if (numel == 0) return 0;
	if (!src || !dst || !tmp || !workspace || tmp == src || tmp == dst) return CLO_HIP_EARGS;
	if (key_bits < 1 || key_shift < 0 || key_shift + key_bits > 8 * elem_size) return CLO_HIP_EARGS;
	if (key_kind < 0 || key_kind > 2) return CLO_HIP_EARGS;
	if (key_kind == 2 && key_bits != 16 && key_bits != 32 && key_bits != 64) return CLO_HIP_EARGS;
	if (digit_bits < 1 || digit_bits > 8) return CLO_HIP_EUNSUPPORTED;
	if (numel > 0xffffffffull) return CLO_HIP_EARGS;  // 32-bit positions, as upstream's uint indices
	if (workspace_bytes < clo_hip_radix_workspace_bytes(numel, elem_size, key_bits, digit_bits)) return CLO_HIP_EWORKSPACE;
	hipStream_t s = (hipStream_t) stream;
	const clo_keyx kx = clo_keyx_make(key_kind, key_shift, key_bits);
	return clo_radix4_sort(src, dst, tmp, numel, elem_size, key_shift, key_bits, digit_bits, kx, first_digits, workspace, s);
}

int clo_hip_radix_preload(void) { return clo_radixw_preload(); }

size_t clo_hip_radix_seg_workspace_bytes(size_t numel, int nseg, int elem_size, int digit_bits) {
	return clo_radix4_seg_workspace_bytes(numel, nseg, elem_size, digit_bits);
}

int clo_hip_radix_sort_segmented(const void* src, void* a, void* b, size_t numel, const size_t* seg_counts, int nseg,
	const size_t* piece_counts, const size_t* piece_offsets, const int* piece_segment, int npieces,
	int elem_size, int key_shift, int key_bits, int digit_bits, void* workspace, size_t workspace_bytes, void* stream, int* result_in_b) {
	return clo_hip_radix_sort_segmented2(src, nullptr, a, b, numel, seg_counts, nseg, piece_counts, piece_offsets, piece_segment, nullptr, npieces,
		elem_size, key_shift, key_bits, digit_bits, workspace, workspace_bytes, stream, result_in_b);
}

int clo_hip_radix_sort_segmented2(const void* src, const void* src2, void* a, void* b, size_t numel, const size_t* seg_counts, int nseg,
	const size_t* piece_counts, const size_t* piece_offsets, const int* piece_segment, const int* piece_source, int npieces,
	int elem_size, int key_shift, int key_bits, int digit_bits, void* workspace, size_t workspace_bytes, void* stream, int* result_in_b) {
	if (!result_in_b) return CLO_HIP_EARGS;
	*result_in_b = 0;
	if (numel == 0) return 0;
	if (!src || !a || !b || a == b || src == b || !workspace || !seg_counts || nseg < 1 || nseg > CLO_SEG_MAX) return CLO_HIP_EARGS;
	if (npieces < 0 || npieces > CLO_SEG_MAX || (npieces > 0 && (!piece_counts || !piece_offsets || !piece_segment))) return CLO_HIP_EARGS;
	if (elem_size != 4 && elem_size != 8) return CLO_HIP_EUNSUPPORTED;
	if (digit_bits != 4 && digit_bits != 8) return CLO_HIP_EUNSUPPORTED;
	if (key_bits < 1 || key_shift < 0 || key_shift + key_bits > 8 * elem_size) return CLO_HIP_EARGS;
	if (numel > 0xffffffffull) return CLO_HIP_EARGS;
	size_t total = 0;
	for (int i = 0; i < nseg; ++i) total += seg_counts[i];
	if (total != numel) return CLO_HIP_EARGS;
	if (npieces > 0) {   // the pieces of a segment add up to it (where they lie in `src` is the caller's business: 32-bit element offsets)
		size_t per[CLO_SEG_MAX];
		for (int k = 0; k < nseg; ++k) per[k] = 0;
		for (int i = 0; i < npieces; ++i) {
			if (piece_segment[i] < 0 || piece_segment[i] >= nseg || piece_offsets[i] > 0xffffffffull || piece_counts[i] > numel) return CLO_HIP_EARGS;
			if (piece_source && piece_source[i] != 0 && (piece_source[i] != 1 || !src2 || src2 == b)) return CLO_HIP_EARGS;   // (a piece of the second source needs one; the first pass writes b)
			per[piece_segment[i]] += piece_counts[i];
		}
		for (int k = 0; k < nseg; ++k) if (per[k] != seg_counts[k]) return CLO_HIP_EARGS;
	}
	const size_t need = clo_radix4_seg_workspace_bytes(numel, nseg, elem_size, digit_bits);
	if (need == 0) return CLO_HIP_EUNSUPPORTED;
	if (workspace_bytes < need) return CLO_HIP_EWORKSPACE;
	return clo_radix4_sort_segmented(src, src2, a, b, numel, seg_counts, nseg, piece_counts, piece_offsets, piece_segment, npieces > 0 ? piece_source : nullptr, npieces,
		elem_size, key_shift, key_bits, digit_bits, workspace, (hipStream_t) stream, result_in_b);
}

size_t clo_hip_msd_workspace_bytes(size_t numel, int elem_size, int bucket_bits) {
	if (bucket_bits < 1 || bucket_bits > 8) return 0;
	return clo_radix4_partition_workspace_bytes(numel, elem_size, bucket_bits);
}

int clo_hip_msd_histogram(const void* src, size_t numel, int elem_size,
	int key_shift, int key_bits, int bucket_bits, uint64_t* counts_dev, void* stream) {
	if (!counts_dev || bucket_bits < 1 || bucket_bits > 3 || bucket_bits > key_bits) return CLO_HIP_EARGS;
	hipStream_t s = (hipStream_t) stream;
	hipError_t e = hipMemsetAsync(counts_dev, 0, sizeof(uint64_t) << bucket_bits, s);
	if (e != hipSuccess) return (int) e;
	if (numel == 0) return 0;
	if (!src) return CLO_HIP_EARGS;
	const unsigned shift = (unsigned) (key_shift + key_bits - bucket_bits), mask = (1u << bucket_bits) - 1u;
	switch (elem_size) {
		case 4: return msd_hist_impl<uint32_t>(src, numel, shift, mask, counts_dev, s);
		case 8: return msd_hist_impl<uint64_t>(src, numel, shift, mask, counts_dev, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

int clo_hip_msd_partition(const void* src, void* dst, size_t numel, int elem_size,
	int key_shift, int key_bits, int bucket_bits, uint64_t* counts_dev,
	void* workspace, size_t workspace_bytes, void* stream) {
	hipStream_t s = (hipStream_t) stream;
	if (bucket_bits < 1 || bucket_bits > 8 || bucket_bits > key_bits) return CLO_HIP_EARGS;
	if (numel == 0) {
		if (counts_dev) return (int) hipMemsetAsync(counts_dev, 0, sizeof(uint64_t) << bucket_bits, s);
		return 0;
	}
	if (!src || !dst || src == dst || !workspace) return CLO_HIP_EARGS;
	if (numel > 0xffffffffull) return CLO_HIP_EARGS;
	if (elem_size != 4 && elem_size != 8) return CLO_HIP_EUNSUPPORTED;
	if (workspace_bytes < clo_hip_msd_workspace_bytes(numel, elem_size, bucket_bits)) return CLO_HIP_EWORKSPACE;
	const unsigned shift = (unsigned) (key_shift + key_bits - bucket_bits);
	// one chain-free radix pass on the top bits (clo_hip_radix4.hip)
	return clo_radix4_partition(src, dst, numel, elem_size, shift, bucket_bits,
		(unsigned long long*) counts_dev, workspace, s);
}

}  // extern "C"
