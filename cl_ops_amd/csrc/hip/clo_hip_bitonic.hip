// clo_hip_bitonic.hip — bitonic sorting networks for gfx950 ("sbitonic" and
// "abitonic" replacements).
//
// The network is the reference's: stage S = 1..T, step p = S..1, pair (i1,i2)
// with i2 = i1 + 2^(p-1), direction bit dir = (i1 >> S) & 1, and
//     swap <=> COMPARE(key[i1], key[i2]) XOR dir
// (sort/clo_sort_sbitonic.cl:45-67, sort/clo_sort_abitonic.cl:31-38,585-601;
// dir is the reference's (gid >> (stage-1)) & 1 expressed on the element index).
// Layers are applied in the same order with the same rule, so the result is
// bit-identical to the reference for any schedule, ties included.
//
// Schedules:
//  * clo_hip_bitonic_simple — one launch per (stage, step), global memory only:
//    what sort/clo_sort_sbitonic.c:102-118 does.
//  * clo_hip_bitonic_tiled  — replaces the 26-kernel strategy of
//    sort/clo_sort_abitonic.c:58-313 (kernels in clo_hip_bitonic_impl.h, one
//    translation unit per element size: clo_hip_bitonic_e{1,2,4,8}.hip):
//      - tile kernels: a work-group owns 2^KL consecutive elements in LDS
//        (KL = 14, 13 for 8-byte elements) and runs every step p <= KL of a
//        stage (the merge kernel) or all of stages 1..KL (the presort). Each
//        thread keeps 2^Q elements in VGPRs and runs up to Q consecutive steps
//        on them between two LDS exchanges; the schedule is fixed at compile
//        time and the LDS image is padded by one slot per 32 so that every
//        exchange pattern is bank-conflict free.
//      - bitonic_strided: steps p > KL; each thread loads 2^NS elements
//        2^(p-NS) apart (adjacent lanes = adjacent addresses, so every access
//        is a full coalesced row), runs NS <= 6 steps in VGPRs, stores them back.
//    For 2^26 4-byte keys that is 31 passes over the array instead of the
//    reference's 58-62.
// Whole-element integer and floating-point keys run bare min/max networks (the
// latter on their order-preserving unsigned image); any other key the general
// compare with keys carried through a register network.
#include <hip/hip_runtime.h>

#include <string>
#include <type_traits>

#include "clo_hip.h"
#include "clo_hip_internal.h"

#include "clo_hip_bitonic_impl.h"

extern "C" {

int clo_hip_gselect(const void* src, void* dst, size_t numel, int elem_size,
	int key_shift, int key_bits, int key_size, int key_kind, int descending, void* stream) {
	if (numel == 0) return 0;
	if (!src || !dst || src == dst) return CLO_HIP_EARGS;
	if (numel > 0xffffffffull) return CLO_HIP_EARGS;
	hipStream_t s = (hipStream_t) stream;
	switch (elem_size) {
		case 1: return gselect_impl<uint8_t>(src, dst, numel, key_shift, key_bits, key_size, key_kind, descending, s);
		case 2: return gselect_impl<uint16_t>(src, dst, numel, key_shift, key_bits, key_size, key_kind, descending, s);
		case 4: return gselect_impl<uint32_t>(src, dst, numel, key_shift, key_bits, key_size, key_kind, descending, s);
		case 8: return gselect_impl<uint64_t>(src, dst, numel, key_shift, key_bits, key_size, key_kind, descending, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

size_t clo_hip_bitonic_padded_numel(size_t numel) { return nlpo2(numel ? numel : 1); }

int clo_hip_bitonic_simple(void* data, size_t numel, int elem_size,
	int key_shift, int key_bits, int key_size, int key_kind, int descending, int* launches, void* stream) {
	if (launches) *launches = 0;
	if (numel <= 1) return 0;
	if (!data) return CLO_HIP_EARGS;
	hipStream_t s = (hipStream_t) stream;
	switch (elem_size) {
		case 1: return simple_impl<uint8_t>(data, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
		case 2: return simple_impl<uint16_t>(data, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
		case 4: return simple_impl<uint32_t>(data, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
		case 8: return simple_impl<uint64_t>(data, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

int clo_hip_bitonic_any(void* data, size_t numel, int elem_size,
	int key_shift, int key_bits, int key_size, int key_kind, int descending, int* launches, void* stream) {
	if (launches) *launches = 0;
	if (numel <= 1) return 0;
	if (!data) return CLO_HIP_EARGS;
	hipStream_t s = (hipStream_t) stream;
	switch (elem_size) {
		case 1: return any_impl<uint8_t>(data, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
		case 2: return any_impl<uint16_t>(data, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
		case 4: return any_impl<uint32_t>(data, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
		case 8: return any_impl<uint64_t>(data, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

int clo_hip_bitonic_tiled(void* data, size_t numel, int elem_size,
	int key_shift, int key_bits, int key_size, int key_kind, int descending, int* launches, void* stream) {
	if (launches) *launches = 0;
	if (numel <= 1) return 0;
	if (!data) return CLO_HIP_EARGS;
	hipStream_t s = (hipStream_t) stream;
	switch (elem_size) {
		// (one translation unit per element size: clo_hip_bitonic_e{1,2,4,8}.hip)
		case 1: return clo_bitonic_tiled_e1(data, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
		case 2: return clo_bitonic_tiled_e2(data, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
		case 4: return clo_bitonic_tiled_e4(data, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
		case 8: return clo_bitonic_tiled_e8(data, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
		default: return CLO_HIP_EUNSUPPORTED;
	}
}

size_t clo_hip_bitonic_lds_bytes(size_t numel, int elem_size, int tiled) {
	if (!tiled) return 0;   // one launch per (stage, step): registers only
	switch (elem_size) {
		case 1: return tiled_lds_bytes<uint8_t>(numel);
		case 2: return tiled_lds_bytes<uint16_t>(numel);
		case 4: return tiled_lds_bytes<uint32_t>(numel);
		case 8: return tiled_lds_bytes<uint64_t>(numel);
		default: return 0;
	}
}

size_t clo_hip_kernel_lds_bytes(const char* family, int elem_size, int param) {
	if (!family) return 0;
	const std::string f(family);
	if (f == "bitonic_tile") {
		const size_t v = elem_size == 8 ? 16 : 32;   // 512-thread groups
		return (512 * v + 512 * v / 32) * (size_t) elem_size;
	}
	if (f == "radix_hist" || f == "radix_pass") {
		const int bits = param < 1 ? 1 : (param > 8 ? 8 : param);
		return f == "radix_hist" ? clo_radixw_lds_bytes(bits <= 4 ? 2 * bits : bits) : clo_radix4_lds_bytes(elem_size, bits);
	}
	if (f == "gselect") return GSEL_STAGE * sizeof(unsigned long long);
	if (f == "scan") return (2 * 8 * 16 + 1) * (size_t) (param > 4 ? 8 : 4) + 4;   // 1024-thread shape, 8 rows
	return 0;  // bitonic_strided, bitonic_step: registers only
}

}  // extern "C"
