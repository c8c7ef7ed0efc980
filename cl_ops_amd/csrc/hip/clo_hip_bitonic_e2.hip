// clo_hip_bitonic_e2.hip — the tiled bitonic schedule (abitonic) for 2-byte
// elements: see clo_hip_bitonic_impl.h / clo_hip_bitonic.hip.
#include "clo_hip_bitonic_impl.h"

int clo_bitonic_tiled_e2(void* data, size_t numel, int key_shift, int key_bits, int key_size, int key_kind, int descending,
	int* launches, hipStream_t s) {
	return tiled_impl<uint16_t>(data, numel, key_shift, key_bits, key_size, key_kind, descending, launches, s);
}
