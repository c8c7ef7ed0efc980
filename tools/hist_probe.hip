// Developer probe (not part of the product): shapes for the per-tile 256-bin histogram over
// the digit stream (one byte per element, tiles of 16 384 bytes) — which of load latency,
// LDS add rate or same-address serialisation bounds clo_radixw_tilehist_bytes_kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hist_probe.hip -o gpurun_out/hist_probe
//   ./hist_probe [log2 n = 28]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int TILE = 16384;
typedef unsigned vec4u __attribute__((ext_vector_type(4)));

__global__ void fill_kernel(unsigned* p, size_t words, unsigned mode) {
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t) gridDim.x * blockDim.x) {
		unsigned x = (unsigned) i * 2654435761u + 12345u;
		x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
		if (mode == 1) x = 0x07070707u;                       // all equal
		if (mode == 2) x &= 0x07070707u;                      // 8 distinct values
		p[i] = x;
	}
}
__global__ void flush_kernel(vec4u* p, size_t n) {
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = vec4u{ 1u, 2u, 3u, (unsigned) i };
}
// HIST_PROBE_CLEAN=1: the caches are filled by READING the junk — nothing dirty is left for the histogram's reads to push out
__global__ void flush_read_kernel(const vec4u* p, size_t n, unsigned* sink) {
	unsigned acc = 0;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) { const vec4u x = p[i]; acc += x[0] ^ x[3]; }
	if (acc == 0x1234567u) *sink = acc;
}

// MODE 0: 32 copies of dword counters (the shipped kernel); 1: 64 copies of packed 16-bit pairs (lane-private);
// 2: 32 copies packed; 3: loads only; 4: adds only (synthetic digits)
template <int THREADS, int MODE, int TPW>   // TPW tiles per work-group, the next tile's bytes requested before the current tile is counted
__global__ __launch_bounds__(THREADS)
void hist_kernel(const unsigned char* __restrict__ dig, unsigned tiles, unsigned* __restrict__ thist) {
	constexpr int BYTES = TILE / THREADS;   // per thread
	constexpr int VECS = BYTES / 16;
	constexpr bool PACKED = MODE == 1 || MODE == 2;
	constexpr int COPIES = MODE == 1 ? 64 : 32;
	constexpr int WORDS = PACKED ? 128 * COPIES : 256 * COPIES;
	__shared__ __attribute__((aligned(16))) unsigned s_cnt[WORDS];
	const unsigned tid = threadIdx.x, lane = tid & 63u;
	const unsigned cp = lane & (COPIES - 1);
	vec4u v[VECS], nv[VECS];
	unsigned tile = blockIdx.x * TPW;
	if (tile >= tiles) return;
	if (MODE != 4) {
		#pragma unroll
		for (int k = 0; k < VECS; ++k) v[k] = reinterpret_cast<const vec4u*>(dig + (size_t) tile * TILE)[k * THREADS + tid];
	}
	for (int t = 0; t < TPW && tile < tiles; ++t, ++tile) {
		if (MODE != 4 && t + 1 < TPW && tile + 1 < tiles) {
			#pragma unroll
			for (int k = 0; k < VECS; ++k) nv[k] = reinterpret_cast<const vec4u*>(dig + (size_t) (tile + 1) * TILE)[k * THREADS + tid];
		}
		for (unsigned i = tid; i < (unsigned) (WORDS / 4); i += THREADS) reinterpret_cast<vec4u*>(s_cnt)[i] = vec4u{ 0u, 0u, 0u, 0u };
		__syncthreads();
		if (MODE == 4) {
			#pragma unroll
			for (int k = 0; k < VECS; ++k) {
				unsigned x = (tid * 16u + k + tile) * 2654435761u;
				v[k] = vec4u{ x, x * 7u + 1u, x ^ (x >> 7), x * 0x9e3779b9u };
			}
		}
		if (MODE == 3) {
			unsigned acc = 0;
			#pragma unroll
			for (int k = 0; k < VECS; ++k) acc += v[k][0] ^ v[k][1] ^ v[k][2] ^ v[k][3];
			if (acc == 0x12345u) s_cnt[tid] = acc;
		} else {
			#pragma unroll
			for (int k = 0; k < VECS; ++k) {
				#pragma unroll
				for (int q = 0; q < 4; ++q) {
					#pragma unroll
					for (int b = 0; b < 4; ++b) {
						const unsigned d = (v[k][q] >> (8 * b)) & 255u;
						if (PACKED) atomicAdd(&s_cnt[(d >> 1) * COPIES + cp], (d & 1u) ? 0x10000u : 1u);
						else atomicAdd(&s_cnt[(d << 5) + cp], 1u);
					}
				}
			}
		}
		__syncthreads();
		if (PACKED) {
			for (unsigned r = tid; r < 128u; r += THREADS) {
				const vec4u* row = reinterpret_cast<const vec4u*>(&s_cnt[r * COPIES]);
				unsigned h = 0;
				#pragma unroll
				for (int k = 0; k < COPIES / 4; ++k) {
					const vec4u x = row[(k + r) & (COPIES / 4 - 1)];
					h += x[0] + x[1] + x[2] + x[3];   // (each half <= 16 384: no carry between the halves)
				}
				reinterpret_cast<uint2*>(thist + (size_t) tile * 256)[r] = uint2{ h & 0xffffu, h >> 16 };
			}
		} else {
			for (unsigned d = tid; d < 256u; d += THREADS) {
				const vec4u* row = reinterpret_cast<const vec4u*>(&s_cnt[d * COPIES]);
				unsigned h = 0;
				#pragma unroll
				for (int k = 0; k < COPIES / 4; ++k) {
					const vec4u x = row[(k + d) & (COPIES / 4 - 1)];
					h += x[0] + x[1] + x[2] + x[3];
				}
				thist[(size_t) tile * 256 + d] = h;
			}
		}
		if (t + 1 < TPW) {
			__syncthreads();
			#pragma unroll
			for (int k = 0; k < VECS; ++k) v[k] = nv[k];
		}
	}
}

// Straight-line pipeline (round 3, second look): TPW consecutive tiles per work-group, the bytes of tile t + DIST requested
// before tile t is counted, every load unconditional (addresses clamped) and the loop fully unrolled, so that the
// compiler can count what is outstanding (the TPW variants above end up behind s_waitcnt vmcnt(0): their prefetch sits
// in a branch). TWO: two counter arrays, the next one zeroed while this one is reduced (one barrier less per tile).
template <int TPW, int DIST, bool TWO>
__global__ __launch_bounds__(1024)
void hist_pipe_kernel(const unsigned char* __restrict__ dig, unsigned tiles, unsigned* __restrict__ thist) {
	constexpr int THREADS = 1024, COPIES = 32, WORDS = 256 * COPIES;
	__shared__ __attribute__((aligned(16))) unsigned s_cnt[TWO ? 2 * WORDS : WORDS];
	const unsigned tid = threadIdx.x, lane = tid & 63u, cp = lane & (COPIES - 1);
	const unsigned tile0 = blockIdx.x * TPW;
	vec4u buf[DIST + 1];
	#pragma unroll
	for (int t = 0; t < DIST; ++t) {
		const unsigned tl = tile0 + t < tiles ? tile0 + t : tiles - 1;
		buf[t] = reinterpret_cast<const vec4u*>(dig + (size_t) tl * TILE)[tid];
	}
	if (TWO) {
		for (unsigned i = tid; i < (unsigned) (WORDS / 4); i += THREADS) reinterpret_cast<vec4u*>(s_cnt)[i] = vec4u{ 0u, 0u, 0u, 0u };
	}
	#pragma unroll
	for (int t = 0; t < TPW; ++t) {
		unsigned* cnt = s_cnt + (TWO ? (t & 1) * WORDS : 0);
		if (t + DIST < TPW) {
			const unsigned tl = tile0 + t + DIST < tiles ? tile0 + t + DIST : tiles - 1;
			buf[(t + DIST) % (DIST + 1)] = reinterpret_cast<const vec4u*>(dig + (size_t) tl * TILE)[tid];
		}
		if (!TWO) {
			for (unsigned i = tid; i < (unsigned) (WORDS / 4); i += THREADS) reinterpret_cast<vec4u*>(cnt)[i] = vec4u{ 0u, 0u, 0u, 0u };
		}
		__syncthreads();
		const vec4u v = buf[t % (DIST + 1)];
		#pragma unroll
		for (int q = 0; q < 4; ++q) {
			#pragma unroll
			for (int b = 0; b < 4; ++b) atomicAdd(&cnt[(((v[q] >> (8 * b)) & 255u) << 5) + cp], 1u);
		}
		__syncthreads();
		if (TWO && t + 1 < TPW) {   // the other array, for the next tile (its last reader finished before the barrier above)
			unsigned* nxt = s_cnt + ((t + 1) & 1) * WORDS;
			for (unsigned i = tid; i < (unsigned) (WORDS / 4); i += THREADS) reinterpret_cast<vec4u*>(nxt)[i] = vec4u{ 0u, 0u, 0u, 0u };
		}
		if (tid < 256u && tile0 + t < tiles) {
			const vec4u* row = reinterpret_cast<const vec4u*>(&cnt[tid * COPIES]);
			unsigned h = 0;
			#pragma unroll
			for (int k = 0; k < COPIES / 4; ++k) {
				const vec4u x = row[(k + tid) & (COPIES / 4 - 1)];
				h += x[0] + x[1] + x[2] + x[3];
			}
			thist[(size_t) (tile0 + t) * 256 + tid] = h;
		}
		if (!TWO && t + 1 < TPW) __syncthreads();
	}
}
template <int TPW, int DIST, bool TWO>
void launch_pipe(const unsigned char* dig, unsigned tiles, unsigned* thist, hipStream_t s) {
	hipLaunchKernelGGL((hist_pipe_kernel<TPW, DIST, TWO>), dim3((tiles + TPW - 1) / TPW), dim3(1024), 0, s, dig, tiles, thist);
}

// Wave-private counting without LDS atomics where a wave's lanes agree: lanes that hold the same byte value are
// found with one pass of v_cmp per DISTINCT value (skewed inputs); kept out: uniform bytes have ~60 distinct per 64.

struct variant { const char* name; void (*launch)(const unsigned char*, unsigned, unsigned*, hipStream_t); };
template <int THREADS, int MODE, int TPW>
void launch(const unsigned char* dig, unsigned tiles, unsigned* thist, hipStream_t s) {
	hipLaunchKernelGGL((hist_kernel<THREADS, MODE, TPW>), dim3((tiles + TPW - 1) / TPW), dim3(THREADS), 0, s, dig, tiles, thist);
}

int main(int argc, char** argv) {
	const int logn = argc > 1 ? atoi(argv[1]) : 28;
	const size_t n = (size_t) 1 << logn;
	const unsigned tiles = (unsigned) (n / TILE);
	unsigned char* dig; unsigned *thist, *ref; vec4u* junk;
	const size_t junk_bytes = (size_t) 1 << 30;
	CK(hipMalloc(&dig, n)); CK(hipMalloc(&thist, (size_t) tiles * 1024)); CK(hipMalloc(&ref, (size_t) tiles * 1024)); CK(hipMalloc(&junk, junk_bytes));
	hipStream_t s; CK(hipStreamCreate(&s));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const variant vs[] = {
		{ "1024x16 u32x32 (shipped)", launch<1024, 0, 1> },
		{ "pipe 2 tiles/wg, 1 ahead", launch_pipe<2, 1, false> },
		{ "pipe 4 tiles/wg, 1 ahead", launch_pipe<4, 1, false> },
		{ "pipe 4 tiles/wg, 2 ahead", launch_pipe<4, 2, false> },
		{ "pipe 8 tiles/wg, 1 ahead", launch_pipe<8, 1, false> },
		{ "pipe 8 tiles/wg, 2 ahead", launch_pipe<8, 2, false> },
		{ "pipe 8 tiles/wg, 3 ahead", launch_pipe<8, 3, false> },
		{ "pipe 16 tiles/wg, 2 ahead", launch_pipe<16, 2, false> },
		{ "pipe 4 tiles/wg, 2 ahead, 2 arrays", launch_pipe<4, 2, true> },
		{ "pipe 8 tiles/wg, 2 ahead, 2 arrays", launch_pipe<8, 2, true> },
		{ "pipe 16 tiles/wg, 2 ahead, 2 arrays", launch_pipe<16, 2, true> },
		{ "1024x16 packed16x64", launch<1024, 1, 1> },
		{ "1024x16 packed16x32", launch<1024, 2, 1> },
		{ "1024x16 loads only", launch<1024, 3, 1> },
		{ "1024x16 adds only", launch<1024, 4, 1> },
		{ "1024x16 u32x32, 2 tiles/wg", launch<1024, 0, 2> },
		{ "1024x16 packed16x64, 2 tiles/wg", launch<1024, 1, 2> },
		{ "1024x16 packed16x64, 4 tiles/wg", launch<1024, 1, 4> },
		{ "512x32 u32x32", launch<512, 0, 1> },
		{ "512x32 packed16x64", launch<512, 1, 1> },
		{ "512x32 packed16x32", launch<512, 2, 1> },
		{ "512x32 loads only", launch<512, 3, 1> },
		{ "512x32 adds only", launch<512, 4, 1> },
		{ "512x32 packed16x64, 2 tiles/wg", launch<512, 1, 2> },
		{ "256x64 u32x32", launch<256, 0, 1> },
		{ "256x64 packed16x64", launch<256, 1, 1> },
		{ "256x64 packed16x32", launch<256, 2, 1> },
		{ "256x64 loads only", launch<256, 3, 1> },
		{ "256x64 adds only", launch<256, 4, 1> },
		{ "256x64 packed16x32, 2 tiles/wg", launch<256, 2, 2> },
	};
	const bool clean = getenv("HIST_PROBE_CLEAN") != nullptr;
	if (clean) hipLaunchKernelGGL(flush_kernel, dim3(4096), dim3(256), 0, s, junk, junk_bytes / 16);
	std::vector<unsigned> h_ref((size_t) tiles * 256), h_got((size_t) tiles * 256);
	for (unsigned mode = 0; mode < 3; ++mode) {
		hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, s, (unsigned*) dig, n / 4, mode);
		printf("== 2^%d bytes, %u tiles, input %s ==\n", logn, tiles, mode == 0 ? "uniform" : (mode == 1 ? "all equal" : "8 distinct values"));
		for (size_t i = 0; i < sizeof(vs) / sizeof(vs[0]); ++i) {
			float best = 1e9f, sum = 0;
			const int reps = 6;
			CK(hipMemsetAsync(thist, 0xff, (size_t) tiles * 1024, s));
			for (int r = 0; r < reps; ++r) {
				if (clean) hipLaunchKernelGGL(flush_read_kernel, dim3(4096), dim3(256), 0, s, (const vec4u*) junk, junk_bytes / 16, ref);
				else hipLaunchKernelGGL(flush_kernel, dim3(4096), dim3(256), 0, s, junk, junk_bytes / 16);   // what a pass kernel leaves in the caches: not the stream, and dirty
				CK(hipEventRecord(e0, s));
				vs[i].launch(dig, tiles, thist, s);
				CK(hipEventRecord(e1, s));
				CK(hipEventSynchronize(e1));
				float ms; CK(hipEventElapsedTime(&ms, e0, e1));
				if (ms < best) best = ms;
				sum += ms;
			}
			CK(hipGetLastError());
			const bool counts = strstr(vs[i].name, "only") == nullptr;
			const char* ok = "-";
			if (counts) {
				CK(hipMemcpy(h_got.data(), thist, h_got.size() * 4, hipMemcpyDeviceToHost));
				if (i == 0) { h_ref = h_got; ok = "ref"; }
				else ok = memcmp(h_ref.data(), h_got.data(), h_got.size() * 4) == 0 ? "same" : "DIFFERENT";
			}
			printf("%-36s best %.4f ms  mean %.4f ms  %.2f TB/s  %s\n", vs[i].name, best, sum / reps, (double) n / (best * 1e-3) / 1e12, ok);
			fflush(stdout);
		}
	}
	return 0;
}
