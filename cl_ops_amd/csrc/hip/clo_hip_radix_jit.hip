// clo_hip_radix_jit.hip — satradix for `get_key` expressions outside the
// ahead-of-time family (shifts, masks, casts).
//
// Upstream pastes CLO_SORT_KEY_GET(x) into its OpenCL C kernels and JIT-builds
// them (sort/clo_sort_abstract.c:144-179), so any expression works with any
// sorter. The HIP radix passes need the key as a bit field of the element; for
// every other expression the key is MATERIALISED once:
//   1. a kernel compiled with hiprtc (the user's macro body pasted in, as
//      upstream does) writes  pair[i] = ordered_bits(get_key(x[i])) << 32 | i
//      — ordered_bits = the order-preserving unsigned image of the typed key;
//   2. the pairs are sorted by their high word with the ordinary radix passes
//      (stable; the index in the low word is carried along);
//   3. a gather kernel writes out[j] = x[pair[j] & 0xffffffff].
// The pair is 64 bits, so a key of 8 bytes takes two rounds, least significant
// half first: round one sorts (low key half, index); a second compiled kernel
// then builds (high key half of x[index], index) IN THAT ORDER, and the stable
// sort of those by the high half finishes the order. numel < 2^32 as everywhere.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "clo_hip.h"
#include "clo_hip_internal.h"
#include "clo_hip_jit_opts.h"

namespace {

const char* k_extract_src = R"CLOJIT(
typedef unsigned char uchar;
typedef unsigned short ushort;
typedef unsigned int uint;
typedef unsigned long ulong;
typedef CLO_SORT_ELEM_TYPE E;
typedef CLO_SORT_KEY_TYPE K;
typedef CLO_SORT_KEY_BITS_TYPE UK;   // unsigned integer of the key's size

__device__ __forceinline__ UK jit_ordered_key(const E x) {
	const K k = (K) (CLO_SORT_KEY_GET_X(x));
	UK u;
	__builtin_memcpy(&u, &k, sizeof(k));
	const UK sign = (UK) 1 << (8 * sizeof(K) - 1);
#if CLO_SORT_KEY_KIND == 1
	u ^= sign;                                   // two's complement
#elif CLO_SORT_KEY_KIND == 2
	u = (u & sign) ? (UK) ~u : (UK) (u | sign);  // IEEE-754
#endif
	return u;
}

// pairs[i] = (low 32 bits of the ordered key, i)  — the whole key when it has <= 32 bits;
// dig (optional): the first radix pass's digit of pair i, one byte (clo_hip_radix_sort_fed)
extern "C" __global__ __launch_bounds__(256)
void jit_extract(const E* __restrict__ in, unsigned long long* __restrict__ pairs, unsigned long n, unsigned char* __restrict__ dig) {
	const unsigned long i = (unsigned long) blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const unsigned long long u = (unsigned long long) jit_ordered_key(in[i]);
	pairs[i] = (u << 32) | (unsigned long long) (unsigned) i;
	if (dig) dig[i] = (unsigned char) u;
}

// 8-byte keys, second round: out[i] = (high 32 bits of the ordered key of x[j], j), j = index in pairs[i]
extern "C" __global__ __launch_bounds__(256)
void jit_extract_hi(const E* __restrict__ in, const unsigned long long* __restrict__ pairs,
	unsigned long long* __restrict__ out, unsigned long n) {
	const unsigned long i = (unsigned long) blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const unsigned j = (unsigned) pairs[i];
	const unsigned long long u = (unsigned long long) jit_ordered_key(in[(unsigned long) j < n ? j : 0]);
	out[i] = (u & 0xffffffff00000000ull) | (unsigned long long) j;
}
)CLOJIT";

struct radix_jit {
	hipModule_t module = nullptr;
	hipFunction_t extract = nullptr, extract_hi = nullptr;
	int elem_size = 0, key_size = 0;
};

template <typename E>
__global__ __launch_bounds__(256)
void clo_radix_gather_kernel(const E* __restrict__ in, const unsigned long long* __restrict__ pairs, E* __restrict__ out, size_t n) {
	const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
	if (i < n) {
		const unsigned j = (unsigned) pairs[i];
		if ((size_t) j < n) out[i] = in[j];
	}
}

const char* ctype_of(int clo_type) {
	static const char* names[] = { "signed char", "unsigned char", "short", "unsigned short", "int", "unsigned int",
		"long", "unsigned long", "_Float16", "float", "double" };
	return (clo_type >= 0 && clo_type <= 10) ? names[clo_type] : nullptr;
}
int type_size(int clo_type) {
	static const int sizes[] = { 1, 1, 2, 2, 4, 4, 8, 8, 2, 4, 8 };
	return (clo_type >= 0 && clo_type <= 10) ? sizes[clo_type] : 0;
}
int type_kind(int clo_type) {   // 0 unsigned, 1 signed, 2 floating point
	static const int kinds[] = { 1, 0, 1, 0, 1, 0, 1, 0, 2, 2, 2 };
	return (clo_type >= 0 && clo_type <= 10) ? kinds[clo_type] : 0;
}
void set_log(char** log, const std::string& text) {
	if (!log) return;
	*log = (char*) malloc(text.size() + 1);
	if (*log) memcpy(*log, text.c_str(), text.size() + 1);
}

}  // namespace

extern "C" {

int clo_hip_radix_jit_create(int elem_type, int key_type, const char* get_key, const char* compiler_opts, void** handle, char** log) {
	if (log) *log = nullptr;
	if (!handle) return CLO_HIP_EARGS;
	*handle = nullptr;
	const char* et = ctype_of(elem_type);
	const char* kt = ctype_of(key_type);
	if (!et || !kt) return CLO_HIP_EUNSUPPORTED;
	const int ks = type_size(key_type);
	const char* uk = ks == 1 ? "unsigned char" : (ks == 2 ? "unsigned short" : (ks == 4 ? "unsigned int" : "unsigned long"));

	std::string src;
	src += std::string("#define CLO_SORT_ELEM_TYPE ") + et + "\n";
	src += std::string("#define CLO_SORT_KEY_TYPE ") + kt + "\n";
	src += std::string("#define CLO_SORT_KEY_BITS_TYPE ") + uk + "\n";
	src += "#define CLO_SORT_KEY_KIND " + std::to_string(type_kind(key_type)) + "\n";
	src += std::string("#define CLO_SORT_KEY_GET_X(x) ") + (get_key ? get_key : "(x)") + "\n";
	src += k_extract_src;

	hiprtcProgram prog = nullptr;
	if (hiprtcCreateProgram(&prog, src.c_str(), "clo_sort_satradix_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
		set_log(log, "hiprtcCreateProgram failed");
		return CLO_HIP_EUNSUPPORTED;
	}
	const std::vector<std::string> optv = clo_jit_options(compiler_opts);
	std::vector<const char*> opts;
	for (const std::string& o : optv) opts.push_back(o.c_str());
	if (hiprtcCompileProgram(prog, (int) opts.size(), opts.data()) != HIPRTC_SUCCESS) {
		size_t n = 0;
		hiprtcGetProgramLogSize(prog, &n);
		std::string text(n ? n : 1, '\0');
		if (n) hiprtcGetProgramLog(prog, &text[0]);
		set_log(log, text);
		hiprtcDestroyProgram(&prog);
		return CLO_HIP_EARGS;  // the user's expression does not compile
	}
	size_t code_size = 0;
	hiprtcGetCodeSize(prog, &code_size);
	std::vector<char> code(code_size);
	hiprtcGetCode(prog, code.data());
	hiprtcDestroyProgram(&prog);

	radix_jit* rj = new radix_jit();
	rj->elem_size = type_size(elem_type);
	rj->key_size = ks;
	hipError_t e = hipModuleLoadData(&rj->module, code.data());
	if (e == hipSuccess) e = hipModuleGetFunction(&rj->extract, rj->module, "jit_extract");
	if (e == hipSuccess) e = hipModuleGetFunction(&rj->extract_hi, rj->module, "jit_extract_hi");
	if (e != hipSuccess) {
		set_log(log, std::string("loading the compiled module failed: ") + hipGetErrorString(e));
		if (rj->module) (void) hipModuleUnload(rj->module);
		delete rj;
		return (int) e;
	}
	*handle = rj;
	return 0;
}

void clo_hip_radix_jit_destroy(void* handle) {
	radix_jit* rj = (radix_jit*) handle;
	if (!rj) return;
	if (rj->module) (void) hipModuleUnload(rj->module);
	delete rj;
}

int clo_hip_radix_jit_sort(void* handle, const void* src, void* dst, void* pairs, void* pairs_tmp, size_t numel,
	int digit_bits, void* workspace, size_t workspace_bytes, void* stream) {
	radix_jit* rj = (radix_jit*) handle;
	if (numel == 0) return 0;
	if (!rj || !src || !dst || !pairs || !pairs_tmp || pairs == pairs_tmp) return CLO_HIP_EARGS;
	if (numel > 0xffffffffull) return CLO_HIP_EARGS;
	hipStream_t s = (hipStream_t) stream;
	unsigned long n = numel;
	const unsigned blocks = (unsigned) ((numel + 255) / 256);
	// The extractor touches every key anyway: it also writes the first pass's digit of every pair, one byte each,
	// into the ping-pong partner of the pairs (free until the first pass kernel writes it), and the sort's first
	// histogram reads those numel bytes instead of numel 8-byte pairs.
	unsigned char* dig = clo_hip_radix_takes_first_digits(numel, 8, 0, digit_bits) ? (unsigned char*) pairs_tmp : nullptr;
	{
		clo_timing_scope timing("radix_extract", s);
		void* args[] = { (void*) &src, &pairs, &n, &dig };
		const hipError_t e = hipModuleLaunchKernel(rj->extract, blocks, 1, 1, 256, 1, 1, 0, s, args, nullptr);
		if (e != hipSuccess) return (int) e;
	}
	const int low_bits = rj->key_size >= 4 ? 32 : 8 * rj->key_size;
	int st = clo_hip_radix_sort_fed(pairs, pairs, pairs_tmp, numel, 8, 32, low_bits, 0, digit_bits, dig,
		workspace, workspace_bytes, stream);
	if (st != 0) return st;
	void* sorted = pairs;      // where the sorted (key, index) pairs are
	void* spare = pairs_tmp;   // the other pair buffer
	if (rj->key_size == 8) {
		{
			clo_timing_scope timing("radix_extract", s);
			void* args[] = { (void*) &src, &pairs, &pairs_tmp, &n };
			const hipError_t e = hipModuleLaunchKernel(rj->extract_hi, blocks, 1, 1, 256, 1, 1, 0, s, args, nullptr);
			if (e != hipSuccess) return (int) e;
		}
		st = clo_hip_radix_sort(pairs_tmp, pairs_tmp, pairs, numel, 8, 32, 32, 0, digit_bits, workspace, workspace_bytes, stream);
		if (st != 0) return st;
		sorted = pairs_tmp;
		spare = pairs;
	}
	// gather through the spare pair buffer when sorting in place (numel * 8 bytes >= numel * elem_size)
	void* out = dst == src ? spare : dst;
	{
		clo_timing_scope timing("radix_gather", s);
		const unsigned long long* p = (const unsigned long long*) sorted;
		switch (rj->elem_size) {
			case 1: hipLaunchKernelGGL((clo_radix_gather_kernel<uint8_t>), dim3(blocks), dim3(256), 0, s, (const uint8_t*) src, p, (uint8_t*) out, numel); break;
			case 2: hipLaunchKernelGGL((clo_radix_gather_kernel<uint16_t>), dim3(blocks), dim3(256), 0, s, (const uint16_t*) src, p, (uint16_t*) out, numel); break;
			case 4: hipLaunchKernelGGL((clo_radix_gather_kernel<uint32_t>), dim3(blocks), dim3(256), 0, s, (const uint32_t*) src, p, (uint32_t*) out, numel); break;
			case 8: hipLaunchKernelGGL((clo_radix_gather_kernel<uint64_t>), dim3(blocks), dim3(256), 0, s, (const uint64_t*) src, p, (uint64_t*) out, numel); break;
			default: return CLO_HIP_EUNSUPPORTED;
		}
	}
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return (int) e;
	if (dst == src) e = hipMemcpyAsync(dst, out, numel * (size_t) rj->elem_size, hipMemcpyDeviceToDevice, s);
	return (int) e;
}

}  // extern "C"
