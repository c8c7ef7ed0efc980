// clo_hip_bitonic_jit.hip — bitonic sorts specialised at run time for arbitrary
// `compare` / `get_key` expressions.
//
// Upstream specialises its kernels by JIT: the sorter constructor prepends
//   #define CLO_SORT_ELEM_TYPE / CLO_SORT_KEY_TYPE / CLO_SORT_COMPARE(a,b) / CLO_SORT_KEY_GET(x)
// to the OpenCL C source and builds it (sort/clo_sort_abstract.c:144-179). The
// ahead-of-time kernels of clo_hip_bitonic.hip cover the common family (shifts,
// masks, casts; a > b, a < b). Anything else takes this path: the same three
// kernels (one step / strided register network / LDS tile — the schedule of
// clo_hip_bitonic_tiled) are compiled with hiprtc for gfx950 with the user's
// two macro bodies pasted in, exactly as upstream pastes them, and launched
// through the module API. The macro bodies are C expressions over OpenCL's
// scalar type names (uchar, ushort, uint, ulong are typedef'd for them).
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "clo_hip.h"
#include "clo_hip_internal.h"
#include "clo_hip_jit_opts.h"

namespace {

// Device source. E/K are the element/key C types; Q = register bits per thread
// of the tile kernel (5 for <= 4-byte elements, 4 for 8-byte ones).
const char* k_src = R"CLOJIT(
typedef unsigned char uchar;
typedef unsigned short ushort;
typedef unsigned int uint;
typedef unsigned long ulong;
typedef CLO_SORT_ELEM_TYPE E;
typedef CLO_SORT_KEY_TYPE K;

__device__ __forceinline__ void cmpxch(E& e1, E& e2, unsigned dir) {
	const K a = (K) (CLO_SORT_KEY_GET_X(e1));
	const K b = (K) (CLO_SORT_KEY_GET_X(e2));
	const bool cmp = (bool) (CLO_SORT_COMPARE_AB(a, b));
	if (cmp != (bool) dir) { const E t = e1; e1 = e2; e2 = t; }
}

template <int V>
__device__ __forceinline__ void reg_network(E (&v)[V], int nsteps, unsigned long idx0, unsigned b0, unsigned S) {
	const unsigned dbase = (unsigned) ((idx0 >> S) & 1);
	const unsigned dsel = (S >= b0 && S - b0 < 31u) ? ((1u << (S - b0)) & (unsigned) (V - 1)) : 0u;
	#pragma unroll
	for (int half = V / 2; half >= 1; half /= 2) {
		if (half < (1 << nsteps)) {
			#pragma unroll
			for (int j = 0; j < V; ++j)
				if ((j & half) == 0) {
					const unsigned dir = dsel ? (((unsigned) j & dsel) ? 1u : 0u) : dbase;
					cmpxch(v[j], v[j + half], dir);
				}
		}
	}
}

extern "C" __global__ __launch_bounds__(256)
void jit_step(E* __restrict__ data, unsigned long npairs, unsigned stage, unsigned step) {
	const unsigned long gid = (unsigned long) blockIdx.x * 256 + threadIdx.x;
	if (gid >= npairs) return;
	const unsigned sh = step - 1;
	const unsigned long i1 = ((gid >> sh) << (sh + 1)) | (gid & ((1ul << sh) - 1));
	const unsigned long i2 = i1 + (1ul << sh);
	E a = data[i1], b = data[i2];
	cmpxch(a, b, (unsigned) ((i1 >> stage) & 1));
	data[i1] = a;
	data[i2] = b;
}

// Any numel (round 3): the network in its "flip" form — the first step of a stage compares position o of a
// block's first half with position B - 1 - o of the block, every other step is a half-cleaner, and EVERY
// comparator puts the smaller element at the lower index. Elements past numel then behave like +infinity
// that never moves: a comparator whose upper index is >= numel is simply skipped. (Upstream's network with
// its direction bit has no such form and upstream sorts powers of two only, so there is no reference order
// to keep for ties here; for a power of two the direction-bit kernels above are used, as upstream's are.)
extern "C" __global__ __launch_bounds__(256)
void jit_step_any(E* __restrict__ data, unsigned long n, unsigned long npairs, unsigned stage, unsigned step) {
	const unsigned long gid = (unsigned long) blockIdx.x * 256 + threadIdx.x;
	if (gid >= npairs) return;
	const unsigned sh = step - 1;
	const unsigned long i1 = ((gid >> sh) << (sh + 1)) | (gid & ((1ul << sh) - 1));
	const unsigned long i2 = step == stage ? (i1 | ((1ul << stage) - 1ul)) - (i1 & ((1ul << sh) - 1ul)) : i1 + (1ul << sh);
	if (i2 >= n) return;
	E a = data[i1], b = data[i2];
	cmpxch(a, b, 0u);
	data[i1] = a;
	data[i2] = b;
}

template <int NS>
__device__ __forceinline__ void strided_body(E* __restrict__ data, unsigned long n, unsigned stage, unsigned p) {
	constexpr int V = 1 << NS;
	const unsigned long t = (unsigned long) blockIdx.x * 256 + threadIdx.x;
	if (t >= (n >> NS)) return;
	const unsigned b0 = p - NS;
	const unsigned long base = ((t >> b0) << (b0 + NS)) | (t & ((1ul << b0) - 1));
	E v[V];
	#pragma unroll
	for (int j = 0; j < V; ++j) v[j] = data[base + ((unsigned long) j << b0)];
	reg_network<V>(v, NS, base, b0, stage);
	#pragma unroll
	for (int j = 0; j < V; ++j) data[base + ((unsigned long) j << b0)] = v[j];
}
extern "C" __global__ __launch_bounds__(256) void jit_strided1(E* d, unsigned long n, unsigned s, unsigned p) { strided_body<1>(d, n, s, p); }
extern "C" __global__ __launch_bounds__(256) void jit_strided2(E* d, unsigned long n, unsigned s, unsigned p) { strided_body<2>(d, n, s, p); }
extern "C" __global__ __launch_bounds__(256) void jit_strided3(E* d, unsigned long n, unsigned s, unsigned p) { strided_body<3>(d, n, s, p); }
extern "C" __global__ __launch_bounds__(256) void jit_strided4(E* d, unsigned long n, unsigned s, unsigned p) { strided_body<4>(d, n, s, p); }
#if CLO_JIT_Q >= 5
extern "C" __global__ __launch_bounds__(256) void jit_strided5(E* d, unsigned long n, unsigned s, unsigned p) { strided_body<5>(d, n, s, p); }
#endif

// The tile kernels: CLO_JIT_TT = 512 threads on tiles of up to 512 * 2^Q elements (2^14 of up to 4 bytes, 2^13 of 8: 67 KiB
// of LDS, two work-groups per CU) — the shape of the ahead-of-time kernels; round 4's 256 x 2^Q tiles meant one more
// stage, and up to one more strided pass per stage, above the tile.
extern "C" __global__ __launch_bounds__(CLO_JIT_TT)
void jit_tile(E* __restrict__ data, unsigned kl, unsigned stage, unsigned p_hi, int mode) {
	constexpr int Q = CLO_JIT_Q;
	constexpr int V = 1 << Q;
	constexpr int TILE_MAX = CLO_JIT_TT * V;
	__shared__ E s[TILE_MAX + TILE_MAX / 32];
	const unsigned tid = threadIdx.x;
	const unsigned tile = 1u << kl;
	const unsigned nthr = tile >> Q;
	const unsigned long gbase = (unsigned long) blockIdx.x << kl;
	#define PHYS(i) ((i) + ((i) >> 5))
	for (unsigned i = tid; i < tile; i += CLO_JIT_TT) s[PHYS(i)] = data[gbase + i];
	__syncthreads();
	E v[V];
	int cur_b0 = -1;
	unsigned base = 0, pbase = 0;
	const unsigned s_first = mode ? 1u : stage;
	for (unsigned S = s_first; S <= stage; ++S) {
		unsigned p = mode ? S : p_hi;
		while (p >= 1) {
			const unsigned b0 = p > (unsigned) Q ? p - Q : 0u;
			const int nsteps = (int) (p - b0);
			if (cur_b0 != (int) b0) {
				if (cur_b0 >= 0) {
					if (tid < nthr) {
						#pragma unroll
						for (int j = 0; j < V; ++j) s[pbase + PHYS((unsigned) j << cur_b0)] = v[j];
					}
					__syncthreads();
				}
				// (the thread's base and the offsets j << b0 share no bits, so the padded slot of their sum is the sum of their
				// padded slots: one per-thread value per group plus wave-uniform offsets — it was a shift and two adds per access)
				base = ((tid >> b0) << (b0 + Q)) | (tid & ((1u << b0) - 1u));
				pbase = PHYS(base);
				if (tid < nthr) {
					#pragma unroll
					for (int j = 0; j < V; ++j) v[j] = s[pbase + PHYS((unsigned) j << b0)];
				}
				cur_b0 = (int) b0;
			}
			if (tid < nthr) reg_network<V>(v, nsteps, gbase + base, b0, S);
			p = b0;
		}
	}
	if (cur_b0 >= 0 && tid < nthr) {
		#pragma unroll
		for (int j = 0; j < V; ++j) s[pbase + PHYS((unsigned) j << cur_b0)] = v[j];
	}
	__syncthreads();
	for (unsigned i = tid; i < tile; i += CLO_JIT_TT) data[gbase + i] = s[PHYS(i)];
}

// Steps KL .. 1 of a stage above the tile (the merge pass, one per stage: 13 of the 38 launches of a 2^26-key sort) with the
// schedule fixed at COMPILE time (round 5): the groups of register bits are always [KL - Q, KL), [KL - 2Q, KL - Q), ...,
// so every LDS access is the thread's padded base plus an immediate offset, and the direction — one bit of the tile's
// number — is a scalar branch around two straight-line networks. (The run-time schedule above spends more VALU on
// addresses and per-pair directions than on the user's compare.)
template <bool DIR>
__device__ __forceinline__ void cmpxch_u(E& e1, E& e2) {
	const K a = (K) (CLO_SORT_KEY_GET_X(e1));
	const K b = (K) (CLO_SORT_KEY_GET_X(e2));
	const bool cmp = (bool) (CLO_SORT_COMPARE_AB(a, b));
	if (cmp != DIR) { const E t = e1; e1 = e2; e2 = t; }
}
template <int V, int NSTEPS, bool DIR>
__device__ __forceinline__ void reg_network_u(E (&v)[V]) {
	#pragma unroll
	for (int half = V / 2; half >= 1; half /= 2) {
		if (half < (1 << NSTEPS)) {
			#pragma unroll
			for (int j = 0; j < V; ++j)
				if ((j & half) == 0) cmpxch_u<DIR>(v[j], v[j + half]);
		}
	}
}
// one group: the values at register bits [B0, B0 + Q) come out of LDS, NSTEPS steps run on them, they go back
template <int P, bool DIR>
__device__ __forceinline__ void merge_groups(E* s, unsigned tid) {
	constexpr int Q = CLO_JIT_Q, V = 1 << Q;
	constexpr int B0 = P > Q ? P - Q : 0, NSTEPS = P - B0;
	const unsigned base = ((tid >> B0) << (B0 + Q)) | (tid & ((1u << B0) - 1u));
	const unsigned pb = PHYS(base);
	E v[V];
	#pragma unroll
	for (int j = 0; j < V; ++j) v[j] = s[pb + PHYS((unsigned) j << B0)];
	reg_network_u<V, NSTEPS, DIR>(v);
	#pragma unroll
	for (int j = 0; j < V; ++j) s[pb + PHYS((unsigned) j << B0)] = v[j];
	__syncthreads();   // (a thread rewrites exactly the slots it read: one barrier per group)
	if constexpr (B0 > 0) merge_groups<B0, DIR>(s, tid);
}
extern "C" __global__ __launch_bounds__(CLO_JIT_TT, CLO_JIT_MERGE_WPE)
void jit_merge(E* __restrict__ data, unsigned stage) {
	constexpr int Q = CLO_JIT_Q, V = 1 << Q, KL = CLO_JIT_TB + Q, TILE = CLO_JIT_TT * V;
	__shared__ E s[TILE + TILE / 32];
	const unsigned tid = threadIdx.x;
	const unsigned long gbase = (unsigned long) blockIdx.x << KL;
	#pragma unroll
	for (int k = 0; k < V; ++k) { const unsigned i = (unsigned) k * CLO_JIT_TT + tid; s[PHYS(i)] = data[gbase + i]; }
	__syncthreads();
	if ((gbase >> stage) & 1ul) merge_groups<KL, true>(s, tid); else merge_groups<KL, false>(s, tid);
	#pragma unroll
	for (int k = 0; k < V; ++k) { const unsigned i = (unsigned) k * CLO_JIT_TT + tid; data[gbase + i] = s[PHYS(i)]; }
}

// gselect with the user's two macro bodies pasted in, as upstream pastes them
// (sort/clo_sort_gselect.cl:38-58): position = number of elements whose key compares
// "before" mine, ties by index. Keys are staged 2048 at a time in LDS.
extern "C" __global__ __launch_bounds__(256)
void jit_gselect(const E* __restrict__ in, E* __restrict__ out, unsigned long n) {
	__shared__ K s_key[2048];
	const unsigned long gid = (unsigned long) blockIdx.x * 256 + threadIdx.x;
	const E mine = in[gid < n ? gid : 0];
	const K km = (K) (CLO_SORT_KEY_GET_X(mine));
	unsigned long pos = 0;
	for (unsigned long base = 0; base < n; base += 2048) {
		const unsigned cnt = n - base < 2048ul ? (unsigned) (n - base) : 2048u;
		__syncthreads();
		for (unsigned i = threadIdx.x; i < cnt; i += 256) { const E e = in[base + i]; s_key[i] = (K) (CLO_SORT_KEY_GET_X(e)); }
		__syncthreads();
		for (unsigned i = 0; i < cnt; ++i) {
			const K ki = s_key[i];
			if ((bool) (CLO_SORT_COMPARE_AB(km, ki)) || ((ki == km) && (base + i < gid))) ++pos;
		}
	}
	if (gid < n && pos < n) out[pos] = mine;
}
)CLOJIT";

struct jit_sorter {
	hipModule_t module = nullptr;
	hipFunction_t step = nullptr, step_any = nullptr, tile = nullptr, merge = nullptr, gselect = nullptr, strided[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
	int elem_size = 0;
	int q = 0;
	int tb = 0;   // thread bits of the tile kernels
};

const char* ctype_of(int clo_type) {
	// CloType numbering (clo_common.h)
	static const char* names[] = { "signed char", "unsigned char", "short", "unsigned short", "int", "unsigned int",
		"long", "unsigned long", "_Float16", "float", "double" };
	return (clo_type >= 0 && clo_type <= 10) ? names[clo_type] : nullptr;
}

int type_size(int clo_type) {
	static const int sizes[] = { 1, 1, 2, 2, 4, 4, 8, 8, 2, 4, 8 };
	return (clo_type >= 0 && clo_type <= 10) ? sizes[clo_type] : 0;
}

void set_log(char** log, const std::string& text) {
	if (!log) return;
	*log = (char*) malloc(text.size() + 1);
	if (*log) memcpy(*log, text.c_str(), text.size() + 1);
}

constexpr int JIT_TB = 9;   // thread bits of the tile kernels: 512 threads

int launch(hipFunction_t f, unsigned blocks, hipStream_t s, void** args, unsigned threads = 256) {
	return (int) hipModuleLaunchKernel(f, blocks, 1, 1, threads, 1, 1, 0, s, args, nullptr);
}

}  // namespace

extern "C" {

int clo_hip_bitonic_jit_create(int elem_type, int key_type, const char* compare, const char* get_key, const char* compiler_opts,
	void** handle, char** log) {
	if (log) *log = nullptr;
	if (!handle) return CLO_HIP_EARGS;
	*handle = nullptr;
	const char* et = ctype_of(elem_type);
	const char* kt = ctype_of(key_type);
	if (!et || !kt) return CLO_HIP_EUNSUPPORTED;
	const int es = type_size(elem_type);
	const int q = es == 8 ? 4 : 5;

	std::string src;
	src += std::string("#define CLO_SORT_ELEM_TYPE ") + et + "\n";
	src += std::string("#define CLO_SORT_KEY_TYPE ") + kt + "\n";
	src += std::string("#define CLO_SORT_COMPARE_AB(a, b) ") + (compare ? compare : "((a) > (b))") + "\n";
	src += std::string("#define CLO_SORT_KEY_GET_X(x) ") + (get_key ? get_key : "(x)") + "\n";
	src += "#define CLO_JIT_Q " + std::to_string(q) + "\n";
	// (experiment switches of round 5, read once per sorter: CLO_JIT_TB = 8 | 9, CLO_JIT_WPE = waves per SIMD the merge kernel is compiled for)
	const char* env_tb = getenv("CLO_JIT_TB");
	const char* env_wpe = getenv("CLO_JIT_WPE");
	const int tb = (env_tb && (atoi(env_tb) == 8 || atoi(env_tb) == 9)) ? atoi(env_tb) : JIT_TB;
	const int wpe = (env_wpe && atoi(env_wpe) >= 1 && atoi(env_wpe) <= 8) ? atoi(env_wpe) : 1;
	src += "#define CLO_JIT_TB " + std::to_string(tb) + "\n#define CLO_JIT_TT " + std::to_string(1 << tb) + "\n#define CLO_JIT_MERGE_WPE " + std::to_string(wpe) + "\n";
	src += k_src;

	hiprtcProgram prog = nullptr;
	if (hiprtcCreateProgram(&prog, src.c_str(), "clo_sort_bitonic_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
		set_log(log, "hiprtcCreateProgram failed");
		return CLO_HIP_EUNSUPPORTED;
	}
	const std::vector<std::string> optv = clo_jit_options(compiler_opts);
	std::vector<const char*> opts;
	for (const std::string& o : optv) opts.push_back(o.c_str());
	const hiprtcResult cr = hiprtcCompileProgram(prog, (int) opts.size(), opts.data());
	if (cr != HIPRTC_SUCCESS) {
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

	jit_sorter* js = new jit_sorter();
	js->elem_size = es;
	js->q = q;
	js->tb = tb;
	hipError_t e = hipModuleLoadData(&js->module, code.data());
	if (e == hipSuccess) e = hipModuleGetFunction(&js->step, js->module, "jit_step");
	if (e == hipSuccess) e = hipModuleGetFunction(&js->step_any, js->module, "jit_step_any");
	if (e == hipSuccess) e = hipModuleGetFunction(&js->tile, js->module, "jit_tile");
	if (e == hipSuccess) e = hipModuleGetFunction(&js->merge, js->module, "jit_merge");
	if (e == hipSuccess) e = hipModuleGetFunction(&js->gselect, js->module, "jit_gselect");
	for (int ns = 1; ns <= q && e == hipSuccess; ++ns) {
		const std::string name = "jit_strided" + std::to_string(ns);
		e = hipModuleGetFunction(&js->strided[ns], js->module, name.c_str());
	}
	if (e != hipSuccess) {
		set_log(log, std::string("loading the compiled module failed: ") + hipGetErrorString(e));
		if (js->module) (void) hipModuleUnload(js->module);
		delete js;
		return (int) e;
	}
	*handle = js;
	return 0;
}

void clo_hip_bitonic_jit_destroy(void* handle) {
	jit_sorter* js = (jit_sorter*) handle;
	if (!js) return;
	if (js->module) (void) hipModuleUnload(js->module);
	delete js;
}

// upstream's O(n^2) rank sort (sort/clo_sort_gselect.cl:38-58) with the user's compare / get_key: src -> dst
int clo_hip_bitonic_jit_gselect(void* handle, const void* src, void* dst, size_t numel, void* stream) {
	jit_sorter* js = (jit_sorter*) handle;
	if (!js || !js->gselect) return CLO_HIP_EARGS;
	if (numel == 0) return 0;
	if (!src || !dst || src == dst) return CLO_HIP_EARGS;
	unsigned long n = numel;
	void* args[] = { &src, &dst, &n };
	return launch(js->gselect, (unsigned) ((numel + 255) / 256), (hipStream_t) stream, args);
}

// Static LDS per work-group of the kernels clo_hip_bitonic_jit_sort launches for `numel` elements (introspection): only the
// tiled schedule of a power of two from 2^Q elements on runs the tile kernel, whose array is sized for 512 x 2^Q elements.
size_t clo_hip_bitonic_jit_lds_bytes(void* handle, size_t numel, int tiled) {
	const jit_sorter* js = (const jit_sorter*) handle;
	if (!js || numel <= 1 || !tiled || (numel & (numel - 1)) != 0 || numel < ((size_t) 1 << js->q)) return 0;
	const size_t tile = (size_t) (1 << js->tb) << js->q;
	return (tile + tile / 32) * (size_t) js->elem_size;
}

// In-place sort of data[0..numel). A power of two: tiled = 0: one launch per step; 1: the tile/strided
// schedule (upstream's network). Any other numel: the flip network, one launch per step.
int clo_hip_bitonic_jit_sort(void* handle, void* data, size_t numel, int tiled, int* launches, void* stream) {
	jit_sorter* js = (jit_sorter*) handle;
	if (launches) *launches = 0;
	if (!js || !data) return CLO_HIP_EARGS;
	if (numel <= 1) return 0;
	hipStream_t s = (hipStream_t) stream;
	unsigned T = 0;
	while (((size_t) 1 << T) < numel) ++T;
	unsigned long n = numel;
	int count = 0, st = 0;
	if ((numel & (numel - 1)) != 0) {   // not a power of two: the flip network over the next one, one launch per step
		unsigned long npairs = ((unsigned long) 1 << T) / 2;
		const unsigned blocks = (unsigned) ((npairs + 255) / 256);
		for (unsigned stage = 1; stage <= T && !st; ++stage)
			for (unsigned step = stage; step >= 1 && !st; --step) {
				void* args[] = { &data, &n, &npairs, &stage, &step };
				st = launch(js->step_any, blocks, s, args);
				++count;
			}
		if (launches) *launches = count;
		return st;
	}
	const unsigned Q = (unsigned) js->q, KL_MAX = (unsigned) js->tb + Q;

	if (!tiled || T < Q) {
		unsigned long npairs = n / 2;
		const unsigned blocks = (unsigned) ((npairs + 255) / 256);
		for (unsigned stage = 1; stage <= T && !st; ++stage)
			for (unsigned step = stage; step >= 1 && !st; --step) {
				void* args[] = { &data, &npairs, &stage, &step };
				st = launch(js->step, blocks, s, args);
				++count;
			}
	} else {
		unsigned kl = T < KL_MAX ? T : KL_MAX;
		const unsigned tiles = (unsigned) (n >> kl);
		int mode = 1;
		unsigned stage = kl, p_hi = kl;
		{
			void* args[] = { &data, &kl, &stage, &p_hi, &mode };
			st = launch(js->tile, tiles, s, args, 1u << js->tb);
			++count;
		}
		mode = 0;
		for (stage = kl + 1; stage <= T && !st; ++stage) {
			unsigned p = stage;
			while (p > kl && !st) {
				unsigned ns = p - kl;
				if (ns > Q) ns = Q;
				const unsigned long threads = n >> ns;
				void* args[] = { &data, &n, &stage, &p };
				st = launch(js->strided[ns], (unsigned) ((threads + 255) / 256), s, args);
				++count;
				p -= ns;
			}
			if (!st) {   // (stages above the tile only exist on full tiles, kl == KL_MAX: the compile-time schedule)
				void* args[] = { &data, &stage };
				st = launch(js->merge, tiles, s, args, 1u << js->tb);
				++count;
			}
		}
	}
	if (launches) *launches = count;
	return st;
}

}  // extern "C"
