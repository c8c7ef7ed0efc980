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

// The tile kernels: CLO_JIT_TT threads on tiles of up to CLO_JIT_TT * 2^Q elements (see JIT_TB on the host side).
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
// so every LDS access is the thread's padded base plus an immediate offset, and the direction is one bit of the tile's
// number. (The run-time schedule above spends more VALU on addresses and per-pair directions than on the user's compare.)
// (`dir` is the same for the whole work-group: the compare's lane mask is flipped by a scalar instruction. Two straight-line
// networks under a scalar branch, one per direction, doubled the time hiprtc takes for these kernels and bought nothing.)
__device__ __forceinline__ void cmpxch_u(E& e1, E& e2, bool dir) {
	const K a = (K) (CLO_SORT_KEY_GET_X(e1));
	const K b = (K) (CLO_SORT_KEY_GET_X(e2));
	const bool cmp = (bool) (CLO_SORT_COMPARE_AB(a, b));
	if (cmp != dir) { const E t = e1; e1 = e2; e2 = t; }
}
template <int V, int NSTEPS>
__device__ __forceinline__ void reg_network_u(E (&v)[V], bool dir) {
	#pragma unroll
	for (int half = V / 2; half >= 1; half /= 2) {
		if (half < (1 << NSTEPS)) {
			#pragma unroll
			for (int j = 0; j < V; ++j)
				if ((j & half) == 0) cmpxch_u(v[j], v[j + half], dir);
		}
	}
}
// one group: the values at register bits [B0, B0 + Q) come out of LDS, NSTEPS steps run on them, they go back
template <int P>
__device__ __forceinline__ void merge_groups(E* s, unsigned tid, bool dir) {
	constexpr int Q = CLO_JIT_Q, V = 1 << Q;
	constexpr int B0 = P > Q ? P - Q : 0, NSTEPS = P - B0;
	const unsigned base = ((tid >> B0) << (B0 + Q)) | (tid & ((1u << B0) - 1u));
	const unsigned pb = PHYS(base);
	E v[V];
	#pragma unroll
	for (int j = 0; j < V; ++j) v[j] = s[pb + PHYS((unsigned) j << B0)];
	reg_network_u<V, NSTEPS>(v, dir);
	#pragma unroll
	for (int j = 0; j < V; ++j) s[pb + PHYS((unsigned) j << B0)] = v[j];
	__syncthreads();   // (a thread rewrites exactly the slots it read: one barrier per group)
	if constexpr (B0 > 0) merge_groups<B0>(s, tid, dir);
}
// (The presort — all of stages 1 .. KL of a tile, 91 steps — keeps the run-time schedule of jit_tile: unrolled at compile time
// it took 0.70 instead of 1.0 ms of a 2^26-key sort, and 2.5 s more of hiprtc per sorter: 3.1 s instead of 0.64.)
extern "C" __global__ __launch_bounds__(CLO_JIT_TT)
void jit_merge(E* __restrict__ data, unsigned stage) {
	constexpr int Q = CLO_JIT_Q, V = 1 << Q, KL = CLO_JIT_TB + Q, TILE = CLO_JIT_TT * V;
	__shared__ E s[TILE + TILE / 32];
	const unsigned tid = threadIdx.x;
	const unsigned long gbase = (unsigned long) blockIdx.x << KL;
	#pragma unroll
	for (int k = 0; k < V; ++k) { const unsigned i = (unsigned) k * CLO_JIT_TT + tid; s[PHYS(i)] = data[gbase + i]; }
	__syncthreads();
	merge_groups<KL>(s, tid, ((gbase >> stage) & 1ul) != 0ul);   // (the direction: one bit of the tile's number)
	#pragma unroll
	for (int k = 0; k < V; ++k) { const unsigned i = (unsigned) k * CLO_JIT_TT + tid; data[gbase + i] = s[PHYS(i)]; }
}

// Two-level strided pass (round 5; the ahead-of-time clo_bitonic_strided2_kernel with the user's compare): steps p .. p-ns+1
// (Q < ns <= 2Q) of a stage in ONE pass. A work-group owns 2^ns rows (index bits [p-ns, p)) of 2^(KL-ns) contiguous elements
// as a tile of 2^KL elements in LDS order row * 2^(KL-ns) + column: the top Q row bits in registers (Q steps), one LDS
// exchange, then tile bits [KL-2Q, KL-Q), whose top ns-Q bits are the remaining row bits (ns-Q steps). The direction is a
// bit of the group's own part of the index: a scalar branch.
template <int V>
__device__ __forceinline__ void reg_network_um(E (&v)[V], int min_half, bool dir) {
	#pragma unroll
	for (int half = V / 2; half >= 1; half /= 2) {
		if (half >= min_half) {   // (the same for the whole launch)
			#pragma unroll
			for (int j = 0; j < V; ++j)
				if ((j & half) == 0) cmpxch_u(v[j], v[j + half], dir);
		}
	}
}
__device__ __forceinline__ void strided2_body(E* __restrict__ data, E* s, unsigned p, unsigned ns, unsigned long wbase, bool dir) {
	constexpr int Q = CLO_JIT_Q, V = 1 << Q, KL = CLO_JIT_TB + Q, B1 = KL - Q, B2 = KL - 2 * Q;
	static_assert(B2 >= 0, "two register groups inside the tile");
	const unsigned tid = threadIdx.x;
	const unsigned C = (unsigned) KL - ns, cmask = (1u << C) - 1u;
	const unsigned t1 = ((tid >> B1) << (B1 + Q)) | (tid & ((1u << B1) - 1u));
	const unsigned t2 = ((tid >> B2) << (B2 + Q)) | (tid & ((1u << B2) - 1u));
	E v[V];
	{
		const E* src = data + wbase + ((unsigned long) (t1 >> C) << (p - ns)) + (t1 & cmask);   // register bits = the top Q row bits = index bits [p-Q, p)
		#pragma unroll
		for (int j = 0; j < V; ++j) v[j] = src[(unsigned long) j << (p - Q)];
	}
	reg_network_um<V>(v, 1, dir);
	{
		const unsigned pf = PHYS(t1), pt = PHYS(t2);
		#pragma unroll
		for (int j = 0; j < V; ++j) s[pf + PHYS((unsigned) j << B1)] = v[j];
		__syncthreads();
		#pragma unroll
		for (int j = 0; j < V; ++j) v[j] = s[pt + PHYS((unsigned) j << B2)];
	}
	reg_network_um<V>(v, V >> (ns - Q), dir);
	#pragma unroll
	for (int j = 0; j < V; ++j) {
		const unsigned t = t2 | ((unsigned) j << B2);
		data[wbase + ((unsigned long) (t >> C) << (p - ns)) + (t & cmask)] = v[j];
	}
}
extern "C" __global__ __launch_bounds__(CLO_JIT_TT)
void jit_strided2l(E* __restrict__ data, unsigned stage, unsigned p, unsigned ns) {
	constexpr int Q = CLO_JIT_Q, V = 1 << Q, KL = CLO_JIT_TB + Q, TILE = CLO_JIT_TT * V;
	__shared__ E s[TILE + TILE / 32];
	const unsigned C = (unsigned) KL - ns;
	const unsigned midbits = p - ns - C;          // index bits between the columns and the rows
	const unsigned long w = blockIdx.x;
	const unsigned long wbase = ((w >> midbits) << p) | ((w & ((1ul << midbits) - 1ul)) << C);
	strided2_body(data, s, p, ns, wbase, ((wbase >> stage) & 1ul) != 0ul);
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
	hipFunction_t step = nullptr, step_any = nullptr, tile = nullptr, merge = nullptr, strided2 = nullptr, gselect = nullptr, strided[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
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

// Thread bits of the tile kernels: 256 threads on tiles of 256 * 2^Q elements. (512 threads on the ahead-of-time kernels' 2^14-
// element tiles measured the same, 3.58 against 3.59 ms per 2^26-key sort, once the merge kernel no longer carried one
// network per direction — with those it needed 164 VGPRs, one work-group per CU: 4.33 against 4.02 ms; profiles/r05_jit_variants.txt.)
constexpr int JIT_TB = 8;

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
	const int tb = JIT_TB;
	src += "#define CLO_JIT_TB " + std::to_string(tb) + "\n#define CLO_JIT_TT " + std::to_string(1 << tb) + "\n";
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
	if (e == hipSuccess) e = hipModuleGetFunction(&js->strided2, js->module, "jit_strided2l");
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
// tiled schedule of a power of two from 2^Q elements on runs the tile kernel, whose array is sized for 2^JIT_TB x 2^Q elements.
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
			// The steps above the tile: plain strided passes of up to Q steps first, then — from Q + 1 steps on — ONE two-level
			// pass of up to NS2 steps right above the tile (rows of at least 128 bytes): 13 steps = 5 + 8, where round 4 made
			// 5 + 5 + 3 (2^26 keys: 18 strided launches instead of 24).
			const unsigned NS2 = (unsigned) js->tb < 2u * Q ? (unsigned) js->tb : 2u * Q;   // rows of 2^(KL - ns) elements: 32 four-byte / 16 eight-byte ones (128 bytes) at ns = NS2
			while (p > kl && !st) {
				const unsigned h = p - kl;
				if (h > Q && h <= NS2 && kl == KL_MAX) {
					unsigned ns = h;
					void* args[] = { &data, &stage, &p, &ns };
					st = launch(js->strided2, tiles, s, args, 1u << js->tb);
					++count;
					p -= ns;
					continue;
				}
				unsigned ns = h > NS2 ? h - NS2 : h;
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
