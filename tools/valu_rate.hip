// Issue rate of a few VALU instructions on gfx950, measured with s_memtime on one
// wave per SIMD (4 waves per work-group, one work-group): cycles per instruction
// of a dependent chain vs 4 independent chains. Build: hipcc --offload-arch=gfx950
// -O3 tools/valu_rate.hip -o benchmarks/bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 256

template <int OP>
__global__ void k(unsigned long long* out, unsigned seed) {
	unsigned long long a = seed + threadIdx.x, b = a * 3, c = a * 5, d = a * 7;
	unsigned s = (seed & 3u) + 1u;
	unsigned x = seed + threadIdx.x, y = x * 3, z = x * 5, w = x * 7;
	unsigned long long t0 = __builtin_readcyclecounter();
	#pragma unroll
	for (int i = 0; i < REP; ++i) {
		if (OP == 0) { asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a) : "v"(s)); asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(b) : "v"(s)); asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(c) : "v"(s)); asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(d) : "v"(s)); }
		if (OP == 1) { asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(x) : "v"(s)); asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(y) : "v"(s)); asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(z) : "v"(s)); asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(w) : "v"(s)); }
		if (OP == 2) { asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(a) : "v"(s)); asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(b) : "v"(s)); asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(c) : "v"(s)); asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(d) : "v"(s)); }
		if (OP == 3) { asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "v"(s)); asm volatile("v_add_u32 %0, %1, %0" : "+v"(y) : "v"(s)); asm volatile("v_add_u32 %0, %1, %0" : "+v"(z) : "v"(s)); asm volatile("v_add_u32 %0, %1, %0" : "+v"(w) : "v"(s)); }
		if (OP == 4) { asm volatile("v_bfe_u32 %0, %0, %1, 4" : "+v"(x) : "v"(s)); asm volatile("v_bfe_u32 %0, %0, %1, 4" : "+v"(y) : "v"(s)); asm volatile("v_bfe_u32 %0, %0, %1, 4" : "+v"(z) : "v"(s)); asm volatile("v_bfe_u32 %0, %0, %1, 4" : "+v"(w) : "v"(s)); }
		if (OP == 5) { asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(s)); asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(y) : "v"(z), "v"(s)); asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(z) : "v"(w), "v"(s)); asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(w) : "v"(x), "v"(s)); }
		if (OP == 6) { asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y)); asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(y) : "v"(z)); asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(z) : "v"(w)); asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(w) : "v"(x)); }
		if (OP == 7) { asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(s)); asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(y) : "v"(s)); asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(z) : "v"(s)); asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(w) : "v"(s)); }
		if (OP == 8) { asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x) : "v"(s)); asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(y) : "v"(s)); asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(z) : "v"(s)); asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(w) : "v"(s)); }
		if (OP == 9) { asm volatile("v_add_co_u32 %0, vcc, %1, %0\n v_addc_co_u32 %2, vcc, 0, %2, vcc" : "+v"(x), "+v"(s), "+v"(y) : : "vcc"); asm volatile("v_add_co_u32 %0, vcc, %1, %0\n v_addc_co_u32 %2, vcc, 0, %2, vcc" : "+v"(z), "+v"(s), "+v"(w) : : "vcc"); }
	}
	unsigned long long t1 = __builtin_readcyclecounter();
	out[threadIdx.x] = a + b + c + d + x + y + z + w;
	if (threadIdx.x == 0) out[1024] = t1 - t0;
}

template <int OP> void run(const char* name, int per_iter, unsigned long long* d) {
	for (int waves = 1; waves <= 8; waves *= 2) {   // waves per SIMD
		hipLaunchKernelGGL(k<OP>, dim3(1), dim3(256 * waves > 1024 ? 1024 : 256 * waves), 0, 0, d, 1u);
		hipLaunchKernelGGL(k<OP>, dim3(1), dim3(256 * waves > 1024 ? 1024 : 256 * waves), 0, 0, d, 1u);
		hipDeviceSynchronize();
		unsigned long long t; hipMemcpy(&t, d + 1024, 8, hipMemcpyDeviceToHost);
		const int w = (256 * waves > 1024 ? 1024 : 256 * waves) / 256;
		printf("%-16s waves/SIMD %d: %6.2f cycles per instruction per wave -> %5.2f cycles of SIMD issue per instruction\n", name, w, (double) t / (REP * per_iter), (double) t / (REP * per_iter) / w);
		if (256 * waves >= 1024) break;
	}
}

int main() {
	unsigned long long* d; hipMalloc(&d, 8 * 1100);
	run<1>("v_lshlrev_b32", 4, d); run<0>("v_lshlrev_b64", 4, d); run<2>("v_lshrrev_b64", 4, d); run<3>("v_add_u32", 4, d);
	run<4>("v_bfe_u32", 4, d); run<5>("v_perm_b32", 4, d); run<6>("v_mov_dpp", 4, d); run<7>("v_mad_u32_u24", 4, d); run<8>("v_lshl_add_u32", 4, d); run<9>("add_co+addc", 4, d);
	return 0;
}
