/* clo_bench_util.c — see clo_bench_util.h. */
#include "clo_bench_util.h"

#include <math.h>
#include <string.h>

void clo_bench_rand_seed(CloBenchRand* r, uint32_t seed) {
	r->mt[0] = seed;
	for (int i = 1; i < 624; ++i)
		r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t) i;
	r->idx = 624;
}

uint32_t clo_bench_rand_int(CloBenchRand* r) {
	if (r->idx >= 624) {
		for (int k = 0; k < 624; ++k) {
			uint32_t y = (r->mt[k] & 0x80000000u) | (r->mt[(k + 1) % 624] & 0x7fffffffu);
			r->mt[k] = r->mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
		}
		r->idx = 0;
	}
	uint32_t y = r->mt[r->idx++];
	y ^= y >> 11;
	y ^= (y << 7) & 0x9d2c5680u;
	y ^= (y << 15) & 0xefc60000u;
	y ^= y >> 18;
	return y;
}

double clo_bench_rand_double(CloBenchRand* r) {
	const double t = 2.3283064365386962890625e-10; /* 2^-32 */
	double v;
	do {
		v = clo_bench_rand_int(r) * t;
		v = (v + clo_bench_rand_int(r)) * t;
	} while (v >= 1.0);
	return v;
}

int32_t clo_bench_rand_int_range(CloBenchRand* r, int32_t begin, int32_t end) {
	const uint32_t dist = (uint32_t) end - (uint32_t) begin;
	if (dist == 0) return begin;
	uint32_t maxvalue;
	if (dist <= 0x80000000u) {
		uint32_t leftover = (0x80000000u % dist) * 2u;
		if (leftover >= dist) leftover -= dist;
		maxvalue = 0xffffffffu - leftover;
	} else {
		maxvalue = dist - 1u;
	}
	uint32_t v;
	do v = clo_bench_rand_int(r); while (v > maxvalue);
	return begin + (int32_t) (v % dist);
}

void clo_bench_rand(CloBenchRand* r, CloType type, void* location) {
	switch (type) {
		case CLO_CHAR: { int8_t v = (int8_t) clo_bench_rand_int_range(r, -128, 127); memcpy(location, &v, 1); break; }
		case CLO_UCHAR: { uint8_t v = (uint8_t) clo_bench_rand_int_range(r, 0, 255); memcpy(location, &v, 1); break; }
		case CLO_SHORT: { int16_t v = (int16_t) clo_bench_rand_int_range(r, -32768, 32767); memcpy(location, &v, 2); break; }
		case CLO_USHORT: { uint16_t v = (uint16_t) clo_bench_rand_int_range(r, 0, 65535); memcpy(location, &v, 2); break; }
		case CLO_INT: { int32_t v = clo_bench_rand_int_range(r, INT32_MIN, INT32_MAX); memcpy(location, &v, 4); break; }
		case CLO_UINT: { uint32_t v = (uint32_t) (clo_bench_rand_double(r) * 4294967295.0); memcpy(location, &v, 4); break; }
		case CLO_LONG: {
			const double u = clo_bench_rand_double(r);
			const int neg = (clo_bench_rand_int(r) & (1u << 15)) != 0; /* g_rand_boolean */
			int64_t v = (int64_t) (u * (neg ? (double) INT64_MIN : (double) INT64_MAX));
			memcpy(location, &v, 8);
			break;
		}
		case CLO_ULONG: {
			const double d = clo_bench_rand_double(r) * 18446744073709551615.0;
			uint64_t v = d >= 18446744073709551615.0 ? UINT64_MAX : (uint64_t) d;
			memcpy(location, &v, 8);
			break;
		}
		case CLO_FLOAT: { float v = (float) (1.17549435e-38 + clo_bench_rand_double(r) * (3.40282347e+38 - 1.17549435e-38)); memcpy(location, &v, 4); break; }
		case CLO_DOUBLE: { double v = 2.2250738585072014e-308 + clo_bench_rand_double(r) * (1.7976931348623157e+308 - 2.2250738585072014e-308); memcpy(location, &v, 8); break; }
		default: break; /* half: not supported by the HIP build */
	}
}

#define CMP_AS(T) do { T x, y; memcpy(&x, a, sizeof(T)); memcpy(&y, b, sizeof(T)); return (x > y) - (x < y); } while (0)

int clo_bench_compare(CloType type, const void* a, const void* b) {
	switch (type) {
		case CLO_CHAR: CMP_AS(int8_t);
		case CLO_UCHAR: CMP_AS(uint8_t);
		case CLO_SHORT: CMP_AS(int16_t);
		case CLO_USHORT: CMP_AS(uint16_t);
		case CLO_INT: CMP_AS(int32_t);
		case CLO_UINT: CMP_AS(uint32_t);
		case CLO_LONG: CMP_AS(int64_t);
		case CLO_ULONG: CMP_AS(uint64_t);
		case CLO_FLOAT: CMP_AS(float);
		case CLO_DOUBLE: CMP_AS(double);
		default: return 0;
	}
}
