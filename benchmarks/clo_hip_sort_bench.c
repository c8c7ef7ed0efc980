/*
 * clo_hip_sort_bench — command-line harness for the CloSort algorithms, written
 * against the public C API only (include/cl_ops.h), the way a cl_ops user
 * program is. It follows the behaviour of the reference's
 * src/benchmarks/clo_sort_bench.c (flags :49-80, sizes 2^4..2^maxpo2 :182, GRand
 * inputs :190-193, clo_sort_with_host_data :196, device time of the exec queue
 * only via CCLProf :201-208, adjacent-pair check :211-226, Mkeys/s formula
 * :233-235, optional TSV of nanoseconds :239-249) and adds a permutation check
 * (xor/sum of the elements), which upstream lacks.
 */
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cl_ops.h>
#include "clo_bench_util.h"

static void usage(const char* argv0) {
	printf("Usage: %s [options]\n"
		"  -a, --algorithm=ALG   sorting algorithm (" CLO_SORT_IMPLS "; default sbitonic)\n"
		"  -g, --alg-opts=STR    algorithm options\n"
		"  -r, --runs=N          runs per size (default 1)\n"
		"  -l, --localsize=N     maximum local work size (accepted, unused by the HIP kernels)\n"
		"  -d, --device=I        device index (default 0)\n"
		"  -s, --rng-seed=S      host RNG seed (default 0)\n"
		"  -t, --type=TYPE       element type (default uint)\n"
		"  -n, --maxpo2=N        sort 2^4 .. 2^N elements (default 24)\n"
		"  -m, --minpo2=N        first size 2^N (default 4)\n"
		"  -o, --out=FILE        write nanoseconds per (size, run) as TSV\n"
		"  -c, --compiler=STR    compiler options (accepted, ignored)\n", argv0);
}

int main(int argc, char** argv) {
	const char* algorithm = "sbitonic";
	const char* alg_options = "";
	const char* type = "uint";
	const char* out = NULL;
	const char* compiler_opts = NULL;
	unsigned runs = 1, maxpo2 = 24, minpo2 = 4, rng_seed = CLO_DEFAULT_SEED;
	size_t lws = 0;
	int dev_idx = -1;

	static const struct option longopts[] = {
		{"algorithm", required_argument, 0, 'a'}, {"alg-opts", required_argument, 0, 'g'},
		{"runs", required_argument, 0, 'r'}, {"localsize", required_argument, 0, 'l'},
		{"device", required_argument, 0, 'd'}, {"rng-seed", required_argument, 0, 's'},
		{"type", required_argument, 0, 't'}, {"maxpo2", required_argument, 0, 'n'},
		{"minpo2", required_argument, 0, 'm'}, {"out", required_argument, 0, 'o'},
		{"compiler", required_argument, 0, 'c'}, {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}
	};
	for (int c; (c = getopt_long(argc, argv, "a:g:r:l:d:s:t:n:m:o:c:h", longopts, NULL)) != -1;) {
		switch (c) {
			case 'a': algorithm = optarg; break;
			case 'g': alg_options = optarg; break;
			case 'r': runs = (unsigned) atoi(optarg); break;
			case 'l': lws = (size_t) atol(optarg); break;
			case 'd': dev_idx = atoi(optarg); break;
			case 's': rng_seed = (unsigned) strtoul(optarg, NULL, 10); break;
			case 't': type = optarg; break;
			case 'n': maxpo2 = (unsigned) atoi(optarg); break;
			case 'm': minpo2 = (unsigned) atoi(optarg); break;
			case 'o': out = optarg; break;
			case 'c': compiler_opts = optarg; break;
			case 'h': usage(argv[0]); return CLO_SUCCESS;
			default: usage(argv[0]); return CLO_ERROR_ARGS;
		}
	}
	if (runs == 0 || maxpo2 > 31 || minpo2 > maxpo2) { usage(argv[0]); return CLO_ERROR_ARGS; }

	int status = CLO_SUCCESS;
	GError* err = NULL;
	CCLContext* ctx = NULL;
	CCLQueue* cq_exec = NULL;
	CCLQueue* cq_comm = NULL;
	CloSort* sorter = NULL;
	unsigned char* host_data = NULL;
	cl_ulong* bench = NULL;
	CloBenchRand rng;

	CloType clotype = clo_type_by_name(type, &err);
	if (err) goto error_handler;
	const size_t bytes = clo_type_sizeof(clotype);
	clo_bench_rand_seed(&rng, rng_seed);

	ctx = ccl_context_new_from_menu_full(&dev_idx, &err);
	if (err) goto error_handler;
	CCLDevice* dev = ccl_context_get_device(ctx, 0, &err);
	if (err) goto error_handler;
	sorter = clo_sort_new(algorithm, alg_options, ctx, &clotype, NULL, NULL, NULL, compiler_opts, &err);
	if (err) goto error_handler;
	cq_exec = ccl_queue_new(ctx, dev, CL_QUEUE_PROFILING_ENABLE, &err);
	if (err) goto error_handler;
	cq_comm = ccl_queue_new(ctx, dev, 0, &err);
	if (err) goto error_handler;

	printf("\n   =========================== Selected options ============================\n\n");
	printf("     Device: %s\n", ccl_device_get_name(dev));
	printf("     Algorithm: %s (options '%s')\n", algorithm, alg_options);
	printf("     Random number generator seed: %u\n", rng_seed);
	printf("     Maximum local worksize (0 is auto-select): %d\n", (int) lws);
	printf("     Type of elements to sort: %s\n", clo_type_get_name(clotype));
	printf("     Number of runs: %u\n\n", runs);

	host_data = (unsigned char*) malloc(bytes << maxpo2);
	bench = (cl_ulong*) calloc((size_t) (maxpo2 + 1) * runs, sizeof(cl_ulong));
	if (!host_data || !bench) { fprintf(stderr, "out of host memory\n"); status = CLO_ERROR_LIBRARY; goto cleanup; }

	for (unsigned N = minpo2; N <= maxpo2; ++N) {
		const size_t num_elems = (size_t) 1 << N;
		int sorted_ok = 1, perm_ok = 1;
		for (unsigned r = 0; r < runs; ++r) {
			unsigned long long x_in = 0, s_in = 0, x_out = 0, s_out = 0;
			for (size_t i = 0; i < num_elems; ++i) {
				clo_bench_rand(&rng, clotype, host_data + bytes * i);
				unsigned long long v = 0;
				memcpy(&v, host_data + bytes * i, bytes);
				x_in ^= v; s_in += v;
			}
			ccl_queue_gc(cq_exec);
			clo_sort_with_host_data(sorter, cq_exec, cq_comm, host_data, host_data, num_elems, lws, &err);
			if (err) goto error_handler;

			/* device time of the exec queue only: transfers travel on cq_comm */
			CCLProf* prof = ccl_prof_new();
			ccl_prof_add_queue(prof, "q_exec", cq_exec);
			ccl_prof_calc(prof, &err);
			if (err) { ccl_prof_destroy(prof); goto error_handler; }
			bench[(size_t) N * runs + r] = ccl_prof_get_duration(prof);
			ccl_prof_destroy(prof);
			ccl_queue_finish(cq_comm, &err);
			if (err) goto error_handler;

			for (size_t i = 0; i < num_elems; ++i) {
				unsigned long long v = 0;
				memcpy(&v, host_data + bytes * i, bytes);
				x_out ^= v; s_out += v;
				if (i + 1 < num_elems && clo_bench_compare(clotype, host_data + bytes * i, host_data + bytes * (i + 1)) > 0)
					sorted_ok = 0;
			}
			if (x_in != x_out || s_in != s_out) perm_ok = 0;
		}
		cl_ulong total = 0;
		for (unsigned r = 0; r < runs; ++r) total += bench[(size_t) N * runs + r];
		printf("       - 2^%u: %lf Mkeys/s %s%s\n", N, (1e-6 * (double) num_elems * runs) / ((double) total * 1e-9),
			sorted_ok ? "" : "(sort did not work)", perm_ok ? "" : "(not a permutation of the input)");
		if (!sorted_ok || !perm_ok) status = CLO_ERROR_LIBRARY;
	}

	if (out) {
		FILE* f = fopen(out, "w");
		if (!f) { fprintf(stderr, "cannot open %s\n", out); status = CLO_ERROR_OPENFILE; goto cleanup; }
		for (unsigned N = minpo2; N <= maxpo2; ++N) {
			fprintf(f, "%u", N);
			for (unsigned r = 0; r < runs; ++r) fprintf(f, "\t%lu", (unsigned long) bench[(size_t) N * runs + r]);
			fprintf(f, "\n");
		}
		fclose(f);
	}
	goto cleanup;

error_handler:
	fprintf(stderr, "Error: %s\n", err ? err->message : "unknown");
	status = err ? err->code : CLO_ERROR_LIBRARY;
	if (status == CLO_SUCCESS) status = CLO_ERROR_LIBRARY;
	clo_gerror_free(err);

cleanup:
	free(host_data);
	free(bench);
	if (sorter) clo_sort_destroy(sorter);
	if (cq_exec) ccl_queue_destroy(cq_exec);
	if (cq_comm) ccl_queue_destroy(cq_comm);
	if (ctx) ccl_context_destroy(ctx);
	return status;
}
