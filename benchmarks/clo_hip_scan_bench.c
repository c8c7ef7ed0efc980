/*
 * clo_hip_scan_bench — command-line harness for the CloScan algorithms, written
 * against the public C API only (include/cl_ops.h). It follows the behaviour of
 * the reference's src/benchmarks/clo_scan_bench.c (flags :53-92, sizes
 * init_elems * 2^k :205-280, values in [0,128) :219-223, clo_scan_with_host_data
 * :227, exec-queue device time :231-239, element-wise comparison with a serial
 * scan incl. the overflow guard :252-271, MValues/s :278).
 */
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cl_ops.h>
#include "clo_bench_util.h"

static void usage(const char* argv0) {
	printf("Usage: %s [options]\n"
		"  -a, --algorithm=ALG   scan algorithm (" CLO_SCAN_IMPLS "; default blelloch)\n"
		"  -p, --alg-opts=STR    algorithm options\n"
		"  -r, --runs=N          runs per size (default 1)\n"
		"  -l, --localsize=N     maximum local work size (accepted, unused by the HIP kernel)\n"
		"  -d, --device=I        device index (default 0)\n"
		"  -s, --rng-seed=S      host RNG seed (default 0)\n"
		"  -t, --type=TYPE       element type (default uint)\n"
		"  -y, --type-sum=TYPE   sum type (default ulong)\n"
		"  -i, --init-elems=N    first size (default 4)\n"
		"  -n, --num-doub=N      number of sizes, each twice the previous (default 24)\n"
		"  -u, --no-check        do not compare with the serial scan\n"
		"  -o, --out=FILE        write nanoseconds per (size, run) as TSV\n"
		"  -c, --compiler=STR    compiler options (accepted, ignored)\n", argv0);
}

static unsigned long long get_elem(const unsigned char* p, size_t bytes) {
	unsigned long long v = 0;
	memcpy(&v, p, bytes);
	return v;
}

int main(int argc, char** argv) {
	const char* algorithm = "blelloch";
	const char* alg_options = "";
	const char* type = "uint";
	const char* type_sum = "ulong";
	const char* out = NULL;
	const char* compiler_opts = NULL;
	unsigned runs = 1, init_elems = 4, num_doub = 24, rng_seed = CLO_DEFAULT_SEED;
	int no_check = 0, dev_idx = -1;
	size_t lws = 0;

	static const struct option longopts[] = {
		{"algorithm", required_argument, 0, 'a'}, {"alg-opts", required_argument, 0, 'p'},
		{"runs", required_argument, 0, 'r'}, {"localsize", required_argument, 0, 'l'},
		{"device", required_argument, 0, 'd'}, {"rng-seed", required_argument, 0, 's'},
		{"type", required_argument, 0, 't'}, {"type-sum", required_argument, 0, 'y'},
		{"init-elems", required_argument, 0, 'i'}, {"num-doub", required_argument, 0, 'n'},
		{"no-check", no_argument, 0, 'u'}, {"out", required_argument, 0, 'o'},
		{"compiler", required_argument, 0, 'c'}, {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}
	};
	for (int c; (c = getopt_long(argc, argv, "a:p:r:l:d:s:t:y:i:n:uo:c:h", longopts, NULL)) != -1;) {
		switch (c) {
			case 'a': algorithm = optarg; break;
			case 'p': alg_options = optarg; break;
			case 'r': runs = (unsigned) atoi(optarg); break;
			case 'l': lws = (size_t) atol(optarg); break;
			case 'd': dev_idx = atoi(optarg); break;
			case 's': rng_seed = (unsigned) strtoul(optarg, NULL, 10); break;
			case 't': type = optarg; break;
			case 'y': type_sum = optarg; break;
			case 'i': init_elems = (unsigned) atoi(optarg); break;
			case 'n': num_doub = (unsigned) atoi(optarg); break;
			case 'u': no_check = 1; break;
			case 'o': out = optarg; break;
			case 'c': compiler_opts = optarg; break;
			case 'h': usage(argv[0]); return CLO_SUCCESS;
			default: usage(argv[0]); return CLO_ERROR_ARGS;
		}
	}
	if (runs == 0 || init_elems == 0 || num_doub == 0 || num_doub > 31) { usage(argv[0]); return CLO_ERROR_ARGS; }

	int status = CLO_SUCCESS;
	GError* err = NULL;
	CCLContext* ctx = NULL;
	CCLQueue* cq_exec = NULL;
	CCLQueue* cq_comm = NULL;
	CloScan* scanner = NULL;
	unsigned char* host_data = NULL;
	unsigned char* host_scanned = NULL;
	cl_ulong* bench = NULL;
	CloBenchRand rng;

	CloType t_elem = clo_type_by_name(type, &err);
	if (err) goto error_handler;
	CloType t_sum = clo_type_by_name(type_sum, &err);
	if (err) goto error_handler;
	const size_t bytes = clo_type_sizeof(t_elem), bytes_sum = clo_type_sizeof(t_sum);
	const size_t max_elems = (size_t) init_elems << (num_doub - 1);
	clo_bench_rand_seed(&rng, rng_seed);

	ctx = ccl_context_new_from_menu_full(&dev_idx, &err);
	if (err) goto error_handler;
	CCLDevice* dev = ccl_context_get_device(ctx, 0, &err);
	if (err) goto error_handler;
	scanner = clo_scan_new(algorithm, alg_options, ctx, t_elem, t_sum, compiler_opts, &err);
	if (err) goto error_handler;
	cq_exec = ccl_queue_new(ctx, dev, CL_QUEUE_PROFILING_ENABLE, &err);
	if (err) goto error_handler;
	cq_comm = ccl_queue_new(ctx, dev, 0, &err);
	if (err) goto error_handler;

	printf("\n   =========================== Selected options ============================\n\n");
	printf("     Device: %s\n", ccl_device_get_name(dev));
	printf("     Algorithm: %s; elements %s -> sums %s; runs %u; seed %u\n\n", algorithm,
		clo_type_get_name(t_elem), clo_type_get_name(t_sum), runs, rng_seed);

	host_data = (unsigned char*) malloc(bytes * max_elems);
	host_scanned = (unsigned char*) malloc(bytes_sum * max_elems);
	bench = (cl_ulong*) calloc((size_t) num_doub * runs, sizeof(cl_ulong));
	if (!host_data || !host_scanned || !bench) { fprintf(stderr, "out of host memory\n"); status = CLO_ERROR_LIBRARY; goto cleanup; }

	const unsigned long long max_sum = bytes_sum >= 8 ? ~0ull : ((1ull << (8 * bytes_sum)) - 1ull);
	size_t num_elems = init_elems;
	for (unsigned N = 0; N < num_doub; ++N, num_elems *= 2) {
		const char* scan_ok = "";
		for (unsigned r = 0; r < runs; ++r) {
			for (size_t i = 0; i < num_elems; ++i) {
				unsigned long long value = (unsigned long long) (clo_bench_rand_double(&rng) * 128);
				memcpy(host_data + bytes * i, &value, bytes);
			}
			ccl_queue_gc(cq_exec);
			clo_scan_with_host_data(scanner, cq_exec, cq_comm, host_data, host_scanned, num_elems, lws, &err);
			if (err) goto error_handler;
			CCLProf* prof = ccl_prof_new();
			ccl_prof_add_queue(prof, "q_exec", cq_exec);
			ccl_prof_calc(prof, &err);
			if (err) { ccl_prof_destroy(prof); goto error_handler; }
			bench[(size_t) N * runs + r] = ccl_prof_get_duration(prof);
			ccl_prof_destroy(prof);
			ccl_queue_finish(cq_comm, &err);
			if (err) goto error_handler;

			if (no_check) {
				scan_ok = "[Unverified]";
			} else {
				unsigned long long value_host = 0;
				for (size_t i = 0; i < num_elems; ++i) {
					if (i > 0) value_host += get_elem(host_data + bytes * (i - 1), bytes);
					if (value_host > max_sum) { scan_ok = "[Overflow]"; break; }
					if (get_elem(host_scanned + bytes_sum * i, bytes_sum) != value_host) {
						scan_ok = "[Scan did not work]";
						status = CLO_ERROR_LIBRARY;
						break;
					}
				}
			}
		}
		cl_ulong total = 0;
		for (unsigned r = 0; r < runs; ++r) total += bench[(size_t) N * runs + r];
		printf("       - %10zu : %f MValues/s %s\n", num_elems, (1e-6 * (double) num_elems * runs) / ((double) total * 1e-9), scan_ok);
	}

	if (out) {
		FILE* f = fopen(out, "w");
		if (!f) { fprintf(stderr, "cannot open %s\n", out); status = CLO_ERROR_OPENFILE; goto cleanup; }
		for (unsigned N = 0; N < num_doub; ++N) {
			fprintf(f, "%zu", (size_t) init_elems << N);
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
	free(host_scanned);
	free(bench);
	if (scanner) clo_scan_destroy(scanner);
	if (cq_exec) ccl_queue_destroy(cq_exec);
	if (cq_comm) ccl_queue_destroy(cq_comm);
	if (ctx) ccl_context_destroy(ctx);
	return status;
}
