/*
 * host_paths_test.c — TEST INFRASTRUCTURE: drives the C host layer (cl_ops_amd/csrc/*.c) over the host-memory stub of
 * the C-ABI (clo_hip_stub.c) so that its threaded and multi-rank paths run on the CPU under the sanitizers:
 *   1. clo_sort_with_host_data of satradix, pipelined (helper thread copying buckets out while later ones are sorted)
 *   2. clo_scan_with_host_data, pipelined (helper thread, chunk carry)
 *   3. the sharded sort (clo_shard.c) with G = 1 (loopback), 2, 4, 8 ranks as THREADS of this process over an
 *      in-memory transport: count exchange, slices, pieces, segmented sorts, growth of the receive buffers, and the
 *      ranks failing together (arguments; memory)
 * Results are checked against qsort / a serial scan. Exit code 0 = everything right.
 * Built by tests/test_host_sanitizers.py with -fsanitize=address,undefined and with -fsanitize=thread, with the
 * size thresholds of the pipelines shrunk (-DSAT_PIPE_MIN_NUMEL=... etc.) so that a run takes seconds.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <time.h>

#include "cl_ops.h"
#include "clo_hip.h"
#include "clo_shard.h"

static int failures;
#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); ++failures; } } while (0)

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static int cmp_u32(const void* a, const void* b) { const uint32_t x = *(const uint32_t*) a, y = *(const uint32_t*) b; return x < y ? -1 : x > y; }
static int cmp_u64(const void* a, const void* b) { const uint64_t x = *(const uint64_t*) a, y = *(const uint64_t*) b; return x < y ? -1 : x > y; }

static void report(GError** err, const char* what) {
	if (err && *err) { fprintf(stderr, "%s: %s\n", what, (*err)->message); clo_gerror_free(*err); *err = NULL; }
}

static uint64_t fuzz_key(int mode, size_t i);

/* ---- 1. pipelined host sort ---- */
static void test_host_sort_mode(CCLContext* ctx, const char* type_name, size_t n, int pairs, int mode);
static void test_host_sort(CCLContext* ctx, const char* type_name, size_t n, int pairs) { test_host_sort_mode(ctx, type_name, n, pairs, 0); }
static void test_host_sort_mode(CCLContext* ctx, const char* type_name, size_t n, int pairs, int mode) {
	GError* err = NULL;
	CloType et = clo_type_by_name(type_name, &err), kt = CLO_UINT;
	const size_t es = clo_type_sizeof(et);
	CloSort* s = pairs ? clo_sort_new("satradix", NULL, ctx, &et, &kt, NULL, "(uint) ((x) >> 32)", NULL, &err)
		: clo_sort_new("satradix", NULL, ctx, &et, NULL, NULL, NULL, NULL, &err);
	CHECK(s != NULL, "clo_sort_new(%s)", type_name);
	report(&err, "clo_sort_new");
	if (!s) return;
	void* in = malloc(n * es);
	void* out = malloc(n * es);
	void* ref = malloc(n * es);
	for (size_t i = 0; i < n; ++i) {
		if (mode >= 10) {   /* fuzz: the key from the drawn distribution; pairs carry the index as the value */
			const uint64_t v = fuzz_key(mode, i);
			if (es == 4) ((uint32_t*) in)[i] = (uint32_t) (v >> 32);
			else if (pairs) ((uint64_t*) in)[i] = (v & 0xffffffff00000000ull) | (uint64_t) i;
			else ((uint64_t*) in)[i] = v;
		}
		else if (es == 4) ((uint32_t*) in)[i] = (uint32_t) rnd();
		else if (pairs) ((uint64_t*) in)[i] = ((rnd() % 1000) << 52) | i;   /* few keys: equal keys keep their order (the value = the index) */
		else ((uint64_t*) in)[i] = rnd();
	}
	memcpy(ref, in, n * es);
	qsort(ref, n, es, es == 4 ? cmp_u32 : cmp_u64);   /* (pairs: key then index = the stable order) */
	CCLQueue* qx = ccl_queue_new(ctx, NULL, 0, &err);
	CCLQueue* qc = ccl_queue_new(ctx, NULL, 0, &err);
	for (int two = 0; two < 2; ++two) {
		memset(out, 0, n * es);
		const cl_bool ok = clo_sort_with_host_data(s, qx, two ? qc : NULL, in, out, n, 0, &err);
		CHECK(ok, "clo_sort_with_host_data(%s, n=%zu, %s)", type_name, n, two ? "two queues" : "one queue");
		report(&err, "clo_sort_with_host_data");
		CHECK(memcmp(out, ref, n * es) == 0, "host sort of %zu %s%s: wrong result", n, type_name, pairs ? " pairs" : "");
	}
	ccl_queue_destroy(qx);
	ccl_queue_destroy(qc);
	clo_sort_destroy(s);
	free(in); free(out); free(ref);
}

/* ---- 2. pipelined host scan ---- */
static void test_host_scan(CCLContext* ctx, size_t n) {
	GError* err = NULL;
	CloScan* sc = clo_scan_new("blelloch", NULL, ctx, CLO_UINT, CLO_ULONG, NULL, &err);
	CHECK(sc != NULL, "clo_scan_new");
	report(&err, "clo_scan_new");
	if (!sc) return;
	uint32_t* in = (uint32_t*) malloc(n * 4);
	uint64_t* out = (uint64_t*) malloc(n * 8);
	for (size_t i = 0; i < n; ++i) in[i] = (uint32_t) (rnd() & 127u);
	CCLQueue* qx = ccl_queue_new(ctx, NULL, 0, &err);
	const cl_bool ok = clo_scan_with_host_data(sc, qx, NULL, in, out, n, 0, &err);
	CHECK(ok, "clo_scan_with_host_data(n=%zu)", n);
	report(&err, "clo_scan_with_host_data");
	uint64_t acc = 0;
	size_t bad = 0;
	for (size_t i = 0; i < n; ++i) { if (out[i] != acc) ++bad; acc += in[i]; }
	CHECK(bad == 0, "host scan of %zu: %zu wrong sums", n, bad);
	ccl_queue_destroy(qx);
	clo_scan_destroy(sc);
	free(in); free(out);
}

/* ---- 3. the sharded sort, ranks = threads, transport = this process's memory ---- */
#define MAXW 8
typedef struct {
	int world;
	pthread_barrier_t bar;
	const void* send[MAXW];
	const size_t* sb[MAXW]; const size_t* so[MAXW];
	const uint64_t* ag[MAXW];
	int fail_alloc[MAXW];   /* recv_alloc of rank r refuses while set */
} fabric;
typedef struct { fabric* f; int rank; } fabric_user;

static int fab_all_gather(void* user, const uint64_t* s, uint64_t* r, size_t count, void* stream) {
	fabric_user* u = (fabric_user*) user;
	(void) stream;
	u->f->ag[u->rank] = s;
	pthread_barrier_wait(&u->f->bar);
	for (int p = 0; p < u->f->world; ++p) memcpy(r + (size_t) p * count, u->f->ag[p], count * sizeof(uint64_t));
	pthread_barrier_wait(&u->f->bar);
	return 0;
}
static int fab_all_to_all_v(void* user, const void* send, const size_t* sb, const size_t* so, void* recv, const size_t* rb, const size_t* ro, void* stream) {
	fabric_user* u = (fabric_user*) user;
	fabric* f = u->f;
	(void) stream;
	f->send[u->rank] = send; f->sb[u->rank] = sb; f->so[u->rank] = so;
	pthread_barrier_wait(&f->bar);
	int st = 0;
	for (int p = 0; p < f->world; ++p) {   /* what rank p sends to me lands at my [ro[p], + rb[p]) */
		if (f->sb[p][u->rank] != rb[p]) { st = CLO_HIP_EARGS; continue; }
		memcpy((char*) recv + ro[p], (const char*) f->send[p] + f->so[p][u->rank], rb[p]);
	}
	pthread_barrier_wait(&f->bar);
	return st;
}
static void* fab_alloc(void* user, size_t bytes) {
	fabric_user* u = (fabric_user*) user;
	return u->f->fail_alloc[u->rank] ? NULL : malloc(bytes ? bytes : 1);
}
static void fab_free(void* user, void* p) { (void) user; free(p); }

typedef struct {
	fabric* f;
	int rank, es, fail_stage, fail_rank, calls;
	size_t n;
	const char* options;
	void* in;           /* this rank's shard */
	void* out;          /* its sorted bucket (malloc'd by the thread) */
	size_t out_n;
	char msg[256];      /* the error of the failing call, if one was staged */
} rank_arg;

static void* rank_main(void* p) {
	rank_arg* a = (rank_arg*) p;
	GError* err = NULL;
	clo_hip_set_device(0);
	CCLContext* ctx = ccl_context_new_from_device_index(0, &err);
	fabric_user fu = { a->f, a->rank };
	CloShardTransport t;
	memset(&t, 0, sizeof(t));
	t.user = &fu; t.rank = a->rank; t.world = a->f->world;
	t.all_gather_u64 = fab_all_gather; t.all_to_all_v = fab_all_to_all_v;
	t.recv_alloc = fab_alloc; t.recv_free = fab_free;
	CloShardSort* ss = clo_shard_sort_new(ctx, &t, a->es == 4 ? CLO_UINT : CLO_ULONG, a->options, &err);
	CHECK(ss != NULL, "clo_shard_sort_new");
	report(&err, "clo_shard_sort_new");
	CCLQueue* q = ccl_queue_new(ctx, NULL, 0, &err);
	CCLBuffer* in = ccl_buffer_new_from_device_ptr(ctx, a->in, (a->n ? a->n : 1) * (size_t) a->es, &err);
	CCLBuffer* out = NULL;
	size_t m = 0;
	a->msg[0] = 0;
	if (a->fail_stage == 1) {          /* the failing rank claims more keys than its buffer holds: found before the count exchange */
		CCLEvent* e = clo_shard_sort_with_device_data(ss, q, in, a->rank == a->fail_rank ? a->n + 1000 : a->n, &out, &m, &err);
		CHECK(e == NULL && err != NULL, "rank %d: the staged failure of rank %d went unnoticed", a->rank, a->fail_rank);
		if (err) { snprintf(a->msg, sizeof(a->msg), "%s", err->message); clo_gerror_free(err); err = NULL; }
	} else if (a->fail_stage == 2) {   /* a first small call sizes the receive buffers; then the failing rank cannot grow its own */
		CCLEvent* e = clo_shard_sort_with_device_data(ss, q, in, a->n / 64, &out, &m, &err);
		CHECK(e != NULL, "rank %d: the small first call failed", a->rank);
		report(&err, "first call");
		ccl_queue_finish(q, NULL);
		if (a->rank == a->fail_rank) a->f->fail_alloc[a->rank] = 1;
		e = clo_shard_sort_with_device_data(ss, q, in, a->n, &out, &m, &err);
		CHECK(e == NULL && err != NULL, "rank %d: rank %d's allocation failure went unnoticed", a->rank, a->fail_rank);
		if (err) { snprintf(a->msg, sizeof(a->msg), "%s", err->message); clo_gerror_free(err); err = NULL; }
		a->f->fail_alloc[a->rank] = 0;
	}
	for (int c = 0; c < a->calls; ++c) {   /* (several calls: the adaptive slice count moves, the buffers are reused) */
		CCLEvent* e = clo_shard_sort_with_device_data(ss, q, in, a->n, &out, &m, &err);
		CHECK(e != NULL, "rank %d, call %d: clo_shard_sort_with_device_data", a->rank, c);
		report(&err, "clo_shard_sort_with_device_data");
		ccl_queue_finish(q, NULL);
	}
	a->out_n = m;
	a->out = malloc((m ? m : 1) * (size_t) a->es);
	if (out && m) memcpy(a->out, ccl_buffer_get_device_ptr(out), m * (size_t) a->es);
	ccl_buffer_destroy(in);
	ccl_queue_destroy(q);
	clo_shard_sort_destroy(ss);
	ccl_context_destroy(ctx);
	return NULL;
}

/* fuzz (skew >= 10): the distribution's parameters and the ranks' sizes, drawn by fuzz_shard */
static uint64_t fz_and, fz_or, fz_lo, fz_span, fz_values[16];
static int fz_nvalues;
static size_t fz_n[8];
static uint64_t fuzz_key(int mode, size_t i) {
	switch (mode) {
		case 10: return rnd();                                            /* uniform */
		case 11: return (rnd() & fz_and) | fz_or;                         /* some bits fixed: whole buckets, sub-buckets and slices stay empty */
		case 12: return fz_values[rnd() % (uint64_t) fz_nvalues];         /* a handful of values */
		case 13: return fz_lo + (fz_span ? rnd() % fz_span : 0);          /* one range of the key space */
		case 14: return fz_lo + (uint64_t) i * (fz_span / 20000 + 1);     /* ascending */
		default: return fz_or;                                            /* all equal */
	}
}

static void test_shard(int world, int es, size_t n_per_rank, const char* options, int skew, int fail_stage, int calls) {
	fabric f;
	memset(&f, 0, sizeof(f));
	f.world = world;
	pthread_barrier_init(&f.bar, NULL, (unsigned) world);
	rank_arg args[MAXW];
	pthread_t th[MAXW];
	size_t total = 0;
	for (int r = 0; r < world; ++r) {
		rank_arg* a = &args[r];
		memset(a, 0, sizeof(*a));
		a->f = &f; a->rank = r; a->es = es; a->options = options; a->calls = calls;
		a->fail_stage = fail_stage; a->fail_rank = world - 1;
		a->n = skew >= 10 ? fz_n[r] : n_per_rank + (size_t) r * 37;
		a->in = malloc((a->n + 1000) * (size_t) es);
		for (size_t i = 0; i < a->n; ++i) {
			uint64_t v = skew >= 10 ? fuzz_key(skew, i) : rnd();
			if (skew == 1 && (i % 3) == 0) v |= 1ull << 63;                 /* uneven buckets */
			if (skew == 2 || fail_stage == 2) v |= 7ull << 61;               /* every key into the last rank's bucket: it must grow */
			if (skew == 3) {                                                  /* every key in the lowest quarter of its rank's range: of four slices only the first carries keys */
				int bits = 0;
				while ((1 << bits) < world) ++bits;
				v &= ~(3ull << (62 - bits));
			}
			if (es == 4) ((uint32_t*) a->in)[i] = (uint32_t) (v >> 32); else ((uint64_t*) a->in)[i] = v;
		}
		total += a->n;
	}
	for (int r = 0; r < world; ++r) pthread_create(&th[r], NULL, rank_main, &args[r]);
	for (int r = 0; r < world; ++r) pthread_join(th[r], NULL);
	/* the ranks' buckets in rank order = the sorted whole */
	char* all = (char*) malloc(total * (size_t) es);
	char* got = (char*) malloc(total * (size_t) es);
	size_t at = 0, gat = 0;
	for (int r = 0; r < world; ++r) { memcpy(all + at * es, args[r].in, args[r].n * (size_t) es); at += args[r].n; }
	qsort(all, total, (size_t) es, es == 4 ? cmp_u32 : cmp_u64);
	for (int r = 0; r < world; ++r) {
		if (gat + args[r].out_n <= total) memcpy(got + gat * es, args[r].out, args[r].out_n * (size_t) es);
		gat += args[r].out_n;
	}
	CHECK(gat == total, "world %d: the buckets hold %zu keys of %zu", world, gat, total);
	CHECK(gat == total && memcmp(all, got, total * (size_t) es) == 0, "world %d, %d-byte keys, options '%s', skew %d: wrong order", world, es, options ? options : "", skew);
	if (fail_stage) {
		for (int r = 0; r < world; ++r)
			CHECK(args[r].msg[0] != 0 && (r == world - 1 || strstr(args[r].msg, "no rank sorted") != NULL), "world %d, stage %d: rank %d says '%s'", world, fail_stage, r, args[r].msg);
	}
	for (int r = 0; r < world; ++r) { free(args[r].in); free(args[r].out); }
	free(all); free(got);
	pthread_barrier_destroy(&f.bar);
}

/* ---- 4. bounded waits (round 5): a peer that never joins, a transport that fails asynchronously, an exchange that never ends ----
 * The stub executes everything at once, so "a collective still waiting for a rank" is played by the stub's stream hook:
 * while `hang_busy` is set every stream of the process has pending work. The transport below never blocks (as RCCL's
 * enqueue does not): a collective that cannot complete just leaves the flag set. */
extern int (*clo_hip_stub_stream_hook)(void* stream);
static atomic_int hang_busy, hang_aborts, hang_async;
static int hang_hook(void* stream) { (void) stream; return atomic_load(&hang_busy) ? CLO_HIP_ENOTREADY : 0; }
typedef struct { int hang_gather, hang_exchange, world; } hang_user;
static int hang_all_gather(void* user, const uint64_t* s, uint64_t* r, size_t count, void* stream) {
	hang_user* u = (hang_user*) user;
	(void) stream;
	if (u->hang_gather) { atomic_store(&hang_busy, 1); return 0; }   /* the other rank never arrives */
	for (int p = 0; p < u->world; ++p) memcpy(r + (size_t) p * count, s, count * sizeof(uint64_t));
	return 0;
}
static int hang_all_to_all_v(void* user, const void* send, const size_t* sb, const size_t* so, void* recv, const size_t* rb, const size_t* ro, void* stream) {
	hang_user* u = (hang_user*) user;
	(void) stream; (void) rb;
	if (u->hang_exchange) { atomic_store(&hang_busy, 1); return 0; }
	memcpy((char*) recv + ro[0], (const char*) send + so[0], sb[0]);
	return 0;
}
static void hang_abort(void* user) { (void) user; atomic_fetch_add(&hang_aborts, 1); atomic_store(&hang_busy, 0); }   /* (an abort ends the pending operations) */
static int hang_async_error(void* user) { (void) user; return atomic_load(&hang_async); }
static void* hang_killer(void* p) { (void) p; struct timespec ts = { 0, 60 * 1000000 }; nanosleep(&ts, NULL); atomic_store(&hang_async, CLO_HIP_ERCCL - 6); return NULL; }
static double wall_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

static void test_shard_bounded_waits(void) {
	GError* err = NULL;
	clo_hip_stub_stream_hook = hang_hook;
	CCLContext* ctx = ccl_context_new_from_device_index(0, &err);
	CCLQueue* q = ccl_queue_new(ctx, NULL, 0, &err);
	const size_t n = 6000;
	uint32_t* keys = (uint32_t*) malloc(n * 4);
	for (size_t i = 0; i < n; ++i) keys[i] = (uint32_t) rnd();
	CCLBuffer* in = ccl_buffer_new_from_device_ptr(ctx, keys, n * 4, &err);
	for (int scenario = 0; scenario < 3; ++scenario) {
		/* 0: rank 1 of 2 never joins the count exchange, timeout_ms bounds the wait; 1: no bound, but the transport reports an
		 * asynchronous failure after 60 ms; 2: one rank (loopback), the key exchange never completes: clo_shard_sort_finish gives up */
		hang_user hu = { scenario < 2, scenario == 2, scenario < 2 ? 2 : 1 };
		CloShardTransport t;
		memset(&t, 0, sizeof(t));
		t.user = &hu; t.rank = 0; t.world = hu.world;
		t.all_gather_u64 = hang_all_gather; t.all_to_all_v = hang_all_to_all_v; t.abort = hang_abort;
		if (scenario == 1) t.async_error = hang_async_error;
		atomic_store(&hang_busy, 0); atomic_store(&hang_aborts, 0); atomic_store(&hang_async, 0);
		CloShardSort* ss = clo_shard_sort_new(ctx, &t, CLO_UINT, scenario == 0 ? "timeout_ms=150" : (scenario == 2 ? "loopback=1,slices=2,slice_min=1" : NULL), &err);
		CHECK(ss != NULL, "bounded waits %d: clo_shard_sort_new", scenario);
		report(&err, "clo_shard_sort_new");
		if (!ss) continue;
		pthread_t killer;
		if (scenario == 1) pthread_create(&killer, NULL, hang_killer, NULL);
		CCLBuffer* out = NULL;
		size_t m = 0;
		const double t0 = wall_ms();
		CCLEvent* e = clo_shard_sort_with_device_data(ss, q, in, n, &out, &m, &err);
		if (scenario < 2) {
			const double dt = wall_ms() - t0;
			CHECK(e == NULL && err != NULL, "bounded waits %d: the call returned although a rank never joined", scenario);
			CHECK(dt < 5000.0 && (scenario == 1 || dt >= 150.0), "bounded waits %d: gave up after %.0f ms", scenario, dt);
			if (err) CHECK(scenario == 0 ? strstr(err->message, "timed out") != NULL : strstr(err->message, "timed out") == NULL, "bounded waits %d: '%s'", scenario, err->message);
			if (err) { clo_gerror_free(err); err = NULL; }
			CHECK(atomic_load(&hang_aborts) == 1, "bounded waits %d: %d aborts", scenario, atomic_load(&hang_aborts));
			e = clo_shard_sort_with_device_data(ss, q, in, n, &out, &m, &err);   /* the object is spent */
			CHECK(e == NULL && err != NULL && strstr(err->message, "destroy it") != NULL, "bounded waits %d: a second call on the aborted object", scenario);
			if (err) { clo_gerror_free(err); err = NULL; }
		} else {
			CHECK(e != NULL, "bounded waits 2: the call itself only enqueues");
			report(&err, "clo_shard_sort_with_device_data");
			const cl_bool ok = clo_shard_sort_finish(ss, q, 120, &err);
			const double dt = wall_ms() - t0;
			CHECK(!ok && err != NULL && strstr(err->message, "timed out") != NULL, "bounded waits 2: finish returned %d", (int) ok);
			CHECK(dt >= 120.0 && dt < 5000.0, "bounded waits 2: gave up after %.0f ms", dt);
			if (err) { clo_gerror_free(err); err = NULL; }
			CHECK(atomic_load(&hang_aborts) == 1, "bounded waits 2: %d aborts", atomic_load(&hang_aborts));
		}
		if (scenario == 1) pthread_join(killer, NULL);
		clo_shard_sort_destroy(ss);
	}
	/* and a healthy sort under the same polling waits: bound set, nothing hangs */
	{
		hang_user hu = { 0, 0, 1 };
		CloShardTransport t;
		memset(&t, 0, sizeof(t));
		t.user = &hu; t.rank = 0; t.world = 1;
		t.all_gather_u64 = hang_all_gather; t.all_to_all_v = hang_all_to_all_v; t.abort = hang_abort; t.async_error = hang_async_error;
		atomic_store(&hang_busy, 0); atomic_store(&hang_aborts, 0); atomic_store(&hang_async, 0);
		CloShardSort* ss = clo_shard_sort_new(ctx, &t, CLO_UINT, "loopback=1,timeout_ms=2000", &err);
		CCLBuffer* out = NULL;
		size_t m = 0;
		CCLEvent* e = ss ? clo_shard_sort_with_device_data(ss, q, in, n, &out, &m, &err) : NULL;
		CHECK(e != NULL && m == n, "bounded waits: the healthy sort");
		report(&err, "healthy sort");
		CHECK(ss && clo_shard_sort_finish(ss, q, 0, &err), "bounded waits: finish of the healthy sort");
		report(&err, "finish");
		if (e && out) {
			const uint32_t* g = (const uint32_t*) ccl_buffer_get_device_ptr(out);
			size_t bad = 0;
			for (size_t i = 1; i < m; ++i) bad += g[i - 1] > g[i];
			CHECK(bad == 0 && atomic_load(&hang_aborts) == 0, "bounded waits: the healthy sort's result");
		}
		clo_shard_sort_destroy(ss);
	}
	ccl_buffer_destroy(in);
	ccl_queue_destroy(q);
	ccl_context_destroy(ctx);
	free(keys);
	clo_hip_stub_stream_hook = NULL;
}

/* Random worlds, sizes (empty ranks included), options and key distributions through the whole protocol. */
static int fuzz_shard(int cases, uint64_t seed) {
	static const char* const opts[] = { NULL, "slices=1", "slices=2", "slices=4", "slices=8", "radix=256", "radix=256,slices=4", "radix=4", "slices=auto" };
	rng_state = seed * 0x9e3779b97f4a7c15ull + 88172645463325252ull;
	for (int c = 0; c < cases; ++c) {
		const int world = 1 << (int) (rnd() % 4), es = (rnd() & 1) ? 4 : 8, mode = 10 + (int) (rnd() % 6);
		const char* o = opts[rnd() % (sizeof(opts) / sizeof(opts[0]))];
		char options[64];
		snprintf(options, sizeof(options), "%s%s%s", o ? o : "", (o && world == 1) ? "," : "", world == 1 ? "loopback=1" : "");
		for (int r = 0; r < world; ++r) fz_n[r] = (rnd() % 5 == 0) ? 0 : (size_t) (rnd() % 14000);
		fz_and = rnd() | rnd(); fz_or = rnd() & rnd() & rnd();
		fz_nvalues = 1 + (int) (rnd() % 16);
		for (int k = 0; k < 16; ++k) fz_values[k] = rnd();
		fz_lo = rnd(); fz_span = (rnd() >> (rnd() % 60)) ; if (fz_lo + fz_span < fz_lo) fz_span = ~fz_lo;
		const int before = failures;
		test_shard(world, es, 0, options[0] ? options : NULL, mode, 0, 1 + (int) (rnd() % 3));
		if (c % 4 == 0) {   /* the pipelined host sort on the same kind of keys (sizes above the build's threshold, below 2^32 / 4096) */
			GError* err = NULL;
			CCLContext* ctx = ccl_context_new_from_device_index(0, &err);
			const int kind = (int) (rnd() % 3);
			test_host_sort_mode(ctx, kind == 0 ? "uint" : "ulong", 4096 + (size_t) (rnd() % 150000), kind == 2, mode);
			ccl_context_destroy(ctx);
		}
		if (failures != before) fprintf(stderr, "   (fuzz case %d of seed %llu: world %d, %d-byte keys, mode %d, options '%s', sizes %zu %zu %zu %zu ...)\n",
			c, (unsigned long long) seed, world, es, mode, options, fz_n[0], fz_n[1], fz_n[2], fz_n[3]);
	}
	return failures;
}

int main(int argc, char** argv) {
	const int quick = argc > 1 && strcmp(argv[1], "quick") == 0;
	if (argc > 1 && strcmp(argv[1], "fuzz") == 0) {
		const int bad = fuzz_shard(argc > 2 ? atoi(argv[2]) : 100, argc > 3 ? strtoull(argv[3], NULL, 10) : 1);
		if (bad) fprintf(stderr, "%d check(s) failed\n", bad); else printf("shard fuzz ok\n");
		return bad ? 1 : 0;
	}
	GError* err = NULL;
	CCLContext* ctx = ccl_context_new_from_device_index(0, &err);
	if (!ctx) { report(&err, "ccl_context_new_from_device_index"); return 2; }
	/* the pipelines' thresholds are shrunk by the build: these sizes are above them */
	test_host_sort(ctx, "uint", (1u << 17) + 12345, 0);
	test_host_sort(ctx, "ulong", (1u << 16) + 7, 0);
	test_host_sort(ctx, "ulong", (1u << 16) + 1, 1);
	test_host_scan(ctx, (1u << 18) + 77);
	ccl_context_destroy(ctx);
	const size_t n = quick ? 5000 : 9000;
	test_shard(1, 4, n * 4, "loopback=1", 0, 0, 3);
	test_shard(1, 8, n * 2, "loopback=1,slices=8", 0, 0, 1);
	test_shard(2, 4, n, NULL, 1, 0, 10);          /* adaptive slices walk through 4, 2, 1, 8 */
	test_shard(2, 8, n, "slices=2,radix=256", 0, 0, 2);
	test_shard(4, 4, n, "slices=8", 1, 0, 2);
	test_shard(8, 8, n, NULL, 0, 0, 3);
	test_shard(8, 4, n, "slices=4", 2, 0, 2);     /* one bucket far fuller than the capacity: every rank grows or agrees */
	test_shard(2, 4, n, "slices=4", 3, 0, 2);     /* empty slices (the last ones): the result's buffer does not depend on them */
	test_shard(1, 8, n * 2, "loopback=1,slices=4", 3, 0, 1);
	test_shard(4, 4, n, "radix=4", 0, 0, 2);      /* a radix without segmented sorts: one exchange, plain sort */
	test_shard(2, 4, 50, NULL, 0, 0, 2);          /* tiny: one exchange */
	test_shard(4, 8, n, NULL, 0, 1, 1);           /* a rank with bad arguments: all fail, then all sort */
	test_shard(2, 4, n, "slices=2", 0, 2, 1);     /* a rank that cannot grow its receive buffer: all fail, then all sort */
	test_shard_bounded_waits();                   /* a peer that never joins / an asynchronous failure / an exchange that never ends */
	if (failures) fprintf(stderr, "%d check(s) failed\n", failures);
	else printf("host paths ok\n");
	return failures ? 1 : 0;
}
