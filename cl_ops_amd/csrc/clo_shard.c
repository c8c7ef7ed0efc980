/*
 * clo_shard.c — the sharded sort of include/clo_shard.h: MSD bucket exchange + local
 * satradix, host side in C over the thin HIP / RCCL C-ABI (clo_hip.h). New
 * functionality (the reference is single-device: sort/clo_sort_abstract.c:335).
 *
 * Round 3: the exchange in slices that travel while earlier ones are being sorted, and
 * ranks that fail together (both described in include/clo_shard.h).
 */
#include "clo_shard.h"
#include "clo_internal.h"

#include <string.h>

#define SHARD_MAX_WORLD 8
#define SHARD_MAX_SLICES 8
#define SHARD_TAIL 2                      /* words after a rank's counts in the gather: status, receive capacity */
#define SHARD_SLICE_MIN_PER_RANK ((uint64_t) 1 << 22)   /* below this many keys per rank (global mean) one exchange is used */

struct clo_shard_sort {
	CCLContext* ctx;
	CloShardTransport* t;
	CloSort* sorter;
	CloType elem_type;
	int elem_size, bucket_bits, slices, slice_bits;
	clo_devbuf send, workspace, counts;   /* partitioned shard; partition workspace; my row + the gathered rows (uint64) */
	CCLBuffer* recv;                      /* what arrived (owned; grown on demand) */
	size_t recv_cap;
	uint64_t* counts_host;                /* G rows (pageable: a few KiB, and the call waits for them anyway) */
	uint64_t tail_host[SHARD_TAIL];       /* status, receive capacity: this rank's words of the gather */
	uint64_t grow_host[SHARD_MAX_WORLD];  /* the second agreement round: every rank's "I could grow" */
	void* comm_stream;                    /* the exchanges of a sliced sort (created at the first one) */
	void* ev_part;                        /* cq_exec has partitioned: the exchanges may read `send` and write `recv` */
	void* ev_arrived[SHARD_MAX_SLICES];   /* sub-bucket j is here */
	void* ev[5];                          /* device time stamps of the phases, on cq_exec */
	void* evx[2];                         /* first all-to-all starts, last one ends (on the stream they run on) */
	int have_phase, last_slices;
	size_t last_out, last_in;
};

/* ---------------- RCCL transport ---------------- */

typedef struct { void* comm; int rank, world; } rccl_user;

static int rccl_all_gather(void* user, const uint64_t* s, uint64_t* r, size_t count, void* stream) {
	rccl_user* u = (rccl_user*) user;
	if (!u->comm) return CLO_HIP_EARGS;   /* aborted */
	return clo_hip_rccl_all_gather_u64(u->comm, s, r, count, stream);
}

static int rccl_all_to_all_v(void* user, const void* send, const size_t* sb, const size_t* so,
	void* recv, const size_t* rb, const size_t* ro, void* stream) {
	rccl_user* u = (rccl_user*) user;
	if (!u->comm) return CLO_HIP_EARGS;
	return clo_hip_rccl_all_to_all_v(u->comm, u->rank, u->world, send, sb, so, recv, rb, ro, stream);
}

static void rccl_abort(void* user) {
	rccl_user* u = (rccl_user*) user;
	if (u && u->comm) { clo_hip_rccl_comm_abort(u->comm); u->comm = NULL; }
}

static void rccl_destroy(void* user) {
	rccl_user* u = (rccl_user*) user;
	if (u) { if (u->comm) clo_hip_rccl_comm_destroy(u->comm); free(u); }
}

cl_bool clo_shard_rccl_unique_id(void* id_out, GError** err) {
	clo_return_val_if_fail(id_out != NULL, CL_FALSE);
	return clo_hip_failed(clo_hip_rccl_unique_id(id_out), err, "ncclGetUniqueId") ? CL_FALSE : CL_TRUE;
}

CloShardTransport* clo_shard_transport_new_rccl(const void* id, int rank, int world, GError** err) {
	clo_return_val_if_fail(id != NULL, NULL);
	if (world < 1 || rank < 0 || rank >= world) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "rank %d of %d is not a valid rank", rank, world);
		return NULL;
	}
	rccl_user* u = (rccl_user*) calloc(1, sizeof(*u));
	CloShardTransport* t = (CloShardTransport*) calloc(1, sizeof(*t));
	if (!u || !t) { free(u); free(t); return NULL; }
	if (clo_hip_failed(clo_hip_rccl_comm_create(&u->comm, id, rank, world), err, "ncclCommInitRank")) { free(u); free(t); return NULL; }
	u->rank = rank; u->world = world;
	t->user = u; t->rank = rank; t->world = world;
	t->all_gather_u64 = rccl_all_gather;
	t->all_to_all_v = rccl_all_to_all_v;
	t->destroy = rccl_destroy;
	t->abort = rccl_abort;
	return t;
}

void clo_shard_transport_destroy(CloShardTransport* t) {
	if (!t) return;
	if (t->destroy) t->destroy(t->user);
	free(t);
}

/* ---------------- plan ---------------- */

void clo_shard_plan(const uint64_t* counts, int world, int rank,
	size_t* send_counts, size_t* send_offsets, size_t* recv_counts, size_t* recv_offsets) {
	size_t so = 0, ro = 0;
	for (int p = 0; p < world; ++p) {
		send_counts[p] = (size_t) counts[(size_t) rank * world + p];   /* my bucket p goes to rank p */
		send_offsets[p] = so;
		so += send_counts[p];
		recv_counts[p] = (size_t) counts[(size_t) p * world + rank];   /* rank p's bucket `rank` comes to me */
		recv_offsets[p] = ro;
		ro += recv_counts[p];
	}
}

/* The partitioned shard holds the sub-buckets in (bucket, slice) order; the result holds
 * sub-bucket 0 of every source rank, then sub-bucket 1 ...: ascending key ranges. */
size_t clo_shard_plan_slice(const uint64_t* counts, size_t row, int world, int slices, int rank, int j,
	size_t* send_counts, size_t* send_offsets, size_t* recv_counts, size_t* recv_offsets,
	size_t* slice_offset, size_t* slice_total) {
	const uint64_t* mine = counts + (size_t) rank * row;
	size_t so = 0;
	for (int p = 0; p < world; ++p)
		for (int k = 0; k < slices; ++k) {
			if (k == j) { send_counts[p] = (size_t) mine[p * slices + k]; send_offsets[p] = so; }
			so += (size_t) mine[p * slices + k];
		}
	size_t ro = 0, total = 0, at = 0, here = 0;
	for (int k = 0; k < slices; ++k) {
		if (k == j) at = ro;
		for (int p = 0; p < world; ++p) {
			const size_t c = (size_t) counts[(size_t) p * row + (size_t) rank * slices + k];
			if (k == j) { recv_counts[p] = c; recv_offsets[p] = ro; here += c; }
			ro += c;
		}
	}
	total = ro;
	if (slice_offset) *slice_offset = at;
	if (slice_total) *slice_total = here;
	return total;
}

/* ---------------- the object ---------------- */

/* "slices=S" is ours, the rest goes to satradix. Returns a malloc'd copy of the rest, or NULL on a bad value. */
static char* shard_options(const char* options, int* slices, int* loopback, GError** err) {
	*slices = 0;
	*loopback = 0;
	const size_t len = options ? strlen(options) : 0;
	char* rest = (char*) calloc(len + 1, 1);
	if (!rest) return NULL;
	const char* p = options ? options : "";
	while (*p) {
		const char* e = strchr(p, ',');
		const size_t n = e ? (size_t) (e - p) : strlen(p);
		if (n > 7 && strncmp(p, "slices=", 7) == 0) {
			const int v = atoi(p + 7);
			if (v != 1 && v != 2 && v != 4 && v != 8) {
				clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "slices must be 1, 2, 4 or 8 (got '%.*s')", (int) (n - 7), p + 7);
				free(rest);
				return NULL;
			}
			*slices = v;
		} else if (n > 9 && strncmp(p, "loopback=", 9) == 0) {
			*loopback = atoi(p + 9) != 0;
		} else if (n > 0) {
			if (rest[0]) strcat(rest, ",");
			strncat(rest, p, n);
		}
		p = e ? e + 1 : p + n;
	}
	return rest;
}

CloShardSort* clo_shard_sort_new(CCLContext* ctx, CloShardTransport* transport, CloType elem_type,
	const char* options, GError** err) {
	clo_return_val_if_fail(ctx != NULL && transport != NULL, NULL);
	clo_return_val_if_fail(err == NULL || *err == NULL, NULL);
	if (elem_type != CLO_UINT && elem_type != CLO_ULONG) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "the sharded sort handles uint and ulong keys (got %s)",
			clo_type_get_name(elem_type) ? clo_type_get_name(elem_type) : "?");
		return NULL;
	}
	const int world = transport->world;
	int bits = 0;
	while ((1 << bits) < world) ++bits;
	if (world < 1 || (1 << bits) != world || bits > 3 || transport->rank < 0 || transport->rank >= world) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "the world size must be 1, 2, 4 or 8 (got %d)", world);
		return NULL;
	}
	int slices = 0, loopback = 0;
	char* sort_options = shard_options(options, &slices, &loopback, err);
	if (!sort_options) return NULL;
	if (slices == 0) slices = 4;
	/* (one rank has nothing to exchange; `loopback=1` keeps the slices and sends the rank's keys to itself
	 * through the whole protocol: a rehearsal of the exchange over the transport on a one-GPU box) */
	if (world == 1 && !loopback) slices = 1;
	int sbits = 0;
	while ((1 << sbits) < slices) ++sbits;
	CloShardSort* ss = (CloShardSort*) calloc(1, sizeof(*ss));
	if (!ss) { free(sort_options); return NULL; }
	ss->sorter = clo_sort_new("satradix", sort_options, ctx, &elem_type, NULL, NULL, NULL, NULL, err);
	free(sort_options);
	if (!ss->sorter) { free(ss); return NULL; }
	ccl_context_ref(ctx);
	ss->ctx = ctx;
	ss->t = transport;
	ss->elem_type = elem_type;
	ss->elem_size = (int) clo_type_sizeof(elem_type);
	ss->bucket_bits = bits;
	ss->slices = slices;
	ss->slice_bits = sbits;
	return ss;
}

void clo_shard_sort_destroy(CloShardSort* ss) {
	if (!ss) return;
	if (ss->comm_stream) { clo_hip_stream_synchronize(ss->comm_stream); clo_hip_stream_destroy(ss->comm_stream); }
	clo_sort_destroy(ss->sorter);
	if (ss->recv) ccl_buffer_destroy(ss->recv);
	clo_devbuf_release(&ss->send);
	clo_devbuf_release(&ss->workspace);
	clo_devbuf_release(&ss->counts);
	free(ss->counts_host);
	for (int i = 0; i < 5; ++i) clo_hip_event_destroy(ss->ev[i]);
	for (int i = 0; i < 2; ++i) clo_hip_event_destroy(ss->evx[i]);
	for (int i = 0; i < SHARD_MAX_SLICES; ++i) clo_hip_event_destroy(ss->ev_arrived[i]);
	clo_hip_event_destroy(ss->ev_part);
	ccl_context_unref(ss->ctx);
	free(ss);
}

void clo_shard_sort_get_phase_ms(CloShardSort* ss, double device_ms[4]) {
	if (!ss || !device_ms) return;
	for (int i = 0; i < 4; ++i) device_ms[i] = 0.0;
	if (!ss->have_phase) return;
	if (clo_hip_event_synchronize(ss->ev[4]) != 0) return;
	for (int i = 0; i < 4; ++i) {
		float ms = 0.f;
		if (clo_hip_event_elapsed_ms(ss->ev[i], ss->ev[i + 1], &ms) == 0) device_ms[i] = ms;
	}
}

void clo_shard_sort_get_exchange(CloShardSort* ss, size_t* bytes_out, size_t* bytes_in, double* device_ms, int* slices) {
	if (bytes_out) *bytes_out = 0;
	if (bytes_in) *bytes_in = 0;
	if (device_ms) *device_ms = 0.0;
	if (slices) *slices = 0;
	if (!ss || !ss->have_phase) return;
	if (bytes_out) *bytes_out = ss->last_out;
	if (bytes_in) *bytes_in = ss->last_in;
	if (slices) *slices = ss->last_slices;
	float ms = 0.f;
	if (device_ms && ss->evx[1] && clo_hip_event_synchronize(ss->evx[1]) == 0
		&& clo_hip_event_elapsed_ms(ss->evx[0], ss->evx[1], &ms) == 0) *device_ms = ms;
}

static int record(void** evt, void* stream) {   /* 0 or a clo_hip status */
	if (!*evt) { const int st = clo_hip_event_create(evt); if (st != 0) return st; }
	return clo_hip_event_record(*evt, stream);
}

/* Test hook: CLO_SHARD_TEST_FAIL="<rank>:<stage>" makes that rank fail on its own at stage 1
 * (before the count exchange) or 2 (while growing its receive buffer) — what an allocation
 * failure on one GPU looks like to the protocol. */
static int injected_failure(int rank, int stage) {
	const char* x = getenv("CLO_SHARD_TEST_FAIL");
	if (!x) return 0;
	int r = -1, s = -1;
	if (sscanf(x, "%d:%d", &r, &s) != 2) return 0;
	return r == rank && s == stage;
}

/* This rank cannot go on and its peers may already be inside a collective: end the transport. */
static void shard_abort(CloShardSort* ss) {
	if (ss->t->abort) ss->t->abort(ss->t->user);
}

CCLEvent* clo_shard_sort_with_device_data(CloShardSort* ss, CCLQueue* cq_exec, CCLBuffer* data_in, size_t numel,
	CCLBuffer** data_out, size_t* numel_out, GError** err) {

	clo_return_val_if_fail(ss != NULL && cq_exec != NULL && data_out != NULL && numel_out != NULL, NULL);
	clo_return_val_if_fail(err == NULL || *err == NULL, NULL);
	clo_return_val_if_fail(numel == 0 || data_in != NULL, NULL);

	const int G = ss->t->world, me = ss->t->rank, es = ss->elem_size, b = ss->bucket_bits;
	const int S = ss->slices, tb = b + ss->slice_bits;
	void* stream = ccl_queue_get_stream(cq_exec);
	const size_t bytes = numel * (size_t) es;
	ss->have_phase = 0;

	if (G == 1 && S == 1) {   /* nothing to exchange: a copy and the local sort */
		if (numel > 0 && bytes > ccl_buffer_get_size(data_in)) {
			clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "numel (%zu) exceeds the size of the device buffer", numel);
			return NULL;
		}
		if (ss->recv_cap < numel || !ss->recv) {
			if (ss->recv) ccl_buffer_destroy(ss->recv);
			ss->recv = ccl_buffer_new(ss->ctx, CL_MEM_READ_WRITE, bytes ? bytes : 4, NULL, err);
			if (!ss->recv) { ss->recv_cap = 0; return NULL; }
			ss->recv_cap = numel;
		}
		*data_out = ss->recv;
		*numel_out = numel;
		return clo_sort_with_device_data(ss->sorter, cq_exec, NULL, data_in, ss->recv, numel, 0, err);
	}

	/* ---- what can fail on this rank alone happens BEFORE the count exchange, and is reported through it ---- */
	const size_t row = (size_t) G * S + SHARD_TAIL;   /* words a rank contributes */
	GError* local = NULL;                             /* this rank's own failure, if any */
	int status = 0;
	if (numel > 0 && bytes > ccl_buffer_get_size(data_in)) {
		clo_gerror_set(&local, CLO_ERROR, CLO_ERROR_ARGS, "numel (%zu) exceeds the size of the device buffer", numel);
		status = CLO_ERROR_ARGS;
	} else if (numel > 0xffffffffull) {
		clo_gerror_set(&local, CLO_ERROR, CLO_ERROR_ARGS, "numel per rank must be below 2^32");
		status = CLO_ERROR_ARGS;
	}
	/* the gather itself needs this buffer: without it the rank cannot even say that it failed */
	if (clo_hip_failed(clo_devbuf_reserve(&ss->counts, (row + (size_t) G * row) * sizeof(uint64_t)), err, "hipMalloc(counts)")) {
		clo_gerror_free(local);
		shard_abort(ss);
		return NULL;
	}
	if (!ss->counts_host) ss->counts_host = (uint64_t*) calloc((size_t) SHARD_MAX_WORLD * (SHARD_MAX_WORLD * SHARD_MAX_SLICES + SHARD_TAIL), sizeof(uint64_t));
	if (!ss->counts_host) {
		clo_gerror_free(local);
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY, "out of host memory for the count matrix");
		shard_abort(ss);
		return NULL;
	}
	uint64_t* my_row = (uint64_t*) ss->counts.ptr;
	uint64_t* all_rows = my_row + row;

	int st = 0;
	if (status == 0) {
		const size_t ws_bytes = clo_hip_msd_workspace_bytes(numel ? numel : 1, es, tb);
		if (injected_failure(me, 1)) st = CLO_HIP_EARGS;
		if (st == 0) st = clo_devbuf_reserve(&ss->send, bytes ? bytes : 4);
		if (st == 0) st = clo_devbuf_reserve(&ss->workspace, ws_bytes);
		if (st == 0 && !ss->recv) {   /* the usual capacity now, so that growing after the plan is the exception */
			const size_t cap = numel + numel / 4 + 1024;
			GError* e2 = NULL;
			ss->recv = ccl_buffer_new(ss->ctx, CL_MEM_READ_WRITE, cap * (size_t) es, NULL, &e2);
			if (ss->recv) ss->recv_cap = cap; else { st = CLO_HIP_EARGS; clo_gerror_free(e2); }
		}
		/* ---- 1. partition (its by-product: the sizes of the G x S sub-buckets) ---- */
		if (st == 0) st = record(&ss->ev[0], stream);
		if (st == 0) st = clo_hip_msd_partition(numel ? ccl_buffer_get_device_ptr(data_in) : NULL, ss->send.ptr, numel, es, 0, 8 * es, tb,
			my_row, ss->workspace.ptr, ss->workspace.bytes, stream);
		if (st == 0) st = record(&ss->ev[1], stream);
		if (st != 0) {
			clo_hip_failed(st, &local, "preparing the exchange (buffers, clo_hip_msd_partition)");
			status = CLO_ERROR_LIBRARY;
		}
	}

	/* ---- 2. all-gather of counts + status + capacity: EVERY rank joins, whatever happened above ---- */
	ss->tail_host[0] = (uint64_t) status;
	ss->tail_host[1] = (uint64_t) ss->recv_cap;
	st = clo_hip_memcpy_h2d_async(my_row + (size_t) G * S, ss->tail_host, sizeof(ss->tail_host), stream);
	if (st == 0) st = ss->t->all_gather_u64(ss->t->user, my_row, all_rows, row, stream);
	if (st == 0) st = clo_hip_memcpy_d2h_async(ss->counts_host, all_rows, (size_t) G * row * sizeof(uint64_t), stream);
	if (st == 0) st = clo_hip_stream_synchronize(stream);
	if (st != 0) {   /* the exchange itself is broken: nothing left to agree through */
		clo_gerror_free(local);
		clo_hip_failed(st, err, "all-gather of the bucket counts");
		shard_abort(ss);
		return NULL;
	}
	const uint64_t* M = ss->counts_host;
	for (int p = 0; p < G; ++p) {
		if (M[(size_t) p * row + (size_t) G * S] == 0) continue;
		if (local) clo_gerror_propagate(err, local);   /* this rank's own story */
		else clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY,
			"rank %d failed before the exchange (error code %llu): no rank sorted", p, (unsigned long long) M[(size_t) p * row + (size_t) G * S]);
		return NULL;
	}

	/* ---- the plan: from the matrix alone, so every rank decides the same ---- */
	uint64_t grand = 0;
	size_t totals[SHARD_MAX_WORLD];
	int any_grow = 0, too_large = -1;
	for (int p = 0; p < G; ++p) {
		uint64_t tot = 0;
		for (int src = 0; src < G; ++src)
			for (int k = 0; k < S; ++k) tot += M[(size_t) src * row + (size_t) p * S + k];
		totals[p] = (size_t) tot;
		grand += tot;
		if (tot > 0xffffffffull && too_large < 0) too_large = p;
		if (tot > M[(size_t) p * row + (size_t) G * S + 1]) any_grow = 1;
	}
	{
		uint64_t sent = 0;
		for (int k = 0; k < G * S; ++k) sent += M[(size_t) me * row + k];
		if (sent != numel) {   /* (a broken partition: cannot happen — and it happens on this rank alone) */
			clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY, "bucket counts (%llu) do not add up to numel (%zu)", (unsigned long long) sent, numel);
			shard_abort(ss);
			return NULL;
		}
	}
	if (too_large >= 0) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "bucket %d holds %zu keys: more than one GPU sorts in one call", too_large, totals[too_large]);
		return NULL;
	}
	const size_t total = totals[me];
	/* small arrays: one exchange, one sort (the sub-buckets of a rank are neighbours in `send`) */
	const int use = (S > 1 && grand / (uint64_t) G >= SHARD_SLICE_MIN_PER_RANK) ? S : 1;

	if (any_grow) {   /* more skew than the capacity allows for, somewhere: grow, then agree that everyone could */
		int gst = 0;
		if (total > ss->recv_cap) {
			if (ss->recv) ccl_buffer_destroy(ss->recv);
			ss->recv = NULL;
			ss->recv_cap = 0;
			GError* e2 = NULL;
			if (!injected_failure(me, 2)) ss->recv = ccl_buffer_new(ss->ctx, CL_MEM_READ_WRITE, total * (size_t) es, NULL, &e2);
			if (ss->recv) ss->recv_cap = total;
			else { gst = CLO_ERROR_LIBRARY; if (e2) local = e2; else clo_gerror_set(&local, CLO_ERROR, CLO_ERROR_LIBRARY, "could not grow the receive buffer to %zu keys", total); }
		}
		ss->tail_host[0] = (uint64_t) gst;
		st = clo_hip_memcpy_h2d_async(my_row, ss->tail_host, sizeof(uint64_t), stream);
		if (st == 0) st = ss->t->all_gather_u64(ss->t->user, my_row, all_rows, 1, stream);
		if (st == 0) st = clo_hip_memcpy_d2h_async(ss->grow_host, all_rows, (size_t) G * sizeof(uint64_t), stream);
		if (st == 0) st = clo_hip_stream_synchronize(stream);
		if (st != 0) {
			clo_gerror_free(local);
			clo_hip_failed(st, err, "agreement after growing the receive buffers");
			shard_abort(ss);
			return NULL;
		}
		for (int p = 0; p < G; ++p) {
			if (ss->grow_host[p] == 0) continue;
			if (local) clo_gerror_propagate(err, local);
			else clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY, "rank %d could not grow its receive buffer: no rank sorted", p);
			return NULL;
		}
	}
	if ((st = record(&ss->ev[2], stream)) != 0) { clo_hip_failed(st, err, "hipEventRecord"); shard_abort(ss); return NULL; }

	/* ---- 3. the sub-buckets travel; 4. each is sorted where it lands while the next one travels ---- */
	size_t sc[SHARD_MAX_WORLD], so[SHARD_MAX_WORLD], rc[SHARD_MAX_WORLD], ro[SHARD_MAX_WORLD];
	size_t sb[SHARD_MAX_WORLD], sob[SHARD_MAX_WORLD], rb[SHARD_MAX_WORLD], rob[SHARD_MAX_WORLD];
	void* recv_ptr = ccl_buffer_get_device_ptr(ss->recv);
	ss->last_out = ss->last_in = 0;
	ss->last_slices = use;
	CCLEvent* evt = NULL;

	if (use == 1) {
		/* the S sub-buckets of a destination are contiguous in `send`: one transfer per pair of ranks */
		size_t s_off = 0, r_off = 0;
		for (int p = 0; p < G; ++p) {
			size_t s_cnt = 0, r_cnt = 0;
			for (int k = 0; k < S; ++k) { s_cnt += (size_t) M[(size_t) me * row + (size_t) p * S + k]; r_cnt += (size_t) M[(size_t) p * row + (size_t) me * S + k]; }
			sb[p] = s_cnt * (size_t) es; sob[p] = s_off * (size_t) es; rb[p] = r_cnt * (size_t) es; rob[p] = r_off * (size_t) es;
			s_off += s_cnt; r_off += r_cnt;
			if (p != me) { ss->last_out += sb[p]; ss->last_in += rb[p]; }
		}
		st = record(&ss->evx[0], stream);
		if (st == 0) st = ss->t->all_to_all_v(ss->t->user, ss->send.ptr, sb, sob, recv_ptr, rb, rob, stream);
		if (st == 0) st = record(&ss->evx[1], stream);
		if (st == 0) st = record(&ss->ev[3], stream);
		if (st != 0) { clo_hip_failed(st, err, "all-to-all of the buckets"); shard_abort(ss); return NULL; }
		evt = clo_sort_with_device_data(ss->sorter, cq_exec, NULL, ss->recv, NULL, total, 0, err);
		if (!evt) { shard_abort(ss); return NULL; }
	} else {
		if (!ss->comm_stream && (st = clo_hip_stream_create_high_priority(&ss->comm_stream)) != 0) {
			clo_hip_failed(st, err, "hipStreamCreate"); shard_abort(ss); return NULL;
		}
		st = record(&ss->ev_part, stream);
		if (st == 0) st = clo_hip_stream_wait_event(ss->comm_stream, ss->ev_part);
		if (st == 0) st = record(&ss->evx[0], ss->comm_stream);
		size_t slice_at[SHARD_MAX_SLICES], slice_n[SHARD_MAX_SLICES];
		for (int j = 0; j < use && st == 0; ++j) {
			clo_shard_plan_slice(M, row, G, S, me, j, sc, so, rc, ro, &slice_at[j], &slice_n[j]);
			for (int p = 0; p < G; ++p) {
				sb[p] = sc[p] * (size_t) es; sob[p] = so[p] * (size_t) es; rb[p] = rc[p] * (size_t) es; rob[p] = ro[p] * (size_t) es;
				if (p != me) { ss->last_out += sb[p]; ss->last_in += rb[p]; }
			}
			st = ss->t->all_to_all_v(ss->t->user, ss->send.ptr, sb, sob, recv_ptr, rb, rob, ss->comm_stream);
			if (st == 0) st = record(&ss->ev_arrived[j], ss->comm_stream);
		}
		if (st == 0) st = record(&ss->evx[1], ss->comm_stream);
		if (st != 0) { clo_hip_failed(st, err, "all-to-all of the sub-buckets"); shard_abort(ss); return NULL; }
		{   /* the sorter's buffers for EVERY slice size now: growing them between two slices would wait for the device */
			const clo_sort_impl_ext* ext = clo_sort_impl_ext_find("satradix");
			for (int j = 0; j < use && ext && ext->reserve; ++j)
				if (!ext->reserve(ss->sorter, cq_exec, slice_n[j], err)) { shard_abort(ss); return NULL; }
		}
		for (int j = 0; j < use; ++j) {
			st = clo_hip_stream_wait_event(stream, ss->ev_arrived[j]);
			if (st == 0 && j == 0) st = record(&ss->ev[3], stream);
			if (st != 0) { clo_hip_failed(st, err, "hipStreamWaitEvent"); shard_abort(ss); return NULL; }
			if (slice_n[j] == 0 && (j + 1 < use || evt != NULL)) continue;
			CCLBuffer* part = ccl_buffer_new_from_device_ptr(ss->ctx, (char*) recv_ptr + slice_at[j] * (size_t) es,
				(slice_n[j] ? slice_n[j] : 1) * (size_t) es, err);
			if (!part) { shard_abort(ss); return NULL; }
			evt = clo_sort_with_device_data(ss->sorter, cq_exec, NULL, part, NULL, slice_n[j], 0, err);
			ccl_buffer_destroy(part);   /* (a view: nothing is freed) */
			if (!evt) { shard_abort(ss); return NULL; }
		}
	}
	if ((st = record(&ss->ev[4], stream)) != 0) { clo_hip_failed(st, err, "hipEventRecord"); return NULL; }
	ss->have_phase = 1;
	*data_out = ss->recv;
	*numel_out = total;
	return evt;
}
