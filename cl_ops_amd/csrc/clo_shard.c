/*
 * clo_shard.c — the sharded sort of include/clo_shard.h: MSD bucket exchange + local
 * satradix, host side in C over the thin HIP / RCCL C-ABI (clo_hip.h). New
 * functionality (the reference is single-device: sort/clo_sort_abstract.c:335).
 *
 * Round 3: the exchange in slices that travel while earlier ones are being sorted, and
 * ranks that fail together. Round 4: the partition takes 8 key bits whatever the world
 * size — 256 sub-buckets, 256 / G per rank — and a rank sorts what it received as ONE
 * segmented sort per slice on the remaining bits (clo_hip_radix_sort_segmented): the
 * partition pass then REPLACES one of the sort's own passes instead of adding one, and a
 * slice's sorts run as launches of the whole slice (round 3 sorted every slice on the full
 * key width: a sharded sort cost 1.5 .. 1.9 times a local one before a byte crossed a
 * link). All of it is described in include/clo_shard.h.
 */
#include "clo_shard.h"
#include "clo_internal.h"

#include <sched.h>
#include <string.h>
#include <time.h>

#define SHARD_MAX_WORLD 8
#define SHARD_MAX_SLICES 8
#define SHARD_PART_BITS 8
#define SHARD_SUBS (1 << SHARD_PART_BITS)   /* sub-buckets of the whole key space; SHARD_SUBS / G per rank */
#define SHARD_TAIL 5                        /* words after a rank's counts in the gather: status, receive capacity, then the
                                             * adaptive slices' sample: sequence number of a finished call, its device us, its slices */
#define SHARD_ROW (SHARD_SUBS + SHARD_TAIL)
/* Below this many BYTES of keys per rank (global mean) one exchange and a plain sort are used: a segmented sort of 256
 * sub-buckets pays off once they span many tiles each. One rank over RCCL, segmented against plain, ms per call
 * (profiles/r04_shard_threshold_probe.txt): uint32 2^24 0.457 / 0.362, 2^25 0.627 / 0.603, 2^26 1.127 / 1.120, 2^27
 * 1.883 / 1.980; uint64 2^24 0.933 / 0.851, 2^25 1.566 / 1.536, 2^26 2.661 / 2.755, 2^27 4.918 / 5.409 — the curves cross at
 * 256 MiB per rank for both widths (round 4 first set 2^22 KEYS). The option "slice_min=<bytes>" overrides it. */
#ifndef SHARD_SLICE_MIN_BYTES_PER_RANK   /* (the sanitizer build of tests/hoststub shrinks it) */
#define SHARD_SLICE_MIN_BYTES_PER_RANK ((uint64_t) 256 << 20)
#endif
#define SHARD_TRIES 2                       /* calls per slice count before the adaptive choice settles */

struct clo_shard_sort {
	CCLContext* ctx;
	CloShardTransport* t;
	CloSort* sorter;
	CloType elem_type;
	int elem_size, bucket_bits, subs;     /* subs = SHARD_SUBS / world: sub-buckets (segments) per rank */
	int slices_opt;                       /* 1, 2, 4, 8 as asked for, 0 = adaptive, -1 = one rank without loopback: no exchange */
	uint64_t slice_min_bytes;             /* per rank (global mean): below it one exchange + a plain sort */
	unsigned timeout_ms;                  /* bound of every host wait for the peers; 0 = none */
	int dead;                             /* a wait gave up and the transport was aborted: only destroy is left */
	int segmented;                        /* the sorter runs segmented sorts (radix 16 / 256) */
	clo_devbuf send, workspace, counts;   /* partitioned shard; partition workspace; my row + the gathered rows (uint64) */
	CCLBuffer* recv;                      /* what arrives (owned; grown on demand; from the transport's recv_alloc when it has one) */
	CCLBuffer* result;                    /* the segmented sorts' second buffer (owned) */
	void* recv_raw; void* result_raw;     /* the transport's allocations behind them */
	size_t recv_cap;
	uint64_t* counts_host;                /* G rows (pageable: a few KiB, and the call waits for them anyway) */
	uint64_t tail_host[SHARD_TAIL];       /* this rank's words of the gather */
	uint64_t grow_host[SHARD_MAX_WORLD];  /* the second agreement round: every rank's "I could grow" */
	void* comm_stream;                    /* the exchanges of a sliced sort */
	void* ev_part;                        /* cq_exec has partitioned: the exchanges may read `send` and write `recv` */
	void* ev_arrived[SHARD_MAX_SLICES];   /* slice j is here */
	void* ev[2][5];                       /* device time stamps of the phases, on cq_exec; two sets, calls alternate */
	void* evx[2];                         /* first all-to-all starts, last one ends (on the stream they run on) */
	int have_phase, last_slices, cur;     /* cur: the event set of the last call */
	size_t last_out, last_in;
	/* adaptive slices: identical on every rank (built from gathered words only) */
	uint64_t seq;                         /* calls made */
	uint64_t ev_seq[2]; int ev_slices[2]; /* which call an event set belongs to; 0 = none / not a sliced call */
	uint64_t sampled_seq;                 /* the last call whose time went into the table */
	double best_us[4];                    /* by log2(slices): best device time seen (max over ranks), 0 = none */
	int tried[4];
	uint64_t class_total;                 /* the global key count the table was built for */
	uint64_t class_seq;                   /* the call that started this size class: samples of earlier calls belong to another */
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

static int rccl_async_error(void* user) {
	rccl_user* u = (rccl_user*) user;
	return (u && u->comm) ? clo_hip_rccl_comm_async_error(u->comm) : 0;
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
	t->async_error = rccl_async_error;
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

/* The partitioned shard holds the sub-buckets in (destination, sub-bucket) order, `subs` per destination. Slice j =
 * the sub-buckets [j * group, (j + 1) * group) of every destination, group = subs / slices: one contiguous range per
 * destination in the shard. What arrives lies slice by slice, inside a slice source by source (one transfer per
 * pair of ranks and slice), inside a source's block sub-bucket by sub-bucket. */
size_t clo_shard_plan_slice(const uint64_t* counts, size_t row, int world, int subs, int slices, int rank, int j,
	size_t* send_counts, size_t* send_offsets, size_t* recv_counts, size_t* recv_offsets,
	size_t* slice_offset, size_t* slice_total) {
	const uint64_t* mine = counts + (size_t) rank * row;
	const int group = subs / slices;
	size_t so = 0;
	for (int p = 0; p < world; ++p) {
		send_counts[p] = 0;
		send_offsets[p] = so;
		for (int k = 0; k < subs; ++k) {
			if (k == j * group) send_offsets[p] = so;
			if (k / group == j) send_counts[p] += (size_t) mine[p * subs + k];
			so += (size_t) mine[p * subs + k];
		}
	}
	size_t ro = 0, at = 0, here = 0;
	for (int s = 0; s < slices; ++s) {
		if (s == j) at = ro;
		for (int p = 0; p < world; ++p) {
			size_t c = 0;
			for (int k = s * group; k < (s + 1) * group; ++k) c += (size_t) counts[(size_t) p * row + (size_t) rank * subs + k];
			if (s == j) { recv_counts[p] = c; recv_offsets[p] = ro; here += c; }
			ro += c;
		}
	}
	if (slice_offset) *slice_offset = at;
	if (slice_total) *slice_total = here;
	return ro;
}

/* ---------------- the object ---------------- */

/* "slices=S", "loopback=0|1", "slice_min=BYTES" and "timeout_ms=N" are ours, the rest goes to satradix. Returns a malloc'd copy of the rest, or NULL on a bad value. */
static char* shard_options(const char* options, int* slices, int* loopback, uint64_t* slice_min, unsigned* timeout_ms, GError** err) {
	*slices = 0;
	*loopback = 0;
	*slice_min = SHARD_SLICE_MIN_BYTES_PER_RANK;
	*timeout_ms = 0;
	const size_t len = options ? strlen(options) : 0;
	char* rest = (char*) calloc(len + 1, 1);
	if (!rest) return NULL;
	const char* p = options ? options : "";
	while (*p) {
		const char* e = strchr(p, ',');
		const size_t n = e ? (size_t) (e - p) : strlen(p);
		if (n > 7 && strncmp(p, "slices=", 7) == 0) {
			if (n == 11 && strncmp(p + 7, "auto", 4) == 0) {
				*slices = 0;
			} else {
				const int v = atoi(p + 7);
				if (v != 1 && v != 2 && v != 4 && v != 8) {
					clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "slices must be 1, 2, 4, 8 or auto (got '%.*s')", (int) (n - 7), p + 7);
					free(rest);
					return NULL;
				}
				*slices = v;
			}
		} else if (n > 9 && strncmp(p, "loopback=", 9) == 0) {
			*loopback = atoi(p + 9) != 0;
		} else if (n > 10 && strncmp(p, "slice_min=", 10) == 0) {
			*slice_min = (uint64_t) strtoull(p + 10, NULL, 10);
		} else if (n > 11 && strncmp(p, "timeout_ms=", 11) == 0) {
			const long long v = atoll(p + 11);
			if (v < 0 || v > 0x7fffffffll) {
				clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "timeout_ms must be between 0 and 2^31 - 1 (got '%.*s')", (int) (n - 11), p + 11);
				free(rest);
				return NULL;
			}
			*timeout_ms = (unsigned) v;
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
	uint64_t slice_min = 0;
	unsigned timeout_ms = 0;
	char* sort_options = shard_options(options, &slices, &loopback, &slice_min, &timeout_ms, err);
	if (!sort_options) return NULL;
	/* (one rank has nothing to exchange; `loopback=1` sends the rank's keys to itself through the whole
	 * protocol: a rehearsal of the exchange over the transport on a one-GPU box) */
	if (world == 1 && !loopback) slices = -1;   /* -1: the shortcut */
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
	ss->subs = SHARD_SUBS / world;
	ss->slices_opt = slices;
	ss->slice_min_bytes = slice_min;
	ss->timeout_ms = timeout_ms;
	{   /* does this sorter run segmented sorts (radix 16 / 256)? Otherwise: one exchange and a plain sort, always */
		const clo_sort_impl_ext* ext = clo_sort_impl_ext_find("satradix");
		int handled = 0;
		if (ext && ext->reserve_segments && ext->sort_segments) ext->reserve_segments(ss->sorter, NULL, 0, 1, &handled, NULL);
		ss->segmented = handled;
	}
	return ss;
}

static void shard_free_recv(CloShardSort* ss) {
	if (ss->recv) ccl_buffer_destroy(ss->recv);
	if (ss->result) ccl_buffer_destroy(ss->result);
	ss->recv = ss->result = NULL;
	if (ss->t->recv_free) {
		if (ss->recv_raw) ss->t->recv_free(ss->t->user, ss->recv_raw);
		if (ss->result_raw) ss->t->recv_free(ss->t->user, ss->result_raw);
	}
	ss->recv_raw = ss->result_raw = NULL;
	ss->recv_cap = 0;
}

/* The receive buffer (and, for segmented sorts, its partner) for `cap` keys: from the transport when it provides memory
 * (registered with the communicator, say), else from the context. 0 or a non-zero status; *e2 may carry the story. */
static int shard_alloc_recv(CloShardSort* ss, size_t cap, int both, GError** e2) {
	const size_t bytes = (cap ? cap : 1) * (size_t) ss->elem_size;
	shard_free_recv(ss);
	for (int i = 0; i < (both ? 2 : 1); ++i) {
		CCLBuffer* b = NULL;
		void* raw = NULL;
		if (ss->t->recv_alloc) {
			raw = ss->t->recv_alloc(ss->t->user, bytes);
			if (raw) b = ccl_buffer_new_from_device_ptr(ss->ctx, raw, bytes, e2);
			if (!b && raw && ss->t->recv_free) { ss->t->recv_free(ss->t->user, raw); raw = NULL; }
		} else {
			b = ccl_buffer_new(ss->ctx, CL_MEM_READ_WRITE, bytes, NULL, e2);
		}
		if (!b) { shard_free_recv(ss); return CLO_HIP_EARGS; }
		if (i == 0) { ss->recv = b; ss->recv_raw = raw; } else { ss->result = b; ss->result_raw = raw; }
	}
	ss->recv_cap = cap;
	return 0;
}

void clo_shard_sort_destroy(CloShardSort* ss) {
	if (!ss) return;
	if (ss->comm_stream) { clo_hip_stream_synchronize(ss->comm_stream); clo_hip_stream_destroy(ss->comm_stream); }
	clo_sort_destroy(ss->sorter);
	shard_free_recv(ss);
	clo_devbuf_release(&ss->send);
	clo_devbuf_release(&ss->workspace);
	clo_devbuf_release(&ss->counts);
	free(ss->counts_host);
	for (int c = 0; c < 2; ++c) for (int i = 0; i < 5; ++i) clo_hip_event_destroy(ss->ev[c][i]);
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
	void** ev = ss->ev[ss->cur];
	if (clo_hip_event_synchronize(ev[4]) != 0) return;
	for (int i = 0; i < 4; ++i) {
		float ms = 0.f;
		if (clo_hip_event_elapsed_ms(ev[i], ev[i + 1], &ms) == 0) device_ms[i] = ms;
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

/* Every event and the transfer stream exist before a call decides to communicate: creating one later could fail on one rank alone. */
static int shard_prepare_events(CloShardSort* ss) {
	int st = 0;
	for (int c = 0; c < 2 && st == 0; ++c) for (int i = 0; i < 5 && st == 0; ++i) if (!ss->ev[c][i]) st = clo_hip_event_create(&ss->ev[c][i]);
	for (int i = 0; i < 2 && st == 0; ++i) if (!ss->evx[i]) st = clo_hip_event_create(&ss->evx[i]);
	for (int i = 0; i < SHARD_MAX_SLICES && st == 0; ++i) if (!ss->ev_arrived[i]) st = clo_hip_event_create(&ss->ev_arrived[i]);
	if (st == 0 && !ss->ev_part) st = clo_hip_event_create(&ss->ev_part);
	if (st == 0 && !ss->comm_stream) st = clo_hip_stream_create_high_priority(&ss->comm_stream);
	return st;
}

/* The sorter's buffers for whatever this call may enqueue on `cap` keys: growing one between two slices would wait for the device. */
static cl_bool shard_reserve_sorter(CloShardSort* ss, CCLQueue* cq_exec, size_t cap, GError** err) {
	const clo_sort_impl_ext* ext = clo_sort_impl_ext_find("satradix");
	if (!ext) return CL_TRUE;
	/* one exchange + plain sort: only calls whose mean shard is below slice_min take it, and a rank's bucket is then at most
	 * G times that mean — with segmented sorts there is no need for the plain sort's partner buffer beyond that size
	 * (config 5: 2.5 GiB per rank that the sliced path never touches) */
	size_t plain_cap = cap;
	if (ss->segmented) {
		const uint64_t most = (uint64_t) ss->t->world * (ss->slice_min_bytes / (uint64_t) ss->elem_size + 1u);
		if (most < (uint64_t) plain_cap) plain_cap = (size_t) most;
	}
	if (ext->reserve && !ext->reserve(ss->sorter, cq_exec, plain_cap, err)) return CL_FALSE;
	if (ss->segmented && ext->reserve_segments) {
		int handled = 0;
		if (!ext->reserve_segments(ss->sorter, cq_exec, cap, ss->subs, &handled, err)) return CL_FALSE;
	}
	return CL_TRUE;
}

/* This rank cannot go on and its peers may already be inside a collective: end the transport. */
static void shard_abort(CloShardSort* ss) {
	if (ss->t->abort) ss->t->abort(ss->t->user);
}

/* ---- bounded waits ----
 * Waiting for a stream that carries a collective is waiting for the PEERS: one that never joins (a crashed process) or
 * that aborted leaves hipStreamSynchronize blocked for ever. So the host polls instead — the stream(s), and the
 * transport's asynchronous error state (RCCL: ncclCommGetAsyncError) — and gives up when `timeout_ms` have passed.
 * 0 = done; CLO_HIP_ENOTREADY = the time is up; anything else = a stream's or the transport's error. Without a
 * bound and without an async_error hook this is the plain blocking wait of rounds 3-4. */
static double shard_now_ms(void) {
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double) ts.tv_sec * 1e3 + (double) ts.tv_nsec * 1e-6;
}

static int shard_wait_streams(CloShardSort* ss, void* s0, void* s1, unsigned timeout_ms) {
	if (timeout_ms == 0 && !ss->t->async_error) {
		int st = s0 ? clo_hip_stream_synchronize(s0) : 0;
		if (st == 0 && s1) st = clo_hip_stream_synchronize(s1);
		return st;
	}
	const double t0 = shard_now_ms();
	for (unsigned spins = 0; ; ++spins) {
		int st = s0 ? clo_hip_stream_query(s0) : 0;
		if (st == 0 && s1) st = clo_hip_stream_query(s1);
		if (st != CLO_HIP_ENOTREADY) return st;
		if (ss->t->async_error && (st = ss->t->async_error(ss->t->user)) != 0) return st;
		const double waited = shard_now_ms() - t0;
		if (timeout_ms != 0 && waited >= (double) timeout_ms) return CLO_HIP_ENOTREADY;
		if (waited < 2.0) sched_yield();   /* (the count exchange sits on the call's critical path: no sleeping while it is young) */
		else { const struct timespec nap = { 0, 50000 }; nanosleep(&nap, NULL); }
	}
}

/* The wait `what` ended in `st` != 0: this rank leaves the transport (its peers' bounded waits, or their async_error, tell them). */
static void shard_wait_failed(CloShardSort* ss, int st, unsigned timeout_ms, const char* what, GError** err) {
	if (st == CLO_HIP_ENOTREADY)
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY, "timed out after %u ms waiting for %s (a peer did not take part): this rank's side of the transport was aborted", timeout_ms, what);
	else
		clo_hip_failed(st, err, what);
	ss->dead = 1;
	if (ss->t->abort) ss->t->abort(ss->t->user);
}

/* ---- adaptive slices ----
 * How many slices pay depends on what only a run shows: the links' rate against the local sort's. Every sliced call is
 * timed on the device (first event to last); a rank reports a finished call's time with a later count exchange
 * (sequence number, microseconds, slices), so after the gather EVERY rank holds the same samples and keeps the same table:
 * best time seen per slice count = max over the ranks, min over the calls. The first calls of a size class try 4, 2, 1
 * and 8 slices SHARD_TRIES times each, later ones use the fastest. Nothing here is decided from local knowledge. */
static int slice_log2(int s) { return s == 1 ? 0 : s == 2 ? 1 : s == 4 ? 2 : 3; }

static void adaptive_sample(CloShardSort* ss, const uint64_t* M, int G) {
	const uint64_t seq0 = M[SHARD_SUBS + 2];
	if (seq0 == 0 || seq0 <= ss->sampled_seq) return;
	/* (a call's time arrives two calls later: after a change of size class the one or two samples still on their way are
	 * the OLD class's — best_us is a minimum that never expires, a much smaller array's time would bias it for good) */
	if (seq0 < ss->class_seq) return;
	uint64_t us = 0;
	for (int p = 0; p < G; ++p) {
		const uint64_t* tail = M + (size_t) p * SHARD_ROW + SHARD_SUBS;
		if (tail[2] != seq0 || tail[4] != M[SHARD_SUBS + 4]) return;   /* the ranks report different calls: no sample this time */
		if (tail[3] > us) us = tail[3];
	}
	const int s = (int) M[SHARD_SUBS + 4];
	if (s != 1 && s != 2 && s != 4 && s != 8) return;
	double* best = &ss->best_us[slice_log2(s)];
	if (*best == 0.0 || (double) us < *best) *best = (double) us;
	ss->sampled_seq = seq0;
}

static int adaptive_choose(CloShardSort* ss, uint64_t grand) {
	if (ss->class_total == 0 || grand > 2 * ss->class_total || 2 * grand < ss->class_total) {   /* another size class: start over */
		ss->class_total = grand;
		ss->class_seq = ss->seq;   /* (the call in progress: the first of the new class) */
		for (int i = 0; i < 4; ++i) { ss->best_us[i] = 0.0; ss->tried[i] = 0; }
	}
	static const int order[4] = { 4, 2, 1, 8 };
	for (int i = 0; i < 4; ++i) {
		const int s = order[i];
		if (s > ss->subs) continue;
		if (ss->tried[slice_log2(s)] < SHARD_TRIES) return s;
	}
	int pick = 4;
	double best = 0.0;
	for (int i = 0; i < 4; ++i) {
		const int s = order[i];
		const double t = ss->best_us[slice_log2(s)];
		if (t > 0.0 && (best == 0.0 || t < best)) { best = t; pick = s; }
	}
	return pick;
}

/* A finished sliced call's sample for the gather (the event set NOT used by the call in progress). */
static void adaptive_report(CloShardSort* ss, int set, uint64_t* tail) {
	tail[2] = tail[3] = tail[4] = 0;
	if (ss->ev_seq[set] == 0 || ss->ev_slices[set] == 0) return;
	float ms = 0.f;
	if (clo_hip_event_query(ss->ev[set][4]) != 0) return;   /* still running: no sample */
	if (clo_hip_event_elapsed_ms(ss->ev[set][0], ss->ev[set][4], &ms) != 0) return;
	tail[2] = ss->ev_seq[set];
	tail[3] = (uint64_t) (ms * 1000.0f);
	tail[4] = (uint64_t) ss->ev_slices[set];
}

CCLEvent* clo_shard_sort_with_device_data(CloShardSort* ss, CCLQueue* cq_exec, CCLBuffer* data_in, size_t numel,
	CCLBuffer** data_out, size_t* numel_out, GError** err) {

	clo_return_val_if_fail(ss != NULL && cq_exec != NULL && data_out != NULL && numel_out != NULL, NULL);
	clo_return_val_if_fail(err == NULL || *err == NULL, NULL);
	clo_return_val_if_fail(numel == 0 || data_in != NULL, NULL);

	const int G = ss->t->world, me = ss->t->rank, es = ss->elem_size, Q = ss->subs;
	void* stream = ccl_queue_get_stream(cq_exec);
	const size_t bytes = numel * (size_t) es;
	ss->have_phase = 0;
	if (ss->dead) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY, "an earlier wait of this sharded sort gave up and aborted its transport: destroy it");
		return NULL;
	}

	if (ss->slices_opt < 0) {   /* one rank, nothing to exchange: a copy and the local sort */
		if (numel > 0 && bytes > ccl_buffer_get_size(data_in)) {
			clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "numel (%zu) exceeds the size of the device buffer", numel);
			return NULL;
		}
		if (ss->recv_cap < numel || !ss->recv) {
			GError* e2 = NULL;
			const int st = shard_alloc_recv(ss, numel ? numel : 1, 0, &e2);
			if (st != 0) { if (e2) clo_gerror_propagate(err, e2); else clo_hip_failed(st, err, "the receive buffer"); return NULL; }
		}
		*data_out = ss->recv;
		*numel_out = numel;
		return clo_sort_with_device_data(ss->sorter, cq_exec, NULL, data_in, ss->recv, numel, 0, err);
	}

	/* ---- what can fail on this rank alone happens BEFORE the count exchange, and is reported through it ---- */
	const size_t row = SHARD_ROW;   /* words a rank contributes */
	GError* local = NULL;           /* this rank's own failure, if any */
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
	if (!ss->counts_host) ss->counts_host = (uint64_t*) calloc((size_t) SHARD_MAX_WORLD * SHARD_ROW, sizeof(uint64_t));
	if (!ss->counts_host) {
		clo_gerror_free(local);
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY, "out of host memory for the count matrix");
		shard_abort(ss);
		return NULL;
	}
	uint64_t* my_row = (uint64_t*) ss->counts.ptr;
	uint64_t* all_rows = my_row + row;
	const int set = ss->cur ^ 1;    /* this call's time stamps; the other set may still belong to a call in flight */
	void** ev = ss->ev[set];
	ss->tail_host[2] = ss->tail_host[3] = ss->tail_host[4] = 0;
	if (ss->ev[set][4]) adaptive_report(ss, set, ss->tail_host);   /* (the call that used this set, two calls ago, has finished or reports nothing) */
	ss->seq += 1;
	ss->ev_seq[set] = 0;

	int st = 0;
	if (status == 0) {
		const size_t ws_bytes = clo_hip_msd_workspace_bytes(numel ? numel : 1, es, SHARD_PART_BITS);
		st = shard_prepare_events(ss);
		if (st == 0) st = clo_devbuf_reserve(&ss->send, bytes ? bytes : 4);
		if (st == 0) st = clo_devbuf_reserve(&ss->workspace, ws_bytes);
		if (st == 0 && !ss->recv) {   /* the usual capacity now, so that growing after the plan is the exception */
			GError* e2 = NULL;
			st = shard_alloc_recv(ss, numel + numel / 4 + 1024, ss->segmented, &e2);
			clo_gerror_free(e2);
		}
		if (st == 0) {   /* the sorter's buffers for anything of that capacity: nothing is left to allocate once the exchange is agreed on */
			GError* e2 = NULL;
			if (!shard_reserve_sorter(ss, cq_exec, ss->recv_cap, &e2)) { st = CLO_HIP_EARGS; clo_gerror_free(e2); }
		}
		/* ---- 1. partition on 8 key bits (its by-product: the sizes of the 256 sub-buckets) ---- */
		if (st == 0) st = clo_hip_event_record(ev[0], stream);
		if (st == 0) st = clo_hip_msd_partition(numel ? ccl_buffer_get_device_ptr(data_in) : NULL, ss->send.ptr, numel, es, 0, 8 * es, SHARD_PART_BITS,
			my_row, ss->workspace.ptr, ss->workspace.bytes, stream);
		if (st == 0) st = clo_hip_event_record(ev[1], stream);
		if (st != 0) {
			clo_hip_failed(st, &local, "preparing the exchange (buffers, clo_hip_msd_partition)");
			status = CLO_ERROR_LIBRARY;
		}
	}

	/* ---- 2. all-gather of counts + status + capacity (+ a timing sample): EVERY rank joins, whatever happened above ---- */
	ss->tail_host[0] = (uint64_t) status;
	ss->tail_host[1] = (uint64_t) ss->recv_cap;
	st = clo_hip_memcpy_h2d_async(my_row + SHARD_SUBS, ss->tail_host, sizeof(ss->tail_host), stream);
	if (st == 0) st = ss->t->all_gather_u64(ss->t->user, my_row, all_rows, row, stream);
	if (st == 0) st = clo_hip_memcpy_d2h_async(ss->counts_host, all_rows, (size_t) G * row * sizeof(uint64_t), stream);
	if (st != 0) {   /* the exchange itself is broken: nothing left to agree through */
		clo_gerror_free(local);
		clo_hip_failed(st, err, "all-gather of the bucket counts");
		shard_abort(ss);
		return NULL;
	}
	if ((st = shard_wait_streams(ss, stream, NULL, ss->timeout_ms)) != 0) {   /* (bounded: a peer that never joins must not hang this rank) */
		clo_gerror_free(local);
		shard_wait_failed(ss, st, ss->timeout_ms, "the all-gather of the bucket counts", err);
		return NULL;
	}
	const uint64_t* M = ss->counts_host;
	for (int p = 0; p < G; ++p) {
		if (M[(size_t) p * row + SHARD_SUBS] == 0) continue;
		if (local) clo_gerror_propagate(err, local);   /* this rank's own story */
		else clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY,
			"rank %d failed before the exchange (error code %llu): no rank sorted", p, (unsigned long long) M[(size_t) p * row + SHARD_SUBS]);
		return NULL;
	}

	/* ---- the plan: from the matrix alone, so every rank decides the same ---- */
	uint64_t grand = 0;
	size_t totals[SHARD_MAX_WORLD];
	int any_grow = 0, too_large = -1;
	for (int p = 0; p < G; ++p) {
		uint64_t tot = 0;
		for (int src = 0; src < G; ++src)
			for (int k = 0; k < Q; ++k) tot += M[(size_t) src * row + (size_t) p * Q + k];
		totals[p] = (size_t) tot;
		grand += tot;
		if (tot > 0xffffffffull && too_large < 0) too_large = p;
		if (tot > M[(size_t) p * row + SHARD_SUBS + 1]) any_grow = 1;
	}
	{
		uint64_t sent = 0;
		for (int k = 0; k < SHARD_SUBS; ++k) sent += M[(size_t) me * row + k];
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
	/* small arrays, or a sorter without segmented sorts: one exchange, one plain sort (the sub-buckets of a destination are neighbours in `send`) */
	const int sliced = ss->segmented && grand > 0 && grand / (uint64_t) G * (uint64_t) es >= ss->slice_min_bytes;
	int use = 1;
	if (sliced) {
		adaptive_sample(ss, M, G);
		use = ss->slices_opt > 0 ? ss->slices_opt : adaptive_choose(ss, grand);
		while (use > Q) use >>= 1;
		ss->tried[slice_log2(use)] += 1;
	}

	if (any_grow) {   /* more skew than the capacity allows for, somewhere: grow, then agree that everyone could */
		int gst = 0;
		if (total > ss->recv_cap) {
			GError* e2 = NULL;
			if (shard_alloc_recv(ss, total, ss->segmented, &e2) != 0 || !shard_reserve_sorter(ss, cq_exec, ss->recv_cap, &e2)) {
				gst = CLO_ERROR_LIBRARY;
				if (e2) local = e2; else clo_gerror_set(&local, CLO_ERROR, CLO_ERROR_LIBRARY, "could not grow the receive buffer to %zu keys", total);
			}
		}
		ss->tail_host[0] = (uint64_t) gst;
		st = clo_hip_memcpy_h2d_async(my_row, ss->tail_host, sizeof(uint64_t), stream);
		if (st == 0) st = ss->t->all_gather_u64(ss->t->user, my_row, all_rows, 1, stream);
		if (st == 0) st = clo_hip_memcpy_d2h_async(ss->grow_host, all_rows, (size_t) G * sizeof(uint64_t), stream);
		if (st != 0) {
			clo_gerror_free(local);
			clo_hip_failed(st, err, "agreement after growing the receive buffers");
			shard_abort(ss);
			return NULL;
		}
		if ((st = shard_wait_streams(ss, stream, NULL, ss->timeout_ms)) != 0) {
			clo_gerror_free(local);
			shard_wait_failed(ss, st, ss->timeout_ms, "the agreement after growing the receive buffers", err);
			return NULL;
		}
		for (int p = 0; p < G; ++p) {
			if (ss->grow_host[p] == 0) continue;
			if (local) clo_gerror_propagate(err, local);
			else clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY, "rank %d could not grow its receive buffer: no rank sorted", p);
			return NULL;
		}
	}
	/* From here on nothing is allocated or created: what can still fail is the runtime refusing to enqueue. While the
	 * exchange is being enqueued that ends the transport (the peers' matching operations would wait for ever); once
	 * every transfer of this rank is enqueued a failure is this rank's alone and is only returned — the exchange
	 * completes for the peers, and aborting it would turn a local error into a global one. */
	if ((st = clo_hip_event_record(ev[2], stream)) != 0) { clo_hip_failed(st, err, "hipEventRecord"); shard_abort(ss); return NULL; }

	/* ---- 3. the slices travel; 4. each is sorted where it landed while the next one travels ---- */
	size_t sc[SHARD_MAX_WORLD], so[SHARD_MAX_WORLD], rc[SHARD_MAX_WORLD], ro[SHARD_MAX_WORLD];
	size_t sb[SHARD_MAX_WORLD], sob[SHARD_MAX_WORLD], rb[SHARD_MAX_WORLD], rob[SHARD_MAX_WORLD];
	char* recv_ptr = (char*) ccl_buffer_get_device_ptr(ss->recv);
	ss->last_out = ss->last_in = 0;
	ss->last_slices = use;
	CCLEvent* evt = NULL;
	CCLBuffer* out_buf = ss->recv;

	if (!sliced) {
		clo_shard_plan_slice(M, row, G, Q, 1, me, 0, sc, so, rc, ro, NULL, NULL);
		for (int p = 0; p < G; ++p) {
			sb[p] = sc[p] * (size_t) es; sob[p] = so[p] * (size_t) es; rb[p] = rc[p] * (size_t) es; rob[p] = ro[p] * (size_t) es;
			if (p != me) { ss->last_out += sb[p]; ss->last_in += rb[p]; }
		}
		st = clo_hip_event_record(ss->evx[0], stream);
		if (st == 0) st = ss->t->all_to_all_v(ss->t->user, ss->send.ptr, sb, sob, recv_ptr, rb, rob, stream);
		if (st != 0) { clo_hip_failed(st, err, "all-to-all of the buckets"); shard_abort(ss); return NULL; }
		st = clo_hip_event_record(ss->evx[1], stream);
		if (st == 0) st = clo_hip_event_record(ev[3], stream);
		if (st != 0) { clo_hip_failed(st, err, "hipEventRecord"); return NULL; }
		evt = clo_sort_with_device_data(ss->sorter, cq_exec, NULL, ss->recv, NULL, total, 0, err);
		if (!evt) return NULL;
	} else {
		const clo_sort_impl_ext* ext = clo_sort_impl_ext_find("satradix");
		const int group = Q / use;
		st = clo_hip_event_record(ss->ev_part, stream);
		if (st == 0) st = clo_hip_stream_wait_event(ss->comm_stream, ss->ev_part);
		if (st == 0) st = clo_hip_event_record(ss->evx[0], ss->comm_stream);
		size_t slice_at[SHARD_MAX_SLICES], slice_n[SHARD_MAX_SLICES], self_at[SHARD_MAX_SLICES];
		for (int j = 0; j < use && st == 0; ++j) {
			clo_shard_plan_slice(M, row, G, Q, use, me, j, sc, so, rc, ro, &slice_at[j], &slice_n[j]);
			self_at[j] = so[me];   /* where this rank's own share of the slice starts in the partitioned shard */
			for (int p = 0; p < G; ++p) {
				sb[p] = sc[p] * (size_t) es; sob[p] = so[p] * (size_t) es; rb[p] = rc[p] * (size_t) es; rob[p] = ro[p] * (size_t) es;
				if (p != me) { ss->last_out += sb[p]; ss->last_in += rb[p]; }
			}
			/* The rank's own share never moves (round 5): the segmented sort's first pass gathers it straight out of `send`
			 * (its pieces below name the second source), so the transport has nothing to copy for this rank — 1 / G of the
			 * exchange's bytes at any G, all of them in the one-rank rehearsal (where the "exchange" was a 0.5-1 ms device copy). */
			sb[me] = rb[me] = 0;
			st = ss->t->all_to_all_v(ss->t->user, ss->send.ptr, sb, sob, recv_ptr, rb, rob, ss->comm_stream);
			if (st == 0) st = clo_hip_event_record(ss->ev_arrived[j], ss->comm_stream);
		}
		if (st != 0) { clo_hip_failed(st, err, "all-to-all of the slices"); shard_abort(ss); return NULL; }
		if ((st = clo_hip_event_record(ss->evx[1], ss->comm_stream)) != 0) { clo_hip_failed(st, err, "hipEventRecord"); return NULL; }
		/* every transfer of this rank is enqueued: the peers get what they wait for whatever happens below */
		char* result_ptr = (char*) ccl_buffer_get_device_ptr(ss->result);
		int in_b = 0;
		for (int j = 0; j < use; ++j) {
			st = clo_hip_stream_wait_event(stream, ss->ev_arrived[j]);
			if (st == 0 && j == 0) st = clo_hip_event_record(ev[3], stream);
			if (st != 0) { clo_hip_failed(st, err, "hipStreamWaitEvent"); return NULL; }
			/* slice j: `group` segments (sub-buckets, ascending key ranges), each in G pieces — one per source rank, in
			 * source order inside the slice — sorted on the bits the partition has not consumed */
			size_t seg_counts[SHARD_SUBS], pn[SHARD_SUBS], po[SHARD_SUBS];
			int ps[SHARD_SUBS], psrc[SHARD_SUBS];
			size_t block_at[SHARD_MAX_WORLD + 1];
			block_at[0] = 0;
			for (int p = 0; p < G; ++p) {
				size_t c = 0;
				for (int kk = 0; kk < group; ++kk) c += (size_t) M[(size_t) p * row + (size_t) me * Q + (size_t) j * group + kk];
				block_at[p + 1] = block_at[p] + c;
			}
			int np = 0;
			for (int kk = 0; kk < group; ++kk) {
				seg_counts[kk] = 0;
				for (int p = 0; p < G; ++p) {
					size_t before = 0;
					for (int k2 = 0; k2 < kk; ++k2) before += (size_t) M[(size_t) p * row + (size_t) me * Q + (size_t) j * group + k2];
					pn[np] = (size_t) M[(size_t) p * row + (size_t) me * Q + (size_t) j * group + kk];
					/* pieces of the other ranks: where they landed in the slice; this rank's own: where the partition left it */
					po[np] = p == me ? self_at[j] + before : block_at[p] + before;
					psrc[np] = p == me;
					ps[np] = kk;
					seg_counts[kk] += pn[np];
					++np;
				}
			}
			int handled = 0, b = 0;
			CCLEvent* e = ext->sort_segments(ss->sorter, cq_exec, recv_ptr + slice_at[j] * (size_t) es, result_ptr + slice_at[j] * (size_t) es,
				slice_n[j], seg_counts, group, pn, po, ps, np, ss->send.ptr, psrc,
				0, 8 * es - SHARD_PART_BITS, &b, &handled, err);
			if (!e || !handled) {
				if (!handled && (err == NULL || *err == NULL)) clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY, "the sorter stopped taking segmented sorts");
				return NULL;
			}
			evt = e;
			if (slice_n[j] > 0 || j == 0) in_b = b;   /* (the same for every slice: it follows from the key bits — an empty slice must not have the last word) */
		}
		out_buf = in_b ? ss->result : ss->recv;
	}
	if ((st = clo_hip_event_record(ev[4], stream)) != 0) { clo_hip_failed(st, err, "hipEventRecord"); return NULL; }
	ss->cur = set;
	ss->ev_seq[set] = ss->seq;
	ss->ev_slices[set] = sliced ? use : 0;
	ss->have_phase = 1;
	*data_out = out_buf;
	*numel_out = total;
	return evt;
}

cl_bool clo_shard_sort_finish(CloShardSort* ss, CCLQueue* cq_exec, unsigned timeout_ms, GError** err) {
	clo_return_val_if_fail(ss != NULL && cq_exec != NULL, CL_FALSE);
	clo_return_val_if_fail(err == NULL || *err == NULL, CL_FALSE);
	if (ss->dead) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY, "an earlier wait of this sharded sort gave up and aborted its transport: destroy it");
		return CL_FALSE;
	}
	if (timeout_ms == 0) timeout_ms = ss->timeout_ms;
	/* the transfer stream first (what the peers owe this rank), then cq_exec (the sorts that wait for it) */
	const int st = shard_wait_streams(ss, ss->slices_opt < 0 ? NULL : ss->comm_stream, ccl_queue_get_stream(cq_exec), timeout_ms);
	if (st != 0) {
		shard_wait_failed(ss, st, timeout_ms, "the key exchange and the local sorts", err);
		return CL_FALSE;
	}
	return ccl_queue_finish(cq_exec, err);   /* (returns at once; reports what the sorts left in their status words) */
}
