/*
 * clo_shard.c — the sharded sort of include/clo_shard.h: MSD bucket exchange + local
 * satradix, host side in C over the thin HIP / RCCL C-ABI (clo_hip.h). New
 * functionality (the reference is single-device: sort/clo_sort_abstract.c:335).
 */
#include "clo_shard.h"
#include "clo_internal.h"

#include <string.h>

struct clo_shard_sort {
	CCLContext* ctx;
	CloShardTransport* t;
	CloSort* sorter;
	CloType elem_type;
	int elem_size, bucket_bits;
	clo_devbuf send, workspace, counts;   /* partitioned shard; partition workspace; G + G*G uint64 */
	CCLBuffer* recv;                      /* what arrived (owned; grown on demand) */
	size_t recv_cap;
	uint64_t* counts_host;                /* G*G (pageable: 512 bytes at G = 8, and the call waits for them anyway) */
	void* ev[5];                          /* device time stamps of the phases */
	double phase_ms[4];
	int have_phase;
};

/* ---------------- RCCL transport ---------------- */

typedef struct { void* comm; int rank, world; } rccl_user;

static int rccl_all_gather(void* user, const uint64_t* s, uint64_t* r, size_t count, void* stream) {
	return clo_hip_rccl_all_gather_u64(((rccl_user*) user)->comm, s, r, count, stream);
}

static int rccl_all_to_all_v(void* user, const void* send, const size_t* sb, const size_t* so,
	void* recv, const size_t* rb, const size_t* ro, void* stream) {
	rccl_user* u = (rccl_user*) user;
	return clo_hip_rccl_all_to_all_v(u->comm, u->rank, u->world, send, sb, so, recv, rb, ro, stream);
}

static void rccl_destroy(void* user) {
	rccl_user* u = (rccl_user*) user;
	if (u) { clo_hip_rccl_comm_destroy(u->comm); free(u); }
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

/* ---------------- the object ---------------- */

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
	CloShardSort* ss = (CloShardSort*) calloc(1, sizeof(*ss));
	if (!ss) return NULL;
	ss->sorter = clo_sort_new("satradix", options, ctx, &elem_type, NULL, NULL, NULL, NULL, err);
	if (!ss->sorter) { free(ss); return NULL; }
	ccl_context_ref(ctx);
	ss->ctx = ctx;
	ss->t = transport;
	ss->elem_type = elem_type;
	ss->elem_size = (int) clo_type_sizeof(elem_type);
	ss->bucket_bits = bits;
	return ss;
}

void clo_shard_sort_destroy(CloShardSort* ss) {
	if (!ss) return;
	clo_sort_destroy(ss->sorter);
	if (ss->recv) ccl_buffer_destroy(ss->recv);
	clo_devbuf_release(&ss->send);
	clo_devbuf_release(&ss->workspace);
	clo_devbuf_release(&ss->counts);
	free(ss->counts_host);
	for (int i = 0; i < 5; ++i) clo_hip_event_destroy(ss->ev[i]);
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

static int stamp(CloShardSort* ss, int i, void* stream) {
	if (!ss->ev[i] && clo_hip_event_create(&ss->ev[i]) != 0) return 0;
	return clo_hip_event_record(ss->ev[i], stream) == 0;
}

CCLEvent* clo_shard_sort_with_device_data(CloShardSort* ss, CCLQueue* cq_exec, CCLBuffer* data_in, size_t numel,
	CCLBuffer** data_out, size_t* numel_out, GError** err) {

	clo_return_val_if_fail(ss != NULL && cq_exec != NULL && data_out != NULL && numel_out != NULL, NULL);
	clo_return_val_if_fail(err == NULL || *err == NULL, NULL);
	clo_return_val_if_fail(numel == 0 || data_in != NULL, NULL);

	const int G = ss->t->world, me = ss->t->rank, es = ss->elem_size, b = ss->bucket_bits;
	void* stream = ccl_queue_get_stream(cq_exec);
	const size_t bytes = numel * (size_t) es;
	if (numel > 0 && bytes > ccl_buffer_get_size(data_in)) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "numel (%zu) exceeds the size of the device buffer", numel);
		return NULL;
	}
	if (numel > 0xffffffffull) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "numel per rank must be below 2^32");
		return NULL;
	}
	ss->have_phase = 0;

	if (G == 1) {   /* nothing to exchange: a copy and the local sort */
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

	/* ---- buffers ---- */
	const size_t ws_bytes = clo_hip_msd_workspace_bytes(numel ? numel : 1, es, b);
	if (clo_hip_failed(clo_devbuf_reserve(&ss->send, bytes ? bytes : 4), err, "hipMalloc(send)")) return NULL;
	if (clo_hip_failed(clo_devbuf_reserve(&ss->workspace, ws_bytes), err, "hipMalloc(partition workspace)")) return NULL;
	if (clo_hip_failed(clo_devbuf_reserve(&ss->counts, (size_t) (G + G * G) * sizeof(uint64_t)), err, "hipMalloc(counts)")) return NULL;
	if (!ss->counts_host) {
		ss->counts_host = (uint64_t*) calloc((size_t) G * G, sizeof(uint64_t));
		if (!ss->counts_host) return NULL;
	}
	uint64_t* my_counts = (uint64_t*) ss->counts.ptr;
	uint64_t* all_counts = my_counts + G;

	/* ---- 1. partition (its by-product: the bucket sizes) ---- */
	if (!stamp(ss, 0, stream)) return NULL;
	if (clo_hip_failed(clo_hip_msd_partition(numel ? ccl_buffer_get_device_ptr(data_in) : NULL, ss->send.ptr, numel, es, 0, 8 * es, b,
		my_counts, ss->workspace.ptr, ss->workspace.bytes, stream), err, "clo_hip_msd_partition")) return NULL;
	if (!stamp(ss, 1, stream)) return NULL;

	/* ---- 2. all-gather of the counts; the host needs them to size the exchange ---- */
	if (clo_hip_failed(ss->t->all_gather_u64(ss->t->user, my_counts, all_counts, (size_t) G, stream), err, "all-gather of the bucket counts")) return NULL;
	if (clo_hip_failed(clo_hip_memcpy_d2h_async(ss->counts_host, all_counts, (size_t) G * G * sizeof(uint64_t), stream), err, "hipMemcpyAsync")) return NULL;
	if (clo_hip_failed(clo_hip_stream_synchronize(stream), err, "hipStreamSynchronize")) return NULL;
	size_t sc[8], so[8], rc[8], ro[8], sb[8], sob[8], rb[8], rob[8];
	clo_shard_plan(ss->counts_host, G, me, sc, so, rc, ro);
	size_t total = 0, sent = 0;
	for (int p = 0; p < G; ++p) {
		total += rc[p]; sent += sc[p];
		sb[p] = sc[p] * (size_t) es; sob[p] = so[p] * (size_t) es; rb[p] = rc[p] * (size_t) es; rob[p] = ro[p] * (size_t) es;
	}
	if (sent != numel) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_LIBRARY, "bucket counts (%zu) do not add up to numel (%zu)", sent, numel);
		return NULL;
	}
	if (total > 0xffffffffull) {
		clo_gerror_set(err, CLO_ERROR, CLO_ERROR_ARGS, "bucket %d holds %zu keys: more than one GPU sorts in one call", me, total);
		return NULL;
	}
	if (total > ss->recv_cap || !ss->recv) {   /* 25 % beyond the even share, or the exact size when the keys are more skewed */
		size_t cap = numel + numel / 4 + 1024;
		if (cap < total) cap = total;
		if (ss->recv) ccl_buffer_destroy(ss->recv);
		ss->recv = ccl_buffer_new(ss->ctx, CL_MEM_READ_WRITE, cap * (size_t) es, NULL, err);
		if (!ss->recv) { ss->recv_cap = 0; return NULL; }
		ss->recv_cap = cap;
	}
	if (!stamp(ss, 2, stream)) return NULL;

	/* ---- 3. the buckets travel ---- */
	if (clo_hip_failed(ss->t->all_to_all_v(ss->t->user, ss->send.ptr, sb, sob, ccl_buffer_get_device_ptr(ss->recv), rb, rob, stream),
		err, "all-to-all of the buckets")) return NULL;
	if (!stamp(ss, 3, stream)) return NULL;

	/* ---- 4. local sort of what arrived ---- */
	CCLEvent* evt = clo_sort_with_device_data(ss->sorter, cq_exec, NULL, ss->recv, NULL, total, 0, err);
	if (!evt) return NULL;
	if (!stamp(ss, 4, stream)) return NULL;
	ss->have_phase = 1;
	*data_out = ss->recv;
	*numel_out = total;
	return evt;
}
