/* clo_sort_bitonic_common.h — shared by the sbitonic and abitonic drivers. */
#ifndef CLO_SORT_BITONIC_COMMON_H
#define CLO_SORT_BITONIC_COMMON_H

#include "clo_sort.h"
#include "clo_internal.h"

typedef struct {
	clo_devbuf padded;   /* used only when numel is not a power of two */
	clo_stream_guard guard;   /* the padded copy is used by one stream at a time */
	/* The launch sequence of a sort depends only on (buffer, numel, stream):
	 * when a call repeats the previous one's, it is captured into a graph and
	 * replayed from then on (sbitonic: 136 launches for 2^16 elements). */
	clo_graph_cache graph;
	int launches;
	int steps;   /* sbitonic: CLO_SBITONIC_STEPS=1 when the sorter was made (one launch per step instead of the tiled schedule) */
} clo_bitonic_state;

CLO_INTERNAL void clo_bitonic_state_release(clo_bitonic_state* state);

/* Runs one of the two HIP schedules on (data_in -> data_out | in place).
 * tiled = 0: one launch per step (sbitonic); 1: LDS/register tiles (abitonic).
 * Returns the event closing the command, or NULL with *err set. */
CLO_INTERNAL CCLEvent* clo_bitonic_run(CloSort* sorter, clo_bitonic_state* state, int tiled, int one_name, const char* evt_name,
	const char* copy_evt_name, CCLQueue* cq_exec, CCLQueue* cq_comm, CCLBuffer* data_in,
	CCLBuffer* data_out, size_t numel, GError** err);

#endif
