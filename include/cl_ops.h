/*
 * cl_ops.h — aggregate public header (reference: src/cl_ops/cl_ops.h:33-48),
 * restricted to the sort/scan hot path this library implements.
 */
#ifndef CL_OPS_H
#define CL_OPS_H

#include "clo_common.h"
#include "clo_ccl.h"
#include "clo_scan.h"
#include "clo_sort.h"
#include "clo_hip.h"
#include "clo_shard.h"

#endif
