#!/bin/bash
# Developer probe (GPU box): the scan bench line with alternative builds of the library (CLO_HIP_LIBRARY), same box, interleaved.
for rep in 1 2; do
	for v in lib lib_a lib_b; do
		CLO_HIP_LIBRARY=$PWD/cl_ops_amd/$v/libcl_ops_hip.so python bench.py --workload scan --steps 200 --warmup 5 --no-cpu-baseline 2>/dev/null \
			| python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
	done
done
