#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): bench lines, rocprofv3 kernel stats, the two PMC passes for the HBM traffic of
# every workload, the harness sweeps and the probes. Raw output goes to gpurun_out/$TAG/; tools/summarize_profiles.py
# turns it into the files under profiles/. One part per gpurun call (a call is limited to 20 minutes):
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh r05 bench'     (parts: bench trace pmc harness probes sq shard)
set -o pipefail
TAG=${1:-r05}
PART=${2:-bench}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
WL="satradix_u32 satradix_pairs satradix_u64 scan abitonic"
case $PART in
bench)
	# the line the driver runs (headline + every other BASELINE config as a leg), then every workload on its own
	python3 "$ROOT/bench.py" --steps 20 --warmup 5 > "$OUT/bench_headline_with_configs.json" 2> "$OUT/bench_headline_with_configs.err" || exit 1
	for w in $WL sbitonic; do
		python3 "$ROOT/bench.py" --workload $w --steps 30 --warmup 3 --no-configs > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err" || exit 1
		echo "bench $w done"
	done
	# the other path at the headline sizes (the library's choice there: chain-free pair passes): single-sweep passes forced on
	for w in satradix_u32 satradix_u64 satradix_pairs; do
		CLO_RADIX_SWEEP=1 python3 "$ROOT/bench.py" --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-configs > "$OUT/bench_${w}_sweep.json" 2> "$OUT/bench_${w}_sweep.err" || exit 1
	done
	echo "bench lines done" ;;
trace)
	for w in $WL; do
		rocprofv3 --kernel-trace --stats -d "$OUT/trace_$w" --output-format csv -- \
			python3 "$ROOT/bench.py" --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-configs > "$OUT/trace_$w.json" 2> "$OUT/trace_$w.log" || exit 1
		echo "trace $w done"
	done
	CLO_RADIX_SWEEP=1 rocprofv3 --kernel-trace --stats -d "$OUT/trace_satradix_u32_sweep" --output-format csv -- \
		python3 "$ROOT/bench.py" --workload satradix_u32 --steps 10 --warmup 2 --no-cpu-baseline --no-configs > "$OUT/trace_satradix_u32_sweep.json" 2> "$OUT/trace_satradix_u32_sweep.log" || exit 1
	echo "traces done" ;;
pmc)
	# HBM traffic: one counter per run, kernel trace only (never combined with other trace domains)
	for w in $WL; do
		for c in FETCH_SIZE WRITE_SIZE; do
			rocprofv3 --pmc $c --kernel-trace -d "$OUT/pmc_${c}_$w" --output-format csv -- \
				python3 "$ROOT/bench.py" --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-configs > "$OUT/pmc_${c}_$w.json" 2> "$OUT/pmc_${c}_$w.log" || exit 1
		done
		echo "pmc $w done"
	done
	for c in FETCH_SIZE WRITE_SIZE; do
		CLO_RADIX_SWEEP=1 rocprofv3 --pmc $c --kernel-trace -d "$OUT/pmc_${c}_satradix_u32_sweep" --output-format csv -- \
			python3 "$ROOT/bench.py" --workload satradix_u32 --steps 2 --warmup 1 --no-cpu-baseline --no-configs > "$OUT/pmc_${c}_satradix_u32_sweep.json" 2> "$OUT/pmc_${c}_satradix_u32_sweep.log" || exit 1
	done
	echo "pmc done" ;;
harness)
	# the reference harness's own sweeps (host data in, exec-queue device time, 5 runs per size)
	B="$ROOT/benchmarks/bin"
	"$B/clo_hip_sort_bench" -a satradix -t uint -n 28 -r 5 > "$OUT/harness_satradix.txt" 2>&1 || exit 1
	"$B/clo_hip_sort_bench" -a satradix -t ulong -n 27 -r 5 > "$OUT/harness_satradix_ulong.txt" 2>&1 || exit 1
	"$B/clo_hip_sort_bench" -a abitonic -t uint -n 26 -r 5 > "$OUT/harness_abitonic.txt" 2>&1 || exit 1
	"$B/clo_hip_sort_bench" -a sbitonic -t uint -n 20 -r 5 > "$OUT/harness_sbitonic.txt" 2>&1 || exit 1
	"$B/clo_hip_sort_bench" -a gselect -t uint -n 16 -r 5 > "$OUT/harness_gselect.txt" 2>&1 || exit 1
	"$B/clo_hip_scan_bench" -t uint -y uint -n 27 -r 5 > "$OUT/harness_scan.txt" 2>&1 || exit 1
	echo "harness sweeps done" ;;
probes)
	python3 "$ROOT/tools/sweep_sizes.py" big > "$OUT/sweep_sizes_big.txt" 2>&1 || exit 1
	python3 "$ROOT/tools/mid_probe.py" uint 22 28 > "$OUT/mid_probe.txt" 2>&1 || exit 1
	python3 "$ROOT/tools/skew_probe.py" 28 u32 > "$OUT/skew_probe.txt" 2>&1 || exit 1
	python3 "$ROOT/tools/hostsort_pipe_probe.py" 24 26 28 > "$OUT/hostsort_pipeline.txt" 2>&1 || exit 1
	python3 "$ROOT/tools/jit_probe.py" > "$OUT/jit_probe.txt" 2>&1 || exit 1
	python3 "$ROOT/tools/scan_sizes_probe.py" > "$OUT/scan_sizes_probe.txt" 2>&1 || exit 1
	echo "probes done" ;;
shard)
	python3 "$ROOT/tools/shard_alone_probe.py" 28 both > "$OUT/shard_alone_probe.txt" 2>&1 || exit 1
	python3 "$ROOT/tools/seg_probe.py" 28 uint 256 > "$OUT/seg_probe.txt" 2>&1 || exit 1
	python3 "$ROOT/tools/seg_probe.py" 28 ulong 256 >> "$OUT/seg_probe.txt" 2>&1 || exit 1
	echo "shard probes done" ;;
sq)
	# SQ counters of the kernels that ship (one --pmc pass per counter group, kernel trace only)
	bash "$ROOT/tools/pmc_busy.sh" "$TAG/sq_u32" satradix_u32 > /dev/null 2>&1 && cp "$ROOT/gpurun_out/$TAG/sq_u32/summary.txt" "$OUT/sq_counters_satradix_u32.txt"
	bash "$ROOT/tools/pmc_busy.sh" "$TAG/sq_u64" satradix_u64 > /dev/null 2>&1 && cp "$ROOT/gpurun_out/$TAG/sq_u64/summary.txt" "$OUT/sq_counters_satradix_u64.txt"
	echo "sq counters done" ;;
*) echo "unknown part $PART"; exit 2 ;;
esac
