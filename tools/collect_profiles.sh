#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): bench lines, rocprofv3 kernel stats and the two
# PMC passes for the HBM traffic of the headline workload. Raw output goes to
# gpurun_out/$TAG/; tools/summarize_profiles.py turns it into the files under profiles/.
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh r03'   (parts: 'benchonly', 'pmc', 'probes' as 2nd argument run a part)
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for w in satradix_u32 satradix_pairs satradix_u64 scan abitonic sbitonic; do
	python3 "$ROOT/bench.py" --workload $w --steps 30 --warmup 3 --no-configs > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err" || exit 1
	echo "bench $w done"
done
# the other path at the headline sizes (the library's choice there: chain-free pair passes): single-sweep passes forced on
for w in satradix_u32 satradix_u64 satradix_pairs; do
	CLO_RADIX_SWEEP=1 python3 "$ROOT/bench.py" --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-configs > "$OUT/bench_${w}_sweep.json" 2> "$OUT/bench_${w}_sweep.err" || exit 1
done
[ "$2" = "benchonly" ] && { echo "bench lines done"; exit 0; }
echo "bench sweep done"
export CLO_RADIX_SWEEP=1
rocprofv3 --kernel-trace --stats -d "$OUT/trace_satradix_u32_sweep" --output-format csv -- \
	python3 "$ROOT/bench.py" --workload satradix_u32 --steps 10 --warmup 2 --no-cpu-baseline --no-configs > "$OUT/trace_satradix_u32_sweep.json" 2> "$OUT/trace_satradix_u32_sweep.log" || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
	rocprofv3 --pmc $c --kernel-trace -d "$OUT/pmc_${c}_satradix_u32_sweep" --output-format csv -- \
		python3 "$ROOT/bench.py" --workload satradix_u32 --steps 2 --warmup 1 --no-cpu-baseline --no-configs > "$OUT/pmc_${c}_satradix_u32_sweep.json" 2> "$OUT/pmc_${c}_satradix_u32_sweep.log" || exit 1
done
unset CLO_RADIX_SWEEP
echo "trace + pmc sweep done"
for w in satradix_u32 satradix_pairs satradix_u64 scan abitonic; do
	rocprofv3 --kernel-trace --stats -d "$OUT/trace_$w" --output-format csv -- \
		python3 "$ROOT/bench.py" --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-configs > "$OUT/trace_$w.json" 2> "$OUT/trace_$w.log" || exit 1
	echo "trace $w done"
done
# HBM traffic (headline workload first): one counter per run, kernel trace only
for w in satradix_u32 satradix_pairs satradix_u64 scan abitonic; do
	for c in FETCH_SIZE WRITE_SIZE; do
		rocprofv3 --pmc $c --kernel-trace -d "$OUT/pmc_${c}_$w" --output-format csv -- \
			python3 "$ROOT/bench.py" --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-configs > "$OUT/pmc_${c}_$w.json" 2> "$OUT/pmc_${c}_$w.log" || exit 1
	done
	echo "pmc $w done"
done
# the reference harness's own sweeps (host data in, exec-queue device time, 5 runs per size)
B="$ROOT/benchmarks/bin"
"$B/clo_hip_sort_bench" -a satradix -t uint -n 28 -r 5 > "$OUT/harness_satradix.txt" 2>&1 || exit 1
"$B/clo_hip_sort_bench" -a abitonic -t uint -n 26 -r 5 > "$OUT/harness_abitonic.txt" 2>&1 || exit 1
"$B/clo_hip_sort_bench" -a sbitonic -t uint -n 20 -r 5 > "$OUT/harness_sbitonic.txt" 2>&1 || exit 1
"$B/clo_hip_sort_bench" -a gselect -t uint -n 16 -r 5 > "$OUT/harness_gselect.txt" 2>&1 || exit 1
"$B/clo_hip_scan_bench" -t uint -y uint -n 27 -r 5 > "$OUT/harness_scan.txt" 2>&1 || exit 1
echo "harness sweeps done"
python3 "$ROOT/tools/sweep_sizes.py" > "$OUT/sweep_sizes.txt" 2>&1 || exit 1
python3 "$ROOT/tools/sweep_sizes.py" big > "$OUT/sweep_sizes_big.txt" 2>&1 || exit 1
python3 "$ROOT/tools/size_sweep.py" > "$OUT/size_sweep.txt" 2>&1 || exit 1
python3 "$ROOT/tools/mid_probe.py" uint 22 28 > "$OUT/mid_probe.txt" 2>&1 || exit 1
echo "size sweeps done"
python3 "$ROOT/tools/skew_probe.py" 28 u32 > "$OUT/skew_probe.txt" 2>&1 || exit 1
python3 "$ROOT/tools/skew_probe.py" 28 u64 > "$OUT/skew_probe_u64.txt" 2>&1 || exit 1
echo "skew probes done"
python3 "$ROOT/tools/hostsort_pipe_probe.py" 24 26 28 > "$OUT/hostsort_pipeline.txt" 2>&1 || exit 1
echo "host pipeline probe done"
# SQ counters of the kernels that ship (one --pmc pass per counter group, kernel trace only)
bash "$ROOT/tools/pmc_busy.sh" "$TAG/sq_u32" satradix_u32 > /dev/null 2>&1 && cp "$ROOT/gpurun_out/$TAG/sq_u32/summary.txt" "$OUT/sq_counters_satradix_u32.txt"
bash "$ROOT/tools/pmc_busy.sh" "$TAG/sq_u64" satradix_u64 > /dev/null 2>&1 && cp "$ROOT/gpurun_out/$TAG/sq_u64/summary.txt" "$OUT/sq_counters_satradix_u64.txt"
echo "sq counters done"
