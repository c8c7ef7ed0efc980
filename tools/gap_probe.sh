#!/bin/bash
# GPU box: rocprofv3 --kernel-trace of back-to-back 2^LOG-key sorts, then tools/gap_probe.py
LOG=${1:-24}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/gap_$LOG
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT" --output-format csv -- python3 "$ROOT/bench.py" --log2n $LOG --steps 40 --warmup 5 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.log" || { tail -5 "$OUT/bench.log"; exit 1; }
python3 "$ROOT/tools/gap_probe.py" "$OUT" 12
