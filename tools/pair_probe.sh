#!/bin/bash
# Pair-kernel experiments: one bench line per environment variant (the switches are read once per process).
# usage: tools/pair_probe.sh OUTDIR "VAR=VAL ..." "VAR=VAL ..." ...
out=$1; shift
mkdir -p "$out"
for wl in satradix_u32 satradix_u64; do
  for v in "$@"; do
    tag=$(echo "$v" | tr ' =/' '___')
    env $v python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-configs > "$out/${wl}_${tag}.json" 2> "$out/${wl}_${tag}.err" || { echo "FAILED $wl $v"; tail -5 "$out/${wl}_${tag}.err"; exit 1; }
    python - "$out/${wl}_${tag}.json" "$wl" "$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ks = {k["name"]: k["avg_launch_ms"] for k in d["roofline"]["kernels"]}
print(f"{sys.argv[2]:14s} {sys.argv[3]:32s} {d['ms_per_step']:8.4f} ms/step  {ks}", flush=True)
PY
  done
done
