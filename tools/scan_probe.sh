#!/bin/bash
# scan bench under environment variants: tools/scan_probe.sh OUTDIR "VAR=VAL ..." ...
out=$1; shift
mkdir -p "$out"
for v in "$@"; do
  tag=$(echo "$v" | tr ' =/' '___')
  for rep in 1 2; do
    env $v python bench.py --workload scan --steps 200 --warmup 20 --no-cpu-baseline > "$out/scan_${tag}_$rep.json" 2> "$out/scan_${tag}_$rep.err" || { echo "FAILED $v"; tail -5 "$out/scan_${tag}_$rep.err"; exit 1; }
    python - "$out/scan_${tag}_$rep.json" "$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ks = {k["name"]: k["avg_launch_ms"] for k in d["roofline"]["kernels"]}
print(f"scan {sys.argv[2]:32s} {d['value']:10.0f} {d['unit']}  {d['ms_per_step']:8.4f} ms/step  {ks}", flush=True)
PY
  done
done
