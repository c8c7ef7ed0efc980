#!/bin/bash
# Developer probe, runs ON THE GPU BOX: instruction-mix counters of one bench workload.
#   gpurun --timeout 600 -- 'bash tools/pmc_probe.sh abitonic'
set -o pipefail
W=${1:-abitonic}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$W
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
	i=$((i+1))
	rocprofv3 --pmc $set --kernel-trace -d "$OUT/p$i" --output-format csv -- \
		python3 "$ROOT/bench.py" --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-configs > "$OUT/p$i.json" 2> "$OUT/p$i.log" || exit 1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("<")[0].split("(")[0][-40:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in acc:
    print(k, {c: "%.4g" % (acc[k][c] / cnt[k][c]) for c in sorted(acc[k])}, "launches", max(cnt[k].values()))
PY
