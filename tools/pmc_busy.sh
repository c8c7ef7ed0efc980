#!/bin/bash
# Runs ON THE GPU BOX: SQ busy / instruction / LDS counters of one bench workload, one rocprofv3 --pmc
# pass per group of counters (never combined with other trace domains), summarised per kernel.
#   bash tools/pmc_busy.sh TAG WORKLOAD [env VAR=VAL ...]
TAG=${1:-pmcb}; WL=${2:-satradix_u32}; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do export "$v"; done
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS_ATOMIC"; do
	i=$((i + 1))
	rocprofv3 --pmc $set --kernel-trace -d "$OUT/set$i" --output-format csv -- \
		python3 "$ROOT/bench.py" --workload "$WL" --steps 2 --warmup 1 --no-cpu-baseline --no-configs > "$OUT/set$i.json" 2> "$OUT/set$i.log" || { echo "set $i failed"; tail -3 "$OUT/set$i.log"; exit 1; }
	echo "set $i done: $set"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/set*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
with open(out + "/summary.txt", "w") as o:
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", 0)):
        if "rocclr" in k: continue
        line = "%s\n" % k
        for c in sorted(acc[k]):
            line += "    %-28s %16.0f per launch (%d launches)\n" % (c, acc[k][c] / cnt[k][c], cnt[k][c])
        o.write(line); print(line, end="")
PY
