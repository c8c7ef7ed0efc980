#!/bin/bash
# Runs ON THE GPU BOX: a list of steps, each logged to gpurun_out/$TAG/<name>.log, with a
# marker line on stdout per step (so a long call never looks silent). A step that is
# killed by its timeout ends the call (no further GPU work after a hang).
#   gpurun -- 'bash tools/gpu_steps.sh TAG "name|timeout_s|command" ...'
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
for spec in "$@"; do
	name=${spec%%|*}; rest=${spec#*|}; tmo=${rest%%|*}; cmd=${rest#*|}
	echo "== $name (limit ${tmo}s): $cmd"
	timeout -k 10 "$tmo" bash -c "$cmd" > "$OUT/$name.log" 2>&1
	rc=$?
	echo "== $name rc=$rc"; tail -n 6 "$OUT/$name.log"
	if [ $rc -ge 124 ]; then echo "== $name hit its limit: stopping"; exit $rc; fi
done
