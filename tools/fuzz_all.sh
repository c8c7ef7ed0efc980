#!/bin/bash
# The four GPU fuzz / soak tools in a row (one MI355X); prints a summary. Usage: bash tools/fuzz_all.sh [seed]
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
S=${1:-61}
echo "== fuzz_gpu"; python tools/fuzz_gpu.py 400 $S 2>&1 | tail -1 || exit 1
echo "== fuzz_gpu (two-tile merge at every stage)"; CLO_BITONIC_MERGE2=2 python tools/fuzz_gpu.py 200 $((S+1)) 2>&1 | tail -1 || exit 1
echo "== fuzz_keyfield"; python tools/fuzz_keyfield_gpu.py 300 $((S+2)) 2>&1 | tail -1 || exit 1
echo "== fuzz_scan"; python tools/fuzz_scan_gpu.py 300 $((S+3)) 2>&1 | tail -1 || exit 1
echo "== soak"; python tools/soak_gpu.py 120 $((S+4)) 2>&1 | tail -1 || exit 1
