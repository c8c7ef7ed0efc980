#!/bin/bash
# CPU-only: rebuilds the C host layer and the oracle with AddressSanitizer +
# UndefinedBehaviorSanitizer (the HIP objects are linked as built), runs the CPU
# test files that exercise them (option / get_key / compare parsing, error paths,
# the oracle algorithms) and counts sanitizer reports. GPU ASan is not available
# on the pool; device code is not covered by this.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SAN=$(mktemp -d)
trap 'cp "$SAN/orig_hip.so" "$ROOT/cl_ops_amd/lib/libcl_ops_hip.so"; cp "$SAN/orig_oracle.so" "$ROOT/oracle/libclo_oracle.so"; rm -rf "$SAN"' EXIT
make -C "$ROOT/cl_ops_amd/csrc" -j4 > /dev/null
make -C "$ROOT/oracle" > /dev/null
cp "$ROOT/cl_ops_amd/lib/libcl_ops_hip.so" "$SAN/orig_hip.so"
cp "$ROOT/oracle/libclo_oracle.so" "$SAN/orig_oracle.so"
INC="-I$ROOT/include -I$ROOT/cl_ops_amd/csrc -I$ROOT/cl_ops_amd/csrc/hip"
for f in "$ROOT"/cl_ops_amd/csrc/*.c; do
	gcc -O1 -g -std=c11 -fPIC -Wall -Wextra $INC -D_GNU_SOURCE -fsanitize=address,undefined -fno-omit-frame-pointer -c "$f" -o "$SAN/$(basename "$f").o"
done
ASAN_LIB=$(gcc -print-file-name=libasan.so)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/cl_ops_amd/lib/libcl_ops_hip.so" \
	"$ROOT"/cl_ops_amd/csrc/build/*.hip.o "$SAN"/*.c.o -lhiprtc -ldl -L"$(dirname "$ASAN_LIB")" -lasan -lubsan
gcc -O1 -g -std=c11 -fPIC -fopenmp -fsanitize=address,undefined -shared -o "$ROOT/oracle/libclo_oracle.so" "$ROOT/oracle/clo_oracle.c"
cd "$ROOT"
LD_PRELOAD=$ASAN_LIB ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
	python -m pytest tests/test_boundary_cpu.py tests/test_oracle.py -x -q -s 2>&1 | tee "$SAN/log" | tail -2
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' "$SAN/log" || true)"
