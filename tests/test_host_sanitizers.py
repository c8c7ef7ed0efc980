"""The C host layer's threaded and multi-rank paths under the sanitizers, on the CPU (no GPU, no hipcc): the host
drivers (cl_ops_amd/csrc/*.c) are compiled with gcc against tests/hoststub/clo_hip_stub.c — a host-memory
implementation of the thin C-ABI, test infrastructure only — and driven by tests/hoststub/host_paths_test.c:
the pipelined clo_sort_with_host_data / clo_scan_with_host_data (helper threads), and the sharded sort of
include/clo_shard.h with 1, 2, 4 and 8 ranks as threads over an in-memory transport (slices, pieces, growth,
ranks failing together). Once with AddressSanitizer + UBSan, once with ThreadSanitizer; a run must end with
exit code 0 and without a single sanitizer report."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the pipelines' size thresholds, shrunk so that a run takes seconds (the product's values: 2^24, 2^25, 2^24 elements, 256 MiB per rank)
SHRINK = ["-DSAT_PIPE_MIN_NUMEL=4096", "-DCLO_SCAN_PIPE_MIN_NUMEL=65536", "-DCLO_SCAN_PIPE_CHUNK_MAX=16384", "-DSHARD_SLICE_MIN_BYTES_PER_RANK=8192"]


def _build(tmp_path, name, flags):
    exe = str(tmp_path / name)
    srcs = sorted(glob.glob(os.path.join(ROOT, "cl_ops_amd", "csrc", "*.c"))) + sorted(glob.glob(os.path.join(ROOT, "tests", "hoststub", "*.c")))
    cmd = ["gcc", "-O1", "-g", "-std=c11", "-D_GNU_SOURCE", "-fno-omit-frame-pointer", "-w", *SHRINK, *flags,
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "cl_ops_amd", "csrc"), *srcs, "-lpthread", "-lm", "-o", exe]
    subprocess.check_call(cmd)
    return exe


@pytest.mark.parametrize("name,flags,needles", [
    ("host_asan", ["-fsanitize=address,undefined"], ("AddressSanitizer", "runtime error", "LeakSanitizer")),
    ("host_tsan", ["-fsanitize=thread"], ("ThreadSanitizer",)),
])
def test_host_paths_under_sanitizers(tmp_path, name, flags, needles):
    exe = _build(tmp_path, name, flags)
    env = dict(os.environ, CLO_NO_WARMUP="1", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=0")
    r = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=900, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "host paths ok" in r.stdout, out[-4000:]
    for n in needles:
        assert n not in out, out[-4000:]
    # the sharded sort's fuzz: random worlds (1 .. 8 ranks, empty ranks included), options and key distributions — fixed
    # bits, a handful of values, one range, ascending, all equal: whole buckets, sub-buckets and slices stay empty
    r = subprocess.run([exe, "fuzz", "150", "7"], capture_output=True, text=True, timeout=900, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "shard fuzz ok" in r.stdout, out[-4000:]
    for n in needles:
        assert n not in out, out[-4000:]


def test_the_stub_is_not_in_the_product():
    """The stub is test infrastructure: the product library is built from cl_ops_amd/csrc alone."""
    mk = open(os.path.join(ROOT, "cl_ops_amd", "csrc", "Makefile")).read()
    assert "hoststub" not in mk and "tests/" not in mk
