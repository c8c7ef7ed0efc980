"""The C harnesses (benchmarks/clo_hip_{sort,scan}_bench.c) are plain C programs
written against include/cl_ops.h and linked with libcl_ops_hip.so: they prove
the drop-in boundary from C, with the reference harness's flags and checks."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "benchmarks", "bin")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "benchmarks"), "-s"])


def test_harnesses_build_and_fail_loudly_without_a_gpu():
    _build()
    for exe in ("clo_hip_sort_bench", "clo_hip_scan_bench"):
        path = os.path.join(BIN, exe)
        assert os.access(path, os.X_OK)
        h = subprocess.run([path, "--help"], capture_output=True, text=True)
        assert h.returncode == 0 and "--algorithm" in h.stdout
    from cl_ops_amd import _hip
    if _hip.device_count() == 0:
        r = subprocess.run([os.path.join(BIN, "clo_hip_sort_bench"), "-a", "satradix", "-n", "8"],
                           capture_output=True, text=True)
        assert r.returncode != 0 and "No HIP device available" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("alg,opts,typ", [("sbitonic", "", "uint"), ("abitonic", "", "uint"), ("satradix", "", "uint"),
                                          ("satradix", "radix=256", "ulong"), ("abitonic", "maxps=2", "int"),
                                          ("satradix", "", "ushort"), ("satradix", "radix=64", "int"),
                                          ("satradix", "", "float"), ("satradix", "", "long")])
def test_sort_harness_on_gpu(alg, opts, typ, tmp_path):
    _build()
    out = tmp_path / "ns.tsv"
    cmd = [os.path.join(BIN, "clo_hip_sort_bench"), "-a", alg, "-t", typ, "-n", "18", "-r", "2", "-s", "7", "-o", str(out)]
    if opts:
        cmd += ["-g", opts]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    lines = re.findall(r"- 2\^(\d+): ([0-9.]+) Mkeys/s(.*)", r.stdout)
    assert [int(l[0]) for l in lines] == list(range(4, 19))
    assert all(l[2].strip() == "" for l in lines), r.stdout          # no "(sort did not work)"
    assert all(float(l[1]) > 0 for l in lines)
    rows = [l.split("\t") for l in out.read_text().strip().split("\n")]
    assert len(rows) == 15 and all(len(x) == 3 and int(x[1]) > 0 for x in rows)


@pytest.mark.gpu
@pytest.mark.parametrize("types", [("uint", "ulong"), ("uint", "uint"), ("uchar", "uint")])
def test_scan_harness_on_gpu(types):
    _build()
    r = subprocess.run([os.path.join(BIN, "clo_hip_scan_bench"), "-t", types[0], "-y", types[1], "-n", "20", "-r", "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    lines = re.findall(r"-\s+(\d+) : ([0-9.]+) MValues/s(.*)", r.stdout)
    assert [int(l[0]) for l in lines] == [4 << k for k in range(20)]
    assert all(l[2].strip() in ("", "[Overflow]") for l in lines), r.stdout


@pytest.mark.gpu
def test_harness_reports_api_errors():
    _build()
    r = subprocess.run([os.path.join(BIN, "clo_hip_sort_bench"), "-a", "quicksort", "-n", "8"], capture_output=True, text=True)
    assert r.returncode == 5 and "was not found" in r.stderr       # CLO_ERROR_IMPL_NOT_FOUND
    r = subprocess.run([os.path.join(BIN, "clo_hip_sort_bench"), "-a", "satradix", "-g", "radix=12", "-n", "8"],
                       capture_output=True, text=True)
    assert r.returncode == 2 and "Radix must be a power of 2." in r.stderr
