"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every
symbol include/*.h declares; host-side logic (types, bit utilities, option and
get_key/compare parsing, introspection, error reporting) behaves like the
reference's. No compute calls: there is no GPU here (an "offline" context lets
objects be constructed; anything that would enqueue work fails loudly)."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

import cl_ops_amd as clo
from cl_ops_amd import _hip
from cl_ops_amd.api import lib, CLO_ERROR_ARGS, CLO_ERROR_IMPL_NOT_FOUND, CLO_ERROR_UNKNOWN_TYPE
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)          # comments
        text = re.sub(r"typedef struct [^{;]*\{.*?\}\s*\w+;", "", text, flags=re.S)  # vtables (function pointers)
        text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
        for m in re.finditer(r"\b((?:clo|ccl)_\w+)\s*\(", text):
            names.add(m.group(1))
    return names


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) > 100
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing
    for n in ("clo_sort_sbitonic_def", "clo_sort_abitonic_def", "clo_sort_satradix_def", "clo_scan_blelloch_def"):
        C.c_void_p.in_dll(lib, n)  # the plugin vtables are data symbols


def test_library_exports_nothing_but_the_declared_interface():
    """The other direction: every clo_* / ccl_* symbol the library exports is declared in include/*.h — the helpers
    shared by the host drivers (clo_internal.h) are hidden."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _hip.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.split() and re.match(r"(clo|ccl)_", l.split()[-1])}
    data = {"clo_sort_sbitonic_def", "clo_sort_abitonic_def", "clo_sort_satradix_def", "clo_sort_gselect_def", "clo_scan_blelloch_def"}
    extra = sorted(exported - declared_functions() - data)
    assert not extra, "exported but not declared in include/*.h: %s" % extra


def test_product_does_not_link_the_oracle():
    import subprocess
    out = subprocess.run(["nm", "-D", _hip.LIB_PATH], capture_output=True, text=True).stdout
    assert "clo_oracle" not in out
    src = ""
    for p in glob.glob(os.path.join(ROOT, "cl_ops_amd", "**", "*"), recursive=True):
        if os.path.isfile(p) and p.endswith((".py", ".c", ".h", ".hip")):
            src += open(p, errors="ignore").read()
    assert "oracle_lib" not in src and "libclo_oracle" not in src


def test_no_gpu_fails_loudly():
    if _hip.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(clo.CloError) as e:
        clo.Context(0)
    assert e.value.domain == "ccl-hip-error-quark"
    ctx = clo.Context(offline=True)
    with pytest.raises(clo.CloError):
        clo.Queue(ctx)
    with pytest.raises(clo.CloError):
        clo.Buffer(ctx, 1024)
    s = clo.Sorter("satradix", ctx, "uint")
    with pytest.raises(clo.CloError):       # the host-data path needs a queue -> a device
        s.with_host_data(np.arange(16, dtype=np.uint32))
    s.close()
    ctx.close()


def test_type_table_follows_clo_common():
    names = ["char", "uchar", "short", "ushort", "int", "uint", "long", "ulong", "half", "float", "double"]
    sizes = [1, 1, 2, 2, 4, 4, 8, 8, 2, 4, 8]
    for i, (n, s) in enumerate(zip(names, sizes)):
        assert lib.clo_type_get_name(i).decode() == n
        assert lib.clo_type_sizeof(i) == s
        assert lib.clo_type_by_name(n.encode(), None) == i
    err = clo.api._Err()
    assert lib.clo_type_by_name(b"quad", err.ref) == -1
    with pytest.raises(clo.CloError) as e:
        err.raise_if_set()
    assert e.value.code == CLO_ERROR_UNKNOWN_TYPE and "Unknown type 'quad'" in e.value.message
    assert e.value.domain == "clo-error-quark"


def test_bit_utilities_match_oracle():
    L = O.lib()
    for x in list(range(0, 70)) + [255, 256, 257, 1000, 65535, 65536, 1 << 30, (1 << 31) - 1, 1 << 31]:
        assert lib.clo_nlpo2(x) == L.clo_oracle_nlpo2(x), x
        assert lib.clo_ones32(x) == L.clo_oracle_ones32(x), x
    for k in range(31):
        assert lib.clo_tzc(1 << k) == k
    assert [lib.clo_sum(x) for x in (0, 1, 4, 100)] == [0, 1, 10, 5050]


@pytest.fixture(scope="module")
def off():
    ctx = clo.Context(offline=True)
    yield ctx
    ctx.close()


def test_sort_impl_dispatch(off):
    for alg in ("sbitonic", "abitonic", "gselect", "satradix"):   # CLO_SORT_IMPLS, clo_sort_abstract.in.h:30
        s = clo.Sorter(alg, off, "uint")
        assert s.element_size == 4 and s.key_size == 4
        s.close()
    for alg in ("quicksort", ""):
        with pytest.raises(clo.CloError) as e:
            clo.Sorter(alg, off, "uint")
        assert e.value.code == CLO_ERROR_IMPL_NOT_FOUND
        assert "was not found" in e.value.message
    with pytest.raises(clo.CloError) as e:
        clo.Scanner("hillis", off, "uint", "uint")
    assert e.value.code == CLO_ERROR_IMPL_NOT_FOUND


def test_kernel_names_are_the_reference_ones(off):
    s = clo.Sorter("sbitonic", off, "uint")
    assert s.num_kernels() == 1 and s.kernel_name(0) == "sbitonic"
    # round 5: the LDS of the tiled kernel `numel` selects (upstream's own kernel uses none and reports 0)
    assert s.localmem_usage(0, 0, 16) == 0 and s.localmem_usage(0, 0, 1 << 12) == (8192 + 256) * 4 and s.localmem_usage(0) == (16384 + 512) * 4
    s.close()
    s = clo.Sorter("gselect", off, "uint")
    assert s.num_kernels() == 1 and s.kernel_name(0) == "gselect"
    s.close()
    s = clo.Sorter("satradix", off, "uint")
    assert s.num_kernels() == 6
    assert [s.kernel_name(i) for i in range(6)] == ["satradix_localsort", "satradix_histogram", "satradix_scatter",
                                                    "workgroupScan", "workgroupSumsScan", "addWorkgroupSums"]
    assert s.localmem_usage(2) > 16 * 1024
    s.close()
    s = clo.Sorter("abitonic", off, "uint")
    assert s.num_kernels() == 26
    names = [s.kernel_name(i) for i in range(26)]
    assert names[0] == "abit_any" and names[1] == "abit_local_s2" and names[10] == "abit_local_s11"
    assert names[11:14] == ["abit_priv_2s4v", "abit_priv_3s8v", "abit_priv_4s16v"]
    assert names[25] == "abit_hyb_s12_4s16v"
    assert s.localmem_usage(0) == 0 and s.localmem_usage(13) == 0 and s.localmem_usage(1) > 0
    s.close()
    sc = clo.Scanner("blelloch", off, "uint", "ulong")
    assert sc.num_kernels() == 3
    assert [sc.kernel_name(i) for i in range(3)] == ["workgroupScan", "workgroupSumsScan", "addWorkgroupSums"]
    sc.close()


@pytest.mark.parametrize("opts,msg", [
    ("radix=12", "Radix must be a power of 2."),
    ("radix=512", "Radix must be between 2 and 256"),
    ("radix", "Invalid option 'radix' for satradix sort."),
    ("bogus=1", "Invalid option key 'bogus' for satradix sort."),
])
def test_satradix_option_errors(off, opts, msg):
    with pytest.raises(clo.CloError) as e:
        clo.Sorter("satradix", off, "uint", options=opts)
    assert e.value.code == CLO_ERROR_ARGS and msg in e.value.message


def test_satradix_options_accepted(off):
    for opts in (None, "", "radix=2", "radix=256", "radix=16,scan=blelloch", ",radix=4,"):
        clo.Sorter("satradix", off, "uint", options=opts).close()
    # Upstream hands the counters to a CloScan of this type at the first sort
    # (clo_sort_satradix.c:94,298) and fails there; here the counter scan is part of
    # the radix kernels, so type and options are checked at clo_sort_new, with the
    # scan API's own error codes and messages, instead of being silently unused.
    with pytest.raises(clo.CloError) as e:
        clo.Sorter("satradix", off, "uint", options="scan=nosuchscan")
    assert e.value.code == CLO_ERROR_IMPL_NOT_FOUND and "'nosuchscan'" in e.value.message
    with pytest.raises(clo.CloError) as e:
        clo.Sorter("satradix", off, "uint", options="scanfoo=1")  # forwarded to blelloch, which takes no options
    assert e.value.code == CLO_ERROR_ARGS and "Invalid options for blelloch scan." in e.value.message


def test_scan_impl_def_has_upstreams_layout():
    """CloScanImplDef is upstream's struct (clo_scan_abstract.in.h:41-103): a name and
    six function pointers, nothing appended — a plugin compiled against upstream's
    header must not be read past its end. The chunk / status extensions live in a
    private side table (clo_internal.h), which the public headers do not mention."""
    import re
    text = open(os.path.join(ROOT, "include", "clo_scan.h")).read()
    body = text[text.index("typedef struct clo_scan_impl_def {"):text.index("} CloScanImplDef;")]
    members = re.findall(r"\(\*(\w+)\)\s*\(", body)
    assert members == ["init", "finalize", "scan_with_device_data", "get_num_kernels", "get_kernel_name",
                       "get_localmem_usage"]
    assert "scan_chunk" not in text
    # the exported vtable is 7 pointers long: the word after it is not a pointer into the library's text
    base = C.addressof(C.c_void_p.in_dll(lib, "clo_scan_blelloch_def"))
    words = (C.c_void_p * 7).from_address(base)
    assert all(words[i] for i in range(7))
    assert C.string_at(words[0]) == b"blelloch"


@pytest.mark.parametrize("opts,msg", [
    ("minps=0", "Option 'minps' must be between 1 and 4."),
    ("maxps=5", "Option 'maxps' must be between 1 and 4."),
    ("minps=3,maxps=2", "'minps' (3) must be less or equal than 'maxps' (2)."),
    ("steps=3", "Invalid option key 'steps' for abitonic sort."),
    ("maxps", "Invalid option 'maxps' for abitonic sort."),
])
def test_abitonic_option_errors(off, opts, msg):
    with pytest.raises(clo.CloError) as e:
        clo.Sorter("abitonic", off, "uint", options=opts)
    assert e.value.code == CLO_ERROR_ARGS and msg in e.value.message


def test_abitonic_options_accepted_and_sbitonic_ignores_options(off):
    clo.Sorter("abitonic", off, "uint", options="minps=2,maxps=3,maxsfs=8").close()
    clo.Sorter("sbitonic", off, "uint", options="whatever").close()


def test_blelloch_rejects_options_and_bad_types(off):
    with pytest.raises(clo.CloError) as e:
        clo.Scanner("blelloch", off, "uint", "uint", options="x=1")
    assert e.value.code == CLO_ERROR_ARGS and "Invalid options for blelloch scan." in e.value.message
    # every pair of types is accepted, as by upstream's generic kernel (clo_scan_abstract.c:122-125): sums narrower
    # than the elements, half sums, floating-point elements into integer sums (round 3; refused before)
    for et, st in (("float", "float"), ("double", "double"), ("float", "double"), ("uint", "float"), ("half", "float"),
                   ("double", "float"), ("ulong", "uint"), ("float", "uint"), ("half", "half"), ("double", "ulong"), ("int", "uchar")):
        clo.Scanner("blelloch", off, et, st).close()
    with pytest.raises(clo.CloError) as e:
        clo.Scanner("blelloch", off, "uint", 42)
    assert e.value.code == clo.api.CLO_ERROR_UNKNOWN_TYPE
    from cl_ops_amd._hip import lib
    # which kernel family scans which pair (CloType numbers: uint 5, ulong 7, uchar 1, half 8, float 9)
    assert [lib.clo_hip_scan_is_typed(a, b) for a, b in ((5, 5), (5, 7), (7, 5), (9, 5), (5, 9), (8, 8), (1, 1), (4, 1))] == [0, 0, 1, 1, 1, 1, 0, 1]


@pytest.mark.parametrize("elem,key,get_key,expect", [
    ("uint", None, None, (4, 4, 0, 32, 0, 0)),
    ("uint", None, "(x)", (4, 4, 0, 32, 0, 0)),
    ("ulong", "uint", "(uint) ((x) >> 32)", (8, 4, 32, 32, 0, 0)),
    ("ulong", "uint", "(uint)(x)", (8, 4, 0, 32, 0, 0)),
    ("ulong", "uint", "((x) >> 16) & 0xFFFF", (8, 4, 16, 16, 0, 0)),
    ("uint", None, "((x) & 0xF)", (4, 4, 0, 4, 0, 0)),
    ("ulong", "ushort", "(ushort)((x) >> 48)", (8, 2, 48, 16, 0, 0)),
    ("int", None, None, (4, 4, 0, 32, 1, 0)),
    ("float", None, None, (4, 4, 0, 32, 2, 0)),
    ("ulong", None, "x >> 8 >> 8", (8, 8, 16, 48, 0, 0)),
])
def test_get_key_parsing(off, elem, key, get_key, expect):
    s = clo.Sorter("sbitonic", off, elem, key_type=key, get_key=get_key)
    assert s.key_spec() == expect
    s.close()


@pytest.mark.parametrize("get_key", ["(x) * 3", "((x) & 0xF0)", "y", "(x) >> 70", "(float)(x)", "((x) >> 4"])
def test_get_key_unsupported_forms_are_refused(off, get_key):
    with pytest.raises(clo.CloError) as e:
        clo.Sorter("satradix", off, "uint", get_key=get_key)
    assert e.value.code == CLO_ERROR_ARGS and "get_key" in e.value.message


def test_compare_parsing(off):
    for cmp_, desc in ((None, 0), ("((a) > (b))", 0), ("(a)>(b)", 0), ("((a) < (b))", 1), ("a < b", 1)):
        s = clo.Sorter("abitonic", off, "uint", compare=cmp_)
        assert s.key_spec()[5] == desc
        s.close()
    with pytest.raises(clo.CloError) as e:
        clo.Sorter("abitonic", off, "uint", compare="((a) >= (b))")
    assert e.value.code == CLO_ERROR_ARGS


def test_half_keys_are_ieee_keys_of_16_bits(off):
    for impl in ("sbitonic", "abitonic", "satradix"):
        s = clo.Sorter(impl, off, "half")
        assert s.key_spec() == (2, 2, 0, 16, 2, 0)
        s.close()


def test_float_key_inside_a_wider_element(off):
    s = clo.Sorter("satradix", off, "ulong", key_type="float", get_key="as_float((uint) ((x) >> 32))")
    assert s.key_spec() == (8, 4, 32, 32, 2, 0)
    s.close()
    with pytest.raises(clo.CloError):   # half of a float is not an IEEE key
        clo.Sorter("satradix", off, "ulong", key_type="float", get_key="as_float((uint) ((x) >> 48))")


def test_workspace_sizes_are_sane():
    w = lib.clo_hip_radix_workspace_bytes(1 << 28, 4, 32, 4)
    assert (1 << 28) + (1 << 20) < w < (1 << 28) + (1 << 26)   # counters + one byte per element (the digit stream of big arrays)
    assert 1 << 20 < lib.clo_hip_radix_workspace_bytes(1 << 24, 4, 32, 4) < 1 << 24   # below 256 MiB: counters only
    assert lib.clo_hip_scan_workspace_bytes(1 << 26, 4, 4) < 1 << 20
    assert lib.clo_hip_bitonic_padded_numel(1000) == 1024
    assert lib.clo_hip_error_string(-4).decode().startswith("clo_hip")


def test_shard_plan_matches_the_python_plan():
    """clo_shard_plan (C, include/clo_shard.h) == ShardedSorter.plan (cl_ops_amd/multigpu.py)."""
    from cl_ops_amd.multigpu import ShardedSorter
    rng = np.random.default_rng(0)
    for world in (1, 2, 4, 8):
        m = rng.integers(0, 1000, (world, world), dtype=np.uint64)
        arr = C.c_size_t * world
        for r in range(world):
            a, b, c, d = arr(), arr(), arr(), arr()
            lib.clo_shard_plan(np.ascontiguousarray(m).ctypes.data_as(C.POINTER(C.c_uint64)), world, r, a, b, c, d)
            sc, so, rc, ro = ShardedSorter.plan(m.astype(np.int64), r)
            assert (list(a), list(b), list(c), list(d)) == (list(sc), list(so), list(rc), list(ro))


def test_shard_sort_argument_errors(off):
    tr = clo.ShardTransport.custom(0, 3, lambda *a: 0, lambda *a: 0)
    with pytest.raises(clo.CloError) as e:
        clo.ShardSort(off, tr, "uint")
    assert e.value.code == CLO_ERROR_ARGS and "world size" in e.value.message
    tr = clo.ShardTransport.custom(0, 2, lambda *a: 0, lambda *a: 0)
    with pytest.raises(clo.CloError) as e:
        clo.ShardSort(off, tr, "float")
    assert e.value.code == CLO_ERROR_ARGS
    clo.ShardSort(off, tr, "ulong", options="radix=256").close()
