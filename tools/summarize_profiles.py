"""Turns gpurun_out/<tag>/ (made by tools/collect_profiles.sh on the GPU box) into the
tracked files under profiles/: bench JSON lines, rocprofv3 kernel stats, and the
PMC-derived HBM traffic per launch of the headline kernel (FETCH_SIZE x2 per the
gfx950 note in MI355X_MICROARCH.md, calibrated on the histogram kernel of the same run).
usage: python tools/summarize_profiles.py r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")

for f in glob.glob(os.path.join(src, "bench_*.json")):
    line = [l for l in open(f).read().splitlines() if l.startswith("{")][-1]
    open(os.path.join(dst, "%s_%s" % (tag, os.path.basename(f))), "w").write(line + "\n")

names = {"satradix_u32": "satradix_u32_2p28", "satradix_pairs": "satradix_pairs", "satradix_u64": "satradix_u64", "scan": "scan",
         "abitonic": "abitonic", "satradix_u32_sweep": "satradix_u32_2p28_sweep"}
for w, out in names.items():
    st = glob.glob(os.path.join(src, "trace_" + w, "*", "*kernel_stats.csv"))
    if st:
        shutil.copy(max(st, key=os.path.getmtime), os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, out)))


def pmc(counter, workload):
    f = max(glob.glob(os.path.join(src, "pmc_%s_%s" % (counter, workload), "*", "*counter_collection.csv")), key=os.path.getmtime)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}


def short(k):
    k = k.replace("void ", "").replace("(anonymous namespace)::", "")
    return k.split("(")[0].strip()


# kernel name (as rocprofv3 prints it) -> the family label bench.py times it under
FAMILY_OF = (("pair_kernel", "radix_pass"), ("tilehist", "radix_hist"), ("chunksum", "radix_offsets"),
             ("chunkscan", "radix_offsets"), ("offsets", "radix_offsets"), ("ghist", "radix_ghist"), ("sweep", "radix_sweep"),
             ("clo_scan_kernel", "scan"), ("tile_merge", "bitonic_tile"), ("presort", "bitonic_presort"),
             ("strided2", "bitonic_strided2"), ("strided", "bitonic_strided"), ("step_kernel", "bitonic_step"))


def families(rows, launches_of):
    """HBM bytes per launch of each kernel family. Template instances of ONE kernel are
    alternatives (averaged over their launches); DIFFERENT kernels of a family run once each
    per family launch (the three counter-scan kernels) and add up."""
    acc = {}
    for name, n, f, w, b in rows:
        fam = next((lab for needle, lab in FAMILY_OF if needle in name), None)
        if fam is None:
            continue
        base = name.split("<")[0].replace("_bytes", "")   # (the two histogram kernels are alternatives: first pass / later passes)
        d = acc.setdefault(fam, {}).setdefault(base, {"kernels": [], "bytes_total": 0, "launches": 0})
        d["kernels"].append(name)
        d["bytes_total"] += b * n
        d["launches"] += n
    out = {}
    for fam, bases in acc.items():
        out[fam] = {"kernels": [k for d in bases.values() for k in d["kernels"]],
                    "hbm_bytes_per_launch": int(sum(d["bytes_total"] / d["launches"] for d in bases.values() if d["launches"]))}
    return out


def traffic_rows(workload):
    fetch, write = pmc("FETCH_SIZE", workload), pmc("WRITE_SIZE", workload)
    rows = []
    for k in fetch:
        n, f = fetch[k]
        w = write.get(k, (0, 0.0))[1]
        rows.append((short(k), n, f, w, int((2 * f + w) * 1024)))
    rows.sort(key=lambda r: -r[4] * r[1])
    return rows


HEADER = ("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate runs, each with --kernel-trace only) of\n"
          "#   python3 bench.py --workload %s --steps 2 --warmup 1 --no-cpu-baseline, MI355X\n"
          "# Counter unit: KB. gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B requests as 64 B, i.e.\n"
          "# reports 1/2 of a coalesced streaming read. %s WRITE_SIZE is taken as is.\n"
          "kernel,launches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_bytes_per_launch(2*FETCH+WRITE)\n")

# ---- headline: satradix, 2^28 uint32 keys ----
rows = traffic_rows("satradix_u32")
hist = [r for r in rows if "tilehist" in r[0] and "unsigned int" in r[0]]
pas = [r for r in rows if "pair_kernel" in r[0] and "unsigned int" in r[0]]
n_keys = 1 << 28
calib = (n_keys * 4 / 1024) / hist[0][2] if hist else float("nan")
path = os.path.join(dst, "%s_satradix_u32_2p28_pmc_hbm_traffic.csv" % tag)
with open(path, "w") as o:
    o.write(HEADER % ("satradix_u32", "Calibration in this very run: the histogram kernel reads exactly 2^28*4 B = 1048576 KB "
                                      "and reports %.0f KB -> factor %.3f (2.0 used)." % (hist[0][2] if hist else 0, calib)))
    for r in rows:
        o.write("%s,%d,%.0f,%.0f,%d\n" % r)
if pas:
    json.dump({"kernel": pas[0][0], "hbm_bytes_per_launch": pas[0][4], "fetch_size_kb": pas[0][2], "write_size_kb": pas[0][3],
               "log2n": 28, "families": families(rows, None),
               "correction": "FETCH_SIZE x2 (gfx950; calibrated on the histogram kernel in the same run: factor %.3f), WRITE_SIZE as is" % calib,
               "source": "profiles/" + os.path.basename(path)},
              open(os.path.join(dst, "traffic_satradix_u32.json"), "w"), indent=1)
# the single-sweep passes (CLO_RADIX_SWEEP=1 runs): own table, families merged into the same JSON
try:
    srows = traffic_rows("satradix_u32_sweep")
    spath = os.path.join(dst, "%s_satradix_u32_2p28_sweep_pmc_hbm_traffic.csv" % tag)
    with open(spath, "w") as o:
        o.write(HEADER % ("satradix_u32 (CLO_RADIX_SWEEP=1)", "(Same x2 as calibrated on the histogram kernel of the chain-free run.)"))
        for r in srows:
            o.write("%s,%d,%.0f,%.0f,%d\n" % r)
    tj = os.path.join(dst, "traffic_satradix_u32.json")
    d = json.load(open(tj))
    d["families"].update(families(srows, None))
    json.dump(d, open(tj, "w"), indent=1)
    for r in srows[:4]:
        print("sweep", r)
except ValueError:
    pass
for extra in ("sweep_sizes.txt", "sweep_probe_u32.txt", "sweep_probe_u64.txt"):
    f = os.path.join(src, extra)
    if os.path.exists(f):
        open(os.path.join(dst, "%s_%s" % (tag, extra)), "w").write("".join(l for l in open(f) if "amdgpu.ids" not in l))
print("profiles/ refreshed from", src)
for r in rows[:6]:
    print(r)

# ---- the 8-byte satradix workloads: key/value pairs (config 4) and uint64 keys (config 5's shard) ----
for workload in ("satradix_pairs", "satradix_u64"):
    try:
        rows8 = traffic_rows(workload)
    except ValueError:
        continue
    path8 = os.path.join(dst, "%s_%s_pmc_hbm_traffic.csv" % (tag, workload))
    with open(path8, "w") as o:
        o.write(HEADER % (workload, "(Same x2 as calibrated on the histogram kernel of the uint32 run.)"))
        for r in rows8:
            o.write("%s,%d,%.0f,%.0f,%d\n" % r)
    pas8 = [r for r in rows8 if "pair_kernel" in r[0]]
    if pas8:
        d8 = max(pas8, key=lambda r: r[1])
        json.dump({"kernel": d8[0], "hbm_bytes_per_launch": d8[4], "fetch_size_kb": d8[2], "write_size_kb": d8[3],
                   "log2n": 28, "families": families(rows8, None),
                   "correction": "FETCH_SIZE x2 (gfx950; calibrated on the histogram kernel of the uint32 run), WRITE_SIZE as is",
                   "source": "profiles/" + os.path.basename(path8)},
                  open(os.path.join(dst, "traffic_%s.json" % workload), "w"), indent=1)
    for r in rows8[:4]:
        print(workload, r)

# ---- SQ counters of the kernels that ship (tools/pmc_busy.sh) and the other probes of the collection ----
for extra in ("sq_counters_satradix_u32.txt", "sq_counters_satradix_u64.txt", "skew_probe.txt", "skew_probe_u64.txt",
              "hostsort_pipeline.txt", "size_sweep.txt", "sweep_sizes_big.txt", "mid_probe.txt", "shard_alone_probe.txt", "seg_probe.txt",
              "bench_headline_with_configs.json"):
    f = os.path.join(src, extra)
    if os.path.exists(f):
        open(os.path.join(dst, "%s_%s" % (tag, extra)), "w").write("".join(
            l for l in open(f) if "amdgpu.ids" not in l and not l.startswith(("RCCL version", "HIP version", "ROCm version", "Hostname", "Librccl path"))))

# ---- scan and abitonic: the dominant kernel of each ----
for workload, needle in (("scan", "clo_scan_kernel"), ("abitonic", "tile_merge_kernel")):
    try:
        rows = traffic_rows(workload)
    except ValueError:
        continue   # no PMC pass for this workload in gpurun_out/<tag>
    path = os.path.join(dst, "%s_%s_pmc_hbm_traffic.csv" % (tag, workload))
    with open(path, "w") as o:
        o.write(HEADER % (workload, "(Same x2 as calibrated on the satradix histogram kernel.)"))
        for r in rows:
            o.write("%s,%d,%.0f,%.0f,%d\n" % r)
    dom = [r for r in rows if needle in r[0]]
    if dom:
        d = max(dom, key=lambda r: r[1])   # the merge launches outnumber the presort
        json.dump({"kernel": d[0], "hbm_bytes_per_launch": d[4], "fetch_size_kb": d[2], "write_size_kb": d[3],
                   "log2n": 26, "families": families(rows, None),
                   "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE as is", "source": "profiles/" + os.path.basename(path)},
                  open(os.path.join(dst, "traffic_%s.json" % workload), "w"), indent=1)
    for r in rows[:4]:
        print(workload, r)

# ---- the C harnesses' sweeps ----
NOTES = {
    "satradix": "benchmarks/bin/clo_hip_sort_bench -a satradix -t uint -n 28 -r 5",
    "satradix_ulong": "benchmarks/bin/clo_hip_sort_bench -a satradix -t ulong -n 27 -r 5",
    "abitonic": "benchmarks/bin/clo_hip_sort_bench -a abitonic -t uint -n 26 -r 5",
    "sbitonic": "benchmarks/bin/clo_hip_sort_bench -a sbitonic -t uint -n 20 -r 5",
    "gselect": "benchmarks/bin/clo_hip_sort_bench -a gselect -t uint -n 16 -r 5",
    "scan": "benchmarks/bin/clo_hip_scan_bench -t uint -y uint -n 27 -r 5",
}
for alg, cmd in NOTES.items():
    f = os.path.join(src, "harness_%s.txt" % alg)
    if not os.path.exists(f):
        continue
    with open(os.path.join(dst, "%s_harness_sweep_%s.txt" % (tag, alg)), "w") as o:
        o.write("# %s on one MI355X (the reference harness's CLI and conventions: host data in, exec-queue device time "
                "only, 5 runs per size; every run follows a fresh host-to-device copy, so these are cold-clock single-call "
                "numbers, below bench.py's back-to-back steps). The scan pipelines chunks of 2^22..2^24 elements from 2^24 elements on: "
                "its device time is the sum of the chunk scans.\n" % cmd)
        o.write(open(f).read())

# ---- what bench.py's live guard compares a later run with: the kernel sources the PMC passes ran (content hash printed by
# bench.py itself on the GPU box), the commit this summary was made at, and the launches per step of every kernel family
# as bench.py's own timing leg counted them in the PMC run ----
import subprocess
try:
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except OSError:
    head = None
for workload in ("satradix_u32", "satradix_pairs", "satradix_u64", "scan", "abitonic"):
    tj = os.path.join(dst, "traffic_%s.json" % workload)
    bj = os.path.join(src, "pmc_FETCH_SIZE_%s.json" % workload)
    if not (os.path.exists(tj) and os.path.exists(bj)):
        continue
    lines = [l for l in open(bj).read().splitlines() if l.startswith("{")]
    if not lines:
        continue
    b = json.loads(lines[-1])
    d = json.load(open(tj))
    d["kernels_sha16"] = (b.get("live_guard") or {}).get("kernels_sha16")
    d["source_head"] = head
    for k in (b.get("roofline") or {}).get("kernels", []):
        if k["name"] in d["families"]:
            d["families"][k["name"]]["launches_per_step"] = k["launches_per_step"]
    sj = os.path.join(src, "pmc_FETCH_SIZE_%s_sweep.json" % workload)   # the forced single-sweep path, where it was collected
    if os.path.exists(sj):
        lines = [l for l in open(sj).read().splitlines() if l.startswith("{")]
        if lines:
            for k in (json.loads(lines[-1]).get("roofline") or {}).get("kernels", []):
                if k["name"] in d["families"]:
                    d["families"][k["name"]]["sweep_launches_per_step"] = k["launches_per_step"]
    json.dump(d, open(tj, "w"), indent=1)
