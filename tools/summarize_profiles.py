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
        # (the two histogram kernels are alternatives: first pass / later passes; so are the one- and the two-tile merge pass)
        base = name.split("<")[0].replace("_bytes", "").replace("merge2", "merge")
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
              "hostsort_pipeline.txt", "size_sweep.txt", "sweep_sizes_big.txt", "mid_probe.txt", "shard_alone_probe.txt", "seg_probe.txt", "jit_probe.txt", "scan_sizes_probe.txt",
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

# ---- round 5: ONE file per workload in which the roofline can be recomputed ----
# For every workload the trace part ran:  rocprofv3 --kernel-trace --stats -- python3 bench.py --workload W ...
# That one process produced BOTH the rocprofv3 per-kernel durations (kernel_stats.csv) and bench.py's own line
# (trace_W.json: HIP-event means over its extra instrumented steps, the step time under the profiler). Joined here with
# the PMC bytes per launch (separate --pmc passes, above) and the unprofiled bench line of the same collection.
def _line(path):
    try:
        return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
    except Exception:
        return None


def _latest(files):
    """The files of the LATEST collection only (gpurun_out/<tag>/ keeps what earlier calls of the same tag merged back)."""
    if not files:
        return []
    newest = max(os.path.getmtime(f) for f in files)
    return [f for f in files if newest - os.path.getmtime(f) < 300]


def _rocprof_families(workload):
    best, best_calls = None, -1
    for f in _latest(glob.glob(os.path.join(src, "trace_" + workload, "*", "*kernel_stats.csv"))):
        rows = list(csv.DictReader(open(f)))
        calls = sum(int(r["Calls"]) for r in rows)
        if calls > best_calls:
            best, best_calls = rows, calls
    fams = {}
    for r in best or []:
        name = short(r["Name"])
        fam = next((lab for needle, lab in FAMILY_OF if needle in name), None)
        if fam is None:
            continue
        d = fams.setdefault(fam, {"calls": 0, "total_ns": 0.0, "kernels": []})
        d["calls"] += int(r["Calls"])
        d["total_ns"] += float(r["TotalDurationNs"])
        d["kernels"].append({"name": name[:120], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 3)})
    return fams


def _rocprof_trace(workload, steps):
    """Per family, from the dispatch records of the same process (kernel_trace.csv, in dispatch order): the mean duration
    over the launches of the TIMED steps (kernels back to back) and over those of the INSTRUMENTED steps (the last `steps`
    steps: the very launches bench.py's event pairs bracket), and the timed steps' length first start -> last end."""
    best = None
    for f in _latest(glob.glob(os.path.join(src, "trace_" + workload, "*", "*kernel_trace.csv"))):
        rows = list(csv.DictReader(open(f)))
        if best is None or len(rows) > len(best):
            best = rows
    if not best:
        return {}, None
    best.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = {}
    for r in best:
        fam = next((lab for needle, lab in FAMILY_OF if needle in r["Kernel_Name"]), None)
        if fam:
            per.setdefault(fam, []).append(r)
    out = dict(per)
    return out, best


for w in ("satradix_u32", "satradix_pairs", "satradix_u64", "scan", "abitonic"):
    prof_line = _line(os.path.join(src, "trace_%s.json" % w))
    rp = _rocprof_families(w)
    if not prof_line or not rp:
        continue
    plain = _line(os.path.join(src, "bench_%s.json" % w))
    try:
        traffic = json.load(open(os.path.join(dst, "traffic_%s.json" % w))).get("families", {})
    except Exception:
        traffic = {}
    steps_profiled = prof_line["steps"]
    ev = {k["name"]: k for k in prof_line["roofline"].get("kernels", [])}
    tr, all_rows = _rocprof_trace(w, steps_profiled)
    fams = []
    timed_window = None
    for fam, d in sorted(rp.items(), key=lambda kv: -kv[1]["total_ns"]):
        e = ev.get(fam)
        b = (traffic.get(fam) or {}).get("hbm_bytes_per_launch") or (e or {}).get("bytes_per_launch")
        rp_ms = d["total_ns"] / d["calls"] / 1e6
        row = {"family": fam, "rocprof_calls": d["calls"], "rocprof_mean_ms": round(rp_ms, 5),
               "events_mean_ms_same_run": (e or {}).get("avg_launch_ms"),
               "pmc_bytes_per_launch": b,
               "frac_rocprof_all_launches": round(b / (rp_ms * 1e-3) / 8e12, 4) if b else None,
               "frac_events_same_run": round(b / (e["avg_launch_ms"] * 1e-3) / 8e12, 4) if b and e and e.get("avg_launch_ms") else None}
        rows = tr.get(fam)
        lps = int(round(e["launches_per_step"])) if e and e.get("launches_per_step") else None
        if rows and lps and len(rows) >= 2 * steps_profiled * lps:
            k = steps_profiled * lps
            dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            instr, timed = rows[-k:], rows[-2 * k:-k]
            row["rocprof_mean_ms_instrumented_launches"] = round(sum(map(dur, instr)) / k, 5)   # the launches the events bracket
            row["rocprof_mean_ms_timed_launches"] = round(sum(map(dur, timed)) / k, 5)          # the launches of the timed region
            row["rocprof_over_events_same_launches"] = round(row["rocprof_mean_ms_instrumented_launches"] / e["avg_launch_ms"], 4)
            if b:
                row["frac_rocprof_instrumented_launches"] = round(b / (row["rocprof_mean_ms_instrumented_launches"] * 1e-3) / 8e12, 4)
                row["frac_rocprof_timed_launches"] = round(b / (row["rocprof_mean_ms_timed_launches"] * 1e-3) / 8e12, 4)
            w0, w1 = int(timed[0]["Start_Timestamp"]), int(timed[-1]["End_Timestamp"])
            timed_window = (min(w0, timed_window[0]), max(w1, timed_window[1])) if timed_window else (w0, w1)
        row["kernels"] = d["kernels"]
        fams.append(row)
    dom = prof_line["roofline"].get("kernel")
    out = {"workload": w,
           "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload %s --steps %d --warmup %d --no-cpu-baseline --no-configs"
                      % (w, prof_line["steps"], prof_line["warmup"]),
           "how_to_read": "ONE process produced every duration here. rocprof_mean_ms: its kernel_stats.csv (all launches: warm-up, timed and "
                          "instrumented steps). bench.py first times `steps` steps with the kernels back to back (no events in between), then runs "
                          "`steps` more with a HIP-event pair around every launch: events_mean_ms_same_run. rocprof_mean_ms_instrumented_launches / "
                          "_timed_launches: rocprofv3's own dispatch records (kernel_trace.csv) of exactly those two groups of launches — the first is "
                          "the same launches the events bracket (rocprof_over_events_same_launches is their ratio), the second is what the kernels "
                          "take inside the timed region, where a kernel starts while its predecessor's stores are still draining. "
                          "pmc_bytes_per_launch: FETCH_SIZE x 2 + WRITE_SIZE of separate --pmc passes (profiles/traffic_%s.json). "
                          "frac = bytes / duration / 8 TB/s." % w,
           "dominant_kernel": dom,
           "line_frac_events_same_run": prof_line["roofline"].get("frac"),
           "frac_from_rocprof_same_launches": next((r.get("frac_rocprof_instrumented_launches") for r in fams if r["family"] == dom), None),
           "frac_from_rocprof_timed_launches": next((r.get("frac_rocprof_timed_launches") for r in fams if r["family"] == dom), None),
           "frac_from_rocprof_all_launches": next((r["frac_rocprof_all_launches"] for r in fams if r["family"] == dom), None),
           "ms_per_step_under_rocprofv3": prof_line.get("ms_per_step"),
           "ms_per_step_from_rocprof_dispatch_records_of_the_timed_steps": round((timed_window[1] - timed_window[0]) / 1e6 / steps_profiled, 5) if timed_window else None,
           "ms_per_step_unprofiled_same_box_collection": plain.get("ms_per_step") if plain else None,
           "line_frac_unprofiled": plain["roofline"].get("frac") if plain else None,
           "step_frac_unprofiled": plain["roofline"].get("step_frac") if plain else None,
           "steps_in_profiled_run": steps_profiled,
           "families": fams}
    json.dump(out, open(os.path.join(dst, "%s_roofline_%s.json" % (tag, w)), "w"), indent=1)
    print("roofline file:", w, "dominant", dom, "frac: events", out["line_frac_events_same_run"], "rocprof same launches", out["frac_from_rocprof_same_launches"],
          "rocprof timed launches", out["frac_from_rocprof_timed_launches"], "step ms bench/rocprof", out["ms_per_step_under_rocprofv3"], out["ms_per_step_from_rocprof_dispatch_records_of_the_timed_steps"])
