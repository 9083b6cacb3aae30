#!/usr/bin/env python3
"""gpurun_out/<tag>/ (tools/collect_profiles.sh) -> profiles/<tag>_*: the kernel-stats tables, the per-launch HBM
traffic of the sweep (FETCH_SIZE x 1024 x read factor, WRITE_SIZE x 1024; factors calibrated in round 1 on a
pure-streaming launch of this kernel's access widths, profiles/r01_pmc_traffic.json), the bench lines, and
profiles/<tag>_recorded.json, which bench.py echoes as *_recorded fields."""
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = ROOT / "gpurun_out" / tag
dst = ROOT / "profiles"
cal = json.load(open(dst / "r01_pmc_traffic.json"))["calibration"]
recorded = {}


def newest(pattern):
    """gpurun_out/ accumulates the files of every collection run (one set per profiled process id): take the latest."""
    import os
    return sorted(glob.glob(pattern), key=os.path.getmtime, reverse=True)


def kernel_stats(name, out):
    f = newest(str(src / name / "*" / "*kernel_stats.csv"))
    if not f:
        return None
    shutil.copy(f[0], dst / out)
    for row in csv.DictReader(open(f[0])):
        if "k_tick_sweep" in row["Name"]:
            return float(row["AverageNs"]) / 1e3


def line(name):
    p = src / f"{name}.json"
    if not p.exists():
        return None
    txt = [l for l in p.read_text().splitlines() if l.startswith("{")]
    return json.loads(txt[-1]) if txt else None


for wl, stats in (("C3", "stats_c3"), ("C3_plain_loop", "stats_c3_plain"), ("C3x4", "stats_c3x4"), ("C5", "stats_c5")):
    us = kernel_stats(stats, f"{tag}_{wl.lower()}_kernel_stats.csv")
    if us:
        recorded[f"sweep_us_{wl}"] = us
    rec = line(stats)
    if rec:
        (dst / f"{tag}_{wl.lower()}_bench_under_rocprofv3.json").write_text(json.dumps(rec, indent=1) + "\n")

traffic = {}
for wl in ("C3", "C3x4"):
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = newest(str(src / f"traffic_{wl}_{c}" / "*" / "*counter_collection.csv"))
        if not f:
            continue
        rows = [r for r in csv.DictReader(open(f[0])) if "k_tick_sweep" in r["Kernel_Name"] and r["Counter_Name"] == c]
        rows = rows[25:]                                   # past the warm-up ticks
        vals[c] = (sum(float(r["Counter_Value"]) for r in rows) / max(len(rows), 1), len(rows))
        with open(f[0]) as s, open(dst / f"{tag}_pmc_{c.lower()}_{wl.lower()}_sweep.csv", "w") as o:
            for k, ln in enumerate(s):
                if k == 0 or "k_tick_sweep" in ln:
                    o.write(ln)
    if len(vals) == 2:
        rd = vals["FETCH_SIZE"][0] * 1024 * cal["read_factor"]
        wr = vals["WRITE_SIZE"][0] * 1024 * cal["write_factor"]
        stats_line = line(f"stats_{wl.lower()}") or {}
        alg = (stats_line.get("roofline") or {}).get("algorithmic_bytes_per_launch")
        traffic[wl] = dict(launches_averaged=vals["FETCH_SIZE"][1], fetch_size_kb=vals["FETCH_SIZE"][0],
                           write_size_kb=vals["WRITE_SIZE"][0], read_bytes=rd, write_bytes=wr, traffic_bytes=rd + wr,
                           algorithmic_bytes=alg, traffic_over_algorithmic=(rd + wr) / alg if alg else None)
        recorded[f"traffic_bytes_{wl}"] = rd + wr
# the same scene without missiles (tools/prof_run.py, PROF_M=0): what the missile phase's gathers add
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = newest(str(src / f"traffic_nomis_{c}" / "*" / "*counter_collection.csv"))
    if f:
        rows = [r for r in csv.DictReader(open(f[0])) if "k_tick_sweep" in r["Kernel_Name"] and r["Counter_Name"] == c][25:]
        vals[c] = sum(float(r["Counter_Value"]) for r in rows) / max(len(rows), 1)
if len(vals) == 2:
    rd, wr = vals["FETCH_SIZE"] * 1024 * cal["read_factor"], vals["WRITE_SIZE"] * 1024 * cal["write_factor"]
    alg = 85.0 * 1_000_000
    traffic["C3_without_missiles"] = dict(read_bytes=rd, write_bytes=wr, traffic_bytes=rd + wr, algorithmic_bytes=alg,
                                          traffic_over_algorithmic=(rd + wr) / alg,
                                          note="tools/prof_run.py with PROF_M=0: 1e6 rows, 16 radars, no missile rows and no missile phase")
(dst / f"{tag}_pmc_traffic.json").write_text(json.dumps(dict(
    kernel="k_tick_sweep<true, true, true>", calibration=cal, workloads=traffic,
    method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py (MI355X_MICROARCH.md HBM section): "
           "FETCH_SIZE x 1024 x 1.998 (gfx950 reports half the bytes of a coalesced stream; calibrated on a pure-streaming "
           "launch with this kernel's 8-byte-per-lane loads), WRITE_SIZE x 1024 x 1.000"), indent=1) + "\n")

for name in ("bench_c3", "bench_c3_20steps", "bench_c3_plain_loop", "bench_c3_exchange_one_rank", "bench_c2", "bench_c5", "bench_c3x4", "bench_c4_1gpu"):
    rec = line(name)
    if rec:
        (dst / f"{tag}_{name}.json").write_text(json.dumps(rec, indent=1) + "\n")
for name, out in (("sq_summary.txt", f"{tag}_pmc_sq.txt"), ("phases.log", f"{tag}_sweep_wave_timeline.txt"),
                  ("host_overhead.log", f"{tag}_host_overhead.txt")):
    if (src / name).exists():
        shutil.copy(src / name, dst / out)
x = ROOT / "gpurun_out" / "r02_exchange_host_time.json"
if x.exists():
    shutil.copy(x, dst / f"{tag}_exchange_host_time.json")
(dst / f"{tag}_recorded.json").write_text(json.dumps(recorded, indent=1) + "\n")
print(json.dumps(recorded, indent=1))
print(json.dumps({k: {kk: v[kk] for kk in ("traffic_bytes", "algorithmic_bytes", "traffic_over_algorithmic")} for k, v in traffic.items()}, indent=1))
