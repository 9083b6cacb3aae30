#!/usr/bin/env python3
"""gpurun_out/r05/ (tools/collect_profiles_r05.sh) -> profiles/r05_*: kernel-stats tables (without the clock spin-up's
self-test kernel and the stamps' reduction kernel, both outside the timed region; percentages recomputed), the GPU timeline of the driver's 20-step run, the per-launch HBM traffic of the
sweep (FETCH_SIZE x 1024 x read factor, WRITE_SIZE x 1024; factors calibrated in round 1 on a pure-streaming launch of
this kernel's access widths, profiles/r01_pmc_traffic.json), the SQ counters, the bench lines, and
profiles/r05_recorded.json, which bench.py echoes as *_recorded fields."""
import csv
import glob
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = "r05"
src = ROOT / "gpurun_out" / tag
dst = ROOT / "profiles"
cal = json.load(open(dst / "r01_pmc_traffic.json"))["calibration"]
recorded = {}


def newest(pattern):
    return sorted(glob.glob(pattern), key=os.path.getmtime, reverse=True)


def kernel_stats(name, out):
    """The stats table without k_selftest_noise (bench.py's 100 ms clock spin-up, outside the timed region), percentages over
    what is left.  Returns {kernel substring: average us}."""
    f = newest(str(src / name / "*" / "*kernel_stats.csv"))
    if not f:
        return {}
    # (bench.py's own kernels outside the timed region: the clock spin-up, the reduction of the sweeps' stamps behind it)
    rows = [r for r in csv.DictReader(open(f[0])) if "k_selftest_noise" not in r["Name"] and "k_reduce_stamps" not in r["Name"]]
    total = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
    with open(dst / out, "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], f"{100 * float(r['TotalDurationNs']) / total:.2f}",
                        r["MinNs"], r["MaxNs"], r["StdDev"]])
    return {r["Name"]: float(r["AverageNs"]) / 1e3 for r in rows}


def line(name):
    p = src / f"{name}.json"
    if not p.exists():
        return None
    txt = [l for l in p.read_text().splitlines() if l.startswith("{")]
    return json.loads(txt[-1]) if txt else None


def pick(stats, *needles):
    for k, v in stats.items():
        if all(n in k for n in needles):
            return v


st = kernel_stats("stats_c3", f"{tag}_c3_kernel_stats.csv")
recorded["sweep_us_C3"] = pick(st, "k_tick_sweep")
recorded["compact_us_C3"] = pick(st, "k_compact_pair")
st = kernel_stats("stats_c3_plain", f"{tag}_c3_plain_loop_kernel_stats.csv")
recorded["sweep_us_C3_plain_loop"] = pick(st, "k_tick_sweep")
st = kernel_stats("stats_c3_driver", f"{tag}_c3_driver_run_kernel_stats.csv")
recorded["sweep_us_C3_driver_run"] = pick(st, "k_tick_sweep")
st = kernel_stats("stats_c3x4", f"{tag}_c3x4_kernel_stats.csv")
recorded["sweep_us_C3x4"] = pick(st, "k_tick_sweep")
st = kernel_stats("stats_c2", f"{tag}_c2_kernel_stats.csv")
recorded["sweep_us_C2"] = pick(st, "k_tick_sweep")
st = kernel_stats("stats_c4", f"{tag}_c4_1gpu_kernel_stats.csv")
try:
    kernel_stats("stats_c2_battery", f"{tag}_c2_battery_kernel_stats.csv")
except Exception as exc:                            # (collected only where the battery workload ran)
    print("no battery stats:", exc)
if (src / "dispatch_rate_probe.txt").exists():
    (dst / f"{tag}_dispatch_rate_probe.txt").write_text((src / "dispatch_rate_probe.txt").read_text())
recorded["sweep_us_C4_1gpu"] = pick(st, "k_tick_sweep")
if (src / "ccp_scale.txt").exists():
    body = [l for l in (src / "ccp_scale.txt").read_text().splitlines() if l.startswith("tick") or l.startswith("both")]
    (dst / f"{tag}_ccp_scale.txt").write_text(
        "# python tools/ccp_scale.py (MI355X, one GPU): the command post's step on the device (zrk_ccp_step), 10^6 tracks, 10^5\n"
        "# detections a tick; candidate pass through the spatial index (default from 8192 tracks) against tiled all pairs\n"
        "# (ZRK_CCP_GRID=0).  Wall clock around the call + synchronisation.  Tick 0 is the dictionaries filling up: 10^6 new\n"
        "# targets, 40 000 missiles handed out one after the other by k_ccp_launch (sequential in the launchers' counts).\n"
        + "\n".join(body) + "\n")
if (src / "pair_sweep_wave_timeline.txt").exists() and (src / "pair_sweep_wave_timeline.txt").stat().st_size > 200:
    (dst / f"{tag}_pair_sweep_wave_timeline.txt").write_text((src / "pair_sweep_wave_timeline.txt").read_text())
for name in ("stats_c3", "stats_c3_plain", "stats_c3_driver"):
    rec = line(name)
    if rec:
        (dst / f"{tag}_{name[6:]}_bench_under_rocprofv3.json").write_text(json.dumps(rec, indent=1) + "\n")

# the GPU's view of the driver's 20-step run: the last 10 sweep launches and everything between them
f = newest(str(src / "stats_c3_driver" / "*" / "*kernel_trace.csv"))
if f:
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
    # (the timed call's launches sweep two ticks each; the one-tick launches behind it are bench.py's calibration of the dispatch overhead)
    idx = [i for i, r in enumerate(rows) if "k_tick_sweep" in r["Kernel_Name"] and "std::conditional<true" in r["Kernel_Name"]]
    if len(idx) >= 10:
        t0 = int(rows[idx[-10]]["Start_Timestamp"])
        with open(dst / f"{tag}_timeline_20steps.txt", "w") as o:
            o.write("# GPU timeline (rocprofv3 --kernel-trace) of the timed call of `bench.py --steps 20 --warmup 5`: start, end, duration [us], queue, kernel\n")
            for r in rows[idx[-10]:idx[-1] + 5]:
                s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
                o.write(f"{s:9.1f} {e:9.1f} {e - s:7.1f}  q{r['Queue_Id']}  {r['Kernel_Name'][:70]}\n")

traffic = {}
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = newest(str(src / f"traffic_C3_{c}" / "*" / "*counter_collection.csv"))
    if not f:
        continue
    rows = [r for r in csv.DictReader(open(f[0])) if "k_tick_sweep" in r["Kernel_Name"] and r["Counter_Name"] == c]
    rows = rows[12:]                                   # past the warm-up launches
    vals[c] = (sum(float(r["Counter_Value"]) for r in rows) / max(len(rows), 1), len(rows))
    with open(f[0]) as s, open(dst / f"{tag}_pmc_{c.lower()}_c3_sweep.csv", "w") as o:
        for k, ln in enumerate(s):
            if k == 0 or "k_tick_sweep" in ln:
                o.write(ln)
if len(vals) == 2:
    rd = vals["FETCH_SIZE"][0] * 1024 * cal["read_factor"]
    wr = vals["WRITE_SIZE"][0] * 1024 * cal["write_factor"]
    stats_line = line("stats_c3") or {}
    roof = stats_line.get("roofline") or {}
    alg = roof.get("algorithmic_bytes_per_launch")
    traffic["C3"] = dict(launches_averaged=vals["FETCH_SIZE"][1], ticks_per_launch=roof.get("ticks_per_launch"),
                         fetch_size_kb=vals["FETCH_SIZE"][0], write_size_kb=vals["WRITE_SIZE"][0], read_bytes=rd, write_bytes=wr,
                         traffic_bytes=rd + wr, algorithmic_bytes=alg, traffic_over_algorithmic=(rd + wr) / alg if alg else None)
    recorded["traffic_bytes_C3"] = rd + wr
# ... and of the pair compaction beside it, from the same two passes
cvals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = newest(str(src / f"traffic_C3_{c}" / "*" / "*counter_collection.csv"))
    if f:
        rows = [r for r in csv.DictReader(open(f[0])) if "k_compact_pair" in r["Kernel_Name"] and r["Counter_Name"] == c][12:]
        if rows:
            cvals[c] = (sum(float(r["Counter_Value"]) for r in rows) / len(rows), len(rows))
if len(cvals) == 2:
    rd = cvals["FETCH_SIZE"][0] * 1024 * cal["read_factor"]
    wr = cvals["WRITE_SIZE"][0] * 1024 * cal["write_factor"]
    traffic["C3_k_compact_pair"] = dict(launches_averaged=cvals["FETCH_SIZE"][1], ticks_per_launch=2, fetch_size_kb=cvals["FETCH_SIZE"][0],
                                        write_size_kb=cvals["WRITE_SIZE"][0], read_bytes=rd, write_bytes=wr, traffic_bytes=rd + wr,
                                        what="the masks of both ticks read (4 B per slot), the lists of both ticks written (4 B per entry), the "
                                             "detected slots' masks cleared")
(dst / f"{tag}_pmc_traffic.json").write_text(json.dumps(dict(
    kernel="k_tick_sweep<true, true, true, true, true> (two ticks per launch)", calibration=cal, workloads=traffic,
    method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py (MI355X_MICROARCH.md HBM section): "
           "FETCH_SIZE x 1024 x 1.998 (gfx950 reports half the bytes of a coalesced stream; calibrated on a pure-streaming "
           "launch with this kernel's 8-byte-per-lane loads), WRITE_SIZE x 1024 x 1.000.  Algorithmic bytes per launch (round 4's "
           "accounting): what a launch of two ticks must move -- 57 B of trajectory columns per live entity read ONCE, 28 B written "
           "per tick swept, 1 B per tombstone (113 B per live entity); `effective` in the bench line keeps SURVEY 8d's 85 B per "
           "entity and tick swept."), indent=1) + "\n")

# SQ counters of the pair sweep and of the pair compaction beside it: per-dispatch averages
sq = {"k_tick_sweep": {}, "k_compact_pair": {}}
for d in glob.glob(str(src / "sq_*")):
    f = newest(d + "/*/*counter_collection.csv")
    if not f:
        continue
    acc = {"k_tick_sweep": {}, "k_compact_pair": {}}
    for r in csv.DictReader(open(f[0])):
        for kern in acc:
            if kern in r["Kernel_Name"] and (kern != "k_tick_sweep" or "std::conditional<true" in r["Kernel_Name"]):
                acc[kern].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for kern in acc:
        for k, v in acc[kern].items():
            v = v[6:] or v
            sq[kern][k] = sum(v) / len(v)
if sq["k_tick_sweep"]:
    with open(dst / f"{tag}_pmc_sq.txt", "w") as o:
        for kern, title in (("k_tick_sweep", "k_tick_sweep (pair launch, C3)"), ("k_compact_pair", "k_compact_pair<1024> (the launch beside it)")):
            q = sq[kern]
            if not q:
                continue
            o.write(f"# {title}, rocprofv3 --pmc, per dispatch (kernels serialised by the counter passes)\n")
            for k in sorted(q):
                o.write(f"{k:24s} {q[k]:16.1f}\n")
            if q.get("SQ_WAVES"):
                for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS"):
                    if k in q:
                        o.write(f"{k + ' / wave':24s} {q[k] / q['SQ_WAVES']:16.1f}\n")
            if q.get("SQ_WAVE_CYCLES"):
                for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                    if k in q:
                        o.write(f"{k + ' / wave cycles':32s} {q[k] / q['SQ_WAVE_CYCLES']:8.3f}\n")

for name in ("bench_c3_driver_1", "bench_c3_driver_2", "bench_c3_driver_3", "bench_c3_driver_no_spinup", "bench_c3", "bench_c3_one_tick_per_launch", "bench_c3_plain_loop",
             "bench_c3_exchange_one_rank", "bench_c2", "bench_c2_one_tick_per_launch", "bench_c5", "bench_c3x4", "bench_c4_1gpu", "bench_c2_battery"):
    rec = line(name)
    if rec:
        (dst / f"{tag}_{name}.json").write_text(json.dumps(rec, indent=1) + "\n")
drv = line("stats_c3_driver")
if drv and recorded.get("sweep_us_C3_driver_run"):
    r = drv["roofline"]
    recorded["driver_run_line_vs_rocprofv3"] = dict(
        line_avg_kernel_us=r["avg_kernel_us"], line_first_wave_in_to_last_wave_out_us=r.get("first_wave_in_to_last_wave_out_us"),
        line_dispatch_overhead_us=r.get("dispatch_overhead_us"), line_samples=r["samples"],
        rocprofv3_avg_us=recorded["sweep_us_C3_driver_run"],
        note="the SAME process: bench.py --steps 20 --warmup 5 under rocprofv3 --kernel-trace; rocprofv3's average is over the 12 pair "
             "launches of warm-up and timed call, the line's over the 10 of the timed call (under the tracer the line's dispatch "
             "overhead, measured with event pairs, carries the tracer's own per-dispatch cost)")
(dst / f"{tag}_recorded.json").write_text(json.dumps(recorded, indent=1) + "\n")
print(json.dumps(recorded, indent=1))
print(json.dumps(traffic, indent=1))
for name in ("bench_c3_driver_1", "bench_c3_driver_2", "bench_c3_driver_3", "bench_c3_driver_no_spinup", "bench_c3", "bench_c3_one_tick_per_launch", "bench_c3_plain_loop",
             "bench_c3_exchange_one_rank", "bench_c2", "bench_c2_one_tick_per_launch", "bench_c5", "bench_c3x4", "bench_c4_1gpu", "bench_c2_battery"):
    rec = line(name)
    if rec:
        r = rec["roofline"]
        print(f"{name:32s} {rec['ms_per_step'] * 1e3:8.2f} us/tick  value {rec['value']:.3e}  launch {r['avg_kernel_us']:7.2f} us x{r.get('ticks_per_launch')} frac {r['frac']:.3f}"
              + (f"  c4_strong {rec['c4_strong']['ms_per_step'] * 1e3:.1f} us/tick" if rec.get("c4_strong") else ""))
