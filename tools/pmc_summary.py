"""Per-wave means of the counters tools/pmc_passes.sh collected (reads gpurun_out/pmc[23]_<tag>/)."""
import collections
import csv
import glob
import sys

for tag in sys.argv[1:] or ["philox_16"]:
    for pre in ("pmc2", "pmc3"):
        f = glob.glob(f"gpurun_out/{pre}_{tag}/*/*counter_collection.csv")
        if not f:
            print(pre, tag, "missing")
            continue
        acc, waves = collections.defaultdict(list), 15625
        for row in csv.DictReader(open(f[0])):
            if "k_tick_sweep" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        kt = glob.glob(f"gpurun_out/{pre}_{tag}/*/*kernel_trace.csv")[0]
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt)) if "k_tick_sweep" in r["Kernel_Name"]]
        if "SQ_WAVES" in acc:
            waves = sum(acc["SQ_WAVES"]) / len(acc["SQ_WAVES"])
        print(pre, tag, f"sweep {sum(d) / len(d) / 1e3:.2f} us;", {k: round(sum(v) / len(v) / waves, 1) for k, v in acc.items()})
