"""Turn the passes of tools/pmc_traffic.sh into profiles/<round>_pmc_traffic.json (per-launch HBM bytes of the
sweep kernel, corrected as MI355X_MICROARCH.md's HBM section prescribes; calibration factors from the
pure-streaming launch recorded in profiles/r01_pmc_traffic.json)."""
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
out = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "profiles" / "r01_pmc_traffic.json"
old = json.load(open(ROOT / "profiles" / "r01_pmc_traffic.json"))
cal = old["calibration"]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(str(ROOT / "gpurun_out" / f"traffic_{c}" / "*" / "*counter_collection.csv"))[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_tick_sweep" in r["Kernel_Name"] and r["Counter_Name"] == c]
    rows = rows[15:]                                   # past the warm-up ticks
    vals[c] = (sum(float(r["Counter_Value"]) for r in rows) / len(rows), len(rows))
    dst = ROOT / "profiles" / "r01_pmc" / f"{c.lower()}_c3_sweep.csv"
    with open(f) as src, open(dst, "w") as o:
        for k, line in enumerate(src):
            if k == 0 or "k_tick_sweep" in line:
                o.write(line)
fetch_kb, n = vals["FETCH_SIZE"]
write_kb, _ = vals["WRITE_SIZE"]
read_b = fetch_kb * 1024 * cal["read_factor"]
write_b = write_kb * 1024 * cal["write_factor"]
rec = dict(old)
rec.update(launches_averaged=n, fetch_size_kb=fetch_kb, write_size_kb=write_kb, read_bytes=read_b, write_bytes=write_b,
           traffic_bytes=read_b + write_b)
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps({k: rec[k] for k in ("fetch_size_kb", "write_size_kb", "read_bytes", "write_bytes", "traffic_bytes")}))
