import csv, glob, sys
for d in sys.argv[1:]:
    f = sorted(glob.glob(d + "/*/*kernel_trace.csv"))
    rows = sorted(csv.DictReader(open(f[-1])), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_tick_sweep" in r["Kernel_Name"] and "std::conditional<true" in r["Kernel_Name"]]
    t0 = int(rows[idx[-10]]["Start_Timestamp"])
    print("==", d)
    for r in rows[idx[-10]:idx[-1] + 6]:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"{s:9.1f} {e:9.1f} {e - s:7.1f}  q{r['Queue_Id']}  {r['Kernel_Name'][:60]}")
