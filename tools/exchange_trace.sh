#!/bin/bash
# kernel trace of the one-rank RCCL flow (ZRK_BENCH_FORCE_EXCHANGE=1): per-kernel durations and the gaps between kernels on
# the compute stream. usage: exchange_trace.sh <outdir under gpurun_out>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$1
mkdir -p $out
if [ "${FORCE:-1}" = "1" ]; then export ZRK_BENCH_FORCE_EXCHANGE=1; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $out/x -- python3 $R/bench.py --steps 200 --warmup 50 --no-cpu-baseline > $out/x.log 2>&1
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + "/x/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-2000:]
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
with open(out + "/timeline.txt", "w") as o:
    t0 = int(rows[-60]["Start_Timestamp"])
    for r in rows[-60:]:
        o.write(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.2f} {(int(r['End_Timestamp'])-t0)/1e3:9.2f} q{r['Queue_Id']} {r['Kernel_Name'][:70]}\n")
    for q, rs in byq.items():
        names = collections.Counter(r["Kernel_Name"][:50] for r in rs)
        o.write(f"queue {q}: {dict(names)}\n")
print(open(out + "/timeline.txt").read())
PY
