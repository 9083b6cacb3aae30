#!/bin/bash
# rocprofv3 kernel-trace stats of tools/prof_run.py under a few env settings: prof_stats.sh <outdir> "ENV=.. ENV=.." ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$1; shift
mkdir -p $out
k=0
for cfg in "$@"; do
  k=$((k+1))
  ( export $cfg _X=1; rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$k -- python3 $R/tools/prof_run.py > $out/p$k.log 2>&1 )
  echo "== [$cfg]" >> $out/summary.txt
  f=$(find $out/p$k -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $out/summary.txt <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    print(f"{row['Name'][:90]:90s} calls {row['Calls']:>6s} avg {float(row['AverageNs'])/1e3:8.2f} us  min {float(row['MinNs'])/1e3:8.2f}  max {float(row['MaxNs'])/1e3:8.2f}")
PY
done
cat $out/summary.txt
