#!/bin/bash
# C3x4, two-level sums on / off, alternating on one box
set -e
out=gpurun_out/r3q; mkdir -p $out
for i in 1 2 3; do
  for g in auto 0; do
    if [ $g = auto ]; then unset ZRK_COMPACT_GROUP; else export ZRK_COMPACT_GROUP=$g; fi
    timeout -k 10 200 python bench.py --workload C3x4 --steps 400 --warmup 20 --no-c4 > $out/b_${g}_$i.json 2> $out/b_${g}_$i.err
    python - $out/b_${g}_$i.json $g <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("C3x4 group", sys.argv[2], "us/tick", round(r["ms_per_step"]*1e3,2), "frac", round(r["roofline"]["frac"],3))
PY
  done
done
