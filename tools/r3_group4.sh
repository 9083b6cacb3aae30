#!/bin/bash
# C3x4: shapes of the compaction beside the sweep
set -e
out=gpurun_out/r3r; mkdir -p $out
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload C3x4 --steps 400 --warmup 20 --no-c4 > $out/b_$name.json 2> $out/b_$name.err
  python - $out/b_$name.json $name <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("C3x4", sys.argv[2], "us/tick", round(r["ms_per_step"]*1e3,2), "frac", round(r["roofline"]["frac"],3))
PY
}
run base ZRK_NOP=1
run pair1024_t512 ZRK_PAIR_COMPACT_BLOCKS=1024 ZRK_PAIR_THREADS=512
run pair1024_t256 ZRK_PAIR_COMPACT_BLOCKS=1024 ZRK_PAIR_THREADS=256
run items4 ZRK_COMPACT_ITEMS=4
run items6 ZRK_COMPACT_ITEMS=6
run nopair ZRK_PAIR=0
run base2 ZRK_NOP=1
