#!/bin/bash
# pair compaction beyond 512 workgroups (C3x4) against two single compactions
set -e
out=gpurun_out/r3p; mkdir -p $out
for b in 512 1024; do
  export ZRK_PAIR_COMPACT_BLOCKS=$b
  timeout -k 10 200 python bench.py --workload C3x4 --steps 400 --warmup 20 --no-c4 > $out/bench_C3x4_b$b.json 2> $out/bench_C3x4_b$b.err
  python - $out/bench_C3x4_b$b.json $b <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("C3x4 pair-compaction blocks", sys.argv[2], "us/tick", round(r["ms_per_step"]*1e3,2), "frac", r["roofline"]["frac"])
PY
done
export ZRK_PAIR_COMPACT_BLOCKS=1024
timeout -k 10 300 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_c4.py -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
