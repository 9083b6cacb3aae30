#!/bin/bash
# the defaults after the pair compaction reaches 1024 workgroups: tests around it, C3x4 / C3 / 2e6 rows
set -e
out=gpurun_out/r3s; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_c4.py tests/test_gpu_compact.py -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
run() { # name workload env...
  name=$1; w=$2; shift; shift
  env "$@" timeout -k 10 200 python bench.py --workload $w --steps 400 --warmup 20 --no-c4 > $out/b_$name.json 2> $out/b_$name.err
  python - $out/b_$name.json $name <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "us/tick", round(r["ms_per_step"]*1e3,2), "frac", round(r["roofline"]["frac"],3))
PY
}
run c3x4 C3x4 ZRK_NOP=1
run c3 C3 ZRK_NOP=1
run c3x4_again C3x4 ZRK_NOP=1
run c3x4_old C3x4 ZRK_PAIR_COMPACT_BLOCKS=512
