#!/bin/bash
# the side stream's done word (no event behind every compaction): the suite around the overlapped loop, then the benches
set -e
out=gpurun_out/r3u; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_c4.py tests/test_gpu_exchange_c.py tests/test_gpu_multirank.py tests/test_gpu_engine.py tests/test_gpu_ensemble.py -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
run() { # name workload steps warmup
  timeout -k 10 200 python bench.py --workload $2 --steps $3 --warmup $4 --no-c4 > $out/b_$1.json 2> $out/b_$1.err
  python - $out/b_$1.json $1 <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "us/tick", round(r["ms_per_step"]*1e3,2), "frac", round(r["roofline"]["frac"],3))
PY
}
run c3_1000 C3 1000 20
run c3_driver C3 20 5
run c3_driver2 C3 20 5
run c3_driver3 C3 20 5
run c2 C2 1000 20
run c3x4 C3x4 400 20
run c5 C5 400 20
