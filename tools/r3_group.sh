#!/bin/bash
# two-level sums in the compaction: the tests around it, then C3 / C3x4 with and without (ZRK_COMPACT_GROUP=0)
set -e
out=gpurun_out/r3o; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_association.py tests/test_gpu_ccp_step.py tests/test_gpu_compact.py tests/test_gpu_overlap.py tests/test_gpu_c4.py -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
for g in auto 0; do
  for w in C3 C3x4; do
    if [ $g = auto ]; then unset ZRK_COMPACT_GROUP; else export ZRK_COMPACT_GROUP=$g; fi
    timeout -k 10 200 python bench.py --workload $w --steps 400 --warmup 20 --no-c4 > $out/bench_${w}_g$g.json 2> $out/bench_${w}_g$g.err
    python - $out/bench_${w}_g$g.json $w $g <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "group", sys.argv[3], "us/tick", round(r["ms_per_step"]*1e3,2), "frac", r["roofline"]["frac"])
PY
  done
done
unset ZRK_COMPACT_GROUP
timeout -k 10 100 python bench.py --steps 20 --warmup 5 --no-c4 > $out/bench_driver.json 2>$out/bench_driver.err
python -c "
import json;r=json.loads(open('$out/bench_driver.json').read().strip().splitlines()[-1]);print('driver20 us/tick',round(r['ms_per_step']*1e3,2),r['roofline']['frac'])"
