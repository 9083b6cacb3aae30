#!/bin/bash
set -o pipefail
out=gpurun_out/r3k
mkdir -p $out
for t in 256 512; do
  ZRK_PAIR_THREADS=$t timeout -k 10 600 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_c4.py -m gpu -x -q > $out/pytest_$t.log 2>&1; echo "pytest threads=$t rc=$?" | tee -a $out/pytest.rc
  tail -2 $out/pytest_$t.log
done
grep -q "rc=1" $out/pytest.rc && exit 1
run() { name=$1; wl=$2; steps=$3; wu=$4; shift 4
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup $wu --no-cpu-baseline --no-c4 > $out/${name}.json 2> $out/${name}.err; }
for t in 1024 512 256; do
  run c3_1000_t$t C3 1000 50 ZRK_PAIR_THREADS=$t
  for i in 1 2 3; do run c3_20_t${t}_$i C3 20 5 ZRK_PAIR_THREADS=$t; done
  run c3x4_t$t C3x4 200 30 ZRK_PAIR_THREADS=$t
  run c2_t$t C2 1000 50 ZRK_PAIR_THREADS=$t
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3k/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f"{f:45s} {d['ms_per_step']*1e3:7.2f} us/tick  launch {r['avg_kernel_us']:6.2f} us x{r.get('ticks_per_launch')}  frac {r['frac']:.3f} value {d['value']:.3e}  call {d['setup']['call_returned_after_us']:.0f} sync {d['setup']['sync_us']:.0f}")
    except Exception as e: print(f, "unreadable", e)
PY
