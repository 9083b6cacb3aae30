#!/bin/bash
set -o pipefail
out=gpurun_out/r3l
mkdir -p $out
ZRK_SIDE_THREE=1 timeout -k 10 600 python -m pytest tests/test_gpu_overlap.py -m gpu -x -q > $out/pytest_three.log 2>&1; echo "pytest three rc=$?" | tee -a $out/pytest.rc
tail -2 $out/pytest_three.log
run() { name=$1; wl=$2; steps=$3; wu=$4; shift 4
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup $wu --no-cpu-baseline --no-c4 > $out/${name}.json 2> $out/${name}.err; }
run c3_1000_base C3 1000 50 ZRK_X=0
run c3_1000_three C3 1000 50 ZRK_SIDE_THREE=1
for i in 1 2 3; do run c3_20_three_$i C3 20 5 ZRK_SIDE_THREE=1; done
run c3x4_three C3x4 200 30 ZRK_SIDE_THREE=1
run c2_three C2 1000 50 ZRK_SIDE_THREE=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3l/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f"{f:45s} {d['ms_per_step']*1e3:7.2f} us/tick  launch {r['avg_kernel_us']:6.2f} us x{r.get('ticks_per_launch')}  frac {r['frac']:.3f} value {d['value']:.3e}  call {d['setup']['call_returned_after_us']:.0f} sync {d['setup']['sync_us']:.0f}")
    except Exception as e: print(f, "unreadable", e)
PY
