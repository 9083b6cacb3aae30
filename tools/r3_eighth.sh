#!/bin/bash
set -o pipefail
out=gpurun_out/r3h
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_c4.py tests/test_gpu_exchange_c.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee $out/pytest.rc
tail -3 $out/pytest.log
[ "$(cat $out/pytest.rc)" = "pytest rc=0" ] || exit 1
run() { name=$1; wl=$2; steps=$3; wu=$4; shift 4
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup $wu --no-cpu-baseline > $out/${name}.json 2> $out/${name}.err; }
run base_1000 C3 1000 50 ZRK_X=0
run hi_1000 C3 1000 50 ZRK_SIDE_PRIORITY=high
run lo_1000 C3 1000 50 ZRK_SIDE_PRIORITY=low
run items2_1000 C3 1000 50 ZRK_COMPACT_ITEMS=2
run items3_1000 C3 1000 50 ZRK_COMPACT_ITEMS=3
run nopc_1000 C3 1000 50 ZRK_PAIR_COMPACT=0
for i in 1 2 3; do run base_20_$i C3 20 5 ZRK_X=0; run hi_20_$i C3 20 5 ZRK_SIDE_PRIORITY=high; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3h/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f"{f:45s} {d['ms_per_step']*1e3:7.2f} us/tick  launch {r['avg_kernel_us']:6.2f} us x{r.get('ticks_per_launch')}  frac {r['frac']:.3f} value {d['value']:.3e}  call {d['setup']['call_returned_after_us']:.0f} sync {d['setup']['sync_us']:.0f}")
    except Exception as e: print(f, "unreadable", e)
PY
