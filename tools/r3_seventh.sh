#!/bin/bash
set -o pipefail
out=gpurun_out/r3g
mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee $out/pytest.rc
tail -4 $out/pytest.log
[ "$(cat $out/pytest.rc)" = "pytest rc=0" ] || exit 1
run() { name=$1; wl=$2; steps=$3; wu=$4; shift 4
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup $wu --no-cpu-baseline > $out/${name}.json 2> $out/${name}.err; }
for i in 1 2 3; do run c3_20_$i C3 20 5 ZRK_X=0; done
run c3_1000 C3 1000 50 ZRK_X=0
run c3_exch_300 C3 300 50 ZRK_BENCH_FORCE_EXCHANGE=1
run c3_exch_20 C3 20 5 ZRK_BENCH_FORCE_EXCHANGE=1
run c3_exch_1helper C3 300 50 ZRK_BENCH_FORCE_EXCHANGE=1 ZRK_HELPERS=1
run c2_1000 C2 1000 50 ZRK_X=0
run c2_20 C2 20 5 ZRK_X=0
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3g/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f"{f:45s} {d['ms_per_step']*1e3:7.2f} us/tick  launch {r['avg_kernel_us']:6.2f} us x{r.get('ticks_per_launch')}  frac {r['frac']:.3f} value {d['value']:.3e}  call {d['setup']['call_returned_after_us']:.0f} sync {d['setup']['sync_us']:.0f}")
    except Exception as e: print(f, "unreadable", e)
PY
