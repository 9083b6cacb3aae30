#!/bin/bash
set -o pipefail
out=gpurun_out/r3e
mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee $out/pytest.rc
tail -4 $out/pytest.log
[ "$(cat $out/pytest.rc)" = "pytest rc=0" ] || exit 1
run() { name=$1; shift
  for i in 1 2 3; do env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/b20_${name}_$i.json 2> $out/b20_${name}_$i.err; done
}
run pairpc
run pair ZRK_PAIR_COMPACT=0
run nopair ZRK_PAIR=0
ZRK_TRACE=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/b20_trace.json 2> $out/b20_trace.err
timeout -k 10 300 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > $out/b1000_pairpc.json 2> $out/b1000_pairpc.err
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $out/b200_prof.json 2> $out/prof.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3e/b*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f"{f:45s} {d['ms_per_step']*1e3:7.2f} us/tick  launch {r['avg_kernel_us']:6.2f} us x{r.get('ticks_per_launch')}  frac {r['frac']:.3f}  call {d['setup']['call_returned_after_us']:.0f} sync {d['setup']['sync_us']:.0f}")
    except Exception as e: print(f, "unreadable", e)
PY
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} head -5 {} | cut -c1-220
