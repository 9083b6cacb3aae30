#!/bin/bash
# round 3, second GPU call: the suite again, then the 20-step run under variations (what do the timed launches, the tail and
# a CU-masked side stream cost / buy)
set -o pipefail
out=gpurun_out/r3b
mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee $out/pytest.rc
tail -5 $out/pytest.log
run() { # name, env...
  name=$1; shift
  for i in 1 2 3; do env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/b20_${name}_$i.json 2> $out/b20_${name}_$i.err; done
}
run base
run stride10 ZRK_BENCH_STRIDE=10
run stride20 ZRK_BENCH_STRIDE=20
run tailflag ZRK_TAIL_EVENT=0 ZRK_BENCH_STRIDE=10
run cus64 ZRK_SIDE_CUS=64 ZRK_BENCH_STRIDE=10
run cus96 ZRK_SIDE_CUS=96 ZRK_BENCH_STRIDE=10
run cus32 ZRK_SIDE_CUS=32 ZRK_BENCH_STRIDE=10
run cus128 ZRK_SIDE_CUS=128 ZRK_BENCH_STRIDE=10
ZRK_TRACE=1 ZRK_BENCH_STRIDE=10 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/b20_trace.json 2> $out/b20_trace.err
for c in 0 64 96; do ZRK_SIDE_CUS=$c python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > $out/b1000_cus$c.json 2> $out/b1000_cus$c.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3b/b*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{f:45s} {d['ms_per_step']*1e3:7.2f} us/tick  sweep {d['roofline']['avg_kernel_us']:6.2f} us  frac {d['roofline']['frac']:.3f}  call {d['setup']['call_returned_after_us']:.0f} sync {d['setup']['sync_us']:.0f}")
    except Exception as e: print(f, "unreadable", e)
PY
