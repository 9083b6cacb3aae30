#!/bin/bash
set -o pipefail
out=gpurun_out/r3c
mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee $out/pytest.rc
tail -3 $out/pytest.log
run() { name=$1; shift
  for i in 1 2 3 4 5; do env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/b20_${name}_$i.json 2> $out/b20_${name}_$i.err; done
}
run base
run activewait ROC_ACTIVE_WAIT_TIMEOUT=1000
run nocache ZRK_TAIL_EVENT=0
ZRK_TRACE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/b20_trace.json 2> $out/b20_trace.err
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > $out/b1000.json 2> $out/b1000.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3c/b*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{f:45s} {d['ms_per_step']*1e3:7.2f} us/tick  sweep {d['roofline']['avg_kernel_us']:6.2f} us  frac {d['roofline']['frac']:.3f}  call {d['setup']['call_returned_after_us']:.0f} sync {d['setup']['sync_us']:.0f}")
    except Exception as e: print(f, "unreadable", e)
PY
grep "zrk trace" $out/b20_trace.err | tail -62 | head -12
