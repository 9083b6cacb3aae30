#!/bin/bash
# round 3, first GPU call: the suite, the driver's bench command twice, a long run, and a kernel trace of the 20-step run
set -o pipefail
out=gpurun_out/r3a
mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee $out/pytest.rc
tail -5 $out/pytest.log
python bench.py --steps 20 --warmup 5 > $out/bench20_a.json 2> $out/bench20_a.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench20_b.json 2> $out/bench20_b.err
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline > $out/bench1000.json 2> $out/bench1000.err
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench20_traced.json 2> $out/trace.err
find $out/trace -name "*kernel_trace.csv" | head
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3a/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"]*1e3, "us/tick", d["roofline"]["avg_kernel_us"], "us sweep", d["roofline"]["frac"])
    except Exception as e: print(f, "unreadable", e)
PY
