#!/bin/bash
set -o pipefail
out=gpurun_out/r3f
mkdir -p $out
run() { name=$1; wl=$2; steps=$3; shift 3
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup 20 --no-cpu-baseline > $out/${name}.json 2> $out/${name}.err; }
run c2_default C2 1000 ZRK_X=0
run c2_overlap C2 1000 ZRK_OVERLAP_MIN_ROWS=0
run c2_overlap_nopair C2 1000 ZRK_OVERLAP_MIN_ROWS=0 ZRK_PAIR=0
run c2_overlap_20 C2 20 ZRK_OVERLAP_MIN_ROWS=0
run c2_default_20 C2 20 ZRK_X=0
run c3x4_pairpc C3x4 300 ZRK_X=0
run c3x4_pair C3x4 300 ZRK_PAIR_COMPACT=0
run c3x4_nopair C3x4 300 ZRK_PAIR=0
run c5 C5 300 ZRK_X=0
run c4 C4 100 ZRK_X=0
run c3_exch C3 300 ZRK_BENCH_FORCE_EXCHANGE=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3f/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f"{f:45s} {d['ms_per_step']*1e3:7.2f} us/tick  launch {r['avg_kernel_us']:6.2f} us x{r.get('ticks_per_launch')}  frac {r['frac']:.3f} value {d['value']:.3e}  {d['config']['loop'][:40]}")
    except Exception as e: print(f, "unreadable", e)
PY
