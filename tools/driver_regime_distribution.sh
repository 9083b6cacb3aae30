#!/bin/bash
# The driver's command many times on one box:  gpurun -- bash tools/driver_regime_distribution.sh [runs]
# -> gpurun_out/${ZRK_DIST_TAG:-r05}_dist/summary.txt (copied to profiles/rNN_driver_regime_distribution.txt): per run the tick, the time the one
# call took to return, the synchronisation behind it, the span of its ten sweeps and the gaps between them (by the launches' own
# stamps), and any host-side wait of the library longer than a millisecond (ZRK_STALL_US).
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${ZRK_DIST_TAG:-r05}_dist; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
runs=${1:-40}
export ZRK_STALL_US=1000
for i in $(seq 1 $runs); do
  python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-c4 --no-cpu-baseline > $out/r_$i.json 2> $out/r_$i.err
done
python3 - "$out" "$runs" > $out/summary.txt <<'PY'
import json, sys, statistics
out, runs = sys.argv[1], int(sys.argv[2])
rows = []
for i in range(1, runs + 1):
    d = [json.loads(l) for l in open(f"{out}/r_{i}.json") if l.startswith("{")][-1]
    s = d["setup"]
    stalls = [l.strip() for l in open(f"{out}/r_{i}.err") if "zrk stall" in l]
    rows.append((d["ms_per_step"] * 1e3, s["call_returned_after_us"], s["sync_us"], s["sweeps_span_us"], s["sweep_gaps_us"], d["roofline"]["avg_kernel_us"], stalls))
t = sorted(r[0] for r in rows)
print(f"# python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-c4 --no-cpu-baseline, {runs} runs on one box (ZRK_STALL_US=1000)")
print(f"# us per tick: min {t[0]:.2f}  p25 {t[len(t) // 4]:.2f}  median {statistics.median(t):.2f}  p75 {t[(3 * len(t)) // 4]:.2f}  max {t[-1]:.2f}")
print("# run  us/tick  call returned [us]  sync [us]  sweeps' span [us]  gaps [us]  sweep launch [us]  stalls")
for i, r in enumerate(rows, 1):
    print(f"{i:4d} {r[0]:8.2f} {r[1]:12.0f} {r[2]:12.0f} {r[3]:14.1f} {r[4]:10.1f} {r[5]:12.2f}   {'; '.join(r[6])}")
PY
tail -n 3 $out/summary.txt
