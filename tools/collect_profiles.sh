#!/bin/bash
# Everything profiles/r02_* is made from, in one GPU call:  gpurun -- bash tools/collect_profiles.sh r02
# (counters in their own passes, kernel trace only next to them; the program itself after `--`).
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
echo "== plain bench lines"
$B --steps 1000 --warmup 50 > $out/bench_c3.json 2> $out/bench_c3.err
$B --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_c3_20steps.json 2>> $out/bench_c3.err
ZRK_OVERLAP=0 $B --steps 1000 --warmup 50 --no-cpu-baseline > $out/bench_c3_plain_loop.json 2>> $out/bench_c3.err
ZRK_BENCH_FORCE_EXCHANGE=1 $B --steps 1000 --warmup 50 --no-cpu-baseline > $out/bench_c3_exchange_one_rank.json 2>> $out/bench_c3.err
$B --workload C2 --steps 2000 --warmup 100 --no-cpu-baseline > $out/bench_c2.json 2>> $out/bench_c3.err
$B --workload C5 --steps 500 --warmup 50 --cpu-budget 5 > $out/bench_c5.json 2>> $out/bench_c3.err
$B --workload C3x4 --steps 200 --warmup 30 --no-cpu-baseline > $out/bench_c3x4.json 2>> $out/bench_c3.err
$B --workload C4 --steps 100 --warmup 30 --no-cpu-baseline > $out/bench_c4_1gpu.json 2>> $out/bench_c3.err
echo "== kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c3 -- $B --steps 300 --warmup 50 --no-cpu-baseline > $out/stats_c3.json 2> $out/stats_c3.err
( export ZRK_OVERLAP=0; rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c3_plain -- $B --steps 300 --warmup 50 --no-cpu-baseline > $out/stats_c3_plain.json 2> $out/stats_c3_plain.err )
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c3x4 -- $B --workload C3x4 --steps 100 --warmup 30 --no-cpu-baseline > $out/stats_c3x4.json 2> $out/stats_c3x4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c5 -- $B --workload C5 --steps 200 --warmup 30 --no-cpu-baseline > $out/stats_c5.json 2> $out/stats_c5.err
echo "== traffic counters"
for w in C3 C3x4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/traffic_${w}_$c -- $B --workload $w --steps 40 --warmup 20 --no-cpu-baseline > $out/traffic_${w}_$c.log 2>&1
  done
done
# what the missile phase (riding in the sweep's grid: ten thousand rows of scattered gathers) adds to the sweep's traffic
for c in FETCH_SIZE WRITE_SIZE; do
  ( export PROF_M=0 PROF_TICKS=60; rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/traffic_nomis_$c -- python3 $R/tools/prof_run.py > $out/traffic_nomis_$c.log 2>&1 )
done
echo "== SQ counters"
bash $R/tools/pmc_sq.sh $tag/sq > $out/sq_summary.txt 2>&1
echo "== host time of the exchange paths, per-wave timeline"
cd $R
python3 tools/exchange_host_time.py --record > $out/exchange_host_time.log 2>&1
python3 tools/sweep_phases.py > $out/phases.log 2>&1
python3 tools/host_overhead.py > $out/host_overhead.log 2>&1
echo done
