#!/bin/bash
# Everything profiles/r05_* is made from, in one GPU call:  gpurun -- bash tools/collect_profiles_r05.sh
# (counters in their own passes, kernel trace only next to them; the program itself after `--`).
tag=r05
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
# (the wave timeline needs the diagnostics build of the SAME source: make -C zrk_modulation_amd/csrc probe, before the call)
if [ ! -f $R/zrk_modulation_amd/csrc/libzrk_hot_probe.so ] || [ $R/zrk_modulation_amd/csrc/libzrk_hot_probe.so -ot $R/zrk_modulation_amd/csrc/zrk_hot.hip ]; then
  echo "libzrk_hot_probe.so is missing or older than zrk_hot.hip: run  make -C zrk_modulation_amd/csrc probe  first" >&2; exit 1
fi
echo "== bench lines"
for i in 1 2 3; do $B --steps 20 --warmup 5 > $out/bench_c3_driver_$i.json 2> $out/bench_c3_driver_$i.err; done
ZRK_BENCH_SPINUP_MS=0 $B --steps 20 --warmup 5 --no-c4 --no-cpu-baseline > $out/bench_c3_driver_no_spinup.json 2>> $out/bench_c3.err
$B --steps 1000 --warmup 50 --no-c4 > $out/bench_c3.json 2> $out/bench_c3.err
ZRK_PAIR=0 $B --steps 1000 --warmup 50 --no-cpu-baseline --no-c4 > $out/bench_c3_one_tick_per_launch.json 2>> $out/bench_c3.err
ZRK_OVERLAP=0 $B --steps 1000 --warmup 50 --no-cpu-baseline --no-c4 > $out/bench_c3_plain_loop.json 2>> $out/bench_c3.err
ZRK_BENCH_FORCE_EXCHANGE=1 $B --steps 1000 --warmup 50 --no-cpu-baseline > $out/bench_c3_exchange_one_rank.json 2>> $out/bench_c3.err
$B --workload C2 --steps 2000 --warmup 100 --no-cpu-baseline > $out/bench_c2.json 2>> $out/bench_c3.err
ZRK_PAIR=0 $B --workload C2 --steps 2000 --warmup 100 --no-cpu-baseline > $out/bench_c2_one_tick_per_launch.json 2>> $out/bench_c3.err
$B --workload C5 --steps 500 --warmup 50 --cpu-budget 5 > $out/bench_c5.json 2>> $out/bench_c3.err
$B --workload C3x4 --steps 200 --warmup 30 --no-cpu-baseline > $out/bench_c3x4.json 2>> $out/bench_c3.err
$B --workload C4 --steps 100 --warmup 30 --no-cpu-baseline > $out/bench_c4_1gpu.json 2>> $out/bench_c3.err
$B --workload C2-battery --steps 400 --warmup 40 --cpu-budget 10 > $out/bench_c2_battery.json 2>> $out/bench_c3.err
echo "== kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c3 -- $B --steps 300 --warmup 50 --no-cpu-baseline --no-c4 > $out/stats_c3.json 2> $out/stats_c3.err
( export ZRK_OVERLAP=0; rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c3_plain -- $B --steps 300 --warmup 50 --no-cpu-baseline --no-c4 > $out/stats_c3_plain.json 2> $out/stats_c3_plain.err )
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c3_driver -- $B --steps 20 --warmup 5 --no-cpu-baseline --no-c4 > $out/stats_c3_driver.json 2> $out/stats_c3_driver.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c3x4 -- $B --workload C3x4 --steps 100 --warmup 30 --no-cpu-baseline > $out/stats_c3x4.json 2> $out/stats_c3x4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c2 -- $B --workload C2 --steps 400 --warmup 50 --no-cpu-baseline > $out/stats_c2.json 2> $out/stats_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c4 -- $B --workload C4 --steps 40 --warmup 10 --no-cpu-baseline > $out/stats_c4.json 2> $out/stats_c4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c2_battery -- $B --workload C2-battery --steps 100 --warmup 20 --no-cpu-baseline > $out/stats_c2_battery.json 2> $out/stats_c2_battery.err
echo "== the dispatcher: refill of a full device, and when a finished wave's slot is given out again"
( cd $R/tools && make -s dispatch_rate_probe ) && $R/tools/dispatch_rate_probe > $out/dispatch_rate_probe.txt 2>&1
echo "== wave timeline of a pair sweep inside the loop (probe build)"
python3 $R/tools/sweep_phases_pair.py > $out/pair_sweep_wave_timeline.txt 2> $out/pair_sweep_wave_timeline.err
echo "== the command post's step at scale"
python3 $R/tools/ccp_scale.py > $out/ccp_scale.txt 2> $out/ccp_scale.err
echo "== traffic counters (separate passes)"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/traffic_C3_$c -- $B --steps 40 --warmup 20 --no-cpu-baseline --no-c4 > $out/traffic_C3_$c.log 2>&1
done
echo "== SQ counters"
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/sq_$n -- $B --steps 24 --warmup 12 --no-cpu-baseline --no-c4 > $out/sq_$n.log 2>&1
done
echo done
