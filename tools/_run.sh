R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r4x; mkdir -p $out; cd $R
ZRK_SERIAL_ROWS=1 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_timed_path.py tests/test_gpu_c4.py -x -q -m gpu > $out/pytest_serial.log 2>&1; echo "rc=$?" >> $out/pytest_serial.log
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py"
for i in 1 2; do
for v in par ser; do
if [ $v = ser ]; then export ZRK_SERIAL_ROWS=1; else unset ZRK_SERIAL_ROWS; fi
$B --workload C4 --steps 60 --warmup 12 --no-cpu-baseline > $out/${v}_c4_$i.json 2> $out/${v}_c4_$i.err
$B --workload C3x4 --steps 100 --warmup 20 --no-cpu-baseline > $out/${v}_c3x4_$i.json 2> $out/${v}_c3x4_$i.err
$B --steps 200 --warmup 20 --no-c4 --no-cpu-baseline > $out/${v}_c3_$i.json 2> $out/${v}_c3_$i.err
done; done
echo done
