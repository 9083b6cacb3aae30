R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r4r; mkdir -p $out; cd $R
python -m pytest tests/test_gpu_compact.py tests/test_gpu_timed_path.py tests/test_gpu_overlap.py tests/test_gpu_ensemble.py tests/test_gpu_engine.py tests/test_gpu_c4.py tests/test_gpu_exchange_c.py tests/test_gpu_multirank.py -x -q -m gpu > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py"
for i in 1 2 3; do
for v in new prev; do
if [ $v = prev ]; then export ZRK_HOT_LIB=$R/zrk_modulation_amd/csrc/libzrk_hot_prev.so; else unset ZRK_HOT_LIB; fi
$B --steps 1000 --warmup 50 --no-c4 --no-cpu-baseline > $out/${v}_1000_$i.json 2> $out/${v}_1000_$i.err
$B --workload C3x4 --steps 100 --warmup 20 --no-cpu-baseline > $out/${v}_c3x4_$i.json 2> $out/${v}_c3x4_$i.err
$B --workload C4 --steps 60 --warmup 12 --no-cpu-baseline > $out/${v}_c4_$i.json 2> $out/${v}_c4_$i.err
$B --workload C2 --steps 2000 --warmup 100 --no-cpu-baseline > $out/${v}_c2_$i.json 2> $out/${v}_c2_$i.err
done; done
echo done
