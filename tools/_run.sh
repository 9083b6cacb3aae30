R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r4F; mkdir -p $out; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_exchange_c.py tests/test_gpu_overlap.py tests/test_gpu_c4.py tests/test_gpu_multirank.py tests/test_gpu_bench.py -x -q -m gpu > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py"
for i in 1 2 3; do
for v in grp nogrp; do
if [ $v = nogrp ]; then export ZRK_EXCHANGE_GROUP=0; else unset ZRK_EXCHANGE_GROUP; fi
ZRK_BENCH_FORCE_EXCHANGE=1 timeout -k 10 200 $B --steps 1000 --warmup 50 --no-cpu-baseline --no-c4 > $out/${v}_$i.json 2> $out/${v}_$i.err
done; done
echo done
