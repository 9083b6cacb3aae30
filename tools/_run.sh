R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r4C; mkdir -p $out; cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_overlap.py -x -q -m gpu -k "alternating" > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
