R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r4E; mkdir -p $out; cd $R
python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "rc=$?" >> $out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; echo "rc=$?" >> $out/smoke.log
