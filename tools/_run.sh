R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r4N; mkdir -p $out; cd $R
python3 tools/_sync_probe.py > $out/sync.txt 2>&1
ZRK_SIDE_QUERY=1 python3 tools/_sync_probe.py > $out/sync_q.txt 2>&1
tail -n 6 $out/sync.txt; echo; tail -n 6 $out/sync_q.txt
