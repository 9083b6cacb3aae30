#!/bin/bash
# HBM traffic of the sweep kernel at C3: FETCH_SIZE and WRITE_SIZE in separate counter passes (kernel trace only),
# on the bench itself.  Run on the GPU box: gpurun -- bash tools/pmc_traffic.sh ; then tools/pmc_traffic_summary.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/traffic_$c -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/traffic_$c.log 2>&1 || exit 1
done
echo done
