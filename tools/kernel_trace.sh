#!/bin/bash
# kernel trace of the steady loop and of the driver's 20-step call
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/trace; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/c3 -- python3 $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-c4 > $out/c3.json 2> $out/c3.err
rocprofv3 --kernel-trace --output-format csv -d $out/drv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c4 > $out/drv.json 2> $out/drv.err
find $out -name "*kernel_trace.csv" | head
