#!/bin/bash
# Counter passes on the sweep kernel (run on the GPU box: gpurun -- bash tools/pmc_passes.sh ["philox 16" ...]).
# Counters in their own runs, kernel trace only -- no other trace domain next to --pmc.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cfgs=("$@"); [ ${#cfgs[@]} -eq 0 ] && cfgs=("philox 16")
for cfg in "${cfgs[@]}"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc2_$tag -- python3 $R/tools/pmc_run.py $cfg > $R/gpurun_out/pmc2_$tag.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc3_$tag -- python3 $R/tools/pmc_run.py $cfg > $R/gpurun_out/pmc3_$tag.log 2>&1 || exit 1
done
echo done
