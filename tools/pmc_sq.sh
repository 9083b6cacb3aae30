#!/bin/bash
# SQ / memory-pipe counter passes on the sweep kernel of tools/prof_run.py: pmc_sq.sh <outdir> [ENV=..]...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$1; shift
mkdir -p $out
for kv in "$@"; do export $kv; done
export PROF_TICKS=${PROF_TICKS:-60}
pass() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $out/$1 -- python3 $R/tools/prof_run.py > $out/$1.log 2>&1 || echo "pass $1 failed"; }
pass a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES"
pass b "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM"
pass c "GRBM_GUI_ACTIVE GRBM_COUNT"
pass d "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS"
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in sorted(glob.glob(out + "/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(p)):
        if "k_tick_sweep" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(f"{p.split('/')[-3]} {k:28s} mean per dispatch {sum(v)/len(v):14.1f}  (n={len(v)})")
PY
