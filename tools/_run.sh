R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r4u; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py"
for i in 1 2; do
rocprofv3 --kernel-trace --output-format csv -d $out/new_$i -- $B --steps 20 --warmup 5 --no-cpu-baseline --no-c4 > $out/new_$i.json 2> $out/new_$i.err
ZRK_TAIL_COMPUTE=0 rocprofv3 --kernel-trace --output-format csv -d $out/old_$i -- $B --steps 20 --warmup 5 --no-cpu-baseline --no-c4 > $out/old_$i.json 2> $out/old_$i.err
done
python3 $R/tools/_tl.py $out/new_1 $out/old_1 $out/new_2 $out/old_2 > $out/tl.txt 2>&1
echo done
