# kernel stats of the finetune2 step (everything on one stream) into gpurun_out/r3g/prof_$1 ; extra env via the caller
O=$GRAFT_REPO_ROOT/gpurun_out/r3g; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
SED_OVERLAP_TEACHER=0 SED_DW_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1 -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline ${@:2} > /dev/null 2>&1
rm -f $O/prof_$1/p_kernel_trace.csv
