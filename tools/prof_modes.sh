mkdir -p gpurun_out/r3h; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h
SED_OVERLAP_TEACHER=0 SED_DW_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer > $O/prof_bench_line.json 2>/dev/null
rm -f $O/prof/p_kernel_trace.csv
SED_DW_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pmam -o p -- python bench.py --mode pmam --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer > /dev/null 2>&1
rm -f $O/prof_pmam/p_kernel_trace.csv
SED_DW_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pretrain -o p -- python bench.py --mode pretrain --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer > /dev/null 2>&1
rm -f $O/prof_pretrain/p_kernel_trace.csv
