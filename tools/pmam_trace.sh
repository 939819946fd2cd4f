# PMAM step: tests, bench line, rocprofv3 kernel stats and the launch sequence of the last step (gpurun_out/<tag>/last_step_order.txt)
TAG=${1:-pm}; O=gpurun_out/$TAG; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_pmam.py tests/test_gpu_pmam_kernels.py tests/test_gpu_kernels.py -x -q -m gpu -k "${2:-pmam or dw_tn or weight_images or gather or bn_finalize or lora or cnn}" > $O/tests.txt 2>&1; tail -15 $O/tests.txt
timeout 600 python bench.py --mode pmam --no-cpu-baseline 2>$O/bench_err.txt | tail -1 > $O/bench_pmam.json; cut -c1-260 $O/bench_pmam.json
SED_OVERLAP_TEACHER=0 SED_DW_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python bench.py --mode pmam --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timer > /dev/null 2>&1
python - $O <<'PY'
import csv, sys
O = sys.argv[1]
rows = list(csv.DictReader(open(O + '/prof/p_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
first = [i for i, r in enumerate(rows) if 'weight_images_kernel' in r['Kernel_Name']]
last = rows[first[-1]:]
with open(O + '/last_step_order.txt', 'w') as f:
    for r in last:
        f.write('%8.1f %s\n' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:150]))
print(len(rows), len(last), sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last) / 1e6, 'ms')
PY
rm -f $O/prof/p_kernel_trace.csv
