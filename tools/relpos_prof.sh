python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "relpos" 2>&1 | grep -E "passed|failed|Error|assert" | head -8
mkdir -p gpurun_out/r3g; O=$GRAFT_REPO_ROOT/gpurun_out/r3g; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rp -o p -- python $GRAFT_REPO_ROOT/tools/relpos_bench.py > /dev/null 2>&1; rm -f $O/prof_rp/p_kernel_trace.csv
python - <<'PY'
import csv, os
for r in csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"] + '/gpurun_out/r3g/prof_rp/p_kernel_stats.csv')):
    if 'relpos' in r['Name'] or 'prep' in r['Name']:
        print(r['Name'][:50], r['Calls'], round(float(r['AverageNs'])/1e3, 1), r['MinNs'], r['MaxNs'])
PY
