mkdir -p gpurun_out/r2; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
for m in finetune2 pretrain pmam; do
for v in 0 1 0 1; do echo "mode=$m dw_stream=$v"; SED_DW_STREAM=$v python bench.py --mode $m --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | cut -c1-200 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo; done; done
