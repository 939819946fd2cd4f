mkdir -p gpurun_out/r2; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
python tools/attn_bench.py 2>&1 | grep mhsa
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
