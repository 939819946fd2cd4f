cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -5
for v in 0 1 0 1; do echo "== persist $v"; SED_GEMM_PERSIST=$v python tools/epi_ab.py 2>&1 | grep TFLOP; done
for v in 0 1; do echo "== persist $v"; SED_GEMM_PERSIST=$v python tools/gemm_shapes.py 2>&1 | grep -v amdgpu.ids | head -9 | cut -c1-150; done
for v in 0 1 0 1; do SED_GEMM_PERSIST=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo " persist=$v"; done
