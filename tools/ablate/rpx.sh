cd $GRAFT_REPO_ROOT
SED_HIP_LIB=$GRAFT_REPO_ROOT/tools/ablate/variants/g_trace.so python tools/epi_trace.py 2>&1 | grep -v amdgpu.ids | tail -9
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -5
python tools/gemm_shapes.py 2>&1 | grep -v amdgpu.ids | head -10 | cut -c1-150
