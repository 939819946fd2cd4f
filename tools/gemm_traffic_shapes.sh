#!/bin/bash
# per-shape fabric traffic of the finetune2 step's GEMMs -> gpurun_out/<tag>_gemm_traffic_shapes.{json,txt}   (tools/gemm_traffic_shapes.py)
TAG=${1:-r6}; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/gts; mkdir -p $O
export SED_OVERLAP_TEACHER=0 SED_DW_STREAM=0
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python tools/gemm_traffic_shapes.py run $O/calls.json > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python tools/gemm_traffic_shapes.py run $O/calls_w.json > $O/w.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/h -o h -- python tools/gemm_traffic_shapes.py run $O/calls_h.json > $O/h.log 2>&1
F=$(find $O/f -name "*counter_collection.csv" | head -1); W=$(find $O/w -name "*counter_collection.csv" | head -1); H=$(find $O/h -name "*counter_collection.csv" | head -1)
python tools/gemm_traffic_shapes.py parse $O/calls.json $F $W $H gpurun_out/${TAG}_gemm_traffic_shapes.json | tee gpurun_out/${TAG}_gemm_traffic_shapes.txt
tail -3 $O/f.log
rm -rf $O/f $O/w $O/h
