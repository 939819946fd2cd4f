#!/bin/bash
# producer_pmc.sh : HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes) and L2 <-> fabric request counters of the LayerNorm-fold producer
# launches of tools/producer_bench.py -> gpurun_out/producer_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ppmc; mkdir -p $O
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "TCC_EA_WRREQ_STALL_sum TCC_EA_RDREQ_LEVEL_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $O/$n -o p -- python tools/producer_bench.py > $O/$n.log 2>&1
done
python - <<'PY' | tee gpurun_out/producer_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/ppmc/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "gemm_nt_pp" not in k: continue
        a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        n, v = acc[k][c]
        print(f"   {c:28s} launches {n:4d}  per launch {v / n:16.1f}")
PY
rm -rf $O
