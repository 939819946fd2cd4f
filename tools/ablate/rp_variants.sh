#!/bin/bash
# rp_variants.sh V1 V2 ... : kernel-level times of tools/relpos_bench.py under library variants tools/ablate/variants/V.so (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  echo "== $v"
  SED_HIP_LIB=$R/tools/ablate/variants/$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o out -- python $R/tools/relpos_bench.py 2>&1 | grep "^relpos"
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys
for r in csv.reader(open(sys.argv[1])):
    if 'relpos' in r[0] or 'prep' in r[0]: print("   %-45s calls %3s avg %8.1f us" % (r[0][:45], r[1], float(r[3])/1e3))
PY
done
