#!/bin/bash
# val_profile.sh TAG [ENV=VAL ...] : rocprofv3 kernel stats of the validation bench (3 steps) -> gpurun_out/val_prof_TAG.csv (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
env "$@" SED_OVERLAP_TEACHER=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp_$tag -o p -- python $R/bench.py --mode val --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timer > /dev/null 2>&1
f=$(find /tmp/vp_$tag -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out
cp $f $R/gpurun_out/val_prof_$tag.csv
python - "$f" <<'PY'
import csv,sys
rows=list(csv.reader(open(sys.argv[1])))[1:]
tot=sum(float(r[2]) for r in rows)
print("total kernel ms: %.1f" % (tot/1e6))
for r in rows[:22]:
    print("  %-80s calls %5s total %8.2f ms avg %8.1f us" % (r[0][:80], r[1], float(r[2])/1e6, float(r[3])/1e3))
PY
