#!/bin/bash
# step_gaps.sh MODE [ENV=VAL ...] : where the GPU idles inside a step of bench.py --mode MODE -- gaps between consecutive kernels of the
# last of 3 steps, from a rocprofv3 kernel trace (GPU box).  How the validation step's 14 ms of idle GPU were found (a blocking pad-mask upload).
R=${GRAFT_REPO_ROOT:-/root/repo}
mode=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/vg
env "$@" SED_OVERLAP_TEACHER=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/vg -o p -- python $R/bench.py --mode $mode --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timer > /dev/null 2>&1
f=$(find /tmp/vg -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# last step: from the last logmel kernel on
starts = [i for i, e in enumerate(ev) if "wav_absmax_kernel" in e[2]]
# (the last step starts at the last frontend launch that is followed by a long run of kernels)
i0 = starts[-1] if len(starts) < 2 or len(ev) - starts[-1] > 100 else starts[-2]
seg = ev[max(0, i0 - 1):]
t0, t1 = seg[0][0], max(e[1] for e in seg)
busy = 0; cur_end = seg[0][0]; gaps = []
for s, e, n in seg:
    if s > cur_end:
        gaps.append((s - cur_end, prev, n))
    busy += max(0, e - max(s, cur_end))
    if e > cur_end: cur_end = e; prev = n
print("last step: %.1f ms span, %.1f ms busy, %d kernels" % ((t1 - t0) / 1e6, busy / 1e6, len(seg)))
big = sorted(gaps, reverse=True)[:14]
print("idle total %.1f ms; gaps > 50 us: %.1f ms in %d gaps" % (sum(g[0] for g in gaps) / 1e6, sum(g[0] for g in gaps if g[0] > 50000) / 1e6, sum(1 for g in gaps if g[0] > 50000)))
for g, a, b in big:
    print("  %8.1f us  after %-50s before %s" % (g / 1e3, a[:50], b[:50]))
PY
