# Host-side de-risking of the 8-rank run: one rank's step under the CPU share it will have when 8 ranks split a
# 16-CPU quota (2 CPUs), and with the RCCL gradient path forced on (SED_DDP_FORCE=1: stage-triggered all-reduces at world size 1).
#   gpurun --timeout 1500 -- 'bash tools/host_contention.sh r5'      -> gpurun_out/<tag>/host_contention.txt
TAG=${1:-r5}; O=gpurun_out/$TAG; mkdir -p $O; OUT=$O/host_contention.txt; : > $OUT
B="python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-timer"
pick() { python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
h=l.get('host',{}); g=l.get('gpu_state') or {}
print('%-44s %8.2f clips/s  %8.2f ms/step   issue %6.2f ms  cpu %6.2f ms  cpus %s  sclk %s  power %s' % (sys.argv[1], l['value'], l['ms_per_step'], h.get('issue_ms_per_step',-1), h.get('cpu_ms_per_step',-1), h.get('cpus_usable'), (g.get('sclk_MHz') or {}).get('mean'), (g.get('socket_power_W') or {}).get('mean')))
" "$1"; }
nproc >> $OUT; python -c "import os;print('affinity',len(os.sched_getaffinity(0)))" >> $OUT
for rep in 1 2; do
  $B 2>/dev/null | pick "all CPUs (rep $rep)" >> $OUT
  taskset -c 0-1 $B 2>/dev/null | pick "taskset 2 CPUs (rep $rep)" >> $OUT
  taskset -c 0 $B 2>/dev/null | pick "taskset 1 CPU (rep $rep)" >> $OUT
done
SED_DDP_FORCE=1 $B 2>/dev/null | pick "all CPUs, SED_DDP_FORCE=1" >> $OUT
SED_DDP_FORCE=1 taskset -c 0-1 $B 2>/dev/null | pick "taskset 2 CPUs, SED_DDP_FORCE=1" >> $OUT
SED_DDP_FORCE=1 SED_DDP_COMM_DTYPE=bf16 taskset -c 0-1 $B 2>/dev/null | pick "taskset 2 CPUs, DDP_FORCE=1, bf16 exchange" >> $OUT
for m in pretrain finetune1; do
  python bench.py --mode $m --steps 20 --warmup 4 --no-cpu-baseline --no-kernel-timer 2>/dev/null | pick "$m, all CPUs" >> $OUT
  taskset -c 0-1 python bench.py --mode $m --steps 20 --warmup 4 --no-cpu-baseline --no-kernel-timer 2>/dev/null | pick "$m, taskset 2 CPUs" >> $OUT
done
# the input pipeline end to end (files -> reader threads -> H2D -> device resampler -> step), at all CPUs and at one rank's share
for st in finetune2 pretrain; do
  python bench.py --mode pipe --pipe-step $st --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>$O/pipe_err_$st.txt | tail -1 > $O/pipe_$st.json
  python -c "
import json;l=json.load(open('$O/pipe_$st.json'));p=l['pipe']
print('pipe %-10s all CPUs      end-to-end %8.2f clips/s  resident %8.2f  ratio %.4f  cpu %6.2f ms/step' % ('$st', l['value'], p['resident_value'], p['ratio_vs_resident'], p['cpu_ms_per_step']))" >> $OUT
  taskset -c 0-1 python bench.py --mode pipe --pipe-step $st --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-timer 2>/dev/null | tail -1 > $O/pipe_${st}_2cpu.json
  python -c "
import json;l=json.load(open('$O/pipe_${st}_2cpu.json'));p=l['pipe']
print('pipe %-10s taskset 2 CPU end-to-end %8.2f clips/s  resident %8.2f  ratio %.4f  cpu %6.2f ms/step' % ('$st', l['value'], p['resident_value'], p['ratio_vs_resident'], p['cpu_ms_per_step']))" >> $OUT
done
cat $OUT
