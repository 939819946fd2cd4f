# in-step A/B of environment switches on ONE box: bash tools/ab_env.sh "VAR=val" "VAR2=val2" ...   (each against the default, interleaved, 2 rounds)
run() { env "$@" python bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | sed -E 's/.*"value": ([0-9.]+).*"ms_per_step": ([0-9.]+).*/\1 clips\/s \2 ms/'; }
for r in 1 2; do
  echo "default: $(run A=1)"
  for v in "$@"; do echo "$v: $(run $v)"; done
done
