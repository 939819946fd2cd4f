# ab_trees.sh DIR : interleaved same-box A/B of two checked-out trees (DIR = an older commit's tree with its own built library, this tree = new):
#   git worktree add -f _old <commit>; (cd _old && python -m transformer4sed_amd.build); gpurun -- 'bash tools/ab_trees.sh _old'
run() { (cd "$1" && python bench.py --no-cpu-baseline --steps 10 --mode "$2" 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}; print(d['value'], 'clips/s', d['ms_per_step'], 'ms  GEMM family', r.get('achieved'), 'TFLOP/s')"); }
for r in 1 2 3; do
  echo "old finetune2: $(run $1 finetune2)"
  echo "new finetune2: $(run . finetune2)"
done
for m in pretrain finetune1; do for r in 1 2; do echo "old $m: $(run $1 $m)"; echo "new $m: $(run . $m)"; done; done
