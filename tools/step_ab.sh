#!/bin/bash
# A/B of environment toggles on the default training bench: prints ms_per_step per variant (3 repetitions each, min)
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  best=999
  for i in 1 2 3; do
    v=$(env "$@" python $R/bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-split-line --no-roofline ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    best=$(python -c "print(min($best, $v))")
  done
  echo "$best  $*"
}
for v in "$@"; do run $v; done
