export TMPDIR=/tmp
O=gpurun_out/r04r; mkdir -p $O
python -m pytest tests/test_gpu_split.py tests/test_gpu_backward.py -x -q 2>&1 | tail -5 > $O/t.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras --no-split-line > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-170 > $O/kstats_head.txt
rm -rf $O/prof
for i in 1 2; do python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extras --no-split-line > $O/bench_$i.json 2>/dev/null; done
cat $O/t.log $O/kstats_head.txt; python - <<'PY'
import json
for f in ("bench_1","bench_2"):
    try:
        d=json.loads([l for l in open("gpurun_out/r04r/%s.json"%f) if l.startswith("{")][-1]); print(f, d["ms_per_step"], d["value"])
    except Exception as e: print(f, "ERR", e)
PY
