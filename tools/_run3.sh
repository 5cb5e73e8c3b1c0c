export TMPDIR=/tmp
O=gpurun_out/r04c; mkdir -p $O
B="python bench.py --config E --precision bf16 --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-split-line --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $B > $O/prof.log 2>&1
for sp in "4,4,4" "3,3,3,3" "2,2,2,2,2,2" "8,4" ; do
  echo "== split $sp" >> $O/ab.log
  MPNHIP_WGRAD_SPLIT=$sp $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $O/ab.log
done
echo "== default" >> $O/ab.log
$B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $O/ab.log
echo "== no side stream" >> $O/ab.log
MPNHIP_NO_SIDE_STREAM=1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" >> $O/ab.log
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
cat $O/ab.log
